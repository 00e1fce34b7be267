// Dependent-chain latencies (one wave per SIMD at most): v_fma_f64, readlane->use, LDS write->broadcast read.
#include <hip/hip_runtime.h>
#include <cstdio>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
constexpr int ITERS = 20000;
template <int MODE>
__global__ void __launch_bounds__(64) k4(double* out, unsigned long long* cyc, double seed)
{
    __shared__ double sm[128];
    const int lane = threadIdx.x;
    double a = seed + lane * 1e-3, b = 1.0000001, c = 1e-9;
    sm[lane] = a; sm[64 + lane] = a;
    __syncthreads();
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < ITERS; ++i) {
        if (MODE == 0) { a = fma(a, b, c); a = fma(a, b, c); a = fma(a, b, c); a = fma(a, b, c); }              // 4 dependent fma_f64
        else if (MODE == 1) {                                                                                 // readlane -> fma dependent x2
            double s = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(a), 5), __builtin_amdgcn_readlane(__double2loint(a), 5));
            a = fma(s, b, c);
            s = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(a), 7), __builtin_amdgcn_readlane(__double2loint(a), 7));
            a = fma(s, b, c);
        } else if (MODE == 2) {                                                                               // LDS write -> broadcast read round trip x2
            sm[lane] = a; a = sm[3] * b + c;
            sm[64 + lane] = a; a = sm[64 + 9] * b + c;
        } else if (MODE == 3) {                                                                               // 4 dependent v_mul_f64
            a = a * b; a = a * b; a = a * b; a = a * b;
        } else if (MODE == 4) {                                                                               // rsq + rcp dependent
            a = __builtin_amdgcn_rsq(a + 2.0); a = __builtin_amdgcn_rcp(a + 2.0);
        } else if (MODE == 5) {                                                                               // 4 dependent DPP max (asm)
            int v = __double2loint(a), r;
            asm volatile("s_nop 1\n\tv_max_i32_dpp %0, %1, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "=v"(r) : "v"(v)); v = r;
            asm volatile("s_nop 1\n\tv_max_i32_dpp %0, %1, %1 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf" : "=v"(r) : "v"(v)); v = r;
            asm volatile("s_nop 1\n\tv_max_i32_dpp %0, %1, %1 row_half_mirror row_mask:0xf bank_mask:0xf" : "=v"(r) : "v"(v)); v = r;
            asm volatile("s_nop 1\n\tv_max_i32_dpp %0, %1, %1 row_mirror row_mask:0xf bank_mask:0xf" : "=v"(r) : "v"(v)); v = r;
            a = __hiloint2double(__double2hiint(a), v);
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * 64 + lane] = a;
    if (lane == 0 && blockIdx.x == 0) cyc[MODE] = t1 - t0;
}
int main()
{
    double* out; unsigned long long* cyc;
    CHECK(hipMalloc(&out, 1024 * 64 * 8)); CHECK(hipMalloc(&cyc, 64));
    CHECK(hipMemset(cyc, 0, 64));
    const char* names[6] = {"4 dependent v_fma_f64", "2 x (2 readlane -> fma)", "2 x (LDS write -> broadcast read -> fma)", "4 dependent v_mul_f64", "rsq -> add -> rcp -> add", "4 dependent v_max_i32_dpp (+s_nop 1)"};
    const int per[6] = {4, 2, 2, 4, 1, 4};
    hipLaunchKernelGGL(k4<0>, dim3(1), dim3(64), 0, 0, out, cyc, 1.0);
    hipLaunchKernelGGL(k4<1>, dim3(1), dim3(64), 0, 0, out, cyc, 1.0);
    hipLaunchKernelGGL(k4<2>, dim3(1), dim3(64), 0, 0, out, cyc, 1.0);
    hipLaunchKernelGGL(k4<3>, dim3(1), dim3(64), 0, 0, out, cyc, 1.0);
    hipLaunchKernelGGL(k4<4>, dim3(1), dim3(64), 0, 0, out, cyc, 1.0);
    hipLaunchKernelGGL(k4<5>, dim3(1), dim3(64), 0, 0, out, cyc, 1.0);
    CHECK(hipDeviceSynchronize());
    unsigned long long h[8];
    CHECK(hipMemcpy(h, cyc, 64, hipMemcpyDeviceToHost));
    for (int m = 0; m < 6; ++m) printf("%-44s %8.1f s_memtime ticks per item (one wave alone)\n", names[m], (double)h[m] / ITERS / per[m]);
    return 0;
}
