// Micro-benchmarks of the instruction classes the QR kernels are built from (gfx950).
// Build: hipcc --offload-arch=gfx950 -O3 tools/ubench.hip -o build/ubench ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

constexpr int ITERS = 4096;

template <int MODE>
__global__ void __launch_bounds__(64) k_bench(double* out, int n_iter, double seed)
{
    const int lane = threadIdx.x;
    double a0 = seed + lane, a1 = a0 * 1.1, a2 = a0 * 1.2, a3 = a0 * 1.3, a4 = a0 * 1.4, a5 = a0 * 1.5, a6 = a0 * 1.6, a7 = a0 * 1.7;
    const double m = 1.0000001, c = 1e-9;
    float f0 = (float)a0, f1 = f0 * 1.1f, f2 = f0 * 1.2f, f3 = f0 * 1.3f, f4 = f0 * 1.4f, f5 = f0 * 1.5f, f6 = f0 * 1.6f, f7 = f0 * 1.7f;
    typedef double d4 __attribute__((ext_vector_type(4)));
    d4 acc0 = {0, 0, 0, 0}, acc1 = acc0, acc2 = acc0, acc3 = acc0;
    for (int i = 0; i < n_iter; ++i) {
        if (MODE == 0) {          // 8 independent v_fma_f64
            a0 = fma(a0, m, c); a1 = fma(a1, m, c); a2 = fma(a2, m, c); a3 = fma(a3, m, c);
            a4 = fma(a4, m, c); a5 = fma(a5, m, c); a6 = fma(a6, m, c); a7 = fma(a7, m, c);
        } else if (MODE == 1) {   // 8 independent v_fma_f32
            f0 = fmaf(f0, 1.0000001f, 1e-9f); f1 = fmaf(f1, 1.0000001f, 1e-9f); f2 = fmaf(f2, 1.0000001f, 1e-9f); f3 = fmaf(f3, 1.0000001f, 1e-9f);
            f4 = fmaf(f4, 1.0000001f, 1e-9f); f5 = fmaf(f5, 1.0000001f, 1e-9f); f6 = fmaf(f6, 1.0000001f, 1e-9f); f7 = fmaf(f7, 1.0000001f, 1e-9f);
        } else if (MODE == 2) {   // 8 x (2 readlane + fma with SGPR operand): the kernel's inner pattern
            const int src = i & 31;
#define RL(x) __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(x), src), __builtin_amdgcn_readlane(__double2loint(x), src))
            double s;
            s = RL(a0); a1 = fma(s, a1, c); s = RL(a1); a2 = fma(s, a2, c); s = RL(a2); a3 = fma(s, a3, c); s = RL(a3); a4 = fma(s, a4, c);
            s = RL(a4); a5 = fma(s, a5, c); s = RL(a5); a6 = fma(s, a6, c); s = RL(a6); a7 = fma(s, a7, c); s = RL(a7); a0 = fma(s, a0, c);
        } else if (MODE == 3) {   // 16 v_readlane_b32 only
            const int src = i & 31;
            int x = __double2loint(a0), y = __double2hiint(a0), acc = 0;
#pragma unroll
            for (int j = 0; j < 8; ++j) { acc += __builtin_amdgcn_readlane(x + j, src); acc ^= __builtin_amdgcn_readlane(y + j, src); }
            a0 = __hiloint2double(y, x + (acc & 1));
        } else if (MODE == 4) {   // 4 independent v_mfma_f64_16x16x4_f64
            acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, a1, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, a1, acc1, 0, 0, 0);
            acc2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, a1, acc2, 0, 0, 0);
            acc3 = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, a1, acc3, 0, 0, 0);
        } else if (MODE == 5) {   // 8 v_mul_f64
            a0 *= m; a1 *= m; a2 *= m; a3 *= m; a4 *= m; a5 *= m; a6 *= m; a7 *= m;
        } else if (MODE == 6) {   // 8 v_rcp_f64
            a0 = __builtin_amdgcn_rcp(a0); a1 = __builtin_amdgcn_rcp(a1); a2 = __builtin_amdgcn_rcp(a2); a3 = __builtin_amdgcn_rcp(a3);
            a4 = __builtin_amdgcn_rcp(a4); a5 = __builtin_amdgcn_rcp(a5); a6 = __builtin_amdgcn_rcp(a6); a7 = __builtin_amdgcn_rcp(a7);
        } else if (MODE == 7) {   // 8 dependent-free v_cndmask pairs (64-bit select)
            a0 = (lane & 1) ? a1 : a0; a2 = (lane & 2) ? a3 : a2; a4 = (lane & 4) ? a5 : a4; a6 = (lane & 8) ? a7 : a6;
            a1 = (lane & 1) ? a0 : a1; a3 = (lane & 2) ? a2 : a3; a5 = (lane & 4) ? a4 : a5; a7 = (lane & 8) ? a6 : a7;
            asm volatile("" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));
        }
    }
    double r = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + f0 + f1 + f2 + f3 + f4 + f5 + f6 + f7;
    r += acc0[0] + acc1[1] + acc2[2] + acc3[3];
    out[blockIdx.x * 64 + lane] = r;
}

template <int MODE>
int run(const char* name, double ops_per_iter, int waves_per_simd)
{
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    const int blocks = cus * 4 * waves_per_simd;
    double* out;
    CHECK(hipMalloc(&out, (size_t)blocks * 64 * sizeof(double)));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL(k_bench<MODE>, dim3(blocks), dim3(64), 0, 0, out, ITERS, 1.0);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    const int reps = 5;
    for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(k_bench<MODE>, dim3(blocks), dim3(64), 0, 0, out, ITERS, 1.0);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    ms /= reps;
    // cycles per wave-instruction per SIMD at an assumed 2.4 GHz: time * f / (instr per SIMD)
    const double instr_per_simd = (double)ITERS * ops_per_iter * waves_per_simd;
    const double cyc = ms * 1e-3 * 2.4e9 / instr_per_simd;
    printf("%-34s waves/SIMD=%d  %.3f ms  -> %.2f cycles per wave-instruction per SIMD (@2.4GHz)\n", name, waves_per_simd, ms, cyc);
    CHECK(hipFree(out));
    return 0;
}

int main()
{
    for (int w : {1, 2, 4}) {
        run<0>("v_fma_f64 x8", 8, w);
        run<1>("v_fma_f32 x8", 8, w);
        run<5>("v_mul_f64 x8", 8, w);
        run<2>("(2 readlane + fma_f64 sgpr) x8", 8, w);
        run<3>("v_readlane_b32 x16 (+16 salu)", 16, w);
        run<4>("v_mfma_f64_16x16x4 x4", 4, w);
        run<6>("v_rcp_f64 x8", 8, w);
        run<7>("v_cndmask 64-bit select x8", 8, w);
    }
    return 0;
}
