// Micro-benchmarks, part 2: broadcast primitives (gfx950).
#include <hip/hip_runtime.h>
#include <cstdio>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
constexpr int ITERS = 8192;

__device__ __forceinline__ double rl64(double v, int l) {
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), l), __builtin_amdgcn_readlane(__double2loint(v), l));
}
__device__ __forceinline__ double bp64(double v, int addr) {
    return __hiloint2double(__builtin_amdgcn_ds_bpermute(addr, __double2hiint(v)), __builtin_amdgcn_ds_bpermute(addr, __double2loint(v)));
}

template <int MODE>
__global__ void __launch_bounds__(64) k2(double* out, int n_iter, double seed)
{
    const int lane = threadIdx.x;
    double a[8], d = 0.0;
#pragma unroll
    for (int j = 0; j < 8; ++j) a[j] = seed + lane * 0.001 + j;
    const double c = 1e-9;
    __shared__ double sm[64];
    sm[lane] = seed;
    __syncthreads();
    for (int i = 0; i < n_iter; ++i) {
        const int src = i & 31;
        if (MODE == 0) {   // batch: 16 readlanes first, then 8 dot FMAs, then 8 update FMAs (as in the QR step)
            double x[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) x[j] = rl64(a[j], src);
#pragma unroll
            for (int j = 0; j < 8; ++j) d = fma(x[j], a[j], d);
            const double nc = d * c;
#pragma unroll
            for (int j = 0; j < 8; ++j) a[j] = fma(nc, x[j], a[j]);
        } else if (MODE == 1) {   // same with ds_bpermute broadcast into VGPRs
            double x[8];
            const int addr = src * 4;
#pragma unroll
            for (int j = 0; j < 8; ++j) x[j] = bp64(a[j], addr);
#pragma unroll
            for (int j = 0; j < 8; ++j) d = fma(x[j], a[j], d);
            const double nc = d * c;
#pragma unroll
            for (int j = 0; j < 8; ++j) a[j] = fma(nc, x[j], a[j]);
        } else if (MODE == 2) {   // FMAs only (16 per iteration), no broadcast
#pragma unroll
            for (int j = 0; j < 8; ++j) d = fma(a[j], a[j], d);
            const double nc = d * c;
#pragma unroll
            for (int j = 0; j < 8; ++j) a[j] = fma(nc, c, a[j]);
        } else if (MODE == 3) {   // FMAs only, upper 32 lanes masked off
            if (lane < 32) {
#pragma unroll
                for (int j = 0; j < 8; ++j) d = fma(a[j], a[j], d);
                const double nc = d * c;
#pragma unroll
                for (int j = 0; j < 8; ++j) a[j] = fma(nc, c, a[j]);
            }
        } else if (MODE == 4) {   // LDS broadcast read (same address all lanes) b128 x4 = 8 doubles, + 16 FMAs
            const double2* p = reinterpret_cast<const double2*>(&sm[(i & 3) * 8]);
            double x[8];
#pragma unroll
            for (int j = 0; j < 4; ++j) { double2 v = p[j]; x[2 * j] = v.x; x[2 * j + 1] = v.y; }
#pragma unroll
            for (int j = 0; j < 8; ++j) d = fma(x[j], a[j], d);
            const double nc = d * c;
#pragma unroll
            for (int j = 0; j < 8; ++j) a[j] = fma(nc, x[j], a[j]);
        } else if (MODE == 5) {   // one-lane LDS writes (8 x ds_write_b64, exec = 1 lane) + 16 FMAs
            if (lane == src) {
#pragma unroll
                for (int j = 0; j < 8; ++j) sm[j + (i & 7) * 8] = a[j];
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) d = fma(a[j], a[j], d);
            const double nc = d * c;
#pragma unroll
            for (int j = 0; j < 8; ++j) a[j] = fma(nc, c, a[j]);
        } else if (MODE == 6) {   // half via readlane (4), half via bpermute (4)
            double x[8];
            const int addr = src * 4;
#pragma unroll
            for (int j = 0; j < 4; ++j) x[j] = rl64(a[j], src);
#pragma unroll
            for (int j = 4; j < 8; ++j) x[j] = bp64(a[j], addr);
#pragma unroll
            for (int j = 0; j < 8; ++j) d = fma(x[j], a[j], d);
            const double nc = d * c;
#pragma unroll
            for (int j = 0; j < 8; ++j) a[j] = fma(nc, x[j], a[j]);
        }
    }
    double r = d;
#pragma unroll
    for (int j = 0; j < 8; ++j) r += a[j];
    out[blockIdx.x * 64 + lane] = r + sm[lane];
}

template <int MODE>
int run(const char* name, int w)
{
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int blocks = prop.multiProcessorCount * 4 * w;
    double* out;
    CHECK(hipMalloc(&out, (size_t)blocks * 64 * sizeof(double)));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL(k2<MODE>, dim3(blocks), dim3(64), 0, 0, out, ITERS, 1.0);
    CHECK(hipDeviceSynchronize());
    float best = 1e30f;
    for (int r = 0; r < 5; ++r) {
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL(k2<MODE>, dim3(blocks), dim3(64), 0, 0, out, ITERS, 1.0);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
    }
    printf("%-52s w/SIMD=%d  %.3f ms  -> %.1f ns per iteration per SIMD\n", name, w, best, best * 1e6 / ((double)ITERS * w));
    CHECK(hipFree(out));
    return 0;
}

int main()
{
    for (int w : {2, 4}) {
        run<2>("16 fma_f64", w);
        run<3>("16 fma_f64, lanes 32-63 masked off", w);
        run<0>("16 readlane (8 doubles) + 16 fma_f64", w);
        run<1>("16 ds_bpermute (8 doubles) + 16 fma_f64", w);
        run<6>("8 readlane + 8 bpermute + 16 fma_f64", w);
        run<4>("4 ds_read_b128 broadcast + 16 fma_f64", w);
        run<5>("8 one-lane ds_write_b64 + 16 fma_f64", w);
    }
    return 0;
}
