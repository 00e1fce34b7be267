// Microbenchmark (gfx950): do the FP64 vector pipe (v_fma_f64) and the FP64 matrix pipe (v_mfma_f64_16x16x4_f64) of one SIMD run side by
// side, or does one occupy the other?  (Round-4 verdict, weak 3: DESIGN.md said "shares its pipe" without a measurement.)
// One workgroup of 512 threads = 8 waves, two per SIMD (HW_ID is read back to prove which waves share a SIMD).  Modes:
//   0  every wave runs NF dependent-free FMAs                      (2 FMA waves per SIMD)
//   1  every wave runs NM dependent-free MFMAs                     (2 MFMA waves per SIMD)
//   2  waves 0..3 FMAs, waves 4..7 MFMAs                           (one of each per SIMD)
//   3  waves 0..3 FMAs, waves 4..7 idle                            (1 FMA wave per SIMD)
//   4  waves 0..3 idle, waves 4..7 MFMAs                           (1 MFMA wave per SIMD)
//   5  every wave interleaves 4 FMAs with 1 MFMA in ONE instruction stream (same counts as 0 + 1 together per wave pair)
// s_memtime ticks per wave (shader clock).  If the pipes are separate, mode 2 takes max(mode 3, mode 4); if shared, their sum.
// Build: hipcc -O3 --offload-arch=gfx950 tools/ubench_coissue.hip -o build/ubench_coissue
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));

#define FMA8 \
    asm volatile("v_fma_f64 %0, %8, %9, %0\n\tv_fma_f64 %1, %8, %9, %1\n\tv_fma_f64 %2, %8, %9, %2\n\tv_fma_f64 %3, %8, %9, %3\n\t" \
                 "v_fma_f64 %4, %8, %9, %4\n\tv_fma_f64 %5, %8, %9, %5\n\tv_fma_f64 %6, %8, %9, %6\n\tv_fma_f64 %7, %8, %9, %7" \
                 : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3), "+v"(f4), "+v"(f5), "+v"(f6), "+v"(f7) : "v"(x), "v"(y));
#define MFMA4 \
    m0 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, m0, 0, 0, 0); m1 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, m1, 0, 0, 0); \
    m2 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, m2, 0, 0, 0); m3 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, m3, 0, 0, 0);
#define MIX \
    asm volatile("v_fma_f64 %0, %8, %9, %0\n\tv_fma_f64 %1, %8, %9, %1\n\tv_fma_f64 %2, %8, %9, %2\n\tv_fma_f64 %3, %8, %9, %3" \
                 : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3), "+v"(f4), "+v"(f5), "+v"(f6), "+v"(f7) : "v"(x), "v"(y)); \
    m0 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, m0, 0, 0, 0); \
    asm volatile("v_fma_f64 %4, %8, %9, %4\n\tv_fma_f64 %5, %8, %9, %5\n\tv_fma_f64 %6, %8, %9, %6\n\tv_fma_f64 %7, %8, %9, %7" \
                 : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3), "+v"(f4), "+v"(f5), "+v"(f6), "+v"(f7) : "v"(x), "v"(y)); \
    m1 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, m1, 0, 0, 0);

__global__ void __launch_bounds__(512) k(int mode, int iters, double* out, long long* ticks, int* hwid)
{
    const int w = threadIdx.x >> 6;
    double x = 1.0 + threadIdx.x * 1e-6, y = 1e-9 * (threadIdx.x + 1);
    double f0 = 0, f1 = 1, f2 = 2, f3 = 3, f4 = 4, f5 = 5, f6 = 6, f7 = 7;
    d4 m0 = {0, 0, 0, 0}, m1 = m0, m2 = m0, m3 = m0;
    const bool do_f = mode == 0 || ((mode == 2 || mode == 3) && w < 4);
    const bool do_m = mode == 1 || ((mode == 2 || mode == 4) && w >= 4);
    __syncthreads();
    const long long t0 = __builtin_amdgcn_s_memtime();
    if (mode == 5) {
        for (int it = 0; it < iters; ++it) { MIX MIX MIX MIX }                 // 32 FMAs + 8 MFMAs per iteration
    } else if (do_f) {
        for (int it = 0; it < iters; ++it) { FMA8 FMA8 FMA8 FMA8 }              // 32 FMAs per iteration
    } else if (do_m) {
        for (int it = 0; it < iters; ++it) { MFMA4 MFMA4 }                      // 8 MFMAs per iteration
    }
    asm volatile("s_nop 7\n\ts_nop 7" ::: "memory");
    const long long t1 = __builtin_amdgcn_s_memtime();
    out[threadIdx.x] = f0 + f1 + f2 + f3 + f4 + f5 + f6 + f7 + m0[0] + m1[1] + m2[2] + m3[3] + m0[3] + m1[0];
    if ((threadIdx.x & 63) == 0) {
        ticks[w] = t1 - t0;
        hwid[w] = (int)__builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11));      // HW_REG_HW_ID, all 32 bits
    }
}

int main()
{
    double* out; long long* ticks; int* hwid;
    hipMalloc(&out, 512 * 8); hipMalloc(&ticks, 64); hipMalloc(&hwid, 32);
    const int iters = 4000;
    const char* names[6] = {"2 FMA waves / SIMD", "2 MFMA waves / SIMD", "1 FMA + 1 MFMA wave / SIMD", "1 FMA wave / SIMD", "1 MFMA wave / SIMD",
                            "2 waves / SIMD, 4 FMA : 1 MFMA interleaved in one stream"};
    for (int mode = 0; mode < 6; ++mode) {
        long long best[8]; int id[8];
        for (int w = 0; w < 8; ++w) best[w] = 1ll << 62;
        for (int rep = 0; rep < 4; ++rep) {
            hipMemset(ticks, 0, 64);
            hipLaunchKernelGGL(k, dim3(1), dim3(512), 0, 0, mode, iters, out, ticks, hwid);
            hipDeviceSynchronize();
            long long t[8]; hipMemcpy(t, ticks, 64, hipMemcpyDeviceToHost); hipMemcpy(id, hwid, 32, hipMemcpyDeviceToHost);
            for (int w = 0; w < 8; ++w) if (t[w] < best[w]) best[w] = t[w];
        }
        printf("mode %d (%s), %d iterations:\n", mode, names[mode], iters);
        for (int w = 0; w < 8; ++w) {
            const bool f = mode == 0 || mode == 5 || ((mode == 2 || mode == 3) && w < 4), m = mode == 1 || mode == 5 || ((mode == 2 || mode == 4) && w >= 4);
            printf("   wave %d  simd %d cu %d : %9lld ticks", w, (id[w] >> 4) & 3, (id[w] >> 8) & 15, best[w]);
            if (f) printf("   %.2f ticks per FMA", (double)best[w] / (32.0 * iters));
            if (m) printf("   %.2f ticks per MFMA", (double)best[w] / (8.0 * iters));
            printf("\n");
        }
    }
    return 0;
}
