// Microbenchmark (gfx950): FP64 FMA with a DPP row_newbcast operand against the plain FMA, dependent-chain latency of both,
// and whether v_mfma_f64_16x16x4_f64 from one wave runs beside v_fmac_f64 from another wave of the same SIMD.
// Build: hipcc -O3 --offload-arch=gfx950 tools/ubench6.hip -o build/ubench6
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));

#define FMAC(d, x, a) asm volatile("v_fmac_f64_e32 %0, %1, %2" : "+v"(d) : "v"(x), "v"(a))
#define FMAC_DPP(d, x, a, N) asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:" #N " row_mask:0xf bank_mask:0xf" : "+v"(d) : "v"(x), "v"(a))

// MODE 0: 8 independent plain FMA chains; 1: the same with DPP row_newbcast; 2: one dependent plain chain; 3: one dependent DPP chain;
// 4: MFMA f64, 4 accumulators; 5: waves 0-3 of the workgroup plain FMA (8 chains), waves 4.. MFMA; 6: all waves FMA but waves 4.. idle early
template <int MODE>
__global__ void __launch_bounds__(1024) k(double* out, long long* cyc, int iters)
{
    const int wave = threadIdx.x >> 6;
    double x = 1.0 + threadIdx.x * 1e-3, a = 1e-9 * (threadIdx.x + 1);
    double d[8];
    for (int i = 0; i < 8; ++i) d[i] = i;
    d4 acc[4];
    for (int i = 0; i < 4; ++i) acc[i] = d4{0, 0, 0, 0};
    __syncthreads();
    const long long t0 = __builtin_amdgcn_s_memtime();
    const bool mfma_role = MODE == 4 || (MODE == 5 && wave >= 4);
    if (mfma_role) {
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(x, a, acc[i], 0, 0, 0);
        }
    } else if (MODE == 6 && wave >= 4) {
        // idle partner
    } else {
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                if (MODE == 0 || MODE == 5 || MODE == 6) {
                    FMAC(d[0], x, a); FMAC(d[1], x, a); FMAC(d[2], x, a); FMAC(d[3], x, a);
                    FMAC(d[4], x, a); FMAC(d[5], x, a); FMAC(d[6], x, a); FMAC(d[7], x, a);
                } else if (MODE == 1) {
                    FMAC_DPP(d[0], x, a, 0); FMAC_DPP(d[1], x, a, 1); FMAC_DPP(d[2], x, a, 2); FMAC_DPP(d[3], x, a, 3);
                    FMAC_DPP(d[4], x, a, 4); FMAC_DPP(d[5], x, a, 13); FMAC_DPP(d[6], x, a, 14); FMAC_DPP(d[7], x, a, 15);
                } else if (MODE == 2) {
                    FMAC(d[0], x, a); FMAC(d[0], x, a); FMAC(d[0], x, a); FMAC(d[0], x, a);
                    FMAC(d[0], x, a); FMAC(d[0], x, a); FMAC(d[0], x, a); FMAC(d[0], x, a);
                } else if (MODE == 3) {
                    FMAC_DPP(d[0], x, a, 0); FMAC_DPP(d[0], x, a, 1); FMAC_DPP(d[0], x, a, 2); FMAC_DPP(d[0], x, a, 3);
                    FMAC_DPP(d[0], x, a, 4); FMAC_DPP(d[0], x, a, 13); FMAC_DPP(d[0], x, a, 14); FMAC_DPP(d[0], x, a, 15);
                }
            }
        }
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    double s = 0;
    for (int i = 0; i < 8; ++i) s += d[i];
    for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0 && blockIdx.x == 0) cyc[wave] = t1 - t0;
}

template <int MODE> void run(const char* what, int threads, int blocks, double* out, long long* cyc)
{
    const int iters = 4000;
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(threads), 0, 0, out, cyc, iters);
    hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0); hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(threads), 0, 0, out, cyc, iters); hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    long long c[16]; hipMemcpy(c, cyc, sizeof(c), hipMemcpyDeviceToHost);
    const int nw = threads / 64;
    // per wave: FMA roles issue 64 FMAs per iteration, MFMA roles 16 MFMAs per iteration
    printf("%-58s waves/SIMD %d blocks %3d: %8.1f us | wave0 %.2f ticks/FMA-or-%.1f/MFMA", what, nw / 4, blocks, ms * 1e3,
           (double)c[0] / (iters * 64.0), (double)c[0] / (iters * 16.0));
    if (nw > 4) printf(" | wave4 %.2f ticks/FMA-or-%.1f/MFMA", (double)c[4] / (iters * 64.0), (double)c[4] / (iters * 16.0));
    printf("\n");
}

__global__ void sem(double* out)
{
    double x = threadIdx.x, a = 1.0, d = 0.0;
    asm volatile("s_nop 4\n\tv_fmac_f64_dpp %0, %1, %2 row_newbcast:5 row_mask:0xf bank_mask:0xf" : "+v"(d) : "v"(x), "v"(a));
    out[threadIdx.x] = d;
}

int main()
{
    double* out; long long* cyc; hipMalloc(&out, 256 * 1024 * 8); hipMalloc(&cyc, 16 * 8);
    hipLaunchKernelGGL(sem, dim3(1), dim3(64), 0, 0, out);
    double h[64]; hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost);
    printf("row_newbcast:5 gives lanes 0,15,16,31,32,63 -> %.0f %.0f %.0f %.0f %.0f %.0f (expect 5 5 21 21 37 53)\n", h[0], h[15], h[16], h[31], h[32], h[63]);
    for (int blocks : {1, 256}) {
        for (int threads : {256, 512}) {
            run<0>("plain v_fmac_f64, 8 chains", threads, blocks, out, cyc);
            run<1>("v_fmac_f64_dpp row_newbcast, 8 chains", threads, blocks, out, cyc);
            run<2>("plain v_fmac_f64, 1 dependent chain", threads, blocks, out, cyc);
            run<3>("v_fmac_f64_dpp row_newbcast, 1 dependent chain", threads, blocks, out, cyc);
            run<4>("v_mfma_f64_16x16x4_f64, 4 accumulators", threads, blocks, out, cyc);
        }
        run<5>("waves 0-3 plain FMA | waves 4-7 MFMA f64 (same SIMDs)", 512, blocks, out, cyc);
        run<6>("waves 0-3 plain FMA | waves 4-7 idle", 512, blocks, out, cyc);
    }
    return 0;
}
