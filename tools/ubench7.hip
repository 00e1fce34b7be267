// Microbenchmark (gfx950): is a long straight-line kernel (code larger than the instruction cache) bound by instruction fetch?
// The same number of independent FP64 FMAs (8 chains, DPP row_newbcast operand) per wave as (A) a loop whose body is 64 instructions
// and (B) one straight line of 12 288 instructions (96 KB of code) run ITERS/192 times; 4 096 one-wave workgroups (four waves per SIMD)
// and 1 024 (one per SIMD).  Build: hipcc -O3 --offload-arch=gfx950 tools/ubench7.hip -o build/ubench7
#include <hip/hip_runtime.h>
#include <cstdio>
#define F(d, N) asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:" #N " row_mask:0xf bank_mask:0xf" : "+v"(d) : "v"(x), "v"(a));
#define F8 F(d0, 0) F(d1, 1) F(d2, 2) F(d3, 3) F(d4, 4) F(d5, 13) F(d6, 14) F(d7, 15)
#define F64 F8 F8 F8 F8 F8 F8 F8 F8
#define F512 F64 F64 F64 F64 F64 F64 F64 F64
#define F4096 F512 F512 F512 F512 F512 F512 F512 F512
#define F12288 F4096 F4096 F4096

template <bool LONG>
__global__ void __launch_bounds__(64, 4) k(double* out, int iters)
{
    double x = 1.0 + threadIdx.x * 1e-3, a = 1e-9 * (threadIdx.x + 1);
    double d0 = 0, d1 = 1, d2 = 2, d3 = 3, d4 = 4, d5 = 5, d6 = 6, d7 = 7;
    if (LONG) {
        for (int it = 0; it < iters / 192; ++it) { F12288 }
    } else {
        for (int it = 0; it < iters; ++it) { F64 }
    }
    out[blockIdx.x * 64 + threadIdx.x] = d0 + d1 + d2 + d3 + d4 + d5 + d6 + d7;
}

int main()
{
    double* out;
    hipMalloc(&out, 4096 * 64 * sizeof(double));
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 192 * 4;                     // 49 152 FMAs per wave
    for (int grid : {1024, 2048, 4096}) {
        for (int mode = 0; mode < 2; ++mode) {
            float best = 1e9f;
            for (int rep = 0; rep < 5; ++rep) {
                hipEventRecord(e0);
                if (mode) hipLaunchKernelGGL(k<true>, dim3(grid), dim3(64), 0, 0, out, iters);
                else hipLaunchKernelGGL(k<false>, dim3(grid), dim3(64), 0, 0, out, iters);
                hipEventRecord(e1);
                hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1);
                if (ms < best) best = ms;
            }
            const double fmas = (double)grid * iters * 64;
            printf("grid %4d (%d waves/SIMD) %-13s %8.2f us   %.2f ns per FMA instruction per SIMD\n", grid, grid / 1024,
                   mode ? "straight-line" : "loop", best * 1e3, best * 1e6 / (fmas / 1024));
        }
    }
    return 0;
}
