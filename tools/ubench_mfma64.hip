// Microbenchmark: issue rate of v_mfma_f64_16x16x4_f64 on gfx950 (cycles per instruction per SIMD) with 1, 2, 4 waves
// per SIMD and 1..4 independent accumulators per wave.  Build: hipcc -O3 --offload-arch=gfx950 tools/ubench_mfma64.hip -o build/ubench_mfma64
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));
template <int NACC>
__global__ void __launch_bounds__(1024) k(double* out, long long* cyc, int iters)
{
    d4 acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = d4{0, 0, 0, 0};
    double a = threadIdx.x * 0.001, b = 1.0 + threadIdx.x * 0.002;
    __syncthreads();
    long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
    }
    long long t1 = __builtin_amdgcn_s_memtime();
    double s = 0; for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[0] = t1 - t0;
}
template <int NACC> void run(int threads, double* out, long long* cyc)
{
    const int iters = 2000;
    hipLaunchKernelGGL(k<NACC>, dim3(1), dim3(threads), 0, 0, out, cyc, iters);
    hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0); hipLaunchKernelGGL(k<NACC>, dim3(1), dim3(threads), 0, 0, out, cyc, iters); hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    const double n_per_simd = (double)iters * NACC * (threads / 64) / 4.0;
    printf("waves %2d acc %d: %.1f us, %.1f ns per MFMA per SIMD (%lld memtime ticks)\n", threads / 64, NACC, ms * 1e3, ms * 1e6 / n_per_simd, c);
}
int main()
{
    double* out; long long* cyc; hipMalloc(&out, 1024 * 8); hipMalloc(&cyc, 8);
    for (int threads : {256, 512, 1024}) { run<1>(threads, out, cyc); run<2>(threads, out, cyc); run<4>(threads, out, cyc); }
    return 0;
}
