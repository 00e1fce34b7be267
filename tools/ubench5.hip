// ubench5: is "pivot lane writes its column to LDS, everybody reads it back" cheaper than
// "image read -> corrections -> publish -> broadcast read"?  One wave per workgroup, 2 waves per SIMD,
// each iteration models the pivot-column phase of one step with NR rows, followed by a dependent dot.
#include <hip/hip_runtime.h>
#include <cstdio>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
constexpr int ITERS = 4000;

template <int MODE, int NR>
__global__ void __launch_bounds__(64, 2) k5(double* out, int iters, double c)
{
    __shared__ __attribute__((aligned(16))) double lds[2 * 1264];
    const int lane = threadIdx.x, half = lane >> 5, j = lane & 31;
    double* hl = lds + half * 1264;
    for (int i = j; i < 1264; i += 32) hl[i] = 1.0 / (1 + i);
    __syncthreads();
    double a[32];
#pragma unroll
    for (int i = 0; i < 32; ++i) a[i] = (lane + 1) * 1e-3 * (i + 1);
    double d = 0.0;
    for (int it = 0; it < iters; ++it) {
        // "pivot lane" of each half depends on the previous iteration's result (serial chain like the real step)
        const int lbl = (__double2loint(d) + it) & 31;
        if (MODE == 0) {
            // current scheme: element (j, lbl) of the image + 2 corrections, publish, broadcast read
            double xi = hl[lbl * 34 + j];
            xi = fma(hl[1088 + 32 + lbl], a[0], xi);
            xi = fma(hl[1088 + 64 + lbl], a[1], xi);
            hl[1088 + j] = xi;
        } else {
            // pivot lane writes NR rows of its column (16-byte writes, 1 active lane per half)
            if (j == lbl) {
#pragma unroll
                for (int i = 32 - NR; i < 32; i += 2)
                    *reinterpret_cast<double2*>(&hl[1088 + i]) = make_double2(a[i], a[i + 1]);
            }
        }
        double x[32];
#pragma unroll
        for (int i = 32 - NR; i < 32; ++i) x[i] = hl[1088 + i];
        double s = 0.0;
#pragma unroll
        for (int i = 32 - NR; i < 32; ++i) s = fma(x[i], a[i], s);
        d = s * c;
#pragma unroll
        for (int i = 32 - NR; i < 32; ++i) a[i] = fma(d, x[i], a[i]) * 0.5;
        if (MODE == 0 && (it & 3) == 3) {   // refresh every 4th step
#pragma unroll
            for (int i = 32 - NR; i < 32; i += 2)
                *reinterpret_cast<double2*>(&hl[j * 34 + i]) = make_double2(a[i], a[i + 1]);
        }
    }
    double r = d;
#pragma unroll
    for (int i = 0; i < 32; ++i) r += a[i];
    out[blockIdx.x * 64 + lane] = r;
}

template <int MODE, int NR>
int run(const char* name)
{
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int blocks = prop.multiProcessorCount * 8;
    double* out;
    CHECK(hipMalloc(&out, (size_t)blocks * 64 * sizeof(double)));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL((k5<MODE, NR>), dim3(blocks), dim3(64), 0, 0, out, ITERS, 1e-3);
    CHECK(hipDeviceSynchronize());
    float best = 1e30f;
    for (int r = 0; r < 3; ++r) {
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL((k5<MODE, NR>), dim3(blocks), dim3(64), 0, 0, out, ITERS, 1e-3);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
    }
    printf("%-44s rows=%2d  %.1f ns per step (8 waves/CU)\n", name, NR, best * 1e6 / ITERS);
    CHECK(hipFree(out));
    return 0;
}

int main()
{
    run<0, 30>("image read + 2 corrections + publish");
    run<1, 30>("pivot lane writes its column");
    run<0, 16>("image read + 2 corrections + publish");
    run<1, 16>("pivot lane writes its column");
    run<0, 6>("image read + 2 corrections + publish");
    run<1, 6>("pivot lane writes its column");
    return 0;
}
