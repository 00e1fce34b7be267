// What the HBM of this box sustains for the read : write mixes of the small-tile kernels (K5): pure read, pure write, copy, and one part
// read to W parts written (7 x 2 tiles: 112 B in, 424 B out = 1 : 3.8; 8 x 6: 384 in, 704 out = 1 : 1.8; 32 x 32: 8 192 in, 12 544 out =
// 1 : 1.5), with ordinary and with non-temporal stores.  16 bytes per lane per access, grid-stride, 2 048 workgroups of 256 lanes.
// Build: hipcc -O3 --offload-arch=gfx950 tools/ubench_stream_mix.hip -o build/ubench_stream_mix
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <algorithm>
#include <vector>
typedef double double2_t __attribute__((ext_vector_type(2)));
template <int W, bool NT, bool RD>
__global__ void __launch_bounds__(256) mix_kernel(const double2_t* __restrict__ src, double2_t* __restrict__ dst, size_t n_in)
{
    // element i of src feeds elements W i .. W i + W - 1 of dst (W = 0: read only, result folded into one store per workgroup)
    double2_t acc = {0.0, 0.0};
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n_in; i += (size_t)gridDim.x * 256) {
        double2_t v = {1.0, 2.0};
        if (RD) v = src[i];
        if (W == 0) acc += v;
#pragma unroll
        for (int w = 0; w < W; ++w) {
            // consecutive lanes write consecutive 16-byte words of block w: coalesced 4 KB runs per workgroup
            const size_t o = (i / 256) * (size_t)(256 * W) + (size_t)w * 256 + (i % 256);
            if (NT) __builtin_nontemporal_store(v, &dst[o]); else dst[o] = v;
        }
    }
    if (W == 0 && acc.x == 123.456) dst[0] = acc;
}
template <int W, bool NT, bool RD>
static void run(const char* name, const double2_t* src, double2_t* dst, size_t n_in)
{
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    std::vector<float> ms;
    for (int rep = 0; rep < 7; ++rep) {
        (void)hipEventRecord(a);
        for (int k = 0; k < 5; ++k) hipLaunchKernelGGL((mix_kernel<W, NT, RD>), dim3(2048), dim3(256), 0, 0, src, dst, n_in);
        (void)hipEventRecord(b); (void)hipEventSynchronize(b);
        float t; (void)hipEventElapsedTime(&t, a, b); ms.push_back(t / 5);
    }
    std::sort(ms.begin(), ms.end());
    const double bytes = (double)n_in * 16.0 * ((RD ? 1 : 0) + W);
    printf("%-46s %8.1f us   %7.0f GB/s = %.3f of 8 TB/s\n", name, ms[ms.size() / 2] * 1e3, bytes / (ms[ms.size() / 2] * 1e-3) / 1e9, bytes / (ms[ms.size() / 2] * 1e-3) / 8e12);
    (void)hipEventDestroy(a); (void)hipEventDestroy(b);
}
int main(int argc, char** argv)
{
    const size_t total_mb = argc > 1 ? (size_t)atoi(argv[1]) : 1024;       // bytes moved per launch, about
    double2_t *src, *dst;
    (void)hipMalloc(&src, total_mb << 20); (void)hipMalloc(&dst, total_mb << 20);
    (void)hipMemset(src, 0, total_mb << 20); (void)hipMemset(dst, 0, total_mb << 20);
    const size_t words = (total_mb << 20) / 16;
    printf("about %zu MB per launch\n", total_mb);
    run<0, false, true>("read only", src, dst, words);
    run<1, false, false>("write only", src, dst, words);
    run<1, true, false>("write only, non-temporal", src, dst, words);
    run<1, false, true>("copy (1 : 1)", src, dst, words / 2);
    run<1, true, true>("copy (1 : 1), non-temporal stores", src, dst, words / 2);
    run<2, false, true>("1 : 2 (8 x 6, 32 x 32 like)", src, dst, words / 3);
    run<2, true, true>("1 : 2, non-temporal stores", src, dst, words / 3);
    run<4, false, true>("1 : 4 (7 x 2 like)", src, dst, words / 5);
    run<4, true, true>("1 : 4, non-temporal stores", src, dst, words / 5);
    return 0;
}
