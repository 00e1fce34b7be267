// Diagnostic for K1: would a producer/consumer split pay?  The A-only build of the pair kernel (-DQRK_ABL=2048: no Q FMAs, no Q stores,
// 119 VGPRs, 8 workgroups per CU by LDS) next to a synthetic kernel that does the Q work of the same tiles (per pair and step: 2 (31 - k)
// dot FMAs and 2 (31 - k) update FMAs with a DPP row_newbcast operand, a few scalar operations, 16-byte stores of the finished entries
// every 8 steps; no LDS, <= 128 VGPRs) on a second stream.  If the two together take about as long as the A-only kernel alone, the
// Q work fits into the bubbles of the dependent chain and a two-wave workgroup (A-wave publishes reflectors, Q-wave applies them) is
// worth building.  hipcc -O3 --offload-arch=gfx950 -DQRK_ABL=2048 -Iinclude -Iqrkit_amd/csrc tools/duo_probe.hip -o tools/abl/duo_probe
#include "../qrkit_amd/csrc/bdqr_pair.hip"

#include <chrono>
#include <cstdio>
#include <vector>

namespace {

template <int N>
__device__ __forceinline__ void fmac_bcast(double& d, double X, double c)
{
    asm("v_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(d) : "v"(X), "v"(c), "n"(N));
}

template <int K>
__device__ __forceinline__ void qstep(double (&q)[32], double xa, double xb, double s, double ng)
{
    double d0 = 0.0, d1 = 0.0;
#define QD(I) if ((I) > K) fmac_bcast<((I) & 15)>(((I) & 1) ? d1 : d0, (I) < 16 ? xa : xb, q[I]);
    QRK_0_31(QD)
#undef QD
    const double ngq = fma(s, q[K], d0 + d1) * ng;
    q[K] = fma(s, ngq, q[K]);
#define QU(I) if ((I) > K) fmac_bcast<((I) & 15)>(q[I], (I) < 16 ? xa : xb, ngq);
    QRK_0_31(QU)
#undef QU
}

__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(4)))
qsim_kernel(int64_t npairs, const double* __restrict__ xs, double* __restrict__ q_vals)
{
    using namespace qrk::pair;
    const int lane = threadIdx.x;
    for (int64_t pi = blockIdx.x; pi < npairs; pi += gridDim.x) {
        double q[32];
#pragma unroll
        for (int i = 0; i < 32; ++i) q[i] = (i == (lane & 31)) ? 1.0 : 0.0;
        const double* xp = xs + ((pi & 63) * 32) * 64 + lane;       // a stand-in for the published reflectors (L2-resident)
#define QS(K)                                                                                                          \
        {                                                                                                              \
            const double xa = xp[(K) * 64], xb = xa * 0.75;                                                            \
            qstep<K>(q, xa, xb, 1.0 + 0.001 * xa, -0.01);                                                              \
            if (((K) & 7) == 7) store_q_half<(K) - 7, 8>(threadIdx.x, pi, 2 * npairs, q, q_vals);                      \
        }
        QRK_0_31(QS)
#undef QS
    }
}

}  // namespace

int main()
{
    const int64_t B = 10000, npairs = B / 2;
    std::vector<double> h((size_t)B * 1024);
    for (size_t i = 0; i < h.size(); ++i) h[i] = 0.5 + (double)((i * 2654435761u) % 4500) / 1000.0;
    double *tiles, *qv, *qv2, *rv, *xs;
    int32_t* perm;
    hipMalloc(&tiles, h.size() * 8); hipMalloc(&qv, h.size() * 8); hipMalloc(&qv2, h.size() * 8); hipMalloc(&rv, (size_t)B * 528 * 8);
    hipMalloc(&perm, (size_t)B * 32 * 4); hipMalloc(&xs, 64 * 32 * 64 * 8);
    hipMemcpy(tiles, h.data(), h.size() * 8, hipMemcpyHostToDevice);
    hipMemcpy(xs, h.data(), 64 * 32 * 64 * 8, hipMemcpyHostToDevice);
    hipStream_t sa, sb;
    hipStreamCreateWithFlags(&sa, hipStreamNonBlocking); hipStreamCreateWithFlags(&sb, hipStreamNonBlocking);
    qrk::WaveBatch nb{};
    nb.num_tiles = B; nb.rows = 32; nb.cols = 32; nb.pivoting = 1;
    auto runA = [&] { qrk::launch_bdqr_pair(nb, true, tiles, qv, rv, perm, nullptr, 2048, nullptr, nullptr, sa); };
    auto runQ = [&](int grid) { hipLaunchKernelGGL(qsim_kernel, dim3(grid), dim3(64), 0, sb, npairs, xs, qv2); };
    auto timeit = [&](const char* name, bool a, bool q, int grid) {
        for (int w = 0; w < 3; ++w) { if (a) runA(); if (q) runQ(grid); }
        hipDeviceSynchronize();
        const int it = 50;
        const auto t0 = std::chrono::steady_clock::now();
        for (int i = 0; i < it; ++i) { if (a) runA(); if (q) runQ(grid); }
        hipDeviceSynchronize();
        const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / it;
        std::printf("%-44s %8.1f us per 10 000 tiles\n", name, us);
    };
    timeit("A-only pair kernel alone", true, false, 0);
    timeit("synthetic Q work alone (2048 waves)", false, true, 2048);
    timeit("synthetic Q work alone (1024 waves)", false, true, 1024);
    timeit("both, two streams (Q: 2048 waves)", true, true, 2048);
    timeit("both, two streams (Q: 1024 waves)", true, true, 1024);
    return 0;
}
