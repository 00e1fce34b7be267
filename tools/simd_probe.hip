// Which SIMD do the waves of small workgroups land on?  (placement probe for a producer / consumer split of K1: 128-thread workgroups
// with 20 KB of LDS, eight per CU).  Build: hipcc --offload-arch=gfx950 -O2 tools/simd_probe.hip -o tools/abl/simd_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <map>
#include <vector>
__global__ void __launch_bounds__(128) probe(unsigned* out)
{
    __shared__ double pad[2532];        // 20256 B as the K1 workgroups
    pad[threadIdx.x] = 0.0;
    unsigned hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    // keep the workgroup resident for a while so that all 2048 are placed together
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    while (__builtin_amdgcn_s_memtime() - t0 < 200000) { }
    if ((threadIdx.x & 63) == 0) { out[2 * (blockIdx.x * 2 + (threadIdx.x >> 6))] = hw; out[2 * (blockIdx.x * 2 + (threadIdx.x >> 6)) + 1] = xcc; }
    if (pad[threadIdx.x] != 0.0) out[0] = 0;
}
int main()
{
    const int nwg = 2048;
    unsigned* d; hipMalloc(&d, nwg * 4 * sizeof(unsigned));
    hipLaunchKernelGGL(probe, dim3(nwg), dim3(128), 0, 0, d);
    std::vector<unsigned> h(nwg * 4);
    hipMemcpy(h.data(), d, h.size() * 4, hipMemcpyDeviceToHost);
    // HW_ID (gfx9): wave[3:0] simd[5:4] pipe[7:6] cu[11:8] sh[12] se[15:13]
    int pairs[4][4] = {};
    std::map<unsigned, std::vector<int> > per_cu_a, per_cu_b;
    for (int b = 0; b < nwg; ++b) {
        const unsigned h0 = h[4 * b], x0 = h[4 * b + 1], h1 = h[4 * b + 2];
        const int s0 = (h0 >> 4) & 3, s1 = (h1 >> 4) & 3;
        pairs[s0][s1]++;
        const unsigned cu = (x0 << 16) | (((h0 >> 13) & 7) << 8) | (((h0 >> 12) & 1) << 4) | ((h0 >> 8) & 15);
        per_cu_a[cu].push_back(s0); per_cu_b[cu].push_back(s1);
    }
    std::printf("(simd of wave 0, simd of wave 1) counts over %d workgroups:\n", nwg);
    for (int a = 0; a < 4; ++a) std::printf("  wave0 on simd %d: wave1 on 0/1/2/3 = %d %d %d %d\n", a, pairs[a][0], pairs[a][1], pairs[a][2], pairs[a][3]);
    int hist[9] = {}, cus = 0;
    for (auto& kv : per_cu_a) {
        int c[4] = {0, 0, 0, 0};
        for (int s : kv.second) c[s]++;
        int mx = 0; for (int s = 0; s < 4; ++s) mx = c[s] > mx ? c[s] : mx;
        hist[mx > 8 ? 8 : mx]++; ++cus;
        if (cus <= 6) { std::printf("  CU %06x: %zu workgroups, wave-0 simds:", kv.first, kv.second.size()); for (int s : kv.second) std::printf(" %d", s); std::printf("  wave-1 simds:"); for (int s : per_cu_b[kv.first]) std::printf(" %d", s); std::printf("\n"); }
    }
    std::printf("%d CUs; histogram of the largest number of wave-0's on one SIMD of a CU (ideal 2): ", cus);
    for (int i = 0; i <= 8; ++i) std::printf("%d:%d ", i, hist[i]);
    std::printf("\n");
    return 0;
}
