// Diagnostic: the panel kernel of caqr.hip alone (hipEvents + s_memtime stamps per phase).  hipcc -O3 --offload-arch=gfx950 -DQRK_CAQR_STAMP
#include "../qrkit_amd/csrc/caqr.hip"
#include <cstdio>
#include <vector>
int main()
{
    using namespace qrk; using namespace qrk::caqr;
    const int m = 40000, S = (m / 32 + FAN - 1) / FAN;
    std::vector<double> h((size_t)m * 32);
    for (size_t i = 0; i < h.size(); ++i) h[i] = 0.5 + (double)((i * 2654435761u) % 1000) / 250.0;
    double *A, *T;
    hipMalloc(&A, h.size() * 8); hipMalloc(&T, (size_t)S * 1024 * 8);
    hipFuncSetAttribute(reinterpret_cast<const void*>(caqr_panel_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)PANEL_LDS);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipFuncSetAttribute(reinterpret_cast<const void*>(caqr_panel_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)PANEL_LDS);
    // grid 1 / S: level 0 (dense slabs); -20 / -3 / -1: the upper levels of the tree (stacks of triangles, chunk stride 8 / 64 / 512)
    for (int grid : {1, S, -20, -3, -1}) {
        for (int it = 0; it < 3; ++it) {
            hipMemcpy(A, h.data(), h.size() * 8, hipMemcpyHostToDevice);
            hipEventRecord(e0);
            if (grid > 0) hipLaunchKernelGGL((caqr_panel_kernel<false>), dim3(grid), dim3(256), PANEL_LDS, 0, A, (int64_t)m, m, 0, 32, Slab{0, 1, m / 32}, T);
            else {
                const int stride = grid == -20 ? 8 : (grid == -3 ? 64 : 512);
                const int nch = (m / 32 + stride - 1) / stride;
                hipLaunchKernelGGL((caqr_panel_kernel<true>), dim3(-grid), dim3(256), PANEL_LDS, 0, A, (int64_t)m, m, 0, 32, Slab{0, stride, nch}, T);
            }
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            std::vector<double> t(1024);
            hipMemcpy(t.data(), T, 1024 * 8, hipMemcpyDeviceToHost);
            unsigned long long st[7];
            for (int q = 0; q < 7; ++q) st[q] = reinterpret_cast<unsigned long long*>(t.data())[32 * (q + 1)];
            printf("grid %4d: %8.1f us   stamps (100 MHz ticks): load %llu  steps0-7 %llu  8-15 %llu  16-23 %llu  24-31 %llu  T %llu  store %llu\n", grid,
                   ms * 1e3, st[0] ? 0ull : 0ull, st[1] - st[0], st[2] - st[1], st[3] - st[2], st[4] - st[3], st[5] - st[4], st[6] - st[5]);
            unsigned long long f[7];
            for (int q = 0; q < 7; ++q) f[q] = reinterpret_cast<unsigned long long*>(t.data())[32 * (q + 9)];
            printf("            step 16: barrier A %llu  v read + dots + red write %llu  barrier B %llu  sums + scalars %llu  update %llu  vb write %llu\n",
                   f[1] - f[0], f[2] - f[1], f[3] - f[2], f[4] - f[3], f[5] - f[4], f[6] - f[5]);
        }
    }
    return 0;
}
