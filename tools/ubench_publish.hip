// Microbenchmark (gfx950), round 6: what the publication of a pivot column costs a wave.  The pattern of bdqr_pair4 / bdqr_quad32: NST
// ds_write_b64 with a few active lanes back to back, then a ds_read_b64 that needs them (the chunk read) and s_waitcnt -- s_memtime around the
// lot, per wave, for 1 .. 16 one-wave workgroups per CU (the grid decides the occupancy: 256 CUs x WPC).  Variants: the stores alone (0); two
// independent VALU instructions before every store (1: the select-then-store form); NFMA independent FMAs issued AFTER the stores and before the
// read (2: what a look-ahead would hide behind the stores); stores of 16 bytes, half as many (3).
// Build: hipcc -O3 --offload-arch=gfx950 tools/ubench_publish.hip -o build/ubench_publish ; run: build/ubench_publish
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>

typedef __attribute__((address_space(3))) double lds_f64;

template <int MODE, int NST, int NFMA>
__global__ void __launch_bounds__(64) k(unsigned long long* out, int iters, int active_mask_sel)
{
    __shared__ __attribute__((aligned(16))) double sm[2560];
    const int lane = threadIdx.x;
    double a[32], acc[4] = {1.0, 2.0, 3.0, 4.0}, x = 1.0 + 1e-9 * lane, y = 1e-12 * (lane + 1);
    for (int i = 0; i < 32; ++i) a[i] = 1.0 + i + lane;
    // four active lanes (one per row of 16), like the pivot lanes of four tiles
    const bool active = active_mask_sel ? true : ((lane & 15) == ((lane >> 4) * 5 + 3) % 16);
    double* base = sm + (lane >> 4) * 640;
    unsigned long long m = 0x5555555555555555ull;
    asm volatile("" : "+s"(m));
    unsigned long long total = 0;
    for (int it = 0; it < iters; ++it) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        const unsigned long long t0 = __builtin_amdgcn_s_memtime();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (active) {
#pragma unroll
            for (int i = 0; i < NST; ++i) {
                if (MODE == 1) {
                    int lo = __double2loint(a[i & 31]), hi = __double2hiint(a[i & 31]);
                    asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(lo) : "v"(lane), "s"(m));
                    asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(hi) : "v"(lane), "s"(m));
                    *(volatile lds_f64*)(base + i) = __hiloint2double(hi, lo);
                } else if (MODE == 3) {
                    typedef double d2a __attribute__((ext_vector_type(2), aligned(16)));
                    if ((i & 1) == 0) *(volatile __attribute__((address_space(3))) d2a*)(base + i) = d2a{a[i & 31], a[(i + 1) & 31]};
                } else {
                    *(volatile lds_f64*)(base + i) = a[i & 31];
                }
            }
        }
        if (MODE == 2) {
#pragma unroll
            for (int i = 0; i < NFMA; ++i) asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(acc[i & 3]) : "v"(x), "v"(y));
        }
        const double r = *(volatile lds_f64*)(base + (lane & 15));
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        const unsigned long long t1 = __builtin_amdgcn_s_memtime();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        total += t1 - t0;
        acc[0] += r * 1e-300;
        // a block of FMAs between publications, so that the waves of a CU are not all publishing all the time (250 cycles of issue)
#pragma unroll
        for (int i = 0; i < 64; ++i) asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(acc[i & 3]) : "v"(x), "v"(y));
    }
    if (lane == 0) out[blockIdx.x] = total + (unsigned long long)(acc[0] + acc[1] + acc[2] + acc[3] == 12345.0);
}

template <int MODE, int NST, int NFMA>
static void run(const char* name, int all_lanes)
{
    unsigned long long* d;
    const int iters = 200;
    hipMalloc(&d, 8192 * sizeof(unsigned long long));
    std::printf("%-58s", name);
    for (int wpc : {1, 2, 4, 8, 16}) {
        const int grid = 256 * wpc;
        hipLaunchKernelGGL((k<MODE, NST, NFMA>), dim3(grid), dim3(64), 0, 0, d, iters, all_lanes);
        hipDeviceSynchronize();
        hipLaunchKernelGGL((k<MODE, NST, NFMA>), dim3(grid), dim3(64), 0, 0, d, iters, all_lanes);
        hipDeviceSynchronize();
        std::vector<unsigned long long> h(grid);
        hipMemcpy(h.data(), d, grid * sizeof(unsigned long long), hipMemcpyDeviceToHost);
        std::sort(h.begin(), h.end());
        std::printf("  %2d/CU: %6.0f", wpc, (double)h[grid / 2] / iters);
    }
    std::printf("   cycles per publication (median wave)\n");
    hipFree(d);
}

int main()
{
    std::printf("# s_memtime ticks from before the first store to after the dependent read, per wave; workgroups (= waves) per CU across the line\n");
    run<0, 0, 0>("no store, the read alone", 0);
    run<0, 11, 0>("11 ds_write_b64, 4 lanes active", 0);
    run<0, 22, 0>("22 ds_write_b64, 4 lanes active", 0);
    run<0, 44, 0>("44 ds_write_b64, 4 lanes active", 0);
    run<0, 22, 0>("22 ds_write_b64, 64 lanes active", 1);
    run<3, 22, 0>("11 ds_write_b128 (the same 22 doubles), 4 lanes", 0);
    run<1, 22, 0>("22 x (2 v_cndmask + ds_write_b64), 4 lanes", 0);
    run<2, 22, 32>("22 ds_write_b64 + 32 FMAs before the read", 0);
    run<2, 22, 64>("22 ds_write_b64 + 64 FMAs before the read", 0);
    run<2, 44, 64>("44 ds_write_b64 + 64 FMAs before the read", 0);
    run<2, 44, 128>("44 ds_write_b64 + 128 FMAs before the read", 0);
    run<2, 0, 64>("64 FMAs + the read (no store)", 0);
    return 0;
}
