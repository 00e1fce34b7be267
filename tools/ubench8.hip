// Microbenchmark (gfx950): issue cost of the instruction kinds of a bdqr_pair4 step with FOUR waves per SIMD (the kernel's occupancy),
// relative to v_fma_f64 -- which of the "45 other VALU instructions per step" are worth more than one slot, and what a single-lane
// ds_write_b128 costs beside them.  Every kind: a loop of 32 independent instructions, 4 096 one-wave workgroups, HIP events.
// Build: hipcc -O3 --offload-arch=gfx950 tools/ubench8.hip -o build/ubench8
#include <hip/hip_runtime.h>
#include <cstdio>

#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)
#define REP32(X) REP8(X) REP8(X) REP8(X) REP8(X)

template <int MODE>
__global__ void __launch_bounds__(64, 4) k(double* out, int iters, int one_lane)
{
    __shared__ __attribute__((aligned(16))) double sm[1280];
    const int lane = threadIdx.x;
    double d[8], x = 1.0 + lane * 1e-3, y = 1e-9 * (lane + 1);
    float f[8];
    int n[8];
    for (int i = 0; i < 8; ++i) { d[i] = 1.5 + i + lane; f[i] = 1.5f + i + lane; n[i] = i * 77 + lane; }
    sm[lane] = x; sm[64 + lane] = y;
    __syncthreads();
    unsigned long long m = 0x5555555555555555ull;
    asm volatile("" : "+s"(m));
    double* wp = sm + 128 + 2 * lane;           // 16 bytes per lane
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) {
#define X(i) asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(d[i]) : "v"(x), "v"(y));
            REP32(X)
#undef X
        } else if (MODE == 1) {
#define X(i) asm volatile("v_rsq_f64 %0, %1" : "=v"(d[i]) : "v"(x));
            REP32(X)
#undef X
        } else if (MODE == 2) {
#define X(i) asm volatile("v_rcp_f64 %0, %1" : "=v"(d[i]) : "v"(x));
            REP32(X)
#undef X
        } else if (MODE == 3) {
#define X(i) asm volatile("v_rsq_f32 %0, %1" : "=v"(f[i]) : "v"(f[(i + 1) & 7]));
            REP32(X)
#undef X
        } else if (MODE == 4) {
#define X(i) asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(n[i]) : "v"(lane), "s"(m));
            REP32(X)
#undef X
        } else if (MODE == 5) {
#define X(i) asm volatile("s_nop 1\n\tv_max_i32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(n[i]));
            REP32(X)
#undef X
        } else if (MODE == 6) {
#define X(i) asm volatile("v_mov_b64_dpp %0, %1 row_newbcast:3 row_mask:0xf bank_mask:0xf" : "=v"(d[i]) : "v"(x));
            REP32(X)
#undef X
        } else if (MODE == 7) {
#define X(i) asm volatile("v_mul_f64 %0, %1, %2" : "=v"(d[i]) : "v"(x), "v"(y));
            REP32(X)
#undef X
        } else if (MODE == 8) {
#define X(i) asm volatile("v_cmp_le_f64 vcc, %0, %1" :: "v"(d[i]), "v"(y) : "vcc");
            REP32(X)
#undef X
        } else if (MODE == 9) {
#define X(i) asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:5 row_mask:0xf bank_mask:0xf" : "+v"(d[i]) : "v"(x), "v"(y));
            REP32(X)
#undef X
        } else if (MODE == 10) {          // ds_write_b128: one lane (one_lane) or all 64
            if (!one_lane || lane == 5) {
#define X(i) asm volatile("ds_write_b128 %0, %1 offset:0" :: "v"((int)(size_t)wp), "v"(*(reinterpret_cast<__attribute__((ext_vector_type(4))) int*>(&d[(i) & 6]))) : "memory");
                REP32(X)
#undef X
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        } else if (MODE == 11) {          // ds_write_b64
            if (!one_lane || lane == 5) {
#define X(i) asm volatile("ds_write_b64 %0, %1 offset:0" :: "v"((int)(size_t)wp), "v"(d[i]) : "memory");
                REP32(X)
#undef X
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        } else if (MODE == 12) {          // ds_read2_b64 (all lanes, 16 bytes per lane)
#define X(i) asm volatile("ds_read2_b64 %0, %1 offset0:0 offset1:16" : "=v"(*(reinterpret_cast<__attribute__((ext_vector_type(4))) int*>(&d[(i) & 6]))) : "v"((int)(size_t)(sm + (lane & 15))) : "memory");
            REP32(X)
#undef X
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        } else if (MODE == 13) {          // ds_read_b64
#define X(i) asm volatile("ds_read_b64 %0, %1" : "=v"(d[i]) : "v"((int)(size_t)(sm + (lane & 15))) : "memory");
            REP32(X)
#undef X
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        } else if (MODE == 14) {          // half the stream FMAs, the other half single-lane ds_write_b128 (do they overlap?)
#define X(i) asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(d[i]) : "v"(x), "v"(y));
            REP32(X)
#undef X
            if (lane == 5) {
#define X(i) asm volatile("ds_write_b128 %0, %1 offset:0" :: "v"((int)(size_t)wp), "v"(*(reinterpret_cast<__attribute__((ext_vector_type(4))) int*>(&d[(i) & 6]))) : "memory");
                REP32(X)
#undef X
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        } else if (MODE == 18) {          // the same 16 bytes per lane as ds_read2_b64, by two ds_read_b64
#define X(i) asm volatile("ds_read_b64 %0, %2\n\tds_read_b64 %1, %2 offset:128" : "=v"(d[(i) & 6]), "=v"(d[((i) & 6) + 1]) : "v"((int)(size_t)(sm + (lane & 15))) : "memory");
            REP32(X)
#undef X
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        } else if (MODE == 19) {          // ds_write_b32, one lane
            if (!one_lane || lane == 5) {
#define X(i) asm volatile("ds_write_b32 %0, %1 offset:0" :: "v"((int)(size_t)wp), "v"(n[i]) : "memory");
                REP32(X)
#undef X
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        } else if (MODE == 20) {          // ds_read_b128, every lane its own 16 bytes, 272 bytes between lanes (the staging read)
#define X(i) asm volatile("ds_read_b128 %0, %1" : "=v"(*(reinterpret_cast<__attribute__((ext_vector_type(4))) int*>(&d[(i) & 6]))) : "v"((int)(size_t)(sm + 34 * (lane & 31))) : "memory");
            REP32(X)
#undef X
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        } else if (MODE == 15) {
#define X(i) asm volatile("v_permlane16_swap_b32 %0, %1" : "+v"(n[i]), "+v"(n[(i + 1) & 7]));
            REP32(X)
#undef X
        } else if (MODE == 16) {
#define X(i) asm volatile("v_rcp_f32 %0, %1" : "=v"(f[i]) : "v"(f[(i + 1) & 7]));
            REP32(X)
#undef X
        } else if (MODE == 17) {
#define X(i) asm volatile("v_bfi_b32 %0, %1, %0, %2" : "+v"(n[i]) : "s"((int)m), "v"(lane));
            REP32(X)
#undef X
        }
    }
    double s = 0;
    for (int i = 0; i < 8; ++i) s += d[i] + f[i] + n[i];
    out[blockIdx.x * 64 + lane] = s + sm[128 + lane];
}

template <int MODE>
static double run(double* out, int grid, int one_lane)
{
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int iters = 400;
    float best = 1e9f;
    for (int rep = 0; rep < 4; ++rep) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(64), 0, 0, out, iters, one_lane);
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    return best * 1e6 / ((double)iters * 32 * (grid / 1024.0));      // ns per instruction per SIMD
}

int main()
{
    double* out;
    (void)hipMalloc(&out, 4096 * 64 * sizeof(double));
    for (int grid : {256, 1024, 4096}) {
        printf("---- %d waves per SIMD (%d workgroups): ns per instruction per SIMD (relative to v_fma_f64); 256 workgroups = one wave per CU: per-SIMD accounting is 4 x too low there\n", grid / 1024, grid);
        const double fma = run<0>(out, grid, 0);
#define P(name, M, ol) { const double v = run<M>(out, grid, ol); printf("  %-44s %7.2f ns   %5.2f x fma\n", name, v, v / fma); }
        printf("  %-44s %7.2f ns\n", "v_fma_f64", fma);
        P("v_rsq_f64", 1, 0) P("v_rcp_f64", 2, 0) P("v_rsq_f32", 3, 0) P("v_rcp_f32", 16, 0) P("v_cndmask_b32 (sgpr mask)", 4, 0)
        P("v_max_i32_dpp + s_nop 1", 5, 0) P("v_mov_b64_dpp row_newbcast", 6, 0) P("v_mul_f64", 7, 0) P("v_cmp_le_f64", 8, 0)
        P("v_fmac_f64_dpp row_newbcast", 9, 0) P("v_permlane16_swap_b32", 15, 0) P("v_bfi_b32", 17, 0)
        P("ds_write_b128, one lane", 10, 1) P("ds_write_b128, 64 lanes", 10, 0) P("ds_write_b64, one lane", 11, 1) P("ds_write_b64, 64 lanes", 11, 0)
        P("ds_write_b32, one lane", 19, 1) P("2 x ds_read_b64 (per pair)", 18, 0) P("ds_read_b128, 272 B between lanes", 20, 0)
        P("ds_read2_b64 (16 B per lane)", 12, 0) P("ds_read_b64", 13, 0) P("32 fma + 32 one-lane ds_write_b128 (per pair of them)", 14, 0)
    }
    return 0;
}
