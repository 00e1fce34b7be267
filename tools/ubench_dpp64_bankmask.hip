// 64-bit DPP operations under a bank mask on gfx950 (v_fmac_f64_dpp / v_mov_b64_dpp with row_newbcast), the building block of the
// 8-lanes-per-tile form of bdqr_quad.hip.
//   test 0-2: does bank_mask work at all (lanes outside the mask keep their value), and the same under an EXEC mask
//   test 3-8: the HAZARD found in round 5 -- two DPP operations with DIFFERENT bank masks writing the same register back to back: the
//             second one must preserve the lanes the first one wrote, and reads them stale unless enough independent work (or an s_nop)
//             sits between the two.  Each test starts from d = 100 and wants d = 100 + x[lane's half-row broadcast] in every lane.
//   test 9:   an ordinary VALU write (v_mov) followed at once by a masked DPP operation: do the masked-off lanes keep the fresh value?
//   test 10:  the SAME mask back to back on one register (the accumulation chain of a dot product): no hazard expected
// Build: hipcc -O3 --offload-arch=gfx950 tools/ubench_dpp64_bankmask.hip -o build/ubench_dpp64_bankmask
#include <hip/hip_runtime.h>
#include <cstdio>
#define LOW(d, n)  "v_fmac_f64_dpp " d ", %1, %2 row_newbcast:" n " row_mask:0xf bank_mask:0x3\n\t"
#define HIGH(d, n) "v_fmac_f64_dpp " d ", %1, %2 row_newbcast:" n " row_mask:0xf bank_mask:0xc\n\t"
__global__ void k(double* out)
{
    const int lane = threadIdx.x;
    double x = lane, one = 1.0, d = 100.0, e = 100.0, m = -1.0;
    asm volatile("s_nop 1\n\tv_fmac_f64_dpp %0, %1, %2 row_newbcast:3 row_mask:0xf bank_mask:0x3" : "+v"(d) : "v"(x), "v"(one));
    asm volatile("s_nop 1\n\tv_mov_b64_dpp %0, %1 row_newbcast:11 row_mask:0xf bank_mask:0xc" : "+v"(m) : "v"(x));
    unsigned long long lo = 0x00ff00ff00ff00ffull;
    asm volatile("s_mov_b64 exec, %3\n\ts_nop 4\n\tv_fmac_f64_dpp %0, %1, %2 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\ts_mov_b64 exec, -1" : "+v"(e) : "v"(x), "v"(one), "s"(lo));
    out[lane] = d; out[64 + lane] = m; out[128 + lane] = e;
    double t3 = 100.0, t4 = 100.0, t5 = 100.0, t6 = 100.0, t7 = 100.0, t8 = 100.0, u = 0.0, w = 0.0; int ui = 0;
    asm volatile("s_nop 4\n\t" LOW("%0", "3") HIGH("%0", "11") : "+v"(t3) : "v"(x), "v"(one));
    asm volatile("s_nop 4\n\t" LOW("%0", "3") "v_add_f64 %3, %1, %2\n\t" HIGH("%0", "11") : "+v"(t4) : "v"(x), "v"(one), "v"(u));
    asm volatile("s_nop 4\n\t" LOW("%0", "3") "v_add_f64 %3, %1, %2\n\tv_add_f64 %4, %1, %2\n\t" HIGH("%0", "11") : "+v"(t5) : "v"(x), "v"(one), "v"(u), "v"(w));
    asm volatile("s_nop 4\n\t" LOW("%0", "3") "s_nop 0\n\t" HIGH("%0", "11") : "+v"(t6) : "v"(x), "v"(one));
    asm volatile("s_nop 4\n\t" LOW("%0", "3") "s_nop 1\n\t" HIGH("%0", "11") : "+v"(t7) : "v"(x), "v"(one));
    asm volatile("s_nop 4\n\t" LOW("%0", "3") "v_mov_b32 %3, 0\n\t" HIGH("%0", "11") : "+v"(t8) : "v"(x), "v"(one), "v"(ui));
    out[192 + lane] = t3; out[256 + lane] = t4; out[320 + lane] = t5; out[384 + lane] = t6; out[448 + lane] = t7; out[512 + lane] = t8;
    // 9: fresh VALU write, then a masked operation at once (4.0 by v_mov_b64; wanted: 4 + x[3] in lanes 0..7 of a row, 4 in 8..15)
    double t9;
    asm volatile("s_nop 4\n\tv_mov_b64 %0, 4.0\n\t" LOW("%0", "3") : "=&v"(t9) : "v"(x), "v"(one));
    out[576 + lane] = t9;
    // 10: same mask twice in a row on one register (wanted 100 + 2 x[3] in lanes 0..7, 100 in 8..15)
    double t10 = 100.0;
    asm volatile("s_nop 4\n\t" LOW("%0", "3") LOW("%0", "3") : "+v"(t10) : "v"(x), "v"(one));
    out[640 + lane] = t10;
}
int main()
{
    const int NT = 11;
    double* o; (void)hipMalloc(&o, NT * 64 * 8);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, o);
    double h[NT * 64]; (void)hipMemcpy(h, o, sizeof(h), hipMemcpyDeviceToHost);
    const char* names[NT] = {"0 v_fmac_f64_dpp row_newbcast:3 bank_mask:0x3 (d = 100 + x[3])", "1 v_mov_b64_dpp row_newbcast:11 bank_mask:0xc (init -1)",
                             "2 v_fmac_f64_dpp row_newbcast:3 under EXEC = lanes 0..7 of every row",
                             "3 low, high back to back (want 103 x8, 111 x8, 119 x8, 127 x8)", "4 low, one independent v_add_f64, high", "5 low, two independent v_add_f64, high",
                             "6 low, s_nop 0, high", "7 low, s_nop 1, high", "8 low, one v_mov_b32, high",
                             "9 v_mov_b64 4.0 then low at once (want 7 x8, 4 x8, 23 x8, 4 x8)", "10 low, low on one register (want 106 x8, 100 x8, 138 x8, 100 x8)"};
    for (int t = 0; t < NT; ++t) { printf("%s\n  ", names[t]); for (int l = 0; l < 32; ++l) printf("%g ", h[64 * t + l]); printf("\n"); }
    return 0;
}
