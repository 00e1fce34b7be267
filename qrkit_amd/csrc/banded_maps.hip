// banded_maps.hip -- the strips form of the banded solver (banded.hip, qrk_bbs_*): Q^T b, Q x and R^-1 y WITHOUT one workgroup walking
// every strip's factors, for gfx950.
//
// Reference seam: BandedBlockedSparseQR::_solve_impl (src/QRKit/BandedBlockedSparseQR.h:290-311) -- y = Q^T b as the product of the
// block reflectors in order (SparseBlockYTY_VecProduct, src/QRKit/SparseBlockYTY.h:100-139, with BlockYTY.h:152-172), then the
// triangular solve with R.  Both are chains from strip to strip, and bb_apply_q_kernel / bb_solve_r_kernel (banded.hip) walk them on ONE
// workgroup, which pulls every strip's Y and T (786 KB at the BASELINE configs[2] shape) and its rows of R through one CU, three
// dependent passes per strip: 52 + 19 us per strip, 3.5 s for the 50 000 strips of configs[2] on a chip whose other 255 CUs idle.
//
// What is sequential in those chains is small.  Panel i of stage B applies A_i = I + Y_i T_i^T Y_i^T to the stack
// [carry of panel i-1 interleaved with the first rows of the strip's own vector; the rest of it]; the only thing panel i + 1 needs from
// panel i is the carry -- lo = n - s numbers (128 at the configs[2] shape) -- and A_i is LINEAR:
//
//      carry_{i+1} = S_o A_i (E_c carry_i + E_y ya_i) = M_i carry_i + c_i,      M_i = S_o A_i E_c  (lo x lo),
//
// with E_c / E_y the placement of the carry / of the strip's vector in the stack and S_o the rows that leave as the next carry.  So:
//   1. bbs_carry_map_kernel    M_i of every panel, once per factorisation, all CUs (two small products per panel, 16 Mflop);
//   2. bbs_panel_apply_kernel  phase 1: every panel at once, carry = 0: c_i                                   (all CUs)
//   3. bbs_carry_chain_kernel  carry_{i+1} = M_i carry_i + c_i: ONE workgroup per right-hand side, 131 KB per strip instead of 786 KB,
//                              one pass instead of three, the next strip's M in flight while this one is summed;
//   4. bbs_panel_apply_kernel  phase 2: every panel at once with its true carry: everything but the carry rows  (all CUs)
// Q x runs the same maps transposed, from the last panel to the first (the carry a panel hands BACK is E_c^T A_i^T S_o^T = M_i^T).
//
// The triangular solve has the same shape: strip i owns rows [i s, i s + s) of R, x_i = D_i^-1 (y_i - U_i x_right) with x_right the next
// lo entries of x.  bbs_backsub_map_kernel forms G_i = D_i^-1 U_i (s x lo) once per factorisation; per solve h_i = D_i^-1 y_i for every
// strip at once (bbs_backsub_diag_kernel), then x_i = h_i - G_i x_right on one workgroup (bbs_backsub_chain_kernel: 65 KB per strip and
// no dependent divisions).  The last strip (all n rows of its triangle) goes through bb_solve_r_kernel as before.
//
// The results differ from the one-workgroup chains by rounding only (the same operators, associated differently); both forms are kept
// (QRK_BBS_MAPS=0 selects the old one) and tests/test_banded_strips_gpu.py compares them.
#include "banded_host.h"
#include "qrk_device.h"

#include <cstdlib>

namespace qrk {

namespace bbm {

// sum over the 64 lanes, the same value in every lane (banded.hip, bb_wave_sum_dpp)
__device__ __forceinline__ double wsum(double v)
{
    v += dpp_f64<0xB1>(v);
    v += dpp_f64<0x4E>(v);
    v += dpp_f64<0x141>(v);
    v += dpp_f64<0x140>(v);
    return (readlane_f64(v, 0) + readlane_f64(v, 16)) + (readlane_f64(v, 32) + readlane_f64(v, 48));
}

// orders LDS only: the global loads of the next step stay in flight across it (a __syncthreads() waits for them)
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

constexpr int PA_THREADS = 512;
constexpr int PA_WAVES = PA_THREADS / 64;
constexpr int CH_THREADS = 1024;

}  // namespace bbm

// ---- 2 / 4: one panel of the chain per workgroup -------------------------------------------------------------------------------------
// blockIdx.x = panel, blockIdx.y = right-hand side.  Layouts as in banded.hip (BBStrips): ya = [N][ms] stage-A vectors, full = the
// vector in the order [R part, cols | per strip: lo chain residuals (strips 1..) | ms - n stage-A residuals].
// carr: [nrhs][N][lo]; carr[p] is the carry that ENTERS panel p -- Q^T: from panel p - 1 (even stack rows below 2 lo), Q: from panel
// p + 1 (stack rows [solved, n)).  phase 1: the entering carry is taken as zero and only the leaving carry is written (added to by the
// chain kernel afterwards); phase 2: the entering carry is read and everything but the leaving carry is written.
__global__ void __launch_bounds__(bbm::PA_THREADS)
bbs_panel_apply_kernel(const BBPanel* __restrict__ panels, int num_panels, const double* __restrict__ y_vals, const double* __restrict__ t_vals,
                       int transpose, int phase, double* __restrict__ ya_all, int64_t ya_ld, double* __restrict__ full_all, int64_t full_ld,
                       double* __restrict__ carr_all, int ms, int s, int lo, int cols, int max_act, int max_n)
{
    using namespace bbm;
    extern __shared__ __attribute__((aligned(16))) double smem[];
    double* seg = smem;                      // [max_act]
    double* w1 = seg + max_act;              // [max_n]
    double* w2 = w1 + max_n;                 // [max_n]
    double* part = w2 + max_n;               // [PA_THREADS]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int pidx = blockIdx.x;
    const int64_t col = blockIdx.y;
    const BBPanel p = panels[pidx];
    const int m = p.act_rows, n = p.ncols, solved = p.solved;
    // panels that hand no carry on have nothing to do in phase 1
    if (phase == 1 && (transpose ? solved >= n : pidx == 0)) return;
    const double* Y = y_vals + p.y_off;
    const double* T = t_vals + p.t_off;
    double* yi = ya_all + col * ya_ld + (int64_t)pidx * ms;
    double* full = full_all + col * full_ld;
    double* carr = carr_all + col * (int64_t)num_panels * lo;
    const int64_t ro = (int64_t)cols + (int64_t)pidx * (ms - n) + (pidx >= 1 ? (int64_t)(pidx - 1) * lo : 0);
    const double* cin = carr + (int64_t)pidx * lo;

    // ---- the stack vector
    if (transpose) {
        for (int i = tid; i < m; i += PA_THREADS) {
            double val;
            if (pidx == 0) val = yi[i];
            else if (i < 2 * lo) val = (i & 1) ? yi[i >> 1] : (phase == 2 ? cin[i >> 1] : 0.0);
            else val = yi[i - lo];
            seg[i] = val;
        }
    } else {
        for (int i = tid; i < m; i += PA_THREADS) {
            double val;
            if (i < solved) val = full[(int64_t)s * pidx + i];
            else if (i < n) val = phase == 2 ? cin[i - solved] : 0.0;
            else val = full[ro + (i - n)];
            seg[i] = val;
        }
    }
    __syncthreads();
    // ---- w1 = Y^T seg (Y = unit-lower view of the panel, row-major m x n): thread = column x row group, coalesced rows
    const int CW = ((n + 63) / 64) * 64, RG = PA_THREADS / CW;
    {
        const int jl = tid % CW, g = tid / CW;
        double acc = 0.0;
        if (g < RG && jl < n) {
            constexpr int U = 8;
            for (int i = jl + 1 + g; i < m; i += U * RG) {
                double yv[U];
#pragma unroll
                for (int u = 0; u < U; ++u) { const int ii = i + u * RG; yv[u] = Y[(int64_t)(ii < m ? ii : m - 1) * n + jl]; }
#pragma unroll
                for (int u = 0; u < U; ++u) { const int ii = i + u * RG; if (ii < m) acc = fma(yv[u], seg[ii], acc); }
            }
        }
        if (g < RG) part[g * CW + jl] = acc;
        __syncthreads();
        if (tid < n) {
            double d = seg[tid];                 // the unit diagonal (m >= n)
            for (int q = 0; q < RG; ++q) d += part[q * CW + tid];
            w1[tid] = d;
        }
        __syncthreads();
    }
    // ---- w2 = T^T w1 (Q^T) or T w1 (Q); T upper triangular, column-major, negated by the factorisation
    if (transpose) {
        constexpr int RB = 4, LB = 4;            // four rows of T^T per wave in flight, n <= 256
        for (int i0 = wave; i0 < n; i0 += RB * PA_WAVES) {
            double tv[RB][LB];
#pragma unroll
            for (int rb = 0; rb < RB; ++rb) {
                const int i = i0 + rb * PA_WAVES;
#pragma unroll
                for (int q = 0; q < LB; ++q) { const int j = lane + 64 * q; tv[rb][q] = (i < n && j <= i) ? T[(int64_t)i * n + j] : 0.0; }
            }
#pragma unroll
            for (int rb = 0; rb < RB; ++rb) {
                const int i = i0 + rb * PA_WAVES;
                double d = 0.0;
#pragma unroll
                for (int q = 0; q < LB; ++q) { const int j = lane + 64 * q; if (j < n) d = fma(tv[rb][q], w1[j], d); }
                d = wsum(d);
                if (lane == 0 && i < n) w2[i] = d;
            }
        }
    } else {
        const int il = tid % CW, g = tid / CW;
        double acc = 0.0;
        if (g < RG && il < n) {
            constexpr int U = 8;
            for (int j = il + g; j < n; j += U * RG) {
                double tv[U];
#pragma unroll
                for (int u = 0; u < U; ++u) { const int jj = j + u * RG; tv[u] = T[(int64_t)(jj < n ? jj : n - 1) * n + il]; }
#pragma unroll
                for (int u = 0; u < U; ++u) { const int jj = j + u * RG; if (jj < n) acc = fma(tv[u], w1[jj], acc); }
            }
        }
        if (g < RG) part[g * CW + il] = acc;
        __syncthreads();
        if (tid < n) {
            double d = 0.0;
            for (int q = 0; q < RG; ++q) d += part[q * CW + tid];
            w2[tid] = d;
        }
    }
    __syncthreads();
    // ---- seg + Y w2 on the rows this phase writes: a wave per row, four rows in flight
    auto carry_row = [&](int i) -> bool {        // stack rows that leave as the carry of the next panel of the chain
        return transpose ? (i >= solved && i < n) : (pidx >= 1 && i < 2 * lo && !(i & 1));
    };
    constexpr int RB3 = 4, LB3 = 4;
    for (int i0 = wave; i0 < m; i0 += RB3 * PA_WAVES) {
        double yv[RB3][LB3];
        bool want[RB3];
#pragma unroll
        for (int rb = 0; rb < RB3; ++rb) {
            const int i = i0 + rb * PA_WAVES;
            want[rb] = i < m && (carry_row(i) == (phase == 1));
            const int je = i < n ? i : n;
#pragma unroll
            for (int q = 0; q < LB3; ++q) { const int j = lane + 64 * q; yv[rb][q] = (want[rb] && j < je) ? Y[(int64_t)i * n + j] : 0.0; }
        }
#pragma unroll
        for (int rb = 0; rb < RB3; ++rb) {
            const int i = i0 + rb * PA_WAVES;
            if (!want[rb]) continue;             // (wave-uniform)
            double d = 0.0;
#pragma unroll
            for (int q = 0; q < LB3; ++q) { const int j = lane + 64 * q; if (j < n) d = fma(yv[rb][q], w2[j], d); }
            d = wsum(d);
            if (lane == 0) {
                const double val = seg[i] + d + (i < n ? w2[i] : 0.0);
                if (transpose) {
                    if (i < solved) full[(int64_t)s * pidx + i] = val;
                    else if (i < n) carr[(int64_t)(pidx + 1) * lo + (i - solved)] = val;
                    else full[ro + (i - n)] = val;
                } else {
                    if (pidx == 0) yi[i] = val;
                    else if (i < 2 * lo) { if (i & 1) yi[i >> 1] = val; else carr[(int64_t)(pidx - 1) * lo + (i >> 1)] = val; }
                    else yi[i - lo] = val;
                }
            }
        }
    }
    if (phase == 2) {
        // the components stage A left out of the chain (rows n.. of the strip's vector) pass through
        const int64_t rp = ro + (pidx >= 1 ? lo : 0);
        for (int q = tid; q < ms - n; q += PA_THREADS) { if (transpose) full[rp + q] = yi[n + q]; else yi[n + q] = full[rp + q]; }
    }
}

// ---- 3: the chain of the carries -------------------------------------------------------------------------------------------------------
// One workgroup per right-hand side.  Q^T (TR): for p = 1 .. N - 2: carr[p + 1] += M_p carr[p];  Q: for p = N - 2 .. 1:
// carr[p - 1] += M_p^T carr[p].  M_p row-major lo x lo at cmap + p lo^2.  FAST (lo <= 128): every thread owns 16 entries of M_p and the
// ones of the next step are in flight while this step is summed.
template <bool TR, bool FAST>
__global__ void __launch_bounds__(bbm::CH_THREADS)
bbs_carry_chain_kernel(const double* __restrict__ cmap, double* __restrict__ carr_all, int num_panels, int lo)
{
    using namespace bbm;
    __shared__ double cv[2][256];
    __shared__ double part[CH_THREADS];
    const int tid = threadIdx.x;
    double* carr = carr_all + (int64_t)blockIdx.x * num_panels * lo;
    if (num_panels < 3) return;
    const int64_t l2 = (int64_t)lo * lo;
    const int first = TR ? 1 : num_panels - 2, last = TR ? num_panels - 2 : 1, dir = TR ? 1 : -1;
    // thread -> entries of M.  TR: chunk e = (row a, 16 columns from 16 q): partial of out[a].  Q: (column b, rows g, g + G, ..): partial of out[b].
    const int Q = lo / 16, G = CH_THREADS / lo;
    const int a = TR ? tid / Q : 0, q = TR ? tid % Q : 0, b = TR ? 0 : tid % lo, g = TR ? 0 : tid / lo;
    const bool act = TR ? tid < lo * Q : g < G;
    if (tid < lo) cv[0][tid] = carr[(int64_t)first * lo + tid];
    int buf = 0;
    if (FAST) {
        typedef double d2 __attribute__((ext_vector_type(2)));
        double cur[16], nxt[16];
        double addc = 0.0, addn = 0.0;
        auto load = [&](int p, double (&dst)[16], double& add) {
            const double* M = cmap + (int64_t)p * l2;
            if (TR) {
                const d2* src = reinterpret_cast<const d2*>(M + (int64_t)(act ? a : 0) * lo + 16 * (act ? q : 0));
#pragma unroll
                for (int u = 0; u < 8; ++u) { const d2 v = src[u]; dst[2 * u] = v.x; dst[2 * u + 1] = v.y; }
            } else {
#pragma unroll
                for (int u = 0; u < 16; ++u) { const int r = g + G * u; dst[u] = M[(int64_t)((act && r < lo) ? r : 0) * lo + (act ? b : 0)]; }
            }
            add = tid < lo ? carr[(int64_t)(p + dir) * lo + tid] : 0.0;
        };
        load(first, cur, addc);
        __syncthreads();
        for (int p = first;; p += dir) {
            const bool more = p != last;
            if (more) load(p + dir, nxt, addn);
            double acc = 0.0;
            if (TR) {
                const double* c = &cv[buf][16 * q];
                double a0 = 0.0, a1 = 0.0;
#pragma unroll
                for (int u = 0; u < 16; u += 2) { a0 = fma(cur[u], c[u], a0); a1 = fma(cur[u + 1], c[u + 1], a1); }
                acc = a0 + a1;
            } else {
                double a0 = 0.0, a1 = 0.0;
#pragma unroll
                for (int u = 0; u < 16; u += 2) {
                    const int r0 = g + G * u, r1 = g + G * (u + 1);
                    a0 = fma(cur[u], r0 < lo ? cv[buf][r0] : 0.0, a0);
                    a1 = fma(cur[u + 1], r1 < lo ? cv[buf][r1] : 0.0, a1);
                }
                acc = a0 + a1;
            }
            if (act) part[tid] = acc;
            lds_barrier();
            if (tid < lo) {
                double sum = addc;
                if (TR) { for (int z = 0; z < Q; ++z) sum += part[tid * Q + z]; }
                else { for (int z = 0; z < G; ++z) sum += part[z * lo + tid]; }
                cv[buf ^ 1][tid] = sum;
                carr[(int64_t)(p + dir) * lo + tid] = sum;
            }
            lds_barrier();
            buf ^= 1;
            if (!more) break;
#pragma unroll
            for (int u = 0; u < 16; ++u) cur[u] = nxt[u];
            addc = addn;
        }
    } else {
        __syncthreads();
        for (int p = first;; p += dir) {
            const double* M = cmap + (int64_t)p * l2;
            double acc = 0.0;
            if (TR) {
                // chunks of 16 entries of a row in passes of CH_THREADS chunks (lo <= 240: at most four); the partial sums of a row meet
                // in LDS after every pass
                const int nch = lo * Q;
                for (int base = 0; base < nch; base += CH_THREADS) {
                    const int e = base + tid;
                    if (e < nch) {
                        const int ra = e / Q, cq = e % Q;
                        const double* src = M + (int64_t)ra * lo + 16 * cq;
                        double v[16];
#pragma unroll
                        for (int u = 0; u < 16; ++u) v[u] = src[u];
                        double sacc = 0.0;
#pragma unroll
                        for (int u = 0; u < 16; ++u) sacc = fma(v[u], cv[buf][16 * cq + u], sacc);
                        part[tid] = sacc;
                    }
                    lds_barrier();
                    if (tid < lo) {
                        int z0 = tid * Q, z1 = z0 + Q;
                        if (z0 < base) z0 = base;
                        if (z1 > base + CH_THREADS) z1 = base + CH_THREADS;
                        if (z1 > nch) z1 = nch;
                        for (int z = z0; z < z1; ++z) acc += part[z - base];
                    }
                    lds_barrier();
                }
            } else {
                if (g < G) {
                    constexpr int U = 8;
                    for (int r = g; r < lo; r += U * G) {
                        double v[U];
#pragma unroll
                        for (int u = 0; u < U; ++u) { const int rr = r + G * u; v[u] = M[(int64_t)(rr < lo ? rr : 0) * lo + b]; }
#pragma unroll
                        for (int u = 0; u < U; ++u) { const int rr = r + G * u; if (rr < lo) acc = fma(v[u], cv[buf][rr], acc); }
                    }
                    part[tid] = acc;
                }
                lds_barrier();
                acc = 0.0;
                if (tid < lo) for (int z = 0; z < G; ++z) acc += part[z * lo + tid];
            }
            if (tid < lo) {
                const double sum = acc + carr[(int64_t)(p + dir) * lo + tid];
                cv[buf ^ 1][tid] = sum;
                carr[(int64_t)(p + dir) * lo + tid] = sum;
            }
            __syncthreads();
            buf ^= 1;
            if (p == last) break;
        }
    }
}

// ---- 1: M_p = S_o (I + Y' T^T Y'^T) E_c of the panels 1 .. N - 2 -------------------------------------------------------------------------
// Two products per 64 columns of M: Z = T^T B (n x 64, kept in LDS) with B[k][b] = Y'(2 b, k), then M = [sel] + Y'(solved + a, :) Z.
// 256 threads as 16 x 16, a 4 x 4 tile of the 64 x 64 block each, operands through LDS in slabs of 16.
__global__ void __launch_bounds__(256)
bbs_carry_map_kernel(const BBPanel* __restrict__ panels, const double* __restrict__ y_vals, const double* __restrict__ t_vals, int lo,
                     double* __restrict__ cmap)
{
    extern __shared__ __attribute__((aligned(16))) double smem[];
    constexpr int LS = 68;                   // row stride of the operand slabs (doubles)
    double* As = smem;                       // [16][LS]
    double* Bs = As + 16 * LS;               // [16][LS]
    double* Zs = Bs + 16 * LS;               // [n][64]
    const int tid = threadIdx.x, tx = tid & 15, ty = tid >> 4;
    const int pi = blockIdx.x + 1;
    const BBPanel p = panels[pi];
    const int n = p.ncols, sv = p.solved;
    const double* Y = y_vals + p.y_off;
    const double* T = t_vals + p.t_off;
    double* M = cmap + (int64_t)pi * lo * lo;
    auto yp = [&](int r, int k) -> double { return k < r ? Y[(int64_t)r * n + k] : (k == r ? 1.0 : 0.0); };      // unit-lower view, k < n
    for (int b0 = 0; b0 < lo; b0 += 64) {
        // ---- Z[j][bb] = sum_{k <= j} T(k, j) Y'(2 (b0 + bb), k)
        for (int j0 = 0; j0 < n; j0 += 64) {
            double acc[4][4] = {};
            int kend = j0 + 64 < n ? j0 + 64 : n;
            if (2 * (b0 + 63) + 1 < kend) kend = 2 * (b0 + 63) + 1;        // (Y'(2 b, k) = 0 for k > 2 b)
            for (int k0 = 0; k0 < kend; k0 += 16) {
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int e = tid + 256 * u, xx = e >> 4, kk = e & 15, k = k0 + kk;
                    const int j = j0 + xx, bq = b0 + xx;
                    As[kk * LS + xx] = (j < n && k <= j) ? T[(int64_t)j * n + k] : 0.0;
                    Bs[kk * LS + xx] = (bq < lo && k < n) ? yp(2 * bq, k) : 0.0;
                }
                __syncthreads();
#pragma unroll
                for (int kk = 0; kk < 16; ++kk) {
                    double av[4], bv[4];
#pragma unroll
                    for (int z = 0; z < 4; ++z) { av[z] = As[kk * LS + 4 * ty + z]; bv[z] = Bs[kk * LS + 4 * tx + z]; }
#pragma unroll
                    for (int r = 0; r < 4; ++r)
#pragma unroll
                        for (int c = 0; c < 4; ++c) acc[r][c] = fma(av[r], bv[c], acc[r][c]);
                }
                __syncthreads();
            }
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int c = 0; c < 4; ++c) { const int j = j0 + 4 * ty + r; if (j < n) Zs[j * 64 + 4 * tx + c] = acc[r][c]; }
        }
        __syncthreads();
        // ---- M[a][b0 + bb] = [solved + a == 2 (b0 + bb)] + sum_j Y'(solved + a, j) Z[j][bb]
        for (int a0 = 0; a0 < lo; a0 += 64) {
            double acc[4][4] = {};
            const int jend = sv + a0 + 64 < n ? sv + a0 + 64 : n;          // (Y'(r, j) = 0 for j > r)
            for (int k0 = 0; k0 < jend; k0 += 16) {
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int e = tid + 256 * u, xx = e >> 4, kk = e & 15, k = k0 + kk;
                    const int aq = a0 + xx;
                    As[kk * LS + xx] = (aq < lo && k < n) ? yp(sv + aq, k) : 0.0;
                }
                __syncthreads();
#pragma unroll
                for (int kk = 0; kk < 16; ++kk) {
                    if (k0 + kk >= n) break;
                    double av[4], bv[4];
#pragma unroll
                    for (int z = 0; z < 4; ++z) { av[z] = As[kk * LS + 4 * ty + z]; bv[z] = Zs[(k0 + kk) * 64 + 4 * tx + z]; }
#pragma unroll
                    for (int r = 0; r < 4; ++r)
#pragma unroll
                        for (int c = 0; c < 4; ++c) acc[r][c] = fma(av[r], bv[c], acc[r][c]);
                }
                __syncthreads();
            }
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const int aq = a0 + 4 * ty + r, bq = b0 + 4 * tx + c;
                    if (aq < lo && bq < lo) M[(int64_t)aq * lo + bq] = acc[r][c] + ((sv + aq == 2 * bq) ? 1.0 : 0.0);
                }
        }
        __syncthreads();
    }
}

// ---- the triangular solve ------------------------------------------------------------------------------------------------------------------
// G_p = D_p^-1 U_p of the panels 0 .. N - 2 (solved = s <= 64 rows each): D = R(0:s, 0:s), U = R(0:s, s:n); R(i, j) = R[j s + i] in the
// staging array.  A thread per column of U, the column in registers, the triangle padded to 64 x 64 with the identity.  gmap: [N][64][lo].
__global__ void __launch_bounds__(256)
bbs_backsub_map_kernel(const BBPanel* __restrict__ panels, const double* __restrict__ r_stage, int lo, double* __restrict__ gmap)
{
    __shared__ double blk[64 * 65];          // blk[j * 65 + i] = D(i, j)
    __shared__ double rd[64];
    const int tid = threadIdx.x;
    const BBPanel p = panels[blockIdx.x];
    const int n = p.ncols, sv = p.solved;
    const double* R = r_stage + p.r_off;
    for (int e = tid; e < 64 * 64; e += 256) {
        const int j = e >> 6, i = e & 63;
        blk[j * 65 + i] = (i < sv && j < sv) ? (i <= j ? R[(int64_t)j * sv + i] : 0.0) : (i == j ? 1.0 : 0.0);
    }
    __syncthreads();
    if (tid < 64) rd[tid] = 1.0 / blk[tid * 65 + tid];
    __syncthreads();
    double* Gp = gmap + (int64_t)blockIdx.x * 64 * lo;
    for (int c = tid; c < n - sv && c < lo; c += 256) {
        double x[64];
#pragma unroll
        for (int i = 0; i < 64; ++i) x[i] = i < sv ? R[(int64_t)(sv + c) * sv + i] : 0.0;
#pragma unroll
        for (int i = 63; i >= 0; --i) {
            const double d = blk[i * 65 + i], r = rd[i];
            double xi = x[i] * r;
            xi = fma(fma(-xi, d, x[i]), r, xi);          // the quotient to the last bit but for rare ties (banded.hip, dense_solve_r_coop_kernel)
            x[i] = xi;
#pragma unroll
            for (int k = 0; k < i; ++k) x[k] = fma(-blk[i * 65 + k], xi, x[k]);
        }
#pragma unroll
        for (int i = 0; i < 64; ++i) if (i < sv) Gp[(int64_t)i * lo + c] = x[i];
    }
}

// h_p = D_p^-1 y_p in place for the panels 0 .. N - 2: blockIdx.x = panel, blockIdx.y = group of four right-hand sides (a wave each)
__global__ void __launch_bounds__(256)
bbs_backsub_diag_kernel(const BBPanel* __restrict__ panels, const double* __restrict__ r_stage, double* __restrict__ v, int64_t ldv, int64_t nrhs)
{
    __shared__ double blk[64 * 65];
    const int tid = threadIdx.x, ln = tid & 63, grp = tid >> 6;
    const BBPanel p = panels[blockIdx.x];
    const int sv = p.solved;
    const double* R = r_stage + p.r_off;
    for (int e = tid; e < sv * sv; e += 256) {
        const int j = e / sv, i = e - j * sv;
        blk[j * 65 + i] = R[(int64_t)j * sv + i];
    }
    __syncthreads();
    const int64_t col = (int64_t)blockIdx.y * 4 + grp;
    if (col >= nrhs) return;
    double* x = v + col * ldv + p.col0;
    double t = ln < sv ? x[ln] : 0.0;
    const double dl = blk[(ln < sv ? ln : 0) * 65 + (ln < sv ? ln : 0)];
    const double rdl = 1.0 / dl;
    for (int i = sv - 1; i >= 0; --i) {
        const double ti = readlane_f64(t, i), di = readlane_f64(dl, i), ri = readlane_f64(rdl, i);
        double xi = ti * ri;
        xi = fma(fma(-xi, di, ti), ri, xi);
        if (ln < i) t = fma(-blk[i * 65 + ln], xi, t);
        else if (ln == i) t = xi;
    }
    if (ln < sv) x[ln] = t;
}

// x_p = h_p - G_p x_right for p = N - 2 .. 0, one workgroup per right-hand side; x of the last panel is final when this starts.  The last
// n + s entries of x live in a ring in LDS; every thread owns 16 entries of G_p and those of the next panel are in flight meanwhile.
__global__ void __launch_bounds__(bbm::CH_THREADS)
bbs_backsub_chain_kernel(const double* __restrict__ gmap, int num_panels, int n, int s, int lo, double* __restrict__ v, int64_t ldv)
{
    using namespace bbm;
    typedef double d2 __attribute__((ext_vector_type(2)));
    __shared__ double xs[512];
    __shared__ double part[CH_THREADS];
    const int tid = threadIdx.x;
    double* x = v + (int64_t)blockIdx.x * ldv;
    const int Q = lo / 16;
    const int r = tid / Q, q = tid % Q;
    const bool act = tid < s * Q;
    const int64_t cl = (int64_t)(num_panels - 1) * s;            // col0 of the last panel
    for (int t = tid; t < n; t += CH_THREADS) xs[(cl + t) & 511] = x[cl + t];
    double cur[16], nxt[16], hc = 0.0, hn = 0.0;
    auto load = [&](int p, double (&dst)[16], double& h) {
        const d2* src = reinterpret_cast<const d2*>(gmap + (int64_t)p * 64 * lo + (int64_t)(act ? r : 0) * lo + 16 * (act ? q : 0));
#pragma unroll
        for (int u = 0; u < 8; ++u) { const d2 g2 = src[u]; dst[2 * u] = g2.x; dst[2 * u + 1] = g2.y; }
        h = tid < s ? x[(int64_t)p * s + tid] : 0.0;
    };
    load(num_panels - 2, cur, hc);
    __syncthreads();
    for (int p = num_panels - 2; p >= 0; --p) {
        if (p > 0) load(p - 1, nxt, hn);
        const int64_t c0 = (int64_t)p * s;
        double a0 = 0.0, a1 = 0.0;
#pragma unroll
        for (int u = 0; u < 16; u += 2) {
            a0 = fma(cur[u], xs[(c0 + s + 16 * q + u) & 511], a0);
            a1 = fma(cur[u + 1], xs[(c0 + s + 16 * q + u + 1) & 511], a1);
        }
        if (act) part[tid] = a0 + a1;
        lds_barrier();
        if (tid < s) {
            double sum = 0.0;
            for (int z = 0; z < Q; ++z) sum += part[tid * Q + z];
            const double xi = hc - sum;
            xs[(c0 + tid) & 511] = xi;
            x[c0 + tid] = xi;
        }
        lds_barrier();
#pragma unroll
        for (int u = 0; u < 16; ++u) cur[u] = nxt[u];
        hc = hn;
    }
}

// ---- launchers ----------------------------------------------------------------------------------------------------------------------------
// the maps of a factorisation: cmap [N][lo][lo] (panels 1 .. N - 2 are written), gmap [N][64][lo] or null (s > 64: the triangular solve
// stays on bb_solve_r_kernel)
hipError_t launch_bbs_maps(const BBPanel* panels, int num_panels, const double* y_vals, const double* t_vals, const double* r_stage, int n,
                           int lo, double* cmap, double* gmap, hipStream_t stream)
{
    if (lo <= 0 || num_panels < 2) return hipSuccess;
    if (num_panels >= 3) {
        const size_t smem = (size_t)(2 * 16 * 68 + n * 64) * sizeof(double);
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(bbs_carry_map_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(bbs_carry_map_kernel, dim3((unsigned)(num_panels - 2)), dim3(256), smem, stream, panels, y_vals, t_vals, lo, cmap);
    }
    if (gmap) hipLaunchKernelGGL(bbs_backsub_map_kernel, dim3((unsigned)(num_panels - 1)), dim3(256), 0, stream, panels, r_stage, lo, gmap);
    return hipGetLastError();
}

// Q^T (transpose) or Q through the carry maps; carr: [nrhs][N][lo] scratch
hipError_t launch_bbs_apply_maps(const BBPanel* panels, int num_panels, const double* y_vals, const double* t_vals, const double* cmap,
                                 int transpose, double* ya, int64_t ya_ld, double* full, int64_t full_ld, int64_t nrhs, int ms, int n, int s,
                                 int lo, int cols, int max_act, double* carr, hipStream_t stream)
{
    using namespace bbm;
    if (nrhs <= 0) return hipSuccess;
    if (nrhs > 65535 || lo <= 0 || lo % 16 || num_panels < 2) return hipErrorInvalidValue;
    const size_t smem = (size_t)(max_act + 2 * n + PA_THREADS) * sizeof(double);
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(bbs_panel_apply_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    if (e != hipSuccess) return e;
    const dim3 grid((unsigned)num_panels, (unsigned)nrhs);
    hipLaunchKernelGGL(bbs_panel_apply_kernel, grid, dim3(PA_THREADS), smem, stream, panels, num_panels, y_vals, t_vals, transpose, 1, ya, ya_ld,
                       full, full_ld, carr, ms, s, lo, cols, max_act, n);
    if (num_panels >= 3) {
        const bool fast = lo <= 128;
        if (transpose) {
            if (fast) hipLaunchKernelGGL((bbs_carry_chain_kernel<true, true>), dim3((unsigned)nrhs), dim3(CH_THREADS), 0, stream, cmap, carr, num_panels, lo);
            else hipLaunchKernelGGL((bbs_carry_chain_kernel<true, false>), dim3((unsigned)nrhs), dim3(CH_THREADS), 0, stream, cmap, carr, num_panels, lo);
        } else {
            if (fast) hipLaunchKernelGGL((bbs_carry_chain_kernel<false, true>), dim3((unsigned)nrhs), dim3(CH_THREADS), 0, stream, cmap, carr, num_panels, lo);
            else hipLaunchKernelGGL((bbs_carry_chain_kernel<false, false>), dim3((unsigned)nrhs), dim3(CH_THREADS), 0, stream, cmap, carr, num_panels, lo);
        }
    }
    hipLaunchKernelGGL(bbs_panel_apply_kernel, grid, dim3(PA_THREADS), smem, stream, panels, num_panels, y_vals, t_vals, transpose, 2, ya, ya_ld,
                       full, full_ld, carr, ms, s, lo, cols, max_act, n);
    return hipGetLastError();
}

// x(0:cols) <- R^-1 x(0:cols) through G_p (s <= 64): the last panel as before, h_p of every other panel at once, then the chain
hipError_t launch_bbs_solve_r_maps(const BBPanel* panels, int num_panels, const double* r_stage, const double* gmap, int n, int s, int lo,
                                   int cols, double* v, int64_t ldv, int64_t nrhs, hipStream_t stream)
{
    using namespace bbm;
    if (nrhs <= 0) return hipSuccess;
    if (nrhs > 65535 || s > 64 || lo <= 0 || lo % 16 || num_panels < 2 || !gmap) return hipErrorInvalidValue;
    hipError_t e = launch_bb_solve_r(panels + (num_panels - 1), 1, r_stage, cols, v, ldv, nrhs, stream);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(bbs_backsub_diag_kernel, dim3((unsigned)(num_panels - 1), (unsigned)((nrhs + 3) / 4)), dim3(256), 0, stream, panels, r_stage,
                       v, ldv, nrhs);
    hipLaunchKernelGGL(bbs_backsub_chain_kernel, dim3((unsigned)nrhs), dim3(CH_THREADS), 0, stream, gmap, num_panels, n, s, lo, v, ldv);
    return hipGetLastError();
}

}  // namespace qrk
