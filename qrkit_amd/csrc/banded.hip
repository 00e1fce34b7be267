// banded.hip -- device side of the block-banded solver, for gfx950.
//
// Replaces the numeric part of QRKit::BandedBlockedSparseQR::factorize
// (src/QRKit/BandedBlockedSparseQR.h:443-519) and the implicit-Q products
// (src/QRKit/SparseBlockYTY.h:100-139 with src/QRKit/BlockYTY.h:152-172):
//   bb_chain_kernel    the sequential chain of dense panels: panel = leftover triangle of the previous
//                      panel stacked on the rows of the next block (:493-507), Eigen::HouseholderQR of
//                      the panel (:468), Y = unit-lower essentials and T = -make_block_householder_
//                      triangular_factor (:471-477), R rows of the panel incl. explicit zeros (:484-491);
//   bb_gather_r_kernel R rows -> CSC value order of m_R;
//   bb_apply_q_kernel  v <- Q^T v (blocks ascending, T^T) or Q v (descending, T), two-segment
//                      gather/scatter per block (SparseQRUtils.h:47-89).
// The chain carries a true dependency from panel to panel, so one workgroup walks it; the panels are
// small (12x8 in the reference's tests, 448x192 in BASELINE configs[2]).
#include "banded_host.h"
#include "qrk_device.h"

#include <float.h>
#include <cstdlib>

namespace qrk {

constexpr int BB_THREADS = 256;
constexpr int BB_WAVES = BB_THREADS / 64;

__device__ __forceinline__ double bb_wave_sum(double v)
{
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off);
    return v;
}

__device__ __forceinline__ double bb_block_sum(double v, double* red)
{
    v = bb_wave_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    double s = 0.0;
#pragma unroll
    for (int w = 0; w < BB_WAVES; ++w) s += red[w];
    return s;
}

// One workgroup walks the whole chain.
//   W      workspace, max_act_rows x max_ncols (column-major, ld = act_rows of the current panel)
//   lo     workspace for the leftover block handed to the next panel
__global__ void __launch_bounds__(BB_THREADS)
bb_chain_kernel(const BBPanel* __restrict__ panels, int num_panels, const int32_t* __restrict__ prowptr,
                const int32_t* __restrict__ pcol, const int64_t* __restrict__ pmap, const double* __restrict__ vals,
                double* __restrict__ W, double* __restrict__ lo, double* __restrict__ y_vals,
                double* __restrict__ t_vals, double* __restrict__ r_stage, int max_act_rows, int max_ncols)
{
    extern __shared__ __attribute__((aligned(16))) double smem[];
    double* xv = smem;                    // [max_act_rows] current Householder vector
    double* hc = xv + max_act_rows;       // [max_ncols] hCoeffs of the panel
    double* uu = hc + max_ncols;          // [max_ncols] row of the T recurrence
    double* red = uu + max_ncols;         // [BB_WAVES]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;

    for (int pi = 0; pi < num_panels; ++pi) {
        const BBPanel p = panels[pi];
        const int m = p.act_rows, n = p.ncols, ld = p.act_rows;

        // ---- Ji = pmat.block(row0, col0, m, n).toDense() (:458, :503) ...
        for (int64_t e = tid; e < (int64_t)m * n; e += BB_THREADS) W[e] = 0.0;
        __syncthreads();
        for (int r = wave; r < m; r += BB_WAVES) {
            const int gr = p.row0 + r;
            for (int e = prowptr[gr] + lane; e < prowptr[gr + 1]; e += 64) {
                const int c = pcol[e] - p.col0;
                if (c >= 0 && c < n) W[(int64_t)c * ld + r] = vals[pmap[e]];
            }
        }
        __syncthreads();
        // ... with its top-left corner replaced by the leftover block of the previous panel (:504-506)
        for (int e = tid; e < p.lo_rows * p.lo_cols; e += BB_THREADS) {
            const int i = e % p.lo_rows, j = e / p.lo_rows;
            W[(int64_t)j * ld + i] = lo[e];
        }
        __syncthreads();

        // ---- Eigen::HouseholderQR of the panel (unblocked; the blocked driver is algebraically the same)
        for (int k = 0; k < n; ++k) {
            double part = 0.0;
            for (int i = k + tid; i < m; i += BB_THREADS) {
                const double v = W[(int64_t)k * ld + i];
                xv[i] = v;
                if (i > k) part = fma(v, v, part);
            }
            const double tailSq = bb_block_sum(part, red);
            const double xk = xv[k];
            double beta, tau, scale;
            if (tailSq <= DBL_MIN) {
                beta = xk; tau = 0.0; scale = 0.0;
            } else {
                const double nrm = sqrt(fma(xk, xk, tailSq));
                beta = xk >= 0.0 ? -nrm : nrm;
                scale = 1.0 / (xk - beta);
                tau = (beta - xk) / beta;
            }
            __syncthreads();
            for (int i = k + 1 + tid; i < m; i += BB_THREADS) {
                const double e = xv[i] * scale;
                xv[i] = e;
                W[(int64_t)k * ld + i] = e;
            }
            if (tid == 0) { W[(int64_t)k * ld + k] = beta; hc[k] = tau; }
            __syncthreads();
            for (int jc = k + 1 + wave; jc < n; jc += BB_WAVES) {
                double* col = W + (int64_t)jc * ld;
                double d = 0.0;
                for (int i = k + 1 + lane; i < m; i += 64) d = fma(xv[i], col[i], d);
                d = bb_wave_sum(d);
                const double tt = tau * (d + col[k]);
                for (int i = k + 1 + lane; i < m; i += 64) col[i] = fma(-tt, xv[i], col[i]);
                if (lane == 0) col[k] -= tt;
            }
            __syncthreads();
        }

        // ---- Y = unit-lower essentials (:471-475)
        double* Y = y_vals + p.y_off;
        for (int64_t e = tid; e < (int64_t)m * n; e += BB_THREADS) {
            const int i = (int)(e % m), j = (int)(e / m);
            Y[e] = i < j ? 0.0 : (i == j ? 1.0 : W[(int64_t)j * ld + i]);
        }
        // ---- T = -make_block_householder_triangular_factor(Y, hCoeffs) (:476-477), rows n-1 .. 0:
        //      T(i, i+1:) = (-h_i Y(i+1:, i)^T Y(i+1:, i+1:)) T(i+1:, i+1:), T(i,i) = h_i
        double* T = t_vals + p.t_off;
        for (int64_t e = tid; e < (int64_t)n * n; e += BB_THREADS) T[e] = 0.0;
        __syncthreads();
        for (int i = n - 1; i >= 0; --i) {
            for (int c = i + 1 + wave; c < n; c += BB_WAVES) {
                // unit-lower view of column c: 1 at row c, essentials below
                double s = 0.0;
                for (int r = c + 1 + lane; r < m; r += 64) s = fma(W[(int64_t)i * ld + r], W[(int64_t)c * ld + r], s);
                s = bb_wave_sum(s);
                if (lane == 0) uu[c] = -hc[i] * (s + W[(int64_t)i * ld + c]);
            }
            __syncthreads();
            for (int c = i + 1 + tid; c < n; c += BB_THREADS) {
                double s = 0.0;
                for (int j = i + 1; j <= c; ++j) s = fma(uu[j], T[(int64_t)c * n + j], s);
                T[(int64_t)c * n + i] = s;
            }
            if (tid == 0) T[(int64_t)i * n + i] = hc[i];
            __syncthreads();
        }
        for (int64_t e = tid; e < (int64_t)n * n; e += BB_THREADS) T[e] = -T[e];

        // ---- rows of R solved by this panel: V = triu(packed QR), explicit zeros kept (:484-491)
        for (int e = tid; e < p.solved * n; e += BB_THREADS) {
            const int br = e % p.solved, bc = e / p.solved;
            r_stage[p.r_off + e] = (br <= bc && br < m) ? W[(int64_t)bc * ld + br] : 0.0;
        }
        // ---- leftover block for the next panel: V.block(lo_from, lo_from, lo_rows, lo_cols) (:505)
        if (pi + 1 < num_panels) {
            const BBPanel q = panels[pi + 1];
            for (int e = tid; e < q.lo_rows * q.lo_cols; e += BB_THREADS) {
                const int i = e % q.lo_rows, j = e / q.lo_rows;
                const int vr = q.lo_from + i, vc = q.lo_from + j;
                lo[e] = (vr <= vc && vr < m && vc < n) ? W[(int64_t)vc * ld + vr] : 0.0;
            }
        }
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------------------------------
// Second version of the chain (panels up to 256 columns): the panel is kept ROW-major and worked on by
// 1024 threads as a 2-D grid, column slot cs = tid % 256, row group rg = tid / 256 (rows interleaved
// modulo 4), so every sweep is coalesced and there is no cross-lane reduction:
//   * Householder QR with the dot and update sweeps FUSED across steps (no pivoting here, the pivot of
//     step k+1 is column k+1): the sweep that applies reflector k to column c also accumulates
//     x'^T c for the next reflector x' = W(:,k+1) - gamma_{k+1} x, two barriers per reflector;
//   * G = Y^T Y on v_mfma_f64_16x16x4_f64 (16x16 tiles, operands from LDS-staged row chunks), written
//     into the T output;
//   * T by the forward (larft) recurrence T(0:c,c) = -h_c T(0:c,0:c) G(0:c,c), in place, T kept packed
//     in LDS when it fits (n <= 192), the negation the reference stores (:477) applied at the end.
// The first version spent ~16 ms per 448x192 panel (BASELINE configs[2] shape), mostly in wave-wide
// reductions: one per column and reflector in the QR, one per (i, c) pair in the T recurrence.
constexpr int BC_THREADS = 1024;
constexpr int BC_CW = 256;                 // most columns of a panel
constexpr int BC_RC = 16;                  // rows per LDS chunk in the Gram matrix
constexpr int BC_NB = 8;                   // columns of a sub-panel of the blocked Householder QR
constexpr int BC_RGT = 4;                  // most row groups in the trailing update (LDS for the partial sums)

__global__ void __launch_bounds__(BC_THREADS)
bb_chain2_kernel(const BBPanel* __restrict__ panels, int num_panels, const int32_t* __restrict__ prowptr,
                 const int32_t* __restrict__ pcol, const int64_t* __restrict__ pmap, const double* __restrict__ vals,
                 double* __restrict__ W, double* __restrict__ lo, double* __restrict__ y_vals,
                 double* __restrict__ t_vals, double* __restrict__ r_stage, int max_act_rows, int max_ncols,
                 int t_in_lds)
{
    extern __shared__ __attribute__((aligned(16))) double smem[];
    double* hc = smem;                         // [BC_CW] hCoeffs of the panel
    double* dpart = hc + BC_CW;                // [BC_THREADS] partial sums of the T recurrence
    double* gs = dpart + BC_THREADS;           // [BC_NB * BC_NB] V^T V of a sub-panel
    double* ts = gs + BC_NB * BC_NB;           // [BC_NB * BC_NB] T of a sub-panel
    double* sc = ts + BC_NB * BC_NB;           // [8] scalars of the current reflector
    double* uni = sc + 8;                      // union: {sub-panel, partial V^T W} / Gram chunk / {packed T, g column}
    double* sp = uni;                          // [max_act_rows * BC_NB] sub-panel, row-major
    double* wpart = sp + (int64_t)max_act_rows * BC_NB;   // [BC_RGT * BC_NB * BC_CW] partial V^T W per row group
    double* ys = uni;                          // [BC_RC * n]
    double* tl = uni;                          // [n (n + 1) / 2] packed upper T by columns
    const int tid = threadIdx.x;
#ifdef QRK_BB_PROF
    unsigned long long pt[6] = {0, 0, 0, 0, 0, 0}, t0 = 0, qt[4] = {0, 0, 0, 0}, q0 = 0;
#define BB_QTICK(n) do { const unsigned long long t1 = __builtin_amdgcn_s_memtime(); qt[n] += t1 - q0; q0 = t1; } while (0)
#define BB_TICK(n) do { const unsigned long long t1 = __builtin_amdgcn_s_memtime(); pt[n] += t1 - t0; t0 = t1; } while (0)
#else
#define BB_TICK(n) do { } while (0)
#define BB_QTICK(n) do { } while (0)
#endif

    for (int pi = 0; pi < num_panels; ++pi) {
        const BBPanel p = panels[pi];
        const int m = p.act_rows, n = p.ncols;          // W is m x n, row-major: W(i, j) = W[i * n + j]
        // 2-D thread grid of this panel: CW column slots (n rounded up to a wave), RG row groups
        const int CW = ((n + 63) / 64) * 64, RG = BC_THREADS / CW;
        const bool on = tid < CW * RG;                  // (threads beyond the grid only help with the copies)
        const int cs = on ? tid % CW : CW - 1, rg = on ? tid / CW : 0;

#ifdef QRK_BB_PROF
        t0 = __builtin_amdgcn_s_memtime();
#endif
        // ---- Ji = pmat.block(row0, col0, m, n).toDense() (:458, :503) ...
        for (int64_t e = tid; e < (int64_t)m * n; e += BC_THREADS) W[e] = 0.0;
        __syncthreads();
        for (int r = tid >> 6; r < m; r += BC_THREADS / 64) {
            const int gr = p.row0 + r;
            for (int e = prowptr[gr] + (tid & 63); e < prowptr[gr + 1]; e += 64) {
                const int c = pcol[e] - p.col0;
                if (c >= 0 && c < n) W[(int64_t)r * n + c] = vals[pmap[e]];
            }
        }
        __syncthreads();
        // ... with its top-left corner replaced by the leftover block of the previous panel (:504-506)
        for (int e = tid; e < p.lo_rows * p.lo_cols; e += BC_THREADS) {
            const int i = e % p.lo_rows, j = e / p.lo_rows;
            W[(int64_t)i * n + j] = lo[e];
        }
        __syncthreads();

        BB_TICK(0);
        // ---- Eigen::HouseholderQR of the panel, blocked: sub-panels of BC_NB columns are factorised in LDS
        // (one wave per column, rows over the lanes), then the block reflector I - V T^T V^T of the sub-panel
        // is applied to the columns to its right in one read-modify-write sweep.  The one-reflector-at-a-time
        // version moved 0.5-1.4 MB per reflector through this CU's L2 port; this one moves it once per
        // BC_NB reflectors.
        for (int jb = 0; jb < n; jb += BC_NB) {
            const int kb = (n - jb) < BC_NB ? (n - jb) : BC_NB;
            const int mr = m - jb;                       // rows jb.. of the panel take part
            const int wv_ = tid >> 6, ln = tid & 63;     // wave = column of the sub-panel
#ifdef QRK_BB_PROF
            q0 = __builtin_amdgcn_s_memtime();
#endif
            // A. sub-panel to LDS, row-major with stride BC_NB
            for (int e = tid; e < mr * BC_NB; e += BC_THREADS) {
                const int i = e / BC_NB, l = e - i * BC_NB;
                sp[e] = l < kb ? W[(int64_t)(jb + i) * n + jb + l] : 0.0;
            }
            __syncthreads();
            BB_QTICK(0);
            // B. Householder QR of the sub-panel in LDS (makeHouseholder + applyHouseholderOnTheLeft,
            //    un-normalised form of bdqr_pair.hip)
            for (int j = 0; j < kb; ++j) {
                if (wv_ == j) {
                    double part = 0.0;
                    for (int i = j + 1 + ln; i < mr; i += 64) { const double v = sp[i * BC_NB + j]; part = fma(v, v, part); }
                    const double tsq = bb_wave_sum(part);
                    const double xk = sp[j * BC_NB + j];
                    double nb_, s2, ng, tau;
                    if (!(tsq > DBL_MIN)) { nb_ = -xk; s2 = 0.0; ng = 0.0; tau = 0.0; }
                    else {
                        const double nrm = sqrt(fma(xk, xk, tsq));
                        nb_ = xk >= 0.0 ? nrm : -nrm;
                        s2 = nb_ + xk;                   // x0 - beta
                        ng = -1.0 / (nb_ * s2);
                        tau = -(s2 * s2) * ng;           // (beta - x0) / beta
                    }
                    if (ln == 0) { sc[0] = s2; sc[1] = ng; sc[2] = -nb_; hc[jb + j] = tau; }
                }
                __syncthreads();
                const double s2 = sc[0], ng = sc[1], betaj = sc[2];   // (read now: the next step's wave rewrites sc)
                if (wv_ > j && wv_ < kb) {
                    const int l = wv_;
                    double part = 0.0;
                    for (int i = j + 1 + ln; i < mr; i += 64) part = fma(sp[i * BC_NB + j], sp[i * BC_NB + l], part);
                    const double d = bb_wave_sum(part);
                    const double ak = sp[j * BC_NB + l];
                    const double ngam = fma(s2, ak, d) * ng;
                    for (int i = j + 1 + ln; i < mr; i += 64) sp[i * BC_NB + l] = fma(ngam, sp[i * BC_NB + j], sp[i * BC_NB + l]);
                    if (ln == 0) sp[j * BC_NB + l] = fma(s2, ngam, ak);       // row j of R
                }
                __syncthreads();
                if (wv_ == j) {
                    const double inv_s = s2 != 0.0 ? 1.0 / s2 : 0.0;
                    for (int i = j + 1 + ln; i < mr; i += 64) sp[i * BC_NB + j] *= inv_s;   // essential part (:471-475)
                    if (ln == 0) sp[j * BC_NB + j] = betaj;                                 // beta
                }
                // (no barrier: the next step reads column j+1 and its wave only touches that column until the
                //  barrier after its scalars; column j is not read again before the barrier below)
            }
            __syncthreads();
            BB_QTICK(1);
            // C. packed sub-panel back to the panel
            for (int e = tid; e < mr * BC_NB; e += BC_THREADS) {
                const int i = e / BC_NB, l = e - i * BC_NB;
                if (l < kb) W[(int64_t)(jb + i) * n + jb + l] = sp[e];
            }
            const int nt = n - jb - kb;                  // columns to the right
            if (nt <= 0) { __syncthreads(); continue; }
            // D. V = unit-lower view of the sub-panel (in place), Ts = larft(V, tau)
            __syncthreads();
            for (int e = tid; e < kb * BC_NB; e += BC_THREADS) {
                const int i = e / BC_NB, l = e - i * BC_NB;
                if (l < kb) { if (i == l) sp[e] = 1.0; else if (i < l) sp[e] = 0.0; }
            }
            __syncthreads();
            for (int pr = wv_; pr < BC_NB * BC_NB; pr += BC_THREADS / 64) {     // G_s = V^T V, one pair per wave pass
                const int a = pr / BC_NB, b2 = pr - a * BC_NB;
                if (a < b2 && b2 < kb) {
                    double part = 0.0;
                    for (int i = b2 + ln; i < mr; i += 64) part = fma(sp[i * BC_NB + a], sp[i * BC_NB + b2], part);
                    part = bb_wave_sum(part);
                    if (ln == 0) gs[pr] = part;
                }
            }
            __syncthreads();
            if (tid < BC_NB) {
                const int a = tid;
                for (int l = 0; l < BC_NB; ++l) ts[a * BC_NB + l] = 0.0;
                for (int l = 0; l < kb; ++l) {
                    const double tau = hc[jb + l];
                    double tv = 0.0;
                    if (a == l) tv = tau;
                    else if (a < l) {
                        double acc = 0.0;
                        for (int b2 = a; b2 < l; ++b2) acc = fma(ts[a * BC_NB + b2], gs[b2 * BC_NB + l], acc);
                        tv = -tau * acc;
                    }
                    ts[a * BC_NB + l] = tv;        // row a only depends on row a
                }
            }
            __syncthreads();
            BB_QTICK(2);
            // E. W(jb:, c) <- (I - V Ts^T V^T) W(jb:, c) for the columns c to the right
            {
                const int CWt = ((nt + 63) / 64) * 64;
                int RGt = BC_THREADS / CWt;                       // row groups: as many as threads and the LDS for
                if (RGt > BC_RGT * BC_CW / CWt) RGt = BC_RGT * BC_CW / CWt;   // the partial sums (wpart) allow
                const bool ont = tid < CWt * RGt;
                const int ct = ont ? tid % CWt : 0, rgt = ont ? tid / CWt : 0;
                const bool colok = ont && ct < nt;
                double* wcol = W + (int64_t)jb * n + jb + kb + ct;        // wcol[i * n] = W(jb + i, jb + kb + ct)
                double w[BC_NB];
#pragma unroll
                for (int l = 0; l < BC_NB; ++l) w[l] = 0.0;
                if (colok) {
                    constexpr int U = 8;
                    for (int i = rgt; i < mr; i += U * RGt) {
                        double xv[U];
#pragma unroll
                        for (int u = 0; u < U; ++u) { const int ii = i + u * RGt; xv[u] = ii < mr ? wcol[(int64_t)ii * n] : 0.0; }
#pragma unroll
                        for (int u = 0; u < U; ++u) {
                            const int ii = i + u * RGt;
                            if (ii < mr) {
#pragma unroll
                                for (int l = 0; l < BC_NB; ++l) w[l] = fma(sp[ii * BC_NB + l], xv[u], w[l]);
                            }
                        }
                    }
                }
                if (ont) {
#pragma unroll
                    for (int l = 0; l < BC_NB; ++l) wpart[(rgt * BC_NB + l) * CWt + ct] = w[l];
                }
                __syncthreads();
                if (colok) {
                    double wt[BC_NB], u2[BC_NB];
#pragma unroll
                    for (int l = 0; l < BC_NB; ++l) {
                        double acc = 0.0;
                        for (int g = 0; g < RGt; ++g) acc += wpart[(g * BC_NB + l) * CWt + ct];
                        wt[l] = acc;
                    }
#pragma unroll
                    for (int a = 0; a < BC_NB; ++a) {          // u = Ts^T w
                        double acc = 0.0;
#pragma unroll
                        for (int b2 = 0; b2 < BC_NB; ++b2) if (b2 <= a) acc = fma(ts[b2 * BC_NB + a], wt[b2], acc);
                        u2[a] = acc;
                    }
                    constexpr int U = 8;
                    for (int i = rgt; i < mr; i += U * RGt) {
                        double xv[U];
#pragma unroll
                        for (int u = 0; u < U; ++u) { const int ii = i + u * RGt; xv[u] = ii < mr ? wcol[(int64_t)ii * n] : 0.0; }
#pragma unroll
                        for (int u = 0; u < U; ++u) {
                            const int ii = i + u * RGt;
                            if (ii < mr) {
                                double v = xv[u];
#pragma unroll
                                for (int l = 0; l < BC_NB; ++l) v = fma(-sp[ii * BC_NB + l], u2[l], v);
                                wcol[(int64_t)ii * n] = v;
                            }
                        }
                    }
                }
            }
            __syncthreads();
            BB_QTICK(3);
        }
        __syncthreads();

        BB_TICK(1);
        // ---- rows of R solved by this panel: V = triu(packed QR), explicit zeros kept (:484-491)
        for (int e = tid; e < p.solved * n; e += BC_THREADS) {
            const int br = e % p.solved, bc = e / p.solved;
            r_stage[p.r_off + e] = (br <= bc && br < m) ? W[(int64_t)br * n + bc] : 0.0;
        }
        // ---- leftover block for the next panel: V.block(lo_from, lo_from, lo_rows, lo_cols) (:505)
        if (pi + 1 < num_panels) {
            const BBPanel q = panels[pi + 1];
            for (int e = tid; e < q.lo_rows * q.lo_cols; e += BC_THREADS) {
                const int i = e % q.lo_rows, j = e / q.lo_rows;
                const int vr = q.lo_from + i, vc = q.lo_from + j;
                lo[e] = (vr <= vc && vr < m && vc < n) ? W[(int64_t)vr * n + vc] : 0.0;
            }
        }
        // ---- Y = unit-lower essentials (:471-475), column-major m x n
        double* Y = y_vals + p.y_off;
        for (int64_t e = tid; e < (int64_t)m * n; e += BC_THREADS) {
            const int i = (int)(e % m), j = (int)(e / m);
            Y[e] = i < j ? 0.0 : (i == j ? 1.0 : W[(int64_t)i * n + j]);
        }
        __syncthreads();

        BB_TICK(2);
        // ---- G = Y^T Y (strict upper part) into the T output: T[c * n + b] = G(b, c), b < c.
        // A real GEMM (n x m by m x n): v_mfma_f64_16x16x4_f64 on 16x16 tiles of G, the operands read from an
        // LDS chunk of 16 rows of Y (unit-lower view of the packed panel); every wave owns up to 9 upper tiles.
        // Lane maps (cdna_hip_programming.md): A[row = l & 15][k = l >> 4], B[k = l >> 4][col = l & 15],
        // D[row = (l >> 4) + 4 j][col = l & 15] in result register j.
        double* T = t_vals + p.t_off;
        {
            typedef double d4 __attribute__((ext_vector_type(4)));
            const int nbt = (n + 15) / 16, n16 = nbt * 16;
            const int ntile = nbt * (nbt + 1) / 2;
            const int wv = tid >> 6, ln = tid & 63;
            constexpr int MAXQ = 9;                      // 16 * 17 / 2 = 136 tiles over 16 waves
            d4 acc[MAXQ];
            int tbi[MAXQ], tbc[MAXQ];
#pragma unroll
            for (int q = 0; q < MAXQ; ++q) {
                acc[q] = d4{0.0, 0.0, 0.0, 0.0};
                int t = wv + q * (BC_THREADS / 64), bi = 0;
                if (t >= ntile) { tbi[q] = -1; tbc[q] = 0; continue; }
                while (t >= nbt - bi) { t -= nbt - bi; ++bi; }   // upper tiles enumerated row by row
                tbi[q] = bi; tbc[q] = bi + t;
            }
            for (int r0 = 0; r0 < m; r0 += BC_RC) {
                __syncthreads();
                for (int e = tid; e < BC_RC * n16; e += BC_THREADS) {
                    const int i = r0 + e / n16, j = e % n16;
                    ys[e] = (i >= m || j >= n || i < j) ? 0.0 : (i == j ? 1.0 : W[(int64_t)i * n + j]);
                }
                __syncthreads();
#pragma unroll
                for (int q = 0; q < MAXQ; ++q) {
                    if (tbi[q] >= 0) {
#pragma unroll
                        for (int ks = 0; ks < BC_RC / 4; ++ks) {
                            const int row = 4 * ks + (ln >> 4);
                            const double av = ys[row * n16 + 16 * tbi[q] + (ln & 15)];
                            const double bv = ys[row * n16 + 16 * tbc[q] + (ln & 15)];
                            acc[q] = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, acc[q], 0, 0, 0);
                        }
                    }
                }
            }
#pragma unroll
            for (int q = 0; q < MAXQ; ++q) {
                if (tbi[q] >= 0) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int gi = 16 * tbi[q] + (ln >> 4) + 4 * j, gc = 16 * tbc[q] + (ln & 15);
                        if (gi < gc && gc < n) T[(int64_t)gc * n + gi] = acc[q][j];
                    }
                }
            }
        }
        __syncthreads();

        BB_TICK(3);
        // ---- T = make_block_householder_triangular_factor(Y, hCoeffs) (:476).
        // With T in LDS: the recursive form of the same factor.  T = [T11 T12; 0 T22] with
        // T12 = -T11 (Y1^T Y2) T22, so the BC_THREADS/64 diagonal blocks are built by the column recurrence, one
        // wave each and without workgroup barriers, and merged pairwise in log2(16) rounds of two triangular
        // products (every thread a few entries).  The column recurrence over the whole panel spent 2.7 us per
        // column in barriers and LDS latency (0.5 ms of a 1.9 ms panel).
        if (t_in_lds) {
            const int wv = tid >> 6, ln = tid & 63;
            constexpr int NBLK = BC_THREADS / 64;
            const int s0 = (n + NBLK - 1) / NBLK;          // <= 16 for n <= 256
            // packed upper storage, by columns: (a, b), a <= b, at b (b + 1) / 2 + a.  G above the diagonal, tau on it
            for (int b = wv; b < n; b += NBLK) {
                const int cb = b * (b + 1) / 2;
                for (int a = ln; a < b; a += 64) tl[cb + a] = T[(int64_t)b * n + a];
                if (ln == 0) tl[cb + b] = hc[b];
            }
            __syncthreads();
            {   // diagonal block wv: t(a, c) = -tau_c sum_{b = a}^{c - 1} T(a, b) G(b, c) inside the block
                const int base = wv * s0, len = (n - base) < s0 ? (n - base) : s0;
                for (int j = 1; j < len; ++j) {
                    const int c = base + j, cc2 = c * (c + 1) / 2;
                    double sum = 0.0;
                    if (ln < j) {
                        const int a = base + ln;
                        int ib = (a * (a + 1)) / 2 + a;          // (a, b = a)
                        for (int b = a; b < c; ++b) { sum = fma(tl[ib], tl[cc2 + b], sum); ib += b + 1; }
                    }
                    const double tc = tl[cc2 + c];
                    __builtin_amdgcn_wave_barrier();
                    if (ln < j) tl[cc2 + base + ln] = -tc * sum;
                    __builtin_amdgcn_wave_barrier();
                }
            }
            __syncthreads();
            constexpr int MAXO = (BC_CW / 2) * (BC_CW / 2) / BC_THREADS;      // entries per thread in the last merge
            for (int sz = s0; sz < n; sz *= 2) {
                const int ss = sz * sz, npair = (n + 2 * sz - 1) / (2 * sz), total = npair * ss;
                double xr[MAXO];
                // X = G12 T22 (in place of G12): X(i, j) = sum_{k <= j} G12(i, k) T22(k, j)
#pragma unroll
                for (int t = 0; t < MAXO; ++t) {
                    const int o = tid + t * BC_THREADS;
                    xr[t] = 0.0;
                    if (o < total) {
                        const int q = o / ss, rem = o - q * ss, j = rem / sz, i = rem - j * sz;
                        const int r0 = 2 * q * sz, c0 = r0 + sz, cj = c0 + j;
                        if (cj < n) {
                            const int cj2 = cj * (cj + 1) / 2 + c0;
                            int ia = c0 * (c0 + 1) / 2 + r0 + i;
                            double acc = 0.0;
#pragma unroll 4
                            for (int k = 0; k <= j; ++k) { acc = fma(tl[ia], tl[cj2 + k], acc); ia += c0 + k + 1; }
                            xr[t] = acc;
                        }
                    }
                }
                __syncthreads();
#pragma unroll
                for (int t = 0; t < MAXO; ++t) {
                    const int o = tid + t * BC_THREADS;
                    if (o < total) {
                        const int q = o / ss, rem = o - q * ss, j = rem / sz, i = rem - j * sz;
                        const int r0 = 2 * q * sz, cj = r0 + sz + j;
                        if (cj < n) tl[cj * (cj + 1) / 2 + r0 + i] = xr[t];
                    }
                }
                __syncthreads();
                // T12 = -T11 X: T12(i, j) = -sum_{k >= i} T11(i, k) X(k, j)
#pragma unroll
                for (int t = 0; t < MAXO; ++t) {
                    const int o = tid + t * BC_THREADS;
                    xr[t] = 0.0;
                    if (o < total) {
                        const int q = o / ss, rem = o - q * ss, j = rem / sz, i = rem - j * sz;
                        const int r0 = 2 * q * sz, cj = r0 + sz + j;
                        if (cj < n) {
                            const int cj2 = cj * (cj + 1) / 2 + r0;
                            const int ri = r0 + i;
                            int ia = ri * (ri + 1) / 2 + ri;     // (ri, ri)
                            double acc = 0.0;
#pragma unroll 4
                            for (int k = i; k < sz; ++k) { acc = fma(tl[ia], tl[cj2 + k], acc); ia += r0 + k + 1; }
                            xr[t] = -acc;
                        }
                    }
                }
                __syncthreads();
#pragma unroll
                for (int t = 0; t < MAXO; ++t) {
                    const int o = tid + t * BC_THREADS;
                    if (o < total) {
                        const int q = o / ss, rem = o - q * ss, j = rem / sz, i = rem - j * sz;
                        const int r0 = 2 * q * sz, cj = r0 + sz + j;
                        if (cj < n) tl[cj * (cj + 1) / 2 + r0 + i] = xr[t];
                    }
                }
                __syncthreads();
            }
        } else {
            // in place in global memory, forward recurrence by columns (panels whose packed T does not fit the LDS)
            for (int cc = 0; cc < n; ++cc) {
                const double hcc = hc[cc];
                double part = 0.0;
                if (on && cs < cc) for (int b = cs + rg; b < cc; b += RG) part = fma(T[(int64_t)b * n + cs], T[(int64_t)cc * n + b], part);
                if (on) dpart[rg * CW + cs] = part;
                __syncthreads();     // every G(b, cc) has been read before column cc is overwritten
                if (on && rg == 0 && cs < cc) {
                    double sum = 0.0;
                    for (int g = 0; g < RG; ++g) sum += dpart[g * CW + cs];
                    T[(int64_t)cc * n + cs] = -hcc * sum;
                }
                if (tid == 0) T[(int64_t)cc * n + cc] = hcc;
                __syncthreads();
            }
        }
        // the reference stores -T (:477); lower part zero
        for (int64_t e = tid; e < (int64_t)n * n; e += BC_THREADS) {
            const int a = (int)(e % n), b = (int)(e / n);
            double v = 0.0;
            if (a <= b) v = t_in_lds ? -tl[(int64_t)b * (b + 1) / 2 + a] : -T[e];
            T[e] = v;
        }
        __syncthreads();
        BB_TICK(4);
#ifdef QRK_BB_PROF
        if (pi == num_panels - 1 && tid == 0) for (int z = 0; z < 5; ++z) T[z] = (double)pt[z];
        if (pi == num_panels - 1 && tid == 0) T[5] = (double)pt[5];
        if (pi == num_panels - 1 && tid == 0) for (int z = 0; z < 4; ++z) T[6 + z] = (double)qt[z];
#endif
    }
}

size_t bb_chain2_smem(int max_act_rows, int max_ncols, int* t_in_lds)
{
    const size_t fixed = (size_t)(BC_CW + BC_THREADS + 2 * BC_NB * BC_NB + 8) * sizeof(double);
    const size_t qr = ((size_t)max_act_rows * BC_NB + (size_t)BC_RGT * BC_NB * BC_CW) * sizeof(double);
    const size_t gram = (size_t)BC_RC * ((max_ncols + 15) / 16 * 16) * sizeof(double);
    const size_t tpk = ((size_t)max_ncols * (max_ncols + 1) / 2 + max_ncols) * sizeof(double);
    size_t uni = qr > gram ? qr : gram;
    *t_in_lds = 0;
    // (QRK_BB_T_GLOBAL forces the in-place T recurrence: lets the tests cover it)
    if (fixed + (tpk > uni ? tpk : uni) <= (size_t)160 * 1024 && !std::getenv("QRK_BB_T_GLOBAL")) { *t_in_lds = 1; if (tpk > uni) uni = tpk; }
    return fixed + uni;
}

__global__ void __launch_bounds__(256)
bb_gather_r_kernel(const double* __restrict__ r_stage, const int64_t* __restrict__ r_src, int64_t nnz,
                   double* __restrict__ r_vals)
{
    const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (i < nnz) r_vals[i] = r_stage[r_src[i]];
}

// One workgroup per right-hand side: seg += Y (T^(T) (Y^T seg)) for every block in order.
__global__ void __launch_bounds__(BB_THREADS)
bb_apply_q_kernel(const BBPanel* __restrict__ panels, int num_panels, const double* __restrict__ y_vals,
                  const double* __restrict__ t_vals, int transpose, double* __restrict__ v, int64_t ldv, int64_t nrhs,
                  int max_act_rows, int max_ncols)
{
    extern __shared__ __attribute__((aligned(16))) double smem[];
    double* seg = smem;                      // [max_act_rows]
    double* w1 = seg + max_act_rows;         // [max_ncols]
    double* w2 = w1 + max_ncols;             // [max_ncols]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int64_t col = blockIdx.x; col < nrhs; col += gridDim.x) {
        double* x = v + col * ldv;
        for (int s = 0; s < num_panels; ++s) {
            const BBPanel p = panels[transpose ? s : num_panels - 1 - s];
            const int m = p.act_rows, n = p.ncols;
            const int seg2 = p.yrow + n + p.num_zeros;     // start of the second row segment
            const double* Y = y_vals + p.y_off;
            const double* T = t_vals + p.t_off;
            for (int i = tid; i < m; i += BB_THREADS) seg[i] = x[i < n ? p.yrow + i : seg2 + (i - n)];
            __syncthreads();
            for (int j = wave; j < n; j += BB_WAVES) {
                double d = 0.0;
                for (int i = lane; i < m; i += 64) d = fma(Y[(int64_t)j * m + i], seg[i], d);
                d = bb_wave_sum(d);
                if (lane == 0) w1[j] = d;
            }
            __syncthreads();
            for (int i = tid; i < n; i += BB_THREADS) {
                double d = 0.0;
                if (transpose) { for (int j = 0; j <= i; ++j) d = fma(T[(int64_t)i * n + j], w1[j], d); }   // T^T w
                else { for (int j = i; j < n; ++j) d = fma(T[(int64_t)j * n + i], w1[j], d); }               // T w
                w2[i] = d;
            }
            __syncthreads();
            for (int i = tid; i < m; i += BB_THREADS) {
                double d = seg[i];
                for (int j = 0; j < n; ++j) d = fma(Y[(int64_t)j * m + i], w2[j], d);
                x[i < n ? p.yrow + i : seg2 + (i - n)] = d;
            }
            __syncthreads();
        }
    }
}

size_t bb_chain_smem(int max_act_rows, int max_ncols) { return (size_t)(max_act_rows + 2 * max_ncols + BB_WAVES) * sizeof(double); }
size_t bb_apply_smem(int max_act_rows, int max_ncols) { return (size_t)(max_act_rows + 2 * max_ncols) * sizeof(double); }

hipError_t launch_bb_chain(const BBPanel* panels, int num_panels, const int32_t* prowptr, const int32_t* pcol,
                           const int64_t* pmap, const double* vals, double* W, double* lo, double* y_vals, double* t_vals,
                           double* r_stage, const int64_t* r_src, int64_t nnz_r, double* r_vals, int max_act_rows,
                           int max_ncols, hipStream_t stream)
{
    int t_in_lds = 0;
    const size_t smem2 = bb_chain2_smem(max_act_rows, max_ncols, &t_in_lds);
    if (max_ncols <= BC_CW && smem2 <= (size_t)160 * 1024 && !std::getenv("QRK_BB_CHAIN_V1")) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(bb_chain2_kernel),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem2);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(bb_chain2_kernel, dim3(1), dim3(BC_THREADS), smem2, stream, panels, num_panels, prowptr, pcol, pmap,
                           vals, W, lo, y_vals, t_vals, r_stage, max_act_rows, max_ncols, t_in_lds);
    } else {
        const size_t smem = bb_chain_smem(max_act_rows, max_ncols);
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(bb_chain_kernel),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(bb_chain_kernel, dim3(1), dim3(BB_THREADS), smem, stream, panels, num_panels, prowptr, pcol, pmap, vals,
                           W, lo, y_vals, t_vals, r_stage, max_act_rows, max_ncols);
    }
    if (nnz_r > 0)
        hipLaunchKernelGGL(bb_gather_r_kernel, dim3((unsigned)((nnz_r + 255) / 256)), dim3(256), 0, stream, r_stage, r_src,
                           nnz_r, r_vals);
    return hipGetLastError();
}

hipError_t launch_bb_apply_q(const BBPanel* panels, int num_panels, const double* y_vals, const double* t_vals,
                             int transpose, double* v, int64_t ldv, int64_t nrhs, int max_act_rows, int max_ncols,
                             hipStream_t stream)
{
    if (nrhs <= 0) return hipSuccess;
    const size_t smem = bb_apply_smem(max_act_rows, max_ncols);
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(bb_apply_q_kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    if (e != hipSuccess) return e;
    const unsigned grid = (unsigned)(nrhs < 1024 ? nrhs : 1024);
    hipLaunchKernelGGL(bb_apply_q_kernel, dim3(grid), dim3(BB_THREADS), smem, stream, panels, num_panels, y_vals, t_vals,
                       transpose, v, ldv, nrhs, max_act_rows, max_ncols);
    return hipGetLastError();
}

}  // namespace qrk
