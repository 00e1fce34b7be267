// bdqr_small.hip -- 64/G tiles per wavefront for uniform batches of SMALL tiles (rows <= G, G = 4, 8 or 16):
// the shapes the reference itself runs -- 7x2 (test/test-qrkit.cpp:49-51), 9x2 LM-damped (test-utils.cpp:254-274),
// 6x6 / 8x6 (BASELINE block-angular left part) -- for gfx950.
// Since round 5 the default for these shapes is elsewhere -- 1 or 2 columns: bdqr_thin.hip (round 2), 5 .. 16 rows: bdqr_quad.hip
// (round 5) -- and this kernel runs the tiles of at most 4 rows with 3 or 4 columns; QRK_THIN=0 / QRK_QUAD=0 bring the others back
// here (the cross-checks of tests/test_quad_gpu.py and tests/test_small_tiles_gpu.py).
//
// Replaces, for those shapes, the body of the hot loop of QRKit::BlockDiagonalSparseQR::factorize
// (src/QRKit/BlockDiagonalSparseQR.h:432-526): blockSolver.compute(block) (:437-438, Eigen ColPivHouseholderQR /
// HouseholderQR), Qi = blockSolver.matrixQ() (:446), the Q / R value assembly (:455-500) and the column-permutation
// splice (:519-521).
//
// The pair kernel (bdqr_pair.hip) gives every tile 32 lanes and 32 row registers whatever its size: an 8x6 tile uses 6
// of the 32 lanes and 8 of the 32 registers, and the batch runs at 5 % of the HBM roofline.  Here a group of G lanes
// owns a tile: lane j of the group holds column j of A (G row registers, zero padded) and column j of Q^T (= row j of
// Q), as in the pair kernel, and reflector k is the same operation on both.  With at most 16 rows everything the pair
// kernel needs LDS for stays in the lanes: the pivot column reaches the group by ds_bpermute of the row registers
// (2(G-k) of them at step k), the reductions over a group are DPP steps inside a row of 16 lanes.  The arithmetic is
// the pair kernel's (squared column norms with the LAWN-176 downdate and recompute rule, first maximum = smallest
// CURRENT position among exact ties, un-normalised reflector with one rsq and one rcp per step, Eigen's degenerate case).
// Column positions are tracked explicitly here (one integer per lane), so ties need no replay.
// Like the pair kernel, it takes a data-dependent decision only when the decision is clear of rounding (bdqr_pair.hip,
// "Decisions and the exact path": pivot margin, recompute band, degenerate reflector, sign of beta, pivot at the noise level)
// and otherwise appends the tile to the redo list of the exact path (bdqr_exact.hip).
//
// I/O is staged through LDS per workgroup of 4 waves: the 4 * 64/G tiles of a workgroup are one contiguous run of the
// tile array, of q_vals (row-major Q_i = CSR order of m_Q, :455-492) and of r_vals (packed upper triangle by columns =
// CSC order of m_R, :475-479), so every global access is a coalesced sweep.
#include "qrk_device.h"

#include <float.h>

#ifndef QRK_SMALL_WAVES8
#define QRK_SMALL_WAVES8 4      // ... and the G = 4 / G = 8 instantiations
#endif
#ifndef QRK_SMALL_WAVES16
#define QRK_SMALL_WAVES16 3     // waves per SIMD the G = 16 instantiation is compiled for
#endif

namespace qrk {

namespace small {

constexpr double SQRT_EPS = 1.4901161193847656e-08;   // sqrt(DBL_EPSILON), Eigen's norm_downdate_threshold
// decision margins: the same constants as bdqr_pair.hip
constexpr double MREL = 0.000244140625;               // 2^-12 = 2^14 eps / sqrt(eps)
constexpr double THR_HI = SQRT_EPS * (1.0 + MREL);
constexpr int SCALE_SHIFT = 60 << 20;                 // scale_hi = high word of 2^-60 |A|^2: a pivot below 2^-30 |A| flags
constexpr int X0_SHIFT = 26 << 20;                    // x0^2 <= 2^-86 |A|^2 = (2^9 eps |A|)^2

// sqrt(x) for a positive normal x: v_rsq_f64 seed, one Goldschmidt iteration and one residual correction (<= 1 ulp)
__device__ __forceinline__ double sqrt_pos(double x)
{
    const double y = __builtin_amdgcn_rsq(x);
    double g = x * y, h = 0.5 * y;
    const double e = fma(-h, g, 0.5);
    g = fma(g, e, g);
    h = fma(h, e, h);
    const double d = fma(-g, g, x);
    return fma(d, h, g);
}

// 1/x by v_rcp_f64 + two Newton steps (<= 1 ulp)
__device__ __forceinline__ double recip(double x)
{
    double y = __builtin_amdgcn_rcp(x);
    double e = fma(-x, y, 1.0);
    y = fma(y, e, y);
    e = fma(-x, y, 1.0);
    y = fma(y, e, y);
    return y;
}

__device__ __forceinline__ double dpp_f64_xor(double v, int stage)
{
    int lo = __double2loint(v), hi = __double2hiint(v);
    switch (stage) {
    case 0: lo = dpp_i32<0xB1>(lo); hi = dpp_i32<0xB1>(hi); break;      // quad_perm [1,0,3,2]
    case 1: lo = dpp_i32<0x4E>(lo); hi = dpp_i32<0x4E>(hi); break;      // quad_perm [2,3,0,1]
    case 2: lo = dpp_i32<0x141>(lo); hi = dpp_i32<0x141>(hi); break;    // row_half_mirror
    default: lo = dpp_i32<0x140>(lo); hi = dpp_i32<0x140>(hi); break;   // row_mirror
    }
    return __hiloint2double(hi, lo);
}

// max / min over the G lanes of a group (G = 4, 8, 16: the groups are aligned inside a DPP row of 16 lanes)
template <int G>
__device__ __forceinline__ double group_max_f64(double v)
{
    v = fmax(v, dpp_f64_xor(v, 0));
    v = fmax(v, dpp_f64_xor(v, 1));
    if (G >= 8) v = fmax(v, dpp_f64_xor(v, 2));
    if (G >= 16) v = fmax(v, dpp_f64_xor(v, 3));
    return v;
}
template <int G>
__device__ __forceinline__ int group_min_i32(int v)
{
    v = min(v, dpp_i32<0xB1>(v));
    v = min(v, dpp_i32<0x4E>(v));
    if (G >= 8) v = min(v, dpp_i32<0x141>(v));
    if (G >= 16) v = min(v, dpp_i32<0x140>(v));
    return v;
}

__device__ __forceinline__ double bperm_f64(int byte_addr, double v)
{
    const int lo = __builtin_amdgcn_ds_bpermute(byte_addr, __double2loint(v));
    const int hi = __builtin_amdgcn_ds_bpermute(byte_addr, __double2hiint(v));
    return __hiloint2double(hi, lo);
}

}  // namespace small

// One workgroup = 4 waves = TW = 4 * 64/G tiles.  PIVOT: ColPivHouseholderQR (else HouseholderQR).
template <int G, bool PIVOT>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(G == 16 ? QRK_SMALL_WAVES16 : QRK_SMALL_WAVES8)))
bdqr_small_kernel(int64_t num_tiles, int r, int c, const double* __restrict__ tiles, double* __restrict__ q_vals,
                  double* __restrict__ r_vals, int32_t* __restrict__ perm, double* __restrict__ hcoeffs,
                  int32_t* __restrict__ redo_count, int32_t* __restrict__ redo_ids)
{
    using namespace small;
    constexpr int TW = 256 / G;                 // tiles per workgroup
    __shared__ double buf[TW * G * G];          // staging: r*c, then c(c+1)/2, then r*r doubles per tile
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int j = tid & (G - 1);                // column of A / row of Q owned by this lane
    const int tl = tid / G;                     // tile of the workgroup
    const int gaddr = (lane & ~(G - 1)) << 2;   // ds_bpermute byte address of lane 0 of this group
    const int rc = r * c, rr = r * r, nr = c * (c + 1) / 2;

    for (int64_t t0 = (int64_t)blockIdx.x * TW; t0 < num_tiles; t0 += (int64_t)gridDim.x * TW) {
        const int nt = num_tiles - t0 < TW ? (int)(num_tiles - t0) : TW;
        // ---- tiles in: one coalesced sweep
        {
            // (all the loads of a thread in flight before the first wait; clamped addresses, not predicated loads: bdqr_thin.hip)
            const double* src = tiles + t0 * rc;
            const int n = nt * rc;
            for (int base = tid; base < n; base += 256 * 8) {
                double v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) { const int e = base + 256 * u; v[u] = QRK_TILE_LOAD(src + (e < n ? e : n - 1)); }
#pragma unroll
                for (int u = 0; u < 8; ++u) { const int e = base + 256 * u; if (e < n) buf[e] = v[u]; }
            }
        }
        __syncthreads();
        const bool valid = tl < nt;
        double a[G], q[G];
#pragma unroll
        for (int i = 0; i < G; ++i) {
            a[i] = (valid && j < c && i < r) ? buf[tl * rc + j * r + i] : 0.0;
            q[i] = (i == j && j < r) ? 1.0 : 0.0;
        }
        __syncthreads();

        bool live = j < c;
        int pos = j;                    // current position of this column (Eigen swaps columns; here only the index moves)
        int kstep = 2 * G;              // step at which this column was chosen = its final position
        double nu2, thr_nd2;            // m_colNormsUpdated^2 and sqrt(eps) (1 + 2^-12) * m_colNormsDirect^2
        bool flag = false;              // a decision of this lane's tile was not clear of rounding
        int scale_hi = 0;               // high word of 2^-60 |A|^2 (|A|^2 = squared norm of the first pivot column)
        {
            double s = 0.0;
#pragma unroll
            for (int i = 0; i < G; ++i) s = fma(a[i], a[i], s);
            nu2 = s;
            thr_nd2 = s * THR_HI;
        }

#pragma unroll
        for (int K = 0; K < G; ++K) {
            if (K < c) {    // uniform: c steps (rows >= cols)
                // ---- pivot: first maximum of the updated norms among the columns at positions >= K
                bool ispiv;
                int pl;         // pivot lane of the group
                if (PIVOT) {
                    const double key = live ? nu2 : -1.0;
                    const double gmax = group_max_f64<G>(key);
                    const bool cand = live && key == gmax;
                    const int pmin = group_min_i32<G>(cand ? pos : 4 * G);
                    ispiv = cand && pos == pmin;
                    const unsigned long long bm = __builtin_amdgcn_ballot_w64(ispiv);
                    const unsigned gm = (unsigned)(bm >> (lane & ~(G - 1))) & ((1u << G) - 1u);
                    pl = gm ? __builtin_ctz(gm) : 0;
                    // decision (1): candidates = live columns whose high word is within one unit of the maximum; more than one
                    // in some group of the wave (rare) -> check the margin there
                    const bool cand2 = live && __double2hiint(key) >= __double2hiint(gmax) - 1;
                    const unsigned long long bm2 = __builtin_amdgcn_ballot_w64(cand2);
                    const unsigned gm2 = (unsigned)(bm2 >> (lane & ~(G - 1))) & ((1u << G) - 1u);
                    if (__builtin_amdgcn_ballot_w64((gm2 & (gm2 - 1u)) != 0u) != 0ull) {
                        const double thrb = bperm_f64(gaddr + (pl << 2), thr_nd2);
                        double margin = MREL * (thr_nd2 + thrb);
                        if (K > 0) margin += 4.547473508864641e-13 /* 2^-41 */ *
                                             __builtin_sqrt(__hiloint2double(scale_hi + SCALE_SHIFT, 0) * (gmax > 0.0 ? gmax : 0.0));
                        flag = flag || (live && !ispiv && nu2 >= gmax - margin);
                    }
                    // the column at position K and the chosen one trade places (m_qr.col(k).swap(m_qr.col(biggest)))
                    if (ispiv) pos = K; else if (pos == K) pos = pmin;
                } else {
                    ispiv = j == K;
                    pl = K;
                }
                if (ispiv) { live = false; kstep = K; }

                // ---- pivot column to every lane of the group
                const int src = gaddr + (pl << 2);
                double x[G];
#pragma unroll
                for (int i = K; i < G; ++i) x[i] = bperm_f64(src, a[i]);
                const double xk = x[K];
                double tailSq = 0.0, dA = 0.0, dQ = 0.0;
#pragma unroll
                for (int i = K + 1; i < G; ++i) {
                    tailSq = fma(x[i], x[i], tailSq);
                    dA = fma(x[i], a[i], dA);
                    dQ = fma(x[i], q[i], dQ);
                }
                // ---- makeHouseholder + applyHouseholderOnTheLeft (Eigen/src/Householder/Householder.h), un-normalised:
                // beta = -sign(x0) sqrt(x0^2 + |tail|^2), w = beta - x0, tau = w/beta, essential = -tail/w,
                // gamma = (d - w c_k)/(beta w): c_k += w gamma, c_i -= gamma x_i.  nb = -beta, s = -w, ng = -1/(beta w).
                const double nrm = sqrt_pos(fma(xk, xk, tailSq));
                // Eigen: if (c0 >= 0) beta = -beta; -0.0 counts as >= 0, hence the + 0.0
                const double nb = __hiloint2double((__double2hiint(nrm) & 0x7fffffff) | (__double2hiint(xk + 0.0) & (int)0x80000000),
                                                   __double2loint(nrm));
                double s = nb + xk;
                double ng = -recip(nb * s);
                bool setdiag = ispiv;
                {
                    // decisions (3), (4), (5) (bdqr_pair.hip): degenerate reflector on a non-empty tail, first entry too small
                    // to fix the sign of beta, pivot column at the noise level of the tile
                    const double nrm2 = fma(xk, xk, tailSq);
                    if (K == 0) scale_hi = __double2hiint(nrm2) - SCALE_SHIFT;
                    const bool tail = K + 1 < r;
                    const bool tiny_x0 = __double2hiint(xk * xk) + X0_SHIFT <= scale_hi;
                    const bool tiny_x = PIVOT && __double2hiint(nrm2) <= scale_hi;
                    flag = flag || (tail && (!(tailSq > DBL_MIN) || tiny_x0)) || tiny_x;
                }
                if (!(tailSq > DBL_MIN)) { ng = 0.0; s = 0.0; setdiag = false; }   // tau = 0, beta = x0, H = I
                if (hcoeffs && ispiv && valid) hcoeffs[(t0 + tl) * c + K] = -(s * s) * ng;   // tau = w^2/(beta w)
                const double ngA = fma(s, a[K], dA) * ng;
                double an = fma(s, ngA, a[K]);
                if (setdiag) an = -nb;                       // R(k,k) = beta
                a[K] = an;                                   // row K of R is final
                const double ngQ = fma(s, q[K], dQ) * ng;
                q[K] = fma(s, ngQ, q[K]);
#pragma unroll
                for (int i = K + 1; i < G; ++i) {
                    a[i] = fma(ngA, x[i], a[i]);
                    q[i] = fma(ngQ, x[i], q[i]);
                }
                // ---- LAWN-176 norm downdate, squared form (see bdqr_pair.hip); recompute from the updated column
                if (PIVOT && K + 1 < G) {
                    const double nn = fma(-an, an, nu2);
                    nu2 = nn;
                    if (live && nn <= thr_nd2) {
                        flag = flag || nn > thr_nd2 * (1.0 - 2.0 * MREL) ||   // decision (2): inside the band around the threshold, or a
                               __double2hiint(thr_nd2) <= scale_hi;   // column below 2^-17 |A| (thr = 2^-26 nd^2 <= 2^-60 |A|^2)
                        double sq = 0.0;
#pragma unroll
                        for (int i = K + 1; i < G; ++i) sq = fma(a[i], a[i], sq);
                        nu2 = sq;
                        thr_nd2 = sq * THR_HI;
                    }
                }
            }
        }

        // ---- a decision inside its error margin: the tile is redone by the exact path
        {
            const unsigned long long fm = __builtin_amdgcn_ballot_w64(flag && valid);
            const unsigned gf = (unsigned)(fm >> (lane & ~(G - 1))) & ((1u << G) - 1u);
            if (redo_count && gf != 0u && j == 0) redo_ids[atomicAdd(redo_count, 1)] = (int32_t)(t0 + tl);
        }
        // ---- R (packed upper triangle by columns, in final column order) and the permutation
        if (valid && j < c) {
#pragma unroll
            for (int i = 0; i < G; ++i)
                if (i <= kstep) buf[tl * nr + kstep * (kstep + 1) / 2 + i] = a[i];
            perm[(t0 + tl) * c + kstep] = (int32_t)((t0 + tl) * c + j);   // m_outputPerm_c.indices()(base_col+k) (:519-521)
        }
        __syncthreads();
        {
            double* dst = r_vals + t0 * nr;
            for (int e = tid; e < nt * nr; e += 256) QRK_OUT_STORE(dst + e, buf[e]);
        }
        __syncthreads();
        // ---- Q: lane j holds row j of Q_i
        if (valid && j < r) {
#pragma unroll
            for (int i = 0; i < G; ++i)
                if (i < r) buf[tl * rr + j * r + i] = q[i];
        }
        __syncthreads();
        {
            double* dst = q_vals + t0 * rr;
            for (int e = tid; e < nt * rr; e += 256) QRK_OUT_STORE(dst + e, buf[e]);
        }
        __syncthreads();
    }
}

// rows <= 16, cols <= rows, uniform batch.
void launch_bdqr_small(int64_t num_tiles, int r, int c, int pivoting, const double* tiles, double* q_vals, double* r_vals,
                       int32_t* perm, double* hcoeffs, int max_blocks, int32_t* redo_count, int32_t* redo_ids, hipStream_t stream)
{
    if (num_tiles <= 0) return;
    const int G = r <= 4 ? 4 : (r <= 8 ? 8 : 16);
    const int tw = 256 / G;
    int64_t nwg = (num_tiles + tw - 1) / tw;
    if (max_blocks > 0 && nwg > max_blocks) nwg = max_blocks;
    const dim3 grid((unsigned)nwg), block(256);
#define QRK_SMALL(GG, P) \
    hipLaunchKernelGGL((bdqr_small_kernel<GG, P>), grid, block, 0, stream, num_tiles, r, c, tiles, q_vals, r_vals, perm, hcoeffs, redo_count, redo_ids)
    if (pivoting) {
        if (G == 4) QRK_SMALL(4, true); else if (G == 8) QRK_SMALL(8, true); else QRK_SMALL(16, true);
    } else {
        if (G == 4) QRK_SMALL(4, false); else if (G == 8) QRK_SMALL(8, false); else QRK_SMALL(16, false);
    }
#undef QRK_SMALL
}

}  // namespace qrk
