// bdqr_wave.hip -- one wavefront factorises one tile (rows, cols <= 32) of a
// block-diagonal matrix: A_i P_i = Q_i R_i with explicit Q_i, for gfx950.
//
// Replaces the body of the hot loop of QRKit::BlockDiagonalSparseQR::factorize
// (src/QRKit/BlockDiagonalSparseQR.h:432-526): blockSolver.compute(block) (:437-438,
// Eigen ColPivHouseholderQR / HouseholderQR), Qi = blockSolver.matrixQ() (:446), the
// Q / R value assembly (:455-500) and the column-permutation splice (:519-521).
//
// Mapping (wave64): lanes 0..31 each own one COLUMN of the tile A_i (32 row
// registers, zero padded), lanes 32..63 each own one column of Q_i^T (= one row of
// Q_i), starting from the identity.  Reflector k is the same operation on all 64
// columns, c <- c - tau v (v^T c), so A -> R and I -> Q^T advance together; the
// pivot column v is broadcast through SGPRs with v_readlane, the dot products and
// the rank-1 update are straight FP64 FMA chains over the row registers, and no
// barrier or LDS traffic is needed inside the factorisation.  Columns are never
// physically swapped: each A lane tracks its current position (Eigen's
// m_colsTranspositions bookkeeping), so the "first maximum" tie rule and the
// final permutation are those of Eigen's ColPivHouseholderQR.
// The tile is staged HBM -> LDS -> registers (coalesced 16-B global loads, padded
// LDS columns for the transposing read), and R / Q^T go registers -> LDS -> HBM the
// same way, so every global access is a full-line coalesced access.
#include "qrk_device.h"

#include <float.h>

namespace qrk {

constexpr int WR = 32;        // row registers per lane
constexpr int LDP = WR + 2;   // LDS column stride in doubles: 272 B, conflict-free b64/b128 access

#define QRK_0_31(M)                                                                              \
    M(0) M(1) M(2) M(3) M(4) M(5) M(6) M(7) M(8) M(9) M(10) M(11) M(12) M(13) M(14) M(15) M(16)  \
    M(17) M(18) M(19) M(20) M(21) M(22) M(23) M(24) M(25) M(26) M(27) M(28) M(29) M(30) M(31)
#define QRK_1_31(M)                                                                              \
    M(1) M(2) M(3) M(4) M(5) M(6) M(7) M(8) M(9) M(10) M(11) M(12) M(13) M(14) M(15) M(16)       \
    M(17) M(18) M(19) M(20) M(21) M(22) M(23) M(24) M(25) M(26) M(27) M(28) M(29) M(30) M(31)

// Row loops of reflector K over the row registers K+1..31.  K is a template parameter so that
// every register index is static; the k loop stays rolled and dispatches through a uniform
// switch, which keeps one copy of the per-step scalar code in the instruction cache.
// sqrt(x) and 1/sqrt(x) for a positive normal x: v_rsq_f64 seed, one Goldschmidt iteration and two
// residual corrections (the same scheme hipcc uses for sqrt(); here it also yields the reciprocal
// without an FP64 division).
__device__ __forceinline__ void sqrt_rsqrt(double x, double& s, double& rs)
{
    const double y = __builtin_amdgcn_rsq(x);
    double g = x * y, h = 0.5 * y;
    double e = fma(-h, g, 0.5);
    g = fma(g, e, g);
    h = fma(h, e, h);
    double d = fma(-g, g, x);
    g = fma(d, h, g);
    d = fma(-g, g, x);
    g = fma(d, h, g);
    e = fma(-h, g, 0.5);
    h = fma(h, e, h);
    s = g;
    rs = h + h;
}

// 1/x by v_rcp_f64 + two Newton steps (<= 1-2 ulp), no v_div_* sequence.
__device__ __forceinline__ double recip(double x)
{
    double y = __builtin_amdgcn_rcp(x);
    double e = fma(-x, y, 1.0);
    y = fma(y, e, y);
    e = fma(-x, y, 1.0);
    y = fma(y, e, y);
    return y;
}

template <int K>
__device__ __forceinline__ void reflect(double (&a)[WR], int lb, bool upd, bool ispiv,
                                        double& aknew, double& tau_out)
{
    // pivot column x = column of lane lb, rows K..31, broadcast wave-uniform (SGPR pairs);
    // d = x_tail^T a_tail per lane; on the pivot lane itself d = |x_tail|^2.
    const double ak = a[K];
    const double xk = readlane_f64(ak, lb);
    double x[WR];
    double d = 0.0;
#pragma unroll
    for (int i = K + 1; i < WR; ++i) {
        x[i] = readlane_f64(a[i], lb);
        d = fma(x[i], a[i], d);
    }
    // makeHouseholder (Eigen/src/Householder/Householder.h):
    //   beta = -sign(x0) sqrt(x0^2 + |tail|^2), tau = (beta - x0)/beta, essential = tail/(x0 - beta)
    const double tailSq = readlane_f64(d, lb);
    const bool degen = tailSq <= DBL_MIN;
    double nrm, inv_nrm;
    sqrt_rsqrt(fma(xk, xk, tailSq), nrm, inv_nrm);
    const bool pos0 = xk >= 0.0;
    double beta = pos0 ? -nrm : nrm;
    const double inv_beta = pos0 ? -inv_nrm : inv_nrm;
    const double w = beta - xk;
    double tau = w * inv_beta;
    double scale = -recip(w);                    // essential = tail * scale
    if (degen) { tau = 0.0; beta = xk; scale = 0.0; }
    // applyHouseholderOnTheLeft: tmp = ess^T bottom + row0; row0 -= tau tmp; bottom -= (tau ess) tmp
    const double tmp = fma(scale, d, ak);
    const double tt = upd ? tau * tmp : 0.0;
    double an = ak - tt;
    if (ispiv) an = beta;
    const double ncoef = -(tt * scale);
    a[K] = an;
#pragma unroll
    for (int i = K + 1; i < WR; ++i) a[i] = fma(ncoef, x[i], a[i]);
    aknew = an;
    tau_out = tau;
}

// Inverse of e = p(p+1)/2 + i (0 <= i <= p): position in the packed upper triangle by columns.
__device__ __forceinline__ void tri_unpack(int e, int& p, int& i)
{
    int q = (int)((sqrtf(8.0f * (float)e + 1.0f) - 1.0f) * 0.5f);
    if ((q + 1) * (q + 2) / 2 <= e) ++q;
    if (q * (q + 1) / 2 > e) --q;
    p = q;
    i = e - q * (q + 1) / 2;
}

template <int K>
__device__ __forceinline__ double tail_sqnorm(const double (&a)[WR])
{
    double s = 0.0;
#pragma unroll
    for (int i = K + 1; i < WR; ++i) s = fma(a[i], a[i], s);
    return s;
}

// One step of ColPivHouseholderQR::computeInPlace (Eigen/src/QR/ColPivHouseholderQR.h) on the
// wave-resident tile: pivot search, reflector, trailing update (also of Q^T), norm downdate.
//
// Column norms are tracked SQUARED (nu2 = m_colNormsUpdated^2, thr_nd2 = sqrt(eps) * m_colNormsDirect^2):
// Eigen's  temp = (1+t)(1-t), t = |a_kj|/normUpd;  normUpd *= sqrt(temp)  is  nu2 <- max(nu2 - a_kj^2, 0),
// and its recompute test  temp * (normUpd/normDir)^2 <= sqrt(eps)  is  nu2_new <= sqrt(eps) * normDir^2:
// the same quantities without two FP64 divisions and a square root per step; the pivot (first maximum)
// is the same column because squaring is monotone.
template <int K>
__device__ __forceinline__ void factor_step_k(double (&a)[WR], double* rrows, int pivoting, int lane,
                                              bool isA, bool& live, int& pos, double& nu2,
                                              double& thr_nd2)
{
    // ---- pivot: first maximum of the updated norms over positions K..c-1.  Non-negative doubles
    // order like their bit patterns, so the wave max is an integer max on (hi, lo).
    int lb;
    if (pivoting) {
        const int khi = live ? __double2hiint(nu2) : (int)0x80000000;
        const unsigned klo = (unsigned)__double2loint(nu2);
        const int mh = half32_max_i32(khi);
        const unsigned ml = half32_max_u32(khi == mh ? klo : 0u);
        unsigned long long tie = __ballot(live && khi == mh && klo == ml);
        if (__popcll(tie) > 1) {
            // exact tie: Eigen takes the first maximum = smallest CURRENT position
            const int pc = ((tie >> lane) & 1ull) ? pos : 64;
            const int pmin = half32_min_i32(pc);
            tie = __ballot(pc == pmin);
        }
        lb = __builtin_amdgcn_readfirstlane(__ffsll((long long)tie) - 1);
        const int bpos = __builtin_amdgcn_readlane(pos, lb);
        if (lane == lb) pos = K;
        else if (live && pos == K) pos = bpos;
    } else {
        lb = K;
    }
    const bool ispiv = lane == lb;
    if (ispiv) live = false;

    // ---- reflector K on all 64 columns (A -> R, I -> Q^T); chosen columns keep their R
    //      entries untouched (upd == false gives a zero coefficient).
    const bool upd = !isA || live;
    double aknew, tau;
    reflect<K>(a, lb, upd, ispiv, aknew, tau);
    if (ispiv) rrows[WR * WR + 16 + K] = tau;   // hcoeffs[K], parked next to lane_of_pos
    // Row K of R is final now: park it in LDS (row-major by ORIGINAL column) so that the row
    // register is dead from here on; the epilogue gathers it through the final permutation.
    if (isA && (live || ispiv)) rrows[K * WR + lane] = aknew;

    // ---- LAWN-176 norm downdate for the remaining columns (squared form, see above)
    if (pivoting) {
        double nn = fma(-aknew, aknew, nu2);
        nn = nn > 0.0 ? nn : 0.0;
        const bool need = live && nn <= thr_nd2;
        nu2 = nn;
        if (__any(need)) {
            const double s = tail_sqnorm<K>(a);
            if (need) { nu2 = s; thr_nd2 = s * 1.4901161193847656e-08; }   // sqrt(DBL_EPSILON)
        }
    }
}

__device__ __forceinline__ void factor_step(int k, double (&a)[WR], double* rrows, int pivoting,
                                            int lane, bool isA, bool& live, int& pos, double& nu2,
                                            double& thr_nd2)
{
    // k is a compile-time constant after unrolling; the switch folds to one case.
    switch (k) {
#define QRK_STEP(K) case K: factor_step_k<K>(a, rrows, pivoting, lane, isA, live, pos, nu2, thr_nd2); break;
        QRK_0_31(QRK_STEP)
#undef QRK_STEP
        default: break;
    }
}

// FULL32: every tile is 32x32 and all arrays are 16-byte aligned (uniform batch).
template <bool FULL32>
__global__ void __launch_bounds__(64, 4)
bdqr_wave_kernel(WaveBatch nb, const double* __restrict__ tiles, double* __restrict__ q_vals,
                 double* __restrict__ r_vals, int32_t* __restrict__ perm,
                 double* __restrict__ hcoeffs)
{
    __shared__ __attribute__((aligned(16))) double lds[WR * LDP];
    for (int64_t t = blockIdx.x; t < nb.num_tiles; t += gridDim.x) {
        // Re-derive the lane id per tile behind an opaque barrier: otherwise hipcc hoists the
        // 32 identity-column constants and the LDS addresses out of the tile loop and spills them.
        int lane = threadIdx.x;
        asm volatile("" : "+v"(lane));
        const bool isA = lane < 32;
        const int col = lane & 31;
        int r, c, cbase;
        int64_t toff, qoff, roff;
        if (FULL32) {
            r = 32; c = 32;
            toff = t * 1024; qoff = t * 1024; roff = t * 528; cbase = (int)(t * 32);
        } else if (nb.tile_ids) {
            const int g = nb.tile_ids[t];
            r = nb.t_rows[g]; c = nb.t_cols[g];
            toff = nb.t_off[g]; qoff = nb.q_off[g]; roff = nb.r_off[g]; cbase = nb.c_off[g];
        } else {
            r = nb.rows; c = nb.cols;
            toff = t * r * c; qoff = t * (int64_t)r * r; roff = t * (int64_t)(c * (c + 1) / 2);
            cbase = (int)(t * c);
        }

        // ---- stage the tile: coalesced global read, LDS image with padded columns
        if (FULL32) {
            const double2* src = reinterpret_cast<const double2*>(tiles + toff);
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const int e2 = lane + 64 * q;           // double2 index, 16 per column
                const double2 v = src[e2];
                *reinterpret_cast<double2*>(&lds[(e2 >> 4) * LDP + ((e2 & 15) << 1)]) = v;
            }
        } else {
            const double* src = tiles + toff;
            const int n_in = r * c;
            for (int e = lane; e < n_in; e += 64) {
                const int cc = e / r;
                lds[cc * LDP + (e - cc * r)] = src[e];
            }
        }
        __syncthreads();

        double a[WR];
        if (isA) {
            if (FULL32) {
#pragma unroll
                for (int i = 0; i < WR; ++i) a[i] = lds[col * LDP + i];
            } else {
#pragma unroll
                for (int i = 0; i < WR; ++i) a[i] = (col < c && i < r) ? lds[col * LDP + i] : 0.0;
            }
        } else {
#pragma unroll
            for (int i = 0; i < WR; ++i) a[i] = (i == col && col < r) ? 1.0 : 0.0;
        }
        __syncthreads();   // the LDS image is reused for the outputs

        // ---- squared column norms (ColPivHouseholderQR: m_colNormsUpdated^2, sqrt(eps) m_colNormsDirect^2)
        double nu2, thr_nd2;
        {
            double s = 0.0;
#pragma unroll
            for (int i = 0; i < WR; ++i) s = fma(a[i], a[i], s);
            nu2 = s;
            thr_nd2 = s * 1.4901161193847656e-08;
        }
        bool live = isA && col < c;   // A column not yet chosen as a pivot
        int pos = col;                // current position of this column (Eigen swaps columns)

        // The k loop is fully unrolled: every row-register index is a compile-time constant.
        // (A rolled loop dispatching through a uniform switch makes hipcc's CFG structurizer
        // copy the whole register tile at every merge point.)
#pragma unroll
        for (int k = 0; k < WR; ++k) {
            if (FULL32 || k < c)
                factor_step(k, a, lds, nb.pivoting, lane, isA, live, pos, nu2, thr_nd2);
        }

        // ---- R: lds[i*32 + l] holds R(i, final position of original column l).  The packed upper
        // triangle by columns is exactly the CSC value order of m_R (BlockDiagonalSparseQR.h:475-479):
        // element e -> (column p, row i), gathered through lane_of_pos[p].
        int* lane_of_pos = reinterpret_cast<int*>(&lds[WR * WR]);
        if (isA && col < c) {
            lane_of_pos[pos] = col;
            perm[cbase + pos] = cbase + col;     // m_outputPerm_c.indices()(base_col+j) (:519-521)
        }
        __syncthreads();
        if (hcoeffs && lane < c) hcoeffs[cbase + lane] = lds[WR * WR + 16 + lane];
        if (FULL32) {
            double2* dst = reinterpret_cast<double2*>(r_vals + roff);
#pragma unroll
            for (int q = 0; q < 5; ++q) {
                const int e2 = lane + 64 * q;
                if (e2 < 264) {
                    int p0, i0, p1, i1;
                    tri_unpack(2 * e2, p0, i0);
                    tri_unpack(2 * e2 + 1, p1, i1);
                    dst[e2] = make_double2(lds[i0 * WR + lane_of_pos[p0]], lds[i1 * WR + lane_of_pos[p1]]);
                }
            }
        } else {
            const int n_r = c * (c + 1) / 2;
            for (int e = lane; e < n_r; e += 64) {
                int p0, i0;
                tri_unpack(e, p0, i0);
                r_vals[roff + e] = lds[i0 * WR + lane_of_pos[p0]];
            }
        }
        __syncthreads();

        // ---- Q: lane 32+j holds row j of Q_i; row-major rows are the CSR value order of m_Q
        // in both FullQ ([U|N] split, :455-471) and BlockDiagonalQ (:480-492) layouts.
        if (!isA && col < r) {
            if (FULL32) {
#pragma unroll
                for (int i = 0; i < WR; i += 2)
                    *reinterpret_cast<double2*>(&lds[col * LDP + i]) = make_double2(a[i], a[i + 1]);
            } else {
#pragma unroll
                for (int i = 0; i < WR; ++i)
                    if (i < r) lds[col * LDP + i] = a[i];
            }
        }
        __syncthreads();
        if (FULL32) {
            double2* dst = reinterpret_cast<double2*>(q_vals + qoff);
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const int e2 = lane + 64 * q;
                dst[e2] = *reinterpret_cast<const double2*>(&lds[(e2 >> 4) * LDP + ((e2 & 15) << 1)]);
            }
        } else {
            const int n_q = r * r;
            for (int e = lane; e < n_q; e += 64) {
                const int j = e / r;
                q_vals[qoff + e] = lds[j * LDP + (e - j * r)];
            }
        }
        __syncthreads();
    }
}

void launch_bdqr_wave(const WaveBatch& nb, bool full32, const double* tiles, double* q_vals,
                      double* r_vals, int32_t* perm, double* hcoeffs, int max_blocks,
                      hipStream_t stream)
{
    if (nb.num_tiles <= 0) return;
    const int64_t want = nb.num_tiles < (int64_t)max_blocks ? nb.num_tiles : (int64_t)max_blocks;
    const dim3 grid((unsigned)want), block(64);
    if (full32)
        hipLaunchKernelGGL(bdqr_wave_kernel<true>, grid, block, 0, stream, nb, tiles, q_vals, r_vals,
                           perm, hcoeffs);
    else
        hipLaunchKernelGGL(bdqr_wave_kernel<false>, grid, block, 0, stream, nb, tiles, q_vals, r_vals,
                           perm, hcoeffs);
}

}  // namespace qrk
