// bdqr_wave.hip -- one wavefront factorises one tile (rows, cols <= 32) of a
// block-diagonal matrix: A_i P_i = Q_i R_i with explicit Q_i, for gfx950.
//
// Replaces the body of the hot loop of QRKit::BlockDiagonalSparseQR::factorize
// (src/QRKit/BlockDiagonalSparseQR.h:432-526): blockSolver.compute(block) (:437-438,
// Eigen ColPivHouseholderQR / HouseholderQR), Qi = blockSolver.matrixQ() (:446), the
// Q / R value assembly (:455-500) and the column-permutation splice (:519-521).
//
// Mapping (wave64): lanes 0..31 each own one COLUMN of the tile A_i (32 row registers, zero
// padded), lanes 32..63 each own one column of Q_i^T (= one row of Q_i), starting from the
// identity.  Reflector k is the same operation on all 64 columns, c <- c - tau v (v^T c), so
// A -> R and I -> Q^T advance together as FP64 FMA chains over the row registers.
//
// The pivot column v has to reach every lane.  v_readlane (VGPR -> SGPR) costs ~7 SIMD cycles per
// dword on gfx950 and ds_bpermute ~8 ns, but an LDS read of ONE address by all lanes is almost
// free (tools/ubench2.hip).  So the wave keeps a column-major image of A in LDS (the staging image
// of the load, refreshed by the owning lanes every 4th step with full-wave stores), lane i < 32
// fetches element (i, pivot) of that image -- consecutive lanes, consecutive addresses -- applies
// the <= 3 rank-1 corrections of the steps since the last refresh from its own registers, and
// publishes the result as a 32-double vector that all lanes then read by broadcast.  No barrier is
// needed inside the factorisation (one wave, in-order LDS queue).
//
// Columns are never physically swapped: each A lane tracks its current position (Eigen's
// m_colsTranspositions bookkeeping), so the "first maximum" tie rule and the final permutation are
// those of Eigen's ColPivHouseholderQR.  Row k of R is final after step k and is parked in the LDS
// slot of the pivot column (dead from then on); the epilogue gathers the packed upper triangle
// through the permutation.  All global accesses are full-line coalesced 16-B accesses via LDS.
#include "qrk_device.h"

#include <float.h>

namespace qrk {

constexpr int WR = 32;               // row registers per lane
constexpr int LDP = WR + 2;          // LDS column stride in doubles: 272 B, conflict-free b64/b128 access
constexpr int RB = 4;                // the LDS image of A is refreshed every RB steps
#ifndef QRK_WAVES_PER_SIMD
#define QRK_WAVES_PER_SIMD 3
#endif
// LDS carve-up (doubles)
constexpr int L_IMG = 0;             // [32][LDP] column-major image of A / staging for Q; R rows parked here
constexpr int L_XBUF = WR * LDP;     // [32] current pivot column
constexpr int L_WBUF = L_XBUF + WR;  // [RB][32] update coefficients of the last RB steps, per A column
constexpr int L_POS = L_WBUF + RB * WR;   // [32] int: lane_of_pos
constexpr int L_TOTAL = L_POS + WR / 2;   // 1264 doubles = 10112 B -> 16 waves per CU

constexpr double SQRT_EPS = 1.4901161193847656e-08;   // sqrt(DBL_EPSILON), Eigen's norm_downdate_threshold

#define QRK_0_31(M)                                                                              \
    M(0) M(1) M(2) M(3) M(4) M(5) M(6) M(7) M(8) M(9) M(10) M(11) M(12) M(13) M(14) M(15) M(16)  \
    M(17) M(18) M(19) M(20) M(21) M(22) M(23) M(24) M(25) M(26) M(27) M(28) M(29) M(30) M(31)

// sqrt(x) for a positive normal x: v_rsq_f64 seed (2^-24), one Goldschmidt iteration and one residual
// correction: <= 1 ulp (tools/ubench3.hip), no FP64 division.
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

// Inverse of e = p(p+1)/2 + i (0 <= i <= p): position in the packed upper triangle by columns.
__device__ __forceinline__ void tri_unpack(int e, int& p, int& i)
{
    int q = (int)((sqrtf(8.0f * (float)e + 1.0f) - 1.0f) * 0.5f);
    if ((q + 1) * (q + 2) / 2 <= e) ++q;
    if (q * (q + 1) / 2 > e) --q;
    p = q;
    i = e - q * (q + 1) / 2;
}

// Per-wave state that lives across the steps.
struct WaveState {
    bool isA;        // lane owns a column of A (lanes 0..31) rather than of Q^T
    bool live;       // A column not yet chosen as a pivot
    int lane;
    int pos;         // current position of this column (Eigen swaps columns physically)
    int rows;        // tile rows (rows >= this are zero padding)
    double nu2;      // m_colNormsUpdated^2
    double thr_nd2;  // sqrt(eps) * m_colNormsDirect^2
    double h[RB];    // lane i < 32: entry i of the pivot columns of the last RB steps
};

// One step of ColPivHouseholderQR::computeInPlace (Eigen/src/QR/ColPivHouseholderQR.h) on the
// wave-resident tile: pivot search, reflector, trailing update (also of Q^T), norm downdate.
//
// Column norms are tracked SQUARED: Eigen's  temp = (1+t)(1-t), t = |a_kj|/normUpd;
// normUpd *= sqrt(temp)  is  nu2 <- max(nu2 - a_kj^2, 0), and its recompute test
// temp (normUpd/normDir)^2 <= sqrt(eps)  is  nu2_new <= sqrt(eps) normDir^2: the same quantities
// without two FP64 divisions and a square root per step; the first maximum is the same column
// because squaring is monotone.
template <int K, bool FULL32>
__device__ __forceinline__ void factor_step_k(double (&a)[WR], double* lds, WaveState& st, int pivoting,
                                              double* __restrict__ hcoeffs_tile)
{
    const int lane = st.lane;
    // ---- pivot: first maximum of the updated norms over positions K..c-1.  Non-negative doubles
    // order like their bit patterns, so the wave max is an integer max on (hi, lo).
    int lb;
    if (pivoting) {
        const int khi = st.live ? __double2hiint(st.nu2) : (int)0x80000000;
        const int mh = half32_max_i32_fast(khi);
        unsigned long long tie = __ballot(st.live && khi == mh);
        if (__popcll(tie) > 1) {
            // several columns share the high word: compare the low words, then Eigen's first-maximum
            // rule = smallest CURRENT position among exact ties
            const unsigned klo = (unsigned)__double2loint(st.nu2);
            const bool cand = (tie >> lane) & 1ull;
            const unsigned ml = half32_max_u32(cand ? klo : 0u);
            tie = __ballot(cand && klo == ml);
            if (__popcll(tie) > 1) {
                const int pc = ((tie >> lane) & 1ull) ? st.pos : 64;
                const int pmin = half32_min_i32(pc);
                tie = __ballot(pc == pmin);
            }
        }
        lb = __builtin_amdgcn_readfirstlane(__ffsll((long long)tie) - 1);
        const int bpos = __builtin_amdgcn_readlane(st.pos, lb);
        if (lane == lb) st.pos = K;
        else if (st.live && st.pos == K) st.pos = bpos;
    } else {
        lb = K;
    }
    const bool ispiv = lane == lb;
    if (ispiv) st.live = false;

    // ---- pivot column: element (lane, lb) of the LDS image (exact through step K0-1) plus the rank-1
    // corrections of steps K0..K-1, then published for broadcast reads.
    constexpr int K0 = (K / RB) * RB;
    {
        double xi = lds[L_IMG + lb * LDP + (lane & 31)];
#pragma unroll
        for (int m = K0; m < K; ++m) xi = fma(lds[L_WBUF + (m % RB) * WR + lb], st.h[m % RB], xi);
        if (!FULL32) xi = (lane & 31) < st.rows ? xi : 0.0;
        st.h[K % RB] = xi;
        if (lane < 32) lds[L_XBUF + lane] = xi;
    }

    // ---- d = x_tail^T a_tail per lane (on the pivot lane itself: |x_tail|^2)
    const double ak = a[K];
    const double xk = lds[L_XBUF + K];
    double d = 0.0;
#pragma unroll
    for (int i = K + 1; i < WR; ++i) {
        d = fma(lds[L_XBUF + i], a[i], d);
        if ((i & 7) == 7) __builtin_amdgcn_sched_barrier(0);   // bound the x registers in flight
    }

    // ---- makeHouseholder + applyHouseholderOnTheLeft (Eigen/src/Householder/Householder.h) in the
    // un-normalised form: with beta = -sign(x0) sqrt(x0^2 + |tail|^2) and w = beta - x0,
    //   tau = w/beta, essential = tail/(x0 - beta) = -tail/w, and for a column c with tail dot d
    //   gamma = (d - w c_k) / (beta w):   c_k <- c_k + w gamma  (= c_k - tau tmp),
    //                                     c_i <- c_i - gamma x_i (= c_i - tau ess_i tmp),
    // which needs one square root and one reciprocal (of beta*w > 0) per step and no division.
    const double tailSq = readlane_f64(d, lb);
    const bool degen = tailSq <= DBL_MIN;           // Eigen: tau = 0, beta = x0, H = I
    const double nrm = sqrt_pos(fma(xk, xk, tailSq));
    double beta = xk >= 0.0 ? -nrm : nrm;
    double w = beta - xk;
    double g = recip(beta * w);
    if (degen) { g = 0.0; beta = xk; w = 0.0; }   // (nrm may be NaN here: rsq(0) = inf)
    if (hcoeffs_tile && ispiv) hcoeffs_tile[K] = (w * w) * g;     // tau = w/beta = w^2/(beta w)

    // Chosen columns keep their R entries untouched (zero coefficient).
    const bool upd = !st.isA || st.live;
    const double gam = upd ? fma(-w, ak, d) * g : 0.0;
    double an = fma(w, gam, ak);
    if (ispiv) an = beta;
    const double ncoef = -gam;
    a[K] = an;   // row K of Q^T (for A columns the value goes to the parked R row below)
    if (lane < 32) lds[L_WBUF + (K % RB) * WR + lane] = ncoef;
#pragma unroll
    for (int i = K + 1; i < WR; ++i) {
        a[i] = fma(ncoef, lds[L_XBUF + i], a[i]);
        if ((i & 7) == 7) __builtin_amdgcn_sched_barrier(0);
    }

    // Row K of R is final: park it in the LDS slot of the pivot column (never read as a column again).
    if (st.isA && (st.live || ispiv)) lds[L_IMG + lb * LDP + lane] = an;

    // ---- refresh the LDS image of the live columns after every RB-th step
    if (K % RB == RB - 1 && K + 1 < WR) {
        if (st.live) {
#pragma unroll
            for (int i = K + 1; i < WR; ++i) lds[L_IMG + lane * LDP + i] = a[i];
        }
    }

    // ---- LAWN-176 norm downdate for the remaining columns (squared form, see above)
    if (pivoting) {
        double nn = fma(-an, an, st.nu2);
        nn = nn > 0.0 ? nn : 0.0;
        const bool need = st.live && nn <= st.thr_nd2;
        st.nu2 = nn;
        if (__any(need)) {
            double s = 0.0;
#pragma unroll
            for (int i = K + 1; i < WR; ++i) s = fma(a[i], a[i], s);
            if (need) { st.nu2 = s; st.thr_nd2 = s * SQRT_EPS; }
        }
    }
}

// FULL32: every tile is 32x32 and all arrays are 16-byte aligned (uniform batch).
template <bool FULL32>
__global__ void __launch_bounds__(64, QRK_WAVES_PER_SIMD)
bdqr_wave_kernel(WaveBatch nb, const double* __restrict__ tiles, double* __restrict__ q_vals,
                 double* __restrict__ r_vals, int32_t* __restrict__ perm,
                 double* __restrict__ hcoeffs)
{
    __shared__ __attribute__((aligned(16))) double lds[L_TOTAL];
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
                *reinterpret_cast<double2*>(&lds[L_IMG + (e2 >> 4) * LDP + ((e2 & 15) << 1)]) = v;
            }
        } else {
            const double* src = tiles + toff;
            const int n_in = r * c;
            for (int e = lane; e < n_in; e += 64) {
                const int cc = e / r;
                lds[L_IMG + cc * LDP + (e - cc * r)] = src[e];
            }
        }
        __syncthreads();

        double a[WR];
        if (isA) {
            if (FULL32) {
#pragma unroll
                for (int i = 0; i < WR; ++i) a[i] = lds[L_IMG + col * LDP + i];
            } else {
#pragma unroll
                for (int i = 0; i < WR; ++i) a[i] = (col < c && i < r) ? lds[L_IMG + col * LDP + i] : 0.0;
            }
        } else {
#pragma unroll
            for (int i = 0; i < WR; ++i) a[i] = (i == col && col < r) ? 1.0 : 0.0;
        }
        // (no barrier: the image stays valid, it is the source of the pivot columns)

        WaveState st;
        st.isA = isA; st.lane = lane; st.pos = col; st.rows = r;
        st.live = isA && col < c;
#pragma unroll
        for (int m = 0; m < RB; ++m) st.h[m] = 0.0;
        {
            // squared column norms (ColPivHouseholderQR: m_colNormsUpdated^2, sqrt(eps) m_colNormsDirect^2)
            double s = 0.0;
#pragma unroll
            for (int i = 0; i < WR; ++i) s = fma(a[i], a[i], s);
            st.nu2 = s;
            st.thr_nd2 = s * SQRT_EPS;
        }
        double* hc_tile = hcoeffs ? hcoeffs + cbase : nullptr;

        // The k loop is expanded by the preprocessor: every row-register index is a compile-time
        // constant.  (A rolled loop dispatching through a uniform switch makes hipcc's CFG
        // structurizer copy the whole register tile at every merge point.)
#define QRK_STEP(K) if (FULL32 || K < c) factor_step_k<K, FULL32>(a, lds, st, nb.pivoting, hc_tile);
        QRK_0_31(QRK_STEP)
#undef QRK_STEP

        // ---- R: row i of R sits in the LDS slot of the column chosen at step i, indexed by ORIGINAL
        // column.  The packed upper triangle by columns is exactly the CSC value order of m_R
        // (BlockDiagonalSparseQR.h:475-479): element e -> (column p, row i), gathered through lane_of_pos.
        int* lane_of_pos = reinterpret_cast<int*>(&lds[L_POS]);
        if (isA && col < c) {
            lane_of_pos[st.pos] = col;
            perm[cbase + st.pos] = cbase + col;     // m_outputPerm_c.indices()(base_col+j) (:519-521)
        }
        __syncthreads();
        if (FULL32) {
            double2* dst = reinterpret_cast<double2*>(r_vals + roff);
#pragma unroll
            for (int q = 0; q < 5; ++q) {
                const int e2 = lane + 64 * q;
                if (e2 < 264) {
                    int p0, i0, p1, i1;
                    tri_unpack(2 * e2, p0, i0);
                    tri_unpack(2 * e2 + 1, p1, i1);
                    dst[e2] = make_double2(lds[L_IMG + lane_of_pos[i0] * LDP + lane_of_pos[p0]],
                                           lds[L_IMG + lane_of_pos[i1] * LDP + lane_of_pos[p1]]);
                }
            }
        } else {
            const int n_r = c * (c + 1) / 2;
            for (int e = lane; e < n_r; e += 64) {
                int p0, i0;
                tri_unpack(e, p0, i0);
                r_vals[roff + e] = lds[L_IMG + lane_of_pos[i0] * LDP + lane_of_pos[p0]];
            }
        }
        __syncthreads();

        // ---- Q: lane 32+j holds row j of Q_i; row-major rows are the CSR value order of m_Q
        // in both FullQ ([U|N] split, :455-471) and BlockDiagonalQ (:480-492) layouts.
        if (!isA && col < r) {
            if (FULL32) {
#pragma unroll
                for (int i = 0; i < WR; i += 2)
                    *reinterpret_cast<double2*>(&lds[L_IMG + col * LDP + i]) = make_double2(a[i], a[i + 1]);
            } else {
#pragma unroll
                for (int i = 0; i < WR; ++i)
                    if (i < r) lds[L_IMG + col * LDP + i] = a[i];
            }
        }
        __syncthreads();
        if (FULL32) {
            double2* dst = reinterpret_cast<double2*>(q_vals + qoff);
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const int e2 = lane + 64 * q;
                dst[e2] = *reinterpret_cast<const double2*>(&lds[L_IMG + (e2 >> 4) * LDP + ((e2 & 15) << 1)]);
            }
        } else {
            const int n_q = r * r;
            for (int e = lane; e < n_q; e += 64) {
                const int j = e / r;
                q_vals[qoff + e] = lds[L_IMG + j * LDP + (e - j * r)];
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
