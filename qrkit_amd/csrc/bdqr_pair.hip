// bdqr_pair.hip -- one wavefront factorises TWO tiles (rows, cols <= 32) of a block-diagonal
// matrix at once: A_i P_i = Q_i R_i with explicit Q_i, for gfx950.
//
// Replaces the body of the hot loop of QRKit::BlockDiagonalSparseQR::factorize
// (src/QRKit/BlockDiagonalSparseQR.h:432-526): blockSolver.compute(block) (:437-438,
// Eigen ColPivHouseholderQR / HouseholderQR), Qi = blockSolver.matrixQ() (:446), the
// Q / R value assembly (:455-500) and the column-permutation splice (:519-521).
//
// Why two tiles per wave: the per-step work that is uniform per tile (pivot search, the reflector
// scalars with their sqrt and reciprocal, the norm downdate) is executed by every lane of a SIMD
// instruction anyway; the single-tile kernel (bdqr_wave.hip) spends ~80 % of its VALU cycles there.
// Giving each half of the wave its own tile makes every one of those instructions serve two tiles.
//
// Mapping (wave64): lane = 32*h + j.  Half h owns tile 2*pair + h; lane j of the half owns column j
// of A (32 row registers a[], zero padded) AND column j of Q^T (= row j of Q, registers q[]),
// starting from the identity.  Reflector k is the same operation on both, c <- c - gamma (x^T c ..),
// so A -> R and I -> Q^T advance together as FP64 FMA chains over the row registers.
//
// The pivot column has to reach all 32 lanes of its half.  v_readlane costs ~7 SIMD cycles per dword
// on gfx950 and ds_bpermute ~8 ns, but an LDS read of one address per half is almost free
// (tools/ubench2.hip).  So each half keeps a column-major image of its A in LDS (the staging image
// of the load, refreshed by the owning lanes every RB-th step with full-wave stores); lane j fetches
// element (j, pivot) of the image -- consecutive lanes, consecutive addresses -- applies the < RB
// rank-1 corrections since the last refresh from its own registers and publishes the result as a
// 32-double vector that the half then reads by broadcast.  No barrier inside the factorisation
// (one wave, in-order LDS queue).
//
// Columns are never physically swapped: each lane tracks the current position of its A column
// (Eigen's m_colsTranspositions bookkeeping), so the "first maximum" tie rule and the final
// permutation are those of Eigen's ColPivHouseholderQR.  Row k of R is final after step k and is
// parked in the LDS slot of the pivot column (dead from then on); the epilogue gathers the packed
// upper triangle through the permutation.  All global accesses are coalesced 16-B accesses via LDS.
#include "qrk_device.h"

#include <float.h>

namespace qrk {

namespace pair {

constexpr int WR = 32;               // row registers per column
constexpr int LDP = WR + 2;          // LDS column stride in doubles: 272 B, conflict-free b64/b128 access
constexpr int RB = 4;                // the LDS image of A is refreshed every RB steps
// LDS carve-up per HALF (doubles)
constexpr int L_IMG = 0;             // [32][LDP] column-major image of A / staging for Q; R rows parked here
constexpr int L_XBUF = WR * LDP;     // [32] current pivot column
constexpr int L_WBUF = L_XBUF + WR;  // [RB][32] update coefficients of the last RB steps, per A column
constexpr int L_POS = L_WBUF + RB * WR;   // [32] int: lane_of_pos
constexpr int L_HALF = L_POS + WR / 2;    // 1264 doubles = 10112 B per half, 20224 B per wave -> 8 waves per CU

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

// 1/x by v_rcp_f64 + two Newton steps (<= 1 ulp), no v_div_* sequence.
__device__ __forceinline__ double recip(double x)
{
    double y = __builtin_amdgcn_rcp(x);
    double e = fma(-x, y, 1.0);
    y = fma(y, e, y);
    e = fma(-x, y, 1.0);
    y = fma(y, e, y);
    return y;
}

// Two FMAs that share one operand, issued back to back.  Written as one asm statement so that
// hipcc cannot separate them (it otherwise defers one accumulation chain and parks the shared
// pivot-column values in scratch).  d0 += x*c0; d1 += x*c1.
__device__ __forceinline__ void fmac2_shared_a(double& d0, double& d1, double x, double c0, double c1)
{
    asm("v_fmac_f64_e32 %0, %2, %3\n\tv_fmac_f64_e32 %1, %2, %4" : "+v"(d0), "+v"(d1) : "v"(x), "v"(c0), "v"(c1));
}
// c0 += n0*x; c1 += n1*x.
__device__ __forceinline__ void fmac2_shared_b(double& c0, double& c1, double n0, double n1, double x)
{
    asm("v_fmac_f64_e32 %0, %2, %4\n\tv_fmac_f64_e32 %1, %3, %4" : "+v"(c0), "+v"(c1) : "v"(n0), "v"(n1), "v"(x));
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

// Per-lane state that lives across the steps.
struct LaneState {
    int lane, j, half;
    bool live;       // this lane's A column is not yet chosen as a pivot
    int pos;         // current position of this column (Eigen swaps columns physically)
    int rows;        // tile rows of this half (rows >= this are zero padding)
    double nu2;      // m_colNormsUpdated^2
    double thr_nd2;  // sqrt(eps) * m_colNormsDirect^2
    double h[RB];    // entry j of the pivot columns of the last RB steps
};

// One step of ColPivHouseholderQR::computeInPlace (Eigen/src/QR/ColPivHouseholderQR.h) on both
// wave-resident tiles: pivot search, reflector, trailing update of A and of Q^T, norm downdate.
//
// Column norms are tracked SQUARED: Eigen's  temp = (1+t)(1-t), t = |a_kj|/normUpd;
// normUpd *= sqrt(temp)  is  nu2 <- max(nu2 - a_kj^2, 0), and its recompute test
// temp (normUpd/normDir)^2 <= sqrt(eps)  is  nu2_new <= sqrt(eps) normDir^2: the same quantities
// without two FP64 divisions and a square root per step; the first maximum is the same column
// because squaring is monotone.
template <int K, bool FULL32>
__device__ __forceinline__ void pair_step(double (&a)[WR], double (&q)[WR], double* hl /* this half's LDS */,
                                          LaneState& st, int pivoting, double* __restrict__ hcoeffs_tile)
{
    const int lane = st.lane, j = st.j;
    // ---- pivot: first maximum of the updated norms over positions K..c-1, per half.  Non-negative
    // doubles order like their bit patterns, so the max is an integer max on (hi, lo).
    int lb;          // pivot lane of this lane's half
    int lbA, lbB;    // pivot lanes of half 0 / half 1 (wave-uniform)
    bool act;        // this half still has a column to eliminate at step K
    if (pivoting) {
        const int khi = st.live ? __double2hiint(st.nu2) : (int)0x80000000;
        const int mh = half32_max_i32_fast(khi);
        unsigned long long tie = __ballot(st.live && khi == mh);
        unsigned tlo = (unsigned)tie, thi = (unsigned)(tie >> 32);
        if (__popc(tlo) > 1 || __popc(thi) > 1) {
            // several columns share the high word: compare the low words, then Eigen's first-maximum
            // rule = smallest CURRENT position among exact ties
            const unsigned klo = (unsigned)__double2loint(st.nu2);
            const bool cand = (tie >> lane) & 1ull;
            const unsigned ml = half32_max_u32(cand ? klo : 0u);
            tie = __ballot(cand && klo == ml);
            const int pc = ((tie >> lane) & 1ull) ? st.pos : 64;
            const int pmin = half32_min_i32(pc);
            tie = __ballot(pc == pmin && pc != 64);
            tlo = (unsigned)tie; thi = (unsigned)(tie >> 32);
        }
        lbA = tlo ? __ffs((int)tlo) - 1 : 0;
        lbB = thi ? __ffs((int)thi) + 31 : 32;
        lb = st.half ? lbB : lbA;
        act = (st.half ? thi : tlo) != 0u;
        const int bposA = __builtin_amdgcn_readlane(st.pos, lbA);
        const int bposB = __builtin_amdgcn_readlane(st.pos, lbB);
        const int bpos = st.half ? bposB : bposA;
        if (lane == lb) st.pos = act ? K : st.pos;
        else if (st.live && st.pos == K) st.pos = bpos;
    } else {
        const unsigned long long lv = __ballot(st.live);
        act = ((st.half ? (unsigned)(lv >> 32) : (unsigned)lv)) != 0u;   // HouseholderQR: column K
        lbA = K; lbB = 32 + K;
        lb = 32 * st.half + K;
    }
    const bool ispiv = act && lane == lb;
    if (ispiv) st.live = false;
    const int lbl = lb & 31;

    // ---- pivot column: element (j, lb) of the LDS image (exact through step K0-1) plus the rank-1
    // corrections of steps K0..K-1, then published for broadcast reads.
    constexpr int K0 = (K / RB) * RB;
    {
        double xi = hl[L_IMG + lbl * LDP + j];
#pragma unroll
        for (int m = K0; m < K; ++m) xi = fma(hl[L_WBUF + (m % RB) * WR + lbl], st.h[m % RB], xi);
        xi = (act && (FULL32 || j < st.rows)) ? xi : 0.0;
        st.h[K % RB] = xi;
        hl[L_XBUF + j] = xi;
    }

    // ---- d = x_tail^T c_tail for the A column and the Q^T column (pivot lane: dA = |x_tail|^2)
    const double ak = a[K], qk = q[K];
    const double xk = hl[L_XBUF + K];
    double dA0 = 0.0, dQ0 = 0.0, dA1 = 0.0, dQ1 = 0.0;   // two accumulators per chain: half the dependent latency
#pragma unroll
    for (int i = K + 1; i < WR; ++i) {
        const double xv = hl[L_XBUF + i];   // broadcast read
        if ((i - K) & 1) fmac2_shared_a(dA0, dQ0, xv, a[i], q[i]);
        else fmac2_shared_a(dA1, dQ1, xv, a[i], q[i]);
    }
    const double dA = dA0 + dA1, dQ = dQ0 + dQ1;

    // ---- makeHouseholder + applyHouseholderOnTheLeft (Eigen/src/Householder/Householder.h) in the
    // un-normalised form: with beta = -sign(x0) sqrt(x0^2 + |tail|^2) and w = beta - x0,
    //   tau = w/beta, essential = tail/(x0 - beta) = -tail/w, and for a column c with tail dot d
    //   gamma = (d - w c_k) / (beta w):   c_k <- c_k + w gamma  (= c_k - tau tmp),
    //                                     c_i <- c_i - gamma x_i (= c_i - tau ess_i tmp),
    // which needs one square root and one reciprocal (of beta*w > 0) per step and no division.
    const double tsA = readlane_f64(dA, lbA);
    const double tsB = readlane_f64(dA, lbB);
    const double tailSq = st.half ? tsB : tsA;
    const bool degen = !act || tailSq <= DBL_MIN;   // Eigen: tau = 0, beta = x0, H = I
    const double nrm = sqrt_pos(fma(xk, xk, tailSq));
    double beta = xk >= 0.0 ? -nrm : nrm;
    double w = beta - xk;
    double g = recip(beta * w);
    if (degen) { g = 0.0; beta = xk; w = 0.0; }   // (nrm may be NaN here: rsq(0) = inf)
    if (hcoeffs_tile && ispiv) hcoeffs_tile[K] = (w * w) * g;     // tau = w/beta = w^2/(beta w)

    // A column: chosen columns keep their R entries untouched (zero coefficient).
    const double gamA = st.live ? fma(-w, ak, dA) * g : 0.0;
    double an = fma(w, gamA, ak);
    if (ispiv) an = beta;
    const double gamQ = fma(-w, qk, dQ) * g;
    q[K] = fma(w, gamQ, qk);
    const double ncA = -gamA, ncQ = -gamQ;
    hl[L_WBUF + (K % RB) * WR + j] = ncA;
    // Re-read the pivot column from LDS for the update (a broadcast read is nearly free); the opaque
    // offset keeps hipcc from carrying the 31 values of the dot pass in registers / scratch instead.
    int xo = L_XBUF;
    // (no launder)
    const double* xb = hl + xo;
#pragma unroll
    for (int i = K + 1; i < WR; ++i) {
        const double xv = xb[i];
        fmac2_shared_b(a[i], q[i], ncA, ncQ, xv);
    }

    // Row K of R is final: park it in the LDS slot of the pivot column (never read as a column again).
    if (act && (st.live || ispiv)) hl[L_IMG + lbl * LDP + j] = an;

    // ---- refresh the LDS image of the live columns after every RB-th step
    if (K % RB == RB - 1 && K + 1 < WR) {
        if (st.live) {
#pragma unroll
            for (int i = K + 1; i < WR; ++i) hl[L_IMG + j * LDP + i] = a[i];
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

}  // namespace pair

// FULL32: every tile is 32x32 and all arrays are 16-byte aligned (uniform batch).
template <bool FULL32>
__global__ void __launch_bounds__(64, 2)
bdqr_pair_kernel(WaveBatch nb, const double* __restrict__ tiles, double* __restrict__ q_vals,
                 double* __restrict__ r_vals, int32_t* __restrict__ perm,
                 double* __restrict__ hcoeffs)
{
    using namespace pair;
    __shared__ __attribute__((aligned(16))) double lds[2 * L_HALF];
    // One pair per workgroup (no grid-stride loop: a loop makes hipcc hoist per-lane addresses out of
    // it and keep them in scratch across the whole factorisation).
    {
        const int64_t pi = blockIdx.x;
        const int lane = threadIdx.x;
        const int half = lane >> 5, j = lane & 31;
        double* hl = lds + half * L_HALF;
        const int64_t t = 2 * pi + half;
        const bool valid = t < nb.num_tiles;
        int r, c, cbase;
        int64_t toff, qoff, roff;
        if (FULL32) {
            r = 32; c = 32;
            toff = t * 1024; qoff = t * 1024; roff = t * 528; cbase = (int)(t * 32);
        } else if (nb.tile_ids) {
            const int gidx = nb.tile_ids[valid ? t : nb.num_tiles - 1];
            r = nb.t_rows[gidx]; c = nb.t_cols[gidx];
            toff = nb.t_off[gidx]; qoff = nb.q_off[gidx]; roff = nb.r_off[gidx]; cbase = nb.c_off[gidx];
        } else {
            r = nb.rows; c = nb.cols;
            toff = t * r * c; qoff = t * (int64_t)r * r; roff = t * (int64_t)(c * (c + 1) / 2);
            cbase = (int)(t * c);
        }
        if (!valid) { r = 0; c = 0; }

        // ---- stage the tiles: coalesced global read (each half its own tile), padded LDS columns
        if (FULL32) {
            if (valid) {
                const double2* src = reinterpret_cast<const double2*>(tiles + toff);
                for (int q0 = 0; q0 < 16; q0 += 8) {   // two batches of eight 16-B loads per lane
#pragma unroll
                    for (int qq = 0; qq < 8; ++qq) {
                        const int e2 = j + 32 * (q0 + qq);    // double2 index, 16 per column
                        const double2 v = src[e2];
                        *reinterpret_cast<double2*>(&hl[L_IMG + (e2 >> 4) * LDP + ((e2 & 15) << 1)]) = v;
                    }
                }
            }
        } else {
            const double* src = tiles + toff;
            const int n_in = r * c;
            for (int e = j; e < n_in; e += 32) {
                const int cc = e / r;
                hl[L_IMG + cc * LDP + (e - cc * r)] = src[e];
            }
        }
        __syncthreads();

        double a[WR], q[WR];
#pragma unroll
        for (int i = 0; i < WR; ++i) {
            if (FULL32) a[i] = valid ? hl[L_IMG + j * LDP + i] : 0.0;
            else a[i] = (j < c && i < r) ? hl[L_IMG + j * LDP + i] : 0.0;
            q[i] = (i == j && j < r) ? 1.0 : 0.0;
        }
        // (no barrier: the image stays valid, it is the source of the pivot columns)

        LaneState st;
        st.lane = lane; st.j = j; st.half = half; st.pos = j; st.rows = r;
        st.live = j < c;
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
        double* hc_tile = (hcoeffs && valid) ? hcoeffs + cbase : nullptr;
        // number of steps = the larger column count of the two tiles
        const int c0 = __builtin_amdgcn_readlane(c, 0), c1 = __builtin_amdgcn_readlane(c, 32);
        const int cmax = c0 > c1 ? c0 : c1;

        // The k loop is expanded by the preprocessor: every row-register index is a compile-time
        // constant.  (A rolled loop dispatching through a uniform switch makes hipcc's CFG
        // structurizer copy the whole register tile at every merge point.)
#define QRK_STEP(K) if (FULL32 || K < cmax) pair_step<K, FULL32>(a, q, hl, st, nb.pivoting, hc_tile);
        QRK_0_31(QRK_STEP)
#undef QRK_STEP

        // ---- R: row i of R sits in the LDS slot of the column chosen at step i, indexed by ORIGINAL
        // column.  The packed upper triangle by columns is exactly the CSC value order of m_R
        // (BlockDiagonalSparseQR.h:475-479): element e -> (column p, row i), gathered through lane_of_pos.
        int* lane_of_pos = reinterpret_cast<int*>(&hl[L_POS]);
        if (j < c) {
            lane_of_pos[st.pos] = j;
            perm[cbase + st.pos] = cbase + j;     // m_outputPerm_c.indices()(base_col+j) (:519-521)
        }
        __syncthreads();
        if (FULL32) {
            if (valid) {
                double2* dst = reinterpret_cast<double2*>(r_vals + roff);
#pragma unroll
                for (int qq = 0; qq < 9; ++qq) {
                    const int e2 = j + 32 * qq;
                    if (e2 < 264) {
                        int p0, i0, p1, i1;
                        tri_unpack(2 * e2, p0, i0);
                        tri_unpack(2 * e2 + 1, p1, i1);
                        dst[e2] = make_double2(hl[L_IMG + lane_of_pos[i0] * LDP + lane_of_pos[p0]],
                                               hl[L_IMG + lane_of_pos[i1] * LDP + lane_of_pos[p1]]);
                    }
                }
            }
        } else {
            const int n_r = c * (c + 1) / 2;
            for (int e = j; e < n_r; e += 32) {
                int p0, i0;
                tri_unpack(e, p0, i0);
                r_vals[roff + e] = hl[L_IMG + lane_of_pos[i0] * LDP + lane_of_pos[p0]];
            }
        }
        __syncthreads();

        // ---- Q: lane j holds row j of Q_i; row-major rows are the CSR value order of m_Q in both
        // FullQ ([U|N] split, :455-471) and BlockDiagonalQ (:480-492) layouts.
        if (j < r) {
            if (FULL32) {
#pragma unroll
                for (int i = 0; i < WR; i += 2)
                    *reinterpret_cast<double2*>(&hl[L_IMG + j * LDP + i]) = make_double2(q[i], q[i + 1]);
            } else {
#pragma unroll
                for (int i = 0; i < WR; ++i)
                    if (i < r) hl[L_IMG + j * LDP + i] = q[i];
            }
        }
        __syncthreads();
        if (FULL32) {
            if (valid) {
                double2* dst = reinterpret_cast<double2*>(q_vals + qoff);
                for (int q0 = 0; q0 < 16; q0 += 8) {
#pragma unroll
                    for (int qq = 0; qq < 8; ++qq) {
                        const int e2 = j + 32 * (q0 + qq);
                        dst[e2] = *reinterpret_cast<const double2*>(&hl[L_IMG + (e2 >> 4) * LDP + ((e2 & 15) << 1)]);
                    }
                }
            }
        } else {
            const int n_q = r * r;
            for (int e = j; e < n_q; e += 32) {
                const int jj = e / r;
                q_vals[qoff + e] = hl[L_IMG + jj * LDP + (e - jj * r)];
            }
        }
        __syncthreads();
    }
}

void launch_bdqr_pair(const WaveBatch& nb, bool full32, const double* tiles, double* q_vals,
                      double* r_vals, int32_t* perm, double* hcoeffs, int max_blocks,
                      hipStream_t stream)
{
    if (nb.num_tiles <= 0) return;
    (void)max_blocks;
    const int64_t npairs = (nb.num_tiles + 1) / 2;
    const dim3 grid((unsigned)npairs), block(64);
    if (full32)
        hipLaunchKernelGGL(bdqr_pair_kernel<true>, grid, block, 0, stream, nb, tiles, q_vals, r_vals,
                           perm, hcoeffs);
    else
        hipLaunchKernelGGL(bdqr_pair_kernel<false>, grid, block, 0, stream, nb, tiles, q_vals, r_vals,
                           perm, hcoeffs);
}

}  // namespace qrk
