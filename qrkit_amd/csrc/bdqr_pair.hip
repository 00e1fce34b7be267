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
// Columns are never physically swapped.  The column chosen at step k ends at position k, which is
// all the final permutation needs; the current positions that Eigen's "first maximum" tie rule
// looks at (m_colsTranspositions bookkeeping) are rebuilt only when an exact tie occurs.  Entry k of
// a lane's column is final after step k (row k of R) and simply stays in its row register; the
// column chosen at step p therefore ends as column p of R in rows 0..p of its lane, and the epilogue
// packs the upper triangle with one unpredicated LDS store per row (see pack_r_column).
//
// Two kernels share the step: bdqr_pair_kernel (one pair per workgroup, any tile shape <= 32x32,
// coalesced I/O staged through LDS) and bdqr_pair32_kernel (uniform 32x32 batches: persistent
// workgroups that prefetch their next tile into dead registers and store Q rows straight from
// registers).  Measured history and the per-phase cycle counts are in HISTORY.md (the current numbers in DESIGN.md).
#include "qrk_device.h"
#include "bdqr_exact_tile.h"

#include <float.h>
#include <cstdlib>

// Diagnostic only (tools/ablate.py): -DQRK_ABL=<mask> removes one phase of the step to see what it
// costs.  Results are wrong with any bit set; the product build never defines it.
#ifndef QRK_ABL
#define QRK_ABL 0
#endif
#ifndef QRK_DECISIONS
#define QRK_DECISIONS 31       // diagnostic only (tools/ab.py): bit mask of the decision checks compiled in, to measure what each
                               // costs: 1 pivot margin, 2 |x0| test, 4 recompute band, 8 pivot at the noise level, 16 degenerate tail
#endif
#ifndef QRK_DOT4
#define QRK_DOT4 0
#endif
#ifndef QRK_RB
#define QRK_RB 4               // steps between refreshes of the LDS image (measured on one box: 4 -> 87.1 us, 3 -> 87.4, 2 -> 91.0; 8 needs 22 KB of LDS per wave)
#endif
#ifndef QRK_DPPX
#define QRK_DPPX 1             // 1: the pivot column reaches the lanes as the DPP operand of the FMAs (row_newbcast), no LDS publish / broadcast reads;
                               // 0: the round-1 form (XBUF in LDS, ds_read_b128 broadcasts), kept for A/B measurements
#endif
// Diagnostic only (occupancy experiments; results are wrong): -DQRK_DIAG_LDS16 folds the LDS image onto 16 columns (11.8 KB of LDS
// per wave instead of 20 KB), -DQRK_MINW=3 compiles the persistent kernel for three waves per SIMD.
#ifdef QRK_DIAG_LDS16
#define QRK_IC(x) ((x) & 15)
#define QRK_IMG_COLS 16
#else
#define QRK_IC(x) (x)
#define QRK_IMG_COLS 32
#endif
#ifndef QRK_MINW
#define QRK_MINW 2
#endif
#ifndef QRK_DOT_SPLIT
#define QRK_DOT_SPLIT 0
#endif
#ifndef QRK_QSTORE_EVERY
#define QRK_QSTORE_EVERY 8     // 4, 8 or 16
#endif
#ifndef QRK_TAILSQ_READLANE
#define QRK_TAILSQ_READLANE 0  // 1: |x_tail|^2 leaves the pivot lane through v_readlane (SGPRs) instead of ds_bpermute (round-4 A/B)
#endif

// Diagnostic only (tools/stamp_run.py): -DQRK_STAMP records s_memtime of lane 0 at phase boundaries
// of every pair into the hcoeffs buffer (12 x int64 per pair).
#ifdef QRK_STAMP
#define QRK_STAMP_AT(slot)                                                                               \
    do {                                                                                                 \
        if (threadIdx.x == 0 && hcoeffs)                                                                 \
            reinterpret_cast<unsigned long long*>(hcoeffs)[(size_t)pi * 12 + (slot)] = __builtin_amdgcn_s_memtime(); \
    } while (0)
#ifndef QRK_STAMP_K
#define QRK_STAMP_K 20
#endif
// ... and inside step QRK_STAMP_K (kept in SGPRs, written out after the pair)
#define QRK_STAMP_IN(n) do { if (K == QRK_STAMP_K) st.tk[n] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define QRK_STAMP_AT(slot) do { } while (0)
#define QRK_STAMP_IN(n) do { } while (0)
#endif

namespace qrk {

namespace pair {

constexpr int WR = 32;               // row registers per column
constexpr int LDP = WR + 2;          // LDS column stride in doubles: 272 B, conflict-free b64/b128 access
constexpr int RB = QRK_RB;           // the LDS image of A is refreshed every RB steps
// LDS carve-up per HALF (doubles)
constexpr int L_IMG = 0;             // [32][LDP] column-major image of A / staging for Q; R rows parked here
constexpr int L_XBUF = QRK_IMG_COLS * LDP;     // [32] current pivot column
constexpr int L_WBUF = L_XBUF + WR;  // [RB][32] update coefficients of the last RB steps, per A column
constexpr int L_POS = L_WBUF + RB * WR;   // [32] int: lane_of_pos
constexpr int L_A2 = L_POS + WR / 2;      // [1] |A|^2: squared norm of the first pivot column (scale of the decision margins)
constexpr int L_FLAG = L_A2 + 1;          // [1] != 0: a decision of this tile was not clear of rounding -> exact path
constexpr int L_HALF = L_FLAG + 1;        // 1266 doubles = 10128 B per half, 20256 B per wave -> 8 waves per CU (162 048 of 163 840 B)

constexpr double SQRT_EPS = 1.4901161193847656e-08;   // sqrt(DBL_EPSILON), Eigen's norm_downdate_threshold
// Decision margins (see "Decisions and the exact path" below): a squared column norm carried by this kernel and the square of
// Eigen's m_colNormsUpdated differ by rounding only, at most ~1000 eps normDirect^2 after 32 downdates (each adds
// <= 2 k eps normDirect^2 through the error of a_kj); decisions closer than MARGIN_M eps normDirect^2 = 2^14 eps normDirect^2
// are not taken here.  thr_hi = sqrt(eps) normDirect^2 (1 + 2^-12) is kept per lane, so MARGIN_M eps normDirect^2 = MREL thr_hi.
constexpr double MREL = 0.000244140625;               // 2^-12 = 2^14 eps / sqrt(eps)
constexpr double THR_HI = SQRT_EPS * (1.0 + MREL);
// The entries of a column carry an absolute error of ~k eps |A| (|A| = the largest column norm of the tile = the first pivot),
// whatever the column has shrunk to: the squared norms of two candidate columns are only known to 2^10 eps |A| (|c_b| + |c_j|),
// i.e. to 2^-42 |A| / |c| relative.  The hot path only looks at the HIGH WORDS of the squared norms: every live column within
// FILTER units (2^-12 relative) of the largest is a candidate, and a single candidate is a clear decision as long as
// 2^-42 |A| / |c| < 2^-12, which (5) guarantees: a pivot column below 2^-30 |A| flags the tile (checked once, in the epilogue,
// on the diagonal of R; generic tiles have |R_kk| > 1e-4 |A|).  (4): an |x0| below 2^9 eps |A| leaves the sign of beta to
// rounding noise; the hot path filters with |x0| <= 2^-13 |x| and the rare branch compares with |A|.  |A|^2 and the flag live
// in LDS (L_A2, L_FLAG): no register is spent on them.
constexpr int FILTER = 256;                            // units of the high word (2^-20 each)
constexpr double X0_FILTER2 = 1.4901161193847656e-08;  // 2^-26: x0^2 <= 2^-26 |x|^2 enters the rare branch ...
constexpr double X0_TINY2 = 1.2924697071141057e-26;    // ... where 2^-86: x0^2 <= (2^9 eps |A|)^2 flags
constexpr double PIV_TINY2 = 8.673617379884035e-19;    // 2^-60: |R_kk|^2 <= 2^-60 |A|^2
constexpr double ND_TINY2 = 5.820766091346741e-11;     // 2^-34: a column whose direct norm is below 2^-17 |A| meets the recompute
                                                       // test with an error that the 2^-12 band does not cover

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
// d0 = x*c0; d1 = x*c1 (starts the two chains without zeroing accumulators).
__device__ __forceinline__ void mul2_shared_a(double& d0, double& d1, double x, double c0, double c1)
{
    asm("v_mul_f64 %0, %2, %3\n\tv_mul_f64 %1, %2, %4" : "=&v"(d0), "=&v"(d1) : "v"(x), "v"(c0), "v"(c1));
}
// c0 += n0*x; c1 += n1*x.
__device__ __forceinline__ void fmac2_shared_b(double& c0, double& c1, double n0, double n1, double x)
{
    asm("v_fmac_f64_e32 %0, %2, %4\n\tv_fmac_f64_e32 %1, %3, %4" : "+v"(c0), "+v"(c1) : "v"(n0), "v"(n1), "v"(x));
}

// FP64 FMAs whose first factor is a DPP row_newbcast operand: X holds one element of the pivot column per lane (element
// lane & 15 of its row of 16), and row_newbcast:N hands lane N's element to all 16 lanes of the row -- the broadcast rides on
// the FMA itself (gfx90a+: the only DPP control the FP64 ALU accepts; measured at 5.25 cycles against 4.94 for the plain FMA,
// tools/ubench6.hip).  d0 += X[N]*c0; d1 += X[N]*c1.  Same products and the same order as the register form: results are
// bit-identical.
template <int N>
__device__ __forceinline__ void fmac2_bcast_a(double& d0, double& d1, double X, double c0, double c1)
{
    asm("v_fmac_f64_dpp %0, %2, %3 row_newbcast:%5 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f64_dpp %1, %2, %4 row_newbcast:%5 row_mask:0xf bank_mask:0xf"
        : "+v"(d0), "+v"(d1) : "v"(X), "v"(c0), "v"(c1), "n"(N));
}
template <int N>
__device__ __forceinline__ void fmac1_bcast_a(double& d0, double X, double c0)
{
    asm("v_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(d0) : "v"(X), "v"(c0), "n"(N));
}
// element N of the row's X in every lane of the row (the s_nop covers the VALU-write -> DPP-read hazard, which hipcc does not
// see through an asm statement)
template <int N>
__device__ __forceinline__ double bcast_f64(double X)
{
    double r;
    asm("s_nop 1\n\tv_mov_b64_dpp %0, %1 row_newbcast:%2 row_mask:0xf bank_mask:0xf" : "=v"(r) : "v"(X), "n"(N));
    return r;
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
    int sh8;         // 8 * half: bit offset of this half's field in packed wave-uniform words
    int hb4;         // 128 * half: ds_bpermute byte address of lane 0 of this half
    bool live;       // this lane's A column is not yet chosen as a pivot
    bool ispiv;      // this lane's column is the pivot of the step whose head (search_fetch) ran last
    int lbl;         // ... and the pivot lane of this half, 0..31
    int lpk;         // wave-uniform: pivot lane of half 0 | pivot lane of half 1 << 8
    unsigned long long livemask;   // the same as a wave-uniform lane mask
    int kstep;       // step at which this column was chosen (= its final position), 64 = not yet
    int rows, cols;  // tile shape of this half (rows/cols beyond are zero padding)
    double nu2;      // m_colNormsUpdated^2
    double thr_nd2;  // sqrt(eps) (1 + 2^-12) * m_colNormsDirect^2: upper edge of the band around Eigen's recompute threshold
    double h[RB];    // entry j of the pivot columns of the last RB steps
#if QRK_DPPX
    double xa, xb;   // the pivot column fetched last, spread over the lanes: xa = element (lane & 15), xb = element 16 + (lane & 15)
#endif
#ifdef QRK_STAMP
    unsigned long long tk[8];
#endif
};

__device__ __forceinline__ unsigned long long ballot64(bool p) { return __builtin_amdgcn_ballot_w64(p); }

__device__ __forceinline__ double bpermute_f64(int byte_addr, double v)
{
    const int lo = __builtin_amdgcn_ds_bpermute(byte_addr, __double2loint(v));
    const int hi = __builtin_amdgcn_ds_bpermute(byte_addr, __double2hiint(v));
    return __hiloint2double(hi, lo);
}

// Rare path of the pivot search: more than one live column has the high word of its squared norm within one unit of the
// largest.  Picks the largest (lowest lane among exact ties: any valid choice will do, see below) and checks the decision:
// a live column within the error margin of the chosen one means that Eigen's recurrence, rounded Eigen's way, may order the
// two differently (exact ties, which Eigen resolves by the CURRENT column positions of its swapped matrix, always land
// here) -- the tile is flagged and redone by the exact path (bdqr_exact.hip), which owns the first-maximum rule.
// Returns a one-hot (per half) pivot flag.
__device__ __forceinline__ bool resolve_near(int lane, bool live, double nu2, double thr_hi, int khi, int mh, int hb4,
                                             double* hl, bool have_scale)
{
    bool cand = live && khi == mh;
    const unsigned klo = (unsigned)__double2loint(nu2);
    const unsigned ml = half32_max_u32(cand ? klo : 0u);
    cand = cand && klo == ml;
    const unsigned long long pm = ballot64(cand);
    const unsigned tlo = (unsigned)pm, thi = (unsigned)(pm >> 32);
    const int lA = tlo ? __builtin_ctz(tlo) : 0, lB = thi ? __builtin_ctz(thi) : 0;
    const int lbl = (lane >> 5) ? lB : lA;
    const bool ispiv = cand && (lane & 31) == lbl;
    const double best = bpermute_f64((lbl << 2) + hb4, nu2), thrb = bpermute_f64((lbl << 2) + hb4, thr_hi);
    // margin: the downdate chain (MREL (thr_j + thr_b) = 2^14 eps (normDirect_j^2 + normDirect_b^2)) plus the absolute error of
    // the column entries, 2^10 eps |A| (|c_b| + |c_j|) <= 2^11 eps sqrt(|A|^2 best)
    double margin = MREL * (thr_hi + thrb);
    if (have_scale) margin += 4.547473508864641e-13 /* 2^-41 */ * __builtin_sqrt(hl[L_A2] * (best > 0.0 ? best : 0.0));
    if ((QRK_DECISIONS & 1) && live && !ispiv && nu2 >= best - margin) hl[L_FLAG] = 1.0;
    return ispiv;
}

// One step of ColPivHouseholderQR::computeInPlace (Eigen/src/QR/ColPivHouseholderQR.h) on both
// wave-resident tiles: pivot search, reflector, trailing update of A and of Q^T, norm downdate.
//
// Column norms are tracked SQUARED: Eigen's  temp = (1+t)(1-t), t = |a_kj|/normUpd;
// normUpd *= sqrt(temp)  is  nu2 <- max(nu2 - a_kj^2, 0), and its recompute test
// temp (normUpd/normDir)^2 <= sqrt(eps)  is  nu2_new <= sqrt(eps) normDir^2: the same quantities in exact arithmetic,
// without two FP64 divisions and a square root per step -- but NOT the same roundings.
//
// Decisions and the exact path.  Every data-dependent decision of the reference algorithm is taken here only when it is
// clear of rounding: (1) the pivot: no other live column within MREL (thr_hi_j + thr_hi_b) of the largest squared norm;
// (2) the LAWN-176 recompute test: the downdated norm not within 2^-12 (relative) of the threshold; (3) Eigen's degenerate
// reflector test tailSqNorm <= DBL_MIN: never true on a non-empty tail; (4) the sign of beta: |x0| > 2^9 eps |A|; (5) no pivot below 2^-30 |A|.  A tile that
// meets any of them is flagged (L_FLAG in LDS) and appended to the redo list at the end; bdqr_exact.hip then recomputes it with
// Eigen's own operation order and rounding.  On generic data nothing is flagged (a pivot inside its margin has probability
// ~1e-9 per tile); on sign / indicator / repeated-column data everything is, and the permutation is Eigen's either way.
//
// The kernel is bound by the number of VALU instructions a wave issues (two waves per SIMD), so the
// step is written to keep everything that is uniform per half out of VGPR selects: the pivot lane is
// a lane mask, its index reaches the lanes as one bit-field extract of a packed scalar, and |tail|^2
// comes from the pivot lane by ds_bpermute.  Columns that were already chosen are NOT masked out of
// the arithmetic: nothing below the diagonal of R is ever read, so their lanes compute garbage.
//
// The steps are software-pipelined: a lone wave stalls ~1100 cycles per step on the chain
// search -> image read -> publish -> broadcast read (tools/stamp_run.py), and with 232 registers only
// two waves share a SIMD.  So the head of step K+1 (search_fetch: pivot search, image read, corrections,
// publish) is issued inside step K, BEFORE the trailing update of step K, whose ~4(31-K) FMAs then
// cover those LDS round trips.  The update therefore works from the pivot column kept in registers
// (x[] below; XBUF already holds the next column), and search_fetch<K+1> sees an image that is one
// update older: it applies the corrections of steps ((K)/RB)*RB .. K (1..RB of them).
template <int K, bool FULL32, bool PIVOT>
__device__ __forceinline__ void search_fetch(double* hl /* this half's LDS */, LaneState& st)
{
    const int j = st.j;
    QRK_STAMP_IN(0);
    // this half still has a column to eliminate at step K (always, for 32x32 tiles)
    const bool act = FULL32 ? true : K < st.cols;
    bool ispiv;      // this lane's column is the pivot of step K
    int lbl;         // pivot lane of this lane's half, 0..31
    if (PIVOT) {
        // first maximum of the updated norms over the live columns, per half.  Non-negative doubles
        // order like their bit patterns, so the max is an integer max on the high word (chosen
        // columns carry a negative norm, see below) ...
        const int khi = __double2hiint(st.nu2);
        const int mh = half32_max_i32_fused(khi);
        // ... and every lane within one unit of it is a candidate (a unit of the high word is 2^-20 relative: far outside the
        // error margin, so a single candidate is a clear decision)
        unsigned long long pm;
        if (FULL32) {
            pm = __builtin_amdgcn_sicmp(khi, mh - ((QRK_DECISIONS & 1) ? FILTER : 0), 39 /* ICMP_SGE */);   // a live column exists: mh >= 0, chosen ones are < -1
            ispiv = khi >= mh - ((QRK_DECISIONS & 1) ? FILTER : 0);
        } else {
            ispiv = st.live && khi >= mh - FILTER;
            pm = ballot64(ispiv);
        }
        unsigned tlo = (unsigned)pm, thi = (unsigned)(pm >> 32);
        if (!(QRK_ABL & 1) && ((tlo & (tlo - 1u)) | (thi & (thi - 1u))) != 0u) {
            // ... unless there are several
            ispiv = resolve_near(st.lane, st.live, st.nu2, st.thr_nd2, khi, mh, st.hb4, hl, K > 0);
            pm = ballot64(ispiv);
            tlo = (unsigned)pm; thi = (unsigned)(pm >> 32);
        }
        const int lA = FULL32 ? __builtin_ctz(tlo) : (tlo ? __builtin_ctz(tlo) : 0);
        const int lB = FULL32 ? __builtin_ctz(thi) : (thi ? __builtin_ctz(thi) : 0);
        lbl = (int)__builtin_amdgcn_ubfe((unsigned)(lA | (lB << 8)), (unsigned)st.sh8, 5u);
        st.lpk = lA | (lB << 8);
        st.livemask &= ~pm;
        // a chosen column leaves the search: negative "norm" (it only decreases from here on)
        st.nu2 = __hiloint2double(ispiv ? (int)0xBF800000 : khi, __double2loint(st.nu2));
    } else {
        ispiv = act && j == K;   // HouseholderQR: column K
        lbl = K;
        st.lpk = K | (K << 8);
    }
    if (ispiv) { st.live = false; st.kstep = K; }
    st.ispiv = ispiv;
    st.lbl = lbl;

    // ---- pivot column: element (j, lbl) of the LDS image (exact through step KR-1) plus the rank-1
    // corrections of steps KR..K-1, then published for broadcast reads.  The image is refreshed after
    // the update of every RB-th step, i.e. after the search_fetch of the following step has run.
    constexpr int KR = K == 0 ? 0 : ((K - 1) / RB) * RB;
    {
        double xi = hl[L_IMG + QRK_IC(lbl) * LDP + j];
        QRK_STAMP_IN(1);
#pragma unroll
        for (int m = (QRK_ABL & 256) ? K : KR; m < K; ++m) xi = fma(hl[L_WBUF + (m % RB) * WR + lbl], st.h[m % RB], xi);
        if (!FULL32) xi = (act && j < st.rows) ? xi : 0.0;
        st.h[K % RB] = xi;
#if QRK_DPPX
        // lane j holds element j: v_permlane16_swap gives every lane the element of its partner row as well
        const auto rl = __builtin_amdgcn_permlane16_swap((unsigned)__double2loint(xi), (unsigned)__double2loint(xi), false, false);
        const auto rh = __builtin_amdgcn_permlane16_swap((unsigned)__double2hiint(xi), (unsigned)__double2hiint(xi), false, false);
        st.xa = __hiloint2double((int)rh[0], (int)rl[0]);
        st.xb = __hiloint2double((int)rh[1], (int)rl[1]);
#else
        hl[L_XBUF + j] = xi;
#endif
        QRK_STAMP_IN(2);
    }
}

// RDUMP: the persistent kernel reuses the row registers of finished rows for the prefetch of its next tile, so every four steps
// the four newest final entries of the lane's column (rows K-3..K of R) go to the lane's own image column, rows that no fetch
// looks at any more (a fetch at step k' uses rows > k' only).
template <int K, bool FULL32, bool PIVOT, bool HC, bool RDUMP>
__device__ __forceinline__ void pair_step(double (&a)[WR], double (&q)[WR], double* hl /* this half's LDS */,
                                          LaneState& st, double* __restrict__ hcoeffs_tile)
{
    const int j = st.j;
    const bool act = FULL32 ? true : K < st.cols;
    const bool ispiv = st.ispiv;     // set by search_fetch<K>
    const int lbl = st.lbl;

    // ---- d = x_tail^T c_tail for the A column and the Q^T column (pivot lane: dA = |x_tail|^2)
    const double ak = a[K], qk = q[K];
    double dA = 0.0, dQ = 0.0;
#if QRK_DPPX
    // the pivot column of THIS step (search_fetch<K+1> below replaces st.xa / st.xb before the trailing update runs)
    const double xa = st.xa, xb = st.xb;
    const double xk = bcast_f64<(K & 15)>(K < 16 ? xa : xb);
#define QRK_XROW(I) ((I) < 16 ? xa : xb)
    if (QRK_ABL & 32) { dA = xk; dQ = xk; }
    else {
#if QRK_DOT_SPLIT
    // one asm statement per FMA, the two chains alternating: gfx950's hazard recogniser pads an asm statement that reads a
    // register written by the asm statement right before it (s_nop 0 between every two steps of a chain otherwise)
#define QRK_DOT(I) if ((I) > K) { fmac1_bcast_a<((I) & 15)>(dA, QRK_XROW(I), a[I]); if (!(QRK_ABL & 2048)) fmac1_bcast_a<((I) & 15)>(dQ, QRK_XROW(I), q[I]); }
#else
#define QRK_DOT(I) if ((I) > K) { if (QRK_ABL & 2048) fmac1_bcast_a<((I) & 15)>(dA, QRK_XROW(I), a[I]); else fmac2_bcast_a<((I) & 15)>(dA, dQ, QRK_XROW(I), a[I], q[I]); }
#endif
        QRK_0_31(QRK_DOT)
#undef QRK_DOT
    }
#define QRK_TRAIL(I) if ((I) > K) { if (QRK_ABL & 2048) fmac1_bcast_a<((I) & 15)>(a[I], QRK_XROW(I), ngA); else fmac2_bcast_a<((I) & 15)>(a[I], q[I], QRK_XROW(I), ngA, ngQ); }
#else
    const double xk = hl[L_XBUF + K];
    double x[WR];                    // rows K+1.. of the pivot column (broadcast reads)
#pragma unroll
    for (int i = K + 1; i < WR; ++i) x[i] = hl[L_XBUF + i];
    if (QRK_ABL & 32) { dA = xk; dQ = xk; }
    else {
        if (QRK_ABL & 2048) {       // diagnostic: the A columns only
#pragma unroll
            for (int i = K + 1; i < WR; ++i) dA = fma(x[i], a[i], dA);
        } else {
#pragma unroll
        for (int i = K + 1; i < WR; ++i) {
            if (i == K + 1) mul2_shared_a(dA, dQ, x[i], a[i], q[i]);
            else fmac2_shared_a(dA, dQ, x[i], a[i], q[i]);
        }
        }
    }
#define QRK_TRAIL(I) if ((I) > K) { if (QRK_ABL & 2048) a[I] = fma(ngA, x[I], a[I]); else fmac2_shared_b(a[I], q[I], ngA, ngQ, x[I]); }
#endif

    QRK_STAMP_IN(3);
    // ---- makeHouseholder + applyHouseholderOnTheLeft (Eigen/src/Householder/Householder.h) in the
    // un-normalised form: with beta = -sign(x0) sqrt(x0^2 + |tail|^2) and w = beta - x0,
    //   tau = w/beta, essential = tail/(x0 - beta) = -tail/w, and for a column c with tail dot d
    //   gamma = (d - w c_k) / (beta w):   c_k <- c_k + w gamma  (= c_k - tau tmp),
    //                                     c_i <- c_i - gamma x_i (= c_i - tau ess_i tmp),
    // which needs one square root and one reciprocal (of beta*w > 0) per step and no division.
    // Kept here: nb = -beta = copysign(norm, x0), s = -w = nb + x0, ng = -1/(beta w).
#if QRK_TAILSQ_READLANE
    double tailSq;
    {
        const int lA = st.lpk & 31, lB = 32 + ((st.lpk >> 8) & 31);
        const int alo = __builtin_amdgcn_readlane(__double2loint(dA), lA), ahi = __builtin_amdgcn_readlane(__double2hiint(dA), lA);
        const int blo = __builtin_amdgcn_readlane(__double2loint(dA), lB), bhi = __builtin_amdgcn_readlane(__double2hiint(dA), lB);
        tailSq = __hiloint2double(st.half ? bhi : ahi, st.half ? blo : alo);
    }
#else
    const double tailSq = (QRK_ABL & 2) ? dA : bpermute_f64((lbl << 2) + st.hb4, dA);
#endif
    const double nrm2 = fma(xk, xk, tailSq);
    const double nrm = (QRK_ABL & 8) ? nrm2 : sqrt_pos(nrm2);
    // Eigen: if (c0 >= 0) beta = -beta; -0.0 counts as >= 0, hence the + 0.0
    double nb = __hiloint2double((__double2hiint(nrm) & 0x7fffffff) | (__double2hiint(xk + 0.0) & (int)0x80000000),
                                 __double2loint(nrm));
    double s = nb + xk;
    QRK_STAMP_IN(4);
    double ng = (QRK_ABL & 8) ? -(nb * s) : -recip(nb * s);
    // Eigen: tailSqNorm <= min() gives tau = 0, beta = x0, H = I.  Rare, so a real branch (the empty
    // asm keeps hipcc from flattening it into selects); s = 0 leaves c_k = x0 in the pivot lane.
    const bool degen = !act || !(tailSq > DBL_MIN);
    // decision (4) (filter: x0^2 <= 2^-26 |x|^2, with a non-empty tail); the scale of the tile goes to LDS at the first step
    if (K == 0 && ispiv) hl[L_A2] = nrm2;
    const unsigned long long dm = (FULL32 ? __builtin_amdgcn_fcmp(tailSq, DBL_MIN, 13 /* FCMP_ULE */) : ballot64(degen)) |
                                  ((QRK_DECISIONS & 2) ? __builtin_amdgcn_fcmp(xk * xk, X0_FILTER2 * nrm2, 13 /* FCMP_ULE */) : 0ull);
    bool setdiag = ispiv;
    if (!(QRK_ABL & 4) && __builtin_expect(dm != 0ull, 0)) {
        asm volatile("");
        // (3) a degenerate reflector on a non-empty tail, (4) a first entry too small to fix the sign of beta
        if ((QRK_DECISIONS & 18) && act && K + 1 < (FULL32 ? WR : st.rows) && (degen || xk * xk <= X0_TINY2 * hl[L_A2])) hl[L_FLAG] = 1.0;
        if (degen) { ng = 0.0; s = 0.0; setdiag = false; }   // (nrm may be NaN here: rsq(0) = inf)
    }
    if (HC) {
        if (hcoeffs_tile && ispiv) hcoeffs_tile[K] = -(s * s) * ng;   // tau = w/beta = w^2/(beta w)
    }

    const double ngA = fma(s, ak, dA) * ng;      // -gamma for the A column
    double an = fma(s, ngA, ak);
    if (setdiag) an = -nb;                       // R(k,k) = beta
    const double ngQ = (QRK_ABL & 2048) ? 0.0 : fma(s, qk, dQ) * ng;
    if (!(QRK_ABL & 2048)) q[K] = fma(s, ngQ, qk);
    QRK_STAMP_IN(5);
    hl[L_WBUF + (K % RB) * WR + j] = ngA;
    // Row K of R is final: it stays in the row register (never touched again: the steps that follow work on rows > K).
    a[K] = an;

    // ---- LAWN-176 norm downdate for the remaining columns (squared form, see above).
    // No clamp at zero: a negative value is <= the threshold and is recomputed exactly.
    bool updated = false;
    if (!(QRK_ABL & 128) && PIVOT && K + 1 < WR) {
        const double nn = fma(-an, an, st.nu2);
        st.nu2 = nn;
        const unsigned long long nm = __builtin_amdgcn_fcmp(nn, st.thr_nd2, 5 /* FCMP_OLE */) & st.livemask;
        if (__builtin_expect(nm != 0ull, 0)) {
            // rare: a column norm has to be recomputed from the updated column before the next search
            asm volatile("");
            // decision (2): inside the band [1 - 2^-12, 1 + 2^-12] around Eigen's threshold the test is rounding noise
            if ((QRK_DECISIONS & 4) && st.live && nn <= st.thr_nd2 &&
                (nn > st.thr_nd2 * (1.0 - 2.0 * MREL) || st.thr_nd2 <= (THR_HI * ND_TINY2) * hl[L_A2])) hl[L_FLAG] = 1.0;
            QRK_0_31(QRK_TRAIL)
            updated = true;
            const bool need = st.live && nn <= st.thr_nd2;
            double sq = 0.0;
#pragma unroll
            for (int i = K + 1; i < WR; ++i) sq = fma(a[i], a[i], sq);
            if (need) { st.nu2 = sq; st.thr_nd2 = sq * THR_HI; }
        }
    }

    // ---- head of the next step, then the trailing update of this one
    if (K + 1 < WR) search_fetch<(K + 1 < WR ? K + 1 : K), FULL32, PIVOT>(hl, st);
    if (!(QRK_ABL & 16) && !updated) { QRK_0_31(QRK_TRAIL) }
#undef QRK_TRAIL
#undef QRK_XROW
    QRK_STAMP_IN(6);

    if (RDUMP && K % 4 == 3) {
        *reinterpret_cast<double2*>(&hl[L_IMG + QRK_IC(j) * LDP + K - 3]) = make_double2(a[K - 3], a[K - 2]);
        *reinterpret_cast<double2*>(&hl[L_IMG + QRK_IC(j) * LDP + K - 1]) = make_double2(a[K - 1], a[K]);
    }
    // ---- refresh the LDS image of the live columns after every RB-th step
    if (!(QRK_ABL & 64) && K % RB == RB - 1 && K + 1 < WR) {
        if (st.live) {   // (the column just chosen for step K+1 is skipped: it was fetched already)
#pragma unroll
            for (int i = K + 1; i < WR; ++i) hl[L_IMG + QRK_IC(j) * LDP + i] = a[i];
        }
    }
    QRK_STAMP_IN(7);
}


// Packs the upper triangle of R of one half.  Lane j holds column p = kstep of R in c[0..p] (p = the step at which its column
// was chosen); the packed CSC value order of m_R (BlockDiagonalSparseQR.h:475-479) puts entry (i, p) at slot p(p+1)/2 + i.
// Every lane stores c[i] for i = 31 down to 0 WITHOUT a predicate: an entry with i > p lands in the slot of a later column,
// (i', p') with p' > p and i' < i, which is stored by a later instruction of this same wave (LDS executes a wave's accesses in
// order) and overwrites it; slots stay below 528 = 32 * 33 / 2.  32 stores per pair instead of a gather through the permutation
// with an integer square root per entry.  pk: 528 doubles of this half's LDS that nothing else uses any more.
__device__ __forceinline__ void pack_r_column(const double (&c)[WR], int kstep, double* pk)
{
    double* dst = pk + ((kstep * (kstep + 1)) >> 1);
#pragma unroll
    for (int i = WR - 1; i >= 0; --i) {
        dst[i] = c[i];
        // one store instruction per row, in this order: merged into ds_write2_b64 pairs, rows i and i + 1 would go out together
        // and the entry (0, 1) would race with the left-over of column 0
        asm volatile("" ::: "memory");
    }
}
}  // namespace pair

// FULL32: every tile is 32x32 and all arrays are 16-byte aligned (uniform batch).
// PIVOT: ColPivHouseholderQR (else HouseholderQR).  HC: also emit the Householder coefficients.
template <bool FULL32, bool PIVOT, bool HC>
__global__ void __launch_bounds__(64, 2)
bdqr_pair_kernel(WaveBatch nb, const double* __restrict__ tiles, double* __restrict__ q_vals,
                 double* __restrict__ r_vals, int32_t* __restrict__ perm,
                 double* __restrict__ hcoeffs, int32_t* __restrict__ redo_count, int32_t* __restrict__ redo_ids)
{
    using namespace pair;
    __shared__ __attribute__((aligned(16))) double lds[2 * L_HALF];
    // One pair per workgroup (no grid-stride loop: a loop makes hipcc hoist per-lane addresses out of
    // it and keep them in scratch across the whole factorisation).
    {
        const int64_t pi = blockIdx.x;
        const int lane = threadIdx.x;
        QRK_STAMP_AT(0);
        const int half = lane >> 5, j = lane & 31;
        double* hl = lds + half * L_HALF;
        const int64_t t = 2 * pi + half;
        const bool valid = t < nb.num_tiles;
        int r, c, cbase;
        int64_t toff, qoff, roff;
        int gid = (int)t;     // global tile index (what the redo list holds)
        if (FULL32) {
            r = 32; c = 32;
            toff = t * 1024; qoff = t * 1024; roff = t * 528; cbase = (int)(t * 32);
        } else if (nb.tile_ids) {
            const int gidx = nb.tile_ids[valid ? t : nb.num_tiles - 1];
            gid = gidx;
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
            } else {
                // missing partner of an odd last tile: diag(64..33) -- distinct norms, no tie-breaking;
                // nothing of it is stored
#pragma unroll 4
                for (int i = 0; i < WR; ++i) hl[L_IMG + j * LDP + i] = (i == j) ? (double)(64 - j) : 0.0;
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
            if (FULL32) a[i] = hl[L_IMG + j * LDP + i];
            else a[i] = (j < c && i < r) ? hl[L_IMG + j * LDP + i] : 0.0;
            q[i] = (i == j && j < r) ? 1.0 : 0.0;
        }
        // (no barrier: the image stays valid, it is the source of the pivot columns)

        LaneState st;
        st.lane = lane; st.j = j; st.half = half; st.kstep = 64; st.rows = r; st.cols = c;
        st.sh8 = half * 8; st.hb4 = half * 128;
        st.live = FULL32 ? true : j < c;
        st.livemask = FULL32 ? ~0ull : __builtin_amdgcn_ballot_w64(j < c);
        if (j == 0) { hl[L_FLAG] = 0.0; hl[L_A2] = 0.0; }
#pragma unroll
        for (int m = 0; m < RB; ++m) st.h[m] = 0.0;
        {
            // squared column norms (ColPivHouseholderQR: m_colNormsUpdated^2, sqrt(eps) m_colNormsDirect^2)
            double s = 0.0;
#pragma unroll
            for (int i = 0; i < WR; ++i) s = fma(a[i], a[i], s);
            st.nu2 = s;
            st.thr_nd2 = s * THR_HI;
        }
        double* hc_tile = (hcoeffs && valid) ? hcoeffs + cbase : nullptr;
        // number of steps = the larger column count of the two tiles
        const int c0 = __builtin_amdgcn_readlane(c, 0), c1 = __builtin_amdgcn_readlane(c, 32);
        const int cmax = c0 > c1 ? c0 : c1;

        // The k loop is expanded by the preprocessor: every row-register index is a compile-time
        // constant.  (A rolled loop dispatching through a uniform switch makes hipcc's CFG
        // structurizer copy the whole register tile at every merge point.)
#define QRK_STEP(K) if (!(QRK_ABL & 1024) && (FULL32 || K < cmax)) pair_step<K, FULL32, PIVOT, HC, false>(a, q, hl, st, hc_tile);
        search_fetch<0, FULL32, PIVOT>(hl, st);   // head of step 0; every step issues the head of the next one
#ifdef QRK_STAMP
        QRK_STAMP_AT(1);
        QRK_STEP(0) QRK_STEP(1) QRK_STEP(2) QRK_STEP(3) QRK_STEP(4) QRK_STEP(5) QRK_STEP(6) QRK_STEP(7)
        QRK_STAMP_AT(2);
        QRK_STEP(8) QRK_STEP(9) QRK_STEP(10) QRK_STEP(11) QRK_STEP(12) QRK_STEP(13) QRK_STEP(14) QRK_STEP(15)
        QRK_STAMP_AT(3);
        QRK_STEP(16) QRK_STEP(17) QRK_STEP(18) QRK_STEP(19) QRK_STEP(20) QRK_STEP(21) QRK_STEP(22) QRK_STEP(23)
        QRK_STAMP_AT(4);
        QRK_STEP(24) QRK_STEP(25) QRK_STEP(26) QRK_STEP(27) QRK_STEP(28) QRK_STEP(29) QRK_STEP(30) QRK_STEP(31)
        QRK_STAMP_AT(5);
#else
        QRK_0_31(QRK_STEP)
#endif
#undef QRK_STEP

        // ---- R: lane j holds column p = kstep of R in a[0..p]; the image is not needed any more and takes the packed triangle
        // (CSC value order of m_R, BlockDiagonalSparseQR.h:475-479)
        __syncthreads();
        if (j < c) {
            pack_r_column(a, st.kstep, hl + L_IMG);
            perm[cbase + st.kstep] = cbase + j;   // the column chosen at step k ends at position k: m_outputPerm_c.indices()(base_col+j) (:519-521)
        }
        __syncthreads();
        // decision (5): a pivot at the noise level of the tile, |R_kk|^2 <= 2^-60 |A|^2
        if (PIVOT && j < c) {
            const double rkk = hl[L_IMG + ((st.kstep * (st.kstep + 3)) >> 1)];
            if (rkk * rkk <= PIV_TINY2 * hl[L_A2]) hl[L_FLAG] = 1.0;
        }
        __syncthreads();
        // a decision inside its error margin: the tile is redone by the exact path (bdqr_exact.hip)
        if (redo_count && valid && j == 0 && hl[L_FLAG] != 0.0) redo_ids[atomicAdd(redo_count, 1)] = gid;
        if (FULL32) {
            if (valid) {
                double2* dst = reinterpret_cast<double2*>(r_vals + roff);
#pragma unroll
                for (int qq = 0; qq < 9; ++qq) {
                    const int e2 = j + 32 * qq;
                    if (e2 < 264) dst[e2] = *reinterpret_cast<const double2*>(&hl[L_IMG + 2 * e2]);
                }
            }
        } else {
            const int n_r = c * (c + 1) / 2;
            for (int e = j; e < n_r; e += 32) r_vals[roff + e] = hl[L_IMG + e];
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
        QRK_STAMP_AT(6);
#ifdef QRK_STAMP
        if (threadIdx.x == 0 && hcoeffs) reinterpret_cast<unsigned long long*>(hcoeffs)[(size_t)pi * 12 + 8] = blockIdx.x;
#endif
    }
}

namespace pair {
// Epilogue of one 32x32 pair: R (packed upper triangle) and perm.  The lane's own image column holds its column of R (rows
// 0..kstep; pair_step<..., RDUMP> put it there four rows at a time).  All per-lane addresses derive from an opaque lane id so
// that none of them is computed (and kept alive) before the factorisation.  Returns (per half, in every lane of the half)
// whether a decision of the tile was not clear of rounding.
__device__ __forceinline__ bool epilogue32(int lane_in, int64_t pi, int64_t num_tiles, int kstep, double* lds,
                                           double* __restrict__ r_vals, int32_t* __restrict__ perm, bool pivot)
{
    int lane = lane_in;
    asm volatile("" : "+v"(lane));
    const int half = lane >> 5, j = lane & 31;
    double* hl = lds + half * L_HALF;
    const int64_t t = 2 * pi + half;
    const bool valid = t < num_tiles;
    const int cbase = (int)(t * 32);
    double c[WR];
#pragma unroll
    for (int i = 0; i < WR; i += 2) {
        const double2 v = *reinterpret_cast<const double2*>(&hl[L_IMG + QRK_IC(j) * LDP + i]);
        c[i] = v.x; c[i + 1] = v.y;
    }
    // (one wave per workgroup: LDS is in order, the loads above are served before the stores below; no s_barrier, whose fence
    // would wait for the prefetch of the next tile)
    pack_r_column(c, kstep, hl + L_IMG);
    if (valid) perm[cbase + kstep] = cbase + j;        // the column chosen at step k ends at position k: m_outputPerm_c.indices()(base_col+j) (BlockDiagonalSparseQR.h:519-521)
    __builtin_amdgcn_wave_barrier();
    // decision (5): a pivot at the noise level of the tile, |R_kk|^2 <= 2^-60 |A|^2
    if (pivot && (QRK_DECISIONS & 8)) {
        const double rkk = hl[L_IMG + ((kstep * (kstep + 3)) >> 1)];
        if (rkk * rkk <= PIV_TINY2 * hl[L_A2]) hl[L_FLAG] = 1.0;
    }
    __builtin_amdgcn_wave_barrier();
    if (valid) {
        double2* dst = reinterpret_cast<double2*>(r_vals + t * 528);
#pragma unroll
        for (int qq = 0; qq < 9; ++qq) {
            const int e2 = j + 32 * qq;
            if (e2 < 264) dst[e2] = *reinterpret_cast<const double2*>(&hl[L_IMG + 2 * e2]);
        }
    }
    const bool flagged = valid && hl[L_FLAG] != 0.0;
    __builtin_amdgcn_wave_barrier();
    return flagged;
}

// Half of the rows of Q straight from the registers: lane j holds row j of Q_i (row-major rows are the
// CSR value order of m_Q in both FullQ ([U|N] split, BlockDiagonalSparseQR.h:455-471) and
// BlockDiagonalQ (:480-492) layouts), and entries q[0..k] are final after step k.  Each lane writes
// 128 contiguous bytes = one cache line of its row with eight 16-byte stores issued back to back
// (they meet again in the L2 line); no LDS staging, the registers are free at once, and the stores
// are spread over the factorisation instead of arriving as one burst at its end.
template <int FIRST, int COUNT = 16>
__device__ __forceinline__ void store_q_half(int lane_in, int64_t pi, int64_t num_tiles, const double (&q)[WR],
                                             double* __restrict__ q_vals)
{
    int lane = lane_in;
    asm volatile("" : "+v"(lane));
    const int64_t t = 2 * pi + (lane >> 5);
    if (t < num_tiles) {
        double2* dst = reinterpret_cast<double2*>(q_vals + t * 1024 + (lane & 31) * 32 + FIRST);
#pragma unroll
        for (int m = 0; m < COUNT / 2; ++m) dst[m] = make_double2(q[FIRST + 2 * m], q[FIRST + 2 * m + 1]);
    }
}

// A tile whose decisions were not clear of rounding, again, by the wave that factorised it: Eigen's own operation order and
// rounding (bdqr_exact_tile.h, 64 threads: lane = column, every sum over the rows a sequential chain), W and Q in the wave's
// LDS.  Bitwise what bdqr_exact_kernel computes.  Rare by construction (generic tiles are never flagged), so it runs after the
// persistent loop, where none of the loop's registers are live.
template <bool PIVOT>
__device__ __noinline__ void redo_exact32(int64_t t, double* lds, const double* __restrict__ tiles, double* __restrict__ q_vals,
                                          double* __restrict__ r_vals, int32_t* __restrict__ perm, double* __restrict__ hcoeffs)
{
    exact::Shared sh;
    double* W = exact::carve_shared<64>(reinterpret_cast<unsigned char*>(lds), 32, 32, sh);
    double* q = W + 1024;
#ifndef QRK_DIAG_LDS16
    static_assert((32 + 3 * 32 + 64) * 8 + (2 * 32 + 64) * 4 + 16 + 2 * 1024 * 8 <= 2 * L_HALF * 8, "the exact path of a 32 x 32 tile must fit the wave's LDS");
#endif
    __syncthreads();
    exact::tile_qr<PIVOT, 64>(32, 32, tiles + t * 1024, W, q, sh);
    exact::tile_store<64>(32, 32, (int)(t * 32), W, q, sh, perm, hcoeffs, r_vals + t * 528, q_vals + t * 1024);
    __syncthreads();
}
}  // namespace pair

// Uniform 32x32 batches (all arrays 16-byte aligned): persistent workgroups, software-pipelined
// against HBM.  With two 213-register waves per SIMD nothing else can hide the tile loads, so each
// wave fetches its NEXT tile while it factorises the current one: row register a[k] is dead after
// the R dump that follows step k, so after step 15 columns 0..15 of the next tile are loaded into a[0..15] (coalesced: lane j
// takes ROW j, 8 bytes per column -- 8-byte loads because 16-byte register tuples that live across
// the loop edge fragment the register file and spill), and after step 31 columns 16..31 into
// a[16..31].  At the top of the next round the rows go to the LDS image (which transposes them to
// lane = column) and the lane reads its column back.  Stores are fire-and-forget.
//
// Self-contained: a tile with a decision inside its error margin is redone by the wave itself with the exact-arithmetic
// routine after its last round (no redo list, no second kernel behind the launch).  The rounds run in chunks of 64 so that one
// 64-bit word per half remembers the flagged rounds.
template <bool PIVOT, bool HC>
__global__ void __launch_bounds__(64, QRK_MINW)
bdqr_pair32_kernel(int64_t num_tiles, const double* __restrict__ tiles, double* __restrict__ q_vals,
                   double* __restrict__ r_vals, int32_t* __restrict__ perm, double* __restrict__ hcoeffs)
{
    using namespace pair;
    __shared__ __attribute__((aligned(16))) double lds[2 * L_HALF];
    const int64_t npairs = (num_tiles + 1) / 2;
    constexpr int CHUNK = 64;

    for (int64_t pi0 = blockIdx.x; pi0 < npairs; pi0 += (int64_t)CHUNK * gridDim.x) {
    unsigned long long flagbits = 0ull;      // bit r: the tile of this half in round r of the chunk was flagged
    {
    double a[WR], q[WR];
    int64_t pi = pi0;
    {
        const int j = threadIdx.x & 31;
        const int64_t t = 2 * pi + (threadIdx.x >> 5);
        if (t < num_tiles) {
            const double* src = tiles + t * 1024 + j;
#pragma unroll
            for (int m = 0; m < WR; ++m) a[m] = src[32 * m];   // element (row j, column m)
        } else {
#pragma unroll
            for (int i = 0; i < WR; ++i) a[i] = 0.0;
        }
    }
    for (int round = 0; round < CHUNK && pi < npairs; ++round, pi += gridDim.x) {
        // Everything per-lane is re-derived from an opaque lane id in every round: otherwise hipcc
        // hoists the (loop-invariant) LDS addresses of the epilogue out of
        // the loop and keeps them in scratch across the factorisation.
        int lane = threadIdx.x;
        asm volatile("" : "+v"(lane));
        QRK_STAMP_AT(0);
        const int half = lane >> 5, j = lane & 31;
        double* hl = lds + half * L_HALF;
        const int64_t t = 2 * pi + half;
        const bool valid = t < num_tiles;
        const int cbase = (int)(t * 32);
        // the next pair of this wave (none after the last round of a chunk: the next chunk loads its own first pair)
        const int64_t pn = round + 1 < CHUNK ? pi + gridDim.x : npairs;

        // ---- stage: chunk m of the lane -> its place in the padded column-major image
#pragma unroll
        for (int m = 0; m < WR; ++m) hl[L_IMG + QRK_IC(m) * LDP + j] = a[m];
        if (!valid) {
            // missing partner of an odd last tile: diag(64..33) -- distinct norms, no tie-breaking;
            // nothing of it is stored
#pragma unroll 4
            for (int i = 0; i < WR; ++i) hl[L_IMG + QRK_IC(j) * LDP + i] = (i == j) ? (double)(64 - j) : 0.0;
        }
        __builtin_amdgcn_wave_barrier();   // one wave per workgroup: LDS is in order, no s_barrier (its fence would wait for the prefetch)
#pragma unroll
        for (int i = 0; i < WR; ++i) {
            a[i] = hl[L_IMG + QRK_IC(j) * LDP + i];
            q[i] = (i == j && valid) ? 1.0 : 0.0;
        }

        LaneState st;
        st.lane = lane; st.j = j; st.half = half; st.kstep = 64; st.rows = 32; st.cols = 32;
        st.sh8 = half * 8; st.hb4 = half * 128;
        st.live = true;
        st.livemask = ~0ull;
        if (j == 0) { hl[L_FLAG] = 0.0; hl[L_A2] = 0.0; }
#pragma unroll
        for (int m = 0; m < RB; ++m) st.h[m] = 0.0;
        {
            // squared column norms (ColPivHouseholderQR: m_colNormsUpdated^2, sqrt(eps) m_colNormsDirect^2)
            double s = 0.0;
#pragma unroll
            for (int i = 0; i < WR; ++i) s = fma(a[i], a[i], s);
            st.nu2 = s;
            st.thr_nd2 = s * THR_HI;
        }
        double* hc_tile = (HC && hcoeffs && valid) ? hcoeffs + cbase : nullptr;
        QRK_STAMP_AT(1);

#define QRK_STEP(K) pair_step<K, true, PIVOT, HC, true>(a, q, hl, st, hc_tile);
        search_fetch<0, true, PIVOT>(hl, st);     // head of step 0; every step issues the head of the next one
        // q[0..k] are final after step k: every QRK_QSTORE_EVERY steps the finished entries of the lane's
        // Q row go out (16-byte stores; the pieces of a cache line meet again in L2) and free their registers
#define QRK_QS(FIRST) { if (!(QRK_ABL & 2048)) store_q_half<FIRST, QRK_QSTORE_EVERY>(threadIdx.x, pi, num_tiles, q, q_vals); }
#define QRK_QS4(FIRST) if (QRK_QSTORE_EVERY == 4) QRK_QS(FIRST)
#define QRK_QS8(FIRST) if (QRK_QSTORE_EVERY == 4) QRK_QS(FIRST + 4) else if (QRK_QSTORE_EVERY == 8) QRK_QS(FIRST)
        QRK_STEP(0) QRK_STEP(1) QRK_STEP(2) QRK_STEP(3) QRK_QS4(0) QRK_STEP(4) QRK_STEP(5) QRK_STEP(6) QRK_STEP(7) QRK_QS8(0)
        QRK_STAMP_AT(2);
        QRK_STEP(8) QRK_STEP(9) QRK_STEP(10) QRK_STEP(11) QRK_QS4(8) QRK_STEP(12) QRK_STEP(13) QRK_STEP(14) QRK_STEP(15)
        QRK_STAMP_AT(3);
        if (QRK_QSTORE_EVERY == 16) { QRK_QS(0) } else { QRK_QS8(8) }
        {
            // columns 0..15 of this half-wave's tile of the next round -> a[0..15] (dead by now)
            int ln = threadIdx.x;
            asm volatile("" : "+v"(ln));
            const int64_t tn = 2 * pn + (ln >> 5);
            if (tn < num_tiles) {
                const double* nsrc = tiles + tn * 1024 + (ln & 31);
#pragma unroll
                for (int m = 0; m < 16; ++m) a[m] = nsrc[32 * m];
            }
        }
        QRK_STEP(16) QRK_STEP(17) QRK_STEP(18) QRK_STEP(19) QRK_QS4(16) QRK_STEP(20) QRK_STEP(21) QRK_STEP(22) QRK_STEP(23) QRK_QS8(16)
        QRK_STAMP_AT(4);
        QRK_STEP(24) QRK_STEP(25) QRK_STEP(26) QRK_STEP(27) QRK_QS4(24) QRK_STEP(28) QRK_STEP(29) QRK_STEP(30) QRK_STEP(31)
#undef QRK_STEP
        QRK_STAMP_AT(5);
        if (QRK_QSTORE_EVERY == 16) { QRK_QS(16) } else { QRK_QS8(24) }
#undef QRK_QS
#undef QRK_QS4
#undef QRK_QS8
        {
            // columns 16..31 of the next tile -> a[16..31]
            int ln = threadIdx.x;
            asm volatile("" : "+v"(ln));
            const int64_t tn = 2 * pn + (ln >> 5);
            if (tn < num_tiles) {
                const double* nsrc = tiles + tn * 1024 + (ln & 31);
#pragma unroll
                for (int m = 16; m < WR; ++m) a[m] = nsrc[32 * m];
            }
        }
        if (epilogue32(threadIdx.x, pi, num_tiles, st.kstep, lds, r_vals, perm, PIVOT)) flagbits |= 1ull << round;
        QRK_STAMP_AT(6);
#ifdef QRK_STAMP
        if (threadIdx.x == 0 && hcoeffs) {
            unsigned hwid;
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
            unsigned xcc;
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
            reinterpret_cast<unsigned long long*>(hcoeffs)[(size_t)pi * 12 + 7] = ((unsigned long long)xcc << 32) | hwid;
            reinterpret_cast<unsigned long long*>(hcoeffs)[(size_t)pi * 12 + 8] = blockIdx.x;
            for (int n = 0; n < 8; ++n)
                reinterpret_cast<unsigned long long*>(hcoeffs)[(size_t)npairs * 12 + (size_t)pi * 8 + n] = st.tk[n];
        }
#endif
    }
    }
#if !QRK_ABL
    // ---- the flagged tiles of this chunk, again, with the reference's own operation order (rare: generic data never gets here)
    {
        const unsigned f0lo = __builtin_amdgcn_readlane((unsigned)flagbits, 0), f0hi = __builtin_amdgcn_readlane((unsigned)(flagbits >> 32), 0);
        const unsigned f1lo = __builtin_amdgcn_readlane((unsigned)flagbits, 32), f1hi = __builtin_amdgcn_readlane((unsigned)(flagbits >> 32), 32);
        unsigned long long f[2] = {((unsigned long long)f0hi << 32) | f0lo, ((unsigned long long)f1hi << 32) | f1lo};
        if (__builtin_expect((f[0] | f[1]) != 0ull, 0)) {
            for (int h2 = 0; h2 < 2; ++h2) {
                unsigned long long m = f[h2];
                while (m) {
                    const int rnd = __builtin_ctzll(m);
                    m &= m - 1;
                    redo_exact32<PIVOT>(2 * (pi0 + (int64_t)rnd * gridDim.x) + h2, lds, tiles, q_vals, r_vals, perm, HC ? hcoeffs : nullptr);
                }
            }
        }
    }
#endif
    }
}
void launch_bdqr_pair(const WaveBatch& nb, bool full32, const double* tiles, double* q_vals,
                      double* r_vals, int32_t* perm, double* hcoeffs, int max_blocks,
                      int32_t* redo_count, int32_t* redo_ids, hipStream_t stream)
{
    if (nb.num_tiles <= 0) return;
#if QRK_ABL
    redo_count = nullptr;   // diagnostic builds compute garbage: never hand their tiles to the exact path
#endif
    const int64_t npairs = (nb.num_tiles + 1) / 2;
    const dim3 grid((unsigned)npairs), block(64);
#define QRK_LAUNCH(F, P, H)                                                                        \
    hipLaunchKernelGGL((bdqr_pair_kernel<F, P, H>), grid, block, 0, stream, nb, tiles, q_vals, r_vals, perm, hcoeffs, redo_count, redo_ids)
    const bool piv = nb.pivoting != 0, hc = hcoeffs != nullptr;
    if (full32) {
        // persistent: one workgroup per resident wave slot (2 waves per SIMD, 20 KB of LDS each)
        // (fewer workgroups with equal round counts -- 1667 x 3 pairs instead of 2048 x 2.44 -- measured
        // slower, 104 vs 91 us: the dispatcher does not spread a partial grid evenly over the CUs)
        const int64_t slots = max_blocks > 0 ? max_blocks : npairs;
        const int64_t nwg = npairs < slots ? npairs : slots;
        const dim3 pgrid((unsigned)nwg);
#define QRK_LAUNCH32(P, H)                                                                         \
    hipLaunchKernelGGL((bdqr_pair32_kernel<P, H>), pgrid, block, 0, stream, nb.num_tiles, tiles, q_vals, r_vals, perm, hcoeffs)
#if QRK_ABL || defined(QRK_STAMP)
        // diagnostic builds: the bench variant only; QRK_PAIR_PERSIST=0 selects the one-pair-per-workgroup kernel
        if (const char* e = std::getenv("QRK_PAIR_PERSIST")) {
            if (e[0] == '0') { QRK_LAUNCH(true, true, false); return; }
        }
        QRK_LAUNCH32(true, false);
        return;
#endif
        if (piv) { if (hc) QRK_LAUNCH32(true, true); else QRK_LAUNCH32(true, false); }
        else { if (hc) QRK_LAUNCH32(false, true); else QRK_LAUNCH32(false, false); }
#undef QRK_LAUNCH32
    } else {
        if (piv) QRK_LAUNCH(false, true, true); else QRK_LAUNCH(false, false, true);
    }
#undef QRK_LAUNCH
}

}  // namespace qrk
