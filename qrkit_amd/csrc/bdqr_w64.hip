// bdqr_w64.hip -- tiles with 32 < max(rows, cols) <= 64 (rows >= cols) of a block-diagonal matrix, factorised ON CHIP by ONE
// wavefront each: A_i P_i = Q_i R_i with explicit Q_i, for gfx950.
//
// Same reference seam as bdqr_pair.hip (the hot loop of QRKit::BlockDiagonalSparseQR::factorize,
// src/QRKit/BlockDiagonalSparseQR.h:432-526: blockSolver.compute(block) :437-438 -- Eigen ColPivHouseholderQR / HouseholderQR --,
// Qi = blockSolver.matrixQ() :446, the Q / R value assembly :455-500 and the column-permutation splice :519-521), for the size
// class that bdqr_col.hip used to serve from an LDS-resident working copy with one thread per column walking down its column
// (round 3: a 33 x 33 tile 11 x slower than a 32 x 32 one, 64 x 64 at 1.6 % of HBM).
//
// Layout: lane j owns column j of the tile in 64 row registers (the tile is aligned to the BOTTOM of a 64-row frame: padded row =
// row + 64 - rows, so the steps are instantiated per padded row and a shorter tile simply starts later in the unrolled sequence).
// No workgroup barrier anywhere: one wave per workgroup, LDS is in order.  Two phases, because A and Q do not fit the registers of a
// wave together (128 + 128 VGPRs, two waves per SIMD):
//   phase 1  A -> R: the step of bdqr_pair.hip (squared norms with the LAWN-176 downdate, integer arg-max on the high words, the
//            un-normalised reflector, decisions inside their error margin flag the tile for the exact path), except that the pivot
//            column reaches the lanes by being PUBLISHED: its lane writes it to LDS (16-byte stores of one lane), where it is also
//            what phase 2 needs -- reflector k.  Every lane then takes element (lane & 15) of each 16-row chunk into a register and
//            the dot product / rank-1 update read the column through the DPP row_newbcast operand of v_fmac_f64.  Row k of R stays
//            in row register k of its lane (later steps only touch the rows below); R leaves the registers at the end of the phase.
//   phase 2  Q = H_0 ... H_{c-1} I by BACKWARD accumulation (HouseholderSequence::evalTo's order) in the registers phase 1 has
//            freed: lane j owns column j of Q; step k reads reflector k from LDS the same way.  No search, no square root: a stream
//            of FMAs that the other wave of the SIMD, which is in another phase of another tile, overlaps with its latency chain.
// 17.9 KB of LDS per wave at 64 rows: nine tiles in flight per CU (round 5: tau no longer kept in LDS; eight before).
#include "qrk_device.h"

#include <float.h>
#include <cstdlib>

#ifndef QRK_W64_OWN
#define QRK_W64_OWN 1          // 1: the pivot lane sums the squares of its own tail and computes the reflector's scalars WHILE its column is on the
#endif                         // way to LDS (s, ng, tau travel with the column): the DPP row sum of |x_tail|^2 and the square-root / reciprocal
                               // chain leave the step's critical path, which the single-lane stores (650 cycles at 64 rows) cover
#ifndef QRK_W64_PIPELINE
#define QRK_W64_PIPELINE 0     // 1: the head of step k + 1 (search, publication) is issued before the trailing update of step k and the
                               // column goes out one update old (corrected on fetch).  Measured SLOWER (64 x 64: 14.4 vs 16.5 M tiles/s,
                               // 33 x 33: 27.0 vs 30.0, profiles/r04_w64_probe.txt): with two waves per SIMD the other wave already
                               // covers the stores, and the correction, the write-back of the corrected elements and the readlane are
                               // added instructions.  Kept behind the switch, parity-tested both ways.
#endif
// Diagnostic only: -DQRK_W64_PROF accumulates s_memtime ticks per phase of the step in workgroup 0 and prints them (never a timed build).
#ifdef QRK_W64_PROF
#define W64_TICK(z) do { const unsigned long long t1_ = __builtin_amdgcn_s_memtime(); st.pt[z] += t1_ - st.pt0; st.pt0 = t1_; } while (0)
#else
#define W64_TICK(z) do { } while (0)
#endif

namespace qrk {

namespace w64 {

using namespace decide;

constexpr int WR = 64;                   // rows of the frame = row registers per lane
constexpr int FILTER = 256;              // pivot candidates: high word of the squared norm within 2^-12 (relative) of the largest
constexpr double SQRT_EPS_HI = THR_HI;   // sqrt(eps) (1 + 2^-12)

// LDS (doubles): reflector KP (the pivot column of the step at padded row KP, as published) holds padded rows (KP & ~1) .. 63 at
// cb(KP): every column starts on a 16-byte boundary.
constexpr int cb(int kp) { int s = 0; for (int k = 0; k < kp; ++k) s += WR - (k & ~1); return s; }
constexpr int L_V = 0;
constexpr int L_S = cb(WR);              // [64] s = x0 - beta of the step at padded row KP
constexpr int L_NG = L_S + WR;           // [64] -1 / (beta (x0 - beta))
// (tau is not kept in LDS since round 5: the lane that computes it stores it to hCoeffs at once -- nothing on chip reads it again -- and
//  the 512 bytes are what a NINTH wave per CU was missing at 64 rows)
constexpr int L_TOTAL = L_NG + WR;       // 2240 doubles = 17 920 B
static_assert(cb(WR) == 2112 && (L_TOTAL * 8) * 9 <= 160 * 1024, "nine waves per CU");

#define QRK_W64_0_63(M)                                                                                                              \
    M(0) M(1) M(2) M(3) M(4) M(5) M(6) M(7) M(8) M(9) M(10) M(11) M(12) M(13) M(14) M(15) M(16) M(17) M(18) M(19) M(20) M(21) M(22)  \
    M(23) M(24) M(25) M(26) M(27) M(28) M(29) M(30) M(31) M(32) M(33) M(34) M(35) M(36) M(37) M(38) M(39) M(40) M(41) M(42) M(43)   \
    M(44) M(45) M(46) M(47) M(48) M(49) M(50) M(51) M(52) M(53) M(54) M(55) M(56) M(57) M(58) M(59) M(60) M(61) M(62) M(63)
#define QRK_W64_63_0(M)                                                                                                              \
    M(63) M(62) M(61) M(60) M(59) M(58) M(57) M(56) M(55) M(54) M(53) M(52) M(51) M(50) M(49) M(48) M(47) M(46) M(45) M(44) M(43)   \
    M(42) M(41) M(40) M(39) M(38) M(37) M(36) M(35) M(34) M(33) M(32) M(31) M(30) M(29) M(28) M(27) M(26) M(25) M(24) M(23) M(22)   \
    M(21) M(20) M(19) M(18) M(17) M(16) M(15) M(14) M(13) M(12) M(11) M(10) M(9) M(8) M(7) M(6) M(5) M(4) M(3) M(2) M(1) M(0)

__device__ __forceinline__ double sqrt_pos(double x)      // <= 1 ulp for positive normal x (bdqr_pair.hip)
{
    const double y = __builtin_amdgcn_rsq(x);
    double g = x * y, h = 0.5 * y;
    const double e = fma(-h, g, 0.5);
    g = fma(g, e, g);
    h = fma(h, e, h);
    const double d = fma(-g, g, x);
    return fma(d, h, g);
}
__device__ __forceinline__ double recip(double x)
{
    double y = __builtin_amdgcn_rcp(x);
    double e = fma(-x, y, 1.0);
    y = fma(y, e, y);
    e = fma(-x, y, 1.0);
    y = fma(y, e, y);
    return y;
}

// d += X[N] * c, X read through DPP row_newbcast (element N of the lane's row of 16 lanes), see bdqr_pair.hip
template <int N>
__device__ __forceinline__ void fmac_bcast(double& d, double X, double c)
{
    asm("v_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(d) : "v"(X), "v"(c), "n"(N));
}
// element N of the row's X in every lane of the row
template <int N>
__device__ __forceinline__ double bcast_f64(double X)
{
    double r;
    asm("s_nop 1\n\tv_mov_b64_dpp %0, %1 row_newbcast:%2 row_mask:0xf bank_mask:0xf" : "=v"(r) : "v"(X), "n"(N));
    return r;
}
// sum over every row of 16 lanes, the same bits in every lane of the row
__device__ __forceinline__ double row16_sum(double v)
{
    v += dpp_f64<0xB1>(v);
    v += dpp_f64<0x4E>(v);
    v += dpp_f64<0x141>(v);
    v += dpp_f64<0x140>(v);
    return v;
}
__device__ __forceinline__ double uniform_f64(double v)      // a wave-uniform value into scalar registers
{
    return __hiloint2double(__builtin_amdgcn_readfirstlane(__double2hiint(v)), __builtin_amdgcn_readfirstlane(__double2loint(v)));
}
// max over the wave, in scalar registers
__device__ __forceinline__ int wave_max_i32(int v)
{
    const int m = half32_max_i32_fused(v);
    const int a = __builtin_amdgcn_readlane(m, 0), b = __builtin_amdgcn_readlane(m, 32);
    return a > b ? a : b;
}
__device__ __forceinline__ unsigned wave_max_u32(unsigned v)
{
    const unsigned m = half32_max_u32(v);
    const unsigned a = (unsigned)__builtin_amdgcn_readlane((int)m, 0), b = (unsigned)__builtin_amdgcn_readlane((int)m, 32);
    return a > b ? a : b;
}

struct Lane {
    int lane;
    bool live;        // this lane's column of A is not yet chosen
    bool unclear;     // a decision of the tile was inside its error margin (any lane)
    int kstep;        // position at which this lane's column was chosen
    double nu2;       // m_colNormsUpdated^2 (a chosen column carries a negative value: it drops out of the integer arg-max)
    double thr;       // sqrt(eps) (1 + 2^-12) m_colNormsDirect^2
    double a2;        // |A|^2: squared norm of the first pivot column (scale of the decision margins); wave-uniform
    int P;            // pivot lane of the step whose head ran last (wave-uniform)
    double betap;     // (QRK_W64_OWN) beta of the step in which this lane's column was chosen: its R(k, k)
    double n2p;       // (QRK_W64_OWN) |x|^2 of that column
    double ngp;       // the coefficient the running step's pivot column got in the step before, if it was published before that update
    double* hc_out;   // (wave-uniform) hCoeffs of this tile, or null: tau of step k goes to hc_out[k] from the lane that computes it
#ifdef QRK_W64_PROF
    unsigned long long pt[16], pt0;
#endif
};

// The elements of the published column that this lane broadcasts: xc[m] = element 16 m + (lane & 15), for the chunks with rows below KP.
template <int KP>
__device__ __forceinline__ void load_chunks(const double* lds, int lane, double (&xc)[4])
{
    constexpr int M0 = (KP + 1) >> 4;                              // first chunk with a row > KP
    const double* vcol = lds + L_V + cb(KP) - (KP & ~1) + (lane & 15);
#pragma unroll
    for (int m = M0; m < 4; ++m) xc[m] = vcol[16 * m];
}

// Head of step k (KP = padded row of its diagonal): the pivot -- first maximum of the updated norms over the live columns -- and
// the PUBLICATION of its column, rows KP.., by its lane.  Non-negative doubles order like their bit patterns: integer max on the high
// words; every lane within FILTER units is a candidate, a single candidate is a clear decision.  The steps are software-pipelined
// (step<KP> issues the head of step KP + 1 BEFORE its own trailing update, whose FMAs then cover the single-lane LDS stores: 650
// cycles per step when they sat on the chain, profiles/r04_w64_step_profile.txt), so the column goes out one update old; `stale`
// says so and the fetch adds the missing rank-1 term (correct_chunks).
template <int KP, bool PIVOT>
__device__ __forceinline__ void search_publish(const double (&a)[WR], double* lds, Lane& st, const int k, const int rows)
{
    const int lane = st.lane;
    int P;
    if (PIVOT) {
        const int khi = __double2hiint(st.nu2);
        const int mh = wave_max_i32(khi);
        unsigned long long pm = __builtin_amdgcn_ballot_w64(khi >= mh - FILTER);
        if (__builtin_expect((pm & (pm - 1ull)) != 0ull, 0)) {
            // several: the largest (lowest lane among exact ties: any valid choice will do, the tile is flagged then) and the check
            // of the decision: a live column within the error margin of the chosen one sends the tile to the exact path, which owns
            // Eigen's first-maximum rule on the current positions
            asm volatile("");
            bool cand = st.live && khi == mh;
            const unsigned klo = (unsigned)__double2loint(st.nu2);
            const unsigned ml = wave_max_u32(cand ? klo : 0u);
            cand = cand && klo == ml;
            pm = __builtin_amdgcn_ballot_w64(cand);
            const int Pn = (int)__builtin_ctzll(pm);
            const double best = readlane_f64(st.nu2, Pn), thrb = readlane_f64(st.thr, Pn);
            double margin = MREL * (st.thr + thrb);
            if (k > 0) margin += 4.547473508864641e-13 /* 2^-41 */ * __builtin_sqrt(st.a2 * (best > 0.0 ? best : 0.0));
            if (st.live && lane != Pn && st.nu2 >= best - margin) st.unclear = true;
            pm = 1ull << Pn;
        }
        P = (int)__builtin_ctzll(pm);
        if (k == 0) st.a2 = readlane_f64(st.nu2, P);
    } else {
        P = k;
    }
    st.P = P;
    if (lane == P) {
        st.live = false; st.kstep = k;
        st.nu2 = __hiloint2double((int)0xBF800000, __double2loint(st.nu2));
        double* vcol = lds + L_V + cb(KP) - (KP & ~1);
#pragma unroll
        for (int i = KP & ~1; i < WR; i += 2) *reinterpret_cast<double2*>(&vcol[i]) = make_double2(a[i], a[i + 1]);
#if QRK_W64_OWN && !QRK_W64_PIPELINE
        __builtin_amdgcn_sched_barrier(0);                 // (the stores go out BEFORE the arithmetic below, which they then cover)
        // |x_tail|^2 from the lane's own registers, the decisions and makeHouseholder in the un-normalised form (see step)
        double q0 = 0.0, q1 = 0.0;
#define QRK_W64_SQ(I) if ((I) > KP) { if ((I) & 1) q1 = fma(a[I], a[I], q1); else q0 = fma(a[I], a[I], q0); }
        QRK_W64_0_63(QRK_W64_SQ)
#undef QRK_W64_SQ
        const double tsq = q0 + q1, xk = a[KP];
        st.n2p = fma(xk, xk, tsq);
        if (unclear_reflector(xk, tsq, k + 1 < rows, PIVOT, (k == 0 && !PIVOT) ? st.n2p : st.a2)) st.unclear = true;
        double beta, s, ng, tau;
        if (!(tsq > DBL_MIN)) { beta = xk; s = 0.0; ng = 0.0; tau = 0.0; }
        else {
            const double nrm = sqrt_pos(st.n2p);
            const double nbv = xk >= 0.0 ? nrm : -nrm;     // (-0.0 counts as >= 0, as in Eigen)
            beta = -nbv;
            s = nbv + xk;
            ng = -recip(nbv * s);
            tau = -(s * s) * ng;
        }
        st.betap = beta;
        lds[L_S + KP] = s; lds[L_NG + KP] = ng;
        if (st.hc_out) st.hc_out[k] = tau;
#endif
    }
}

// One step of ColPivHouseholderQR::computeInPlace (Eigen/src/QR/ColPivHouseholderQR.h) / HouseholderQR on the wave's tile; KP = padded
// row of the diagonal, k = KP - off the step; its head (search_publish<KP>) has run.  xp: the reflector of the step before (this
// lane's elements), st.ngp: the coefficient that step gave the pivot column of this one (0: the published column is up to date).
template <int KP, bool PIVOT>
__device__ __forceinline__ void step(double (&a)[WR], double* lds, Lane& st, double (&xp)[4], const int k, const int rows, const int cols)
{
    const int lane = st.lane;
    if (!QRK_W64_PIPELINE) search_publish<KP, PIVOT>(a, lds, st, k, rows);
    const bool ispiv = lane == st.P;
    __builtin_amdgcn_wave_barrier();
    W64_TICK(1);
    // ---- 3. the lanes' elements of the pivot column (+ the rank-1 term of the step before, when it went out one update old; the
    // corrected elements go back to LDS: they are reflector k of phase 2), |x_tail|^2 (every row of 16 lanes the same sum in the
    // same order), x0
#if QRK_W64_OWN && !QRK_W64_PIPELINE
    // (the pivot lane has computed the reflector's scalars beside its stores: search_publish)
    double xc[4] = {0.0, 0.0, 0.0, 0.0};
    constexpr int M0 = (KP + 1) >> 4;
    if (KP + 1 < WR) load_chunks<KP>(lds, lane, xc);
    const double s = uniform_f64(lds[L_S + KP]), ng = uniform_f64(lds[L_NG + KP]);
    const double beta = st.betap;                            // (meaningful in the pivot lane, the only one that uses it)
    if (k == 0 && !PIVOT) st.a2 = readlane_f64(st.n2p, st.P);
    W64_TICK(2);
#else
    double xc[4] = {0.0, 0.0, 0.0, 0.0};
    double tsq = 0.0, xk;
    constexpr int M0 = (KP + 1) >> 4, MK = KP >> 4;
    if (KP + 1 < WR) {
        load_chunks<KP>(lds, lane, xc);
        if (QRK_W64_PIPELINE) {
            const double ngp = st.ngp;
#pragma unroll
            for (int m = M0; m < 4; ++m) xc[m] = fma(ngp, xp[m], xc[m]);
            // (the chunk that holds the diagonal starts above the column's storage: only the rows the column owns go back)
            double* vcol = lds + L_V + cb(KP) - (KP & ~1) + lane;
#pragma unroll
            for (int m = M0; m < 4; ++m)
                if (lane < 16 && (m > MK || 16 * m + lane >= (KP & ~1))) vcol[16 * m] = xc[m];
        }
        double p = 0.0;
#pragma unroll
        for (int m = M0; m < 4; ++m) {
            const double x = (m == MK) ? (((lane & 15) > (KP & 15)) ? xc[m] : 0.0) : xc[m];
            p = fma(x, x, p);
        }
        tsq = uniform_f64(row16_sum(p));
    }
    if (MK >= M0) xk = uniform_f64(bcast_f64<(KP & 15)>(xc[MK]));
    else if (QRK_W64_PIPELINE) xk = uniform_f64(fma(st.ngp, readlane_f64(xp[MK], 15), lds[L_V + cb(KP) + (KP & 1)]));   // (row KP is the last one of its chunk: not among the loaded ones)
    else xk = uniform_f64(lds[L_V + cb(KP) + (KP & 1)]);
    W64_TICK(2);
    if (k == 0 && !PIVOT) st.a2 = fma(xk, xk, tsq);
    if (unclear_reflector(xk, tsq, k + 1 < rows, PIVOT, st.a2)) st.unclear = true;
    // ---- 4. makeHouseholder in the un-normalised form (bdqr_pair.hip): nb = -beta = copysign(norm, x0), s = x0 - beta,
    // ng = -1 / (beta (x0 - beta)); Eigen: tailSqNorm <= min() gives tau = 0, beta = x0, H = I.  The scalars go through the scalar
    // registers and the degenerate case is a real (uniform) branch: as per-lane values with selects the step becomes one basic block
    // whose schedule needs 256 VGPRs and 636 spills (64 x 64: 10.8 instead of 16.5 M tiles/s).
    double beta, s, ng, tau;
    if (!(tsq > DBL_MIN)) { beta = xk; s = 0.0; ng = 0.0; tau = 0.0; }
    else {
        const double nrm = sqrt_pos(fma(xk, xk, tsq));
        const double nbv = xk >= 0.0 ? nrm : -nrm;         // (-0.0 counts as >= 0, as in Eigen)
        beta = uniform_f64(-nbv);
        s = uniform_f64(nbv + xk);
        ng = uniform_f64(-recip(nbv * s));
        tau = -(s * s) * ng;
    }
    if (lane == 0) { lds[L_S + KP] = s; lds[L_NG + KP] = ng; if (st.hc_out) st.hc_out[k] = tau; }
#endif
    W64_TICK(3);
    // ---- 5. d = x_tail^T a_tail, the coefficient of the column, row k of R
    const double ak = a[KP];
    double d0 = 0.0, d1 = 0.0;
#pragma unroll
    for (int m = M0; m < 4; ++m) asm volatile("s_nop 1" : "+v"(xc[m]));      // (VALU write -> DPP read hazard, hidden from hipcc by the asm)
#define QRK_W64_DOT(I) if ((I) > KP) fmac_bcast<((I) & 15)>(((I) & 1) ? d1 : d0, xc[(I) >> 4], a[I]);
    QRK_W64_0_63(QRK_W64_DOT)
#undef QRK_W64_DOT
    const double ngam = fma(s, ak, d0 + d1) * ng;            // -gamma of this column
    double an = fma(s, ngam, ak);
    if (ispiv) an = beta;                                    // R(k, k)
    a[KP] = an;                                              // final: later steps work on the rows below
    W64_TICK(4);
    // the trailing update a_tail -= gamma x_tail (columns already chosen are not masked out: nothing below the diagonal of R is
    // ever read, and what they hold stays bounded -- the reflectors are orthogonal)
#define QRK_W64_UPD(I) if ((I) > KP) fmac_bcast<((I) & 15)>(a[I], xc[(I) >> 4], ngam);
#if QRK_W64_PIPELINE
    // ---- 6. LAWN-176 norm downdate (squared form; no clamp at zero: a negative value is <= the threshold and recomputed)
    bool updated = false;
    if (PIVOT && KP + 1 < WR) {
        const double nn = fma(-an, an, st.nu2);
        st.nu2 = nn;
        const bool need = st.live && nn <= st.thr;
        if (__builtin_expect(__builtin_amdgcn_ballot_w64(need) != 0ull, 0)) {
            // rare: a column norm has to be recomputed from the UPDATED column before the next search
            asm volatile("");
            if (need && in_recompute_band(nn, st.thr, st.a2)) st.unclear = true;      // decision (2)
            QRK_W64_0_63(QRK_W64_UPD)
            updated = true;
            double sq = 0.0;
#define QRK_W64_SQ(I) if ((I) > KP) sq = fma(a[I], a[I], sq);
            QRK_W64_0_63(QRK_W64_SQ)
#undef QRK_W64_SQ
            if (need) { st.nu2 = sq; st.thr = sq * SQRT_EPS_HI; }
        }
    }
    W64_TICK(5);
    // ---- 7. head of the next step, then the trailing update of this one
    if (KP + 1 < WR && k + 1 < cols) {
        search_publish<(KP + 1 < WR ? KP + 1 : KP), PIVOT>(a, lds, st, k + 1, rows);
        st.ngp = updated ? 0.0 : readlane_f64(ngam, st.P);
    }
    W64_TICK(0);
    if (!updated) { QRK_W64_0_63(QRK_W64_UPD) }
#pragma unroll
    for (int m = 0; m < 4; ++m) xp[m] = xc[m];
#else
    // ---- 6. the trailing update, then the LAWN-176 norm downdate (squared form; no clamp at zero: a negative value is <= the
    // threshold and recomputed from the updated column)
    QRK_W64_0_63(QRK_W64_UPD)
    W64_TICK(5);
    if (PIVOT && KP + 1 < WR) {
        const double nn = fma(-an, an, st.nu2);
        st.nu2 = nn;
        const bool need = st.live && nn <= st.thr;
        if (__builtin_expect(__builtin_amdgcn_ballot_w64(need) != 0ull, 0)) {
            asm volatile("");
            if (need && in_recompute_band(nn, st.thr, st.a2)) st.unclear = true;      // decision (2)
            double sq = 0.0;
#define QRK_W64_SQ(I) if ((I) > KP) sq = fma(a[I], a[I], sq);
            QRK_W64_0_63(QRK_W64_SQ)
#undef QRK_W64_SQ
            if (need) { st.nu2 = sq; st.thr = sq * SQRT_EPS_HI; }
        }
    }
    (void)cols; (void)xp;
#endif
#undef QRK_W64_UPD
    W64_TICK(6);
}

// Q_k = H_k Q_{k+1} on the wave's columns of Q: reflector k from LDS (x_tail as published, s, ng).
template <int KP>
__device__ __forceinline__ void back_step(double (&q)[WR], const double* lds, const int lane, Lane& st)
{
    constexpr int M0 = (KP + 1) >> 4;
    const double s = uniform_f64(lds[L_S + KP]), ng = uniform_f64(lds[L_NG + KP]);
    double xc[4] = {0.0, 0.0, 0.0, 0.0};
    if (KP + 1 < WR) load_chunks<KP>(lds, lane, xc);
    const double qk = q[KP];
    double d0 = 0.0, d1 = 0.0;
#pragma unroll
    for (int m = M0; m < 4; ++m) asm volatile("s_nop 1" : "+v"(xc[m]));
    W64_TICK(7);
#define QRK_W64_DOT(I) if ((I) > KP) fmac_bcast<((I) & 15)>(((I) & 1) ? d1 : d0, xc[(I) >> 4], q[I]);
    QRK_W64_0_63(QRK_W64_DOT)
#undef QRK_W64_DOT
    const double ngam = fma(s, qk, d0 + d1) * ng;
    q[KP] = fma(s, ngam, qk);
    W64_TICK(8);
#define QRK_W64_UPD(I) if ((I) > KP) fmac_bcast<((I) & 15)>(q[I], xc[(I) >> 4], ngam);
    QRK_W64_0_63(QRK_W64_UPD)
#undef QRK_W64_UPD
    W64_TICK(9);
}

}  // namespace w64

// PIVOT: ColPivHouseholderQR (else HouseholderQR).  One wave per workgroup; the workgroups take their tiles through `queue`
// (largest first: tile_ids is sorted by size).
// WPS: waves per SIMD the instantiation is compiled for.  The reflectors of a tile with r rows occupy the TAIL of the LDS layout
// (cb(64 - r) ..): the launch allocates only what its tallest tile needs (lds_shift = the unused head, in doubles), and tiles of up to
// 52 rows leave room for twelve waves per CU -- three per SIMD at the kernel's 167 registers -- instead of eight.
// R0: the first frame row the instantiation works on -- its tiles have at most WR - R0 rows, the row registers a[0 .. R0) do not exist for
// the register allocator and the steps above R0 are not instantiated: R0 = 20 (tiles of 33 .. 44 rows) fits 128 VGPRs and 10 KB of LDS per
// wave, FOUR waves per SIMD (round 5).
template <bool PIVOT, int WPS, int R0>
__global__ void __launch_bounds__(64, WPS)
bdqr_w64_kernel(WaveBatch nb, const double* __restrict__ tiles, double* __restrict__ q_vals, double* __restrict__ r_vals,
                int32_t* __restrict__ perm, double* __restrict__ hcoeffs, int32_t* __restrict__ redo_count,
                int32_t* __restrict__ redo_ids, int32_t* __restrict__ queue, int lds_shift)
{
    using namespace w64;
    extern __shared__ __attribute__((aligned(16))) double lds_dyn[];
    double* const lds = lds_dyn - lds_shift;

    for (int64_t t = blockIdx.x; t < nb.num_tiles;) {
        const int gidx = nb.tile_ids ? nb.tile_ids[t] : (int)t;
        int r, c, cbase;
        int64_t toff, qoff, roff;
        if (nb.t_rows) {
            r = nb.t_rows[gidx]; c = nb.t_cols[gidx];
            toff = nb.t_off[gidx]; qoff = nb.q_off[gidx]; roff = nb.r_off[gidx]; cbase = nb.c_off[gidx];
        } else {
            r = nb.rows; c = nb.cols;
            toff = t * (int64_t)r * c; qoff = t * (int64_t)r * r; roff = t * (int64_t)(c * (c + 1) / 2);
            cbase = (int)(t * c);
        }
        r = __builtin_amdgcn_readfirstlane(r); c = __builtin_amdgcn_readfirstlane(c);
        const int off = WR - r;                               // padded row of row 0
        // (per-lane values are re-derived from an opaque lane id in every phase: hipcc otherwise hoists loop-invariant address
        //  arithmetic out of the tile loop and keeps it in registers across the factorisation)
        int lane = threadIdx.x;
        asm volatile("" : "+v"(lane));

        Lane st;
        st.lane = lane; st.unclear = false; st.kstep = 0; st.a2 = 0.0; st.hc_out = hcoeffs ? hcoeffs + cbase : nullptr;
#ifdef QRK_W64_PROF
        for (int z = 0; z < 16; ++z) st.pt[z] = 0;
        st.pt0 = __builtin_amdgcn_s_memtime();
#endif
        {
            // =============== phase 1: A -> R ===============
            // wave priority by phase (s_setprio): the FMA stream of phase 2 ahead of the latency chain of phase 1 desynchronises the waves of
            // a SIMD: 56 x 56 20.9 -> 22.0 M tiles/s, 64 x 64 16.3 -> 17.0 M, smaller tiles and small batches unchanged; the reverse (1)
            // gains as much at 56 / 64 rows and loses 15 % at 33 x 33 (profiles/r04_w64_probe.txt)
#ifndef QRK_W64_PRIO
#define QRK_W64_PRIO 2
#endif
            if (QRK_W64_PRIO == 1) __builtin_amdgcn_s_setprio(2);      // (1: the chain of phase 1 ahead of the FMA stream of phase 2; 2: the reverse)
            if (QRK_W64_PRIO == 2) __builtin_amdgcn_s_setprio(0);
            double a[WR];
            const bool isA = lane < c;
            st.live = isA;
            {
                // lane j reads its column (unconditional loads from clamped addresses, then a select)
                const double* colp = tiles + toff + (int64_t)(isA ? lane : 0) * r;
#pragma unroll
                for (int i = R0; i < WR; ++i) {
                    const int row = i - off;
                    const double v = colp[row > 0 ? row : 0];     // (plain loads: the 64 row loads of a lane share its cache lines; non-temporal: 16.5 -> 13 M tiles/s)
                    a[i] = (isA && row >= 0) ? v : 0.0;
                }
            }
            {
                double s0 = 0.0, s1 = 0.0;
#pragma unroll
                for (int i = R0; i < WR; i += 2) { s0 = fma(a[i], a[i], s0); s1 = fma(a[i + 1], a[i + 1], s1); }
                const double s = s0 + s1;
                st.nu2 = isA ? s : -1.0;
                st.thr = s * SQRT_EPS_HI;
            }
            W64_TICK(10);
            double xp[4] = {0.0, 0.0, 0.0, 0.0};
            st.ngp = 0.0; st.P = 0;
#define QRK_W64_HEAD(KP) if ((KP) >= R0 && QRK_W64_PIPELINE && (KP) == off) search_publish<KP, PIVOT>(a, lds, st, 0, r);
            QRK_W64_0_63(QRK_W64_HEAD)
#undef QRK_W64_HEAD
#define QRK_W64_STEP(KP) if ((KP) >= R0 && (KP) >= off && (KP) - off < c) step<KP, PIVOT>(a, lds, st, xp, (KP) - off, r, c);
            QRK_W64_0_63(QRK_W64_STEP)
#undef QRK_W64_STEP

            W64_TICK(15);
            // ---- R: lane j holds column p = kstep of R in padded rows off .. off + p; the packed CSC value order of m_R
            // (BlockDiagonalSparseQR.h:475-479) puts entry (i, p) at p (p + 1) / 2 + i -- a contiguous run per lane, stored straight
            // from the registers; the permutation splice (:519-521): the column chosen at step p ends at position p
            int ln = threadIdx.x;
            asm volatile("" : "+v"(ln));
            if (ln < c) {
                const int p = st.kstep;
                perm[cbase + p] = cbase + ln;
                double* dst = r_vals + roff + ((p * (p + 1)) >> 1) - off;
#pragma unroll
                for (int i = R0; i < WR; ++i)
                    if (i >= off && i - off <= p) dst[i] = a[i];
            }
            // a decision inside its error margin: the tile is redone by the exact path (bdqr_exact.hip)
            if (__builtin_amdgcn_ballot_w64(st.unclear) != 0ull && redo_count && ln == 0) redo_ids[atomicAdd(redo_count, 1)] = gidx;
        }
        W64_TICK(11);
        {
            // =============== phase 2: Q = H_0 ... H_{c-1}, backward ===============
            if (QRK_W64_PRIO == 1) __builtin_amdgcn_s_setprio(0);
            if (QRK_W64_PRIO == 2) __builtin_amdgcn_s_setprio(2);
            int ln = threadIdx.x;
            asm volatile("" : "+v"(ln));
            double q[WR];
#pragma unroll
            for (int i = R0; i < WR; ++i) q[i] = (i == ln + off) ? 1.0 : 0.0;      // (lanes >= rows: zero columns, never stored)
#define QRK_W64_BACK(KP) if ((KP) >= R0 && (KP) >= off && (KP) - off < c) back_step<KP>(q, lds, ln, st);
            QRK_W64_63_0(QRK_W64_BACK)
#undef QRK_W64_BACK
            W64_TICK(15);
            // row-major rows of Q_i are the CSR value order of m_Q in both FullQ ([U|N] split, :455-471) and BlockDiagonalQ
            // (:480-492) layouts: one coalesced store per row
            if (ln < r) {
                double* dst = q_vals + qoff + ln;
#pragma unroll
                for (int i = R0; i < WR; ++i)
                    if (i >= off) dst[(int64_t)(i - off) * r] = q[i];
            }
        }
        W64_TICK(12);
#ifdef QRK_W64_PROF
        if (blockIdx.x == 0 && threadIdx.x == 0)
            printf("w64 prof %d x %d (s_memtime ticks per tile): load + norms %llu | steps: search %llu  publish %llu  chunks + |x|^2 %llu  scalars %llu  dot %llu  "
                   "update %llu  downdate %llu | R out %llu | back: loads %llu  dot %llu  update %llu | Q out %llu\n", r, c, st.pt[10], st.pt[0], st.pt[1],
                   st.pt[2], st.pt[3], st.pt[4], st.pt[5], st.pt[6], st.pt[11], st.pt[7], st.pt[8], st.pt[9], st.pt[12]);
#endif
        // next tile (the LDS of this one is dead: every lane is past its last read)
        int nxt = 0;
        if (threadIdx.x == 0) nxt = (int)gridDim.x + atomicAdd(queue, 1);
        t = __builtin_amdgcn_readfirstlane(nxt);
    }
}

bool bdqr_w64_supported(int rows, int cols) { return rows >= cols && rows <= w64::WR && rows > 32; }

// Tiles with 32 < rows <= 64, cols <= rows.  queue: one int32 (zeroed here) through which the workgroups take their next tile.
// max_rows: the tallest tile of the launch (33 .. 64); num_cus: the grid is 8, 12 or 16 workgroups per CU by what the LDS of the launch allows (9 .. 11 resident ones run the 12-per-CU grid).
hipError_t launch_bdqr_w64(const WaveBatch& nb, const double* tiles, double* q_vals, double* r_vals, int32_t* perm, double* hcoeffs,
                           int num_cus, int max_rows, int32_t* redo_count, int32_t* redo_ids, int32_t* queue, hipStream_t stream)
{
    if (nb.num_tiles <= 0) return hipSuccess;
    if (max_rows < 33 || max_rows > w64::WR) return hipErrorInvalidValue;
    if (hipError_t e = hipMemsetAsync(queue, 0, sizeof(int32_t), stream)) return e;
    // the reflector of padded row KP starts at cb(KP) and load_chunks reads up to 15 doubles below the column it fetches
    int shift = w64::cb(w64::WR - max_rows) - 16;
    if (shift < 0) shift = 0;
    const size_t lds_bytes = (size_t)(w64::L_TOTAL - shift) * sizeof(double);
    static const int wps_env = std::getenv("QRK_W64_WPS") ? std::atoi(std::getenv("QRK_W64_WPS")) : 0;      // (diagnostic: 2 / 3 cap the occupancy)
    int per_cu = (int)((size_t)(160 * 1024) / lds_bytes);     // waves of one CU by LDS: 8 (64 rows) .. 16
    const int cap = wps_env == 2 ? 8 : (wps_env == 3 ? 12 : 16);
    per_cu = per_cu > cap ? cap : (per_cu < 8 ? 8 : per_cu);
    // four waves per SIMD: the instantiation without the frame rows above 20 (tiles of at most 44 rows), whose LDS leaves room for 16 waves
    const bool wps4 = per_cu >= 16 && max_rows <= w64::WR - 20;
    if (!wps4 && per_cu > 12) per_cu = 12;
    const bool wps3 = !wps4 && per_cu > 8;
    if (wps3) per_cu = 12; else if (!wps4) per_cu = 8;
    const int64_t num_wg = (int64_t)num_cus * per_cu;
    const int64_t want = nb.num_tiles < num_wg ? nb.num_tiles : num_wg;
#define QRK_W64_LAUNCH(P, W, R) hipLaunchKernelGGL((bdqr_w64_kernel<P, W, R>), dim3((unsigned)want), dim3(64), lds_bytes, stream, nb, tiles, q_vals, \
                                                   r_vals, perm, hcoeffs, redo_count, redo_ids, queue, shift)
    if (nb.pivoting) { if (wps4) QRK_W64_LAUNCH(true, 4, 20); else if (wps3) QRK_W64_LAUNCH(true, 3, 0); else QRK_W64_LAUNCH(true, 2, 0); }
    else { if (wps4) QRK_W64_LAUNCH(false, 4, 20); else if (wps3) QRK_W64_LAUNCH(false, 3, 0); else QRK_W64_LAUNCH(false, 2, 0); }
#undef QRK_W64_LAUNCH
    return hipGetLastError();
}

}  // namespace qrk
