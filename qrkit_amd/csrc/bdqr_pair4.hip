// bdqr_pair4.hip -- uniform batches of 32 x 32 tiles, TWO tiles per wavefront and FOUR wavefronts per SIMD: A_i P_i = Q_i R_i with
// explicit Q_i, for gfx950.  The second generation of the headline kernel (bdqr_pair.hip is the first; QRK_PAIR_V2 selects).
//
// Same reference seam as bdqr_pair.hip: the body of the hot loop of QRKit::BlockDiagonalSparseQR::factorize
// (src/QRKit/BlockDiagonalSparseQR.h:432-526): blockSolver.compute(block) (:437-438, Eigen ColPivHouseholderQR / HouseholderQR),
// Qi = blockSolver.matrixQ() (:446), the Q / R value assembly (:455-500) and the column-permutation splice (:519-521).
//
// Why a second generation.  bdqr_pair.hip is bounded by the dependent chain of its steps, not by memory or issue: a pair that has its
// SIMD to itself takes 41 790 cycles, two per SIMD overlap almost completely, and 10 000 tiles are three rounds of that chain
// (profiles/r04_k1_critical_path.txt: 0.37 of HBM at any schedule).  More chains per SIMD are what it needs, and two resources cap it at
// two waves per SIMD: the LDS image of A through which the pivot column is fetched (8.7 KB per tile) and the 128 data registers of a
// wave that carries A and Q^T together.  This kernel is bdqr_w64.hip's design at 32 rows:
//   * no image: the pivot lane of each half PUBLISHES its column to LDS (8-byte stores of two lanes), every lane takes element
//     lane & 15 of the two 16-row chunks, and the dot / update FMAs read it through the DPP row_newbcast operand; |x_tail|^2 is the
//     pivot lane's own dot product, handed to its half through one LDS word; the published column is also reflector k of phase 2
//     -- 4.75 KB of LDS per tile;
//   * two phases in the same registers: A -> R (row k of R stays in row register k of its lane and leaves at the end of the phase as
//     one contiguous run per lane), then Q = H_0 ... H_31 by backward accumulation (HouseholderSequence::evalTo's order) from the
//     reflectors in LDS -- 64 data registers per wave instead of 128;
//   * <= 128 VGPRs and 10 KB of LDS per wave: sixteen waves per CU, 4 096 pairs resident on the chip: 10 000 tiles are ONE round and a
//     fifth of a second one, spread evenly over the SIMDs.
// Decisions and the exact path exactly as bdqr_pair.hip ("Decisions and the exact path" there): integer arg-max on the high words
// with a filter, margins, LAWN-176 band, degenerate reflector, noise-level pivot; a flagged tile is redone by the wave itself with the
// exact-arithmetic routine after its rounds (working copy in a global scratch: the LDS of this kernel is too small for it).
#include "qrk_device.h"
#include "bdqr_exact_tile.h"

#include <float.h>
#include <cstdlib>

namespace qrk {

namespace p4 {

using namespace decide;

// Diagnostic only (tools/p4_stamps.py): -DQRK_P4_STAMP records s_memrealtime (100 MHz, one clock for the whole chip) of every pair at
// the start of its round, when its tiles are in registers, at the end of phase 1 and at the end -- in the array passed as `hcoeffs`
#ifdef QRK_P4_STAMP
#define QRK_P4_STAMP_AT(slot) do { if (threadIdx.x == 0) reinterpret_cast<long long*>(hcoeffs)[pi * 4 + (slot)] = (long long)__builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define QRK_P4_STAMP_AT(slot) do { } while (0)
#endif

constexpr int WR = 32;
constexpr int FILTER = 256;              // pivot candidates: high word of the squared norm within 2^-12 (relative) of the largest
// LDS per HALF (doubles): reflector K (the pivot column of step K, as published) holds rows (K & ~1) .. 31 at cb(K): 16-byte aligned
constexpr int cb(int k) { int s = 0; for (int q = 0; q < k; ++q) s += WR - (q & ~1); return s; }
constexpr int L_V = 0;
constexpr int L_S = cb(WR);              // [32] s = x0 - beta
constexpr int L_NG = L_S + WR;           // [32] 1 / (beta (beta - x0)) (the sign of -gamma goes into the multiply's source modifier)
constexpr int L_TAU = L_NG + WR;         // [32]
constexpr int L_HALF = L_TAU + WR;       // 640 doubles = 5 120 B per half, 10 240 B per wave: 16 waves per CU
static_assert(cb(WR) == 544 && L_HALF * 8 * 2 * 16 <= 160 * 1024, "sixteen waves per CU");
constexpr int STAGE_LD = WR + 2;         // the staging of a tile (lane = two rows -> lane = column) uses [32][34] doubles of the wave's LDS
static_assert(WR * STAGE_LD <= 2 * L_HALF, "the staging buffer fits the wave's LDS");

// QRK_P4_NT: 1 = the tiles are loaded, 2 = Q and R are stored, 3 = both with the non-temporal hint.  Measured (profiles/r04_p4_nt.txt):
// loads 78.0 -> 74.3 us per 10 000 tiles (the input does not displace the results being merged in L2); stores 130 us (16-byte pieces of
// R no longer merge)
// Round 6: bit 4 = the rows of Q alone with the hint (every store instruction fills whole 128-byte lines, unlike the pieces of R): 1 250 tiles
// 25.9 -> 25.0 us, 5 000 46.4 -> 45.6, 10 000 72.6 -> 72.2, 160 000 923 -> 920 (two interleaved A/B runs each, profiles/r06_k1_ramp.txt)
#ifndef QRK_P4_NT
#define QRK_P4_NT 5
#endif
#if QRK_P4_NT & 1
#define QRK_P4_LOAD(p) __builtin_nontemporal_load(p)
#else
#define QRK_P4_LOAD(p) (*(p))
#endif
#if QRK_P4_NT & 2
#define QRK_P4_STORE(v, p) __builtin_nontemporal_store((v), (p))
#else
#define QRK_P4_STORE(v, p) (*(p) = (v))
#endif

#if QRK_P4_NT & 4
#define QRK_P4_STOREQ(v, p) __builtin_nontemporal_store((v), (p))
#else
#define QRK_P4_STOREQ(v, p) (*(p) = (v))
#endif

#ifndef QRK_P4_EXACT_LDS
#define QRK_P4_EXACT_LDS 2
#endif
// Round 5 (profiles/r05_k1_experiments.txt; every switch that was measured and NOT kept is in tools/experiments/bdqr_pair4_r5_switches.hip.txt):
// with four waves per SIMD the kernel sits on TWO resources at once, the VALU issue port and the CU's one LDS pipe, both ~80 % busy
// (tools/ubench8.hip: a ds_write_b128 takes the LDS 22.5 ns per SIMD whether one lane or 64 are active, a ds_read2_b64 13.6 ns, against
// 2.2 ns per VALU instruction).  Kept: the pivot lane publishes with 8-byte stores (10.3 ns against 22.5 for 16 bytes), the two chunks of a
// column come by two ds_read_b64 (3.7 ns each), phase 2 takes s_K and 1 / (beta (beta - x0)) from registers through DPP instead of 32 LDS
// reads, the decision masks come straight from v_cmp, R(K, K) is the updated entry itself, and launches of at most half a round or of
// two rounds and more load every lane's column directly (no staging through LDS).
#ifndef QRK_P4_PRIO
#define QRK_P4_PRIO 0
#endif
// doubles of global scratch per workgroup: what the exact routine does not keep in LDS (its working copy; with QRK_P4_EXACT_LDS != 2 also Q)
constexpr int EXACT_SCRATCH = QRK_P4_EXACT_LDS == 2 ? 1024 : 2048;

#define QRK_P4_0_31(M)                                                                           \
    M(0) M(1) M(2) M(3) M(4) M(5) M(6) M(7) M(8) M(9) M(10) M(11) M(12) M(13) M(14) M(15) M(16)  \
    M(17) M(18) M(19) M(20) M(21) M(22) M(23) M(24) M(25) M(26) M(27) M(28) M(29) M(30) M(31)
#define QRK_P4_31_0(M)                                                                           \
    M(31) M(30) M(29) M(28) M(27) M(26) M(25) M(24) M(23) M(22) M(21) M(20) M(19) M(18) M(17)    \
    M(16) M(15) M(14) M(13) M(12) M(11) M(10) M(9) M(8) M(7) M(6) M(5) M(4) M(3) M(2) M(1) M(0)

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
// d += X[N] * c, X read through DPP row_newbcast (element N of the lane's row of 16 lanes)
template <int N>
__device__ __forceinline__ void fmac_bcast(double& d, double X, double c)
{
    asm("v_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(d) : "v"(X), "v"(c), "n"(N));
}
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
__device__ __forceinline__ double bpermute_f64(int byte_addr, double v)
{
    const int lo = __builtin_amdgcn_ds_bpermute(byte_addr, __double2loint(v));
    const int hi = __builtin_amdgcn_ds_bpermute(byte_addr, __double2hiint(v));
    return __hiloint2double(hi, lo);
}

typedef __attribute__((address_space(3))) double lds_f64;      // (volatile accesses through a generic pointer would become flat_*)

struct Lane {
    int lane, j, half;
    unsigned long long livemask;   // (wave-uniform) the lanes whose column of A is not yet chosen
    unsigned long long unclearm;   // (wave-uniform) lanes that saw a decision of their tile inside its error margin
    int kstep;        // position at which this lane's column was chosen
    double nu2;       // m_colNormsUpdated^2 (a chosen column carries a negative value)
    double thr;       // sqrt(eps) (1 + 2^-12) m_colNormsDirect^2
    double a2;        // |A|^2 of this half's tile: squared norm of its first pivot column (scale of the decision margins)
};

// The elements of the published column that this lane broadcasts: xc[m] = element 16 m + (lane & 15) of the half's column
template <int K>
__device__ __forceinline__ void load_chunks(const double* hl, int lane, double (&xc)[2])
{
    constexpr int M0 = (K + 1) >> 4;
    const double* vcol = hl + L_V + cb(K) - (K & ~1) + (lane & 15);
#pragma unroll
    for (int m = M0; m < 2; ++m) xc[m] = *(const volatile lds_f64*)(vcol + 16 * m);     // (two ds_read_b64: cheaper than one ds_read2_b64, tools/ubench8.hip)
}

// One step of ColPivHouseholderQR::computeInPlace / HouseholderQR on both tiles of the wave (see bdqr_pair.hip for the arithmetic: squared
// norms, un-normalised reflector, decisions).
// OWN: the pivot lane itself sums the squares of its tail and computes the reflector's scalars while its column is on the way to LDS, and
// s, ng travel with the column: the step loses the |x_tail|^2 hand-off (an LDS write and read behind the dot product) for 31 - K more
// FMAs.  The sums are the same products in the same order as the pivot lane's dot product of the other form: bitwise the same factors.
// Built in round 5 as a LATENCY form for small launches -- and measured the other way round (profiles/r05_k1_own_norm.txt): a launch of
// at most one round is 2-4 % SLOWER with it (1 250 tiles 26.2 -> 27.1 us: the chain it removes is not the critical one there), a launch
// of more than one round 1-1.6 % faster (10 000 tiles 72.3 -> 71.4 us, 100 000 tiles 581 -> 571: two LDS instructions fewer per step
// where the LDS pipe is the co-bottleneck).  The launcher picks it for more than one round.
template <int K, bool PIVOT, bool HC, bool OWN>
__device__ __forceinline__ void step(double (&a)[WR], double* hl /* this half's LDS */, Lane& st)
{
    const int lane = st.lane;
    if (!PIVOT) __builtin_amdgcn_sched_barrier(0);          // (no branch separates the steps here: keep hipcc from interleaving them)
    // ---- 1. pivot of each half
    bool ispiv;
    unsigned long long pm;                                  // ballot of ispiv
    if (PIVOT) {
        const int khi = __double2hiint(st.nu2);
        const int mh = half32_max_i32_fused(khi);
        ispiv = khi >= mh - FILTER;
        pm = __builtin_amdgcn_ballot_w64(ispiv);
        unsigned tlo = (unsigned)pm, thi = (unsigned)(pm >> 32);
        if (__builtin_expect(((tlo & (tlo - 1u)) | (thi & (thi - 1u))) != 0u, 0)) {
            // several candidates in a half: the largest (lowest lane among exact ties: the tile is flagged then) and the check of the
            // decision -- a live column within the error margin of the chosen one sends the tile to the exact path, which owns
            // Eigen's first-maximum rule on the current positions
            asm volatile("");
            const bool live = ((st.livemask >> lane) & 1ull) != 0ull;
            bool cand = live && khi == mh;
            const unsigned klo = (unsigned)__double2loint(st.nu2);
            const unsigned ml = half32_max_u32(cand ? klo : 0u);
            cand = cand && klo == ml;
            const unsigned long long cm = __builtin_amdgcn_ballot_w64(cand);
            const unsigned clo = (unsigned)cm, chi = (unsigned)(cm >> 32);
            const int lA = clo ? __builtin_ctz(clo) : 0, lB = chi ? __builtin_ctz(chi) : 0;
            const int lbl = st.half ? lB : lA;
            ispiv = cand && st.j == lbl;
            const int src = ((st.half << 5) + lbl) << 2;
            const double best = bpermute_f64(src, st.nu2), thrb = bpermute_f64(src, st.thr);
            double margin = MREL * (st.thr + thrb);
            if (K > 0) margin += 4.547473508864641e-13 /* 2^-41 */ * __builtin_sqrt(st.a2 * (best > 0.0 ? best : 0.0));
            st.unclearm |= __builtin_amdgcn_ballot_w64(live && !ispiv && st.nu2 >= best - margin);
            pm = __builtin_amdgcn_ballot_w64(ispiv);
        }
        st.livemask &= ~pm;
    } else {
        ispiv = st.j == K;
        pm = __builtin_amdgcn_ballot_w64(ispiv);
    }
    if (K == 0 && PIVOT) {
        // the scale of the tile: the squared norm of its first pivot, to every lane of the half
        const unsigned tlo = (unsigned)pm, thi = (unsigned)(pm >> 32);
        const int lA = tlo ? __builtin_ctz(tlo) : 0, lB = thi ? __builtin_ctz(thi) : 0;
        st.a2 = bpermute_f64(((st.half << 5) + (st.half ? lB : lA)) << 2, st.nu2);
    }
    if (ispiv) {
        st.kstep = K;
        st.nu2 = __hiloint2double((int)0xBF800000, __double2loint(st.nu2));
        // ---- 2. publish the column (it is reflector K of phase 2 as well)
        double* vcol = hl + L_V + cb(K) - (K & ~1);
#pragma unroll
#if defined(QRK_P4_ABL) && (QRK_P4_ABL & 1)      // (timing ablation only: results are wrong) 1: publish 16 bytes instead of the column
        for (int i = K & ~1; i < ((QRK_P4_ABL & 1) ? (K & ~1) + 2 : WR); i += 2) *reinterpret_cast<double2*>(&vcol[i]) = make_double2(a[i], a[i + 1]);
#else
        // (8-byte stores: a single-lane ds_write_b64 takes the LDS pipe less than half of a ds_write_b128, tools/ubench8.hip; volatile
        //  keeps hipcc from merging them back)
        for (int i = K; i < WR; ++i) *(volatile lds_f64*)(&vcol[i]) = a[i];
#endif
        if (OWN) __builtin_amdgcn_sched_barrier(0);          // (the stores are issued BEFORE the arithmetic below, which they then hide)
    }
    double n2p = 0.0;                                       // (OWN) |x|^2 of the pivot column, in its lane
    if (OWN) {
        bool u = false;
        if (ispiv) {
            // the same products in the same order as the dot product of the column with itself (QRK_P4_DOT below)
            double q0 = 0.0, q1 = 0.0;
#define QRK_P4_SQ(I) if ((I) > K) { if ((I) & 1) q1 = fma(a[I], a[I], q1); else q0 = fma(a[I], a[I], q0); }
            QRK_P4_0_31(QRK_P4_SQ)
#undef QRK_P4_SQ
            const double tsq = q0 + q1, xk = a[K];
            n2p = fma(xk, xk, tsq);
            const double a2 = (K == 0 && !PIVOT) ? n2p : st.a2;
            const bool degenerate = !(tsq > DBL_MIN);
            if (K + 1 < WR) u = degenerate | (xk * xk <= X0_TINY2 * a2);
            if (PIVOT) u = u | (n2p <= PIV_TINY2 * a2);
            const double nrm = sqrt_pos(n2p);
            double nbv = __builtin_copysign(nrm, xk);        // beta = -nbv
            double s = nbv + xk;
            double ngp = recip(nbv * s);                     // -ng
            if (degenerate) { nbv = -xk; s = 0.0; ngp = 0.0; }
            *(volatile lds_f64*)(&hl[L_S + K]) = s;
            *(volatile lds_f64*)(&hl[L_NG + K]) = ngp;
            if (HC) hl[L_TAU + K] = (s * s) * ngp;
        }
        st.unclearm |= __builtin_amdgcn_ballot_w64(u);
        if (K == 0 && !PIVOT) st.a2 = bpermute_f64((st.half << 5) << 2, n2p);      // (the pivot lane of step 0 is lane 0 of its half)
    }
    __builtin_amdgcn_wave_barrier();
    // ---- 3. the lanes' elements of it, x0
    double xc[2] = {0.0, 0.0};
    double xk = 0.0;
    constexpr int M0 = (K + 1) >> 4, MK = K >> 4;
    if (K + 1 < WR) load_chunks<K>(hl, lane, xc);
    double s = 0.0, ngp = 0.0;
    if (OWN) {
        s = *(const volatile lds_f64*)(&hl[L_S + K]);
        ngp = *(const volatile lds_f64*)(&hl[L_NG + K]);
    } else {
        if (MK >= M0) xk = bcast_f64<(K & 15)>(xc[MK]);
        else xk = hl[L_V + cb(K) + (K & 1)];                  // (row K is the last one of its chunk: not among the loaded ones)
    }
    // ---- 4. d = x_tail^T a_tail of every column; the pivot lane's own is |x_tail|^2, handed to its half through LDS (the slot of
    // tau_K, which is written after it) -- no cross-lane sum
    const double ak = a[K];
    double d0 = 0.0, d1 = 0.0;
#pragma unroll
    for (int m = M0; m < 2; ++m) asm volatile("s_nop 1" : "+v"(xc[m]));      // (VALU write -> DPP read hazard, hidden from hipcc by the asm)
#define QRK_P4_DOT(I) if ((I) > K) fmac_bcast<((I) & 15)>(((I) & 1) ? d1 : d0, xc[(I) >> 4], a[I]);
    QRK_P4_0_31(QRK_P4_DOT)
#undef QRK_P4_DOT
    const double dsum = d0 + d1;
    if (!OWN) {
        double tsq = 0.0;
        if (K + 1 < WR) {
            if (ispiv) hl[L_TAU + K] = dsum;
            __builtin_amdgcn_wave_barrier();
            tsq = hl[L_TAU + K];
            __builtin_amdgcn_wave_barrier();
        }
        if (K == 0 && !PIVOT) st.a2 = fma(xk, xk, tsq);
        // (decide::unclear_reflector without short-circuit evaluation: three compares straight into wave masks, no control flow)
        unsigned long long degm = 0ull;          // lanes whose tail is empty to rounding: !(tsq > DBL_MIN)
        {
            const double n2 = fma(xk, xk, tsq);
            unsigned long long um = 0ull;
            if (K + 1 < WR) {
                degm = __builtin_amdgcn_fcmp(tsq, DBL_MIN, 13 /* ULE */);
                um = degm | __builtin_amdgcn_fcmp(xk * xk, X0_TINY2 * st.a2, 5 /* OLE */);
            } else {
                degm = ~0ull;
            }
            if (PIVOT) um |= __builtin_amdgcn_fcmp(n2, PIV_TINY2 * st.a2, 5 /* OLE */);
            st.unclearm |= um;
        }
        // ---- 5. makeHouseholder in the un-normalised form: nb = -beta = copysign(norm, x0), s = x0 - beta, ng = -1 / (beta (x0 - beta));
        // Eigen: tailSqNorm <= min() gives tau = 0, beta = x0, H = I (rare: a real branch on a wave-level test, selects inside)
        const double nrm = sqrt_pos(fma(xk, xk, tsq));
        // (Eigen's test is x0 >= 0, which takes -0.0 as positive: a zero x0 with a tail is one of unclear_reflector's cases -- the tile is
        //  redone by the exact path -- and without a tail the branch below overrides)
        double nbv = __builtin_copysign(nrm, xk);                // beta = -nbv
        double s_ = nbv + xk;
        double ngp_ = recip(nbv * s_);                             // -ng
        if (__builtin_expect(degm != 0ull, 0)) {
            asm volatile("");
            if ((degm >> lane) & 1ull) { nbv = -xk; s_ = 0.0; ngp_ = 0.0; }
        }
        if (st.j == 0) {
            hl[L_S + K] = s_; hl[L_NG + K] = ngp_;
            if (HC) hl[L_TAU + K] = (s_ * s_) * ngp_;
        }
        s = s_; ngp = ngp_;
    }
    const double ngam = fma(s, ak, dsum) * -ngp;             // -gamma of this column
    // R(K, K): in the pivot lane s x0 + |x_tail|^2 = beta (beta - x0), so its updated entry x0 - s (1 + delta) IS beta to a few ulp -- no
    // select of the beta computed from the norm (which is what Eigen stores: the fast path answers for 1e-12, flagged tiles are redone)
    const double an = fma(s, ngam, ak);
    a[K] = an;                                               // final: later steps work on the rows below
    if (!PIVOT) asm volatile("" : "+v"(a[K]));                  // (hipcc otherwise sinks the 32 selects to the store of R and keeps every beta alive)
    // ---- 6. the trailing update (columns already chosen are not masked out: nothing below the diagonal of R is ever read, and what
    // they hold stays bounded -- the reflectors are orthogonal)
#define QRK_P4_UPD(I) if ((I) > K) fmac_bcast<((I) & 15)>(a[I], xc[(I) >> 4], ngam);
    QRK_P4_0_31(QRK_P4_UPD)
#undef QRK_P4_UPD
    // ---- 7. LAWN-176 norm downdate (squared form; no clamp at zero: a negative value is <= the threshold and recomputed)
    if (PIVOT && K + 1 < WR) {
        const double nn = fma(-an, an, st.nu2);
        st.nu2 = nn;
        const unsigned long long needm = __builtin_amdgcn_ballot_w64(nn <= st.thr) & st.livemask;
        if (__builtin_expect(needm != 0ull, 0)) {
            asm volatile("");
            const bool need = ((needm >> lane) & 1ull) != 0ull;
            st.unclearm |= __builtin_amdgcn_ballot_w64(need && in_recompute_band(nn, st.thr, st.a2));      // decision (2)
            double sq = 0.0;
#define QRK_P4_SQ(I) if ((I) > K) sq = fma(a[I], a[I], sq);
            QRK_P4_0_31(QRK_P4_SQ)
#undef QRK_P4_SQ
            if (need) { st.nu2 = sq; st.thr = sq * THR_HI; }
        }
    }
}

// Q_k = H_k Q_{k+1} on the wave's columns of Q (both tiles): reflector K from the half's LDS (x_tail as published, s, ng)
template <int K>
__device__ __forceinline__ void back_step(double (&q)[WR], const double* hl, const int lane, const double (&sv)[2], const double (&ngv)[2])
{
    constexpr int M0 = (K + 1) >> 4;
    double xc[2] = {0.0, 0.0};
    if (K + 1 < WR) load_chunks<K>(hl, lane, xc);
    const double qk = q[K];
    double d0 = 0.0, d1 = 0.0;
#pragma unroll
    for (int m = M0; m < 2; ++m) asm volatile("s_nop 1" : "+v"(xc[m]));
#define QRK_P4_DOT(I) if ((I) > K) fmac_bcast<((I) & 15)>(((I) & 1) ? d1 : d0, xc[(I) >> 4], q[I]);
    QRK_P4_0_31(QRK_P4_DOT)
#undef QRK_P4_DOT
    // gamma and row K with s_K and ng_K read from lane K & 15 of the row through DPP (bitwise what reading them from LDS gives)
    double t = d0 + d1;
    fmac_bcast<(K & 15)>(t, sv[K >> 4], qk);
    const double ngam = t * -bcast_f64<(K & 15)>(ngv[K >> 4]);
    double qn = qk;
    fmac_bcast<(K & 15)>(qn, sv[K >> 4], ngam);
    q[K] = qn;
#define QRK_P4_UPD(I) if ((I) > K) fmac_bcast<((I) & 15)>(q[I], xc[(I) >> 4], ngam);
    QRK_P4_0_31(QRK_P4_UPD)
#undef QRK_P4_UPD
}

// A flagged tile again, by the wave that factorised it, in Eigen's own operation order (bdqr_exact_tile.h; bitwise what
// bdqr_exact_kernel computes).  The small tables live in the wave's LDS, the working copy and Q in the wave's global scratch.
template <bool PIVOT>
__device__ __noinline__ void redo_exact(int64_t t, double* lds, double* scratch, const double* __restrict__ tiles, double* __restrict__ q_vals,
                                        double* __restrict__ r_vals, int32_t* __restrict__ perm, double* __restrict__ hcoeffs)
{
    exact::Shared sh;
    double* rest = exact::carve_shared<64>(reinterpret_cast<unsigned char*>(lds), 32, 32, sh);
    // one of the two 8 KB arrays fits the wave's LDS next to the tables (QRK_P4_EXACT_LDS: 1 = the working copy, 2 = Q, 0 = neither).  10 000
    // tiles of +-1 (every tile redone): 2.14 / 1.64 / 1.60 ms for 0 / 1 / 2; the first-generation kernel 1.59 (profiles/r04_p4_vs_k1.txt)
    static_assert(2 * L_HALF * 8 >= 2048 + 8192, "tables + one 32 x 32 array in the wave's LDS");
    double* W = QRK_P4_EXACT_LDS == 1 ? rest : scratch;
    double* q = QRK_P4_EXACT_LDS == 2 ? rest : scratch + 1024;
    __syncthreads();
    exact::tile_qr<PIVOT, 64>(32, 32, tiles + t * 1024, W, q, sh);
    exact::tile_store<64>(32, 32, (int)(t * 32), W, q, sh, perm, hcoeffs, r_vals + t * 528, q_vals + t * 1024);
    __syncthreads();
}

}  // namespace p4

// PIVOT: ColPivHouseholderQR (else HouseholderQR).  HC: also emit the Householder coefficients.  One wave per workgroup, persistent over
// the pairs blockIdx.x, blockIdx.x + gridDim.x, ..; scratch: p4::EXACT_SCRATCH doubles per workgroup (the exact path's working copy).
template <bool PIVOT, bool HC, bool OWN>
__global__ void __launch_bounds__(64, 4)
bdqr_pair4_kernel(int64_t num_tiles, const double* __restrict__ tiles, double* __restrict__ q_vals, double* __restrict__ r_vals,
                  int32_t* __restrict__ perm, double* __restrict__ hcoeffs, double* __restrict__ scratch)
{
    using namespace p4;
    __shared__ __attribute__((aligned(16))) double lds[2 * L_HALF];
    const int64_t npairs = (num_tiles + 1) / 2;
    constexpr int CHUNK = 32;                // rounds per chunk: one 32-bit word per half remembers the flagged rounds
    const bool steady = npairs >= 3 * (int64_t)gridDim.x;
    const bool direct = npairs >= 2 * (int64_t)gridDim.x || 2 * npairs <= (int64_t)gridDim.x;     // (see the loads)
    // QRK_P4_PRIO: 1 = the waves that have one pair more than the others go first on their SIMD and in the memory queues (their two
    // chains are the critical path of a launch of 1 .. 2 rounds); 2 = the later a wave is dispatched the higher its priority
    if (QRK_P4_PRIO == 1 && (int64_t)blockIdx.x + (npairs / gridDim.x) * gridDim.x < npairs) __builtin_amdgcn_s_setprio(3);
    if (QRK_P4_PRIO == 2) {
        switch ((int)(((uint64_t)blockIdx.x * 4u) / gridDim.x)) {
            case 1: __builtin_amdgcn_s_setprio(1); break;
            case 2: __builtin_amdgcn_s_setprio(2); break;
            case 3: __builtin_amdgcn_s_setprio(3); break;
            default: break;
        }
    }
    // QRK_P4_PERM (experiment, verdict r05 item 2b: channel camping by the 16 KB wave stride?): workgroup -> pair map permuted.  > 1: the
    // pair of workgroup b is (b * QRK_P4_PERM) mod gridDim (odd multiplier, gridDim a power of two); 1: every XCD a contiguous range of pairs
#if defined(QRK_P4_PERM)
    const unsigned gd = gridDim.x;
    const int64_t bid0 = (gd & (gd - 1u)) ? (int64_t)blockIdx.x
                                          : (QRK_P4_PERM == 1 ? (int64_t)((blockIdx.x & 7u) * (gd >> 3) + (blockIdx.x >> 3))
                                                              : (int64_t)((blockIdx.x * (unsigned)QRK_P4_PERM) & (gd - 1u)));
#else
    const int64_t bid0 = blockIdx.x;
#endif
    for (int64_t pi0 = bid0; pi0 < npairs; pi0 += (int64_t)CHUNK * gridDim.x) {
    unsigned flagbits = 0u;                  // bit r: the tile of this half in round r of the chunk was flagged
    int64_t pi = pi0;
    for (int round = 0; round < CHUNK && pi < npairs; ++round, pi += gridDim.x) {
        // (per-lane values are re-derived from an opaque lane id in every round: hipcc otherwise hoists loop-invariant address
        //  arithmetic out of the loop and keeps it in registers across the factorisation)
        int lane = threadIdx.x;
        asm volatile("" : "+v"(lane));
        const int half = lane >> 5, j = lane & 31;
        double* hl = lds + half * L_HALF;
        const int64_t t = 2 * pi + half;
        const bool valid = t < num_tiles;
        Lane st;
        st.lane = lane; st.j = j; st.half = half; st.unclearm = 0ull; st.kstep = 0; st.a2 = 0.0; st.livemask = ~0ull;
        {
            // =============== phase 1: A -> R ===============
            QRK_P4_STAMP_AT(0);
            // steady state (three rounds or more per wave): the latency chain of phase 1 goes ahead of the FMA stream of phase 2 on the
            // SIMD: 100 000 tiles 621 -> 599 us; at BASELINE's 10 000 tiles (1.2 rounds) the same costs 10 % (the first waves' phase 2
            // is starved and their second pairs start late), so the launch decides.  QRK_P4_PRIO 6 / 7: always / the reverse (measured)
            if ((QRK_P4_PRIO == 0 && steady) || QRK_P4_PRIO == 6) __builtin_amdgcn_s_setprio(2);
            if (QRK_P4_PRIO == 7) __builtin_amdgcn_s_setprio(0);
            if (QRK_P4_PRIO == 8) __builtin_amdgcn_s_setprio(3);      // (experiment: a wave that is still loading / staging goes ahead of the computing ones)
            double a[WR];
            if (direct) {
                // every lane its own column, straight from the tile: 16 loads of 16 bytes, 256 bytes between lanes (plain loads: the lines
                // are re-used by the following loads of the same lane) -- no trip through LDS: 100 000 tiles -2 %, 20 000 -3.5 %, 1 250
                // -1.5 %.  A launch of one round and a bit starts with 4 096 waves loading at once, and there the 64 separate lines
                // of every load instruction cost more than the staging (+1.8 us at 10 000 tiles): such a launch stages
                typedef double d2u __attribute__((ext_vector_type(2), aligned(8)));
                const double* src = tiles + (valid ? t : 2 * pi) * 1024 + j * 32;
#pragma unroll
                for (int i = 0; i < WR; i += 2) {
                    const d2u v = *reinterpret_cast<const d2u*>(src + i);
                    a[i] = v.x; a[i + 1] = v.y;
                }
                if (!valid) {
#pragma unroll
                    for (int i = 0; i < WR; ++i) a[i] = (i == j) ? (double)(64 - j) : 0.0;
                }
            } else {
                // both tiles of the pair, one after the other through the wave's LDS: every load instruction takes 1 KB of a tile (lane l
                // rows 2 (l & 15), +1 of column 4 m + (l >> 4)), the tile is written column by column with a padded stride (16-byte
                // stores) and the lanes of its half read their columns back (16-byte loads, conflict-free: 34 doubles between lanes).
                // The loads of the second tile are in flight while the first one is staged.
                typedef double d2u __attribute__((ext_vector_type(2), aligned(8)));
                typedef double d2a __attribute__((ext_vector_type(2), aligned(16)));
                d2u ld0[8], ld1[8];
                const int64_t t0 = 2 * pi, t1 = 2 * pi + 1;
                const double* s0 = tiles + t0 * 1024 + 2 * lane;
                const double* s1 = tiles + (t1 < num_tiles ? t1 : t0) * 1024 + 2 * lane;
#pragma unroll
#if defined(QRK_P4_COPY) && (QRK_P4_COPY & 8)
                for (int m = 0; m < 8; ++m) { ld0[m] = d2u{(double)(lane + m), 1.0 + (double)pi}; ld1[m] = d2u{(double)(lane - m), 2.0 + (double)pi}; }
                (void)s0; (void)s1;
#else
                for (int m = 0; m < 8; ++m) ld0[m] = QRK_P4_LOAD(reinterpret_cast<const d2u*>(s0 + 128 * m));
#pragma unroll
                for (int m = 0; m < 8; ++m) ld1[m] = QRK_P4_LOAD(reinterpret_cast<const d2u*>(s1 + 128 * m));
#endif
                if (QRK_P4_PRIO >= 3 && round == 0 && pi0 == blockIdx.x) {
                    // (after the loads are queued in dispatch order: the later a wave's tiles arrive, the higher its priority on the SIMD;
                    //  4, 5: the waves that carry one pair more than the others -- two chains back to back -- are in the top class)
                    const bool longw = QRK_P4_PRIO >= 4 && (int64_t)blockIdx.x + (npairs / gridDim.x) * gridDim.x < npairs;
                    switch (longw ? 3 : (int)(((uint64_t)blockIdx.x * 4u) / gridDim.x)) {
                        case 1: __builtin_amdgcn_s_setprio(1); break;
                        case 2: __builtin_amdgcn_s_setprio(2); break;
                        case 3: __builtin_amdgcn_s_setprio(3); break;
                        default: break;
                    }
                }
                double* sw = lds + (lane >> 4) * STAGE_LD + 2 * (lane & 15);
#pragma unroll
                for (int m = 0; m < 8; ++m) *reinterpret_cast<d2a*>(sw + 4 * m * STAGE_LD) = d2a{ld0[m].x, ld0[m].y};
                __builtin_amdgcn_wave_barrier();
                // (both halves read: a conditional first definition would leave a[] undefined on one side, which hipcc carries
                //  around the round loop as live values)
#pragma unroll
                for (int i = 0; i < WR; i += 2) {
                    const d2a v = *reinterpret_cast<const d2a*>(lds + j * STAGE_LD + i);
                    a[i] = v.x; a[i + 1] = v.y;
                }
                __builtin_amdgcn_wave_barrier();
#pragma unroll
                for (int m = 0; m < 8; ++m) *reinterpret_cast<d2a*>(sw + 4 * m * STAGE_LD) = d2a{ld1[m].x, ld1[m].y};
                __builtin_amdgcn_wave_barrier();
                if (half == 1) {
                    // (the missing partner of an odd last tile: diag(64..33) -- distinct norms, no tie-breaking; nothing of it is stored)
#pragma unroll
                    for (int i = 0; i < WR; i += 2) {
                        const d2a v = *reinterpret_cast<const d2a*>(lds + j * STAGE_LD + i);
                        a[i] = valid ? v.x : ((i == j) ? (double)(64 - j) : 0.0);
                        a[i + 1] = valid ? v.y : ((i + 1 == j) ? (double)(64 - j) : 0.0);
                    }
                }
                __builtin_amdgcn_wave_barrier();
            }
            QRK_P4_STAMP_AT(1);
            if (QRK_P4_PRIO == 8) __builtin_amdgcn_s_setprio(0);
            {
                double s0 = 0.0, s1 = 0.0;
#pragma unroll
                for (int i = 0; i < WR; i += 2) { s0 = fma(a[i], a[i], s0); s1 = fma(a[i + 1], a[i + 1], s1); }
                st.nu2 = s0 + s1;
                st.thr = st.nu2 * THR_HI;
            }
#if defined(QRK_P4_COPY)                        // (diagnostic: the launch's memory traffic without its arithmetic -- tools/p4_stamps.py, profiles/r06_k1_ramp.txt)
#define QRK_P4_STEP(K) st.kstep = st.j;
#elif defined(QRK_P4_ABL) && (QRK_P4_ABL & 2)     // (timing ablation: every second pair of a wave skips steps 16..31 and back steps 31..16)
#define QRK_P4_STEP(K) if ((K) < 16 || !(round & 1)) step<K, PIVOT, HC, OWN>(a, hl, st);
#else
#define QRK_P4_STEP(K) step<K, PIVOT, HC, OWN>(a, hl, st);
#endif
            QRK_P4_0_31(QRK_P4_STEP)
#undef QRK_P4_STEP
            // ---- R: lane j holds column p = kstep of R in rows 0 .. p; the packed CSC value order of m_R (BlockDiagonalSparseQR.h:475-479)
            // puts entry (i, p) at p (p + 1) / 2 + i -- a contiguous run per lane, stored straight from the registers, two rows at a time;
            // the permutation splice (:519-521): the column chosen at step p ends at position p
            int ln = threadIdx.x;
            asm volatile("" : "+v"(ln));
#if (defined(QRK_P4_COPY) && (QRK_P4_COPY & 2)) || (defined(QRK_P4_ABL) && (QRK_P4_ABL & 4))      // (timing only: without the stores of R)
            if (valid && st.nu2 == 12345.678) {
#else
            if (valid) {
#endif
                const int p = st.kstep, jj = ln & 31;
                const int cbase = (int)(t * 32);
                perm[cbase + p] = cbase + jj;
                double* dst = r_vals + t * 528 + ((p * (p + 1)) >> 1);
                typedef double d2u __attribute__((ext_vector_type(2), aligned(8)));
#pragma unroll
                for (int i = 0; i < WR; i += 2) {
                    if (i + 1 <= p) QRK_P4_STORE((d2u{a[i], a[i + 1]}), reinterpret_cast<d2u*>(dst + i));
                    else if (i == p) QRK_P4_STORE(a[i], dst + i);
                }
                if (HC && hcoeffs) hcoeffs[cbase + jj] = lds[(ln >> 5) * L_HALF + L_TAU + jj];
            }
        }
        // a decision inside its error margin, anywhere in the half: the tile is redone by the exact path after the rounds
        {
#ifdef QRK_P4_ABL
            const unsigned long long um = 0ull;
#else
            const unsigned long long um = st.unclearm;
#endif
            const bool f = half ? (um >> 32) != 0ull : (um & 0xffffffffull) != 0ull;
            if (f && valid) flagbits |= 1u << round;
        }
        {
            // =============== phase 2: Q = H_0 ... H_31, backward ===============
            QRK_P4_STAMP_AT(2);
            if ((QRK_P4_PRIO == 0 && steady) || QRK_P4_PRIO == 6) __builtin_amdgcn_s_setprio(0);
            if (QRK_P4_PRIO == 7) __builtin_amdgcn_s_setprio(2);
            int ln = threadIdx.x;
            asm volatile("" : "+v"(ln));
            const int jj = ln & 31;
            const double* hl2 = lds + (ln >> 5) * L_HALF;
            double q[WR];
#pragma unroll
            for (int i = 0; i < WR; ++i) q[i] = (i == jj) ? 1.0 : 0.0;
            double sv[2], ngv[2];
            // (s_K and ng_K of the 32 reflectors: lane l keeps entries l & 15 and 16 + (l & 15); back_step reads them through DPP)
#pragma unroll
            for (int m = 0; m < 2; ++m) { sv[m] = hl2[L_S + 16 * m + (ln & 15)]; ngv[m] = hl2[L_NG + 16 * m + (ln & 15)]; }
#if defined(QRK_P4_COPY)
#define QRK_P4_BACK(K) q[K] += sv[K >> 4];
#elif defined(QRK_P4_ABL) && (QRK_P4_ABL & 2)
#define QRK_P4_BACK(K) if ((K) < 16 || !(round & 1)) back_step<K>(q, hl2, ln, sv, ngv);
#else
#define QRK_P4_BACK(K) back_step<K>(q, hl2, ln, sv, ngv);
#endif
            QRK_P4_31_0(QRK_P4_BACK)
#undef QRK_P4_BACK
            // row-major rows of Q_i are the CSR value order of m_Q in both FullQ ([U|N] split, :455-471) and BlockDiagonalQ (:480-492)
            // layouts: lane j holds COLUMN j of Q_i, one coalesced store of 256 bytes per row and half
#if (defined(QRK_P4_COPY) && (QRK_P4_COPY & 4)) || (defined(QRK_P4_ABL) && (QRK_P4_ABL & 8))      // (timing only: without the stores of Q)
            if (valid && q[5] == 12345.678) {
#else
            if (valid) {
#endif
                double* dst = q_vals + t * 1024 + jj;
#pragma unroll
                for (int i = 0; i < WR; ++i) QRK_P4_STOREQ(q[i], dst + 32 * i);
            }
            QRK_P4_STAMP_AT(3);
        }
        __builtin_amdgcn_wave_barrier();
        if (QRK_P4_PRIO == 2 || QRK_P4_PRIO == 3) __builtin_amdgcn_s_setprio(3);     // (a further pair of this wave starts later than anything else)
        if (QRK_P4_PRIO == 5 && !((int64_t)blockIdx.x + (npairs / gridDim.x) * gridDim.x < npairs)) __builtin_amdgcn_s_setprio(0);
    }
    // ---- the flagged tiles, again, with the reference's own operation order (rare: generic data never gets here)
    {
        const unsigned f0 = (unsigned)__builtin_amdgcn_readlane((int)flagbits, 0), f1 = (unsigned)__builtin_amdgcn_readlane((int)flagbits, 32);
        if (__builtin_expect((f0 | f1) != 0u, 0)) {
            double* sc = scratch + (int64_t)blockIdx.x * EXACT_SCRATCH;
            for (int h2 = 0; h2 < 2; ++h2) {
                unsigned m = h2 ? f1 : f0;
                while (m) {
                    const int rnd = __builtin_ctz(m);
                    m &= m - 1;
                    redo_exact<PIVOT>(2 * (pi0 + (int64_t)rnd * gridDim.x) + h2, lds, sc, tiles, q_vals, r_vals, perm,
                                      HC ? hcoeffs : nullptr);
                }
            }
        }
    }
    }
}

// step<.., OWN> (the pivot lane's own norm) for launches of more than one round of the resident waves, see step; QRK_P4_OWN=0 / 1 forces
// the choice (diagnostic)
bool bdqr_pair4_own_norm(int64_t num_tiles, int num_wg)
{
    if (const char* e = std::getenv("QRK_P4_OWN")) return std::atoi(e) != 0;
    return (num_tiles + 1) / 2 > (int64_t)num_wg;
}

int64_t bdqr_pair4_scratch_doubles(int num_wg) { return (int64_t)num_wg * p4::EXACT_SCRATCH; }

// Uniform 32 x 32 batches (any alignment).  num_wg: resident wave slots (16 per CU).
hipError_t launch_bdqr_pair4(int64_t num_tiles, int pivoting, const double* tiles, double* q_vals, double* r_vals, int32_t* perm,
                             double* hcoeffs, double* scratch, int num_wg, hipStream_t stream)
{
    if (num_tiles <= 0) return hipSuccess;
    const int64_t npairs = (num_tiles + 1) / 2;
    const int64_t nwg = npairs < num_wg ? npairs : num_wg;
    const dim3 grid((unsigned)nwg), block(64);
    const bool own = bdqr_pair4_own_norm(num_tiles, num_wg);
#define QRK_P4_LAUNCH(P, H, L) hipLaunchKernelGGL((bdqr_pair4_kernel<P, H, L>), grid, block, 0, stream, num_tiles, tiles, q_vals, r_vals, perm, hcoeffs, scratch)
#define QRK_P4_LAUNCH2(P, H) do { if (own) QRK_P4_LAUNCH(P, H, true); else QRK_P4_LAUNCH(P, H, false); } while (0)
#ifdef QRK_P4_STAMP
    if (pivoting) QRK_P4_LAUNCH2(true, false); else QRK_P4_LAUNCH2(false, false);
#else
    if (pivoting) { if (hcoeffs) QRK_P4_LAUNCH2(true, true); else QRK_P4_LAUNCH2(true, false); }
    else { if (hcoeffs) QRK_P4_LAUNCH2(false, true); else QRK_P4_LAUNCH2(false, false); }
#endif
#undef QRK_P4_LAUNCH2
#undef QRK_P4_LAUNCH
    return hipGetLastError();
}

}  // namespace qrk
