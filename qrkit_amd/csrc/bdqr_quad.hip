// bdqr_quad.hip -- uniform batches of tiles with 5 .. 16 rows (cols <= rows), FOUR tiles per wavefront (9 .. 16 rows) or EIGHT (5 .. 8
// rows): A_i P_i = Q_i R_i with explicit Q_i, for gfx950.  The design of bdqr_pair4.hip (round 4 / 5) at 16 and at 8 rows; round 5.
//
// Same reference seam as the other block-diagonal kernels: the body of the hot loop of QRKit::BlockDiagonalSparseQR::factorize
// (src/QRKit/BlockDiagonalSparseQR.h:432-526): blockSolver.compute(block) (:437-438, Eigen ColPivHouseholderQR / HouseholderQR),
// Qi = blockSolver.matrixQ() (:446), the Q / R value assembly (:455-500) and the column-permutation splice (:519-521).
//
// Why.  bdqr_small.hip gives a tile of up to 16 rows a group of 16 lanes that carries A and Q^T TOGETHER (64 data registers) and moves
// the pivot column through ds_bpermute (2 (16 - k) LDS-pipe instructions per step): 16 x 16 ran at 0.24 of the HBM roofline with 167
// registers, three waves per SIMD.  Here:
//   * lane 16 g + j owns column j of tile g in 16 row registers -- a tile is exactly one DPP row;
//   * the pivot lane of each tile publishes its column to LDS (8-byte stores of four lanes at once), every lane takes element
//     lane & 15 and the dot / update FMAs read it through the DPP row_newbcast operand -- no register of the broadcast, no bpermute;
//   * two phases in the same registers: A -> R, then Q = H_0 ... H_{c-1} by backward accumulation from the reflectors in LDS;
//   * 32 data registers, 1.5 KB of LDS per tile: more waves per SIMD than the LDS pipe can use.
// Tiles of 5 .. 8 rows (WR = 8): lane 8 g + j owns column j of tile g in 8 row registers, TWO tiles share a DPP row; every FMA with a
// row_newbcast operand becomes two, told apart by the bank mask (lanes 0..7 read lane N of the row, lanes 8..15 lane 8 + N); the maximum
// over a tile stops at the half-row mirror.  See the hazard note at fmac_bcast.
// Tiles with fewer than 16 (8) rows are zero-padded below (zero rows change neither the reflectors nor any sum); the steps run to the
// launch's number of columns.  Decisions exactly as bdqr_pair.hip ("Decisions and the exact path"): integer arg-max on the high
// words with a filter, margins, LAWN-176 band, degenerate reflector, sign of beta, noise-level pivot; a flagged tile is redone by the
// wave itself after its rounds, in Eigen's own operation order (bdqr_exact_tile.h, as bdqr_pair4.hip does) -- round 6: no redo list and no
// second launch behind this one, which at BASELINE configs[3]'s 20 000 tiles of 8 x 6 was a third of the time of factorize().
#include "qrk_device.h"
#include "bdqr_exact_tile.h"

#include <float.h>

namespace qrk {

namespace q16 {

using namespace decide;

constexpr int FILTER = 256;              // pivot candidates: high word of the squared norm within 2^-12 (relative) of the largest
// WR: row registers per lane = lanes per tile (16: tiles of 9 .. 16 rows, four per wavefront; 8: tiles of 5 .. 8 rows, eight per
// wavefront).  LDS per TILE (doubles): reflector K (the pivot column of step K, as published) holds rows K .. WR - 1 at cb(K)
template <int WR>
struct Lay {
    static constexpr int cb(int k) { int s = 0; for (int q = 0; q < k; ++q) s += WR - q; return s; }
    static constexpr int L_V = 0;
    static constexpr int L_S = cb(WR);           // [WR] s = x0 - beta
    static constexpr int L_NG = L_S + WR;        // [WR] 1 / (beta (beta - x0))
    static constexpr int L_TAU = L_NG + WR;      // [WR] tau; before that, the hand-off word of |x_tail|^2 of the step
    static constexpr int L_TILE = L_TAU + WR;    // WR = 16: 184 doubles = 1 472 B per tile, 5 888 B per wave
};
static_assert(Lay<16>::cb(16) == 136 && Lay<8>::L_TILE == 60, "the triangle of the reflectors");

typedef __attribute__((address_space(3))) double lds_f64;      // (volatile accesses through a generic pointer would become flat_*)

#define QRK_Q16_0_15(M) M(0) M(1) M(2) M(3) M(4) M(5) M(6) M(7) M(8) M(9) M(10) M(11) M(12) M(13) M(14) M(15)
#define QRK_Q16_15_0(M) M(15) M(14) M(13) M(12) M(11) M(10) M(9) M(8) M(7) M(6) M(5) M(4) M(3) M(2) M(1) M(0)

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
// d += X[N] * c with X[N] = element N of the lane's TILE, through the DPP row_newbcast operand.  WR = 16: a tile is a DPP row of 16 lanes.
// WR = 8: two tiles share a row -- lanes 0..7 read lane N, lanes 8..15 lane 8 + N, two instructions told apart by the bank mask (a bank
// is four lanes; a lane outside the mask keeps its value).
// HAZARD (gfx950, tools/ubench_dpp64_bankmask.hip, profiles/r05_quad.txt): two DPP operations with DIFFERENT bank masks that write the
// same register back to back lose the first one's lanes (the second takes its accumulator from the forwarding path, which holds nothing
// valid for the lanes the first one masked off).  One instruction of any kind between the two is enough; the same mask twice in a row,
// or an ordinary VALU write before a masked operation, is fine.  Hence the s_nop 0 here and the two-row form below, where the second
// row's instruction is the one in between.
template <int WR, int N>
__device__ __forceinline__ void fmac_bcast(double& d, double X, double c)
{
    if (WR == 16) {
        asm("v_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(d) : "v"(X), "v"(c), "n"(N));
    } else {
        asm("v_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0x3\n\t"
            "s_nop 0\n\t"
            "v_fmac_f64_dpp %0, %1, %2 row_newbcast:%4 row_mask:0xf bank_mask:0xc" : "+v"(d) : "v"(X), "v"(c), "n"(N & 7), "n"((N & 7) + 8));
    }
}
// (WR = 8) dN += X[N] * cN and dM += X[M] * cM: low halves of both, then high halves of both -- no wait state
template <int N, int M>
__device__ __forceinline__ void fmac_bcast2_oct(double& dN, double& dM, double X, double cN, double cM)
{
    asm("v_fmac_f64_dpp %0, %2, %3 row_newbcast:%5 row_mask:0xf bank_mask:0x3\n\t"
        "v_fmac_f64_dpp %1, %2, %4 row_newbcast:%6 row_mask:0xf bank_mask:0x3\n\t"
        "v_fmac_f64_dpp %0, %2, %3 row_newbcast:%7 row_mask:0xf bank_mask:0xc\n\t"
        "v_fmac_f64_dpp %1, %2, %4 row_newbcast:%8 row_mask:0xf bank_mask:0xc"
        : "+v"(dN), "+v"(dM) : "v"(X), "v"(cN), "v"(cM), "n"(N & 7), "n"(M & 7), "n"((N & 7) + 8), "n"((M & 7) + 8));
}
// (WR = 8) the rows I..7 of a dot product (accumulators by the parity of the row, as in the 16-lane form) and of an update, two rows at
// a time
template <int I>
__device__ __forceinline__ void dot_rows_oct(double& d0, double& d1, double xc, const double (&a)[8])
{
    if constexpr (I + 1 < 8) {
        if constexpr ((I & 1) != 0) fmac_bcast2_oct<I, I + 1>(d1, d0, xc, a[I], a[I + 1]);
        else fmac_bcast2_oct<I, I + 1>(d0, d1, xc, a[I], a[I + 1]);
        dot_rows_oct<I + 2>(d0, d1, xc, a);
    } else if constexpr (I < 8) {
        fmac_bcast<8, I>((I & 1) ? d1 : d0, xc, a[I]);
    }
}
template <int I>
__device__ __forceinline__ void upd_rows_oct(double (&a)[8], double xc, double ngam)
{
    if constexpr (I + 1 < 8) {
        fmac_bcast2_oct<I, I + 1>(a[I], a[I + 1], xc, ngam, ngam);
        upd_rows_oct<I + 2>(a, xc, ngam);
    } else if constexpr (I < 8) {
        fmac_bcast<8, I>(a[I], xc, ngam);
    }
}
template <int WR, int N>
__device__ __forceinline__ double bcast_f64(double X)
{
    double r;
    if (WR == 16) {
        asm("s_nop 1\n\tv_mov_b64_dpp %0, %1 row_newbcast:%2 row_mask:0xf bank_mask:0xf" : "=v"(r) : "v"(X), "n"(N));
    } else {
        asm("s_nop 1\n\tv_mov_b64_dpp %0, %1 row_newbcast:%2 row_mask:0xf bank_mask:0x3\n\t"
            "s_nop 0\n\t"
            "v_mov_b64_dpp %0, %1 row_newbcast:%3 row_mask:0xf bank_mask:0xc" : "=&v"(r) : "v"(X), "n"(N & 7), "n"((N & 7) + 8));
    }
    return r;
}
__device__ __forceinline__ double bpermute_f64(int byte_addr, double v)
{
    const int lo = __builtin_amdgcn_ds_bpermute(byte_addr, __double2loint(v));
    const int hi = __builtin_amdgcn_ds_bpermute(byte_addr, __double2hiint(v));
    return __hiloint2double(hi, lo);
}
// max over the WR lanes of every tile, in every lane of the tile: fused DPP stages (quad xor 1, quad xor 2, half-row mirror = 8 lanes;
// row mirror = 16)
template <int WR>
__device__ __forceinline__ int tile_max_i32_fused(int v)
{
    int m;
    if (WR == 16)
        asm("s_nop 1\n\t"
            "v_max_i32_dpp %0, %1, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
            "s_nop 1\n\t"
            "v_max_i32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
            "s_nop 1\n\t"
            "v_max_i32_dpp %0, %0, %0 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
            "s_nop 1\n\t"
            "v_max_i32_dpp %0, %0, %0 row_mirror row_mask:0xf bank_mask:0xf"
            : "=&v"(m) : "v"(v));
    else
        asm("s_nop 1\n\t"
            "v_max_i32_dpp %0, %1, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
            "s_nop 1\n\t"
            "v_max_i32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
            "s_nop 1\n\t"
            "v_max_i32_dpp %0, %0, %0 row_half_mirror row_mask:0xf bank_mask:0xf"
            : "=&v"(m) : "v"(v));
    return m;
}
template <int WR>
__device__ __forceinline__ unsigned tile_max_u32(unsigned v)
{
    v = max(v, (unsigned)dpp_i32<0xB1>((int)v));
    v = max(v, (unsigned)dpp_i32<0x4E>((int)v));
    v = max(v, (unsigned)dpp_i32<0x141>((int)v));
    if (WR == 16) v = max(v, (unsigned)dpp_i32<0x140>((int)v));
    return v;
}

struct Lane {
    int lane, j, g;
    unsigned long long livemask;   // (wave-uniform) the lanes whose column of A is not yet chosen
    unsigned long long unclearm;   // (wave-uniform) lanes that saw a decision of their tile inside its error margin
    int kstep;        // position at which this lane's column was chosen
    double nu2;       // m_colNormsUpdated^2 (a chosen or non-existent column carries a negative value)
    double thr;       // sqrt(eps) (1 + 2^-12) m_colNormsDirect^2
    double a2;        // |A|^2 of this lane's tile: squared norm of its first pivot column (scale of the decision margins)
};

// One step of ColPivHouseholderQR::computeInPlace / HouseholderQR on the four tiles of the wave (bdqr_pair4.hip's step; see
// bdqr_pair.hip for the arithmetic: squared norms, un-normalised reflector, decisions).  r: rows of the tiles (the launch is uniform).
template <int WR, int K, bool PIVOT, bool HC>
__device__ __forceinline__ void step(double (&a)[WR], double* tl /* this tile's LDS */, Lane& st, const int r)
{
    typedef Lay<WR> L;
    constexpr int L_V = L::L_V, L_S = L::L_S, L_NG = L::L_NG, L_TAU = L::L_TAU;
    const int lane = st.lane;
    // ---- 1. pivot of each tile
    bool ispiv;
    unsigned long long pm;                                  // ballot of ispiv: one lane per tile
    if (PIVOT) {
        const int khi = __double2hiint(st.nu2);
        const int mh = tile_max_i32_fused<WR>(khi);
        ispiv = khi >= mh - FILTER;
        pm = __builtin_amdgcn_ballot_w64(ispiv);
        // (every tile has at least one candidate, so subtracting one from each WR-bit field never borrows across fields)
        constexpr unsigned long long ONES = WR == 16 ? 0x0001000100010001ull : 0x0101010101010101ull;
        if (__builtin_expect((pm & (pm - ONES)) != 0ull, 0)) {
            // several candidates in a tile: the largest (lowest lane among exact ties: the tile is flagged then) and the check of the
            // decision -- a live column within the error margin of the chosen one sends the tile to the exact path, which owns
            // Eigen's first-maximum rule on the current positions
            asm volatile("");
            const bool live = ((st.livemask >> lane) & 1ull) != 0ull;
            bool cand = live && khi == mh;
            const unsigned klo = (unsigned)__double2loint(st.nu2);
            const unsigned ml = tile_max_u32<WR>(cand ? klo : 0u);
            cand = cand && klo == ml;
            const unsigned long long cm = __builtin_amdgcn_ballot_w64(cand);
            const unsigned f = (unsigned)(cm >> (WR * st.g)) & ((1u << WR) - 1u);
            const int lbl = f ? __builtin_ctz(f) : 0;
            ispiv = cand && st.j == lbl;
            const int src = (st.g * WR + lbl) << 2;
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
        // the scale of the tile: the squared norm of its first pivot, to every lane of the tile
        const unsigned f = (unsigned)(pm >> (WR * st.g)) & ((1u << WR) - 1u);
        st.a2 = bpermute_f64((st.g * WR + (f ? __builtin_ctz(f) : 0)) << 2, st.nu2);
    }
    if (ispiv) {
        st.kstep = K;
        st.nu2 = __hiloint2double((int)0xBF800000, __double2loint(st.nu2));
        // ---- 2. publish the column (it is reflector K of phase 2 as well): 8-byte stores (tools/ubench8.hip)
        double* vcol = tl + L_V + L::cb(K) - K;
#pragma unroll
        for (int i = K; i < WR; ++i) *(volatile lds_f64*)(&vcol[i]) = a[i];
    }
    __builtin_amdgcn_wave_barrier();
    // ---- 3. the lanes' elements of it (element lane & 15; the elements above row K are not data and are never used), x0
    double xc = *(const volatile lds_f64*)(tl + L_V + L::cb(K) - K + (lane & (WR - 1)));
    const double xk = bcast_f64<WR, K>(xc);
    // ---- 4. d = x_tail^T a_tail of every column; the pivot lane's own is |x_tail|^2, handed to its tile through LDS (the slot of
    // tau_K, which is written after it) -- no cross-lane sum
    const double ak = a[K];
    double d0 = 0.0, d1 = 0.0;
    asm volatile("s_nop 1" : "+v"(xc));                      // (VALU write -> DPP read hazard, hidden from hipcc by the asm)
    if constexpr (WR == 8) {
        dot_rows_oct<K + 1>(d0, d1, xc, a);
    } else {
#define QRK_Q16_DOT(I) if ((I) > K && (I) < WR) fmac_bcast<WR, (I)>(((I) & 1) ? d1 : d0, xc, a[(I) < WR ? (I) : 0]);
        QRK_Q16_0_15(QRK_Q16_DOT)
#undef QRK_Q16_DOT
    }
    const double dsum = d0 + d1;
    double tsq = 0.0;
    if (K + 1 < WR) {
        if (ispiv) tl[L_TAU + K] = dsum;
        __builtin_amdgcn_wave_barrier();
        tsq = tl[L_TAU + K];
        __builtin_amdgcn_wave_barrier();
    }
    if (K == 0 && !PIVOT) st.a2 = fma(xk, xk, tsq);
    // (decide::unclear_reflector without short-circuit evaluation: compares straight into wave masks, no control flow; a tail exists
    //  when row K is not the tile's last)
    unsigned long long degm;                 // lanes whose tail is empty to rounding: !(tsq > DBL_MIN)
    {
        const double n2 = fma(xk, xk, tsq);
        unsigned long long um = 0ull;
        if (K + 1 < WR) {
            degm = __builtin_amdgcn_fcmp(tsq, DBL_MIN, 13 /* ULE */);
            if (K + 1 < r) um = degm | __builtin_amdgcn_fcmp(xk * xk, X0_TINY2 * st.a2, 5 /* OLE */);
        } else {
            degm = ~0ull;
        }
        if (PIVOT) um |= __builtin_amdgcn_fcmp(n2, PIV_TINY2 * st.a2, 5 /* OLE */);
        st.unclearm |= um;
    }
    // ---- 5. makeHouseholder in the un-normalised form: nb = -beta = copysign(norm, x0), s = x0 - beta, ng = -1 / (beta (x0 - beta));
    // Eigen: tailSqNorm <= min() gives tau = 0, beta = x0, H = I (rare: a real branch on a wave-level test, selects inside)
    const double nrm = sqrt_pos(fma(xk, xk, tsq));
    double nbv = __builtin_copysign(nrm, xk);                // beta = -nbv
    double s = nbv + xk;
    double ngp = recip(nbv * s);                             // -ng
    if (__builtin_expect(degm != 0ull, 0)) {
        asm volatile("");
        if ((degm >> lane) & 1ull) { nbv = -xk; s = 0.0; ngp = 0.0; }
    }
    if (st.j == 0) {
        tl[L_S + K] = s; tl[L_NG + K] = ngp;
        if (HC) tl[L_TAU + K] = (s * s) * ngp;
    }
    const double ngam = fma(s, ak, dsum) * -ngp;             // -gamma of this column
    // R(K, K): in the pivot lane s x0 + |x_tail|^2 = beta (beta - x0), so its updated entry IS beta to a few ulp (bdqr_pair4.hip)
    const double an = fma(s, ngam, ak);
    a[K] = an;                                               // final: later steps work on the rows below
    if (!PIVOT) asm volatile("" : "+v"(a[K]));
    // ---- 6. the trailing update (columns already chosen are not masked out: nothing below the diagonal of R is ever read)
    if constexpr (WR == 8) {
        upd_rows_oct<K + 1>(a, xc, ngam);
    } else {
#define QRK_Q16_UPD(I) if ((I) > K && (I) < WR) fmac_bcast<WR, (I)>(a[(I) < WR ? (I) : 0], xc, ngam);
        QRK_Q16_0_15(QRK_Q16_UPD)
#undef QRK_Q16_UPD
    }
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
#define QRK_Q16_SQ(I) if ((I) > K && (I) < WR) sq = fma(a[(I) < WR ? (I) : 0], a[(I) < WR ? (I) : 0], sq);
            QRK_Q16_0_15(QRK_Q16_SQ)
#undef QRK_Q16_SQ
            if (need) { st.nu2 = sq; st.thr = sq * THR_HI; }
        }
    }
}

// Q_k = H_k Q_{k+1} on the wave's columns of Q (four tiles): reflector K from the tile's LDS (x_tail as published); s_K and ng_K come
// from lane K of the tile through DPP (lane l holds entry l & 15 of both)
template <int WR, int K>
__device__ __forceinline__ void back_step(double (&q)[WR], const double* tl, const int lane, const double sv, const double ngv)
{
    typedef Lay<WR> L;
    double xc = 0.0;
    if (K + 1 < WR) xc = *(const volatile lds_f64*)(tl + L::L_V + L::cb(K) - K + (lane & (WR - 1)));
    const double qk = q[K];
    double d0 = 0.0, d1 = 0.0;
    asm volatile("s_nop 1" : "+v"(xc));
    if constexpr (WR == 8) {
        dot_rows_oct<K + 1>(d0, d1, xc, q);
    } else {
#define QRK_Q16_DOT(I) if ((I) > K && (I) < WR) fmac_bcast<WR, (I)>(((I) & 1) ? d1 : d0, xc, q[(I) < WR ? (I) : 0]);
        QRK_Q16_0_15(QRK_Q16_DOT)
#undef QRK_Q16_DOT
    }
    double t = d0 + d1;
    fmac_bcast<WR, K>(t, sv, qk);
    const double ngam = t * -bcast_f64<WR, K>(ngv);
    double qn = qk;
    fmac_bcast<WR, K>(qn, sv, ngam);
    q[K] = qn;
    if constexpr (WR == 8) {
        upd_rows_oct<K + 1>(q, xc, ngam);
    } else {
#define QRK_Q16_UPD(I) if ((I) > K && (I) < WR) fmac_bcast<WR, (I)>(q[(I) < WR ? (I) : 0], xc, ngam);
        QRK_Q16_0_15(QRK_Q16_UPD)
#undef QRK_Q16_UPD
    }
}

// A flagged tile again, by the wave that factorised it, in Eigen's own operation order (bitwise what bdqr_exact_kernel computes): the
// tables, the working copy and Q in the wave's LDS
template <int WR, bool PIVOT>
__device__ __noinline__ void redo_exact(int64_t t, int r, int c, double* lds, const double* __restrict__ tiles, double* __restrict__ q_vals,
                                        double* __restrict__ r_vals, int32_t* __restrict__ perm, double* __restrict__ hcoeffs)
{
    static_assert((64 / WR) * Lay<WR>::L_TILE * 8 >= (4 * WR + 64) * 8 + (2 * WR + 64) * 4 + 16 + 2 * WR * WR * 8, "tables + W + Q in the wave's LDS");
    exact::Shared sh;
    double* W = exact::carve_shared<64>(reinterpret_cast<unsigned char*>(lds), WR, WR, sh);
    double* q = W + r * c;
    const int nr = (c * (c + 1)) >> 1;
    __syncthreads();
    exact::tile_qr<PIVOT, 64>(r, c, tiles + t * (int64_t)(r * c), W, q, sh);
    exact::tile_store<64>(r, c, (int)(t * c), W, q, sh, perm, hcoeffs, r_vals + t * nr, q_vals + t * (int64_t)(r * r));
    __syncthreads();
}

// the flagged tiles of a chunk of rounds: bit rnd of flagbits in the lanes of tile slot g2
template <int WR, bool PIVOT>
__device__ __noinline__ void redo_flagged(unsigned flagbits, int64_t qi0, int r, int c, double* lds, const double* __restrict__ tiles,
                                          double* __restrict__ q_vals, double* __restrict__ r_vals, int32_t* __restrict__ perm,
                                          double* __restrict__ hcoeffs)
{
    constexpr int TPW = 64 / WR;
    for (int g2 = 0; g2 < TPW; ++g2) {
        unsigned m = (unsigned)__builtin_amdgcn_readlane((int)flagbits, g2 * WR);
        while (m) {
            const int rnd = __builtin_ctz(m);
            m &= m - 1;
            redo_exact<WR, PIVOT>((int64_t)TPW * (qi0 + (int64_t)rnd * gridDim.x) + g2, r, c, lds, tiles, q_vals, r_vals, perm, hcoeffs);
        }
    }
}

}  // namespace q16

#ifndef QRK_QUAD_WAVES
#define QRK_QUAD_WAVES 5       // waves per SIMD the 16-row instantiation is compiled for (96 VGPRs, no spill; 6: 80 VGPRs with 3-5 spills in the staging)
#endif
#ifndef QRK_QUAD_WAVES8
#define QRK_QUAD_WAVES8 8      // ... and the 8-row instantiation
#endif

// PIVOT: ColPivHouseholderQR (else HouseholderQR).  HC: also emit the Householder coefficients.  One wave per workgroup, persistent over
// the quads blockIdx.x, blockIdx.x + gridDim.x, ..; r x c: the size of every tile (9 <= r <= 16, c <= r).
template <int WR, bool PIVOT, bool HC>
__global__ void __launch_bounds__(64, (WR == 16 ? QRK_QUAD_WAVES : QRK_QUAD_WAVES8))
bdqr_quad_kernel(int64_t num_tiles, int r, int c, const double* __restrict__ tiles, double* __restrict__ q_vals, double* __restrict__ r_vals,
                 int32_t* __restrict__ perm, double* __restrict__ hcoeffs, int32_t* __restrict__ redo_count, int32_t* __restrict__ redo_ids)
{
    using namespace q16;
    typedef Lay<WR> L;
    constexpr int L_TILE = L::L_TILE, L_S = L::L_S, L_NG = L::L_NG, L_TAU = L::L_TAU;
    constexpr int TPW = 64 / WR, HT = TPW / 2;                        // tiles per wave (4 or 8); tiles per staging pass
    constexpr int LG = WR == 16 ? 4 : 3;                              // log2 of the lanes of a tile
    constexpr int MLD = (HT * WR * WR + 63) / 64;                     // load instructions per staging pass
    __shared__ __attribute__((aligned(16))) double lds[TPW * L_TILE];
    const int64_t nquads = (num_tiles + TPW - 1) / TPW;
    const int rc = r * c, rr = r * r, nr = (c * (c + 1)) >> 1;
    constexpr int CHUNK = 32;                // rounds per chunk: one 32-bit word per tile remembers the flagged rounds
    for (int64_t qi0 = blockIdx.x; qi0 < nquads; qi0 += (int64_t)CHUNK * gridDim.x) {
    unsigned flagbits = 0u;                  // bit r: the tile of this group of lanes in round r of the chunk was flagged
    int64_t qi = qi0;
    for (int round = 0; round < CHUNK && qi < nquads; ++round, qi += gridDim.x) {
        // (per-lane values are re-derived from an opaque lane id in every round: hipcc otherwise hoists loop-invariant address
        //  arithmetic out of the loop and keeps it in registers across the factorisation)
        int lane = threadIdx.x;
        asm volatile("" : "+v"(lane));
        const int g = lane >> LG, j = lane & (WR - 1);
        double* tl = lds + g * L_TILE;
        const int64_t t = (int64_t)TPW * qi + g;
        const bool valid = t < num_tiles;
        Lane st;
        st.lane = lane; st.j = j; st.g = g; st.unclearm = 0ull; st.kstep = 0; st.a2 = 0.0;
        const bool isA = j < c;
        st.livemask = __builtin_amdgcn_ballot_w64(isA);
        double a[WR];
        {
            // =============== phase 1: A -> R ===============
            // The four tiles of the quad are 4 r c consecutive doubles: sixteen coalesced 512-byte loads, then through the wave's LDS (idle
            // until the first publication) to lane = column, two tiles at a time with an odd column stride.  (Every lane loading its own
            // column directly costs 64 separate cache lines per load instruction: 20 000 tiles of 16 x 16 took 99 us that way, 69 with
            // bdqr_small.hip's staged sweeps.)
            const int half = HT * rc;                                 // doubles of one staging pass: half the wave's tiles (<= 512)
            const int n4 = (int)(num_tiles - TPW * qi < TPW ? num_tiles - TPW * qi : TPW) * rc;
            const double* qbase = tiles + (int64_t)TPW * qi * rc;
            const int pad = (r & 1) ? 0 : 1, RS = r + pad;
            const unsigned M = (65536u + (unsigned)r - 1u) / (unsigned)r;      // (e M) >> 16 = e / r for e < 1024, 9 <= r <= 16
            const double* colp = lds + ((g & (HT - 1)) * c + (isA ? j : 0)) * RS;
            {
                // (the loads of the second pair are issued when the first pair's registers are free: 16 + 16 live values otherwise)
                double ld[MLD];
#pragma unroll
                for (int m = 0; m < MLD; ++m) { const int e = 64 * m + lane; ld[m] = QRK_TILE_LOAD(qbase + ((e < half && e < n4) ? e : 0)); }
#pragma unroll
                for (int m = 0; m < MLD; ++m) {
                    const int e = 64 * m + lane;
                    if (e < half && e < n4) lds[e + (pad ? (int)(((unsigned)e * M) >> 16) : 0)] = ld[m];
                }
            }
            double ld2[MLD];
#pragma unroll
            for (int m = 0; m < MLD; ++m) { const int e2 = 64 * m + lane; ld2[m] = QRK_TILE_LOAD(qbase + ((e2 < half && half + e2 < n4) ? half + e2 : 0)); }
            __builtin_amdgcn_wave_barrier();
            // (every lane reads in the first pass: an array that is first defined under a condition is carried around the loop as live values)
#pragma unroll
            for (int i = 0; i < WR; ++i) a[i] = colp[i < r ? i : 0];
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int m = 0; m < MLD; ++m) {
                const int e2 = 64 * m + lane;
                if (e2 < half && half + e2 < n4) lds[e2 + (pad ? (int)(((unsigned)e2 * M) >> 16) : 0)] = ld2[m];
            }
            __builtin_amdgcn_wave_barrier();
            if (g >= HT) {
#pragma unroll
                for (int i = 0; i < WR; ++i) a[i] = colp[i < r ? i : 0];
            }
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int i = 0; i < WR; ++i) {
                // rows >= r are zero; a lane without a column holds zeros; a tile beyond the batch: diag(64 .. 49), nothing of it is stored
                const double v = (isA && i < r) ? a[i] : 0.0;
                a[i] = valid ? v : ((i == j) ? (double)(64 - j) : 0.0);
            }
            {
                double s0 = 0.0, s1 = 0.0;
#pragma unroll
                for (int i = 0; i < WR; i += 2) { s0 = fma(a[i], a[i], s0); s1 = fma(a[i + 1], a[i + 1], s1); }
                const double s = s0 + s1;
                st.nu2 = isA ? s : -1.0;                     // (a lane without a column never becomes a pivot)
                st.thr = s * THR_HI;
            }
#define QRK_Q16_STEP(K) if ((K) < WR && (K) < c) step<WR, ((K) < WR ? (K) : 0), PIVOT, HC>(a, tl, st, r);
            QRK_Q16_0_15(QRK_Q16_STEP)
#undef QRK_Q16_STEP
            // ---- R: lane j holds column p = kstep of R in rows 0 .. p; the packed CSC value order of m_R (BlockDiagonalSparseQR.h:475-479)
            // puts entry (i, p) at p (p + 1) / 2 + i -- a contiguous run per lane; the permutation splice (:519-521): the column chosen at
            // step p ends at position p
            int ln = threadIdx.x;
            asm volatile("" : "+v"(ln));
            const int jj = ln & (WR - 1);
            if (valid && jj < c) {
                const int p = st.kstep;
                const int cbase = (int)(t * c);
                perm[cbase + p] = cbase + jj;
                double* dst = r_vals + t * nr + ((p * (p + 1)) >> 1);
#pragma unroll
                for (int i = 0; i < WR; ++i)
                    if (i <= p) dst[i] = a[i];
                if (HC && hcoeffs) hcoeffs[cbase + jj] = lds[(ln >> LG) * L_TILE + L_TAU + jj];
            }
            // a decision inside its error margin, anywhere in the tile: the wave redoes the tile after its rounds
            const bool f = ((st.unclearm >> (WR * (ln >> LG))) & ((1ull << WR) - 1ull)) != 0ull;
            if (f && valid) flagbits |= 1u << round;
        }
        {
            // =============== phase 2: Q = H_0 ... H_{c-1}, backward ===============
            int ln = threadIdx.x;
            asm volatile("" : "+v"(ln));
            const int jj = ln & (WR - 1);
            const double* tl2 = lds + (ln >> LG) * L_TILE;
            double q[WR];
#pragma unroll
            for (int i = 0; i < WR; ++i) q[i] = (i == jj) ? 1.0 : 0.0;
            // (s_K and ng_K of the tile's reflectors: lane l keeps entry l & 15; back_step reads them through DPP)
            const double sv = tl2[L_S + jj], ngv = tl2[L_NG + jj];
#define QRK_Q16_BACK(K) if ((K) < WR && (K) < c) back_step<WR, ((K) < WR ? (K) : 0)>(q, tl2, ln, sv, ngv);
            QRK_Q16_15_0(QRK_Q16_BACK)
#undef QRK_Q16_BACK
            // row-major rows of Q_i are the CSR value order of m_Q in both FullQ ([U|N] split, :455-471) and BlockDiagonalQ (:480-492)
            // layouts: lane j holds COLUMN j of Q_i, one coalesced store of r doubles per row and tile
            if (valid && jj < r) {
                double* dst = q_vals + t * rr + jj;
#pragma unroll
                for (int i = 0; i < WR; ++i)
                    if (i < r) dst[i * r] = q[i];
            }
        }
        __builtin_amdgcn_wave_barrier();
    }
    // ---- the flagged tiles, again, with the reference's own operation order (rare: generic data never gets here; one call, out of
    // line, so that the loop above keeps its registers)
    if (__builtin_expect(__builtin_amdgcn_ballot_w64(flagbits != 0u) != 0ull, 0))
        redo_flagged<WR, PIVOT>(flagbits, qi0, r, c, lds, tiles, q_vals, r_vals, perm, HC ? hcoeffs : nullptr);
    }
    (void)redo_count; (void)redo_ids;
}

int bdqr_quad_waves_per_cu(int r) { return 4 * (r > 8 ? QRK_QUAD_WAVES : QRK_QUAD_WAVES8); }
// 9 .. 16 rows: four tiles per wave; 5 .. 8 rows: eight (two tiles per DPP row, the FMAs told apart by the bank mask)
bool bdqr_quad_supported(int r, int c) { return r >= 5 && r <= 16 && c >= 1 && c <= r; }

// Uniform batches of r x c tiles, r <= 16, c <= r.  num_wg: resident wave slots.
void launch_bdqr_quad(int64_t num_tiles, int r, int c, int pivoting, const double* tiles, double* q_vals, double* r_vals, int32_t* perm,
                      double* hcoeffs, int num_wg, int32_t* redo_count, int32_t* redo_ids, hipStream_t stream)
{
    if (num_tiles <= 0) return;
    const int tpw = r > 8 ? 4 : 8;
    const int64_t nquads = (num_tiles + tpw - 1) / tpw;
    const int64_t nwg = nquads < num_wg ? nquads : num_wg;
    const dim3 grid((unsigned)nwg), block(64);
#define QRK_Q16_LAUNCH(W, P, H) hipLaunchKernelGGL((bdqr_quad_kernel<W, P, H>), grid, block, 0, stream, num_tiles, r, c, tiles, q_vals, r_vals, perm, hcoeffs, redo_count, redo_ids)
#define QRK_Q16_LAUNCH2(W) do { if (pivoting) { if (hcoeffs) QRK_Q16_LAUNCH(W, true, true); else QRK_Q16_LAUNCH(W, true, false); } \
                                else { if (hcoeffs) QRK_Q16_LAUNCH(W, false, true); else QRK_Q16_LAUNCH(W, false, false); } } while (0)
    if (r > 8) QRK_Q16_LAUNCH2(16); else QRK_Q16_LAUNCH2(8);
#undef QRK_Q16_LAUNCH2
#undef QRK_Q16_LAUNCH
}

}  // namespace qrk
