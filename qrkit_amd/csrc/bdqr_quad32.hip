// bdqr_quad32.hip -- uniform batches of 32 x 32 tiles, FOUR tiles per wavefront: A_i P_i = Q_i R_i with explicit Q_i, for gfx950.
// The third generation of the headline kernel (bdqr_pair.hip, bdqr_pair4.hip are the first two; the plan picks per launch size).
//
// Same reference seam: the body of the hot loop of QRKit::BlockDiagonalSparseQR::factorize
// (src/QRKit/BlockDiagonalSparseQR.h:432-526): blockSolver.compute(block) (:437-438, Eigen ColPivHouseholderQR / HouseholderQR),
// Qi = blockSolver.matrixQ() (:446), the Q / R value assembly (:455-500) and the column-permutation splice (:519-521).
//
// Why a third generation.  bdqr_pair4.hip sits on its instruction count, not on memory: 4 581 VALU instructions per pair of tiles, of
// which only 1 984 are the FMAs of the dot products and updates -- the rest (arg-max, the reflector's square root and reciprocal, the
// decision compares, the pivot lane's own norm) is issued once per step for the whole wave and serves TWO tiles
// (profiles/r05_k1_pmc_summary.txt: a 50 us issue floor under a 68 us launch).  Here the same per-step instructions serve FOUR:
//   * a tile is one DPP row of 16 lanes; lane 16 g + c owns columns c (slot 0) and 16 + c (slot 1) of tile g in 2 x 32 row registers
//     -- row_newbcast already broadcasts inside a row, so every dot / update FMA serves four tiles;
//   * the pivot lane of each tile publishes its column to LDS from whichever slot holds it (two masked runs of 8-byte stores: the LDS
//     pipe takes a store for the same time whether one lane or 64 are active, and now four lanes are), every lane takes elements c and
//     16 + c, the FMAs read them through the row_newbcast operand; |x_tail|^2 is the pivot lane's own dot product, spread over its
//     row by an integer OR over the row (every other lane contributes zero) -- no LDS round trip on the critical path;
//   * arg-max over a row of 16 lanes (four DPP stages, no v_permlane16_swap), reflector scalars, compares and ballots once per step;
//   * phase 2 (Q = H_0 ... H_31 by backward accumulation in the same registers) skips slot 0 for K >= 16: those columns of Q are still
//     unit vectors there;
//   * 128 data registers + the working set under __launch_bounds__(64, 2): two waves per SIMD = 8 tiles per SIMD, 32 per CU, as
//     before; 4 992 B of LDS per tile (the reflectors of phase 2), 19 968 B per wave.
// Arithmetic, decisions and the exact path exactly as bdqr_pair4.hip (squared norms with the LAWN-176 downdate, integer arg-max on the
// high words with a filter, un-normalised reflector, margins; a flagged tile is redone by the wave itself in Eigen's own operation
// order after its rounds): on the fast path the two kernels compute the same products in the same order.
#include "qrk_device.h"
#include "bdqr_exact_tile.h"

#include <float.h>
#include <cstdlib>

namespace qrk {

namespace q32 {

using namespace decide;

// Diagnostic only (tools/q32_stamps.py): -DQRK_Q32_STAMP records s_memrealtime (100 MHz, one clock for the whole chip) of every quad at
// the start of its round, when its tiles are in registers, at the end of phase 1 and at the end -- in the array passed as `hcoeffs`
#ifdef QRK_Q32_STAMP
#define QRK_Q32_STAMP_AT(slot) do { if (threadIdx.x == 0) reinterpret_cast<long long*>(hcoeffs)[qi * 4 + (slot)] = (long long)__builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define QRK_Q32_STAMP_AT(slot) do { } while (0)
#endif

// Diagnostic only (tools/q32_prof.py): -DQRK_Q32_PROF accumulates s_memtime ticks (= shader cycles) per phase of the step in workgroup 0 and
// prints them; every stamp drains the LDS queue, so a phase's figure includes the wait for the LDS operations issued in it.  Never timed.
#ifdef QRK_Q32_PROF
#define Q32_TICK(z) do { __builtin_amdgcn_sched_barrier(0); const unsigned long long t1_ = __builtin_amdgcn_s_memtime(); st.pt[z] += t1_ - st.pt0; st.pt0 = t1_; __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define Q32_TICK(z) do { } while (0)
#endif

constexpr int WR = 32;
constexpr int FILTER = 256;              // pivot candidates: high word of the squared norm within 2^-12 (relative) of the largest
// LDS per TILE (doubles): reflector K (the pivot column of step K, as published) holds rows K .. 31 at cb(K)
constexpr int cb(int k) { int s = 0; for (int q = 0; q < k; ++q) s += WR - q; return s; }
constexpr int L_V = 0;
constexpr int L_S = cb(WR);              // [32] s = x0 - beta
constexpr int L_NG = L_S + WR;           // [32] 1 / (beta (beta - x0)) (the sign of -gamma goes into the multiply's source modifier)
constexpr int L_TAU = L_NG + WR;         // [32] tau; before that the hand-off word of |x_tail|^2 (QRK_Q32_TSQ_LDS)
constexpr int L_TILE = L_TAU + WR;       // 624 doubles = 4 992 B per tile, 19 968 B per wave: eight waves per CU
static_assert(cb(WR) == 528 && L_TILE * 8 * 4 * 8 <= 160 * 1024, "eight waves per CU");
constexpr int STAGE_LD = WR + 2;         // the staging of a tile (lane = two rows -> lane = column) uses [32][34] doubles; two tiles at a time
constexpr int STAGE_TILE = WR * STAGE_LD;
static_assert(2 * STAGE_TILE <= 4 * L_TILE, "two staged tiles fit the wave's LDS");

// |x_tail|^2 from the pivot lane to its row: 0 = an OR over the row through DPP (8 VALU instructions, no LDS), 1 = through one LDS word
#ifndef QRK_Q32_TSQ_LDS
#define QRK_Q32_TSQ_LDS 0
#endif
#ifndef QRK_Q32_TSQ_EARLY
#define QRK_Q32_TSQ_EARLY 0
#endif
// QRK_Q32_ILV 1 (experiment; implies TSQ_EARLY): the dependent operations of |x_tail|^2 / square root / reciprocal are issued BETWEEN groups of
// four dot-product FMAs, order pinned -- a wave issues in order, so a stalled dependent operation otherwise holds up the independent FMAs behind it
#ifndef QRK_Q32_ILV
#define QRK_Q32_ILV 0
#endif
#if QRK_Q32_ILV
#undef QRK_Q32_TSQ_EARLY
#define QRK_Q32_TSQ_EARLY 1
#endif
// QRK_Q32_PRIO (experiments): 1 = the wave in an odd slot of its SIMD runs at priority 1 for the whole launch (static asymmetry between
// the two waves of a SIMD); 2 = that wave starts its first phase 1 half a step late
#ifndef QRK_Q32_PRIO
#define QRK_Q32_PRIO 0
#endif
#ifndef QRK_Q32_PUB
#define QRK_Q32_PUB 0
#endif
// QRK_Q32_PUB_WIDE 1: the publication stores may be merged by hipcc (ds_write2_b64: half the instructions, the same store-path cycles)
#ifndef QRK_Q32_SKEW
#define QRK_Q32_SKEW 3
#endif
#ifndef QRK_Q32_PUB_WIDE
#define QRK_Q32_PUB_WIDE 0
#endif
#if QRK_Q32_PUB_WIDE
#define QRK_Q32_PUBQ
#else
#define QRK_Q32_PUBQ volatile
#endif
// doubles of global scratch per workgroup: the exact routine's working copy (its Q is in the wave's LDS)
constexpr int EXACT_SCRATCH = 1024;

#define QRK_Q32_0_31(M)                                                                          \
    M(0) M(1) M(2) M(3) M(4) M(5) M(6) M(7) M(8) M(9) M(10) M(11) M(12) M(13) M(14) M(15) M(16)  \
    M(17) M(18) M(19) M(20) M(21) M(22) M(23) M(24) M(25) M(26) M(27) M(28) M(29) M(30) M(31)
#define QRK_Q32_31_0(M)                                                                          \
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
// d += X[N] * c, X read through DPP row_newbcast (element N of the lane's row of 16 lanes = of its tile)
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
// max over every row of 16 lanes, in every lane of the row: four fused DPP stages (quad xor 1, quad xor 2, half-row mirror, row mirror)
__device__ __forceinline__ int row16_max_i32_fused(int v)
{
    int m;
    asm("s_nop 1\n\t"
        "v_max_i32_dpp %0, %1, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\t"
        "v_max_i32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\t"
        "v_max_i32_dpp %0, %0, %0 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\t"
        "v_max_i32_dpp %0, %0, %0 row_mirror row_mask:0xf bank_mask:0xf"
        : "=&v"(m) : "v"(v));
    return m;
}
__device__ __forceinline__ unsigned row16_max_u32(unsigned v)
{
    v = max(v, (unsigned)dpp_i32<0xB1>((int)v));
    v = max(v, (unsigned)dpp_i32<0x4E>((int)v));
    v = max(v, (unsigned)dpp_i32<0x141>((int)v));
    v = max(v, (unsigned)dpp_i32<0x140>((int)v));
    return v;
}
// OR over every row of 16 lanes of the two words of a double, in every lane of the row.  With the value in ONE lane of the row and
// zero in the others this is the broadcast of that lane, whichever it is (64-bit DPP operations know row_newbcast only, and that names
// its lane at compile time).  The two words alternate, so one s_nop 0 per stage covers the VALU write -> DPP read hazard.
__device__ __forceinline__ double row16_or_f64(double v)
{
    int lo = __double2loint(v), hi = __double2hiint(v);
    asm("s_nop 1\n\t"
        "v_or_b32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
        "v_or_b32_dpp %1, %1, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 0\n\t"
        "v_or_b32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
        "v_or_b32_dpp %1, %1, %1 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 0\n\t"
        "v_or_b32_dpp %0, %0, %0 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
        "v_or_b32_dpp %1, %1, %1 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 0\n\t"
        "v_or_b32_dpp %0, %0, %0 row_mirror row_mask:0xf bank_mask:0xf\n\t"
        "v_or_b32_dpp %1, %1, %1 row_mirror row_mask:0xf bank_mask:0xf"
        : "+v"(lo), "+v"(hi));
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double bpermute_f64(int byte_addr, double v)
{
    const int lo = __builtin_amdgcn_ds_bpermute(byte_addr, __double2loint(v));
    const int hi = __builtin_amdgcn_ds_bpermute(byte_addr, __double2hiint(v));
    return __hiloint2double(hi, lo);
}

typedef __attribute__((address_space(3))) double lds_f64;      // (volatile accesses through a generic pointer would become flat_*)

struct Lane {
    int lane, c, g;               // lane = 16 g + c: columns c and 16 + c of tile g of the quad
    unsigned long long live0, live1;   // (wave-uniform) the lanes whose column of A in slot 0 / 1 is not yet chosen
    unsigned long long unclearm;       // (wave-uniform) lanes that saw a decision of their tile inside its error margin
    int kstep0, kstep1;           // position at which the lane's column was chosen
    double nu2_0, nu2_1;          // m_colNormsUpdated^2 (a chosen column carries a negative value)
    double thr0, thr1;            // sqrt(eps) (1 + 2^-12) m_colNormsDirect^2
    double a2;                    // |A|^2 of the lane's tile: squared norm of its first pivot column (scale of the decision margins)
#ifdef QRK_Q32_PROF
    unsigned long long pt[16], pt0;
#endif
};

// The elements of the published column that this lane broadcasts: xc[m] = element 16 m + c of its tile's column
template <int K>
__device__ __forceinline__ void load_chunks(const double* tl, int c, double (&xc)[2])
{
    constexpr int M0 = (K + 1) >> 4;
    const double* vcol = tl + L_V + cb(K) - K + c;
#pragma unroll
    for (int m = M0; m < 2; ++m) xc[m] = *(const volatile lds_f64*)(vcol + 16 * m);     // (two ds_read_b64: cheaper than one ds_read2_b64, tools/ubench8.hip)
}

// One step of ColPivHouseholderQR::computeInPlace / HouseholderQR on the four tiles of the wave (see bdqr_pair.hip for the arithmetic:
// squared norms, un-normalised reflector, decisions; bdqr_pair4.hip's step with two columns per lane).
template <int K, bool PIVOT, bool HC>
__device__ __forceinline__ void step(double (&a0)[WR], double (&a1)[WR], double* tl /* this tile's LDS */, Lane& st)
{
    const int lane = st.lane;
    if (!PIVOT) __builtin_amdgcn_sched_barrier(0);          // (no branch separates the steps here: keep hipcc from interleaving them)
    // ---- 1. pivot of each tile
    bool ispiv0, ispiv1;
    if (PIVOT) {
        const int khi0 = __double2hiint(st.nu2_0), khi1 = __double2hiint(st.nu2_1);
        const int mh = row16_max_i32_fused(max(khi0, khi1));
        const int lim = mh - FILTER;
        ispiv0 = khi0 >= lim;
        ispiv1 = khi1 >= lim;
        unsigned long long pm0 = __builtin_amdgcn_ballot_w64(ispiv0), pm1 = __builtin_amdgcn_ballot_w64(ispiv1);
        // (every tile has at least one candidate, so subtracting one from each 16-bit field never borrows across fields)
        const unsigned long long x = pm0 | pm1;
        if (__builtin_expect(((x & (x - 0x0001000100010001ull)) | (pm0 & pm1)) != 0ull, 0)) {
            // several candidates in a tile: the largest (lowest COLUMN among exact ties: the tile is flagged then) and the check of the
            // decision -- a live column within the error margin of the chosen one sends the tile to the exact path, which owns
            // Eigen's first-maximum rule on the current positions
            asm volatile("");
            const bool l0 = ((st.live0 >> lane) & 1ull) != 0ull, l1 = ((st.live1 >> lane) & 1ull) != 0ull;
            bool c0 = l0 && khi0 == mh, c1 = l1 && khi1 == mh;
            const unsigned klo0 = (unsigned)__double2loint(st.nu2_0), klo1 = (unsigned)__double2loint(st.nu2_1);
            const unsigned ml = row16_max_u32(max(c0 ? klo0 : 0u, c1 ? klo1 : 0u));
            c0 = c0 && klo0 == ml;
            c1 = c1 && klo1 == ml;
            const unsigned long long cm0 = __builtin_amdgcn_ballot_w64(c0), cm1 = __builtin_amdgcn_ballot_w64(c1);
            const unsigned f0 = (unsigned)(cm0 >> (16 * st.g)) & 0xffffu, f1 = (unsigned)(cm1 >> (16 * st.g)) & 0xffffu;
            const int slot = f0 ? 0 : 1;
            const unsigned f = f0 ? f0 : f1;
            const int lbl = f ? __builtin_ctz(f) : 0;
            ispiv0 = c0 && slot == 0 && st.c == lbl;
            ispiv1 = c1 && slot == 1 && st.c == lbl;
            const int src = (16 * st.g + lbl) << 2;
            const double b0 = bpermute_f64(src, st.nu2_0), b1 = bpermute_f64(src, st.nu2_1);
            const double t0 = bpermute_f64(src, st.thr0), t1 = bpermute_f64(src, st.thr1);
            const double best = slot ? b1 : b0, thrb = slot ? t1 : t0;
            double m0 = MREL * (st.thr0 + thrb), m1 = MREL * (st.thr1 + thrb);
            if (K > 0) {
                const double e = 4.547473508864641e-13 /* 2^-41 */ * __builtin_sqrt(st.a2 * (best > 0.0 ? best : 0.0));
                m0 += e; m1 += e;
            }
            st.unclearm |= __builtin_amdgcn_ballot_w64(l0 && !ispiv0 && st.nu2_0 >= best - m0) |
                           __builtin_amdgcn_ballot_w64(l1 && !ispiv1 && st.nu2_1 >= best - m1);
            pm0 = __builtin_amdgcn_ballot_w64(ispiv0);
            pm1 = __builtin_amdgcn_ballot_w64(ispiv1);
        }
        st.live0 &= ~pm0;
        st.live1 &= ~pm1;
    } else {
        ispiv0 = K < 16 && st.c == K;
        ispiv1 = K >= 16 && st.c == K - 16;
    }
    Q32_TICK(0);
    if (K == 0 && PIVOT) {
        // the scale of the tile: the squared norm of its first pivot, to every lane of the tile
        double v = ispiv0 ? st.nu2_0 : 0.0;
        v = ispiv1 ? st.nu2_1 : v;
        st.a2 = row16_or_f64(v);
    }
    // ---- 2. publish the column (it is reflector K of phase 2 as well): 8-byte stores (a single-lane ds_write_b64 takes the LDS pipe less
    // than half of a ds_write_b128, tools/ubench8.hip; volatile keeps hipcc from merging them back).  A store costs the CU's store path
    // 2 cycles per source dword whatever the number of active lanes (MI355X_MICROARCH.md, LDS), and that path, not the VALU, is what
    // two runs (one per slot that holds a pivot) saturate: QRK_Q32_PUB 1 = the pivot's slot is selected first (2 v_cndmask per
    // element) and ONE run stores it, 0 = one run per slot
    {
        double* vcol = tl + L_V + cb(K) - K;
        if (ispiv0) { st.kstep0 = K; st.nu2_0 = __hiloint2double((int)0xBF800000, __double2loint(st.nu2_0)); }
        if (ispiv1) { st.kstep1 = K; st.nu2_1 = __hiloint2double((int)0xBF800000, __double2loint(st.nu2_1)); }
        if (!PIVOT) {
            if (K < 16) {
                if (ispiv0) {
#pragma unroll
                    for (int i = K; i < WR; ++i) *(QRK_Q32_PUBQ lds_f64*)(&vcol[i]) = a0[i];
                }
            } else if (ispiv1) {
#pragma unroll
                for (int i = K; i < WR; ++i) *(QRK_Q32_PUBQ lds_f64*)(&vcol[i]) = a1[i];
            }
        } else if (QRK_Q32_PUB == 2) {
            // select first, store afterwards, sixteen elements at a time: a ds_write behind a VALU instruction of the same wave waits for the
            // vector pipe to drain (tools/ubench_publish.hip: 42 cycles per element when every store follows its two v_cndmask, 12.7 for a
            // store in a run of stores), so the selects of a batch all come before its stores
            if (ispiv0 || ispiv1) {
#pragma unroll
                for (int i0 = K; i0 < WR; i0 += 16) {
                    double x[16];
#pragma unroll
                    for (int q = 0; q < 16; ++q) {
                        if (i0 + q < WR) {
                            x[q] = ispiv1 ? a1[i0 + q < WR ? i0 + q : 0] : a0[i0 + q < WR ? i0 + q : 0];
                            asm volatile("" : "+v"(x[q]));
                        }
                    }
#pragma unroll
                    for (int q = 0; q < 16; ++q)
                        if (i0 + q < WR) *(volatile lds_f64*)(&vcol[i0 + q]) = x[q];
                }
            }
        } else if (QRK_Q32_PUB) {
            if (ispiv0 || ispiv1) {
                // (the selects run QRK_Q32_SKEW elements ahead of the stores: a store that waits for the select just before it holds up
                //  the wave's in-order issue -- 630 cycles per step for 16 stores against 520 for the 33 of the two-run form)
                constexpr int SK = QRK_Q32_SKEW;
                double xs[SK + 1];
#pragma unroll
                for (int q = 0; q < SK; ++q) {
                    xs[q] = (K + q < WR) ? (ispiv1 ? a1[K + q < WR ? K + q : 0] : a0[K + q < WR ? K + q : 0]) : 0.0;
                    asm volatile("" : "+v"(xs[q]));
                }
#pragma unroll
                for (int i = K; i < WR; ++i) {
                    if (i + SK < WR) {
                        xs[SK] = ispiv1 ? a1[i + SK < WR ? i + SK : 0] : a0[i + SK < WR ? i + SK : 0];
                        asm volatile("" : "+v"(xs[SK]));
                    }
                    *(volatile lds_f64*)(&vcol[i]) = xs[0];
#pragma unroll
                    for (int q = 0; q < SK; ++q) xs[q] = xs[q + 1];
                }
            }
        } else {
            if (ispiv0) {
#pragma unroll
                for (int i = K; i < WR; ++i) *(QRK_Q32_PUBQ lds_f64*)(&vcol[i]) = a0[i];
            }
            if (ispiv1) {
#pragma unroll
                for (int i = K; i < WR; ++i) *(QRK_Q32_PUBQ lds_f64*)(&vcol[i]) = a1[i];
            }
        }
    }
    Q32_TICK(1);
    __builtin_amdgcn_wave_barrier();
    // ---- 3. the lanes' elements of it, x0
    double xc[2] = {0.0, 0.0};
    double xk;
    constexpr int M0 = (K + 1) >> 4, MK = K >> 4;
    if (K + 1 < WR) load_chunks<K>(tl, st.c, xc);
    if (MK >= M0) xk = bcast_f64<(K & 15)>(xc[MK]);
    else xk = *(const volatile lds_f64*)(tl + L_V + cb(K));          // (row K is the last one of its chunk: not among the loaded ones)
    asm volatile("" : "+v"(xk));
    Q32_TICK(2);
#if QRK_Q32_ILV
    const double ak0 = a0[K], ak1 = a1[K];
    double d0a = 0.0, d0b = 0.0, d1a = 0.0, d1b = 0.0;
#pragma unroll
    for (int m = M0; m < 2; ++m) asm volatile("s_nop 1" : "+v"(xc[m]));
    // group G: rows K + 1 + 2 G and K + 2 + 2 G of both slots (the same products in the same order per accumulator as the plain form)
#define QRK_Q32_DOTG(G)                                                                                                  \
    {                                                                                                                    \
        constexpr int I0 = K + 1 + 2 * (G), I1 = I0 + 1;                                                                 \
        if constexpr (I0 < WR) { fmac_bcast<(I0 & 15)>((I0 & 1) ? d0b : d0a, xc[I0 >> 4], a0[I0]); fmac_bcast<(I0 & 15)>((I0 & 1) ? d1b : d1a, xc[I0 >> 4], a1[I0]); } \
        if constexpr (I1 < WR) { fmac_bcast<(I1 & 15)>((I1 & 1) ? d0b : d0a, xc[I1 >> 4], a0[I1]); fmac_bcast<(I1 & 15)>((I1 & 1) ? d1b : d1a, xc[I1 >> 4], a1[I1]); } \
        __builtin_amdgcn_sched_barrier(0);                                                                               \
    }
    double tsq = 0.0, s = 0.0, ngp = 0.0;
    unsigned long long degm = ~0ull;
    {
        double sq = 0.0;
        if (K + 1 < WR) {
            if (M0 == 0) { const double u = st.c > K ? xc[0] : 0.0; sq = u * u; }
            { const double u = 16 + st.c > K ? xc[1] : 0.0; sq = fma(u, u, sq); }
        }
        __builtin_amdgcn_sched_barrier(0);
        QRK_Q32_DOTG(0)
        sq += dpp_f64<0xB1>(sq);
        __builtin_amdgcn_sched_barrier(0);
        QRK_Q32_DOTG(1)
        sq += dpp_f64<0x4E>(sq);
        __builtin_amdgcn_sched_barrier(0);
        QRK_Q32_DOTG(2)
        sq += dpp_f64<0x141>(sq);
        __builtin_amdgcn_sched_barrier(0);
        QRK_Q32_DOTG(3)
        sq += dpp_f64<0x140>(sq);
        tsq = K + 1 < WR ? sq : 0.0;
        const double n2 = fma(xk, xk, tsq);
        __builtin_amdgcn_sched_barrier(0);
        QRK_Q32_DOTG(4)
        if (K == 0 && !PIVOT) st.a2 = n2;
        {
            unsigned long long um = 0ull;
            if (K + 1 < WR) {
                degm = __builtin_amdgcn_fcmp(tsq, DBL_MIN, 13 /* ULE */);
                um = degm | __builtin_amdgcn_fcmp(xk * xk, X0_TINY2 * st.a2, 5 /* OLE */);
            }
            if (PIVOT) um |= __builtin_amdgcn_fcmp(n2, PIV_TINY2 * st.a2, 5 /* OLE */);
            st.unclearm |= um;
        }
        const double y = __builtin_amdgcn_rsq(n2);
        __builtin_amdgcn_sched_barrier(0);
        QRK_Q32_DOTG(5) QRK_Q32_DOTG(6)
        double g = n2 * y, hh = 0.5 * y;
        __builtin_amdgcn_sched_barrier(0);
        QRK_Q32_DOTG(7)
        const double e = fma(-hh, g, 0.5);
        __builtin_amdgcn_sched_barrier(0);
        QRK_Q32_DOTG(8)
        g = fma(g, e, g); hh = fma(hh, e, hh);
        __builtin_amdgcn_sched_barrier(0);
        QRK_Q32_DOTG(9)
        const double dd = fma(-g, g, n2);
        __builtin_amdgcn_sched_barrier(0);
        QRK_Q32_DOTG(10)
        const double nrm = fma(dd, hh, g);
        double nbv = __builtin_copysign(nrm, xk);            // beta = -nbv
        __builtin_amdgcn_sched_barrier(0);
        QRK_Q32_DOTG(11)
        s = nbv + xk;
        const double pr = nbv * s;
        double y2 = __builtin_amdgcn_rcp(pr);
        __builtin_amdgcn_sched_barrier(0);
        QRK_Q32_DOTG(12) QRK_Q32_DOTG(13)
        double e2 = fma(-pr, y2, 1.0);
        __builtin_amdgcn_sched_barrier(0);
        QRK_Q32_DOTG(14)
        y2 = fma(y2, e2, y2);
        __builtin_amdgcn_sched_barrier(0);
        QRK_Q32_DOTG(15)
        e2 = fma(-pr, y2, 1.0);
        y2 = fma(y2, e2, y2);
        ngp = y2;
        if (__builtin_expect(degm != 0ull, 0)) {
            asm volatile("");
            if ((degm >> lane) & 1ull) { nbv = -xk; s = 0.0; ngp = 0.0; }
        }
    }
#undef QRK_Q32_DOTG
    double ds0 = d0a + d0b, ds1 = d1a + d1b;
    if (st.c == 0) {
        tl[L_S + K] = s; tl[L_NG + K] = ngp;
        if (HC) tl[L_TAU + K] = (s * s) * ngp;
    }
#else
#if QRK_Q32_TSQ_EARLY
    // (experiment) |x_tail|^2 from the chunks, summed over the row: independent of the dot products below, so the reflector's scalars
    // overlap with them; another summation order than the pivot lane's own dot product
    double tsq_early = 0.0;
    if (K + 1 < WR) {
        double sq = 0.0;
        if (M0 == 0) { const double u = st.c > K ? xc[0] : 0.0; sq = u * u; }
        { const double u = 16 + st.c > K ? xc[1] : 0.0; sq = fma(u, u, sq); }
        sq += dpp_f64<0xB1>(sq);
        sq += dpp_f64<0x4E>(sq);
        sq += dpp_f64<0x141>(sq);
        sq += dpp_f64<0x140>(sq);
        tsq_early = sq;
    }
#endif
    // ---- 4. d = x_tail^T a_tail of every column (accumulators by the parity of the row, as in bdqr_pair4.hip); the pivot lane's own is
    // |x_tail|^2
    const double ak0 = a0[K], ak1 = a1[K];
    double d0a = 0.0, d0b = 0.0, d1a = 0.0, d1b = 0.0;
#pragma unroll
    for (int m = M0; m < 2; ++m) asm volatile("s_nop 1" : "+v"(xc[m]));      // (VALU write -> DPP read hazard, hidden from hipcc by the asm)
#define QRK_Q32_DOT(I) if ((I) > K) { fmac_bcast<((I) & 15)>(((I) & 1) ? d0b : d0a, xc[(I) >> 4], a0[I]); fmac_bcast<((I) & 15)>(((I) & 1) ? d1b : d1a, xc[(I) >> 4], a1[I]); }
    QRK_Q32_0_31(QRK_Q32_DOT)
#undef QRK_Q32_DOT
    double ds0 = d0a + d0b, ds1 = d1a + d1b;
#ifdef QRK_Q32_PROF
    asm volatile("s_nop 0" : "+v"(ds0), "+v"(ds1));
#endif
    Q32_TICK(3);
    double tsq = 0.0;
#if QRK_Q32_TSQ_EARLY
    tsq = tsq_early;
#else
    if (K + 1 < WR) {
#if QRK_Q32_TSQ_LDS
        if (ispiv0) tl[L_TAU + K] = ds0;
        if (ispiv1) tl[L_TAU + K] = ds1;
        __builtin_amdgcn_wave_barrier();
        tsq = tl[L_TAU + K];
        __builtin_amdgcn_wave_barrier();
#else
        double v = ispiv0 ? ds0 : 0.0;
        v = ispiv1 ? ds1 : v;
        tsq = row16_or_f64(v);
#endif
    }
#endif
#ifdef QRK_Q32_PROF
    asm volatile("s_nop 0" : "+v"(tsq));
#endif
    Q32_TICK(4);
    if (K == 0 && !PIVOT) st.a2 = fma(xk, xk, tsq);
    // (decide::unclear_reflector without short-circuit evaluation: three compares straight into wave masks, no control flow)
    unsigned long long degm;                 // lanes whose tail is empty to rounding: !(tsq > DBL_MIN)
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
    double s = nbv + xk;
    double ngp = recip(nbv * s);                             // -ng
    if (__builtin_expect(degm != 0ull, 0)) {
        asm volatile("");
        if ((degm >> lane) & 1ull) { nbv = -xk; s = 0.0; ngp = 0.0; }
    }
    if (st.c == 0) {
        tl[L_S + K] = s; tl[L_NG + K] = ngp;
        if (HC) tl[L_TAU + K] = (s * s) * ngp;
    }
#endif
    double ngam0 = fma(s, ak0, ds0) * -ngp, ngam1 = fma(s, ak1, ds1) * -ngp;      // -gamma of the lane's columns
#ifdef QRK_Q32_PROF
    asm volatile("s_nop 0" : "+v"(ngam0), "+v"(ngam1));
#endif
    Q32_TICK(5);
    // R(K, K): in the pivot lane s x0 + |x_tail|^2 = beta (beta - x0), so its updated entry x0 - s (1 + delta) IS beta to a few ulp -- no
    // select of the beta computed from the norm (which is what Eigen stores: the fast path answers for 1e-12, flagged tiles are redone)
    const double an0 = fma(s, ngam0, ak0), an1 = fma(s, ngam1, ak1);
    a0[K] = an0;                                             // final: later steps work on the rows below
    a1[K] = an1;
    if (!PIVOT) { asm volatile("" : "+v"(a0[K])); asm volatile("" : "+v"(a1[K])); }      // (see bdqr_pair4.hip: keeps hipcc from sinking selects to the store of R)
    // ---- 6. the trailing update (columns already chosen are not masked out: nothing below the diagonal of R is ever read, and what
    // they hold stays bounded -- the reflectors are orthogonal)
#define QRK_Q32_UPD(I) if ((I) > K) { fmac_bcast<((I) & 15)>(a0[I], xc[(I) >> 4], ngam0); fmac_bcast<((I) & 15)>(a1[I], xc[(I) >> 4], ngam1); }
    QRK_Q32_0_31(QRK_Q32_UPD)
#undef QRK_Q32_UPD
#ifdef QRK_Q32_PROF
    asm volatile("s_nop 0" : "+v"(a0[WR - 1]), "+v"(a1[WR - 1]));
#endif
    Q32_TICK(6);
    // ---- 7. LAWN-176 norm downdate (squared form; no clamp at zero: a negative value is <= the threshold and recomputed)
    if (PIVOT && K + 1 < WR) {
        const double nn0 = fma(-an0, an0, st.nu2_0), nn1 = fma(-an1, an1, st.nu2_1);
        st.nu2_0 = nn0;
        st.nu2_1 = nn1;
        const unsigned long long need0 = __builtin_amdgcn_ballot_w64(nn0 <= st.thr0) & st.live0;
        const unsigned long long need1 = __builtin_amdgcn_ballot_w64(nn1 <= st.thr1) & st.live1;
        if (__builtin_expect((need0 | need1) != 0ull, 0)) {
            asm volatile("");
            const bool n0 = ((need0 >> lane) & 1ull) != 0ull, n1 = ((need1 >> lane) & 1ull) != 0ull;
            st.unclearm |= __builtin_amdgcn_ballot_w64(n0 && in_recompute_band(nn0, st.thr0, st.a2)) |
                           __builtin_amdgcn_ballot_w64(n1 && in_recompute_band(nn1, st.thr1, st.a2));      // decision (2)
            double sq0 = 0.0, sq1 = 0.0;
#define QRK_Q32_SQ(I) if ((I) > K) { sq0 = fma(a0[I], a0[I], sq0); sq1 = fma(a1[I], a1[I], sq1); }
            QRK_Q32_0_31(QRK_Q32_SQ)
#undef QRK_Q32_SQ
            if (n0) { st.nu2_0 = sq0; st.thr0 = sq0 * THR_HI; }
            if (n1) { st.nu2_1 = sq1; st.thr1 = sq1 * THR_HI; }
        }
    }
    Q32_TICK(7);
}

// Q_k = H_k Q_{k+1} on the wave's columns of Q (four tiles): reflector K from the tile's LDS (x_tail as published), s_K and ng_K from the
// registers of lane K & 15 of the row through DPP.  Columns 0 .. K - 1 of Q_{k+1} are unit vectors that H_k leaves alone: for
// K >= 16 that is the whole of slot 0.
template <int K>
__device__ __forceinline__ void back_step(double (&q0)[WR], double (&q1)[WR], const double* tl, const int c, const double (&sv)[2], const double (&ngv)[2])
{
    constexpr int M0 = (K + 1) >> 4;
    constexpr bool S0 = K < 16;              // slot 0 takes part
    double xc[2] = {0.0, 0.0};
    if (K + 1 < WR) load_chunks<K>(tl, c, xc);
    const double qk0 = q0[K], qk1 = q1[K];
    double d0a = 0.0, d0b = 0.0, d1a = 0.0, d1b = 0.0;
#pragma unroll
    for (int m = M0; m < 2; ++m) asm volatile("s_nop 1" : "+v"(xc[m]));
#define QRK_Q32_DOT(I) if ((I) > K) { if (S0) fmac_bcast<((I) & 15)>(((I) & 1) ? d0b : d0a, xc[(I) >> 4], q0[I]); fmac_bcast<((I) & 15)>(((I) & 1) ? d1b : d1a, xc[(I) >> 4], q1[I]); }
    QRK_Q32_0_31(QRK_Q32_DOT)
#undef QRK_Q32_DOT
    // gamma and row K with s_K and ng_K read from lane K & 15 of the row through DPP (bitwise what reading them from LDS gives)
    const double ngk = -bcast_f64<(K & 15)>(ngv[K >> 4]);
    double t1 = d1a + d1b;
    fmac_bcast<(K & 15)>(t1, sv[K >> 4], qk1);
    const double ngam1 = t1 * ngk;
    double qn1 = qk1;
    fmac_bcast<(K & 15)>(qn1, sv[K >> 4], ngam1);
    q1[K] = qn1;
    double ngam0 = 0.0;
    if (S0) {
        double t0 = d0a + d0b;
        fmac_bcast<(K & 15)>(t0, sv[K >> 4], qk0);
        ngam0 = t0 * ngk;
        double qn0 = qk0;
        fmac_bcast<(K & 15)>(qn0, sv[K >> 4], ngam0);
        q0[K] = qn0;
    }
#define QRK_Q32_UPD(I) if ((I) > K) { if (S0) fmac_bcast<((I) & 15)>(q0[I], xc[(I) >> 4], ngam0); fmac_bcast<((I) & 15)>(q1[I], xc[(I) >> 4], ngam1); }
    QRK_Q32_0_31(QRK_Q32_UPD)
#undef QRK_Q32_UPD
}

// A flagged tile again, by the wave that factorised it, in Eigen's own operation order (bdqr_exact_tile.h; bitwise what
// bdqr_exact_kernel computes).  The small tables and Q live in the wave's LDS, the working copy in the wave's global scratch.
template <bool PIVOT>
__device__ __noinline__ void redo_exact(int64_t t, double* lds, double* scratch, const double* __restrict__ tiles, double* __restrict__ q_vals,
                                        double* __restrict__ r_vals, int32_t* __restrict__ perm, double* __restrict__ hcoeffs)
{
    exact::Shared sh;
    double* rest = exact::carve_shared<64>(reinterpret_cast<unsigned char*>(lds), 32, 32, sh);
    static_assert(4 * L_TILE * 8 >= 2048 + 8192, "tables + one 32 x 32 array in the wave's LDS");
    double* W = scratch;
    double* q = rest;
    __syncthreads();
    exact::tile_qr<PIVOT, 64>(32, 32, tiles + t * 1024, W, q, sh);
    exact::tile_store<64>(32, 32, (int)(t * 32), W, q, sh, perm, hcoeffs, r_vals + t * 528, q_vals + t * 1024);
    __syncthreads();
}

}  // namespace q32

// PIVOT: ColPivHouseholderQR (else HouseholderQR).  HC: also emit the Householder coefficients.  One wave per workgroup, persistent over
// the quads blockIdx.x, blockIdx.x + gridDim.x, ..; scratch: q32::EXACT_SCRATCH doubles per workgroup (the exact path's working copy).
// direct: every lane loads its two columns straight from the tiles (else the tiles are staged through the wave's LDS, two at a time).
template <bool PIVOT, bool HC>
__global__ void __launch_bounds__(64, 2)
bdqr_quad32_kernel(int64_t num_tiles, const double* __restrict__ tiles, double* __restrict__ q_vals, double* __restrict__ r_vals,
                   int32_t* __restrict__ perm, double* __restrict__ hcoeffs, double* __restrict__ scratch, int direct)
{
    using namespace q32;
    __shared__ __attribute__((aligned(16))) double lds[4 * L_TILE];
    const int64_t nquads = (num_tiles + 3) / 4;
    constexpr int CHUNK = 32;                // rounds per chunk: one 32-bit word per tile remembers the flagged rounds
    if (QRK_Q32_PRIO) {
        const unsigned wid = __builtin_amdgcn_s_getreg(4 /* HW_REG_HW_ID */ | (0 << 6) | (3 << 11));      // wave slot on its SIMD
        if (wid & 1u) {
            if (QRK_Q32_PRIO == 1) __builtin_amdgcn_s_setprio(1);
            if (QRK_Q32_PRIO == 2) { __builtin_amdgcn_s_sleep(12); }
        }
    }
    for (int64_t qi0 = blockIdx.x; qi0 < nquads; qi0 += (int64_t)CHUNK * gridDim.x) {
    unsigned flagbits = 0u;                  // bit r: the tile of this row of lanes in round r of the chunk was flagged
    int64_t qi = qi0;
    for (int round = 0; round < CHUNK && qi < nquads; ++round, qi += gridDim.x) {
        // (per-lane values are re-derived from an opaque lane id in every round: hipcc otherwise hoists loop-invariant address
        //  arithmetic out of the loop and keeps it in registers across the factorisation)
        int lane = threadIdx.x;
        asm volatile("" : "+v"(lane));
        const int g = lane >> 4, c = lane & 15;
        double* tl = lds + g * L_TILE;
        const int64_t t = 4 * qi + g;
        const bool valid = t < num_tiles;
        Lane st;
        st.lane = lane; st.c = c; st.g = g; st.unclearm = 0ull; st.kstep0 = 0; st.kstep1 = 0; st.a2 = 0.0;
        st.live0 = ~0ull; st.live1 = ~0ull;
#ifdef QRK_Q32_PROF
        for (int z = 0; z < 16; ++z) st.pt[z] = 0;
        st.pt0 = __builtin_amdgcn_s_memtime();
#endif
        {
            // =============== phase 1: A -> R ===============
            QRK_Q32_STAMP_AT(0);
            double a0[WR], a1[WR];
            typedef double d2u __attribute__((ext_vector_type(2), aligned(8)));
            typedef double d2a __attribute__((ext_vector_type(2), aligned(16)));
            if (direct) {
                // every lane its own two columns, straight from the tile: 2 x 16 loads of 16 bytes, 256 bytes between lanes (plain loads:
                // the lines are re-used by the following loads of the same lane)
                const double* src = tiles + (valid ? t : 4 * qi) * 1024 + c * 32;
#pragma unroll
                for (int i = 0; i < WR; i += 2) {
                    const d2u v = *reinterpret_cast<const d2u*>(src + i);
                    const d2u w = *reinterpret_cast<const d2u*>(src + 512 + i);
                    a0[i] = v.x; a0[i + 1] = v.y;
                    a1[i] = w.x; a1[i + 1] = w.y;
                }
            } else {
                // the four tiles of the quad, two at a time through the wave's LDS: every load instruction takes 1 KB of a tile (lane l
                // rows 2 (l & 15), +1 of column 4 m + (l >> 4)), a tile is written column by column with a padded stride (16-byte
                // stores) and the lanes of its row read their two columns back (16-byte loads, conflict-free: 34 doubles between lanes).
                // All 32 loads are in flight before the first one is staged.
                d2u ld[4][8];
#pragma unroll
                for (int tt = 0; tt < 4; ++tt) {
                    const int64_t ts = 4 * qi + tt < num_tiles ? 4 * qi + tt : 4 * qi;
                    const double* s = tiles + ts * 1024 + 2 * lane;
#pragma unroll
                    for (int m = 0; m < 8; ++m) ld[tt][m] = QRK_TILE_LOAD(reinterpret_cast<const d2u*>(s + 128 * m));
                }
                double* sw = lds + (lane >> 4) * STAGE_LD + 2 * (lane & 15);
                const double* sr = lds + (g & 1) * STAGE_TILE + c * STAGE_LD;
#pragma unroll
                for (int tt = 0; tt < 2; ++tt)
#pragma unroll
                    for (int m = 0; m < 8; ++m) *reinterpret_cast<d2a*>(sw + tt * STAGE_TILE + 4 * m * STAGE_LD) = d2a{ld[tt][m].x, ld[tt][m].y};
                __builtin_amdgcn_wave_barrier();
                // (every lane reads in the first pass: an array that is first defined under a condition is carried around the round
                //  loop as live values)
#pragma unroll
                for (int i = 0; i < WR; i += 2) {
                    const d2a v = *reinterpret_cast<const d2a*>(sr + i);
                    const d2a w = *reinterpret_cast<const d2a*>(sr + 16 * STAGE_LD + i);
                    a0[i] = v.x; a0[i + 1] = v.y;
                    a1[i] = w.x; a1[i + 1] = w.y;
                }
                __builtin_amdgcn_wave_barrier();
#pragma unroll
                for (int tt = 0; tt < 2; ++tt)
#pragma unroll
                    for (int m = 0; m < 8; ++m) *reinterpret_cast<d2a*>(sw + tt * STAGE_TILE + 4 * m * STAGE_LD) = d2a{ld[2 + tt][m].x, ld[2 + tt][m].y};
                __builtin_amdgcn_wave_barrier();
                if (g >= 2) {
#pragma unroll
                    for (int i = 0; i < WR; i += 2) {
                        const d2a v = *reinterpret_cast<const d2a*>(sr + i);
                        const d2a w = *reinterpret_cast<const d2a*>(sr + 16 * STAGE_LD + i);
                        a0[i] = v.x; a0[i + 1] = v.y;
                        a1[i] = w.x; a1[i + 1] = w.y;
                    }
                }
                __builtin_amdgcn_wave_barrier();
            }
            if (!valid) {
                // (a tile beyond the batch: diag(64 .. 33) -- distinct norms, no tie-breaking; nothing of it is stored)
#pragma unroll
                for (int i = 0; i < WR; ++i) {
                    a0[i] = (i == c) ? (double)(64 - c) : 0.0;
                    a1[i] = (i == 16 + c) ? (double)(48 - c) : 0.0;
                }
            }
            QRK_Q32_STAMP_AT(1);
            {
                double s0 = 0.0, s1 = 0.0, u0 = 0.0, u1 = 0.0;
#pragma unroll
                for (int i = 0; i < WR; i += 2) {
                    s0 = fma(a0[i], a0[i], s0); s1 = fma(a0[i + 1], a0[i + 1], s1);
                    u0 = fma(a1[i], a1[i], u0); u1 = fma(a1[i + 1], a1[i + 1], u1);
                }
                st.nu2_0 = s0 + s1;
                st.nu2_1 = u0 + u1;
                st.thr0 = st.nu2_0 * THR_HI;
                st.thr1 = st.nu2_1 * THR_HI;
            }
            Q32_TICK(8);
#define QRK_Q32_STEP(K) step<K, PIVOT, HC>(a0, a1, tl, st);
            QRK_Q32_0_31(QRK_Q32_STEP)
#undef QRK_Q32_STEP
            // ---- R: the lane holds columns p = kstep0, kstep1 of R in rows 0 .. p; the packed CSC value order of m_R
            // (BlockDiagonalSparseQR.h:475-479) puts entry (i, p) at p (p + 1) / 2 + i -- a contiguous run per column, stored straight
            // from the registers, two rows at a time; the permutation splice (:519-521): the column chosen at step p ends at position p
            int ln = threadIdx.x;
            asm volatile("" : "+v"(ln));
            if (valid) {
                const int cc = ln & 15;
                const int cbase = (int)(t * 32);
                const int p0 = st.kstep0, p1 = st.kstep1;
                perm[cbase + p0] = cbase + cc;
                perm[cbase + p1] = cbase + 16 + cc;
                double* d0 = r_vals + t * 528 + ((p0 * (p0 + 1)) >> 1);
                double* d1 = r_vals + t * 528 + ((p1 * (p1 + 1)) >> 1);
#pragma unroll
                for (int i = 0; i < WR; i += 2) {
                    if (i + 1 <= p0) *reinterpret_cast<d2u*>(d0 + i) = d2u{a0[i], a0[i + 1]};
                    else if (i == p0) d0[i] = a0[i];
                    if (i + 1 <= p1) *reinterpret_cast<d2u*>(d1 + i) = d2u{a1[i], a1[i + 1]};
                    else if (i == p1) d1[i] = a1[i];
                }
                if (HC && hcoeffs) {
                    const double* tt = lds + (ln >> 4) * L_TILE + L_TAU;
                    hcoeffs[cbase + cc] = tt[cc];
                    hcoeffs[cbase + 16 + cc] = tt[16 + cc];
                }
            }
        }
        Q32_TICK(9);
        // a decision inside its error margin, anywhere in the tile: the tile is redone by the exact path after the rounds
        {
            const bool f = ((st.unclearm >> (16 * g)) & 0xffffull) != 0ull;
            if (f && valid) flagbits |= 1u << round;
        }
        {
            // =============== phase 2: Q = H_0 ... H_31, backward ===============
            QRK_Q32_STAMP_AT(2);
            int ln = threadIdx.x;
            asm volatile("" : "+v"(ln));
            const int cc = ln & 15;
            const double* tl2 = lds + (ln >> 4) * L_TILE;
            double q0[WR], q1[WR];
#pragma unroll
            for (int i = 0; i < WR; ++i) { q0[i] = (i == cc) ? 1.0 : 0.0; q1[i] = (i == 16 + cc) ? 1.0 : 0.0; }
            double sv[2], ngv[2];
            // (s_K and ng_K of the 32 reflectors: lane l keeps entries l & 15 and 16 + (l & 15); back_step reads them through DPP)
#pragma unroll
            for (int m = 0; m < 2; ++m) { sv[m] = tl2[L_S + 16 * m + cc]; ngv[m] = tl2[L_NG + 16 * m + cc]; }
#define QRK_Q32_BACK(K) back_step<K>(q0, q1, tl2, cc, sv, ngv);
            QRK_Q32_31_0(QRK_Q32_BACK)
#undef QRK_Q32_BACK
#ifdef QRK_Q32_PROF
            asm volatile("s_nop 0" : "+v"(q0[WR - 1]), "+v"(q1[WR - 1]));
#endif
            Q32_TICK(10);
            // row-major rows of Q_i are the CSR value order of m_Q in both FullQ ([U|N] split, :455-471) and BlockDiagonalQ (:480-492)
            // layouts: the lane holds COLUMNS c and 16 + c of Q_i, two coalesced stores of 128 bytes per row and tile
            if (valid) {
                double* dst = q_vals + t * 1024 + cc;
#pragma unroll
                for (int i = 0; i < WR; ++i) { dst[32 * i] = q0[i]; dst[32 * i + 16] = q1[i]; }
            }
            QRK_Q32_STAMP_AT(3);
        }
        Q32_TICK(11);
#ifdef QRK_Q32_PROF
        if (blockIdx.x == 0 && threadIdx.x == 0)
            printf("q32 prof (s_memtime ticks per quad): load + norms %llu | steps: search %llu  publish %llu  chunks %llu  dot %llu  |x|^2 %llu  scalars %llu  "
                   "update %llu  downdate %llu | R out %llu | back steps %llu | Q out %llu\n", st.pt[8], st.pt[0], st.pt[1], st.pt[2], st.pt[3], st.pt[4],
                   st.pt[5], st.pt[6], st.pt[7], st.pt[9], st.pt[10], st.pt[11]);
#endif
        __builtin_amdgcn_wave_barrier();
    }
    // ---- the flagged tiles, again, with the reference's own operation order (rare: generic data never gets here)
    {
        unsigned f[4];
#pragma unroll
        for (int g2 = 0; g2 < 4; ++g2) f[g2] = (unsigned)__builtin_amdgcn_readlane((int)flagbits, 16 * g2);
        if (__builtin_expect((f[0] | f[1] | f[2] | f[3]) != 0u, 0)) {
            double* sc = scratch + (int64_t)blockIdx.x * EXACT_SCRATCH;
            for (int g2 = 0; g2 < 4; ++g2) {
                unsigned m = f[g2];
                while (m) {
                    const int rnd = __builtin_ctz(m);
                    m &= m - 1;
                    redo_exact<PIVOT>(4 * (qi0 + (int64_t)rnd * gridDim.x) + g2, lds, sc, tiles, q_vals, r_vals, perm, HC ? hcoeffs : nullptr);
                }
            }
        }
    }
    }
}

int64_t bdqr_quad32_scratch_doubles(int num_wg) { return (int64_t)num_wg * q32::EXACT_SCRATCH; }

// Which of the two-phase 32 x 32 kernels a launch of num_tiles tiles takes (qrk_bd_plan_create; QRK_K1_FORM overrides)
// Measured (profiles/r06_k1_quad32.txt): the four-tile form issues 0.73 x the VALU instructions of bdqr_pair4 and is never faster -- two
// waves per SIMD do not cover its dependent step (1 690 cycles alone against 580 of issue), four of bdqr_pair4's do (875 against 468).
// So the plan keeps bdqr_pair4 at every launch size; QRK_K1_FORM=quad32 selects this kernel.
bool bdqr_quad32_preferred(int64_t num_tiles, int num_wg)
{
    (void)num_tiles; (void)num_wg;
    return false;
}

// Uniform 32 x 32 batches (any 8-byte alignment).  num_wg: resident wave slots (8 per CU).  direct < 0: chosen here by the launch size.
hipError_t launch_bdqr_quad32(int64_t num_tiles, int pivoting, const double* tiles, double* q_vals, double* r_vals, int32_t* perm,
                              double* hcoeffs, double* scratch, int num_wg, int direct, hipStream_t stream)
{
    if (num_tiles <= 0) return hipSuccess;
    const int64_t nquads = (num_tiles + 3) / 4;
    const int64_t nwg = nquads < num_wg ? nquads : num_wg;
    if (direct < 0) {
        // (measured, profiles/r06_k1_quad32.txt: every lane loading its two columns directly is slower than the staging through LDS at every
        //  launch size -- +0.3 / +7 / +64 us at 1 250 / 8 192 / 100 000 tiles; QRK_Q32_DIRECT=1 is the diagnostic switch)
        direct = 0;
        if (const char* e = std::getenv("QRK_Q32_DIRECT")) direct = std::atoi(e) != 0;
    }
    const dim3 grid((unsigned)nwg), block(64);
#define QRK_Q32_LAUNCH(P, H) hipLaunchKernelGGL((bdqr_quad32_kernel<P, H>), grid, block, 0, stream, num_tiles, tiles, q_vals, r_vals, perm, hcoeffs, scratch, direct)
#ifdef QRK_Q32_STAMP
    if (pivoting) QRK_Q32_LAUNCH(true, false); else QRK_Q32_LAUNCH(false, false);
#else
    if (pivoting) { if (hcoeffs) QRK_Q32_LAUNCH(true, true); else QRK_Q32_LAUNCH(true, false); }
    else { if (hcoeffs) QRK_Q32_LAUNCH(false, true); else QRK_Q32_LAUNCH(false, false); }
#endif
#undef QRK_Q32_LAUNCH
    return hipGetLastError();
}

}  // namespace qrk
