// bdqr_reg.hip -- the large tiles of a block-diagonal matrix (more than 64 columns, up to 256 x 256, rows >= cols) factorised ON CHIP:
// A_i P_i = Q_i R_i with explicit Q_i, one workgroup of 512 threads per tile, one workgroup per CU, for gfx950.
//
// Same reference seam as bdqr_col.hip (the hot loop of BlockDiagonalSparseQR::factorize, src/QRKit/BlockDiagonalSparseQR.h:432-526,
// Eigen ColPivHouseholderQR / HouseholderQR behind blockSolver.compute), for the tiles whose working copy bdqr_col.hip keeps in a
// global workspace.  There every step of the column-pivoted factorisation reads the live part of the tile once (the norm downdate
// needs row k of the UPDATED matrix, i.e. v^T A over the whole trailing matrix): n^3/3 * 8 bytes = 45 MB for a 256 x 256 tile, 60 MB
// with the rest, served by the Infinity Cache at 6 TB/s over the chip (profiles/r03_k2_pmc.txt) -- 50x the bytes of the tile.
// A 256 x 256 tile is 512 KB, exactly the vector register file of a CU; here 384 KB of it live in registers and 128 KB in LDS:
//
//   * The tile is aligned to the BOTTOM of a padded 256-row frame (padded row = row + 256 - rows): the rows that die first -- the
//     top ones -- are the ones in LDS, and a shorter tile simply starts with its top register rows dead.
//   * Padded rows 0..63 are in LDS (row-major, stride 257: conflict-free by column thread and by row).
//   * Padded rows 64..255 are in registers: two threads per column (lanes l and l + 32 of a wave, 32 columns per wave), thread
//     (j, h) holds the six 16-row chunks 64 + 32 m + 16 h .. + 15 (m = 0..5) of column j: 96 doubles.  Register rows need static
//     indices, so the step is instantiated six times, for m >= V live chunks; a dead chunk costs nothing.
//   * A step is the level-2 Householder step in the un-normalised form of bdqr_pair.hip: the pivot column is published to LDS by
//     its two threads, every thread takes the elements of ITS rows into six registers (lane l holds element l % 16 of each chunk)
//     and the dot product / rank-1 update read them through the DPP row_newbcast operand of v_fmac_f64 -- no LDS operand per FMA.
//     The two halves of a column meet through v_permlane32_swap.
//   * Two barriers per step: after the pivot candidates of the eight waves are written, after the pivot column is published.
//
// Phase 1 ends with the packed factorisation written once to the workgroup's workspace; R, the permutation and Q (blocked backward
// accumulation on the matrix cores) are bdqr_col_finish.h, shared with bdqr_col.hip.  Decisions inside their error margin flag the
// tile for the exact path exactly as in bdqr_col.hip (same tests, qrk_device.h decide::).
#include "qrk_device.h"
#include "bdqr_col_finish.h"

#include <float.h>
#include <cstdlib>
#include <type_traits>

namespace qrk {

namespace reg {

using namespace decide;
// Two instantiations: NWV = 8 waves (512 threads, up to 256 columns, one workgroup per CU: the LDS rows) and NWV = 4 waves (256
// threads, tiles of at most 128 columns and 192 rows: no LDS rows, TWO workgroups per CU -- the steps are a latency chain, a second
// tile in flight on the CU hides half of it).
constexpr int PR = 256;            // rows of the padded frame
constexpr int LR = 64;             // padded rows [0, LR) live in LDS
constexpr int CS = 257;            // stride of an LDS row (doubles)
constexpr int NCH = 6;             // register chunks of 16 rows per thread
constexpr int NB = colfin::NB;

// Candidate of the pivot search: squared updated norm (< 0: none) and (current position << 8 | column): the first maximum in
// Eigen's order is the largest norm, then the smallest position (positions are distinct, the column rides along).
struct Cand { double val; int key; };
__device__ __forceinline__ bool better(const Cand& a, const Cand& b) { return a.val > b.val || (a.val == b.val && a.key < b.key); }

// DPP butterflies (qrk_device.h: dpp_f64 / dpp_i32): quad xor 1, quad xor 2, half-row mirror, row mirror, then the rows through
// v_permlane16_swap -- no LDS round trips.  Both halves of the wave hold the same candidates (two threads per column): 32 lanes.
__device__ __forceinline__ double swap16_f64(double v)
{
    const unsigned vl = (unsigned)__double2loint(v), vh = (unsigned)__double2hiint(v);
    const auto rl = __builtin_amdgcn_permlane16_swap(vl, vl, false, false);
    const auto rh = __builtin_amdgcn_permlane16_swap(vh, vh, false, false);
    // [0] / [1]: the value of the even and of the odd row, in every lane: own ^ both = the partner's
    return __hiloint2double((int)(rh[0] ^ rh[1] ^ vh), (int)(rl[0] ^ rl[1] ^ vl));
}
__device__ __forceinline__ int swap16_i32(int v)
{
    const auto r = __builtin_amdgcn_permlane16_swap((unsigned)v, (unsigned)v, false, false);
    return (int)(r[0] ^ r[1] ^ (unsigned)v);
}
// arg-max in two passes: the largest norm (three instructions per stage: two DPP moves and v_max_f64), then the smallest key among the
// lanes that hold it (one v_min_i32 with a DPP operand per stage) -- the same first maximum as comparing (norm, key) pairs stage by
// stage, in half the instructions
__device__ __forceinline__ Cand best8(Cand c)          // over every group of 8 lanes
{
    double m = c.val;
    m = fmax(m, dpp_f64<0xB1>(m));
    m = fmax(m, dpp_f64<0x4E>(m));
    m = fmax(m, dpp_f64<0x141>(m));
    int kk = c.val == m ? c.key : 0x7fffffff;
    kk = min(kk, dpp_i32<0xB1>(kk));
    kk = min(kk, dpp_i32<0x4E>(kk));
    kk = min(kk, dpp_i32<0x141>(kk));
    return Cand{m, kk};
}
__device__ __forceinline__ Cand half_best(Cand c)      // over the 32 lanes of a half (both halves hold the same candidates)
{
    double m = c.val;
    m = fmax(m, dpp_f64<0xB1>(m));
    m = fmax(m, dpp_f64<0x4E>(m));
    m = fmax(m, dpp_f64<0x141>(m));
    m = fmax(m, dpp_f64<0x140>(m));
    m = fmax(m, swap16_f64(m));
    int kk = c.val == m ? c.key : 0x7fffffff;
    kk = min(kk, dpp_i32<0xB1>(kk));
    kk = min(kk, dpp_i32<0x4E>(kk));
    kk = min(kk, dpp_i32<0x141>(kk));
    kk = min(kk, dpp_i32<0x140>(kk));
    kk = min(kk, swap16_i32(kk));
    return Cand{m, kk};
}

// sum over the 64 lanes, the same bits in every lane (each stage adds the same two partial sums in either order)
__device__ __forceinline__ double wave_sum_d(double v)
{
    v += dpp_f64<0xB1>(v);
    v += dpp_f64<0x4E>(v);
    v += dpp_f64<0x141>(v);
    v += dpp_f64<0x140>(v);
    v += swap16_f64(v);
    double lo, hi;
    {
        const unsigned vl = (unsigned)__double2loint(v), vh = (unsigned)__double2hiint(v);
        const auto rl = __builtin_amdgcn_permlane32_swap(vl, vl, false, false);
        const auto rh = __builtin_amdgcn_permlane32_swap(vh, vh, false, false);
        lo = __hiloint2double((int)rh[0], (int)rl[0]);
        hi = __hiloint2double((int)rh[1], (int)rl[1]);
    }
    return lo + hi;
}

__device__ __forceinline__ double uniform_f64(double v)      // a wave-uniform value into scalar registers
{
    return __hiloint2double(__builtin_amdgcn_readfirstlane(__double2hiint(v)), __builtin_amdgcn_readfirstlane(__double2loint(v)));
}

// lo / hi: the value of lane (l & 31) / (l & 31) + 32, in both lanes
__device__ __forceinline__ void halves(double v, double& lo, double& hi)
{
    const unsigned vl = (unsigned)__double2loint(v), vh = (unsigned)__double2hiint(v);
    const auto rl = __builtin_amdgcn_permlane32_swap(vl, vl, false, false);
    const auto rh = __builtin_amdgcn_permlane32_swap(vh, vh, false, false);
    lo = __hiloint2double((int)rh[0], (int)rl[0]);
    hi = __hiloint2double((int)rh[1], (int)rl[1]);
}
__device__ __forceinline__ double halves_sum(double v) { double lo, hi; halves(v, lo, hi); return lo + hi; }

// d += X[N] * c with X read through DPP row_newbcast (element N of the lane's row of 16), see bdqr_pair.hip
template <int N>
__device__ __forceinline__ void fmac_bcast(double& d, double X, double c)
{
    asm("v_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(d) : "v"(X), "v"(c), "n"(N));
}

// Barrier of the step loop: orders LDS only.  __syncthreads() also waits for the global stores of the step before (row k of R and
// the reflector, fire-and-forget: ~1.5 us of write latency per step); nothing on chip depends on them until the Q accumulation,
// which is behind a full __syncthreads().
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

#define QRK_REG_16(M) M(0) M(1) M(2) M(3) M(4) M(5) M(6) M(7) M(8) M(9) M(10) M(11) M(12) M(13) M(14) M(15)

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

// doubles of the first LDS region: the LDS rows (8 waves) and, after phase 1, the scratch of the Q accumulation
constexpr int scratch_doubles(int nw) { return (nw == 8 ? 2 : 1) * (PR * (NB + 1) + (2 + nw) * NB * NB); }   // (8 waves: two panels per pass)
constexpr int region_doubles(int nw) { return nw == 8 ? LR * CS : scratch_doubles(nw); }
constexpr size_t lds_bytes(int nw)
{
    return (size_t)(region_doubles(nw) + PR /* xv */ + PR /* taus */ + 2 * nw /* cands */) * sizeof(double) + (size_t)(4 + PR) * sizeof(int) + 16;
}
static_assert(scratch_doubles(8) <= LR * CS, "the scratch of the Q accumulation reuses the LDS rows");
static_assert(region_doubles(4) % 2 == 0 && (LR * CS) % 2 == 0, "16-byte alignment of what follows");

}  // namespace reg

template <int NWV>
__global__ void __launch_bounds__(64 * NWV) __attribute__((amdgpu_waves_per_eu(2, 2)))
bdqr_reg_kernel(WaveBatch nb, const double* __restrict__ tiles, double* __restrict__ q_vals, double* __restrict__ r_vals,
                int32_t* __restrict__ perm, double* __restrict__ hcoeffs, double* __restrict__ workspace, int64_t ws_stride,
                int32_t* __restrict__ redo_count, int32_t* __restrict__ redo_ids, int32_t* __restrict__ queue)
{
    using namespace reg;
    constexpr int CT = 64 * NWV, NW = NWV;
    extern __shared__ __attribute__((aligned(16))) double smem[];
    __shared__ int next_tile;
    double* ldsA = smem;                                   // [LR][CS] padded rows 0..63 (8 waves; 4 waves: tiles have no such rows)
    double* xv = ldsA + region_doubles(NW);                // [PR] pivot column of the step, by padded row
    double* taus = xv + PR;                                // [PR]
    double2* cands = reinterpret_cast<double2*>(taus + PR);   // [NW] candidates of the waves: {norm, key in the low word of .y}
    int* flags = reinterpret_cast<int*>(cands + NW);       // [4]
    int* col_of_pos = flags + 4;                           // [PR]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int h = lane >> 5, j = wave * 32 + (lane & 31);

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
        const int pivoting = nb.pivoting;
        // the workgroup's workspace: rows of R by original column, the reflectors by step, the padded frame of Q (bdqr_col_finish.h)
        double* Rw = workspace + (int64_t)blockIdx.x * ws_stride;
        double* Vb = Rw + PR * PR;
        double* Qp = Vb + PR * PR;
        const double* src = tiles + toff;                  // column-major: A(i, jj) = src[jj * r + i]
        const int off = PR - r;                            // padded row of row 0
        const bool isA = j < c;
        const int nl = off < LR ? LR - off : 0;            // rows of the tile that live in LDS
        const int lim = isA ? LR : 0;                      // end of this thread's loops over LDS rows

#ifdef QRK_REG_PROF
        unsigned long long pt[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, pt0 = __builtin_amdgcn_s_memtime();
#define REG_TICK(z) do { const unsigned long long t1 = __builtin_amdgcn_s_memtime(); pt[z] += t1 - pt0; pt0 = t1; } while (0)
#else
#define REG_TICK(z) do { } while (0)
#endif
        // ---- load: register chunks (16 consecutive rows of one column: one cache line), LDS rows
        double a[NCH][16];
        {
            // (unconditional loads from clamped addresses, then a select: a branch per element otherwise)
            const double* colp = src + (int64_t)(isA ? j : 0) * r;
            // 16 bytes per load (the hardware takes 8-byte alignment; a lane reads 16 consecutive rows of its column, one or two lines)
            typedef double d2u __attribute__((ext_vector_type(2), aligned(8)));
#pragma unroll
            for (int m = 0; m < NCH; ++m)
#pragma unroll
                for (int u = 0; u < 16; u += 2) {
                    const int i = LR + 32 * m + 16 * h + u - off;      // rows i, i + 1 of the tile; i = -1: only the second one exists
                    const d2u v = *reinterpret_cast<const d2u*>(colp + (i > 0 ? i : 0));
                    a[m][u] = (isA && i >= 0) ? v.x : 0.0;
                    a[m][u + 1] = (isA && i >= 0) ? v.y : ((isA && i == -1) ? v.x : 0.0);
                }
        }
        if (nl > 0) {
            // lane = padded row (the top nl <= 64 rows of a column are consecutive in memory), a wave per column, eight loads in flight;
            // the padded rows above the tile are zeroed: the step reads and rewrites whole 16-row chunks of the LDS rows
            const int i = lane >= off ? lane - off : 0;
            for (int j0 = wave; j0 < c; j0 += 8 * NW) {
                double v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) { const int jj = j0 + NW * u; v[u] = src[(int64_t)(jj < c ? jj : c - 1) * r + i]; }
#pragma unroll
                for (int u = 0; u < 8; ++u) { const int jj = j0 + NW * u; if (jj < c) ldsA[lane * CS + jj] = lane >= off ? v[u] : 0.0; }
            }
        }
        if (tid < 4) flags[tid] = 0;
        __syncthreads();

        bool live = isA, unclear = false;
        int pos = j;
        double nu2 = -1.0, thr = 0.0, a2 = 0.0;
        {
            double s0 = 0.0, s1 = 0.0;
#pragma unroll
            for (int m = 0; m < NCH; ++m)
#pragma unroll
                for (int u = 0; u < 16; u += 2) { s0 = fma(a[m][u], a[m][u], s0); s1 = fma(a[m][u + 1], a[m][u + 1], s1); }
            for (int i = off + ((off ^ h) & 1); i < lim; i += 2) { const double v = ldsA[i * CS + j]; s0 = fma(v, v, s0); }
            const double s = halves_sum(s0 + s1);
            if (isA) { nu2 = s; thr = s * THR_HI; }
        }

        REG_TICK(0);
        auto step = [&](auto tag, const int k) {
            constexpr int V = decltype(tag)::value;        // register chunks [V, NCH) are live
            const int kp = k + off;                        // padded row of the diagonal
            const bool in_lds = V == 0 && kp < LR;         // (from level 1 on the diagonal is in a register row: kp >= LR + 32 V)
            // Everything per-lane is re-derived from an opaque thread id in every step: otherwise hipcc hoists the (loop-invariant)
            // LDS addresses out of the step loop and keeps them in scratch across the factorisation.
            int tid = threadIdx.x;
            asm volatile("" : "+v"(tid));
            const int lane = tid & 63, wave = tid >> 6;
            const int h = lane >> 5, j = wave * 32 + (lane & 31), l16 = lane & 15;
            // ---- 1. pivot
            int P = k, ppos = k;
            if (pivoting) {
                // The wave's candidate.  Fast path (round 5): the largest HIGH WORD of the squared norms by an integer DPP max (four fused
                // stages + one swap instead of five three-instruction f64 stages and five key stages); a lane that holds it ALONE holds the
                // largest norm (a > b implies hi(a) >= hi(b)) and writes the candidate itself.  Several holders (norms within 2^-20 of each
                // other, or no live column in the wave): the exact (norm, position) arg-max as before.
                const double cv = live ? nu2 : -1.0;
                const int khi = __double2hiint(cv);
                const int mh = half32_max_i32_fused(khi);
                const bool top = khi == mh;
                const unsigned tm = (unsigned)__builtin_amdgcn_ballot_w64(top);      // (lanes 0..31: both halves hold the same columns)
                if (__builtin_expect((tm & (tm - 1u)) == 0u && mh >= 0, 1)) {
                    if (top && h == 0) cands[wave] = make_double2(cv, __hiloint2double(0, (pos << 8) | j));
                } else {
                    Cand cd{cv, (pos << 8) | j};
                    cd = half_best(cd);
                    if (lane == 0) cands[wave] = make_double2(cd.val, __hiloint2double(0, cd.key));
                }
            }
            REG_TICK(1);
            lds_barrier();      // (also: every read of xv of the step before is done)
            REG_TICK(11);
            if (pivoting) {
                // lane l takes the candidate of wave l % 8: one LDS read, three DPP stages
                const double2 cw = cands[lane & (NW - 1)];
                Cand bb;
                {
                    // the same shortcut over the eight candidates: the wave whose candidate alone has the largest high word wins, and
                    // its norm and key come out as scalars (v_readlane from that lane); otherwise the exact comparison
                    const int ch = __double2hiint(cw.x);
                    int cmx;
                    asm("s_nop 1\n\t"
                        "v_max_i32_dpp %0, %1, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
                        "s_nop 1\n\t"
                        "v_max_i32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
                        "s_nop 1\n\t"
                        "v_max_i32_dpp %0, %0, %0 row_half_mirror row_mask:0xf bank_mask:0xf"
                        : "=&v"(cmx) : "v"(ch));
                    const unsigned wm = (unsigned)__builtin_amdgcn_ballot_w64(ch == cmx) & (NW == 8 ? 0xffu : 0xfu);
                    const int cm0 = __builtin_amdgcn_readfirstlane(cmx);
                    if (__builtin_expect((wm & (wm - 1u)) == 0u && cm0 >= 0, 1)) {
                        const int wl = __builtin_ctz(wm);
                        bb.val = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(cw.x), wl), __builtin_amdgcn_readlane(__double2loint(cw.x), wl));
                        bb.key = __builtin_amdgcn_readlane(__double2loint(cw.y), wl);
                    } else {
                        bb = best8(Cand{cw.x, __double2loint(cw.y)});
                    }
                }
                P = bb.key & 255; ppos = bb.key >> 8;
                if (k == 0) a2 = uniform_f64(bb.val);
                if (live && j != P) {
                    // decision (1), qrk_device.h near_best: its margin is at most MREL (thr + THR_HI a2) + 2^-41 max(a2, best), which
                    // rules the test out for (almost) every column without the square root
                    const double mub = MREL * (thr + THR_HI * a2) + 4.547473508864641e-13 * fmax(a2, bb.val);
                    if (nu2 >= bb.val - mub && near_best(nu2, thr, bb.val, a2)) unclear = true;
                }
                if (isA) { if (j == P) pos = k; else if (pos == k) pos = ppos; }
            }
            REG_TICK(3);
            // ---- 2. publish the pivot column: its two threads their register rows, 64 threads the LDS rows
            if (j == P) {
                live = false;
                if (h == 0) col_of_pos[k] = j;
#pragma unroll
                for (int m = V; m < NCH; ++m) {
                    double2* dst = reinterpret_cast<double2*>(&xv[LR + 32 * m + 16 * h]);
#pragma unroll
                    for (int u = 0; u < 8; ++u) dst[u] = make_double2(a[m][2 * u], a[m][2 * u + 1]);
                }
            }
            if (in_lds && tid < LR) xv[tid] = ldsA[tid * CS + P];
            lds_barrier();
            REG_TICK(4);
            // ---- 3. |x_tail|^2 (every wave the same sum in the same order), the reflector
            double tsq;
            {
                double p = 0.0;
                // (a group of 64 rows is skipped when the level says at compile time that all of it lies above the diagonal: from level 1
                //  on kp >= LR + 32 V; level 0 also holds the steps whose diagonal is in an LDS row)
#pragma unroll
                for (int q4 = 0; q4 < 4; ++q4) {
                    if (V == 0 || 64 * q4 + 63 >= LR + 32 * V) {        // otherwise every row of the group is <= kp
                        const int row = 64 * q4 + lane; double x = xv[row]; x = row > kp ? x : 0.0; p = fma(x, x, p);
                    }
                }
                tsq = uniform_f64(wave_sum_d(p));
            }
            const double xk = uniform_f64(xv[kp]);
            if (k == 0 && !pivoting) a2 = fma(xk, xk, tsq);
            if (unclear_reflector(xk, tsq, k + 1 < r, pivoting != 0, a2)) unclear = true;
            const bool degen = !(tsq > DBL_MIN);
            double beta, s, ng, tau, inv_s;
            if (degen) { beta = xk; s = 0.0; ng = 0.0; tau = 0.0; inv_s = 0.0; }    // makeHouseholder: H = I
            else {
                const double nrm = sqrt_pos(fma(xk, xk, tsq));
                const double nbv = xk >= 0.0 ? nrm : -nrm;   // -beta (Eigen: if (c0 >= 0) beta = -beta; -0.0 counts as >= 0)
                beta = uniform_f64(-nbv);
                s = uniform_f64(nbv + xk);                   // x0 - beta
                ng = uniform_f64(-recip(nbv * s));           // -1 / (beta w)
                tau = -(s * s) * ng;                         // w / beta
                inv_s = uniform_f64(-ng * nbv);              // 1 / (x0 - beta): the essential part is x_tail / (x0 - beta)
            }
            if (tid == 0) { taus[k] = tau; if (hcoeffs) hcoeffs[cbase + k] = tau; }
            REG_TICK(5);
            // the elements of this thread's register rows: x' = (0 .. 0, s, x_tail)
            double xc[NCH];
#pragma unroll
            for (int m = V; m < NCH; ++m) {
                const int row = LR + 32 * m + 16 * h + l16;
                const double x = xv[row];
                // (only chunk V can hold the diagonal or rows above it; with the diagonal in an LDS row none does)
                xc[m] = (m > V || in_lds) ? x : (row > kp ? x : (row == kp ? s : 0.0));
            }
            // ---- 4. row kp of this column before the step
            double ak;
            if (in_lds) ak = isA ? ldsA[kp * CS + j] : 0.0;
            else {
                // element kp & 15 of chunk V, without a dynamic register index (a chain of selects is folded by hipcc into exactly that,
                // and the whole array moves to scratch): the dot product with a one-hot vector, on the DPP broadcast
                double onehot = l16 == (kp & 15) ? 1.0 : 0.0;
                asm volatile("s_nop 1" : "+v"(onehot));
                double sel = 0.0;
#define QRK_REG_SEL(U) fmac_bcast<U>(sel, onehot, a[V][U]);
                QRK_REG_16(QRK_REG_SEL)
#undef QRK_REG_SEL
                double lo, hi;
                halves(sel, lo, hi);
                ak = ((kp >> 4) & 1) ? hi : lo;
            }
            REG_TICK(6);
            // ---- 5. d = x'^T a  (the s_nop covers the VALU write -> DPP read hazard of xc, which hipcc does not see through asm)
            double d0 = 0.0, d1 = 0.0;
#pragma unroll
            for (int m = V; m < NCH; ++m) asm volatile("s_nop 1" : "+v"(xc[m]));
#pragma unroll
            for (int m = V; m < NCH; ++m) {
#define QRK_REG_DOT(U) fmac_bcast<U>((U & 1) ? d1 : d0, xc[m], a[m][U]);
                QRK_REG_16(QRK_REG_DOT)
#undef QRK_REG_DOT
            }
            // the LDS rows: half h takes the 16-row chunks h and h + 2 of padded rows 0..63, with its elements of x' in two more registers
            // and the same DPP broadcast -- one LDS read per row and pass, eight in flight (a loop that reads x and a for every row and
            // waits for both made a step with LDS rows 40 % longer than one without)
            double xl[2];
            if (in_lds) {
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    const int row = 16 * (h + 2 * q) + l16;
                    const double x = xv[row];
                    xl[q] = row > kp ? x : (row == kp ? s : 0.0);
                }
                // (every lane executes the DPP FMAs: a broadcast from a lane that EXEC has switched off does not arrive; lanes without
                //  a column read column 0 and keep nothing)
                {
#pragma unroll
                    for (int q = 0; q < 2; ++q) {
                        if (q == 0 && kp >= 32) continue;        // (rows 0..31 all lie above the diagonal)
                        asm volatile("s_nop 1" : "+v"(xl[q]));
                        const double* ap = ldsA + (16 * (h + 2 * q)) * CS + (isA ? j : 0);
#pragma unroll
                        for (int u0 = 0; u0 < 16; u0 += 8) {
                            double al[8];
#pragma unroll
                            for (int u = 0; u < 8; ++u) al[u] = ap[(u0 + u) * CS];
#define QRK_REG_LDOT(U) fmac_bcast<U>((U & 1) ? d1 : d0, xl[q], al[U & 7]);
                            if (u0 == 0) { QRK_REG_LDOT(0) QRK_REG_LDOT(1) QRK_REG_LDOT(2) QRK_REG_LDOT(3) QRK_REG_LDOT(4) QRK_REG_LDOT(5) QRK_REG_LDOT(6) QRK_REG_LDOT(7) }
                            else { QRK_REG_LDOT(8) QRK_REG_LDOT(9) QRK_REG_LDOT(10) QRK_REG_LDOT(11) QRK_REG_LDOT(12) QRK_REG_LDOT(13) QRK_REG_LDOT(14) QRK_REG_LDOT(15) }
#undef QRK_REG_LDOT
                        }
                    }
                }
            }
            double d = halves_sum(d0 + d1);                  // (x' holds s at the diagonal row: d includes s a_k)
            const double ngam = live ? d * ng : 0.0;         // -gamma of this column
            const double an = fma(s, ngam, ak);              // row kp of the updated column
            REG_TICK(7);
            // ---- 6. update a += (-gamma) x'
#pragma unroll
            for (int m = V; m < NCH; ++m) {
#define QRK_REG_UPD(U) fmac_bcast<U>(a[m][U], xc[m], ngam);
                QRK_REG_16(QRK_REG_UPD)
#undef QRK_REG_UPD
            }
            if (in_lds) {
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    if (q == 0 && kp >= 32) continue;
                    double* ap = ldsA + (16 * (h + 2 * q)) * CS + (isA ? j : 0);
#pragma unroll
                    for (int u0 = 0; u0 < 16; u0 += 8) {
                        double al[8];
#pragma unroll
                        for (int u = 0; u < 8; ++u) al[u] = ap[(u0 + u) * CS];
#define QRK_REG_LUPD(U) fmac_bcast<U>(al[U & 7], xl[q], ngam);
                        if (u0 == 0) { QRK_REG_LUPD(0) QRK_REG_LUPD(1) QRK_REG_LUPD(2) QRK_REG_LUPD(3) QRK_REG_LUPD(4) QRK_REG_LUPD(5) QRK_REG_LUPD(6) QRK_REG_LUPD(7) }
                        else { QRK_REG_LUPD(8) QRK_REG_LUPD(9) QRK_REG_LUPD(10) QRK_REG_LUPD(11) QRK_REG_LUPD(12) QRK_REG_LUPD(13) QRK_REG_LUPD(14) QRK_REG_LUPD(15) }
#undef QRK_REG_LUPD
                        if (live) {                          // (only the stores are predicated)
#pragma unroll
                            for (int u = 0; u < 8; ++u) ap[(u0 + u) * CS] = al[u];
                        }
                    }
                }
            }
            REG_TICK(8);
            // row k of R and reflector k leave for the workspace now (fire and forget): nothing of a chosen column, and no row above
            // the diagonal, is needed on chip again -- dead register rows are simply garbage
            if (h == 0 && (live || j == P)) Rw[k * PR + j] = j == P ? beta : an;
            { const int row = kp + 1 + tid; if (row < PR) Vb[k * PR + row - off] = xv[row] * inv_s; }
            REG_TICK(9);
            // ---- 7. LAWN-176 norm downdate (squared form); Eigen's recompute uses the up-to-date column
            if (pivoting) {
                bool need = false;
                if (live) {
                    const double nn = fma(-an, an, nu2);
                    nu2 = nn;
                    need = nn <= thr;
                    if (need && in_recompute_band(nn, thr, a2)) unclear = true;
                }
                if (__builtin_amdgcn_ballot_w64(need) != 0ull) {
                    double s0 = 0.0;
                    if (need) {
#pragma unroll
                        for (int m = V; m < NCH; ++m)
#pragma unroll
                            for (int u = 0; u < 16; ++u) {
                                const int row = LR + 32 * m + 16 * h + u;
                                s0 = row > kp ? fma(a[m][u], a[m][u], s0) : s0;
                            }
                        if (in_lds)
                            for (int i = kp + 1 + ((kp + 1 + h) & 1); i < lim; i += 2) { const double v = ldsA[i * CS + j]; s0 = fma(v, v, s0); }
                    }
                    const double s2 = halves_sum(s0);
                    if (need) { nu2 = s2; thr = s2 * THR_HI; }
                }
            }
            REG_TICK(10);
        };

        // One loop per number of live register chunks (chunk m dies when the diagonal passes padded row LR + 32 (m + 1)): after
        // its loop a chunk is dead for the register allocator too.
        {
            int k = 0;
#define QRK_REG_LOOP(VV) for (; k < c && k + off < LR + 32 * (VV + 1); ++k) step(std::integral_constant<int, VV>(), k);
            QRK_REG_LOOP(0) QRK_REG_LOOP(1) QRK_REG_LOOP(2) QRK_REG_LOOP(3) QRK_REG_LOOP(4) QRK_REG_LOOP(5)
#undef QRK_REG_LOOP
        }

        REG_TICK(1);
        // ---- a decision inside its error margin: the tile is redone by the exact path
        if (unclear) flags[2] = 1;
        __syncthreads();
        if (tid == 0 && flags[2] != 0 && redo_count) redo_ids[atomicAdd(redo_count, 1)] = gidx;
        __syncthreads();
        // ---- R, the permutation, Q (scratch in the LDS rows, which are dead now)
        constexpr int NPF = NW == 8 ? 2 : 1;                 // panels per pass of the Q accumulation
        double* vs = ldsA;
        double* gm = vs + NPF * PR * (NB + 1);
        double* tm = gm + NPF * NB * NB;
        double* gp = tm + NPF * NB * NB;
        colfin::finish_tile_strips<CT, NW == 8>(Rw, Vb, Qp, r, c, cbase, col_of_pos, taus, vs, gm, tm, gp, q_vals + qoff, r_vals + roff, perm);
        __syncthreads();
        REG_TICK(2);
#ifdef QRK_REG_PROF
        if (blockIdx.x == 0 && threadIdx.x == 0)
            printf("reg prof %d x %d (s_memtime ticks): load + norms %llu  R / perm / Q %llu  steps: search %llu  publish %llu  reflector %llu  "
                   "x chunks + a_k %llu  dot %llu  update %llu  stores %llu  downdate %llu\n", r, c, pt[0], pt[2], pt[3], pt[4], pt[5], pt[6],
                   pt[7], pt[8], pt[9], pt[10]);
        if (blockIdx.x == 0 && threadIdx.x == 0) printf("   search: candidates of the wave %llu  barrier %llu  best of 8 + margins %llu\n", pt[1], pt[11], pt[3]);
#endif
        if (threadIdx.x == 0) next_tile = (int)gridDim.x + atomicAdd(queue, 1);
        __syncthreads();
        t = next_tile;
    }
}

size_t bdqr_reg_smem_bytes(int small) { return reg::lds_bytes(small ? 4 : 8); }
// the 4-wave instantiation takes tiles of at most this size
bool bdqr_reg_small(int max_rows, int max_cols) { return max_cols <= 128 && max_rows <= 192; }
int64_t bdqr_reg_ws_doubles() { return 3 * (int64_t)reg::PR * reg::PR; }

// Tiles with more than 64 columns, rows >= cols, rows <= 256.  workspace: num_wg * ws_stride doubles, ws_stride >=
// bdqr_reg_ws_doubles() (R rows, reflectors, frame of Q: 3 x 256 x 256); queue: one int32 (zeroed here) through which the workgroups take their next tile.
hipError_t launch_bdqr_reg(const WaveBatch& nb, const double* tiles, double* q_vals, double* r_vals, int32_t* perm, double* hcoeffs,
                           double* workspace, int64_t ws_stride, int num_wg, int max_rows, int max_cols, int32_t* redo_count,
                           int32_t* redo_ids, int32_t* queue, hipStream_t stream)
{
    if (nb.num_tiles <= 0) return hipSuccess;
    if (max_rows > reg::PR || max_cols > max_rows || ws_stride < bdqr_reg_ws_doubles()) return hipErrorInvalidValue;
    if (hipError_t e = hipMemsetAsync(queue, 0, sizeof(int32_t), stream)) return e;
    const int64_t want = nb.num_tiles < (int64_t)num_wg ? nb.num_tiles : (int64_t)num_wg;
    if (bdqr_reg_small(max_rows, max_cols)) {
        const size_t smem = reg::lds_bytes(4);
        if (hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(bdqr_reg_kernel<4>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem)) return e;
        hipLaunchKernelGGL(bdqr_reg_kernel<4>, dim3((unsigned)want), dim3(256), smem, stream, nb, tiles, q_vals, r_vals, perm, hcoeffs,
                           workspace, ws_stride, redo_count, redo_ids, queue);
    } else {
        const size_t smem = reg::lds_bytes(8);
        if (hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(bdqr_reg_kernel<8>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem)) return e;
        hipLaunchKernelGGL(bdqr_reg_kernel<8>, dim3((unsigned)want), dim3(512), smem, stream, nb, tiles, q_vals, r_vals, perm, hcoeffs,
                           workspace, ws_stride, redo_count, redo_ids, queue);
    }
    return hipGetLastError();
}

}  // namespace qrk
