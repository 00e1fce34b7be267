// banded.hip -- device side of the block-banded solver, for gfx950.
//
// Replaces the numeric part of QRKit::BandedBlockedSparseQR::factorize
// (src/QRKit/BandedBlockedSparseQR.h:443-519) and the implicit-Q products
// (src/QRKit/SparseBlockYTY.h:100-139 with src/QRKit/BlockYTY.h:152-172):
//   bb_chain_kernel    the sequential chain of dense panels: panel = leftover triangle of the previous
//                      panel stacked on the rows of the next block (:493-507), Eigen::HouseholderQR of
//                      the panel (:468), Y = unit-lower essentials and T = -make_block_householder_
//                      triangular_factor (:471-477), R rows of the panel incl. explicit zeros (:484-491);
//   bb_gather_r_kernel R rows -> CSC value order of m_R;
//   bb_apply_q_kernel  v <- Q^T v (blocks ascending, T^T) or Q v (descending, T), two-segment
//                      gather/scatter per block (SparseQRUtils.h:47-89).
// The chain carries a true dependency from panel to panel, so one workgroup walks it; the panels are
// small (12x8 in the reference's tests, 448x192 in BASELINE configs[2]).
#include "banded_host.h"
#include "qrk_device.h"

#include <float.h>
#include <cstdlib>
#include <type_traits>

namespace qrk {

constexpr int BB_THREADS = 256;
constexpr int BB_WAVES = BB_THREADS / 64;

__device__ __forceinline__ double bb_wave_sum(double v)
{
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off);
    return v;
}

__device__ __forceinline__ double bb_block_sum(double v, double* red)
{
    v = bb_wave_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    double s = 0.0;
#pragma unroll
    for (int w = 0; w < BB_WAVES; ++w) s += red[w];
    return s;
}

// One workgroup walks the whole chain.
//   W      workspace, max_act_rows x max_ncols (column-major, ld = act_rows of the current panel)
//   lo     workspace for the leftover block handed to the next panel
__global__ void __launch_bounds__(BB_THREADS)
bb_chain_kernel(const BBPanel* __restrict__ panels, int num_panels, const int32_t* __restrict__ prowptr,
                const int32_t* __restrict__ pcol, const int64_t* __restrict__ pmap, const double* __restrict__ vals,
                double* __restrict__ W, double* __restrict__ lo, double* __restrict__ y_vals,
                double* __restrict__ t_vals, double* __restrict__ r_stage, int max_act_rows, int max_ncols)
{
    extern __shared__ __attribute__((aligned(16))) double smem[];
    double* xv = smem;                    // [max_act_rows] current Householder vector
    double* hc = xv + max_act_rows;       // [max_ncols] hCoeffs of the panel
    double* uu = hc + max_ncols;          // [max_ncols] row of the T recurrence
    double* red = uu + max_ncols;         // [BB_WAVES]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;

    for (int pi = 0; pi < num_panels; ++pi) {
        const BBPanel p = panels[pi];
        const int m = p.act_rows, n = p.ncols, ld = p.act_rows;

        // ---- Ji = pmat.block(row0, col0, m, n).toDense() (:458, :503) ...
        for (int64_t e = tid; e < (int64_t)m * n; e += BB_THREADS) W[e] = 0.0;
        __syncthreads();
        for (int r = wave; r < m; r += BB_WAVES) {
            const int gr = p.row0 + r;
            for (int e = prowptr[gr] + lane; e < prowptr[gr + 1]; e += 64) {
                const int c = pcol[e] - p.col0;
                if (c >= 0 && c < n) W[(int64_t)c * ld + r] = vals[pmap[e]];
            }
        }
        __syncthreads();
        // ... with its top-left corner replaced by the leftover block of the previous panel (:504-506)
        for (int e = tid; e < p.lo_rows * p.lo_cols; e += BB_THREADS) {
            const int i = e % p.lo_rows, j = e / p.lo_rows;
            W[(int64_t)j * ld + i] = lo[e];
        }
        __syncthreads();

        // ---- Eigen::HouseholderQR of the panel (unblocked; the blocked driver is algebraically the same)
        for (int k = 0; k < n; ++k) {
            double part = 0.0;
            for (int i = k + tid; i < m; i += BB_THREADS) {
                const double v = W[(int64_t)k * ld + i];
                xv[i] = v;
                if (i > k) part = fma(v, v, part);
            }
            const double tailSq = bb_block_sum(part, red);
            const double xk = xv[k];
            double beta, tau, scale;
            if (tailSq <= DBL_MIN) {
                beta = xk; tau = 0.0; scale = 0.0;
            } else {
                const double nrm = sqrt(fma(xk, xk, tailSq));
                beta = xk >= 0.0 ? -nrm : nrm;
                scale = 1.0 / (xk - beta);
                tau = (beta - xk) / beta;
            }
            __syncthreads();
            for (int i = k + 1 + tid; i < m; i += BB_THREADS) {
                const double e = xv[i] * scale;
                xv[i] = e;
                W[(int64_t)k * ld + i] = e;
            }
            if (tid == 0) { W[(int64_t)k * ld + k] = beta; hc[k] = tau; }
            __syncthreads();
            for (int jc = k + 1 + wave; jc < n; jc += BB_WAVES) {
                double* col = W + (int64_t)jc * ld;
                double d = 0.0;
                for (int i = k + 1 + lane; i < m; i += 64) d = fma(xv[i], col[i], d);
                d = bb_wave_sum(d);
                const double tt = tau * (d + col[k]);
                for (int i = k + 1 + lane; i < m; i += 64) col[i] = fma(-tt, xv[i], col[i]);
                if (lane == 0) col[k] -= tt;
            }
            __syncthreads();
        }

        // ---- Y = unit-lower essentials (:471-475), stored as the panel itself: row-major m x n (see bb_chain2_kernel)
        double* Y = y_vals + p.y_off;
        for (int64_t e = tid; e < (int64_t)m * n; e += BB_THREADS) {
            const int i = (int)(e / n), j = (int)(e % n);
            Y[e] = i < j ? 0.0 : (i == j ? 1.0 : W[(int64_t)j * ld + i]);
        }
        // ---- T = -make_block_householder_triangular_factor(Y, hCoeffs) (:476-477), rows n-1 .. 0:
        //      T(i, i+1:) = (-h_i Y(i+1:, i)^T Y(i+1:, i+1:)) T(i+1:, i+1:), T(i,i) = h_i
        double* T = t_vals + p.t_off;
        for (int64_t e = tid; e < (int64_t)n * n; e += BB_THREADS) T[e] = 0.0;
        __syncthreads();
        for (int i = n - 1; i >= 0; --i) {
            for (int c = i + 1 + wave; c < n; c += BB_WAVES) {
                // unit-lower view of column c: 1 at row c, essentials below
                double s = 0.0;
                for (int r = c + 1 + lane; r < m; r += 64) s = fma(W[(int64_t)i * ld + r], W[(int64_t)c * ld + r], s);
                s = bb_wave_sum(s);
                if (lane == 0) uu[c] = -hc[i] * (s + W[(int64_t)i * ld + c]);
            }
            __syncthreads();
            for (int c = i + 1 + tid; c < n; c += BB_THREADS) {
                double s = 0.0;
                for (int j = i + 1; j <= c; ++j) s = fma(uu[j], T[(int64_t)c * n + j], s);
                T[(int64_t)c * n + i] = s;
            }
            if (tid == 0) T[(int64_t)i * n + i] = hc[i];
            __syncthreads();
        }
        for (int64_t e = tid; e < (int64_t)n * n; e += BB_THREADS) T[e] = -T[e];

        // ---- rows of R solved by this panel: V = triu(packed QR), explicit zeros kept (:484-491)
        for (int e = tid; e < p.solved * n; e += BB_THREADS) {
            const int br = e % p.solved, bc = e / p.solved;
            r_stage[p.r_off + e] = (br <= bc && br < m) ? W[(int64_t)bc * ld + br] : 0.0;
        }
        // ---- leftover block for the next panel: V.block(lo_from, lo_from, lo_rows, lo_cols) (:505)
        if (pi + 1 < num_panels) {
            const BBPanel q = panels[pi + 1];
            for (int e = tid; e < q.lo_rows * q.lo_cols; e += BB_THREADS) {
                const int i = e % q.lo_rows, j = e / q.lo_rows;
                const int vr = q.lo_from + i, vc = q.lo_from + j;
                lo[e] = (vr <= vc && vr < m && vc < n) ? W[(int64_t)vc * ld + vr] : 0.0;
            }
        }
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------------------------------
// Second version of the chain (panels up to 256 columns): the panel is kept ROW-major and worked on by
// 1024 threads as a 2-D grid, column slot cs = tid % 256, row group rg = tid / 256 (rows interleaved
// modulo 4), so every sweep is coalesced and there is no cross-lane reduction:
//   * Householder QR with the dot and update sweeps FUSED across steps (no pivoting here, the pivot of
//     step k+1 is column k+1): the sweep that applies reflector k to column c also accumulates
//     x'^T c for the next reflector x' = W(:,k+1) - gamma_{k+1} x, two barriers per reflector;
//   * G = Y^T Y on v_mfma_f64_16x16x4_f64 (16x16 tiles, operands from LDS-staged row chunks), written
//     into the T output;
//   * T by the forward (larft) recurrence T(0:c,c) = -h_c T(0:c,0:c) G(0:c,c), in place, T kept packed
//     in LDS when it fits (n <= 192), the negation the reference stores (:477) applied at the end.
// The first version spent ~16 ms per 448x192 panel (BASELINE configs[2] shape), mostly in wave-wide
// reductions: one per column and reflector in the QR, one per (i, c) pair in the T recurrence.
constexpr int BC_THREADS = 1024;
constexpr int BC_CW = 256;                 // most columns of a panel
constexpr int BC_RC = 16;                  // rows per LDS chunk in the Gram matrix

// Sum over the 64 lanes, the same value in every lane: four DPP steps inside the rows of 16, then the four row sums
// through SGPRs.  (A fixed order: results do not depend on timing.)
__device__ __forceinline__ double bb_wave_sum_dpp(double v)
{
    v += dpp_f64<0xB1>(v);    // quad_perm [1,0,3,2]
    v += dpp_f64<0x4E>(v);    // quad_perm [2,3,0,1]
    v += dpp_f64<0x141>(v);   // row_half_mirror
    v += dpp_f64<0x140>(v);   // row_mirror
    return (readlane_f64(v, 0) + readlane_f64(v, 16)) + (readlane_f64(v, 32) + readlane_f64(v, 48));
}

typedef double bb_d4 __attribute__((ext_vector_type(4)));
#ifndef QRK_BB_PIPE_OB16
#define QRK_BB_PIPE_OB16 1    // blocks of 16 columns in the pipelined strips chain (0: 32 as in round 4, for A/B builds)
#endif
#ifndef QRK_BB_SAME_XCD
#define QRK_BB_SAME_XCD 1     // the workgroups of the pipelined chain on one XCD: its L2 serves the hand-over of the carry rows (0: spread over the XCDs, for A/B)
#endif
#ifndef QRK_BB_P1_ROWS
#define QRK_BB_P1_ROWS 64    // blocks of at most this many rows: a wave per strip of the block update, all rows (no row parts); 64 / 80 / 96 / 112 / 128 measured
#endif
#ifndef QRK_BB_PMAX
#define QRK_BB_PMAX 8    // most row parts a strip of the block update is split into (each part costs a barrier when the partial sums meet)
#endif
#ifndef QRK_BB_ABL
#define QRK_BB_ABL 0      // timing experiments only (wrong results): 1 = no MFMA in the block update, 2 = no W loads, 4 = no W stores
#endif

// sqrt and reciprocal without the FP64 division sequences (as in bdqr_pair.hip: v_rsq / v_rcp seeds, <= 1 ulp)
__device__ __forceinline__ double bb_sqrt_pos(double x)
{
    const double y = __builtin_amdgcn_rsq(x);
    double g = x * y, h = 0.5 * y;
    const double e = fma(-h, g, 0.5);
    g = fma(g, e, g);
    h = fma(h, e, h);
    const double d = fma(-g, g, x);
    return fma(d, h, g);
}
__device__ __forceinline__ double bb_recip(double x)
{
    double y = __builtin_amdgcn_rcp(x);
    double e = fma(-x, y, 1.0);
    y = fma(y, e, y);
    e = fma(-x, y, 1.0);
    y = fma(y, e, y);
    return y;
}

// Pipelining of the strips form over several workgroups (round 4).  The chain is a true dependency from strip to strip -- but not
// from the END of a strip: the carry that panel i takes over from panel i - 1 is rows lo_from .. of panel i - 1, row lo_from + r is
// final as soon as reflector lo_from + r has been applied (later reflectors only touch the rows below), and a block of panel i only
// works on the carry rows its staircase reaches (stack rows below rlim).  So workgroup g factorises the panels g, g + G, ..; after
// every block of columns it publishes how many rows of its panel are final (one word per panel, agent-scope release), and before a
// block it waits until the previous panel has finalised the carry rows that block needs and copies exactly those rows in -- straight
// from the previous panel's storage, there is no leftover buffer.  Round 5: the pipelined chain works in blocks of 16 columns (the grain of
// the blocks is the grain of the overlap) and a panel publishes TWO words per block -- "own": the rows below are final in the columns
// below (after the block's reflectors), "done": final everywhere (after its update) -- so that the next panel's block starts its
// reflectors on the first and only needs the second before its own update.  For 256 x 192 strips at column step 64 the next panel
// starts when this one has run the reflectors of its first 80 columns; three workgroups, two to three panels in flight.
struct BBPipe {
    const double* prev;        // storage of the previous panel (row-major, prev_n columns), or null for the first panel
    const int* prev_done;      // its rows-final word: rows below this are final in EVERY column (published after a block's update)
    int* my_done;              // this panel's
    const int* prev_own;       // its second word: rows below this are final in the columns below it (published after a block's own
    int* my_own;               //   reflectors, before the block's update of the columns to the right); this panel's
    int* abortw;               // the chain's abort word (done[num_panels]): set by the workgroup whose wait ran out, seen by every waiter
    int prev_n, lo_from, lo_rows, lo_cols, lo_stride;
    unsigned spin_limit;       // polls of a rows-final word before the chain is given up
    int aborted;               // this workgroup leaves (its own wait ran out, or a partner's did)
};
// Waits until `word` reaches `target`.  The workgroups of the chain are not guaranteed to be co-resident (three 160 KB workgroups of
// an ordinary launch), so the wait is bounded: when it runs out the workgroup raises the chain's abort word and leaves, every other
// waiter sees the word and leaves too, and the host (launch_bbs_chain's caller, qrk_bbs_factorize) reads the word at its
// synchronisation point and runs the chain again on ONE workgroup.  Returns false (to every thread) when the chain is aborted.
__device__ __forceinline__ bool bb_pipe_wait(BBPipe* pipe, const int* word, int target)
{
    if (threadIdx.x == 0) {
        unsigned spins = 0;
        int bad = 0;
        while (__hip_atomic_load(word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
            if (__hip_atomic_load(pipe->abortw, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) { bad = 1; break; }
            if (++spins > pipe->spin_limit) {
                __hip_atomic_store(pipe->abortw, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                bad = 1;
                break;
            }
            __builtin_amdgcn_s_sleep(2);
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        if (bad) pipe->aborted = 1;
    }
    __syncthreads();
    return pipe->aborted == 0;
}
__device__ __forceinline__ void bb_pipe_publish(int* word, int rows_done, int* word2 = nullptr)
{
    __syncthreads();           // every thread's stores of the block have been issued and waited for (the barrier's release)
    if (threadIdx.x == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        __hip_atomic_store(word, rows_done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (word2) __hip_atomic_store(word2, rows_done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// LDS doubles the blocked QR needs next to the OB-column block: T of the block, then the larger of {two published
// Householder vectors, partial V^T W of 8 column strips}
__host__ __device__ constexpr int bb_qr_aux_doubles(int OB) { return OB * OB + 8 * (OB / 16) * 256; }

// Eigen::HouseholderQR of the m x n panel W (row-major, in place, packed; hCoeffs to hc), blocked two ways:
//  - a block of OB columns lives in registers, one or two columns per wave with the rows over the lanes, and is
//    factorised one reflector at a time (makeHouseholder + applyHouseholderOnTheLeft in the un-normalised form of
//    bdqr_pair.hip): the wave that owns column j publishes its tail in LDS, one barrier, every wave updates its own
//    columns.  No memory traffic but the published vector.
//  - the block reflector I - V T^T V^T of those OB reflectors is then applied to the columns to the right with
//    v_mfma_f64_16x16x4_f64: w = V^T W (V from the LDS block, W streamed from global memory), u = T^T w (the result
//    registers of one MFMA are laid out as the B operand of the next, so w and u never leave the registers),
//    W -= V u.  W crosses this CU's L2 port twice per OB reflectors; with one reflector, or eight, at a time that
//    port was the bound (0.55 ms of a 1.6 ms panel).
// uni: [OB * ld] block (column-major, ld = m | 1), then bb_qr_aux_doubles(OB).
// rlim (or null): rlim[g] = rows of the panel that can be nonzero in the columns [16 g, 16 g + 16) before this factorisation
// (a staircase profile, non-decreasing in g): the block of columns that ends with group g then works on the rows below its
// first one and above rlim[g] only -- the rows beyond are zero in these columns and no reflector of the block touches them.
// NR: 64-row registers per lane that hold a column of the block (a block never works on more than 64 NR rows: the caller picks
// the smallest instantiation that covers its tallest block -- every reflector costs NR FMAs per column and NR LDS reads)
template <int OB, int NR = 256 / OB>
__device__ __attribute__((noinline)) void bb_panel_qr(double* __restrict__ W, const int m, const int n, const int* __restrict__ rlim,
                                                      BBPipe* __restrict__ pipe
#ifdef QRK_BB_PROF
                                            , unsigned long long* qt
#endif
                                            )
{
    // (not inlined: the kernel around it keeps ~100 values live across the panel loop, and the register allocation
    //  of this function should not pay for them.  The LDS carve-up is the kernel's: hc, sc, uni.)
    extern __shared__ __attribute__((aligned(16))) double smem[];
    double* hc = smem;
    double* sc = hc + BC_CW;
    double* uni = sc + 8;
    const int tid = threadIdx.x;
    constexpr int CPW = OB / 16;               // columns of the block per wave
    constexpr int MT = OB / 16;                // 16-row tiles of reflectors
    const int wv = tid >> 6, ln = tid & 63;
    const int ld = m | 1;
    double* blk = uni;
    double* tb = blk + (int64_t)OB * ld;       // [OB * OB] T of the block (row-major, upper; G above the diagonal first)
    constexpr int VL = NR * 64;                // rows a published Householder tail can have
    double* vcol = tb + OB * OB;               // [2 * VL] published Householder tails (zero beyond the rows of the block)
    double* red = vcol;                        // [8 * MT * 256] partial sums of w (after the factorisation)
#ifdef QRK_BB_PROF
    unsigned long long q0 = __builtin_amdgcn_s_memtime();
#define BB_QTICK(z) do { const unsigned long long t1 = __builtin_amdgcn_s_memtime(); qt[z] += t1 - q0; q0 = t1; } while (0)
#else
#define BB_QTICK(z) do { } while (0)
#endif

    int cp_own = 0, cp_full = 0;               // carry rows taken over: up to the current block's last column / in full (the same in every thread)
    for (int jb = 0; jb < n; jb += OB) {
        const int ob = (n - jb) < OB ? (n - jb) : OB;
        int mtop = m;
        if (rlim) { const int lim = rlim[(jb + ob - 1) >> 4]; mtop = lim < m ? lim : m; }
        const int mr = mtop - jb;              // rows jb.. of the panel take part (local row i = panel row jb + i)
        // 0. (pipelined chain) the carry rows this block reaches, in TWO steps (round 5): their entries in the columns up to this block's
        //    last one as soon as the previous panel has run the reflectors of the block that holds them (its "own" word: those entries
        //    are final then, the previous panel's update of its columns to the right does not touch them) -- enough for this block's
        //    reflectors --, the entries to the right before this block's own update (step 5), when the previous panel's update is done.
        //    The block's reflectors so run beside the previous panel's block update instead of behind it.
        int pipe_need = 0;
        if (pipe && pipe->prev && cp_full < pipe->lo_rows) {
            int need = (mtop + pipe->lo_stride - 1) / pipe->lo_stride;
            if (need > pipe->lo_rows) need = pipe->lo_rows;
            pipe_need = need;
            if (need > cp_own) {
#ifdef QRK_BB_PROF
                const unsigned long long tq0 = __builtin_amdgcn_s_memtime();
#endif
                if (!bb_pipe_wait(pipe, pipe->prev_own, pipe->lo_from + need)) return;       // (uniform: the chain is given up)
#ifdef QRK_BB_PROF
                qt[13] += __builtin_amdgcn_s_memtime() - tq0;
#endif
                const int c0 = cp_own, lc = pipe->lo_cols, ce = (jb + ob) < lc ? (jb + ob) : lc;
                for (int e = tid; e < (need - c0) * ce; e += BC_THREADS) {
                    const int i = c0 + e / ce, j = e % ce;
                    W[(int64_t)(i * pipe->lo_stride) * n + j] =
                        i <= j ? pipe->prev[(int64_t)(pipe->lo_from + i) * pipe->prev_n + pipe->lo_from + j] : 0.0;
                }
                __syncthreads();                 // (the copied entries before the block load reads them)
                cp_own = need;                   // (every thread keeps the count itself)
            }
        }
        // 1. block to LDS (coalesced rows of W), then to the registers of the owning waves
        for (int e = tid; e < mr * OB; e += BC_THREADS) {
            const int i = e / OB, l = e - i * OB;
            blk[l * ld + i] = l < ob ? W[(int64_t)(jb + i) * n + jb + l] : 0.0;
        }
        for (int e = tid; e < OB * OB; e += BC_THREADS) tb[e] = 0.0;
        for (int e = tid; e < 2 * VL; e += BC_THREADS) vcol[e] = 0.0;
        __syncthreads();
        double col[CPW][NR];
#pragma unroll
        for (int s = 0; s < CPW; ++s) {
            const int c = wv + 16 * s;
#pragma unroll
            for (int r = 0; r < NR; ++r) { const int i = r * 64 + ln; col[s][r] = (i < mr && c < ob) ? blk[c * ld + i] : 0.0; }
        }
        BB_QTICK(0);
        // 2. the reflectors of the block.  Local row j is the pivot row of reflector j: lane j of the first row register.
        //    Rows beyond mr hold zeros in col[] and in the published vectors, so only that register needs a mask.
        //    The loop is instantiated for the row registers the BLOCK reaches (64 rows each): the first blocks of a staircase panel are short
        //    (32 .. 96 rows in the strips form) and they are the ones the next panel of the pipelined chain waits for.
        auto reflectors = [&](auto nrb_tag) {
        constexpr int NRB = decltype(nrb_tag)::value;
        for (int j = 0; j < ob; ++j) {
            double* vj = vcol + (j & 1) * VL;
            double* scj = sc + (j & 1) * 4;
            if (j >= mr) { if (tid == 0) hc[jb + j] = 0.0; continue; }      // (no rows left: identity)
            if (wv == (j & 15)) {
                auto head = [&](double (&x)[NR]) {
                    // (the raw tail goes to LDS first: its stores are under way while the norm is reduced and the scalars are computed)
                    if (ln > j) vj[ln] = x[0];
#pragma unroll
                    for (int r = 1; r < NRB; ++r) vj[r * 64 + ln] = x[r];
                    double part = ln > j ? x[0] * x[0] : 0.0;
#pragma unroll
                    for (int r = 1; r < NRB; ++r) part = fma(x[r], x[r], part);
                    const double tsq = bb_wave_sum_dpp(part);
                    const double xk = readlane_f64(x[0], j);
                    double nb_, s2, ng, tau, inv_s;
                    if (!(tsq > DBL_MIN)) { nb_ = -xk; s2 = 0.0; ng = 0.0; tau = 0.0; inv_s = 0.0; }
                    else {
                        const double nrm = bb_sqrt_pos(fma(xk, xk, tsq));
                        nb_ = xk >= 0.0 ? nrm : -nrm;
                        s2 = nb_ + xk;                   // x0 - beta
                        inv_s = bb_recip(s2);
                        ng = -bb_recip(nb_) * inv_s;     // -1 / (beta (beta - x0))
                        tau = -(s2 * s2) * ng;           // (beta - x0) / beta
                    }
                    if (ln == 0) { scj[0] = s2; scj[1] = ng; hc[jb + j] = tau; tb[j * OB + j] = tau; }
                    x[0] = ln > j ? x[0] * inv_s : (ln == j ? -nb_ : x[0]);      // essential part (:471-475), beta
#pragma unroll
                    for (int r = 1; r < NRB; ++r) x[r] *= inv_s;
                };
                if (CPW == 1 || (j >> 4) == 0) head(col[0]); else head(col[CPW - 1]);
            }
#ifdef QRK_BB_PROF
            const unsigned long long tw0 = __builtin_amdgcn_s_memtime();
#endif
            __syncthreads();
#ifdef QRK_BB_PROF
            if (wv != (j & 15)) qt[11] += __builtin_amdgcn_s_memtime() - tw0;      // a non-owner's wait for the head of the step
#endif
            const double s2 = scj[0], ng = scj[1];
            bool any = false;
#pragma unroll
            for (int s = 0; s < CPW; ++s) any = any || (wv + 16 * s > j && wv + 16 * s < ob);
            if (any) {
                double v[NR];
                v[0] = ln > j ? vj[ln] : 0.0;
#pragma unroll
                for (int r = 1; r < NRB; ++r) v[r] = vj[r * 64 + ln];
#pragma unroll
                for (int s = 0; s < CPW; ++s) {
                    const int c = wv + 16 * s;
                    if (c > j && c < ob) {
                        double part = 0.0;
#pragma unroll
                        for (int r = 0; r < NRB; ++r) part = fma(v[r], col[s][r], part);
                        const double d = bb_wave_sum_dpp(part);
                        const double ak = readlane_f64(col[s][0], j);
                        const double ngam = fma(s2, ak, d) * ng;
                        const double rjv = fma(s2, ngam, ak);              // row j of R
                        col[s][0] = ln == j ? rjv : fma(ngam, v[0], col[s][0]);
#pragma unroll
                        for (int r = 1; r < NRB; ++r) col[s][r] = fma(ngam, v[r], col[s][r]);
                    }
                }
            }
        }
        };
        if (NR >= 3 && mr <= 64) reflectors(std::integral_constant<int, 1>());
        else if (NR >= 3 && mr <= 128) reflectors(std::integral_constant<int, 2>());
        else reflectors(std::integral_constant<int, NR>());
        BB_QTICK(1);
        // 3. packed block back to LDS and to the panel
        __syncthreads();
#pragma unroll
        for (int s = 0; s < CPW; ++s) {
            const int c = wv + 16 * s;
#pragma unroll
            for (int r = 0; r < NR; ++r) { const int i = r * 64 + ln; if (i < mr && c < ob) blk[c * ld + i] = col[s][r]; }
        }
        __syncthreads();
        // (the same pass leaves V = the unit-lower view of the block in LDS for step 4: an entry is read, stored and masked by one thread)
        for (int e = tid; e < mr * OB; e += BC_THREADS) {
            const int i = e / OB, l = e - i * OB;
            const double bv = blk[l * ld + i];
            if (l < ob) W[(int64_t)(jb + i) * n + jb + l] = bv;
            if (i == l) blk[l * ld + i] = 1.0; else if (i < l) blk[l * ld + i] = 0.0;
        }
        const int c_first = jb + OB, nt = n - c_first;     // columns to the right
        if (nt <= 0) { __syncthreads(); if (pipe) bb_pipe_publish(pipe->my_done, jb + ob, pipe->my_own); BB_QTICK(8); continue; }
        if (pipe) bb_pipe_publish(pipe->my_own, jb + ob);      // (begins with the barrier this point needs anyway)
        else __syncthreads();
        BB_QTICK(8);
        // 4. T of the block from V (unit-lower, in LDS since the pass above)
        BB_QTICK(9);
        {   // G = V^T V above the diagonal (v_mfma_f64_16x16x4_f64, the rows split over the waves), into tb
            constexpr int NT = MT * (MT + 1) / 2, PG = OB == 32 ? 5 : 8;      // tiles of G, row parts: NT * PG <= 16 waves
            const int tile = wv % NT, part = wv / NT;
            const int ta = tile == 2 ? 1 : 0, tbn = tile == 0 ? 0 : 1;       // (0,0) (0,1) (1,1)
            const int kq = ln >> 4, l15 = ln & 15;
            if (wv < NT * PG) {
                const int K = (mr + 3) >> 2, k0 = part * K / PG, k1 = (part + 1) * K / PG;
                bb_d4 g = bb_d4{0.0, 0.0, 0.0, 0.0};
                for (int k = k0; k < k1; ++k) {
                    const int row = 4 * k + kq;
                    const double av = row < mr ? blk[(16 * ta + l15) * ld + row] : 0.0;
                    const double bv = row < mr ? blk[(16 * tbn + l15) * ld + row] : 0.0;
                    g = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, g, 0, 0, 0);
                }
#pragma unroll
                for (int z = 0; z < 4; ++z) red[((tile * PG + part) * 4 + z) * 64 + ln] = g[z];
            }
            __syncthreads();
            for (int e = tid; e < NT * 256; e += BC_THREADS) {
                const int t = e >> 8, z = (e >> 6) & 3, l = e & 63;
                double sum = 0.0;
                for (int pp = 0; pp < PG; ++pp) sum += red[((t * PG + pp) * 4 + z) * 64 + l];
                const int a = 16 * (t == 2 ? 1 : 0) + (l >> 4) + 4 * z, b = 16 * (t == 0 ? 0 : 1) + (l & 15);
                if (a < b) { tb[a * OB + b] = sum; tb[b * OB + a] = sum; }     // (the mirror: read below as rows without bank conflicts)
            }
        }
        __syncthreads();
        BB_QTICK(10);
        {
            // T = (diag(1 / tau) + strict upper part of G)^-1 by back substitution on e_c, a column at a time per wave (two per wave for
            // OB = 32): lane i < OB keeps row i of G in registers (zero up to the diagonal) and y_i; step k (c .. 1): the wave reads
            // x_k = y_k tau_k from lane k and y_i <- y_i - G(i, k) x_k (lanes i >= k hold a zero there and keep what they have); column c
            // of T is y tau.  A reflector with tau = 0 drops out by itself.  Four instructions per step, c steps per column.
            // (Round 5: the column recurrence on 8 x 8 diagonal blocks + two merge levels took 20 000 cycles per block -- ten barriers
            //  and dependent loops over LDS; a first form of this one with two columns per wave side by side and every step run for
            //  every column 13 000; this one see profiles/r05_banded_chain_T.txt.)
            const bool rowact = ln < OB;
            const int li = rowact ? ln : 0;
            double srow[OB];                             // G(li, k), k > li, from the mirror below the diagonal: consecutive lanes, consecutive words
#pragma unroll
            for (int k = 0; k < OB; ++k) { const double gv = tb[k * OB + li]; srow[k] = (rowact && k > li) ? gv : 0.0; }
            const double tau_own = rowact ? tb[li * OB + li] : 0.0;
            __syncthreads();                             // every lane has its row of G before T overwrites it
            for (int c = wv; c < OB; c += 16) {
                double y = ln == c ? 1.0 : 0.0;
#pragma unroll
                for (int k = OB - 1; k >= 1; --k) {
                    if (k <= c) {                        // (wave-uniform)
                        const double xk = readlane_f64(y * tau_own, k);
                        y = fma(-srow[k], xk, y);
                    }
                }
                if (rowact) tb[li * OB + c] = y * tau_own;       // (zero below the diagonal: the mirror is gone with this)
            }
        }
        __syncthreads();
        BB_QTICK(2);
        // 5. W(jb:, c_first:) <- (I - V T^T V^T) W(jb:, c_first:): strips of 16 columns, the rows split over the waves
        //    that are left
        if (pipe && pipe->prev && pipe_need > cp_full) {
            // the rest of the carry rows of step 0: their entries right of this block, final once the previous panel's update is
#ifdef QRK_BB_PROF
            const unsigned long long tq0 = __builtin_amdgcn_s_memtime();
#endif
            if (!bb_pipe_wait(pipe, pipe->prev_done, pipe->lo_from + pipe_need)) return;
#ifdef QRK_BB_PROF
            qt[13] += __builtin_amdgcn_s_memtime() - tq0;
#endif
            const int c0 = cp_full, lc = pipe->lo_cols, cb = jb + ob;
            if (cb < lc) {
                const int wd = lc - cb;
                for (int e = tid; e < (pipe_need - c0) * wd; e += BC_THREADS) {
                    const int i = c0 + e / wd, j = cb + e % wd;
                    W[(int64_t)(i * pipe->lo_stride) * n + j] =
                        i <= j ? pipe->prev[(int64_t)(pipe->lo_from + i) * pipe->prev_n + pipe->lo_from + j] : 0.0;
                }
            }
            __syncthreads();                     // (the copied entries before the update reads them)
            cp_full = pipe_need;
        }
        const int S_all = (nt + 15) >> 4;
        // A SHORT block (the first blocks of a staircase panel: 32 .. 96 rows, the ones the next panel of the pipelined chain waits for)
        // gives every strip of 16 columns to one wave, all rows: up to 16 strips at once, no partial sums to meet in LDS, no barrier
        // per row part.  Taller blocks split the rows of a strip over the waves that are left, at most 8 strips at a time (LDS for
        // their partial sums).
        const bool whole = mr <= QRK_BB_P1_ROWS;
        const int GS = whole ? 16 : 8;
        for (int g0 = 0; g0 < S_all; g0 += GS) {
            const int S = (S_all - g0) < GS ? (S_all - g0) : GS;
            int P = whole ? 1 : 16 / S; if (P > QRK_BB_PMAX) P = QRK_BB_PMAX;
            const bool act = wv < S * P;
            const int strip = act ? wv % S : 0, part = act ? wv / S : 0;
            const int colg = c_first + 16 * (g0 + strip) + (ln & 15);
            const bool cok = act && colg < n;
            const int kq = ln >> 4, l15 = ln & 15;
            double* wcol = W + (int64_t)jb * n + colg;                // wcol[i * n] = W(jb + i, colg)
            bb_d4 acc[MT];
#pragma unroll
            for (int t = 0; t < MT; ++t) acc[t] = bb_d4{0.0, 0.0, 0.0, 0.0};
            if (act) {
                const int K = (mr + 3) >> 2;
                const int k0 = part * K / P, k1 = (part + 1) * K / P;
                constexpr int U = 16;               // W loads in flight per wave: the latency of this CU's path to L2
                for (int k = k0; k < k1; k += U) {
                    double bv[U];
#pragma unroll
                    for (int u = 0; u < U; ++u) {
                        const int row = 4 * (k + u) + kq;
                        bv[u] = (QRK_BB_ABL & 2) ? (double)row : ((k + u < k1 && row < mr && cok) ? wcol[(int64_t)row * n] : 0.0);
                    }
#pragma unroll
                    for (int u = 0; u < U; ++u) {
                        if (k + u < k1) {
                            int row = 4 * (k + u) + kq; if (row > mr - 1) row = mr - 1;
#pragma unroll
                            for (int t = 0; t < MT; ++t)
                                if (QRK_BB_ABL & 1) acc[t][0] += bv[u]; else
                                acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(blk[(16 * t + l15) * ld + row], bv[u], acc[t], 0, 0, 0);
                        }
                    }
                }
            }
            BB_QTICK(4);
            if (P > 1) {                       // partial sums of the row parts, added in a fixed order
                for (int pp = 0; pp < P; ++pp) {
                    if (act && part == pp) {
#pragma unroll
                        for (int t = 0; t < MT; ++t)
#pragma unroll
                            for (int z = 0; z < 4; ++z) {
                                const int idx = ((strip * MT + t) * 4 + z) * 64 + ln;
                                red[idx] = pp == 0 ? acc[t][z] : red[idx] + acc[t][z];
                            }
                    }
                    __syncthreads();
                }
                if (act) {
#pragma unroll
                    for (int t = 0; t < MT; ++t)
#pragma unroll
                        for (int z = 0; z < 4; ++z) acc[t][z] = red[((strip * MT + t) * 4 + z) * 64 + ln];
                }
            }
            BB_QTICK(5);
            if (act) {
                // u = -T^T w: the result layout D[row = (lane >> 4) + 4 z][col = lane & 15] of tile t is the B operand
                // [k = lane >> 4][col] of the k-step 4 t + z
                bb_d4 uu[MT];
#pragma unroll
                for (int t = 0; t < MT; ++t) {
                    uu[t] = bb_d4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
                    for (int ks = 0; ks < 4 * MT; ++ks)
                        uu[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(tb[(4 * ks + kq) * OB + 16 * t + l15], acc[ks >> 2][ks & 3], uu[t], 0, 0, 0);
                    uu[t] = -uu[t];
                }
                BB_QTICK(6);
                const int RT = (mr + 15) >> 4;
                const int t0 = part * RT / P, t1 = (part + 1) * RT / P;
                constexpr int UT = 4;
                for (int rt = t0; rt < t1; rt += UT) {
                    bb_d4 dv[UT];
#pragma unroll
                    for (int u = 0; u < UT; ++u)
#pragma unroll
                        for (int z = 0; z < 4; ++z) {
                            const int row = 16 * (rt + u) + kq + 4 * z;
                            dv[u][z] = (QRK_BB_ABL & 2) ? (double)row : ((rt + u < t1 && row < mr && cok) ? wcol[(int64_t)row * n] : 0.0);
                        }
#pragma unroll
                    for (int u = 0; u < UT; ++u) {
                        if (rt + u < t1) {
                            int arow = 16 * (rt + u) + l15; if (arow > mr - 1) arow = mr - 1;
#pragma unroll
                            for (int ks = 0; ks < 4 * MT; ++ks)
                                if (QRK_BB_ABL & 1) dv[u][0] += uu[ks >> 2][ks & 3]; else
                                dv[u] = __builtin_amdgcn_mfma_f64_16x16x4f64(blk[(4 * ks + kq) * ld + arow], uu[ks >> 2][ks & 3], dv[u], 0, 0, 0);
#pragma unroll
                            for (int z = 0; z < 4; ++z) {
                                const int row = 16 * (rt + u) + kq + 4 * z;
                                if (row < mr && cok && !((QRK_BB_ABL & 4) && dv[u][z] != 12345.678)) wcol[(int64_t)row * n] = dv[u][z];
                            }
                        }
                    }
                }
            }
            BB_QTICK(7);
            __syncthreads();
            BB_QTICK(3);
        }
#ifdef QRK_BB_PROF
        { const unsigned long long tp0 = __builtin_amdgcn_s_memtime(); if (pipe) bb_pipe_publish(pipe->my_done, jb + ob, pipe->my_own); qt[12] += __builtin_amdgcn_s_memtime() - tp0; q0 = __builtin_amdgcn_s_memtime(); }
#else
        if (pipe) bb_pipe_publish(pipe->my_done, jb + ob, pipe->my_own);
#endif
    }
#undef BB_QTICK
}

// Dense inputs of all the panels (Ji = pmat.block(row0, col0, m, n).toDense(), :458/:503) into the panels' own
// storage in y_vals, row-major m x n: one workgroup per panel, before the chain.  The rows of a panel are one
// contiguous run of CSR entries; every thread walks it with stride BC_THREADS (coalesced index loads, eight entries in
// flight) and finds the row of its entry by stepping through the row pointers kept in LDS.  (The top-left corner is
// replaced by the previous panel's leftover block inside the chain.)
__global__ void __launch_bounds__(BC_THREADS)
bb_scatter_kernel(const BBPanel* __restrict__ panels, const int32_t* __restrict__ prowptr, const int32_t* __restrict__ pcol,
                  const int64_t* __restrict__ pmap, const double* __restrict__ vals, double* __restrict__ y_vals)
{
    extern __shared__ __attribute__((aligned(16))) double smem[];
    int* ptrs = reinterpret_cast<int*>(smem);           // [m + 1]
    const int tid = threadIdx.x;
    const BBPanel p = panels[blockIdx.x];
    const int m = p.act_rows, n = p.ncols;
    double* W = y_vals + p.y_off;
    for (int i = tid; i <= m; i += BC_THREADS) ptrs[i] = prowptr[p.row0 + i];
    for (int64_t e = tid; e < (int64_t)m * n; e += BC_THREADS) W[e] = 0.0;
    __syncthreads();
    const int e0 = ptrs[0], e1 = ptrs[m];
    int r = 0;
    constexpr int U = 8;
    for (int e = e0 + tid; e < e1; e += U * BC_THREADS) {
        // (indices clamped instead of predicated loads: every value is defined on every path)
        int cc[U]; int64_t pm[U]; double vv[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            int ee = e + u * BC_THREADS; if (ee > e1 - 1) ee = e1 - 1;
            cc[u] = pcol[ee]; pm[u] = pmap[ee];
        }
#pragma unroll
        for (int u = 0; u < U; ++u) vv[u] = vals[pm[u]];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int ee = e + u * BC_THREADS;
            if (ee < e1) {
                while (ee >= ptrs[r + 1]) ++r;
                const int c = cc[u] - p.col0;
                if (c >= 0 && c < n) W[(int64_t)r * n + c] = vv[u];
            }
        }
    }
}

__global__ void __launch_bounds__(BC_THREADS)
bb_chain2_kernel(const BBPanel* __restrict__ panels, int num_panels, const int32_t* __restrict__ prowptr,
                 const int32_t* __restrict__ pcol, const int64_t* __restrict__ pmap, const double* __restrict__ vals,
                 double* __restrict__ lo, double* __restrict__ y_vals,
                 double* __restrict__ t_vals, double* __restrict__ r_stage, int max_act_rows, int max_ncols,
                 int uni_doubles, const int* __restrict__ rlim_first, const int* __restrict__ rlim_rest, int* __restrict__ done,
                 unsigned spin_limit)
{
    extern __shared__ __attribute__((aligned(16))) double smem[];
    double* hc = smem;                         // [BC_CW] hCoeffs of the panel
    double* sc = hc + BC_CW;                   // [8] scalars of the current reflector (two sets)
    double* uni = sc + 8;                      // the blocked QR's (bb_panel_qr); a tile of R rows on the way out
    const int tid = threadIdx.x;
    // done != null: the pipelined chain (see BBPipe): this workgroup takes the panels blockIdx.x, blockIdx.x + gridDim.x, ..
    // (its record lives in the last 96 bytes of the dynamic LDS: the kernel already asks for all 160 KB)
    BBPipe& s_pipe = *reinterpret_cast<BBPipe*>(uni + uni_doubles);
    static_assert(sizeof(BBPipe) <= 96, "BBPipe fits its LDS slot (bb_chain2_smem)");
    const bool piped = done != nullptr;
#ifdef QRK_BB_PROF
    unsigned long long pt[6] = {0, 0, 0, 0, 0, 0}, t0 = 0, qt[14] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#define BB_TICK(n) do { const unsigned long long t1 = __builtin_amdgcn_s_memtime(); pt[n] += t1 - t0; t0 = t1; } while (0)
#else
#define BB_TICK(n) do { } while (0)
#endif

#if QRK_BB_SAME_XCD
    // The workgroups of the pipelined chain on ONE XCD (round 5): workgroup i of a launch goes to XCD i mod 8, so the launch has 8 (G - 1) + 1
    // workgroups and only every eighth works -- the panels' rows then pass from one workgroup to the next through the XCD's L2 instead of
    // memory: 0.1174 -> 0.1143 ms per strip.  (Were the dispatch order different, the result would be the same and the time the old one.)
    if (piped && (blockIdx.x & 7) != 0) return;
    const int wg_id = piped ? (int)blockIdx.x >> 3 : 0, wg_n = piped ? ((int)gridDim.x + 7) >> 3 : 1;
#else
    const int wg_id = piped ? (int)blockIdx.x : 0, wg_n = piped ? (int)gridDim.x : 1;
#endif
    for (int pi = wg_id; pi < num_panels; pi += wg_n) {
        const BBPanel p = panels[pi];
        const int m = p.act_rows, n = p.ncols;          // W is m x n, row-major: W(i, j) = W[i * n + j]
#ifdef QRK_BB_PROF
        t0 = __builtin_amdgcn_s_memtime();
#endif
        // ---- Ji = pmat.block(row0, col0, m, n).toDense() (:458, :503) was written by bb_scatter_kernel into the panel's
        // own storage (the factorisation works in place there: what is left below the diagonal is Y) ...
        double* W = y_vals + p.y_off;
        // ... with its top-left corner replaced by the leftover block of the previous panel (:504-506)
        if (!piped) {
            for (int e = tid; e < p.lo_rows * p.lo_cols; e += BC_THREADS) {
                const int i = e / p.lo_cols, j = e - i * p.lo_cols;
                W[(int64_t)(i * p.lo_stride) * n + j] = lo[e];
            }
        } else if (tid == 0) {
            // (pipelined: bb_panel_qr takes the carry rows over block by block, from the previous panel's own storage)
            const bool has = pi > 0 && p.lo_rows > 0;
            s_pipe.prev = has ? y_vals + panels[pi - 1].y_off : nullptr;
            s_pipe.prev_done = has ? done + pi - 1 : nullptr;
            s_pipe.my_done = done + pi;
            s_pipe.prev_own = has ? done + num_panels + 1 + pi - 1 : nullptr;
            s_pipe.my_own = done + num_panels + 1 + pi;
            s_pipe.abortw = done + num_panels;
            s_pipe.spin_limit = spin_limit;
            if (pi == wg_id) s_pipe.aborted = 0;
            s_pipe.prev_n = has ? panels[pi - 1].ncols : 0;
            s_pipe.lo_from = p.lo_from; s_pipe.lo_rows = p.lo_rows; s_pipe.lo_cols = p.lo_cols; s_pipe.lo_stride = p.lo_stride;
        }
        __syncthreads();
        const int* rlim = pi == 0 ? rlim_first : rlim_rest;
        BBPipe* pipe = piped ? &s_pipe : nullptr;

        BB_TICK(0);
        // ---- Eigen::HouseholderQR of the panel (:459-470): blocks of 32 columns when the block fits the LDS
        // next to its T and the partial sums (m <= ~470), else of 16
        // tallest block of this panel: all of it without a staircase, else the largest rlim[g] - (first row of its block)
        int tall = m;
        if (rlim) {
            tall = 0;
            for (int jb = 0; jb < n; jb += 32) {
                const int ob = (n - jb) < 32 ? (n - jb) : 32;
                int lim = rlim[(jb + ob - 1) >> 4]; lim = lim < m ? lim : m;
                tall = (lim - jb) > tall ? (lim - jb) : tall;
            }
        }
#if QRK_BB_PIPE_OB16
        // The pipelined chain works in blocks of 16 columns (round 5): a panel publishes its final rows block by block and the next panel's
        // block b waits for the rows of column step + 16 (b + 1) columns, so the grain of the blocks is the grain of the overlap -- with
        // 32-column blocks a panel starts when the one before has done 96 of its 192 columns, with 16-column blocks after 80:
        // 0.1654 -> 0.1336 ms per strip at the BASELINE configs[2] shape although a panel ALONE is slower in blocks of 16 (twice the
        // block updates).  One workgroup alone (no partner to wait for) keeps the 32-column blocks.
        if (pipe) {
#ifdef QRK_BB_PROF
            if (tall <= 192) bb_panel_qr<16, 3>(W, m, n, rlim, pipe, qt); else if (tall <= 256) bb_panel_qr<16, 4>(W, m, n, rlim, pipe, qt); else bb_panel_qr<16>(W, m, n, rlim, pipe, qt);
#else
            if (tall <= 192) bb_panel_qr<16, 3>(W, m, n, rlim, pipe); else if (tall <= 256) bb_panel_qr<16, 4>(W, m, n, rlim, pipe); else bb_panel_qr<16>(W, m, n, rlim, pipe);
#endif
        } else
#endif
        if ((int64_t)32 * (m | 1) + bb_qr_aux_doubles(32) <= (int64_t)uni_doubles) {
#ifdef QRK_BB_PROF
            if (tall <= 192) bb_panel_qr<32, 3>(W, m, n, rlim, pipe, qt); else if (tall <= 256) bb_panel_qr<32, 4>(W, m, n, rlim, pipe, qt); else bb_panel_qr<32>(W, m, n, rlim, pipe, qt);
#else
            if (tall <= 192) bb_panel_qr<32, 3>(W, m, n, rlim, pipe); else if (tall <= 256) bb_panel_qr<32, 4>(W, m, n, rlim, pipe); else bb_panel_qr<32>(W, m, n, rlim, pipe);
#endif
        } else {
#ifdef QRK_BB_PROF
            bb_panel_qr<16>(W, m, n, rlim, pipe, qt);
#else
            bb_panel_qr<16>(W, m, n, rlim, pipe);
#endif
        }
        __syncthreads();
        if (piped && s_pipe.aborted) break;            // (uniform: the chain was given up -- bb_pipe_wait; the host runs it again on one workgroup)

        BB_TICK(1);
        // ---- rows of R solved by this panel: V = triu(packed QR), explicit zeros kept (:484-491).  r_stage is
        // column-major (solved x n): the rows go through an LDS tile so that both sides are coalesced.
        for (int b0 = 0; b0 < p.solved; b0 += 64) {
            const int nr = (p.solved - b0) < 64 ? (p.solved - b0) : 64, tld = n | 1;
            __syncthreads();
            for (int e = tid; e < nr * n; e += BC_THREADS) {
                const int il = e / n, bc = e - il * n, br = b0 + il;
                uni[il * tld + bc] = (br <= bc && br < m) ? W[(int64_t)br * n + bc] : 0.0;
            }
            __syncthreads();
            for (int e = tid; e < nr * n; e += BC_THREADS) {
                const int bc = e / nr, il = e - bc * nr;
                r_stage[p.r_off + (int64_t)bc * p.solved + b0 + il] = uni[il * tld + bc];
            }
        }
        // ---- leftover block for the next panel: V.block(lo_from, lo_from, lo_rows, lo_cols) (:505), row-major
        if (pi + 1 < num_panels && !piped) {
            const BBPanel q = panels[pi + 1];
            for (int e = tid; e < q.lo_rows * q.lo_cols; e += BC_THREADS) {
                const int i = e / q.lo_cols, j = e - i * q.lo_cols;
                const int vr = q.lo_from + i, vc = q.lo_from + j;
                lo[e] = (vr <= vc && vr < m && vc < n) ? W[(int64_t)vr * n + vc] : 0.0;
            }
        }
        // (Y = unit-lower essentials (:471-475) is the part of the panel below the diagonal: nothing to write)
        __syncthreads();

        BB_TICK(2);
        // ---- hCoeffs parked on the diagonal of the T output: bb_t_kernel (one workgroup per panel, after the chain)
        // builds T from Y and these.  T is not needed by the next panel, so it is not the chain's work.
        {
            double* T = t_vals + p.t_off;
            for (int c = tid; c < n; c += BC_THREADS) T[(int64_t)c * n + c] = hc[c];
        }
        __syncthreads();
#ifdef QRK_BB_PROF
        if (pi == num_panels - 1 && tid == 0) {
            double* T = t_vals + p.t_off;
            for (int z = 0; z < 6; ++z) T[z] = (double)pt[z];
            for (int z = 0; z < 14; ++z) T[6 + z] = (double)qt[z];
        }
#endif
    }
}

// T of every panel from its Y and hCoeffs (parked on T's diagonal by the chain): one workgroup per panel.
__global__ void __launch_bounds__(BC_THREADS)
bb_t_kernel(const BBPanel* __restrict__ panels, int num_panels, const double* __restrict__ y_vals,
            double* __restrict__ t_vals, int t_in_lds)
{
    extern __shared__ __attribute__((aligned(16))) double smem[];
    double* hc = smem;                         // [BC_CW] hCoeffs of the panel
    double* uni = hc + BC_CW;
    double* dpart = uni;                       // [BC_THREADS] partial sums of the T recurrence in global memory
    double* ys = uni;                          // [BC_RC * (n16 + 1)] rows of Y
    double* tl = uni;                          // [n (n + 1) / 2] packed upper T by columns
    const int tid = threadIdx.x;
    const int pi = blockIdx.x;
#ifdef QRK_BB_PROF
    if (pi == num_panels - 1) return;          // (the chain left its tick counts in this panel's T)
#endif
    const BBPanel p = panels[pi];
    const int m = p.act_rows, n = p.ncols;
    const int CW = ((n + 63) / 64) * 64, RG = BC_THREADS / CW;
    const bool on = tid < CW * RG;
    const int cs = on ? tid % CW : CW - 1, rg = on ? tid / CW : 0;
    const double* Y = y_vals + p.y_off;
    double* T = t_vals + p.t_off;
    for (int c = tid; c < n; c += BC_THREADS) hc[c] = T[(int64_t)c * n + c];
    __syncthreads();
    // ---- G = Y^T Y (strict upper part) into the T output: T[c * n + b] = G(b, c), b < c.
    // A real GEMM (n x m by m x n): v_mfma_f64_16x16x4_f64 on 16x16 tiles of G, the operands read from an
    // LDS chunk of 16 rows of Y (unit-lower view of the packed panel); every wave owns up to 9 upper tiles.
    // Lane maps (cdna_hip_programming.md): A[row = l & 15][k = l >> 4], B[k = l >> 4][col = l & 15],
    // D[row = (l >> 4) + 4 j][col = l & 15] in result register j.
    {
        typedef double d4 __attribute__((ext_vector_type(4)));
        const int nbt = (n + 15) / 16, n16 = nbt * 16, ysld = n16 + 1;
        const int ntile = nbt * (nbt + 1) / 2;
        const int wv = tid >> 6, ln = tid & 63;
        constexpr int MAXQ = 9;                      // 16 * 17 / 2 = 136 tiles over 16 waves
        d4 acc[MAXQ];
        int tbi[MAXQ], tbc[MAXQ];
#pragma unroll
        for (int q = 0; q < MAXQ; ++q) {
            acc[q] = d4{0.0, 0.0, 0.0, 0.0};
            int t = wv + q * (BC_THREADS / 64), bi = 0;
            if (t >= ntile) { tbi[q] = -1; tbc[q] = 0; continue; }
            while (t >= nbt - bi) { t -= nbt - bi; ++bi; }   // upper tiles enumerated row by row
            tbi[q] = bi; tbc[q] = bi + t;
        }
        for (int r0 = 0; r0 < m; r0 += BC_RC) {
            __syncthreads();
            for (int e = tid; e < BC_RC * n16; e += BC_THREADS) {
                const int il = e / n16, j = e - il * n16, i = r0 + il;      // unit-lower view of the packed panel
                ys[il * ysld + j] = (i >= m || j >= n || i < j) ? 0.0 : (i == j ? 1.0 : Y[(int64_t)i * n + j]);
            }
            __syncthreads();
#pragma unroll
            for (int q = 0; q < MAXQ; ++q) {
                if (tbi[q] >= 0) {
#pragma unroll
                    for (int ks = 0; ks < BC_RC / 4; ++ks) {
                        const int row = 4 * ks + (ln >> 4);
                        const double av = ys[row * ysld + 16 * tbi[q] + (ln & 15)];
                        const double bv = ys[row * ysld + 16 * tbc[q] + (ln & 15)];
                        acc[q] = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, acc[q], 0, 0, 0);
                    }
                }
            }
        }
#pragma unroll
        for (int q = 0; q < MAXQ; ++q) {
            if (tbi[q] >= 0) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int gi = 16 * tbi[q] + (ln >> 4) + 4 * j, gc = 16 * tbc[q] + (ln & 15);
                    if (gi < gc && gc < n) T[(int64_t)gc * n + gi] = acc[q][j];
                }
            }
        }
    }
    __syncthreads();

    // ---- T = make_block_householder_triangular_factor(Y, hCoeffs) (:476).
    // With T in LDS: the recursive form of the same factor.  T = [T11 T12; 0 T22] with
    // T12 = -T11 (Y1^T Y2) T22, so the BC_THREADS/64 diagonal blocks are built by the column recurrence, one
    // wave each and without workgroup barriers, and merged pairwise in log2(16) rounds of two triangular
    // products (every thread a few entries).  The column recurrence over the whole panel spent 2.7 us per
    // column in barriers and LDS latency (0.5 ms of a 1.9 ms panel).
    if (t_in_lds) {
        const int wv = tid >> 6, ln = tid & 63;
        constexpr int NBLK = BC_THREADS / 64;
        const int s0 = (n + NBLK - 1) / NBLK;          // <= 16 for n <= 256
        // packed upper storage, by columns: (a, b), a <= b, at b (b + 1) / 2 + a.  G above the diagonal, tau on it
        for (int b = wv; b < n; b += NBLK) {
            const int cb = b * (b + 1) / 2;
            for (int a = ln; a < b; a += 64) tl[cb + a] = T[(int64_t)b * n + a];
            if (ln == 0) tl[cb + b] = hc[b];
        }
        __syncthreads();
        {   // diagonal block wv: t(a, c) = -tau_c sum_{b = a}^{c - 1} T(a, b) G(b, c) inside the block
            const int base = wv * s0, len = (n - base) < s0 ? (n - base) : s0;
            for (int j = 1; j < len; ++j) {
                const int c = base + j, cc2 = c * (c + 1) / 2;
                double sum = 0.0;
                if (ln < j) {
                    const int a = base + ln;
                    int ib = (a * (a + 1)) / 2 + a;          // (a, b = a)
                    for (int b = a; b < c; ++b) { sum = fma(tl[ib], tl[cc2 + b], sum); ib += b + 1; }
                }
                const double tc = tl[cc2 + c];
                __builtin_amdgcn_wave_barrier();
                if (ln < j) tl[cc2 + base + ln] = -tc * sum;
                __builtin_amdgcn_wave_barrier();
            }
        }
        __syncthreads();
        constexpr int MAXO = (BC_CW / 2) * (BC_CW / 2) / BC_THREADS;      // entries per thread in the last merge
        for (int sz = s0; sz < n; sz *= 2) {
            const int ss = sz * sz, npair = (n + 2 * sz - 1) / (2 * sz), total = npair * ss;
            double xr[MAXO];
            // X = G12 T22 (in place of G12): X(i, j) = sum_{k <= j} G12(i, k) T22(k, j)
#pragma unroll
            for (int t = 0; t < MAXO; ++t) {
                const int o = tid + t * BC_THREADS;
                xr[t] = 0.0;
                if (o < total) {
                    const int q = o / ss, rem = o - q * ss, j = rem / sz, i = rem - j * sz;
                    const int r0 = 2 * q * sz, c0 = r0 + sz, cj = c0 + j;
                    if (cj < n) {
                        const int cj2 = cj * (cj + 1) / 2 + c0;
                        int ia = c0 * (c0 + 1) / 2 + r0 + i;
                        double acc = 0.0;
#pragma unroll 4
                        for (int k = 0; k <= j; ++k) { acc = fma(tl[ia], tl[cj2 + k], acc); ia += c0 + k + 1; }
                        xr[t] = acc;
                    }
                }
            }
            __syncthreads();
#pragma unroll
            for (int t = 0; t < MAXO; ++t) {
                const int o = tid + t * BC_THREADS;
                if (o < total) {
                    const int q = o / ss, rem = o - q * ss, j = rem / sz, i = rem - j * sz;
                    const int r0 = 2 * q * sz, cj = r0 + sz + j;
                    if (cj < n) tl[cj * (cj + 1) / 2 + r0 + i] = xr[t];
                }
            }
            __syncthreads();
            // T12 = -T11 X: T12(i, j) = -sum_{k >= i} T11(i, k) X(k, j)
#pragma unroll
            for (int t = 0; t < MAXO; ++t) {
                const int o = tid + t * BC_THREADS;
                xr[t] = 0.0;
                if (o < total) {
                    const int q = o / ss, rem = o - q * ss, j = rem / sz, i = rem - j * sz;
                    const int r0 = 2 * q * sz, cj = r0 + sz + j;
                    if (cj < n) {
                        const int cj2 = cj * (cj + 1) / 2 + r0;
                        const int ri = r0 + i;
                        int ia = ri * (ri + 1) / 2 + ri;     // (ri, ri)
                        double acc = 0.0;
#pragma unroll 4
                        for (int k = i; k < sz; ++k) { acc = fma(tl[ia], tl[cj2 + k], acc); ia += r0 + k + 1; }
                        xr[t] = -acc;
                    }
                }
            }
            __syncthreads();
#pragma unroll
            for (int t = 0; t < MAXO; ++t) {
                const int o = tid + t * BC_THREADS;
                if (o < total) {
                    const int q = o / ss, rem = o - q * ss, j = rem / sz, i = rem - j * sz;
                    const int r0 = 2 * q * sz, cj = r0 + sz + j;
                    if (cj < n) tl[cj * (cj + 1) / 2 + r0 + i] = xr[t];
                }
            }
            __syncthreads();
        }
    } else {
        // in place in global memory, forward recurrence by columns (panels whose packed T does not fit the LDS)
        for (int cc = 0; cc < n; ++cc) {
            const double hcc = hc[cc];
            double part = 0.0;
            if (on && cs < cc) for (int b = cs + rg; b < cc; b += RG) part = fma(T[(int64_t)b * n + cs], T[(int64_t)cc * n + b], part);
            if (on) dpart[rg * CW + cs] = part;
            __syncthreads();     // every G(b, cc) has been read before column cc is overwritten
            if (on && rg == 0 && cs < cc) {
                double sum = 0.0;
                for (int g = 0; g < RG; ++g) sum += dpart[g * CW + cs];
                T[(int64_t)cc * n + cs] = -hcc * sum;
            }
            if (tid == 0) T[(int64_t)cc * n + cc] = hcc;
            __syncthreads();
        }
    }
    // the reference stores -T (:477); lower part zero
    for (int64_t e = tid; e < (int64_t)n * n; e += BC_THREADS) {
        const int a = (int)(e % n), b = (int)(e / n);
        double v = 0.0;
        if (a <= b) v = t_in_lds ? -tl[(int64_t)b * (b + 1) / 2 + a] : -T[e];
        T[e] = v;
    }
    __syncthreads();
}

// LDS of bb_chain2_kernel: all of it (one workgroup per chain), 0 if the 16-column block of the tallest panel does not fit
size_t bb_chain2_smem(int max_act_rows, int* uni_doubles)
{
    const size_t all = (size_t)160 * 1024;
    const size_t fixed = (size_t)(BC_CW + 8 + 12) * sizeof(double);      // hc, sc, and (at the end) the BBPipe record (96 bytes)
    const size_t qr = ((size_t)16 * (max_act_rows | 1) + bb_qr_aux_doubles(16)) * sizeof(double);
    *uni_doubles = 0;
    if (fixed + qr > all || max_act_rows > 1024) return 0;      // (bb_panel_qr<16> keeps 16 x 64 rows of a column per wave)
    *uni_doubles = (int)((all - fixed) / sizeof(double));
    return all;
}

// LDS of bb_t_kernel
size_t bb_t_smem(int max_ncols, int* t_in_lds)
{
    const size_t all = (size_t)160 * 1024;
    const size_t fixed = (size_t)BC_CW * sizeof(double);
    const size_t gram = (size_t)BC_RC * ((max_ncols + 15) / 16 * 16 + 1) * sizeof(double);
    const size_t tpk = (size_t)max_ncols * (max_ncols + 1) / 2 * sizeof(double);
    size_t uni = gram > BC_THREADS * sizeof(double) ? gram : BC_THREADS * sizeof(double);
    *t_in_lds = 0;
    // (QRK_BB_T_GLOBAL forces the in-place T recurrence: lets the tests cover it)
    if (fixed + tpk <= all && !std::getenv("QRK_BB_T_GLOBAL")) { *t_in_lds = 1; if (tpk > uni) uni = tpk; }
    return fixed + uni;
}

// x(0:cols) = R(0:cols, 0:cols).triangularView<Upper>().solve(v(0:cols)), in place: the last step of
// BandedBlockedSparseQR::_solve_impl (src/QRKit/BandedBlockedSparseQR.h:290-311).  R is read from the staging array the
// chain leaves behind (panel p: the dense rows [col0, col0 + solved) x [col0, col0 + ncols), column-major), panels in
// descending order, 64 rows at a time: the columns to the right of the 64 x 64 diagonal block are a matrix-vector
// product with entries of x that are already final (256 threads, rows over the lanes: coalesced), the block itself is
// solved by one wave from LDS (lane = row, the pivot value broadcast with v_readlane).  One workgroup per right-hand
// side; the chain over the panels is sequential (x of a panel needs the x of the panels after it).
constexpr int BS_THREADS = 256;
__global__ void __launch_bounds__(BS_THREADS)
bb_solve_r_kernel(const BBPanel* __restrict__ panels, const BBPanel single, int num_panels, const double* __restrict__ r_stage,
                  int cols, double* __restrict__ v, int64_t ldv)
{
    // (panels == nullptr: one dense upper triangle described by `single` - qrk_dense_solve_r)
    __shared__ double blk[64 * 65];          // diagonal block, blk[j * 65 + i] = R(i, j)
    __shared__ double part[4 * 64];          // partial sums of the four column groups
    const int tid = threadIdx.x, ln = tid & 63, grp = tid >> 6;
    double* x = v + (int64_t)blockIdx.x * ldv;
    for (int pi = num_panels - 1; pi >= 0; --pi) {
        const BBPanel p = panels ? panels[pi] : single;
        const int n = p.ncols, sv = p.solved;
        int ns = sv < n ? sv : n;                        // rows of this panel inside the triangle
        if (p.col0 + ns > cols) ns = cols - p.col0;
        if (ns <= 0) continue;
        const double* R = r_stage + p.r_off;             // R(i, j) = R[j * sv + i]
        for (int c1 = ns; c1 > 0; c1 -= 64) {            // rows [c0, c1) of the panel, last chunk first
            const int c0 = c1 > 64 ? c1 - 64 : 0, nr = c1 - c0;
            __syncthreads();                             // (x of the previous chunk is visible; blk and part are free)
            // diagonal block to LDS
            for (int e = tid; e < nr * nr; e += BS_THREADS) {
                const int j = e / nr, i = e - j * nr;
                blk[j * 65 + i] = R[(int64_t)(c0 + j) * sv + c0 + i];
            }
            // t_i = v_i - sum_{j >= c1} R(i, j) x_j, the columns dealt round-robin to the four waves
            double acc = 0.0;
            if (ln < nr) {
                constexpr int U = 8;
                for (int j = c1 + grp; j < n; j += 4 * U) {
                    double rv[U], xv[U];
#pragma unroll
                    for (int u = 0; u < U; ++u) {
                        const int jj = j + 4 * u, jc = jj < n ? jj : n - 1;
                        rv[u] = R[(int64_t)jc * sv + c0 + ln];
                        xv[u] = jj < n ? x[p.col0 + jc] : 0.0;
                    }
#pragma unroll
                    for (int u = 0; u < U; ++u) acc = fma(rv[u], xv[u], acc);
                }
            }
            part[grp * 64 + ln] = acc;
            __syncthreads();
            if (grp == 0) {
                double t = 0.0;
                if (ln < nr) t = x[p.col0 + c0 + ln] - ((part[ln] + part[64 + ln]) + (part[128 + ln] + part[192 + ln]));
                for (int i = nr - 1; i >= 0; --i) {      // back substitution inside the block
                    const double xi = readlane_f64(t, i) / blk[i * 65 + i];
                    if (ln < i) t = fma(-blk[i * 65 + ln], xi, t);
                    else if (ln == i) t = xi;
                }
                if (ln < nr) x[p.col0 + c0 + ln] = t;
            }
        }
    }
}

__global__ void __launch_bounds__(256)
bb_gather_r_kernel(const double* __restrict__ r_stage, const int64_t* __restrict__ r_src, int64_t nnz,
                   double* __restrict__ r_vals)
{
    const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (i < nnz) r_vals[i] = r_stage[r_src[i]];
}

// SparseBlockYTY_VecProduct (src/QRKit/SparseBlockYTY.h:100-139): seg += Y (T^(T) (Y^T seg)) for every block in order
// (transpose: ascending with T^T; else descending with T).  One workgroup of 1024 threads per right-hand side; the
// chain over the blocks is sequential (consecutive blocks share rows).  Y is the unit-lower view of the factorised panel
// (row-major m x n); every pass over Y and T is coalesced:
//   w1 = Y^T seg   threads as column x row group, partial sums through LDS;
//   w2 = T w1      (i, group) over the columns j >= i of the column-major T;  T^T w1: a wave per row of T^T, lanes over j;
//   seg += Y w2    a wave per row, lanes over the columns left of the diagonal.
// 0.080 ms per 448 x 192 block (256 threads with strided passes: 0.120): ~18 dependent round trips to HBM per block.
// Tried without gain: eight rows per wave with all their loads in flight (0.087), touching the next block's lines
// ahead of time (0.093: vmcnt retires in order, so the first real load waits for the prefetches).
constexpr int BA_THREADS = 1024;
constexpr int BA_WAVES = BA_THREADS / 64;
// Strips form (BBStrips, launch_bbs_apply): the chain of stage B of the two-stage banded factorisation.  Panel i works on the
// stack [carry of panel i-1 interleaved with the first lo rows of R_i; the other rows of R_i]; its vector is gathered from the
// carry (LDS) and from the strip's stage-A vector ya_i, and leaves as: solved rows of the R part, the carry of the next panel
// and lo residual components.  Layout of the full vector (rows = N m_s): [R part, cols] then per strip [lo chain residuals
// (strips 1..) | m_s - n stage-A residuals].
struct BBStrips {
    double* ya;          // [nrhs][N][ms]: stage-A vectors Q_i^T b_i (transpose: in; else: out)
    double* full;        // [nrhs][rows]: the vector in the layout above (transpose: out; else: in)
    int64_t ya_ld, full_ld;
    int ms, n, s, lo, cols;
};
__device__ __forceinline__ int64_t bbs_res_off(const BBStrips& S, int i)
{
    return (int64_t)S.cols + (int64_t)i * (S.ms - S.n) + (i >= 1 ? (int64_t)(i - 1) * S.lo : 0);
}

__global__ void __launch_bounds__(BA_THREADS)
bb_apply_q_kernel(const BBPanel* __restrict__ panels, int num_panels, const double* __restrict__ y_vals,
                  const double* __restrict__ t_vals, int transpose, double* __restrict__ v, int64_t ldv, int64_t nrhs,
                  int max_act_rows, int max_ncols, const BBStrips S)
{
    extern __shared__ __attribute__((aligned(16))) double smem[];
    double* seg = smem;                      // [max_act_rows]
    double* w1 = seg + max_act_rows;         // [max_ncols]
    double* w2 = w1 + max_ncols;             // [max_ncols]
    double* part = w2 + max_ncols;           // [BA_THREADS]
    double* carry = part + BA_THREADS;       // [lo] (strips form)
    const bool strips = S.ya != nullptr;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int64_t col = blockIdx.x; col < nrhs; col += gridDim.x) {
        double* x = strips ? nullptr : v + col * ldv;
        double* ya = strips ? S.ya + col * S.ya_ld : nullptr;
        double* full = strips ? S.full + col * S.full_ld : nullptr;
        for (int s = 0; s < num_panels; ++s) {
            const int pidx = transpose ? s : num_panels - 1 - s;
            const BBPanel p = panels[pidx];
            const int m = p.act_rows, n = p.ncols;
            const int seg2 = p.yrow + n + p.num_zeros;     // start of the second row segment
            const double* Y = y_vals + p.y_off;
            const double* T = t_vals + p.t_off;
            if (!strips) {
                for (int i = tid; i < m; i += BA_THREADS) seg[i] = x[i < n ? p.yrow + i : seg2 + (i - n)];
            } else if (transpose) {
                // stack rows in: panel 0 = the strip's own vector; else even rows 2t = carry, odd rows 2t+1 and the tail = ya_i
                const double* yi = ya + (int64_t)pidx * S.ms;
                for (int i = tid; i < m; i += BA_THREADS) {
                    double val;
                    if (pidx == 0) val = yi[i];
                    else if (i < 2 * S.lo) val = (i & 1) ? yi[i >> 1] : carry[i >> 1];
                    else val = yi[i - S.lo];
                    seg[i] = val;
                }
            } else {
                // stack rows out (the input of Q): solved rows of the R part, the carry handed back by panel i+1, residuals
                const int64_t ro = bbs_res_off(S, pidx);
                for (int i = tid; i < m; i += BA_THREADS) {
                    double val;
                    if (i < p.solved) val = full[(int64_t)S.s * pidx + i];
                    else if (i < n) val = carry[i - p.solved];
                    else val = full[ro + (i - n)];
                    seg[i] = val;
                }
            }
            __syncthreads();
            // ---- w1 = Y^T seg
            for (int j0 = 0; j0 < n; j0 += BA_THREADS) {
                const int nc = (n - j0) < BA_THREADS ? (n - j0) : BA_THREADS;
                const int CW = ((nc + 63) / 64) * 64, RG = BA_THREADS / CW;
                const int jl = tid % CW, g = tid / CW, j = j0 + jl;
                double acc = 0.0;
                if (g < RG && jl < nc) {
                    constexpr int U = 8;           // loads in flight per thread: the panel streams from memory once, latency is all there is to hide
                    for (int i = j + 1 + g; i < m; i += U * RG) {
                        double yv[U];
#pragma unroll
                        for (int u = 0; u < U; ++u) { const int ii = i + u * RG; yv[u] = Y[(int64_t)(ii < m ? ii : m - 1) * n + j]; }
#pragma unroll
                        for (int u = 0; u < U; ++u) { const int ii = i + u * RG; if (ii < m) acc = fma(yv[u], seg[ii], acc); }
                    }
                }
                if (g < RG) part[g * CW + jl] = acc;
                __syncthreads();
                if (g == 0 && jl < nc) {
                    double d = j < m ? seg[j] : 0.0;          // the unit diagonal
                    for (int q = 0; q < RG; ++q) d += part[q * CW + jl];
                    w1[j] = d;
                }
                __syncthreads();
            }
            // ---- w2 = T^T w1 or T w1 (T upper triangular, column-major)
            if (transpose) {
                // a wave per row of T^T, four rows at a time: every load of the batch is issued before the first sum (one row at a
                // time was one memory latency per row: 12 of them per wave and panel)
                constexpr int RB = 4, LB = BC_CW / 64;
                for (int i0 = wave; i0 < n; i0 += RB * BA_WAVES) {
                    double tv[RB][LB];
#pragma unroll
                    for (int rb = 0; rb < RB; ++rb) {
                        const int i = i0 + rb * BA_WAVES;
#pragma unroll
                        for (int q = 0; q < LB; ++q) { const int j = lane + 64 * q; tv[rb][q] = (i < n && j <= i) ? T[(int64_t)i * n + j] : 0.0; }
                    }
#pragma unroll
                    for (int rb = 0; rb < RB; ++rb) {
                        const int i = i0 + rb * BA_WAVES;
                        double d = 0.0;
#pragma unroll
                        for (int q = 0; q < LB; ++q) { const int j = lane + 64 * q; if (j < n) d = fma(tv[rb][q], w1[j], d); }
                        d = bb_wave_sum_dpp(d);
                        if (lane == 0 && i < n) w2[i] = d;
                    }
                }
            } else {
                for (int i0 = 0; i0 < n; i0 += BA_THREADS) {
                    const int nc = (n - i0) < BA_THREADS ? (n - i0) : BA_THREADS;
                    const int CW = ((nc + 63) / 64) * 64, RG = BA_THREADS / CW;
                    const int il = tid % CW, g = tid / CW, i = i0 + il;
                    double acc = 0.0;
                    if (g < RG && il < nc) {
                        constexpr int U = 8;
                        for (int j = i + g; j < n; j += U * RG) {
                            double tv[U];
#pragma unroll
                            for (int u = 0; u < U; ++u) { const int jj = j + u * RG; tv[u] = T[(int64_t)(jj < n ? jj : n - 1) * n + i]; }
#pragma unroll
                            for (int u = 0; u < U; ++u) { const int jj = j + u * RG; if (jj < n) acc = fma(tv[u], w1[jj], acc); }
                        }
                    }
                    if (g < RG) part[g * CW + il] = acc;
                    __syncthreads();
                    if (g == 0 && il < nc) {
                        double d = 0.0;
                        for (int q = 0; q < RG; ++q) d += part[q * CW + il];
                        w2[i] = d;
                    }
                    __syncthreads();
                }
            }
            __syncthreads();
            // ---- seg += Y w2, a wave per row
            constexpr int RB3 = 4, LB3 = BC_CW / 64;
            for (int i0 = wave; i0 < m; i0 += RB3 * BA_WAVES) {
              // (four rows of Y per wave in flight, as above: 16 loads per lane; eight spill at 1 024 threads)
              double yv[RB3][LB3];
#pragma unroll
              for (int rb = 0; rb < RB3; ++rb) {
                  const int i = i0 + rb * BA_WAVES;
                  const int je = i < n ? i : n;
#pragma unroll
                  for (int q = 0; q < LB3; ++q) { const int j = lane + 64 * q; yv[rb][q] = (i < m && j < je) ? Y[(int64_t)i * n + j] : 0.0; }
              }
#pragma unroll
              for (int rb = 0; rb < RB3; ++rb) {
                const int i = i0 + rb * BA_WAVES;
                if (i >= m) break;
                double d = 0.0;
#pragma unroll
                for (int q = 0; q < LB3; ++q) { const int j = lane + 64 * q; if (j < n) d = fma(yv[rb][q], w2[j], d); }
                d = bb_wave_sum_dpp(d);
                if (lane == 0) {
                    const double val = seg[i] + d + (i < n ? w2[i] : 0.0);
                    if (!strips) x[i < n ? p.yrow + i : seg2 + (i - n)] = val;
                    else if (transpose) {
                        if (i < p.solved) full[(int64_t)S.s * pidx + i] = val;
                        else if (i < n) carry[i - p.solved] = val;
                        else full[bbs_res_off(S, pidx) + (i - n)] = val;
                    } else {
                        double* yi = ya + (int64_t)pidx * S.ms;
                        if (pidx == 0) yi[i] = val;
                        else if (i < 2 * S.lo) { if (i & 1) yi[i >> 1] = val; else carry[i >> 1] = val; }
                        else yi[i - S.lo] = val;
                    }
                }
              }
            }
            if (strips) {
                // the components stage A left out of the chain (rows n.. of the strip's vector) pass through
                const int64_t ro = bbs_res_off(S, pidx) + (pidx >= 1 ? S.lo : 0);
                double* yi = ya + (int64_t)pidx * S.ms;
                for (int q = tid; q < S.ms - S.n; q += BA_THREADS) { if (transpose) full[ro + q] = yi[S.n + q]; else yi[S.n + q] = full[ro + q]; }
            }
            __syncthreads();
        }
    }
}

size_t bb_chain_smem(int max_act_rows, int max_ncols) { return (size_t)(max_act_rows + 2 * max_ncols + BB_WAVES) * sizeof(double); }
size_t bb_apply_smem(int max_act_rows, int max_ncols) { return (size_t)(max_act_rows + 2 * max_ncols + BA_THREADS + max_ncols) * sizeof(double); }

hipError_t launch_bb_chain(const BBPanel* panels, int num_panels, const int32_t* prowptr, const int32_t* pcol,
                           const int64_t* pmap, const double* vals, double* W, double* lo, double* y_vals, double* t_vals,
                           double* r_stage, const int64_t* r_src, int64_t nnz_r, double* r_vals, int max_act_rows,
                           int max_ncols, hipStream_t stream)
{
    int t_in_lds = 0, uni_doubles = 0;
    const size_t smem2 = bb_chain2_smem(max_act_rows, &uni_doubles);
    if (max_ncols <= BC_CW && smem2 > 0 && !std::getenv("QRK_BB_CHAIN_V1")) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(bb_chain2_kernel),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem2);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(bb_scatter_kernel, dim3((unsigned)num_panels), dim3(BC_THREADS),
                           (size_t)(max_act_rows + 2) * sizeof(int), stream, panels, prowptr, pcol, pmap, vals, y_vals);
        hipLaunchKernelGGL(bb_chain2_kernel, dim3(1), dim3(BC_THREADS), smem2, stream, panels, num_panels, prowptr, pcol, pmap,
                           vals, lo, y_vals, t_vals, r_stage, max_act_rows, max_ncols, uni_doubles, (const int*)nullptr, (const int*)nullptr,
                           (int*)nullptr, 0u);
        const size_t smem_t = bb_t_smem(max_ncols, &t_in_lds);
        e = hipFuncSetAttribute(reinterpret_cast<const void*>(bb_t_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)smem_t);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(bb_t_kernel, dim3((unsigned)num_panels), dim3(BC_THREADS), smem_t, stream, panels, num_panels,
                           y_vals, t_vals, t_in_lds);
    } else {
        const size_t smem = bb_chain_smem(max_act_rows, max_ncols);
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(bb_chain_kernel),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(bb_chain_kernel, dim3(1), dim3(BB_THREADS), smem, stream, panels, num_panels, prowptr, pcol, pmap, vals,
                           W, lo, y_vals, t_vals, r_stage, max_act_rows, max_ncols);
    }
    if (nnz_r > 0)
        hipLaunchKernelGGL(bb_gather_r_kernel, dim3((unsigned)((nnz_r + 255) / 256)), dim3(256), 0, stream, r_stage, r_src,
                           nnz_r, r_vals);
    return hipGetLastError();
}

hipError_t launch_bb_apply_q(const BBPanel* panels, int num_panels, const double* y_vals, const double* t_vals,
                             int transpose, double* v, int64_t ldv, int64_t nrhs, int max_act_rows, int max_ncols,
                             hipStream_t stream)
{
    if (nrhs <= 0) return hipSuccess;
    const size_t smem = bb_apply_smem(max_act_rows, max_ncols);
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(bb_apply_q_kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    if (e != hipSuccess) return e;
    const unsigned grid = (unsigned)(nrhs < 1024 ? nrhs : 1024);
    hipLaunchKernelGGL(bb_apply_q_kernel, dim3(grid), dim3(BA_THREADS), smem, stream, panels, num_panels, y_vals, t_vals,
                       transpose, v, ldv, nrhs, max_act_rows, max_ncols, BBStrips{});
    return hipGetLastError();
}

// ---- the strips form: stage B of the two-stage banded factorisation ---------------------------------------------------------
// Stage A (the caller: the block-diagonal solver on the strips, HouseholderQR) left R_i (n x n upper triangle, packed by columns)
// of every strip.  bbs_scatter_kernel writes the stack of panel i into the panel's own storage (row-major act x n): panel 0 is
// R_0 itself; panel i >= 1 holds R_i's first lo rows on the ODD rows 1, 3, .. 2 lo - 1 (the even rows receive the carry inside
// the chain) and its other rows from row 2 lo on.  Interleaved like this, row r of the stack has its first nonzero in column
// r / 2 (r < 2 lo) or r - lo: a staircase, so that a plain Householder QR in the natural row order only ever touches the rows
// [j, 2 j + 2) (j < lo) or [j, lo + j + 1) of column j -- the two triangles are merged without ever visiting their zeros
// (4.5 instead of 18.9 Mflop for lo = 128, n = 192).
__global__ void __launch_bounds__(256)
bbs_scatter_kernel(const BBPanel* __restrict__ panels, const double* __restrict__ r_packed, int64_t r_stride, int n, int lo,
                   double* __restrict__ y_vals)
{
    const int pi = blockIdx.x;
    const BBPanel p = panels[pi];
    const double* R = r_packed + (int64_t)pi * r_stride;      // R(r, c) = R[c (c + 1) / 2 + r], r <= c
    double* W = y_vals + p.y_off;
    for (int e = threadIdx.x; e < p.act_rows * n; e += 256) {
        const int i = e / n, c = e - i * n;
        int r;                                                // row of R_i that lives on stack row i (-1: a carry row)
        if (pi == 0) r = i;
        else if (i < 2 * lo) r = (i & 1) ? (i >> 1) : -1;
        else r = i - lo;
        W[e] = (r >= 0 && r <= c) ? R[(int64_t)c * (c + 1) / 2 + r] : 0.0;
    }
}

// done: 2 num_panels + 2 ints (the rows-final words of the pipelined chain, its abort word at [num_panels], the second word of every
// panel from [num_panels + 1] on; zeroed here), or null: one workgroup
// walks the strips.  single != 0: one workgroup whatever QRK_BBS_PIPE says (the caller's second run after an aborted chain).
// *piped_out: the chain ran on more than one workgroup -- the caller must read done[num_panels] once the stream has drained.
hipError_t launch_bbs_chain(const BBPanel* panels, int num_panels, const double* r_packed, int64_t r_stride, int n, int lo,
                            int max_act_rows, double* lo_buf, double* y_vals, double* t_vals, double* r_stage,
                            const int* rlim_first, const int* rlim_rest, int* done, int single, int* piped_out, hipStream_t stream)
{
    int t_in_lds = 0, uni_doubles = 0;
    const size_t smem2 = bb_chain2_smem(max_act_rows, &uni_doubles);
    if (n > BC_CW || smem2 == 0) return hipErrorInvalidValue;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(bb_chain2_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem2);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(bbs_scatter_kernel, dim3((unsigned)num_panels), dim3(256), 0, stream, panels, r_packed, r_stride, n, lo, y_vals);
    // workgroups of the pipelined chain (QRK_BBS_PIPE: 1 = the single-workgroup chain of round 3): with the carry taken over block by
    // block two panels overlap; a third and fourth workgroup only add slack
    int G = 3;
    if (const char* e2 = std::getenv("QRK_BBS_PIPE")) { const int v = std::atoi(e2); if (v >= 1 && v <= 16) G = v; }
    if (G > num_panels) G = num_panels;
    if (!done || lo <= 0 || single) G = 1;
    // polls of a rows-final word before a workgroup gives the chain up (about a microsecond each: seconds, not minutes;
    // QRK_BBS_PIPE_SPINS=0 makes the first unsatisfied wait abort -- the test of the fall-back)
    unsigned spin_limit = 1u << 21;
    if (const char* e3 = std::getenv("QRK_BBS_PIPE_SPINS")) spin_limit = (unsigned)std::strtoul(e3, nullptr, 10);
    if (G > 1) { e = hipMemsetAsync(done, 0, (size_t)(2 * num_panels + 2) * sizeof(int), stream); if (e != hipSuccess) return e; }
    if (piped_out) *piped_out = G > 1;
    hipLaunchKernelGGL(bb_chain2_kernel, dim3((unsigned)(QRK_BB_SAME_XCD && G > 1 ? 8 * (G - 1) + 1 : G)), dim3(BC_THREADS), smem2, stream, panels, num_panels, (const int32_t*)nullptr,
                       (const int32_t*)nullptr, (const int64_t*)nullptr, (const double*)nullptr, lo_buf, y_vals, t_vals, r_stage,
                       max_act_rows, n, uni_doubles, rlim_first, rlim_rest, G > 1 ? done : (int*)nullptr, spin_limit);
    const size_t smem_t = bb_t_smem(n, &t_in_lds);
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(bb_t_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem_t);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(bb_t_kernel, dim3((unsigned)num_panels), dim3(BC_THREADS), smem_t, stream, panels, num_panels, y_vals, t_vals, t_in_lds);
    return hipGetLastError();
}

hipError_t launch_bbs_apply(const BBPanel* panels, int num_panels, const double* y_vals, const double* t_vals, int transpose,
                            double* ya, int64_t ya_ld, double* full, int64_t full_ld, int64_t nrhs, int ms, int n, int s, int lo,
                            int cols, int max_act_rows, hipStream_t stream)
{
    if (nrhs <= 0) return hipSuccess;
    const size_t smem = bb_apply_smem(max_act_rows, n);
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(bb_apply_q_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    if (e != hipSuccess) return e;
    BBStrips S{ya, full, ya_ld, full_ld, ms, n, s, lo, cols};
    const unsigned grid = (unsigned)(nrhs < 1024 ? nrhs : 1024);
    hipLaunchKernelGGL(bb_apply_q_kernel, dim3(grid), dim3(BA_THREADS), smem, stream, panels, num_panels, y_vals, t_vals, transpose,
                       (double*)nullptr, (int64_t)0, nrhs, max_act_rows, n, S);
    return hipGetLastError();
}

hipError_t launch_bb_solve_r(const BBPanel* panels, int num_panels, const double* r_stage, int cols, double* v, int64_t ldv,
                             int64_t nrhs, hipStream_t stream)
{
    if (nrhs <= 0 || num_panels <= 0) return hipSuccess;
    hipLaunchKernelGGL(bb_solve_r_kernel, dim3((unsigned)nrhs), dim3(BS_THREADS), 0, stream, panels, BBPanel{}, num_panels, r_stage,
                       cols, v, ldv);
    return hipGetLastError();
}

// One dense upper triangle, a few right-hand sides, MANY workgroups (round 5).  bb_solve_r_kernel gives a right-hand side one workgroup,
// which pulls the 16 MB of a 2 000 x 2 000 triangle through one CU, 64 rows at a time: 1.37 ms of the 8 ms of configs[3]'s solve().  Here
// the block of 64 rows kb has a workgroup of its own (blockIdx.x = 0 is the BOTTOM block): it takes the column blocks jb > kb in the
// order in which their x becomes available -- bottom first -- waiting on one flag word each, subtracts R(kb, jb) x_jb with its four
// waves, solves its diagonal block as bb_solve_r_kernel does and publishes x_kb.  A workgroup only ever waits for workgroups with a
// SMALLER blockIdx.x, which were dispatched before it: no co-residency assumption, no deadlock; the wait is bounded all the same.
// flags: [nrhs][nblk] ints, zero before the launch.  (The sums run over the column blocks from the bottom up: not the order of the
// one-workgroup kernel, the same result to rounding.)
// Block kb of one right-hand side.  WAIT: the workgroup waits for the flag of every block below (2 = aborted: it stops); else every block
// below is final.  Returns false when a wait ran out or met an aborted block: nothing of x has been written then.
template <bool WAIT>
__device__ __forceinline__ bool dense_solve_r_block(const double* __restrict__ R, int64_t lda, int n, double* __restrict__ x, int* __restrict__ fl,
                                                    int nblk, int kb, double* blk /* [64 * 65] */, double* part /* [4 * 64] */, int* ok /* LDS word */,
                                                    int max_spins)
{
    const int tid = threadIdx.x, ln = tid & 63, grp = tid >> 6;
    const int c0 = kb * 64, nr = (n - c0) < 64 ? (n - c0) : 64;
    for (int e = tid; e < nr * nr; e += BS_THREADS) {
        const int j = e / nr, i = e - j * nr;
        blk[j * 65 + i] = R[(int64_t)(c0 + j) * lda + c0 + i];
    }
    double acc = 0.0;
    const double* rrow = R + c0 + (ln < nr ? ln : 0);
    for (int jb = nblk - 1; jb > kb; --jb) {
        if (WAIT) {
            if (tid == 0) {
                int spins = 0, f;
                while ((f = __hip_atomic_load(&fl[jb], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT)) == 0 && ++spins < max_spins) __builtin_amdgcn_s_sleep(2);
                *ok = f == 1;
            }
            __syncthreads();
            const bool go = *ok != 0;
            __syncthreads();
            if (!go) return false;                           // (never on with x_jb that is not there: the finishing kernel takes over)
            __threadfence();                                 // (x of block jb as its owner wrote it, not a stale line of this CU's L1)
        }
        const int j0 = jb * 64, j1 = (n - j0) < 64 ? n : j0 + 64;
        // the 64 columns of the block dealt round-robin to the four waves, eight loads in flight
        constexpr int U = 8;
        for (int j = j0 + grp; j < j1; j += 4 * U) {
            double rv[U], xv[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int jj = j + 4 * u, jc = jj < j1 ? jj : j1 - 1;
                rv[u] = rrow[(int64_t)jc * lda];
                xv[u] = jj < j1 ? __builtin_nontemporal_load(&x[jc]) : 0.0;
            }
#pragma unroll
            for (int u = 0; u < U; ++u) acc = fma(rv[u], xv[u], acc);
        }
    }
    part[grp * 64 + ln] = acc;
    __syncthreads();
    if (grp == 0) {
        double t = 0.0;
        if (ln < nr) t = x[c0 + ln] - ((part[ln] + part[64 + ln]) + (part[128 + ln] + part[192 + ln]));
        // back substitution inside the block.  The 64 dependent divisions were 60 % of the chain of a block (32 blocks in a row at
        // n = 2 000): lane i divides ONCE, ahead of the chain, and a step is t_i * (1 / d_i) with one residual correction
        // (x + (t - x d) / d: the quotient to the last bit but for rare ties)
        const double dl = blk[(ln < nr ? ln : 0) * 65 + (ln < nr ? ln : 0)];
        const double rdl = 1.0 / dl;
        for (int i = nr - 1; i >= 0; --i) {
            const double ti = readlane_f64(t, i), di = readlane_f64(dl, i), ri = readlane_f64(rdl, i);
            double xi = ti * ri;
            xi = fma(fma(-xi, di, ti), ri, xi);
            if (ln < i) t = fma(-blk[i * 65 + ln], xi, t);
            else if (ln == i) t = xi;
        }
        if (ln < nr) x[c0 + ln] = t;
        __threadfence();
        if (ln == 0) __hip_atomic_store(&fl[kb], 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
    return true;
}

__global__ void __launch_bounds__(BS_THREADS)
dense_solve_r_coop_kernel(const double* __restrict__ R, int64_t lda, int n, double* __restrict__ b, int64_t ldb, int* __restrict__ flags,
                          int nblk, int* __restrict__ abort_word, int max_spins)
{
    __shared__ double blk[64 * 65];          // diagonal block, blk[j * 65 + i] = R(i, j)
    __shared__ double part[4 * 64];
    __shared__ int ok;
    const int kb = nblk - 1 - (int)blockIdx.x;
    int* fl = flags + (int64_t)blockIdx.y * nblk;
    if (!dense_solve_r_block<true>(R, lda, n, b + (int64_t)blockIdx.y * ldb, fl, nblk, kb, blk, part, &ok, max_spins) && threadIdx.x == 0) {
        // a wait ran out (a preempted or contended GPU): this block's rows of b are untouched; its flag says so to the blocks above, the
        // word to the finishing kernel
        __hip_atomic_store(&fl[kb], 2, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(abort_word, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// Queued behind the many-workgroup kernel: nothing to do unless a wait ran out there -- then the blocks that stopped (a prefix from the top:
// a block stops exactly when one below it did) are solved here, bottom first, on one workgroup per right-hand side, from the rows of b
// they left untouched.  No wait inside: every workgroup of the first kernel has ended.
__global__ void __launch_bounds__(BS_THREADS)
dense_solve_r_finish_kernel(const double* __restrict__ R, int64_t lda, int n, double* __restrict__ b, int64_t ldb, int* __restrict__ flags,
                            int nblk, const int* __restrict__ abort_word)
{
    __shared__ double blk[64 * 65];
    __shared__ double part[4 * 64];
    __shared__ int ok;
    if (__hip_atomic_load(abort_word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0) return;
    int* fl = flags + (int64_t)blockIdx.y * nblk;
    for (int kb = nblk - 1; kb >= 0; --kb) {
        if (__hip_atomic_load(&fl[kb], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) == 1) continue;      // (uniform: every thread reads the same word)
        __threadfence();
        (void)dense_solve_r_block<false>(R, lda, n, b + (int64_t)blockIdx.y * ldb, fl, nblk, kb, blk, part, &ok, 0);
    }
}

// b(0:n, :) <- R^-1 b(0:n, :) for the upper triangle R (n x n) of a column-major array with leading dimension lda.
// flags (or null): flags_cap ints of device scratch for the many-workgroup form (n >= 512 and nrhs * ceil(n / 64) <= flags_cap).
hipError_t launch_dense_solve_r(const double* qr, int64_t lda, int n, double* b, int64_t ldb, int64_t nrhs, hipStream_t stream, int* flags,
                                int flags_cap)
{
    if (nrhs <= 0 || n <= 0) return hipSuccess;
    if (lda > INT32_MAX) return hipErrorInvalidValue;
    // (diagnostic switches, read on every call like QRK_BBS_MAPS: QRK_SOLVE_R_COOP=0 keeps the one-workgroup kernel, QRK_SOLVE_R_SPINS
    //  bounds the flag waits -- 0 makes every block but the bottom one stop, which is how the tests reach the finishing kernel)
    const char* csw = std::getenv("QRK_SOLVE_R_COOP");
    const bool coop = !(csw && csw[0] == '0');
    int max_spins = 1 << 22;
    if (const char* e = std::getenv("QRK_SOLVE_R_SPINS")) max_spins = std::atoi(e);
    const int nblk = (n + 63) / 64;
    if (coop && flags && n >= 512 && nrhs * nblk + 1 <= flags_cap) {
        // flags: [nrhs][nblk] (0 = not yet, 1 = x of the block is final, 2 = the block stopped on a wait that ran out) + the abort word
        int* abort_word = flags + nrhs * nblk;
        if (hipError_t e = hipMemsetAsync(flags, 0, (size_t)(nrhs * nblk + 1) * sizeof(int), stream)) return e;
        hipLaunchKernelGGL(dense_solve_r_coop_kernel, dim3((unsigned)nblk, (unsigned)nrhs), dim3(BS_THREADS), 0, stream, qr, lda, n, b, ldb,
                           flags, nblk, abort_word, max_spins);
        hipLaunchKernelGGL(dense_solve_r_finish_kernel, dim3(1, (unsigned)nrhs), dim3(BS_THREADS), 0, stream, qr, lda, n, b, ldb, flags, nblk,
                           abort_word);
        return hipGetLastError();
    }
    BBPanel one{};
    one.col0 = 0; one.ncols = n; one.solved = (int32_t)lda; one.r_off = 0;     // R(i, j) = qr[j * lda + i]
    hipLaunchKernelGGL(bb_solve_r_kernel, dim3((unsigned)nrhs), dim3(BS_THREADS), 0, stream, (const BBPanel*)nullptr, one, 1, qr, n, b,
                       ldb);
    return hipGetLastError();
}

}  // namespace qrk
