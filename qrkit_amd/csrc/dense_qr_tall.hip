// dense_qr_tall.hip -- dense Householder QR with implicit Q for TALL right blocks, on the whole GPU.
//
// Same seam as dense_qr.hip: the right-block solver of QRKit::BlockAngularSparseQR (Eigen
// ColPivHouseholderQR / HouseholderQR, src/QRKit/BlockAngularSparseQR.h:361-369, :488, :498-503),
// for matrices that do not fit the single-workgroup kernel -- BASELINE configs[3] has a 40000 x 2000
// bottom block.
//
// The rows are cut into slabs, one workgroup per slab; a reflector is applied slab by slab and only
// per-column scalars cross the slabs.  Per step k the level-2 algorithm is FUSED as in bdqr_col.hip:
// the dot products of step k give row k of the updated matrix, hence the downdated norms and the next
// pivot, so one read-modify-write sweep applies update k, swaps the next pivot column into place and
// accumulates the partial dot products with it.  A step is a short sequence of kernels on the caller's
// stream (no host synchronisation, no spin barriers):
//   head   (1 workgroup)  reduce the slab partials, reflector scalars, row k of R, norm downdate,
//                         next pivot -- or "slow" when Eigen's norm-recompute test fires;
//   sweep  (all slabs)    update + swap + partial dots with the next reflector (slow: update + partial
//                         column norms only);
//   swap_dots (slow steps only; a no-op otherwise) the dots with the pivot that the last workgroup of the slow sweep
//                         chose from the recomputed norms.
// Traffic: the trailing matrix is read and written once per step (1.3 TB for 40000 x 2000).
#include "qrk_device.h"

#include <float.h>

namespace qrk {

namespace tall {

constexpr int TT = 1024;                 // threads per workgroup
constexpr int TW = TT / 64;
using namespace decide;                  // decisions inside their error margin send the matrix to the exact path (qrk_device.h)

struct State {
    double s, ng, inv_s;     // reflector of the current step: s = x0 - beta, ng = -1/(beta w), 1/s (0 if H = I)
    int P;                   // column to swap into position k+1 (already swapped in the bookkeeping arrays)
    int slow;                // the norm-recompute test fired in this step
    unsigned ticket;         // head_kernel: workgroups that have finished their columns (reset by the last one)
    int anyneed;             // head_kernel: some column failed the downdate test (reset by the last one)
    unsigned ticket2;        // sweep / swap_dots: workgroups that have finished (reset by the last one), see slab_tail()
    unsigned bar_count;      // persistent kernel: workgroups that have arrived at the grid barrier
    unsigned bar_gen;        // ... and its generation
    unsigned bar_abort;      // ... set when a workgroup gave up waiting (the kernel then drains without computing)
    double a2;               // |A|^2: squared norm of the first pivot column (scale of the decision margins)
    int unclear;             // some decision was not clear of rounding: the exact path redoes the factorisation
};

// Value another workgroup wrote during this kernel (persistent form): read at agent scope, never from a stale line.
__device__ __forceinline__ int ld_agent(const int* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ double ld_agent(const double* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

struct Work {                // device workspace of a plan
    double* partial;         // [G][cpad] partial dot products with the next reflector
    double* sqpart;          // [G][cpad] partial squared column norms
    double* tpart;           // [G] partial |x_tail|^2
    double* nu2;             // [cpad] m_colNormsUpdated^2
    double* thr;             // [cpad] sqrt(eps) m_colNormsDirect^2
    double* ngamv;           // [cpad] -gamma per column for the step being applied
    int* need;               // [cpad] recompute flags
    int* pidx;               // [cpad] permutation indices
    State* st;
    int G, cpad, rows_per;
};

__device__ __forceinline__ double wave_sum(double v)
{
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off);
    return v;
}

// Block-wide first maximum of (val, idx): larger val wins, ties -> smaller idx.  red/ired: [TW].
__device__ __forceinline__ void block_argmax(double& best, int& bi, double* red, int* ired)
{
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        const double ob = __shfl_xor(best, off);
        const int oi = __shfl_xor(bi, off);
        if (ob > best || (ob == best && oi < bi)) { best = ob; bi = oi; }
    }
    __syncthreads();
    if ((threadIdx.x & 63) == 0) { red[threadIdx.x >> 6] = best; ired[threadIdx.x >> 6] = bi; }
    __syncthreads();
    best = red[0]; bi = ired[0];
#pragma unroll
    for (int w = 1; w < TW; ++w)
        if (red[w] > best || (red[w] == best && ired[w] < bi)) { best = red[w]; bi = ired[w]; }
}

// partial squared column norms of every slab
__device__ __forceinline__ void norms_body(const double* __restrict__ A, int64_t lda, int r, int c, Work w, int vb)
{
    const int g = vb, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r0 = g * w.rows_per, r1 = min(r, r0 + w.rows_per);
    for (int jc = wave; jc < c; jc += TW) {
        double s = 0.0;
        for (int i = r0 + lane; i < r1; i += 64) { const double v = A[(int64_t)jc * lda + i]; s = fma(v, v, s); }
        s = wave_sum(s);
        if (lane == 0) w.sqpart[(int64_t)g * w.cpad + jc] = s;
    }
}

// Pivot of position `kpos` among columns kpos..c-1 by the bookkeeping norms, then swap the bookkeeping.
__device__ __forceinline__ void choose_and_swap(int kpos, int c, int pivoting, Work& w, double* red, int* ired)
{
    int P = kpos;
    if (pivoting) {
        double best = -1.0; int bi = c;
        for (int jc = kpos + threadIdx.x; jc < c; jc += TT) { const double v = w.nu2[jc]; if (v > best) { best = v; bi = jc; } }
        block_argmax(best, bi, red, ired);
        P = bi < c ? bi : kpos;
        // decision (1): another column within the error margin of the chosen one
        const double a2 = kpos == 0 ? best : ld_agent(&w.st->a2);
        if (kpos == 0 && threadIdx.x == 0) w.st->a2 = best;
        bool nr = false;
        for (int jc = kpos + threadIdx.x; jc < c; jc += TT) nr = nr || (jc != P && near_best(w.nu2[jc], w.thr[jc], best, a2));
        if (nr) atomicOr(&w.st->unclear, 1);
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        w.st->P = P;
        if (P != kpos) {
            double t = w.nu2[kpos]; w.nu2[kpos] = w.nu2[P]; w.nu2[P] = t;
            t = w.thr[kpos]; w.thr[kpos] = w.thr[P]; w.thr[P] = t;
            const int tp = w.pidx[kpos]; w.pidx[kpos] = w.pidx[P]; w.pidx[P] = tp;
        }
    }
}

__device__ __forceinline__ void init_body(int c, int pivoting, Work w)
{
    __shared__ double red[TW];
    __shared__ int ired[TW];
    for (int jc = threadIdx.x; jc < c; jc += TT) {
        double s = 0.0;
        for (int g = 0; g < w.G; ++g) s += w.sqpart[(int64_t)g * w.cpad + jc];
        w.nu2[jc] = s; w.thr[jc] = s * THR_HI; w.pidx[jc] = jc;
    }
    if (threadIdx.x == 0) { w.st->slow = 0; w.st->ticket = 0u; w.st->anyneed = 0; w.st->ticket2 = 0u; w.st->unclear = 0; w.st->a2 = 0.0; }
    __syncthreads();
    choose_and_swap(0, c, pivoting, w, red, ired);
}

// The workgroup that finishes a slow sweep LAST also does what used to be a kernel of its own - the recomputed norms and
// the next pivot - so that the sequence is one kernel shorter per reflector (a boundary costs ~8.5 us).  Same ticket scheme
// as the head: device-scope fence, atomic ticket, nobody waits.  (Running the HEAD of the next step there as well, for
// problems whose slab partials one workgroup can sum, was measured slower: 20.3 vs 12.9 ms at 5120 x 384.)
enum { TAIL_NONE = 0, TAIL_RECOMPUTE = 1 };
__device__ __forceinline__ void recompute_body(int c, int k, int pivoting, Work w);
__device__ __forceinline__ bool slab_is_last(Work& w, int nslabs)
{
    __shared__ unsigned s_t2;
    __threadfence();
    __syncthreads();
    if (threadIdx.x == 0) s_t2 = atomicAdd(&w.st->ticket2, 1u);
    __syncthreads();
    if (s_t2 != (unsigned)nslabs - 1u) return false;
    __threadfence();
    if (threadIdx.x == 0) w.st->ticket2 = 0u;
    return true;
}
// Swap columns kpos and st->P (all rows of the slab), then partial dots of x = column kpos (rows > kpos)
// with every column to its right, and the partial |x_tail|^2.  No-op unless `always` or the step was slow.
__device__ __forceinline__ void swap_dots_body(double* __restrict__ A, int64_t lda, int r, int c, int kpos, int always, Work w, int vb, double* xs)
{
    if (!always && !ld_agent(&w.st->slow)) return;
    const int g = vb, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r0 = g * w.rows_per, r1 = min(r, r0 + w.rows_per);
    const int P = ld_agent(&w.st->P);
    if (P != kpos) {
        for (int i = r0 + tid; i < r1; i += TT) {
            const double t = A[(int64_t)kpos * lda + i];
            A[(int64_t)kpos * lda + i] = A[(int64_t)P * lda + i];
            A[(int64_t)P * lda + i] = t;
        }
    }
    __syncthreads();
    for (int i = r0 + tid; i < r1; i += TT) xs[i - r0] = A[(int64_t)kpos * lda + i];
    __syncthreads();
    const int i0 = max(r0, kpos + 1);
    if (wave == 0) {
        double t = 0.0;
        for (int i = i0 + lane; i < r1; i += 64) t = fma(xs[i - r0], xs[i - r0], t);
        t = wave_sum(t);
        if (lane == 0) w.tpart[g] = t;
    }
    for (int jc = kpos + 1 + wave; jc < c; jc += TW) {
        double d = 0.0;
        for (int i = i0 + lane; i < r1; i += 64) d = fma(xs[i - r0], A[(int64_t)jc * lda + i], d);
        d = wave_sum(d);
        if (lane == 0) w.partial[(int64_t)g * w.cpad + jc] = d;
    }
}

// Head of step k: reflector scalars, row k of R, norm downdate, next pivot (or slow).
// The slab partials of a step are G x (c - k) doubles (4 MB at 40000 x 2000): summed by ONE workgroup they cost as
// much as the sweep itself (97 us per step, one CU's path to L2).  So the columns are dealt to several workgroups
// (128 columns each, eight groups of slabs per column, combined through LDS in a fixed order), and the part that needs
// all the columns - the slow/fast decision and the next pivot - is done by the workgroup that finishes last
// (a ticket counter after a device-scope fence: no workgroup ever waits for another).
constexpr int HC = 128;                  // columns per workgroup
constexpr int HG = TT / HC;              // slab groups per column
__device__ __forceinline__ void head_body(double* __restrict__ A, int64_t lda, int r, int c, int k, int pivoting, double* __restrict__ hcoeffs,
                                          int32_t* __restrict__ perm, Work w, int vb, int vgrid)
{
    __shared__ double red[TW];
    __shared__ int ired[TW];
    __shared__ double part[TT];
    __shared__ unsigned s_ticket;
    __shared__ int s_any;
    const int tid = threadIdx.x;
    // |x_tail|^2 from the slab partials (every workgroup: it needs the reflector scalars)
    double t = 0.0;
    for (int g = tid; g < w.G; g += TT) t += w.tpart[g];
    t = wave_sum(t);
    if ((tid & 63) == 0) red[tid >> 6] = t;
    __syncthreads();
    double tsq = 0.0;
#pragma unroll
    for (int q = 0; q < TW; ++q) tsq += red[q];
    const double xk = A[(int64_t)k * lda + k];       // (rewritten with beta by the last workgroup only)
    // makeHouseholder, un-normalised form of bdqr_pair.hip: nb = -beta, s = x0 - beta, ng = -1/(beta w)
    double nb, s, ng, tau;
    const bool degen = !(tsq > DBL_MIN);
    if (degen) { nb = -xk; s = 0.0; ng = 0.0; tau = 0.0; }
    else {
        const double nrm = sqrt(fma(xk, xk, tsq));
        nb = xk >= 0.0 ? nrm : -nrm;
        s = nb + xk;
        ng = -1.0 / (nb * s);
        tau = -(s * s) * ng;
    }
    // this workgroup's columns
    {
        const int cl = tid % HC, gg = tid / HC;
        const int jc = k + 1 + vb * HC + cl;
        double d = 0.0;
        if (jc < c) for (int g = gg; g < w.G; g += HG) d += w.partial[(int64_t)g * w.cpad + jc];
        part[gg * HC + cl] = d;
        __syncthreads();
        if (gg == 0 && jc < c) {
            d = 0.0;
#pragma unroll
            for (int q = 0; q < HG; ++q) d += part[q * HC + cl];
            const double ak = A[(int64_t)jc * lda + k];
            const double ngam = fma(s, ak, d) * ng;
            const double an = fma(s, ngam, ak);
            w.ngamv[jc] = ngam;
            A[(int64_t)jc * lda + k] = an;            // row k of R
            int nd = 0;
            if (pivoting) {
                // LAWN-176 downdate in squared form (see bdqr_pair.hip); no clamp: a negative value is recomputed
                const double nn = fma(-an, an, w.nu2[jc]);
                w.nu2[jc] = nn;
                nd = nn <= w.thr[jc];
                if (nd) atomicOr(&w.st->anyneed, 1);
                if (nd && in_recompute_band(nn, w.thr[jc], ld_agent(&w.st->a2))) atomicOr(&w.st->unclear, 1);   // decision (2)
            }
            w.need[jc] = nd;
        }
    }
    __threadfence();
    __syncthreads();
    if (tid == 0) s_ticket = atomicAdd(&w.st->ticket, 1u);
    __syncthreads();
    if (s_ticket != (unsigned)vgrid - 1u) return;
    // ---- the last workgroup: every column of the step is in memory
    __threadfence();
    if (tid == 0) {
        s_any = atomicOr(&w.st->anyneed, 0);
        w.st->anyneed = 0; w.st->ticket = 0u;
        w.st->s = s; w.st->ng = ng; w.st->inv_s = degen ? 0.0 : 1.0 / s;
        A[(int64_t)k * lda + k] = -nb;            // beta (= x0 when H = I)
        hcoeffs[k] = tau;
        if (k == 0 && !pivoting) w.st->a2 = fma(xk, xk, tsq);
        if (unclear_reflector(xk, tsq, k + 1 < r, pivoting != 0, k == 0 && !pivoting ? fma(xk, xk, tsq) : w.st->a2,
                              (pivoting & PIVOTING_SIGN_FREE) != 0))
            w.st->unclear = 1;                    // decisions (3), (4), (5)
    }
    __syncthreads();
    const int size = r < c ? r : c;
    if (k + 1 >= size) {
        if (tid == 0) { w.st->slow = 0; w.st->P = k + 1; }
        for (int jc = tid; jc < c; jc += TT) perm[jc] = w.pidx[jc];     // colsPermutation().indices()
        return;
    }
    if (s_any) { if (tid == 0) w.st->slow = 1; return; }
    if (tid == 0) w.st->slow = 0;
    choose_and_swap(k + 1, c, pivoting, w, red, ired);
}

// Sweep of step k over every slab.
__device__ __forceinline__ void sweep_body(double* __restrict__ A, int64_t lda, int r, int c, int k, Work w, int vb, double* sm,
                                           int tail = TAIL_NONE, int pivoting = 0, int nslabs = 0)
{
    double* xs = sm;                    // [rows_per] x = column k
    double* xp = sm + w.rows_per;       // [rows_per] x' = next reflector column
    const int g = vb, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r0 = g * w.rows_per, r1 = min(r, r0 + w.rows_per);
    const double inv_s = ld_agent(&w.st->inv_s);
    const int slow = ld_agent(&w.st->slow);
    const int P = ld_agent(&w.st->P);
    const int size = r < c ? r : c;
    const bool last = k + 1 >= size;
    for (int i = r0 + tid; i < r1; i += TT) {
        const double v = A[(int64_t)k * lda + i];
        xs[i - r0] = v;
        if (i > k) A[(int64_t)k * lda + i] = v * inv_s;      // essential part x_tail / (x0 - beta), 0 if H = I
    }
    __syncthreads();
    const int iu = max(r0, k + 1);           // first row the update touches
    if (slow || last) {
        // update only (+ partial squared norms for the recompute)
        for (int jc = k + 1 + wave; jc < c; jc += TW) {
            const double ngam = w.ngamv[jc];
            double s2 = 0.0;
            for (int i = iu + lane; i < r1; i += 64) {
                const double a = fma(ngam, xs[i - r0], A[(int64_t)jc * lda + i]);
                A[(int64_t)jc * lda + i] = a;
                s2 = fma(a, a, s2);
            }
            s2 = wave_sum(s2);
            if (lane == 0) w.sqpart[(int64_t)g * w.cpad + jc] = s2;
        }
        if (tail != TAIL_NONE && slow && !last && slab_is_last(w, nslabs)) recompute_body(c, k, pivoting, w);
        return;
    }
    // fused: the next pivot column P goes to position k+1, everything is updated, dots with x' accumulate
    const int id = max(r0, k + 2);           // first row of the next reflector's tail
    {
        const double ngP = w.ngamv[P], ngK = w.ngamv[k + 1];
        for (int i = r0 + tid; i < r1; i += TT) {
            const double a2 = A[(int64_t)P * lda + i];
            if (i > k) {
                const double x = xs[i - r0];
                const double v = fma(ngP, x, a2);
                xp[i - r0] = v;
                if (P != k + 1) {
                    const double a1 = A[(int64_t)(k + 1) * lda + i];
                    A[(int64_t)P * lda + i] = fma(ngK, x, a1);
                }
                A[(int64_t)(k + 1) * lda + i] = v;
            } else if (P != k + 1) {           // rows of R: plain swap
                const double a1 = A[(int64_t)(k + 1) * lda + i];
                A[(int64_t)(k + 1) * lda + i] = a2;
                A[(int64_t)P * lda + i] = a1;
            }
        }
    }
    __syncthreads();
    if (wave == 0) {
        double t = 0.0;
        for (int i = id + lane; i < r1; i += 64) t = fma(xp[i - r0], xp[i - r0], t);
        t = wave_sum(t);
        if (lane == 0) w.tpart[g] = t;
    }
    // CB columns per wave pass: their loads are independent, so CB row-chunks are in flight per lane (one
    // column at a time the wave waits a full memory round trip per 64 rows)
    constexpr int CB = 4;
    for (int j0 = k + 2 + wave * CB; j0 < c; j0 += TW * CB) {
        double d[CB], ngam[CB];
        bool upd[CB];
#pragma unroll
        for (int u = 0; u < CB; ++u) {
            const int jc = j0 + u;
            d[u] = 0.0;
            upd[u] = jc < c && jc != P;          // column P already holds the updated old column k+1: dots only
            ngam[u] = upd[u] ? w.ngamv[jc] : 0.0;
        }
        for (int i = iu + lane; i < r1; i += 64) {
            double av[CB];
#pragma unroll
            for (int u = 0; u < CB; ++u) av[u] = (j0 + u < c) ? A[(int64_t)(j0 + u) * lda + i] : 0.0;
            const double x = xs[i - r0], x2 = i >= id ? xp[i - r0] : 0.0;
#pragma unroll
            for (int u = 0; u < CB; ++u) {
                const double a = fma(ngam[u], x, av[u]);
                if (upd[u]) A[(int64_t)(j0 + u) * lda + i] = a;
                d[u] = fma(x2, a, d[u]);
            }
        }
#pragma unroll
        for (int u = 0; u < CB; ++u) {
            const double dd = wave_sum(d[u]);
            if (lane == 0 && j0 + u < c) w.partial[(int64_t)g * w.cpad + j0 + u] = dd;
        }
    }
}

// Slow steps only: recomputed norms for the flagged columns, then the pivot of position k+1.
__device__ __forceinline__ void recompute_body(int c, int k, int pivoting, Work w)
{
    __shared__ double red[TW];
    __shared__ int ired[TW];
    if (!ld_agent(&w.st->slow)) return;
    for (int jc = k + 1 + threadIdx.x; jc < c; jc += TT) {
        if (w.need[jc]) {
            double s = 0.0;
            for (int g = 0; g < w.G; ++g) s += w.sqpart[(int64_t)g * w.cpad + jc];
            w.nu2[jc] = s; w.thr[jc] = s * THR_HI;
        }
    }
    __syncthreads();
    choose_and_swap(k + 1, c, pivoting, w, red, ired);
}

// ---- the steps as separate kernels on the caller's stream (fall-back when the slabs cannot all be resident)
__global__ void __launch_bounds__(TT)
norms_kernel(const double* __restrict__ A, int64_t lda, int r, int c, Work w) { norms_body(A, lda, r, c, w, blockIdx.x); }
__global__ void __launch_bounds__(TT)
init_kernel(int c, int pivoting, Work w) { init_body(c, pivoting, w); }
__global__ void __launch_bounds__(TT)
swap_dots_kernel(double* __restrict__ A, int64_t lda, int r, int c, int kpos, int always, Work w)
{
    extern __shared__ double dyn_sm[];
    swap_dots_body(A, lda, r, c, kpos, always, w, blockIdx.x, dyn_sm);
}
__global__ void __launch_bounds__(TT)
head_kernel(double* __restrict__ A, int64_t lda, int r, int c, int k, int pivoting, double* __restrict__ hcoeffs,
            int32_t* __restrict__ perm, Work w)
{
    head_body(A, lda, r, c, k, pivoting, hcoeffs, perm, w, blockIdx.x, gridDim.x);
}
__global__ void __launch_bounds__(TT)
sweep_kernel(double* __restrict__ A, int64_t lda, int r, int c, int k, int pivoting, Work w)
{
    extern __shared__ double dyn_sm[];
    sweep_body(A, lda, r, c, k, w, blockIdx.x, dyn_sm, TAIL_RECOMPUTE, pivoting, gridDim.x);
}

// ---- the same steps inside ONE kernel (opt-in, QRK_DENSE_PERSISTENT=1): a workgroup per slab, all resident (cooperative
// launch, G <= number of CUs), grid barriers where the kernel boundaries were.  The boundaries cost ~8.5 us each (4 per
// reflector), which is why this was tried; measured, the barriers cost MORE: 472 vs 363 ms for the block-angular
// BASELINE shape (40000 x 2000 bottom block), 14.0 vs 12.9 ms at the reference's test size (5120 x 384).  A barrier is
// an atomic counter + generation word behind agent-scope fences, and on this chip every workgroup's release/acquire
// pair writes back and invalidates its XCD's L2 (8 XCDs, 256 workgroups), which a kernel boundary does once.
// Waiting is BOUNDED: a workgroup that does not see the generation change within ~2 s raises bar_abort, after which
// every workgroup runs through the remaining barriers without waiting or computing, so the grid always drains.
constexpr unsigned BAR_SPIN_LIMIT = 1u << 24;

__device__ __forceinline__ bool grid_barrier(State* st, unsigned nblocks)
{
    __shared__ unsigned s_ok;
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned ok = 1u;
        if (__hip_atomic_load(&st->bar_abort, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) ok = 0u;
        else {
            __threadfence();                                   // release: this workgroup's writes
            const unsigned gen = __hip_atomic_load(&st->bar_gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (atomicAdd(&st->bar_count, 1u) == nblocks - 1u) {
                atomicExch(&st->bar_count, 0u);
                __threadfence();
                atomicAdd(&st->bar_gen, 1u);
            } else {
                unsigned spins = 0;
                while (__hip_atomic_load(&st->bar_gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gen) {
                    __builtin_amdgcn_s_sleep(2);
                    if (++spins > BAR_SPIN_LIMIT ||
                        __hip_atomic_load(&st->bar_abort, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
                        atomicExch(&st->bar_abort, 1u);
                        ok = 0u;
                        break;
                    }
                }
            }
            __threadfence();                                   // acquire: the other workgroups' writes
        }
        s_ok = ok;
    }
    __syncthreads();
    return s_ok != 0u;
}

__global__ void __launch_bounds__(TT)
tall_persistent_kernel(double* __restrict__ A, int64_t lda, int r, int c, int pivoting, double* __restrict__ hcoeffs,
                       int32_t* __restrict__ perm, Work w)
{
    extern __shared__ double dyn_sm[];          // [2 * rows_per]
    const int g = blockIdx.x;
    const unsigned nb = gridDim.x;              // = G
    const int size = r < c ? r : c;
    norms_body(A, lda, r, c, w, g);
    if (!grid_barrier(w.st, nb)) return;
    if (g == 0) init_body(c, pivoting, w);
    if (!grid_barrier(w.st, nb)) return;
    swap_dots_body(A, lda, r, c, 0, 1, w, g, dyn_sm);
    if (!grid_barrier(w.st, nb)) return;
    for (int k = 0; k < size; ++k) {
        const int nh0 = (c - k - 1 + HC - 1) / HC, nh = nh0 > 0 ? nh0 : 1;
        for (int hb = g; hb < nh; hb += (int)nb) {
            head_body(A, lda, r, c, k, pivoting, hcoeffs, perm, w, hb, nh);
            __syncthreads();
        }
        if (!grid_barrier(w.st, nb)) return;
        sweep_body(A, lda, r, c, k, w, g, dyn_sm);
        if (!grid_barrier(w.st, nb)) return;
        if (pivoting && k + 1 < size && ld_agent(&w.st->slow)) {      // (uniform over the grid: written before the sweep)
            if (g == 0) recompute_body(c, k, pivoting, w);
            if (!grid_barrier(w.st, nb)) return;
            swap_dots_body(A, lda, r, c, k + 1, 0, w, g, dyn_sm);
            if (!grid_barrier(w.st, nb)) return;
        }
    }
}

// B <- Q^T B or Q B for tall B: the column stays in global memory.
__global__ void __launch_bounds__(TT)
apply_q_kernel(const double* __restrict__ QR, int64_t lda, int r, int nrefl, const double* __restrict__ hcoeffs,
               int transpose, double* __restrict__ B, int64_t ldb, int64_t nrhs)
{
    __shared__ double red[TW];
    const int tid = threadIdx.x;
    for (int64_t col = blockIdx.x; col < nrhs; col += gridDim.x) {
        double* b = B + col * ldb;
        for (int s = 0; s < nrefl; ++s) {
            const int k = transpose ? s : nrefl - 1 - s;
            const double tau = hcoeffs[k];
            const double* v = QR + (int64_t)k * lda;
            double part = 0.0;
            for (int i = k + 1 + tid; i < r; i += TT) part = fma(v[i], b[i], part);
            part = wave_sum(part);
            __syncthreads();
            if ((tid & 63) == 0) red[tid >> 6] = part;
            __syncthreads();
            double tmp = b[k];
#pragma unroll
            for (int wv = 0; wv < TW; ++wv) tmp += red[wv];
            const double tt = tau * tmp;
            __syncthreads();
            if (tid == 0) b[k] -= tt;
            for (int i = k + 1 + tid; i < r; i += TT) b[i] = fma(-tt, v[i], b[i]);
            __syncthreads();
        }
    }
}

}  // namespace tall

size_t dense_tall_workspace_bytes(int r, int c, int num_cus, int* G_out, int* cpad_out, int* rows_per_out)
{
    int G = (r + 127) / 128;
    if (G > num_cus) G = num_cus;
    if (G < 1) G = 1;
    int rows_per = ((r + G - 1) / G + 63) / 64 * 64;
    G = (r + rows_per - 1) / rows_per;
    const int cpad = (c + 63) / 64 * 64;
    *G_out = G; *cpad_out = cpad; *rows_per_out = rows_per;
    return (size_t)(2 * (size_t)G * cpad + G + 3 * cpad) * sizeof(double) + (size_t)2 * cpad * sizeof(int) + 320;
}

// Can the persistent form run: all G workgroups of 1024 threads resident at once (one per CU)?
bool dense_tall_persistent_ok(int G, int rows_per, int num_cus)
{
    int per_cu = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, tall::tall_persistent_kernel, tall::TT,
                                                     2 * (size_t)rows_per * sizeof(double)) != hipSuccess)
        return false;
    return per_cu >= 1 && G <= per_cu * num_cus;
}

// Device address of the "unclear" word of a tall plan's workspace (read by the exact path after the factorisation)
int* dense_tall_unclear_ptr(void* workspace, int G, int cpad)
{
    char* p = static_cast<char*>(workspace);
    p += (size_t)(2 * (size_t)G * cpad + G + 3 * cpad) * sizeof(double) + (size_t)2 * cpad * sizeof(int);
    p = reinterpret_cast<char*>((reinterpret_cast<uintptr_t>(p) + 63) & ~(uintptr_t)63);
    return &reinterpret_cast<tall::State*>(p)->unclear;
}

hipError_t launch_dense_qr_tall(double* A, int64_t lda, int r, int c, int pivoting, double* hcoeffs, int32_t* perm,
                                void* workspace, int G, int cpad, int rows_per, bool persistent, hipStream_t stream)
{
    using namespace tall;
    Work w;
    char* p = static_cast<char*>(workspace);
    w.partial = reinterpret_cast<double*>(p); p += (size_t)G * cpad * sizeof(double);
    w.sqpart = reinterpret_cast<double*>(p); p += (size_t)G * cpad * sizeof(double);
    w.tpart = reinterpret_cast<double*>(p); p += (size_t)G * sizeof(double);
    w.nu2 = reinterpret_cast<double*>(p); p += (size_t)cpad * sizeof(double);
    w.thr = reinterpret_cast<double*>(p); p += (size_t)cpad * sizeof(double);
    w.ngamv = reinterpret_cast<double*>(p); p += (size_t)cpad * sizeof(double);
    w.need = reinterpret_cast<int*>(p); p += (size_t)cpad * sizeof(int);
    w.pidx = reinterpret_cast<int*>(p); p += (size_t)cpad * sizeof(int);
    p = reinterpret_cast<char*>((reinterpret_cast<uintptr_t>(p) + 63) & ~(uintptr_t)63);
    w.st = reinterpret_cast<State*>(p);
    w.G = G; w.cpad = cpad; w.rows_per = rows_per;
    const int size = r < c ? r : c;
    const size_t sm1 = (size_t)rows_per * sizeof(double), sm2 = 2 * sm1;
    if (sm2 > 64 * 1024) {     // very tall slabs: dynamic LDS above the default limit has to be asked for
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(sweep_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm2);
        if (e == hipSuccess && sm1 > 64 * 1024)
            e = hipFuncSetAttribute(reinterpret_cast<const void*>(swap_dots_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm1);
        if (e != hipSuccess) return e;
    }
    if (persistent) {
        // one kernel, a workgroup per slab, all resident (checked by the caller and again by the cooperative launch)
        hipError_t e = hipMemsetAsync(&w.st->bar_count, 0, 3 * sizeof(unsigned), stream);
        if (e != hipSuccess) return e;
        void* args[] = {(void*)&A, (void*)&lda, (void*)&r, (void*)&c, (void*)&pivoting, (void*)&hcoeffs, (void*)&perm, (void*)&w};
        return hipLaunchCooperativeKernel((const void*)tall_persistent_kernel, dim3(G), dim3(TT), args, (unsigned)sm2, stream);
    }
    hipLaunchKernelGGL(norms_kernel, dim3(G), dim3(TT), 0, stream, A, lda, r, c, w);
    hipLaunchKernelGGL(init_kernel, dim3(1), dim3(TT), 0, stream, c, pivoting, w);
    hipLaunchKernelGGL(swap_dots_kernel, dim3(G), dim3(TT), sm1, stream, A, lda, r, c, 0, 1, w);
    for (int k = 0; k < size; ++k) {
        const int nh = (c - k - 1 + HC - 1) / HC;
        hipLaunchKernelGGL(head_kernel, dim3(nh > 0 ? nh : 1), dim3(TT), 0, stream, A, lda, r, c, k, pivoting, hcoeffs, perm, w);
        hipLaunchKernelGGL(sweep_kernel, dim3(G), dim3(TT), sm2, stream, A, lda, r, c, k, pivoting, w);   // (slow: + recompute)
        if (pivoting && k + 1 < size)     // (a no-op unless the step was slow)
            hipLaunchKernelGGL(swap_dots_kernel, dim3(G), dim3(TT), sm1, stream, A, lda, r, c, k + 1, 0, w);
    }
    return hipGetLastError();
}

hipError_t launch_dense_apply_q_tall(const double* QR, int64_t lda, int r, int nrefl, const double* hcoeffs, int transpose,
                                     double* B, int64_t ldb, int64_t nrhs, hipStream_t stream)
{
    if (nrhs <= 0) return hipSuccess;
    const unsigned grid = (unsigned)(nrhs < 4096 ? nrhs : 4096);
    hipLaunchKernelGGL(tall::apply_q_kernel, dim3(grid), dim3(tall::TT), 0, stream, QR, lda, r, nrefl, hcoeffs, transpose, B,
                       ldb, nrhs);
    return hipGetLastError();
}

}  // namespace qrk
