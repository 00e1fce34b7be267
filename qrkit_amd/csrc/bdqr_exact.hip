// bdqr_exact.hip -- the exact-arithmetic path of the per-block QR (gfx950).
//
// The fast kernels (bdqr_pair / bdqr_small / bdqr_col / bdqr_wg) evaluate Eigen's ColPivHouseholderQR with FMA chains,
// squared column norms and an un-normalised reflector: the same mathematics, other roundings.  That is invisible in Q and R
// (1e-15) but NOT in the column permutation whenever two candidate columns are tied in exact arithmetic (sign / indicator /
// repeated-structure Jacobians): Eigen then breaks the tie by the rounding noise of ITS operation order.  Each fast kernel
// therefore checks every decision it takes (pivot choice, LAWN-176 recompute test, degenerate reflector, sign of beta) against
// an error margin and appends the tile to a list when a decision is not clear-cut; this file redoes the listed tiles with the
// reference's arithmetic itself:
//
//   * Eigen/src/QR/ColPivHouseholderQR.h  ColPivHouseholderQR::computeInPlace  (norm tables, first maximum by CURRENT
//     position, column swaps as position bookkeeping, LAWN-176 downdate `normUpd *= sqrt((1+t)(1-t))` / recompute test),
//   * Eigen/src/QR/HouseholderQR.h  householder_qr_inplace_unblocked,
//   * Eigen/src/Householder/Householder.h  makeHouseholder / applyHouseholderOnTheLeft
//     (tmp = essential^T bottom; tmp += row0; row0 -= tau tmp; bottom -= (tau essential) tmp),
//   * Eigen/src/Householder/HouseholderSequence.h  evalTo (Q = I, reflectors applied last to first on the shrinking corner),
//
// one IEEE-754 double operation at a time in Eigen's scalar order (sequential sums over the rows, no FMA contraction,
// correctly rounded division and square root): what a scalar build of Eigen without FMA contraction computes.  The results of
// this path are therefore reproducible bit for bit on any IEEE machine: permutation, tau, R and Q.
// Call site in the reference: blockSolver.compute(block) / matrixQ() / matrixR() / colsPermutation(),
// src/QRKit/BlockDiagonalSparseQR.h:437-447,519-521.
//
// Shape: one workgroup of 256 threads per listed tile; a thread owns whole columns (j, j+256, ...), so every sum over the rows
// is a sequential chain in one thread exactly as in the scalar reference, and the columns run in parallel.  The working copy
// of the tile (row-major: consecutive threads, consecutive addresses) and Q live in LDS when they fit, else in a global
// workspace / directly in the output.  This is the slow path by design (it trades the FMA chains and the register tile for
// reproducibility); generic inputs never reach it.
#include "qrk_device.h"

#include <float.h>

#pragma clang fp contract(off)

#include "bdqr_exact_tile.h"

namespace qrk {

// LDS carve-up: [xbuf maxr][nu maxc][nd maxc][hc maxc][sval T] doubles, [pos maxc][col_at maxc][spos T] ints,
// then `lds_tile_doubles` doubles for W and Q of tiles that fit.
__host__ __device__ static size_t exact_fixed_lds_bytes(int maxr, int maxc)
{
    size_t b = ((size_t)maxr + 3 * (size_t)maxc + exact::T) * sizeof(double);
    b += (2 * (size_t)maxc + exact::T) * sizeof(int);
    return (b + 15) & ~(size_t)15;
}

// ids/count: the tiles to redo (count == nullptr: all nb.num_tiles tiles, ids == nullptr: 0..num_tiles-1).
// The list counter of the NEXT factorisation (next_count) is reset by this kernel, so no memset sits on the stream.
template <bool PIVOT>
__global__ void __launch_bounds__(exact::T)
bdqr_exact_kernel(WaveBatch nb, const int32_t* __restrict__ ids, const int32_t* __restrict__ count, int32_t* next_count,
                  const double* __restrict__ tiles, double* __restrict__ q_vals, double* __restrict__ r_vals,
                  int32_t* __restrict__ perm, double* __restrict__ hcoeffs, double* __restrict__ workspace,
                  int64_t ws_stride, int maxr, int maxc, int lds_tile_doubles)
{
    using namespace exact;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    Shared sh;
    carve_shared<T>(smem, maxr, maxc, sh);
    const int64_t total_tiles = nb.num_tiles;
    double* lds_tile = reinterpret_cast<double*>(smem + exact_fixed_lds_bytes(maxr, maxc));

    if (next_count && blockIdx.x == 0 && threadIdx.x == 0) *next_count = 0;
    // (a list can never be longer than the batch, nor name a tile outside it: a corrupted counter must not walk the workgroups
    // out of the arrays)
    int64_t n = count ? (int64_t)*count : nb.num_tiles;
    if (n > total_tiles) n = total_tiles;
    for (int64_t li = blockIdx.x; li < n; li += gridDim.x) {
        const int64_t t = ids ? (int64_t)ids[li] : li;
        if (t < 0 || t >= total_tiles) continue;
        int r, c, cbase;
        int64_t toff, qoff, roff;
        if (nb.t_rows) {
            r = nb.t_rows[t]; c = nb.t_cols[t];
            toff = nb.t_off[t]; qoff = nb.q_off[t]; roff = nb.r_off[t]; cbase = nb.c_off[t];
        } else {
            r = nb.rows; c = nb.cols;
            toff = t * (int64_t)r * c; qoff = t * (int64_t)r * r; roff = t * (int64_t)(c * (c + 1) / 2);
            cbase = (int)(t * c);
        }
        const bool in_lds = (int64_t)r * c + (int64_t)r * r <= lds_tile_doubles;
        double* W = in_lds ? lds_tile : workspace + (size_t)blockIdx.x * ws_stride;
        double* q = in_lds ? lds_tile + (size_t)r * c : q_vals + qoff;
        tile_qr<PIVOT, T>(r, c, tiles + toff, W, q, sh);
        tile_store<T>(r, c, cbase, W, in_lds ? q : nullptr, sh, perm, hcoeffs, r_vals + roff, q_vals + qoff);
        __syncthreads();
    }
}

hipError_t launch_bdqr_exact(const WaveBatch& nb, const int32_t* ids, const int32_t* count, int32_t* next_count,
                             const double* tiles, double* q_vals, double* r_vals, int32_t* perm, double* hcoeffs,
                             double* workspace, int64_t ws_stride, int num_wg, int maxr, int maxc, hipStream_t stream)
{
    if (num_wg <= 0) return hipSuccess;
    const size_t fixed = exact_fixed_lds_bytes(maxr, maxc);
    // W and Q of a tile in LDS when both fit beside the fixed part in 64 KB (tiles up to ~60 x 60), else global memory
    const size_t budget = 64 * 1024;
    int64_t tile_doubles = fixed < budget ? (int64_t)((budget - fixed) / sizeof(double)) : 0;
    const int64_t want = (int64_t)maxr * maxc + (int64_t)maxr * maxr;
    if (tile_doubles > want) tile_doubles = want;
    if (!workspace && want > tile_doubles) return hipErrorInvalidValue;   // (the plan sizes the workspace for its largest tile)
    const size_t smem = fixed + (size_t)tile_doubles * sizeof(double);
    const dim3 grid((unsigned)num_wg), block(exact::T);
    if (smem > 64 * 1024) {    // (tiles with thousands of rows/columns: the fixed part alone passes the default limit)
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&bdqr_exact_kernel<true>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        if (e == hipSuccess)
            e = hipFuncSetAttribute(reinterpret_cast<const void*>(&bdqr_exact_kernel<false>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        if (e != hipSuccess) return e;
    }
    if (nb.pivoting)
        hipLaunchKernelGGL(bdqr_exact_kernel<true>, grid, block, smem, stream, nb, ids, count, next_count, tiles, q_vals, r_vals,
                           perm, hcoeffs, workspace, ws_stride, maxr, maxc, (int)tile_doubles);
    else
        hipLaunchKernelGGL(bdqr_exact_kernel<false>, grid, block, smem, stream, nb, ids, count, next_count, tiles, q_vals, r_vals,
                           perm, hcoeffs, workspace, ws_stride, maxr, maxc, (int)tile_doubles);
    return hipGetLastError();
}

// ---- the dense solver's exact path -------------------------------------------------------------------------------------
// The right-block solver of BlockAngularSparseQR (Eigen ColPivHouseholderQR / HouseholderQR on one dense matrix with implicit
// Q, src/QRKit/BlockAngularSparseQR.h:361-369) in Eigen's operation order, in place in the caller's column-major array with
// Eigen's physical column swaps: ONE workgroup of 1024 threads, a thread owns whole columns (every sum over the rows is the
// sequential chain of the scalar reference), the pivot column / essential vector, the two norm tables and the permutation
// in a global workspace.  Launched after every dense factorisation and a no-op unless the fast kernels flagged a decision
// (then the matrix is first restored from the copy the plan keeps).  Slow by design: the fall-back for inputs whose pivot
// order is decided by rounding noise.
namespace exact {
constexpr int DT = 1024;

template <bool PIVOT>
__global__ void __launch_bounds__(DT)
dense_exact_kernel(double* __restrict__ A, int64_t lda, int r, int c, const double* __restrict__ copy, double* __restrict__ hcoeffs,
                   int32_t* __restrict__ perm, const int* __restrict__ unclear, double* __restrict__ ws)
{
    if (unclear && *unclear == 0) return;
    __shared__ double sval[DT];
    __shared__ int spos[DT];
    const int t = threadIdx.x;
    double* xbuf = ws;                    // [r]
    double* nu = xbuf + r;                // [c] m_colNormsUpdated
    double* nd = nu + c;                  // [c] m_colNormsDirect
    int* pidx = reinterpret_cast<int*>(nd + c);   // [c]
    double* pivn = reinterpret_cast<double*>(pidx + ((c + 1) & ~1));   // [c] m_colNormsUpdated of the chosen column, per step (wide_ws below)
    if (copy)
        for (int64_t e = t; e < (int64_t)r * c; e += DT) { const int64_t j = e / r; A[j * lda + (e - j * r)] = copy[e]; }
    for (int j = t; j < c; j += DT) pidx[j] = j;
    __syncthreads();
    if (PIVOT) {
        for (int j = t; j < c; j += DT) {
            const double* col = A + (int64_t)j * lda;
            double s = 0.0;
            for (int i = 0; i < r; ++i) { const double v = col[i]; s += v * v; }
            const double n = sqrt(s);
            nu[j] = n; nd[j] = n;
        }
        __syncthreads();
    }
    const int size = r < c ? r : c;
    for (int k = 0; k < size; ++k) {
        if (PIVOT) {
            double bv = -1.0; int bp = 0x7fffffff;
            for (int j = k + t; j < c; j += DT) if (better(nu[j], j, bv, bp)) { bv = nu[j]; bp = j; }
            sval[t] = bv; spos[t] = bp;
            __syncthreads();
            for (int s = DT / 2; s > 0; s >>= 1) {
                if (t < s && better(sval[t + s], spos[t + s], sval[t], spos[t])) { sval[t] = sval[t + s]; spos[t] = spos[t + s]; }
                __syncthreads();
            }
            const int b = spos[0];
            if (t == 0) pivn[k] = sval[0];        // (Eigen: biggest_col_sq_norm = abs2 of this; nonzeroPivots() is counted from it)
            __syncthreads();
            if (b != k) {       // m_qr.col(k).swap(m_qr.col(b)), the two norm tables, the transposition
                double* ck = A + (int64_t)k * lda;
                double* cb = A + (int64_t)b * lda;
                for (int i = t; i < r; i += DT) { const double v = ck[i]; ck[i] = cb[i]; cb[i] = v; }
                if (t == 0) {
                    double v = nu[k]; nu[k] = nu[b]; nu[b] = v;
                    v = nd[k]; nd[k] = nd[b]; nd[b] = v;
                    const int p = pidx[k]; pidx[k] = pidx[b]; pidx[b] = p;
                }
            }
            __syncthreads();
        }
        double* ck = A + (int64_t)k * lda;
        for (int i = k + t; i < r; i += DT) xbuf[i] = ck[i];
        __syncthreads();
        const double c0 = xbuf[k];
        double tail = 0.0;
        for (int i = k + 1; i < r; ++i) { const double v = xbuf[i]; tail += v * v; }
        double tau, beta, denom = 1.0;
        const bool degen = tail <= DBL_MIN;
        if (degen) { tau = 0.0; beta = c0; }
        else {
            beta = sqrt(c0 * c0 + tail);
            if (c0 >= 0.0) beta = -beta;
            denom = c0 - beta;
            tau = (beta - c0) / beta;
        }
        __syncthreads();
        for (int i = k + 1 + t; i < r; i += DT) { const double e = degen ? 0.0 : xbuf[i] / denom; xbuf[i] = e; ck[i] = e; }
        if (t == 0) { ck[k] = beta; hcoeffs[k] = tau; }
        __syncthreads();
        const int m = r - k;
        for (int j = k + 1 + t; j < c; j += DT) {
            double* col = A + (int64_t)j * lda;
            if (m == 1) col[k] *= (1.0 - tau);
            else if (tau != 0.0) {
                double tmp = 0.0;
                for (int i = k + 1; i < r; ++i) tmp += xbuf[i] * col[i];
                tmp += col[k];
                col[k] -= tau * tmp;
                for (int i = k + 1; i < r; ++i) col[i] -= (tau * xbuf[i]) * tmp;
            }
            if (PIVOT) {
                const double nuj = nu[j];
                if (nuj != 0.0) {
                    double temp = fabs(col[k]) / nuj;
                    temp = (1.0 + temp) * (1.0 - temp);
                    temp = temp < 0.0 ? 0.0 : temp;
                    const double ratio = nuj / nd[j];
                    const double temp2 = temp * (ratio * ratio);
                    if (temp2 <= SQRT_EPS) {
                        double s = 0.0;
                        for (int i = k + 1; i < r; ++i) { const double v = col[i]; s += v * v; }
                        const double n = sqrt(s);
                        nd[j] = n; nu[j] = n;
                    } else nu[j] = nuj * sqrt(temp);
                }
            }
        }
        __syncthreads();
    }
    for (int j = t; j < c; j += DT) perm[j] = pidx[j];
}
}  // namespace exact

// ---- the same, wide: for blocks where one workgroup would take minutes (40 000 x 2 000: 200 s).  The arithmetic of a column is
// a sequential chain in one thread, as above - that is what makes the result Eigen's, bit for bit - but the columns are
// independent, so they are spread over the chip: per reflector one `head` launch (one workgroup: pivot search, Eigen's
// physical column swap, |x_tail|^2 as one chain, beta / tau, the essential vector) and one `apply` launch (a thread per
// remaining column, 16 columns per workgroup so that a wave touches 16 cache lines per access, not 64; the LAWN-176 update of
// its norms included).  Launched by the host only after it has read the `unclear` word (the sequence is 2 launches per
// reflector: not something to queue as no-ops behind every factorisation).
namespace exact {
constexpr int WIDE_COLS = 16;
constexpr int HT = 256;       // threads of the head kernel (256 VGPRs each: the two batches of the chain stay in registers)
constexpr int WU = 32;        // loads issued together, two batches in flight; the sums stay ONE sequential chain in row order

#define QRK_WU_LOAD(dst, src, at) _Pragma("unroll") for (int u = 0; u < WU; ++u) dst[u] = src[(at) + u]
// Walks rows i0 .. r-1 of y in batches of WU with the next batch already in flight; BODY(buf, i) consumes batch `buf` whose first
// row is i, TAIL(i) a single row.  (The columns of a 40 000 x 2 000 block come from HBM: without the second batch in flight the
// chains run at the memory latency, 30 s per factorisation instead of a few.)
#define QRK_WU_STREAM(y, i0, r, BODY, TAIL)                                          \
    do {                                                                             \
        int i_ = (i0);                                                               \
        const int nb_ = ((r) - (i0)) > 0 ? ((r) - (i0)) / WU : 0;                    \
        double b0_[WU], b1_[WU];                                                     \
        if (nb_ > 0) { QRK_WU_LOAD(b0_, y, i_); }                                    \
        int t_ = 0;                                                                  \
        for (; t_ + 2 <= nb_; t_ += 2) {                                             \
            QRK_WU_LOAD(b1_, y, i_ + WU);                                            \
            BODY(b0_, i_);                                                           \
            i_ += WU;                                                                \
            if (t_ + 2 < nb_) { QRK_WU_LOAD(b0_, y, i_ + WU); }                      \
            BODY(b1_, i_);                                                           \
            i_ += WU;                                                                \
        }                                                                            \
        if (t_ < nb_) { BODY(b0_, i_); i_ += WU; }                                   \
        for (; i_ < (r); ++i_) { TAIL(i_); }                                         \
    } while (0)

// s += v[i]^2, i = i0 .. r-1, in that order
__device__ __forceinline__ double chain_sumsq(const double* v, int i0, int r, double s)
{
#define QRK_BODY(buf, i) _Pragma("unroll") for (int u = 0; u < WU; ++u) s += buf[u] * buf[u]
#define QRK_TAIL(i) { const double a = v[i]; s += a * a; }
    QRK_WU_STREAM(v, i0, r, QRK_BODY, QRK_TAIL);
#undef QRK_BODY
#undef QRK_TAIL
    return s;
}
// s += x[i] * y[i], i = i0 .. r-1, in that order (y streamed from memory, x the shared reflector: cache hits)
__device__ __forceinline__ double chain_dot(const double* x, const double* y, int i0, int r, double s)
{
#define QRK_BODY(buf, i) { double a[WU]; QRK_WU_LOAD(a, x, i); _Pragma("unroll") for (int u = 0; u < WU; ++u) s += a[u] * buf[u]; }
#define QRK_TAIL(i) s += x[i] * y[i]
    QRK_WU_STREAM(y, i0, r, QRK_BODY, QRK_TAIL);
#undef QRK_BODY
#undef QRK_TAIL
    return s;
}
// y[i] -= (tau x[i]) tmp, i = i0 .. r-1
__device__ __forceinline__ void chain_update(const double* x, double* y, int i0, int r, double tau, double tmp)
{
#define QRK_BODY(buf, i) { double a[WU]; QRK_WU_LOAD(a, x, i); _Pragma("unroll") for (int u = 0; u < WU; ++u) y[(i) + u] = buf[u] - (tau * a[u]) * tmp; }
#define QRK_TAIL(i) y[i] -= (tau * x[i]) * tmp
    QRK_WU_STREAM(y, i0, r, QRK_BODY, QRK_TAIL);
#undef QRK_BODY
#undef QRK_TAIL
}

// pivn [c]: m_colNormsUpdated of the column chosen at every step (the biggest one, Eigen's biggest_col_sq_norm before squaring): what
// nonzeroPivots() is counted from (ColPivHouseholderQR::computeInPlace; BlockedThinSparseQR.h:250-256 reads it)
struct WideWs { double* xbuf; double* nu; double* nd; int* pidx; double* pivn; };
__host__ __device__ inline WideWs wide_ws(double* ws, int r, int c)
{
    WideWs w;
    w.xbuf = ws; w.nu = ws + r; w.nd = w.nu + c; w.pidx = reinterpret_cast<int*>(w.nd + c);
    w.pivn = reinterpret_cast<double*>(w.pidx + ((c + 1) & ~1));
    return w;
}

template <bool PIVOT>
__global__ void __launch_bounds__(64)
dense_exact_wide_init_kernel(const double* __restrict__ A, int64_t lda, int r, int c, double* __restrict__ ws)
{
    const WideWs w = wide_ws(ws, r, c);
    const int j = blockIdx.x * WIDE_COLS + threadIdx.x;
    if (threadIdx.x >= WIDE_COLS || j >= c) return;
    w.pidx[j] = j;
    if (PIVOT) {
        const double* col = A + (int64_t)j * lda;
        const double n = sqrt(chain_sumsq(col, 0, r, 0.0));
        w.nu[j] = n; w.nd[j] = n;
    }
}

template <bool PIVOT>
__global__ void __launch_bounds__(HT)
dense_exact_wide_head_kernel(double* __restrict__ A, int64_t lda, int r, int c, int k, double* __restrict__ hcoeffs,
                             int32_t* __restrict__ perm, double* __restrict__ ws)
{
    __shared__ double sval[HT];
    __shared__ int spos[HT];
    const int t = threadIdx.x;
    const WideWs w = wide_ws(ws, r, c);
    double* xbuf = w.xbuf; double* nu = w.nu; double* nd = w.nd; int* pidx = w.pidx;
    if (PIVOT) {
        double bv = -1.0; int bp = 0x7fffffff;
        for (int j = k + t; j < c; j += HT) if (better(nu[j], j, bv, bp)) { bv = nu[j]; bp = j; }
        sval[t] = bv; spos[t] = bp;
        __syncthreads();
        for (int s = HT / 2; s > 0; s >>= 1) {
            if (t < s && better(sval[t + s], spos[t + s], sval[t], spos[t])) { sval[t] = sval[t + s]; spos[t] = spos[t + s]; }
            __syncthreads();
        }
        const int b = spos[0];
        if (t == 0) w.pivn[k] = sval[0];
        __syncthreads();
        if (b != k) {       // m_qr.col(k).swap(m_qr.col(b)), the two norm tables, the transposition
            double* ck = A + (int64_t)k * lda;
            double* cb = A + (int64_t)b * lda;
            for (int i = t; i < r; i += HT) { const double v = ck[i]; ck[i] = cb[i]; cb[i] = v; }
            if (t == 0) {
                double v = nu[k]; nu[k] = nu[b]; nu[b] = v;
                v = nd[k]; nd[k] = nd[b]; nd[b] = v;
                const int p = pidx[k]; pidx[k] = pidx[b]; pidx[b] = p;
            }
        }
        __syncthreads();
    }
    double* ck = A + (int64_t)k * lda;
    for (int i = k + t; i < r; i += HT) xbuf[i] = ck[i];
    __syncthreads();
    const double c0 = xbuf[k];
    // (every thread runs the same chain on the same data, as in the one-workgroup kernel: no broadcast needed)
    const double tail = chain_sumsq(xbuf, k + 1, r, 0.0);
    double tau, beta, denom = 1.0;
    const bool degen = tail <= DBL_MIN;
    if (degen) { tau = 0.0; beta = c0; }
    else {
        beta = sqrt(c0 * c0 + tail);
        if (c0 >= 0.0) beta = -beta;
        denom = c0 - beta;
        tau = (beta - c0) / beta;
    }
    __syncthreads();
    for (int i = k + 1 + t; i < r; i += HT) { const double e = degen ? 0.0 : xbuf[i] / denom; xbuf[i] = e; ck[i] = e; }
    if (t == 0) { ck[k] = beta; hcoeffs[k] = tau; }
    const int size = r < c ? r : c;
    if (k == size - 1) {
        __syncthreads();
        for (int j = t; j < c; j += HT) perm[j] = pidx[j];
    }
}

template <bool PIVOT>
__global__ void __launch_bounds__(64)
dense_exact_wide_apply_kernel(double* __restrict__ A, int64_t lda, int r, int c, int k, const double* __restrict__ hcoeffs,
                              double* __restrict__ ws)
{
    const WideWs w = wide_ws(ws, r, c);
    const double* xbuf = w.xbuf; double* nu = w.nu; double* nd = w.nd;
    const int j = k + 1 + blockIdx.x * WIDE_COLS + threadIdx.x;
    if (threadIdx.x >= WIDE_COLS || j >= c) return;
    const double tau = hcoeffs[k];
    const int m = r - k;
    double* col = A + (int64_t)j * lda;
    if (m == 1) col[k] *= (1.0 - tau);
    else if (tau != 0.0) {
        double tmp = chain_dot(xbuf, col, k + 1, r, 0.0);
        tmp += col[k];
        col[k] -= tau * tmp;
        chain_update(xbuf, col, k + 1, r, tau, tmp);
    }
    if (PIVOT) {
        const double nuj = nu[j];
        if (nuj != 0.0) {
            double temp = fabs(col[k]) / nuj;
            temp = (1.0 + temp) * (1.0 - temp);
            temp = temp < 0.0 ? 0.0 : temp;
            const double ratio = nuj / nd[j];
            const double temp2 = temp * (ratio * ratio);
            if (temp2 <= SQRT_EPS) {
                const double n = sqrt(chain_sumsq(col, k + 1, r, 0.0));
                nd[j] = n; nu[j] = n;
            } else nu[j] = nuj * sqrt(temp);
        }
    }
}
}  // namespace exact

// A (restored by the caller) -> Eigen's packed QR, hcoeffs, perm; workspace as for launch_dense_exact
hipError_t launch_dense_exact_wide(double* A, int64_t lda, int r, int c, int pivoting, double* hcoeffs, int32_t* perm,
                                   double* workspace, hipStream_t stream)
{
    using namespace exact;
    const int size = r < c ? r : c;
    const unsigned gi = (unsigned)((c + WIDE_COLS - 1) / WIDE_COLS);
#define QRK_WIDE(PV)                                                                                                         \
    do {                                                                                                                     \
        hipLaunchKernelGGL(dense_exact_wide_init_kernel<PV>, dim3(gi), dim3(64), 0, stream, A, lda, r, c, workspace);        \
        for (int k = 0; k < size; ++k) {                                                                                     \
            hipLaunchKernelGGL(dense_exact_wide_head_kernel<PV>, dim3(1), dim3(HT), 0, stream, A, lda, r, c, k, hcoeffs, perm, workspace); \
            const int rest = c - k - 1;                                                                                      \
            if (rest > 0)                                                                                                    \
                hipLaunchKernelGGL(dense_exact_wide_apply_kernel<PV>, dim3((unsigned)((rest + WIDE_COLS - 1) / WIDE_COLS)), dim3(64), 0, \
                                   stream, A, lda, r, c, k, hcoeffs, workspace);                                             \
        }                                                                                                                    \
    } while (0)
    if (pivoting) QRK_WIDE(true); else QRK_WIDE(false);
#undef QRK_WIDE
    return hipGetLastError();
}

size_t dense_exact_workspace_bytes(int r, int c) { return ((size_t)r + 3 * (size_t)c) * sizeof(double) + (size_t)(c + 1) * sizeof(int) + 64; }
const double* dense_exact_pivot_norms(const double* workspace, int r, int c) { return exact::wide_ws(const_cast<double*>(workspace), r, c).pivn; }

hipError_t launch_dense_exact(double* A, int64_t lda, int r, int c, int pivoting, const double* copy, double* hcoeffs,
                              int32_t* perm, const int* unclear, double* workspace, hipStream_t stream)
{
    if (pivoting)
        hipLaunchKernelGGL(exact::dense_exact_kernel<true>, dim3(1), dim3(exact::DT), 0, stream, A, lda, r, c, copy, hcoeffs, perm,
                           unclear, workspace);
    else
        hipLaunchKernelGGL(exact::dense_exact_kernel<false>, dim3(1), dim3(exact::DT), 0, stream, A, lda, r, c, copy, hcoeffs, perm,
                           unclear, workspace);
    return hipGetLastError();
}

// Whether launch_bdqr_exact needs a global workspace for tiles of up to maxr x maxc.
bool bdqr_exact_needs_workspace(int maxr, int maxc)
{
    const size_t fixed = exact_fixed_lds_bytes(maxr, maxc);
    const size_t budget = 64 * 1024;
    const int64_t have = fixed < budget ? (int64_t)((budget - fixed) / sizeof(double)) : 0;
    return (int64_t)maxr * maxc + (int64_t)maxr * maxr > have;
}

}  // namespace qrk
