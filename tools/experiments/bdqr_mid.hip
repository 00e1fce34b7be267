// EXPERIMENT (round 2), not part of the product build: measured, it does not beat bdqr_col.hip except on 57..64-column tiles in
// large batches (64 x 64: 2.30 vs 1.54 M tiles/s at 10 000 tiles, 1.05 vs 1.42 at 2 000; 48 x 48: 4.1 vs 4.1; 33 x 33: 6.3 vs 10.0).
// s_memtime stamps: ~5 900 cycles per reflector with 2 waves per SIMD -- pivot search and publishing x 2 200 (before the DPP search),
// partial dots 1 200, reflector scalars + update 1 300, the two barriers 1 200: every wave repeats ~250 instructions of
// bookkeeping per step for 64-96 FMAs of its own.  A K1-grade version (scalars by one wave, DPP broadcasts, packed I/O) is what it
// would take.  To try it: copy to qrkit_amd/csrc/, declare bdqr_mid_supported / launch_bdqr_mid in qrk_device.h and call it at the
// top of launch_bdqr_col when bdqr_mid_supported(max_rows, max_cols).
//
// bdqr_mid.hip -- one workgroup of 2 x (64 / RW) wavefronts factorises one tile with 32 < max(rows, cols) <= 64 (rows >= cols) entirely in
// registers: A_i P_i = Q_i R_i with explicit Q_i, for gfx950.
//
// Same reference seam as bdqr_pair.hip / bdqr_col.hip (the hot loop of BlockDiagonalSparseQR::factorize,
// src/QRKit/BlockDiagonalSparseQR.h:432-526: Eigen ColPivHouseholderQR / HouseholderQR behind blockSolver.compute,
// HouseholderSequence behind matrixQ()).  bdqr_col.hip gives such a tile ONE wavefront and keeps the matrix in LDS (a thread per
// column walks down its column through LDS): with 36 KB per tile four tiles fit a CU, one wave per SIMD, and every LDS round trip
// is exposed -- 64 x 64 tiles ran at 1.5 M tiles/s, 10 us per reflector.  Here the tile lives in the register file, K1's way, spread
// over a workgroup:
//   * waves 0..3 hold [A]: wave w rows 16 w .. 16 w + 15, lane j column j -- 16 row registers per thread;
//     waves 4..7 hold Q^T the same way, starting from the identity: every reflector is applied to [A | I] (H_{c-1} ... H_0 I = Q^T);
//   * per reflector TWO barriers: the threads of the pivot column publish x in LDS | every thread forms the partial dot product of
//     its 16 rows with its column (and |x_tail|^2 of its rows), the wave that owns row k publishes row k | the four partial sums
//     of a column are added in a fixed order by every thread that needs them, so that the reflector scalars, row k of R, the
//     LAWN-176 downdate of the squared column norms and the next pivot are computed redundantly and identically in every wave
//     (nothing else crosses waves), then the rank-1 update of the 16 registers;
//   * squared norms, un-normalised reflector (nb = copysign(norm, x0), s = nb + x0, ng = -1 / (nb s)) and the decision margins
//     of the other fast kernels (qrk_device.h, namespace decide): a pivot inside the margin, a degenerate reflector, a sign of
//     beta at the noise level or a column norm that passes Eigen's recompute test (the recomputation itself is not done here:
//     generic tiles never get there) sends the tile to the exact path (bdqr_exact.hip).
// Outputs as bdqr_col.hip: Q row-major (CSR value order of m_Q), R packed by columns (CSC value order of m_R), the permutation
// splice, tau when asked for.
#include "qrk_device.h"

#include <float.h>
#include <cstdlib>

namespace qrk {

namespace mid {

using namespace decide;
#ifndef QRK_MID_RW
#define QRK_MID_RW 32
#endif
constexpr int RW = QRK_MID_RW;          // rows per wave (registers per thread)
constexpr int HW = 64 / RW;             // waves that hold [A]; as many hold Q^T
constexpr int T = 2 * HW * 64;          // threads
constexpr int MAXD = 64;      // largest tile dimension

struct Best { double val; int pos; int lane; };

// sqrt and reciprocal from the v_rsq / v_rcp seeds (<= 1 ulp; as in bdqr_pair.hip and banded.hip): the division sequences of the
// compiler are ~40 instructions each, and every wave evaluates the reflector scalars itself
__device__ __forceinline__ double fast_sqrt(double x)
{
    const double y = __builtin_amdgcn_rsq(x);
    double g = x * y, h = 0.5 * y;
    const double e = fma(-h, g, 0.5);
    g = fma(g, e, g);
    h = fma(h, e, h);
    const double d = fma(-g, g, x);
    return fma(d, h, g);
}
__device__ __forceinline__ double fast_recip(double x)
{
    double y = __builtin_amdgcn_rcp(x);
    double e = fma(-x, y, 1.0);
    y = fma(y, e, y);
    e = fma(-x, y, 1.0);
    y = fma(y, e, y);
    return y;
}

// First maximum of the 64 lanes by CURRENT position (Eigen's tie rule) without LDS round trips: the maximum by DPP steps inside the
// rows of 16 and four readlanes, the smallest position among the lanes that hold it the same way, the lane by a ballot.
template <int CTRL>
__device__ __forceinline__ int dpp_int(int v) { return __builtin_amdgcn_update_dpp(v, v, CTRL, 0xF, 0xF, false); }

__device__ __forceinline__ Best wave_best(Best b)
{
    double m = row16_max(b.val);
    m = fmax(fmax(readlane_f64(m, 0), readlane_f64(m, 16)), fmax(readlane_f64(m, 32), readlane_f64(m, 48)));
    int p = b.val == m ? b.pos : 0x7fffffff;
    p = min(p, dpp_int<0xB1>(p));
    p = min(p, dpp_int<0x4E>(p));
    p = min(p, dpp_int<0x141>(p));
    p = min(p, dpp_int<0x140>(p));
    p = min(min(__builtin_amdgcn_readlane(p, 0), __builtin_amdgcn_readlane(p, 16)),
            min(__builtin_amdgcn_readlane(p, 32), __builtin_amdgcn_readlane(p, 48)));
    const unsigned long long who = __ballot(b.val == m && b.pos == p);
    Best o;
    o.val = m; o.pos = p; o.lane = (int)__builtin_ctzll(who);
    return o;
}

template <bool PIVOT>
__global__ void __launch_bounds__(T) __attribute__((amdgpu_waves_per_eu(RW == 32 ? 2 : 4, RW == 32 ? 2 : 4)))
bdqr_mid_kernel(WaveBatch nb, const double* __restrict__ tiles, double* __restrict__ q_vals, double* __restrict__ r_vals,
                int32_t* __restrict__ perm, double* __restrict__ hcoeffs, int32_t* __restrict__ redo_count,
                int32_t* __restrict__ redo_ids, int32_t* __restrict__ queue)
{
    __shared__ double xs[MAXD];              // the pivot column
    __shared__ double pd[2 * HW][MAXD];      // partial dot products of the waves
    __shared__ double akr[2][MAXD];          // row k of [A], of [Q^T]
    __shared__ double tsqp[HW];              // partial |x_tail|^2
    __shared__ double rrow[MAXD * MAXD];     // rows of R as they are finished: rrow[k * 64 + column]
    __shared__ int col_of_pos[MAXD];
    __shared__ int s_unclear, next_tile;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const bool isQ = wave >= HW;
    const int hw = isQ ? wave - HW : wave;   // index of the wave inside its half
    const int rw = hw * RW;                  // first row of this thread
    const int j = lane;

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
        // ---- [A | I] into the registers (all loads in flight: clamped addresses, values replaced afterwards)
        double a[RW];
        {
            const double* base = tiles + toff;
            const double* src = base + (int64_t)j * r + rw;
#pragma unroll
            for (int i = 0; i < RW; ++i) { const bool in = !isQ && j < c && rw + i < r; a[i] = *(in ? src + i : base); }
#pragma unroll
            for (int i = 0; i < RW; ++i) {
                const bool in = !isQ && j < c && rw + i < r;
                a[i] = in ? a[i] : ((isQ && rw + i == j && j < r) ? 1.0 : 0.0);
            }
        }
        bool live = !isQ && j < c;
        int pos = j;
        double nu2 = -1.0, thr = 0.0, a2 = 0.0;
        bool unclear = false;
        if (tid == 0) s_unclear = 0;
        if (PIVOT) {
            double s0 = 0.0;
#pragma unroll
            for (int i = 0; i < RW; ++i) s0 = fma(a[i], a[i], s0);
            pd[wave][j] = s0;
            __syncthreads();
            if (live) { double sn = pd[0][j]; for (int w2 = 1; w2 < HW; ++w2) sn += pd[w2][j]; nu2 = sn; thr = nu2 * THR_HI; }
            __syncthreads();
        }

#ifdef QRK_MID_PROF
        unsigned long long pt[6] = {0, 0, 0, 0, 0, 0}, pt0 = __builtin_amdgcn_s_memtime();
#define MID_TICK(z) do { const unsigned long long t1 = __builtin_amdgcn_s_memtime(); pt[z] += t1 - pt0; pt0 = t1; } while (0)
#else
#define MID_TICK(z) do { } while (0)
#endif
        for (int k = 0; k < c; ++k) {
            int P = k;
            MID_TICK(5);
            if (!isQ) {
                if (PIVOT) {
                    Best b{live ? nu2 : -1.0, pos, lane};
                    b = wave_best(b);
                    P = b.lane;
                    if (k == 0) a2 = b.val;
                    {   // decision (1), decide::near_best with the fast square root
                        const double margin = MREL * (thr + THR_HI * a2) + 4.547473508864641e-13 /* 2^-41 */ * fast_sqrt(a2 * (b.val > 0.0 ? b.val : 0.0));
                        if (live && lane != P && nu2 >= b.val - margin) unclear = true;
                    }
                    if (lane == P) pos = k; else if (pos == k) pos = b.pos;                       // Eigen's transposition
                }
                if (lane == P) {
#pragma unroll
                    for (int i = 0; i < RW; ++i) xs[rw + i] = a[i];
                    if (wave == 0) col_of_pos[k] = P;
                }
            }
            MID_TICK(0);
            __syncthreads();
            MID_TICK(1);
            // ---- partial dot products with x (rows below k), |x_tail|^2, row k.  k is uniform: a wave whose rows are all below
            // row k needs no masks, one whose rows are all finished does nothing but keep the barriers
            const int where = rw > k ? 0 : (rw + RW - 1 < k ? 2 : 1);      // 0: all rows take part, 1: row k is here, 2: finished
            double xm[RW];
            const double xk = xs[k];
            double pdv = 0.0, tq = 0.0, akv = 0.0;
            if (where == 0) {
#pragma unroll
                for (int i = 0; i < RW; ++i) xm[i] = xs[rw + i];
                double p4[4] = {0.0, 0.0, 0.0, 0.0}, q4[4] = {0.0, 0.0, 0.0, 0.0};      // four chains each
#pragma unroll
                for (int i = 0; i < RW; ++i) { p4[i & 3] = fma(xm[i], a[i], p4[i & 3]); q4[i & 3] = fma(xm[i], xm[i], q4[i & 3]); }
                pdv = (p4[0] + p4[1]) + (p4[2] + p4[3]); tq = (q4[0] + q4[1]) + (q4[2] + q4[3]);
            } else if (where == 1) {
#pragma unroll
                for (int i = 0; i < RW; ++i) { const double xv = xs[rw + i]; xm[i] = rw + i > k ? xv : 0.0; }
                double p4[4] = {0.0, 0.0, 0.0, 0.0}, q4[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
                for (int i = 0; i < RW; ++i) {
                    p4[i & 3] = fma(xm[i], a[i], p4[i & 3]);
                    q4[i & 3] = fma(xm[i], xm[i], q4[i & 3]);
                    akv = rw + i == k ? a[i] : akv;
                }
                pdv = (p4[0] + p4[1]) + (p4[2] + p4[3]); tq = (q4[0] + q4[1]) + (q4[2] + q4[3]);
                akr[isQ ? 1 : 0][j] = akv;
            } else {
#pragma unroll
                for (int i = 0; i < RW; ++i) xm[i] = 0.0;
            }
            pd[wave][j] = pdv;
            if (!isQ && lane == 0) tsqp[wave] = tq;
            MID_TICK(2);
            __syncthreads();
            MID_TICK(3);
            double d = pd[isQ ? HW : 0][j], tsq = tsqp[0];
#pragma unroll
            for (int w2 = 1; w2 < HW; ++w2) { d += pd[(isQ ? HW : 0) + w2][j]; tsq += tsqp[w2]; }
            const double ak = akr[isQ ? 1 : 0][j];
            // ---- makeHouseholder, un-normalised (bdqr_pair.hip): degenerate -> H = I
            if (k == 0 && !PIVOT) a2 = fma(xk, xk, tsq);
            if (!isQ && unclear_reflector(xk, tsq, k + 1 < r, PIVOT, a2)) unclear = true;          // decisions (3), (4), (5)
            const bool degen = !(tsq > DBL_MIN);
            double nb_, s, ng;
            if (degen) { nb_ = -xk; s = 0.0; ng = 0.0; }
            else {
                const double nrm = fast_sqrt(fma(xk, xk, tsq));
                nb_ = xk >= 0.0 ? nrm : -nrm;       // Eigen: if (c0 >= 0) beta = -beta  (-0.0 counts as >= 0)
                s = nb_ + xk;
                ng = -fast_recip(nb_ * s);
            }
            const bool active = isQ ? j < r : live;               // (the pivot column is still live here)
            const double ngam = active ? fma(s, ak, d) * ng : 0.0;
            double an = fma(s, ngam, ak);
            if (!isQ && lane == P && !degen) an = -nb_;           // R(k,k) = beta
            if (!isQ && wave == 0) {
                if (active) rrow[k * MAXD + j] = an;              // row k of R
                if (lane == P && hcoeffs) hcoeffs[cbase + k] = -(s * s) * ng;     // tau = w / beta
            }
            if (!isQ && lane == P) { live = false; nu2 = -1.0; }
            if (PIVOT && live) {
                // LAWN-176 downdate in the squared form; a column that passes Eigen's recompute test sends the tile to the exact path
                const double nn = fma(-an, an, nu2);
                nu2 = nn;
                if (nn <= thr) unclear = true;                    // decision (2) and the recomputation this kernel does not do
            }
            // ---- rank-1 update (rows above k are finished; row k takes its new value)
            if (where == 0) {
#pragma unroll
                for (int i = 0; i < RW; ++i) a[i] = fma(ngam, xm[i], a[i]);
            } else if (where == 1) {
#pragma unroll
                for (int i = 0; i < RW; ++i) { const double up = fma(ngam, xm[i], a[i]); a[i] = rw + i == k ? an : up; }
            }
#ifdef QRK_MID_PROF
            if (a[0] == 1.2345e300) pt[5]++;
#endif
            MID_TICK(4);
        }
#ifdef QRK_MID_PROF
        if (blockIdx.x == 0 && lane == 0)
            printf("mid prof wave %d (100 MHz ticks over %d steps): pivot+publish %llu  barrier x %llu  dots %llu  barrier d %llu  scalars+update %llu\n",
                   wave, c, pt[0], pt[1], pt[2], pt[3], pt[4]);
#endif
        if (unclear) s_unclear = 1;
        __syncthreads();
        if (tid == 0 && s_unclear != 0 && redo_count) redo_ids[atomicAdd(redo_count, 1)] = gidx;
        // ---- outputs: permutation splice (:519-521), R packed by columns, Q row-major
        for (int p = tid; p < c; p += T) perm[cbase + p] = cbase + col_of_pos[p];
        {
            double* rv = r_vals + roff;
            const int nr = c * (c + 1) / 2;
            for (int e = tid; e < nr; e += T) {
                int p = (int)((sqrt(8.0 * (double)e + 1.0) - 1.0) * 0.5);
                while (p * (p + 1) / 2 > e) --p;
                while ((p + 1) * (p + 2) / 2 <= e) ++p;
                const int i = e - p * (p + 1) / 2;
                rv[e] = rrow[i * MAXD + col_of_pos[p]];
            }
        }
        if (isQ && j < r) {
            double* qd = q_vals + qoff + (int64_t)j * r + rw;
#pragma unroll
            for (int i = 0; i < RW; ++i) if (rw + i < r) qd[i] = a[i];
        }
        __syncthreads();
        if (tid == 0) next_tile = (int)gridDim.x + atomicAdd(queue, 1);
        __syncthreads();
        t = next_tile;
    }
}

}  // namespace mid

bool bdqr_mid_supported(int max_rows, int max_cols)
{
    if (const char* e = std::getenv("QRK_MID")) { if (e[0] == '0') return false; }
    return max_rows <= mid::MAXD && max_cols <= max_rows;
}

hipError_t launch_bdqr_mid(const WaveBatch& nb, const double* tiles, double* q_vals, double* r_vals, int32_t* perm, double* hcoeffs,
                           int32_t* redo_count, int32_t* redo_ids, int32_t* queue, hipStream_t stream)
{
    if (nb.num_tiles <= 0) return hipSuccess;
    static int num_cus = 0;
    if (num_cus == 0) {
        int dev = 0;
        if (hipError_t e = hipGetDevice(&dev)) return e;
        if (hipError_t e = hipDeviceGetAttribute(&num_cus, hipDeviceAttributeMultiprocessorCount, dev)) return e;
    }
    if (hipError_t e = hipMemsetAsync(queue, 0, sizeof(int32_t), stream)) return e;
    const int64_t slots = (int64_t)num_cus * 2;     // 8 (RW = 32: 2 per SIMD, no spills) or 16 waves per CU
    const unsigned grid = (unsigned)(nb.num_tiles < slots ? nb.num_tiles : slots);
    if (nb.pivoting)
        hipLaunchKernelGGL(mid::bdqr_mid_kernel<true>, dim3(grid), dim3(mid::T), 0, stream, nb, tiles, q_vals, r_vals, perm, hcoeffs,
                           redo_count, redo_ids, queue);
    else
        hipLaunchKernelGGL(mid::bdqr_mid_kernel<false>, dim3(grid), dim3(mid::T), 0, stream, nb, tiles, q_vals, r_vals, perm, hcoeffs,
                           redo_count, redo_ids, queue);
    return hipGetLastError();
}

}  // namespace qrk
