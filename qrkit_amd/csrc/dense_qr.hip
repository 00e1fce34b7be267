// dense_qr.hip -- single dense Householder QR with IMPLICIT Q (packed reflectors), for gfx950.
//
// This is the right-block solver of QRKit::BlockAngularSparseQR: the tests instantiate it with
// Eigen::ColPivHouseholderQR<MatrixXd> (test/test-qrkit.cpp:46-48), called at
// src/QRKit/BlockAngularSparseQR.h:361-369 (rightSolver.compute(J2.bottomRows(n1+n2-m1))), :488
// (matrixR()), :498-503 (colsPermutation()) and :619-622 / :636-638 (matrixQ() products).
// Like Eigen it keeps Q as the sequence of reflectors: essentials below the diagonal of the packed
// matrix plus hCoeffs; Q is never formed.
//
// One workgroup works in place on the caller's column-major matrix (level-2 algorithm, the same
// step structure as bdqr_wg.hip).  A multi-workgroup panel-blocked version is the planned replacement
// for right blocks of BASELINE configs[3] size (40000 x 2000).
#include "qrk_device.h"

#include <cstdlib>

#include <float.h>

namespace qrk {

constexpr int DQ_THREADS = 1024;
constexpr int DQ_WAVES = DQ_THREADS / 64;

__device__ __forceinline__ double dq_wave_sum(double v)
{
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off);
    return v;
}

// the same sum by DPP steps inside the rows of 16 lanes and four v_readlane pairs: no trip through the LDS pipe (__shfl_xor is
// ds_bpermute: six dependent LDS operations per sum; dense_apply_q_kernel does one per reflector, 2 000 in a row)
__device__ __forceinline__ double dq_wave_sum_dpp(double v)
{
    v += dpp_f64<0xB1>(v);    // quad_perm [1,0,3,2]
    v += dpp_f64<0x4E>(v);    // quad_perm [2,3,0,1]
    v += dpp_f64<0x141>(v);   // row_half_mirror
    v += dpp_f64<0x140>(v);   // row_mirror
    return (readlane_f64(v, 0) + readlane_f64(v, 16)) + (readlane_f64(v, 32) + readlane_f64(v, 48));
}

__device__ __forceinline__ double dq_block_sum(double v, double* red)
{
    v = dq_wave_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    double s = 0.0;
#pragma unroll
    for (int w = 0; w < DQ_WAVES; ++w) s += red[w];
    return s;
}

// ColPivHouseholderQR::computeInPlace / HouseholderQR::compute on A (r x c, ld = lda), in place.
__global__ void __launch_bounds__(DQ_THREADS)
dense_qr_kernel(double* __restrict__ A, int64_t lda, int r, int c, int pivoting, double* __restrict__ hcoeffs,
                int32_t* __restrict__ perm, int* __restrict__ unclear)
{
    using namespace decide;   // decisions inside their error margin send the matrix to the exact path (qrk_device.h)
    double a2 = 0.0;          // |A|^2: squared norm of the first pivot column
    extern __shared__ __attribute__((aligned(16))) double smem[];
    double* xv = smem;                 // [r] pivot column, then essential vector
    double* nu2 = xv + r;              // [c]
    double* thr = nu2 + c;             // [c]
    double* red = thr + c;             // [2*DQ_WAVES]
    int* pidx = reinterpret_cast<int*>(red + 2 * DQ_WAVES);   // [c]
    int* ired = pidx + c;              // [2*DQ_WAVES]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int size = r < c ? r : c;

    for (int jc = wave; jc < c; jc += DQ_WAVES) {
        double s = 0.0;
        for (int i = lane; i < r; i += 64) { const double v = A[(int64_t)jc * lda + i]; s = fma(v, v, s); }
        s = dq_wave_sum(s);
        if (lane == 0) { nu2[jc] = s; thr[jc] = s * THR_HI; pidx[jc] = jc; }
    }
    __syncthreads();

    for (int k = 0; k < size; ++k) {
        if (pivoting) {
            double best = -1.0;
            int bi = c;
            for (int jc = k + tid; jc < c; jc += DQ_THREADS) {
                const double v = nu2[jc];
                if (v > best) { best = v; bi = jc; }
            }
#pragma unroll
            for (int off = 32; off >= 1; off >>= 1) {
                const double ob = __shfl_xor(best, off);
                const int oi = __shfl_xor(bi, off);
                if (ob > best || (ob == best && oi < bi)) { best = ob; bi = oi; }
            }
            if (lane == 0) { red[wave] = best; ired[wave] = bi; }
            __syncthreads();
            best = red[0]; bi = ired[0];
#pragma unroll
            for (int w = 1; w < DQ_WAVES; ++w)
                if (red[w] > best || (red[w] == best && ired[w] < bi)) { best = red[w]; bi = ired[w]; }
            const int b = bi < c ? bi : k;
            if (k == 0) a2 = best;
            for (int jc = k + tid; jc < c; jc += DQ_THREADS)
                if (jc != b && near_best(nu2[jc], thr[jc], best, a2)) *unclear = 1;          // decision (1)
            __syncthreads();
            if (b != k) {
                for (int i = tid; i < r; i += DQ_THREADS) {
                    const double tk = A[(int64_t)k * lda + i];
                    A[(int64_t)k * lda + i] = A[(int64_t)b * lda + i];
                    A[(int64_t)b * lda + i] = tk;
                }
                if (tid == 0) {
                    double tn = nu2[k]; nu2[k] = nu2[b]; nu2[b] = tn;
                    tn = thr[k]; thr[k] = thr[b]; thr[b] = tn;
                    const int tp = pidx[k]; pidx[k] = pidx[b]; pidx[b] = tp;
                }
            }
            __syncthreads();
        }

        double part = 0.0;
        for (int i = k + tid; i < r; i += DQ_THREADS) {
            const double v = A[(int64_t)k * lda + i];
            xv[i] = v;
            if (i > k) part = fma(v, v, part);
        }
        const double tailSq = dq_block_sum(part, red);
        const double xk = xv[k];
        double beta, tau, scale;
        if (k == 0 && !pivoting) a2 = fma(xk, xk, tailSq);
        if (tid == 0 && unclear_reflector(xk, tailSq, k + 1 < r, pivoting != 0, a2, (pivoting & PIVOTING_SIGN_FREE) != 0)) *unclear = 1;   // (3), (4), (5)
        if (tailSq <= DBL_MIN) {           // makeHouseholder: tau = 0, beta = x0, essential = 0
            beta = xk; tau = 0.0; scale = 0.0;
        } else {
            const double nrm = sqrt(fma(xk, xk, tailSq));
            beta = xk >= 0.0 ? -nrm : nrm;
            scale = 1.0 / (xk - beta);
            tau = (beta - xk) / beta;
        }
        __syncthreads();
        // essential vector in place and in LDS
        for (int i = k + 1 + tid; i < r; i += DQ_THREADS) {
            const double e = xv[i] * scale;
            xv[i] = e;
            A[(int64_t)k * lda + i] = e;
        }
        if (tid == 0) { A[(int64_t)k * lda + k] = beta; hcoeffs[k] = tau; }
        __syncthreads();

        // applyHouseholderOnTheLeft on the trailing columns: tmp = ess^T bottom + row0; row0 -= tau tmp;
        // bottom -= tau ess tmp; then the LAWN-176 norm downdate in squared form
        for (int jc = k + 1 + wave; jc < c; jc += DQ_WAVES) {
            double* col = A + (int64_t)jc * lda;
            double d = 0.0;
            for (int i = k + 1 + lane; i < r; i += 64) d = fma(xv[i], col[i], d);
            d = dq_wave_sum(d);
            const double ck = col[k];
            const double tt = tau * (d + ck);
            const double cknew = ck - tt;
            double s2 = 0.0;
            for (int i = k + 1 + lane; i < r; i += 64) {
                const double v = fma(-tt, xv[i], col[i]);
                col[i] = v;
                s2 = fma(v, v, s2);
            }
            if (lane == 0) col[k] = cknew;
            if (pivoting) {
                double nn = fma(-cknew, cknew, nu2[jc]);
                nn = nn > 0.0 ? nn : 0.0;
                if (nn <= thr[jc]) {
                    s2 = dq_wave_sum(s2);
                    if (lane == 0) {
                        if (in_recompute_band(nn, thr[jc], a2)) *unclear = 1;                         // decision (2)
                        nu2[jc] = s2; thr[jc] = s2 * THR_HI;
                    }
                } else if (lane == 0) {
                    nu2[jc] = nn;
                }
            }
        }
        __syncthreads();
    }
    for (int jc = tid; jc < c; jc += DQ_THREADS) perm[jc] = pidx[jc];
}

constexpr int APQ_EPT = 8;     // entries of a reflector per thread in the register-prefetch form of dense_apply_q_kernel
// B <- Q^T B (transpose != 0) or Q B, Q = H_0 ... H_{n-1} from the packed reflectors
// (HouseholderSequence::applyThisOnTheLeft).  One workgroup per right-hand-side column.
__global__ void __launch_bounds__(256)
dense_apply_q_kernel(const double* __restrict__ QR, int64_t lda, int r, int nrefl, const double* __restrict__ hcoeffs,
                     int transpose, double* __restrict__ B, int64_t ldb, int64_t nrhs)
{
    extern __shared__ double bs[];   // [r] the column, [4] reduction
    double* red = bs + r;
    const int tid = threadIdx.x;
    for (int64_t col = blockIdx.x; col < nrhs; col += gridDim.x) {
        double* b = B + col * ldb;
        for (int i = tid; i < r; i += 256) bs[i] = b[i];
        __syncthreads();
        if (r <= 256 * APQ_EPT + 1) {
            // short columns (the n x n second-stage factor of the two-stage form, n <= 2049): the reflector of the NEXT step is
            // loaded into registers while this one is applied, and is read once for dot and update -- a step was the latency of
            // two dependent loads from L2 plus three barriers (2.5 us; 5.1 ms for the 2000 reflectors of a solve)
            double vn[APQ_EPT];
            auto load_v = [&](int k, double (&dst)[APQ_EPT]) {
                const double* v = QR + (int64_t)k * lda;
#pragma unroll
                for (int e = 0; e < APQ_EPT; ++e) { const int i = k + 1 + tid + 256 * e; dst[e] = v[i < r ? i : r - 1]; }
            };
            int kn = transpose ? 0 : nrefl - 1;
            double taun = nrefl > 0 ? hcoeffs[kn] : 0.0;
            if (nrefl > 0) load_v(kn, vn);
            for (int s = 0; s < nrefl; ++s) {
                const int k = kn;
                const double tau = taun;
                double vc[APQ_EPT];
#pragma unroll
                for (int e = 0; e < APQ_EPT; ++e) vc[e] = (k + 1 + tid + 256 * e < r) ? vn[e] : 0.0;
                if (s + 1 < nrefl) { kn = transpose ? s + 1 : nrefl - 2 - s; taun = hcoeffs[kn]; load_v(kn, vn); }
                const double xk = bs[k];
                double part = 0.0;
#pragma unroll
                for (int e = 0; e < APQ_EPT; ++e) { const int i = k + 1 + tid + 256 * e; part = fma(vc[e], bs[i < r ? i : r - 1], part); }
                part = dq_wave_sum_dpp(part);
                if ((tid & 63) == 0) red[tid >> 6] = part;
                __syncthreads();
                const double tt = tau * (red[0] + red[1] + red[2] + red[3] + xk);
                if (tid == 0) bs[k] = xk - tt;
#pragma unroll
                for (int e = 0; e < APQ_EPT; ++e) { const int i = k + 1 + tid + 256 * e; if (i < r) bs[i] = fma(-tt, vc[e], bs[i]); }
                __syncthreads();
            }
        } else
        for (int s = 0; s < nrefl; ++s) {
            const int k = transpose ? s : nrefl - 1 - s;
            const double tau = hcoeffs[k];
            const double* v = QR + (int64_t)k * lda;
            double part = 0.0;
            for (int i = k + 1 + tid; i < r; i += 256) part = fma(v[i], bs[i], part);
            part = dq_wave_sum(part);
            if ((tid & 63) == 0) red[tid >> 6] = part;
            __syncthreads();
            const double tmp = red[0] + red[1] + red[2] + red[3] + bs[k];
            const double tt = tau * tmp;
            __syncthreads();
            if (tid == 0) bs[k] -= tt;
            for (int i = k + 1 + tid; i < r; i += 256) bs[i] = fma(-tt, v[i], bs[i]);
            __syncthreads();
        }
        for (int i = tid; i < r; i += 256) b[i] = bs[i];
        __syncthreads();
    }
}

// ---- blocked application of a Householder sequence (round 5) ---------------------------------------------------------------
// dense_apply_q_kernel applies the reflectors one after the other: ~1 us each (a dot product over a workgroup, three barriers), 2.0 ms
// for the 2 000 reflectors of the second stage of BASELINE configs[3] per solve().  With the triangular factors T_j of blocks of 32
// reflectors (Q = prod_j (I - V_j T_j V_j^T), Eigen's make_block_householder_triangular_factor / LAPACK larft, forward columnwise)
// a block is three matrix-vector products.  T is computed once per factorisation (dense_q_tfactors_kernel, a workgroup per block).
constexpr int QB = 32;                 // reflectors per block

// T_j of every block: G = V^T V (strict upper part) through LDS tiles of 64 rows, then the column recurrence
// T(l, l) = tau_l, T(0:l, l) = -tau_l T(0:l, 0:l) G(0:l, l).  V = unit-lower view of QR(:, k0 .. k0 + 31) from row k0.
__global__ void __launch_bounds__(256)
dense_q_tfactors_kernel(const double* __restrict__ QR, int64_t lda, int n, int nrefl, const double* __restrict__ tau, double* __restrict__ T)
{
    __shared__ double vt[64][QB + 1];
    __shared__ double G[QB][QB + 1];
    __shared__ double Ts[QB][QB + 1];
    const int tid = threadIdx.x, blk = blockIdx.x;
    const int k0 = blk * QB, kb = (nrefl - k0) < QB ? (nrefl - k0) : QB;
    // thread -> up to four (a, b) pairs of the 32 x 32 square (a = tid >> 3, b = 4 (tid & 7) .. + 3); only a < b < kb is kept
    const int a = tid >> 3, b0 = 4 * (tid & 7);
    double acc[4] = {0.0, 0.0, 0.0, 0.0};
    for (int r0 = k0; r0 < n; r0 += 64) {
        __syncthreads();
        // tile rows r0 .. r0 + 63: lane = row (coalesced along a column), eight columns per pass
        for (int e = tid; e < 64 * QB; e += 256) {
            const int l = e >> 6, i = r0 + (e & 63);
            double v = 0.0;
            if (l < kb && i < n) v = i > k0 + l ? QR[(int64_t)(k0 + l) * lda + i] : (i == k0 + l ? 1.0 : 0.0);
            vt[e & 63][l] = v;
        }
        __syncthreads();
#pragma unroll 8
        for (int i = 0; i < 64; ++i) {
            const double va = vt[i][a];
#pragma unroll
            for (int q = 0; q < 4; ++q) acc[q] = fma(va, vt[i][b0 + q], acc[q]);
        }
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) G[a][b0 + q] = acc[q];
    for (int e = tid; e < QB * QB; e += 256) Ts[e >> 5][e & 31] = 0.0;
    __syncthreads();
    if (tid < QB) {
        // row `tid` of T depends on row `tid` only: T(a, l) = -tau_l sum_{b = a .. l-1} T(a, b) G(b, l)
        const int ar = tid;
        for (int l = 0; l < kb; ++l) {
            const double tl = tau[k0 + l];
            double tv = 0.0;
            if (ar == l) tv = tl;
            else if (ar < l) {
                double sacc = 0.0;
                for (int bb = ar; bb < l; ++bb) sacc = fma(Ts[ar][bb], G[bb][l], sacc);
                tv = -tl * sacc;
            }
            Ts[ar][l] = tv;
        }
    }
    __syncthreads();
    for (int e = tid; e < QB * QB; e += 256) T[(int64_t)blk * QB * QB + e] = Ts[e >> 5][e & 31];
}

// Q^T B with the blocks' T factors on MANY workgroups (round 5): ONE launch per block of 32 reflectors (+ one ahead of the first),
// blockIdx.x = a slab of the rows below the block's first, blockIdx.y = right-hand side.  The one-workgroup kernel (reflector by reflector) pulls the 16 MB of V of a
// 2 000-reflector sequence through ONE CU -- 13 GB/s, 1.8 ms of configs[3]'s solve().  Here a launch sums the slabs' shares of the block's
// w = V^T x (32 numbers per slab, left by the launch before), applies T^T, updates its slab in place and, rows in hand, leaves the slab's
// share of the NEXT block's w: every workgroup reads its slab of two 32-column panels and nothing else.  (A first form with ONE launch per block, every workgroup forming all of w itself and x
// ping-ponging between two work vectors, pulled the whole panel through every CU: 20 us per block, profiles/r05_solve.txt.)
__device__ __forceinline__ void dq_panel_row(const double* __restrict__ V, int64_t lda, int i, int k0, int kb, double (&vv)[QB])
{
#pragma unroll
    for (int l = 0; l < QB; ++l) vv[l] = V[(int64_t)(l < kb ? l : 0) * lda + i];      // (all 32 loads of the row in flight)
    if (i < k0 + QB) {
#pragma unroll
        for (int l = 0; l < QB; ++l) vv[l] = i > k0 + l ? vv[l] : (i == k0 + l ? 1.0 : 0.0);
    }
#pragma unroll
    for (int l = 0; l < QB; ++l) vv[l] = l < kb ? vv[l] : 0.0;
}

constexpr int QT_MAXW = 32;            // most slabs (workgroups) of a launch of the blocked Q^T
__global__ void __launch_bounds__(256)
dense_qt_dot_kernel(const double* __restrict__ QR, int64_t lda, int n, int k0, int kb, const double* __restrict__ B, int64_t ldb,
                    double* __restrict__ partial)
{
    __shared__ double part[4 * QB];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const double* x = B + (int64_t)blockIdx.y * ldb;
    const double* V = QR + (int64_t)k0 * lda;
    const int per = (n - k0 + (int)gridDim.x - 1) / (int)gridDim.x;
    const int r0 = k0 + (int)blockIdx.x * per, r1 = (r0 + per) < n ? (r0 + per) : n;
    double wl[QB];
#pragma unroll
    for (int l = 0; l < QB; ++l) wl[l] = 0.0;
    for (int i = r0 + tid; i < r1; i += 256) {
        const double xi = x[i];
        double vv[QB];
        dq_panel_row(V, lda, i, k0, kb, vv);
#pragma unroll
        for (int l = 0; l < QB; ++l) wl[l] = fma(vv[l], xi, wl[l]);
    }
#pragma unroll
    for (int l = 0; l < QB; ++l) {
        const double sgl = dq_wave_sum_dpp(wl[l]);      // (DPP, not __shfl_xor: 32 sums of six ds_bpermute pairs each were 6 us of LDS pipe per launch)
        if (lane == 0) part[wave * QB + l] = sgl;
    }
    __syncthreads();
    if (tid < QB)
        partial[((int64_t)blockIdx.y * QT_MAXW + blockIdx.x) * QB + tid] = (part[tid] + part[QB + tid]) + (part[2 * QB + tid] + part[3 * QB + tid]);
}

// the slabs' shares of block g (nparts of them, left by the launch before) summed, T^T, the slab updated in place -- and, rows in hand, the
// slab's share of the NEXT block's w = V_next^T x (knext > 0), so that a block costs one launch
__global__ void __launch_bounds__(256)
dense_qt_update_kernel(const double* __restrict__ QR, int64_t lda, int n, int k0, int kb, const double* __restrict__ Tb,
                       const double* __restrict__ partial, int nparts, double* __restrict__ B, int64_t ldb, int knext,
                       double* __restrict__ partial_next)
{
    __shared__ double w[QB];
    __shared__ double w2[QB];
    __shared__ double part[4 * QB];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    double* x = B + (int64_t)blockIdx.y * ldb;
    const double* V = QR + (int64_t)k0 * lda;
    const int per = (n - k0 + (int)gridDim.x - 1) / (int)gridDim.x;      // (<= 256 by the launcher: a row per thread)
    const int r0 = k0 + (int)blockIdx.x * per, r1 = (r0 + per) < n ? (r0 + per) : n;
    const int kn = k0 + QB;                              // first row / reflector of the next block
    // (the last block has no next panel: column kn lies past the reflectors -- and, for the square second stage, past the buffer; its
    //  32 loads then read this block's own panel and are discarded)
    const double* Vn = knext > 0 ? QR + (int64_t)kn * lda : V;
    // every load of the launch is issued before the first barrier: the row of the two panels (64 strided loads), x, T's column and the
    // slabs' shares -- one trip to memory instead of three in a row
    const int i = r0 + tid;
    const bool rowact = i < r1, nextact = knext > 0 && rowact && i >= kn;
    double vv[QB], vn[QB];
    dq_panel_row(V, lda, rowact ? i : r0, k0, kb, vv);
    dq_panel_row(Vn, lda, nextact ? i : (knext > 0 ? kn : k0), kn, knext > 0 ? knext : 0, vn);
    const double xi = rowact ? x[i] : 0.0;
    double tv[QB];
    if (tid < QB) {
#pragma unroll
        for (int q = 0; q < QB; ++q) tv[q] = Tb[q * QB + tid];          // column tid of T (upper triangular, row-major)
        double pv[QT_MAXW];
#pragma unroll
        for (int q = 0; q < QT_MAXW; ++q) pv[q] = q < nparts ? partial[((int64_t)blockIdx.y * QT_MAXW + q) * QB + tid] : 0.0;
        double acc = 0.0;
#pragma unroll
        for (int q = 0; q < QT_MAXW; ++q) acc += pv[q];                       // (a fixed order)
        w[tid] = acc;
    }
    __syncthreads();
    if (tid < QB) {
        double acc = 0.0;
#pragma unroll
        for (int q = 0; q < QB; ++q) acc = fma(q <= tid ? tv[q] : 0.0, w[q], acc);      // T^T w
        w2[tid] = tid < kb ? acc : 0.0;
    }
    __syncthreads();
    double acc = xi;
#pragma unroll
    for (int l = 0; l < QB; ++l) acc = fma(-vv[l], w2[l], acc);
    if (rowact) x[i] = acc;
    if (knext > 0) {
        const double xa = nextact ? acc : 0.0;
#pragma unroll
        for (int l = 0; l < QB; ++l) {
            const double sgl = dq_wave_sum_dpp(vn[l] * xa);
            if (lane == 0) part[wave * QB + l] = sgl;
        }
        __syncthreads();
        if (tid < QB)
            partial_next[((int64_t)blockIdx.y * QT_MAXW + blockIdx.x) * QB + tid] = (part[tid] + part[QB + tid]) + (part[2 * QB + tid] + part[3 * QB + tid]);
    }
}

// B(0:n, :) <- Q^T B for the sequence of nrefl reflectors packed in QR with the T factors of launch_dense_q_tfactors; work: 2 * QT_MAXW * 32 nrhs doubles
hipError_t launch_dense_apply_qt_blocks(const double* QR, int64_t lda, int n, int nrefl, const double* T, double* B, int64_t ldb, int64_t nrhs,
                                        double* work, hipStream_t stream)
{
    if (nrhs <= 0 || nrefl <= 0) return hipSuccess;
    if (nrhs > 65535) return hipErrorInvalidValue;
    const int nblk = (nrefl + QB - 1) / QB;
    double* pbuf[2] = {work, work + (int64_t)QT_MAXW * QB * nrhs};
    // slabs of 64 .. 256 rows (a row per thread), at most QT_MAXW per right-hand side (n <= 256 QT_MAXW)
    auto slabs = [&](int k0) { int W = (n - k0 + 63) / 64; if (W > QT_MAXW) W = QT_MAXW; if (W < 1) W = 1; return W; };
    int Wprev = slabs(0);
    hipLaunchKernelGGL(dense_qt_dot_kernel, dim3((unsigned)Wprev, (unsigned)nrhs), dim3(256), 0, stream, QR, lda, n, 0, nrefl < QB ? nrefl : QB, B, ldb,
                       pbuf[0]);
    for (int g = 0; g < nblk; ++g) {
        const int k0 = g * QB, kb = (nrefl - k0) < QB ? (nrefl - k0) : QB;
        const int kn = k0 + QB, knext = g + 1 < nblk ? ((nrefl - kn) < QB ? (nrefl - kn) : QB) : 0;
        const int W = slabs(k0);
        hipLaunchKernelGGL(dense_qt_update_kernel, dim3((unsigned)W, (unsigned)nrhs), dim3(256), 0, stream, QR, lda, n, k0, kb,
                           T + (int64_t)g * QB * QB, pbuf[g & 1], Wprev, B, ldb, knext, pbuf[(g + 1) & 1]);
        Wprev = W;
    }
    return hipGetLastError();
}

size_t dense_q_tfactors_doubles(int nrefl) { return (size_t)((nrefl + QB - 1) / QB) * QB * QB; }

hipError_t launch_dense_q_tfactors(const double* QR, int64_t lda, int n, int nrefl, const double* tau, double* T, hipStream_t stream)
{
    if (nrefl <= 0) return hipSuccess;
    hipLaunchKernelGGL(dense_q_tfactors_kernel, dim3((unsigned)((nrefl + QB - 1) / QB)), dim3(256), 0, stream, QR, lda, n, nrefl, tau, T);
    return hipGetLastError();
}

size_t dense_qr_smem_bytes(int r, int c)
{
    return (size_t)(r + 2 * c + 2 * DQ_WAVES) * sizeof(double) + (size_t)(c + 2 * DQ_WAVES) * sizeof(int);
}

hipError_t launch_dense_qr(double* A, int64_t lda, int r, int c, int pivoting, double* hcoeffs, int32_t* perm, int* unclear,
                           hipStream_t stream)
{
    const size_t smem = dense_qr_smem_bytes(r, c);
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(dense_qr_kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(dense_qr_kernel, dim3(1), dim3(DQ_THREADS), smem, stream, A, lda, r, c, pivoting, hcoeffs, perm, unclear);
    return hipGetLastError();
}

hipError_t launch_dense_apply_q(const double* QR, int64_t lda, int r, int nrefl, const double* hcoeffs, int transpose,
                                double* B, int64_t ldb, int64_t nrhs, hipStream_t stream)
{
    if (nrhs <= 0) return hipSuccess;
    const size_t smem = (size_t)(r + 4) * sizeof(double);
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(dense_apply_q_kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    if (e != hipSuccess) return e;
    const unsigned grid = (unsigned)(nrhs < 4096 ? nrhs : 4096);
    hipLaunchKernelGGL(dense_apply_q_kernel, dim3(grid), dim3(256), smem, stream, QR, lda, r, nrefl, hcoeffs, transpose, B,
                       ldb, nrhs);
    return hipGetLastError();
}

}  // namespace qrk
