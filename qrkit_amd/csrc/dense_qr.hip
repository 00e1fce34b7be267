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
