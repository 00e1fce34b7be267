// bdqr_exact_tile.h -- one tile of the exact-arithmetic path (see bdqr_exact.hip for what it is and why): Eigen's
// ColPivHouseholderQR::computeInPlace / householder_qr_inplace_unblocked, makeHouseholder, applyHouseholderOnTheLeft and
// HouseholderSequence::evalTo, one IEEE-754 double operation at a time in Eigen's scalar order.  Shared by the exact kernel
// (bdqr_exact.hip, a workgroup of 256 threads per listed tile) and by the uniform 32 x 32 kernel (bdqr_pair.hip), whose
// wavefronts redo their own flagged tiles with it (64 threads) at the end of the launch.  Every function body switches FMA
// contraction off itself, so the including file may be compiled with contraction on.
// Call site in the reference: blockSolver.compute(block) / matrixQ() / matrixR() / colsPermutation(),
// src/QRKit/BlockDiagonalSparseQR.h:437-447,519-521.
#ifndef QRK_BDQR_EXACT_TILE_H
#define QRK_BDQR_EXACT_TILE_H
#include "qrk_device.h"

#include <float.h>

namespace qrk {
namespace exact {

constexpr int T = 256;          // threads per tile of the exact kernel
constexpr double SQRT_EPS = 1.4901161193847656e-08;   // sqrt(DBL_EPSILON): Eigen's norm_downdate_threshold

// Inverse of e = p(p+1)/2 + i (0 <= i <= p): position in the packed upper triangle by columns.
__device__ __forceinline__ void tri_unpack(int64_t e, int& p, int& i)
{
#pragma clang fp contract(off)
    int64_t q = (int64_t)((sqrt(8.0 * (double)e + 1.0) - 1.0) * 0.5);
    while ((q + 1) * (q + 2) / 2 <= e) ++q;
    while (q * (q + 1) / 2 > e) --q;
    p = (int)q;
    i = (int)(e - q * (q + 1) / 2);
}

// a is a better pivot than b: larger norm, or the same norm at a smaller current position (Eigen's "first maximum").
__device__ __forceinline__ bool better(double va, int pa, double vb, int pb) { return va > vb || (va == vb && pa < pb); }

struct Shared {
    double* xbuf;     // [maxr] pivot column, then the essential part of the reflector
    double* nu;       // [maxc] m_colNormsUpdated
    double* nd;       // [maxc] m_colNormsDirect
    double* hc;       // [maxc] m_hCoeffs
    int* pos;         // [maxc] current position of original column j
    int* col_at;      // [maxc] original column at position p
    double* sval;     // [T] reduction scratch
    int* spos;        // [T]
};

// One tile.  tile: r x c column-major input; W: r x c row-major working copy (LDS or global); q: r x r row-major Q
// (LDS or the output itself).
template <bool PIVOT, int T>
__device__ void tile_qr(int r, int c, const double* __restrict__ tile, double* W, double* q, const Shared& sh)
{
#pragma clang fp contract(off)
    const int t = threadIdx.x;
    for (int e = t; e < r * c; e += T) {
        const int j = e / r, i = e - j * r;
        W[(size_t)i * c + j] = tile[e];
    }
    for (int j = t; j < c; j += T) { sh.pos[j] = j; sh.col_at[j] = j; }
    __syncthreads();
    if (PIVOT) {
        for (int j = t; j < c; j += T) {
            double s = 0.0;
            for (int i = 0; i < r; ++i) { const double v = W[(size_t)i * c + j]; s += v * v; }
            const double n = sqrt(s);
            sh.nu[j] = n; sh.nd[j] = n;
        }
        __syncthreads();
    }
    for (int k = 0; k < c; ++k) {      // size = min(rows, cols) = cols (portrait tiles only)
        int jb = k;
        if (PIVOT) {
            // biggest remaining column norm, first maximum over the CURRENT positions k..c-1
            double bv = -1.0; int bp = 0x7fffffff;
            for (int j = t; j < c; j += T) {
                const int p = sh.pos[j];
                if (p >= k && better(sh.nu[j], p, bv, bp)) { bv = sh.nu[j]; bp = p; }
            }
            sh.sval[t] = bv; sh.spos[t] = bp;
            __syncthreads();
            for (int s = T / 2; s > 0; s >>= 1) {
                if (t < s && better(sh.sval[t + s], sh.spos[t + s], sh.sval[t], sh.spos[t])) {
                    sh.sval[t] = sh.sval[t + s]; sh.spos[t] = sh.spos[t + s];
                }
                __syncthreads();
            }
            if (t == 0) {
                // m_qr.col(k).swap(m_qr.col(biggest)) and the two norm tables: position bookkeeping only
                const int b = sh.spos[0];
                const int cb = sh.col_at[b], ck = sh.col_at[k];
                sh.col_at[k] = cb; sh.col_at[b] = ck; sh.pos[cb] = k; sh.pos[ck] = b;
            }
            __syncthreads();
            jb = sh.col_at[k];
        }
        for (int i = k + t; i < r; i += T) sh.xbuf[i] = W[(size_t)i * c + jb];
        __syncthreads();
        // makeHouseholder (every thread evaluates the same scalars in the same order)
        const double c0 = sh.xbuf[k];
        double tail = 0.0;
        for (int i = k + 1; i < r; ++i) { const double v = sh.xbuf[i]; tail += v * v; }
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
        for (int i = k + 1 + t; i < r; i += T) {
            const double e = degen ? 0.0 : sh.xbuf[i] / denom;
            sh.xbuf[i] = e;
            W[(size_t)i * c + jb] = e;        // packed QR: essential part below the diagonal
        }
        if (t == 0) { W[(size_t)k * c + jb] = beta; sh.hc[k] = tau; }
        __syncthreads();
        // applyHouseholderOnTheLeft on the remaining columns + norm downdate
        const int m = r - k;
        for (int j = t; j < c; j += T) {
            if (!(PIVOT ? sh.pos[j] > k : j > k)) continue;
            double* colk = W + (size_t)k * c + j;
            if (m == 1) *colk *= (1.0 - tau);
            else if (tau != 0.0) {
                double tmp = 0.0;
                for (int i = k + 1; i < r; ++i) tmp += sh.xbuf[i] * W[(size_t)i * c + j];
                tmp += *colk;
                *colk -= tau * tmp;
                for (int i = k + 1; i < r; ++i) W[(size_t)i * c + j] -= (tau * sh.xbuf[i]) * tmp;
            }
            if (PIVOT) {
                const double nuj = sh.nu[j];
                if (nuj != 0.0) {
                    double temp = fabs(*colk) / nuj;
                    temp = (1.0 + temp) * (1.0 - temp);
                    temp = temp < 0.0 ? 0.0 : temp;
                    const double ratio = nuj / sh.nd[j];
                    const double temp2 = temp * (ratio * ratio);
                    if (temp2 <= SQRT_EPS) {
                        double s = 0.0;
                        for (int i = k + 1; i < r; ++i) { const double v = W[(size_t)i * c + j]; s += v * v; }
                        const double n = sqrt(s);
                        sh.nd[j] = n; sh.nu[j] = n;
                    } else sh.nu[j] = nuj * sqrt(temp);
                }
            }
        }
        __syncthreads();
    }
    // HouseholderSequence::evalTo: Q = I, then H_k on the corner Q(k:, k:) for k = c-1 .. 0
    for (int e = t; e < r * r; e += T) q[e] = (e / r == e % r) ? 1.0 : 0.0;
    __syncthreads();
    for (int k = c - 1; k >= 0; --k) {
        const int jb = sh.col_at[k];
        const double tau = sh.hc[k];
        for (int i = k + 1 + t; i < r; i += T) sh.xbuf[i] = W[(size_t)i * c + jb];
        __syncthreads();
        const int m = r - k;
        for (int j = k + t; j < r; j += T) {
            double* qk = q + (size_t)k * r + j;
            if (m == 1) *qk *= (1.0 - tau);
            else if (tau != 0.0) {
                double tmp = 0.0;
                for (int i = k + 1; i < r; ++i) tmp += sh.xbuf[i] * q[(size_t)i * r + j];
                tmp += *qk;
                *qk -= tau * tmp;
                for (int i = k + 1; i < r; ++i) q[(size_t)i * r + j] -= (tau * sh.xbuf[i]) * tmp;
            }
        }
        __syncthreads();
    }
}


// Outputs of one tile after tile_qr: permutation (m_outputPerm_c splice, BlockDiagonalSparseQR.h:519-521), tau, the packed
// upper triangle of R in CSC order (:475-479); Q rows (:455-471 / :480-492) when Q was built in LDS (q_lds != nullptr).
template <int T>
__device__ __forceinline__ void tile_store(int r, int c, int cbase, const double* W, const double* q_lds, const Shared& sh,
                                           int32_t* __restrict__ perm, double* __restrict__ hcoeffs, double* __restrict__ r_out,
                                           double* __restrict__ q_out)
{
    const int tid = threadIdx.x;
    for (int p = tid; p < c; p += T) {
        perm[cbase + p] = cbase + sh.col_at[p];
        if (hcoeffs) hcoeffs[cbase + p] = sh.hc[p];
    }
    const int64_t n_r = (int64_t)c * (c + 1) / 2;
    for (int64_t e = tid; e < n_r; e += T) {
        int p, i;
        tri_unpack(e, p, i);
        r_out[e] = W[(size_t)i * c + sh.col_at[p]];
    }
    if (q_lds)
        for (int e = tid; e < r * r; e += T) q_out[e] = q_lds[e];
}

// LDS carve-up of the fixed part for T threads: [xbuf maxr][nu maxc][nd maxc][hc maxc][sval T] doubles, [pos maxc][col_at maxc][spos T] ints.
template <int T>
__device__ __forceinline__ double* carve_shared(unsigned char* smem, int maxr, int maxc, Shared& sh)
{
    double* d = reinterpret_cast<double*>(smem);
    sh.xbuf = d; d += maxr;
    sh.nu = d; d += maxc;
    sh.nd = d; d += maxc;
    sh.hc = d; d += maxc;
    sh.sval = d; d += T;
    int* ip = reinterpret_cast<int*>(d);
    sh.pos = ip; ip += maxc;
    sh.col_at = ip; ip += maxc;
    sh.spos = ip; ip += T;
    return reinterpret_cast<double*>((reinterpret_cast<uintptr_t>(ip) + 15) & ~(uintptr_t)15);
}

}  // namespace exact
}  // namespace qrk
#endif
