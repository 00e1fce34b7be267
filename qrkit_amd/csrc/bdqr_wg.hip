// bdqr_wg.hip -- one workgroup factorises one tile LARGER than 32x32 (rows >= cols, up to
// QRK_WG_MAX_DIM) of a block-diagonal matrix: A_i P_i = Q_i R_i with explicit Q_i, for gfx950.
//
// Same reference seam as bdqr_pair.hip (the hot loop of BlockDiagonalSparseQR::factorize,
// src/QRKit/BlockDiagonalSparseQR.h:432-526, with Eigen ColPivHouseholderQR / HouseholderQR behind
// blockSolver.compute and HouseholderSequence behind matrixQ()), for the tiles of mixed-size batches
// (BASELINE configs[4]: sizes 8..256) that do not fit one half-wave.
//
// The working copy W of the tile lives in a per-workgroup global workspace (L2 resident: a 256x256
// tile is 512 KB) and Q^T is accumulated directly in the output array: column-major Q^T IS the
// row-major Q_i of the CSR value order.  Per step k:
//   pivot     block-wide first-maximum of the squared column norms kept in LDS; Eigen's physical
//             column swap (the norm tables and the permutation swap with it);
//   reflector the pivot column goes to LDS; |tail|^2 by a block reduction; beta, w, 1/(beta w);
//   update    every wave takes columns of [W(:,k+1:) | Q^T] round robin, lanes stride the rows k..r-1:
//             d = x_tail^T c_tail (wave reduction), gamma = (d - w c_k)/(beta w),
//             c_k += w gamma, c_tail -= gamma x_tail, and for W columns the LAWN-176 norm downdate
//             (recompute from the freshly updated values when the test fires).
// This is the plain (level-2) algorithm; a panel-blocked MFMA variant is the planned replacement.
#include "qrk_device.h"

#include <float.h>

namespace qrk {

constexpr int WG_THREADS = 256;
constexpr int WG_WAVES = WG_THREADS / 64;

__device__ __forceinline__ double wave_sum(double v)
{
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off);
    return v;
}

// Block-wide sum; every thread gets the result.  red[] has WG_WAVES doubles.
__device__ __forceinline__ double block_sum(double v, double* red)
{
    v = wave_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    double s = 0.0;
#pragma unroll
    for (int w = 0; w < WG_WAVES; ++w) s += red[w];
    return s;
}

__global__ void __launch_bounds__(WG_THREADS)
bdqr_wg_kernel(WaveBatch nb, const double* __restrict__ tiles, double* __restrict__ q_vals,
               double* __restrict__ r_vals, int32_t* __restrict__ perm, double* __restrict__ hcoeffs,
               double* __restrict__ workspace, int64_t ws_stride, int max_dim,
               int32_t* __restrict__ redo_count, int32_t* __restrict__ redo_ids)
{
    using namespace decide;   // decisions inside their error margin send the tile to the exact path (qrk_device.h)
    extern __shared__ __attribute__((aligned(16))) double smem[];
    double* xv = smem;                       // [max_dim] pivot column (rows k..r-1 valid)
    double* nu2 = xv + max_dim;              // [max_dim] m_colNormsUpdated^2
    double* thr = nu2 + max_dim;             // [max_dim] sqrt(eps) * m_colNormsDirect^2
    double* red = thr + max_dim;             // [2*WG_WAVES] reduction scratch
    int* pidx = reinterpret_cast<int*>(red + 2 * WG_WAVES);   // [max_dim] permutation indices
    int* ired = pidx + max_dim;              // [2*WG_WAVES], then [1] redo flag
    int* unclear = ired + 2 * WG_WAVES;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    double* W = workspace + (int64_t)blockIdx.x * ws_stride;

    for (int64_t t = blockIdx.x; t < nb.num_tiles; t += gridDim.x) {
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
        const double* src = tiles + toff;
        double* QT = q_vals + qoff;          // column-major r x r Q^T == row-major Q_i

        // ---- working copy, identity, squared column norms
        for (int64_t e = tid; e < (int64_t)r * c; e += WG_THREADS) W[e] = src[e];
        for (int64_t e = tid; e < (int64_t)r * r; e += WG_THREADS) QT[e] = (e / r == e % r) ? 1.0 : 0.0;
        if (tid == 0) *unclear = 0;
        double a2 = 0.0;                     // |A|^2: squared norm of the first pivot column
        __syncthreads();
        for (int jc = wave; jc < c; jc += WG_WAVES) {
            double s = 0.0;
            for (int i = lane; i < r; i += 64) { const double v = W[(int64_t)jc * r + i]; s = fma(v, v, s); }
            s = wave_sum(s);
            if (lane == 0) { nu2[jc] = s; thr[jc] = s * THR_HI; pidx[jc] = jc; }
        }
        __syncthreads();

        for (int k = 0; k < c; ++k) {
            // ---- pivot: first maximum of nu2[k..c-1] (ColPivHouseholderQR: maxCoeff of the tail)
            int b = k;
            if (nb.pivoting) {
                double best = -1.0;
                int bi = c;
                for (int jc = k + tid; jc < c; jc += WG_THREADS) {
                    const double v = nu2[jc];
                    if (v > best) { best = v; bi = jc; }   // ascending scan keeps the first maximum
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
                for (int w = 1; w < WG_WAVES; ++w)
                    if (red[w] > best || (red[w] == best && ired[w] < bi)) { best = red[w]; bi = ired[w]; }
                b = bi < c ? bi : k;
                if (k == 0) a2 = best;
                for (int jc = k + tid; jc < c; jc += WG_THREADS)
                    if (jc != b && near_best(nu2[jc], thr[jc], best, a2)) *unclear = 1;      // decision (1)
                __syncthreads();
                // Eigen swaps columns k and b physically, and the norm tables with them
                if (b != k) {
                    for (int i = tid; i < r; i += WG_THREADS) {
                        const double tk = W[(int64_t)k * r + i];
                        W[(int64_t)k * r + i] = W[(int64_t)b * r + i];
                        W[(int64_t)b * r + i] = tk;
                    }
                    if (tid == 0) {
                        double tn = nu2[k]; nu2[k] = nu2[b]; nu2[b] = tn;
                        tn = thr[k]; thr[k] = thr[b]; thr[b] = tn;
                        const int tp = pidx[k]; pidx[k] = pidx[b]; pidx[b] = tp;
                    }
                }
                __syncthreads();
            }

            // ---- pivot column to LDS, |tail|^2, reflector scalars (un-normalised form, see bdqr_pair.hip)
            double part = 0.0;
            for (int i = k + tid; i < r; i += WG_THREADS) {
                const double v = W[(int64_t)k * r + i];
                xv[i] = v;
                if (i > k) part = fma(v, v, part);
            }
            const double tailSq = block_sum(part, red);
            const double xk = xv[k];
            double beta, w, g;
            if (k == 0 && !nb.pivoting) a2 = fma(xk, xk, tailSq);
            if (tid == 0 && unclear_reflector(xk, tailSq, k + 1 < r, nb.pivoting != 0, a2)) *unclear = 1;   // (3), (4), (5)
            if (tailSq <= DBL_MIN) {               // Eigen: tau = 0, beta = x0, H = I
                beta = xk; w = 0.0; g = 0.0;
            } else {
                const double nrm = sqrt(fma(xk, xk, tailSq));
                beta = xk >= 0.0 ? -nrm : nrm;
                w = beta - xk;
                g = 1.0 / (beta * w);
            }
            if (tid == 0) {
                W[(int64_t)k * r + k] = beta;
                if (hcoeffs) hcoeffs[cbase + k] = (w * w) * g;   // tau = w / beta
            }

            // ---- update the trailing columns of W and all columns of Q^T
            const int nA = c - k - 1;
            for (int cc = wave; cc < nA + r; cc += WG_WAVES) {
                const bool isA = cc < nA;
                double* col = isA ? W + (int64_t)(k + 1 + cc) * r : QT + (int64_t)(cc - nA) * r;
                double d = 0.0;
                for (int i = k + 1 + lane; i < r; i += 64) d = fma(xv[i], col[i], d);
                d = wave_sum(d);
                const double ck = col[k];
                const double gam = fma(-w, ck, d) * g;
                const double cknew = fma(w, gam, ck);
                double s2 = 0.0;
                for (int i = k + 1 + lane; i < r; i += 64) {
                    const double v = fma(-gam, xv[i], col[i]);
                    col[i] = v;
                    s2 = fma(v, v, s2);
                }
                if (lane == 0) col[k] = cknew;
                if (isA && nb.pivoting) {
                    // LAWN-176 downdate in squared form (ColPivHouseholderQR.h; see bdqr_pair.hip)
                    const int jc = k + 1 + cc;
                    double nn = fma(-cknew, cknew, nu2[jc]);
                    nn = nn > 0.0 ? nn : 0.0;
                    if (nn <= thr[jc]) {
                        s2 = wave_sum(s2);
                        if (lane == 0) {
                            if (in_recompute_band(nn, thr[jc], a2)) *unclear = 1;                 // decision (2)
                            nu2[jc] = s2; thr[jc] = s2 * THR_HI;
                        }
                    } else if (lane == 0) {
                        nu2[jc] = nn;
                    }
                }
            }
            __syncthreads();
        }

        // ---- R (packed upper triangle by columns = CSC value order of m_R) and the permutation splice
        const int n_r = c * (c + 1) / 2;
        for (int jc = wave; jc < c; jc += WG_WAVES)
            for (int i = lane; i <= jc; i += 64) r_vals[roff + (int64_t)jc * (jc + 1) / 2 + i] = W[(int64_t)jc * r + i];
        (void)n_r;
        for (int jc = tid; jc < c; jc += WG_THREADS) perm[cbase + jc] = cbase + pidx[jc];
        if (tid == 0 && *unclear != 0 && redo_count) redo_ids[atomicAdd(redo_count, 1)] = gidx;
        __syncthreads();
    }
}

size_t bdqr_wg_smem_bytes(int max_dim)
{
    return (size_t)(3 * max_dim + 2 * WG_WAVES) * sizeof(double) + (size_t)(max_dim + 2 * WG_WAVES + 2) * sizeof(int);
}

void launch_bdqr_wg(const WaveBatch& nb, const double* tiles, double* q_vals, double* r_vals, int32_t* perm,
                    double* hcoeffs, double* workspace, int64_t ws_stride, int num_wg, int max_dim,
                    int32_t* redo_count, int32_t* redo_ids, hipStream_t stream)
{
    if (nb.num_tiles <= 0) return;
    const int64_t want = nb.num_tiles < (int64_t)num_wg ? nb.num_tiles : (int64_t)num_wg;
    hipLaunchKernelGGL(bdqr_wg_kernel, dim3((unsigned)want), dim3(WG_THREADS), bdqr_wg_smem_bytes(max_dim), stream, nb,
                       tiles, q_vals, r_vals, perm, hcoeffs, workspace, ws_stride, max_dim, redo_count, redo_ids);
}

}  // namespace qrk
