// bdqr_col.hip -- one workgroup factorises one tile with 32 < max(rows, cols) <= 256 (rows >= cols) of a
// block-diagonal matrix: A_i P_i = Q_i R_i with explicit Q_i, for gfx950.
//
// Same reference seam as bdqr_pair.hip (the hot loop of BlockDiagonalSparseQR::factorize,
// src/QRKit/BlockDiagonalSparseQR.h:432-526, Eigen ColPivHouseholderQR / HouseholderQR behind
// blockSolver.compute and HouseholderSequence behind matrixQ()), for the mid-size tiles of mixed
// batches (BASELINE configs[4]: sizes 8..256).
//
// One THREAD per column of the working matrix W = [A | Q^T] (rows x (cols + rows), ROW-major, so that the
// threads of a wave touch consecutive addresses).  W lives in LDS when it fits in 64 KB (tiles up to
// 64x64), otherwise in a per-workgroup global workspace.  With 512 workgroups in flight that workspace
// (256 KB for a 128x128 tile) does not fit the L2, and the kernel then runs at the bandwidth of the
// Infinity Cache / HBM (measured 6.8 TB/s of sweep traffic on 128x128 tiles, 1.1 TFLOP/s): the next
// step for these sizes is the panel-blocked (dlaqps-style) variant, which reads only the A part once per
// column and applies the updates as GEMMs.  A reflector
// is the same operation on every column, c <- c - gamma x, so A -> R and I -> Q^T advance together and
// no cross-lane reduction is needed: every thread walks down its own column.
//
// The level-2 algorithm needs two sweeps over the trailing matrix per step (dot products, then the
// update).  They are fused across steps into ONE read-modify-write sweep: the dot products of step k
// give row k of the updated matrix (c_k + w gamma), that row is all the LAWN-176 norm downdate needs,
// so the pivot of step k+1 is known BEFORE the update of step k is applied; its column is brought up
// to date on the fly (x' = W(:,p') - gamma_p' x) and the sweep that applies update k also accumulates
// the dot products with x' (and |x'_tail|^2, and the recomputed column norms).  Only when Eigen's
// recompute test fires (rare) the step falls back to separate sweeps, because then the next pivot
// depends on the recomputed norms.
//
// Columns are never swapped: the column chosen at step k ends at position k; the "first maximum" tie
// rule compares current positions, which every A thread tracks (Eigen's transpositions).
#include "qrk_device.h"

#include <float.h>

namespace qrk {

namespace col {

constexpr double SQRT_EPS = 1.4901161193847656e-08;   // sqrt(DBL_EPSILON), Eigen's norm_downdate_threshold
constexpr int W_LDS_DOUBLES = 8192;                   // 64 KB of LDS for W when the tile fits

struct Cand {          // candidate of the pivot search
    double val;        // squared updated norm, < 0 = none
    int pos;           // current position (tie rule: smallest)
    int tidx;          // owning thread = original column
    double ngam;       // -gamma of that column in the step being applied
};

__device__ __forceinline__ bool better(const Cand& a, const Cand& b)   // a beats b
{
    return a.val > b.val || (a.val == b.val && a.pos < b.pos);
}

__device__ __forceinline__ Cand wave_best(Cand c)
{
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        Cand o;
        o.val = __shfl_xor(c.val, off);
        o.pos = __shfl_xor(c.pos, off);
        o.tidx = __shfl_xor(c.tidx, off);
        o.ngam = __shfl_xor(c.ngam, off);
        if (better(o, c)) c = o;
    }
    return c;
}

}  // namespace col

template <int CT>
__global__ void __launch_bounds__(CT)
bdqr_col_kernel(WaveBatch nb, const double* __restrict__ tiles, double* __restrict__ q_vals,
                double* __restrict__ r_vals, int32_t* __restrict__ perm, double* __restrict__ hcoeffs,
                double* __restrict__ workspace, int64_t ws_stride, int max_rows)
{
    using namespace col;
    constexpr int NW = CT / 64;
    extern __shared__ __attribute__((aligned(16))) double smem[];
    double* wl = smem;                                   // [W_LDS_DOUBLES] W when it fits
    double* xv0 = wl + W_LDS_DOUBLES;                    // [max_rows] pivot column, even steps
    double* xv1 = xv0 + max_rows;                        // [max_rows] odd steps
    double* cval = xv1 + max_rows;                       // [NW] candidates of the waves
    double* cngam = cval + NW;                           // [NW]
    int* cpos = reinterpret_cast<int*>(cngam + NW);      // [NW]
    int* ctid = cpos + NW;                               // [NW]
    int* flags = ctid + NW;                              // [2] any-need flags (double buffered)
    int* col_of_pos = flags + 2;                         // [max cols <= max_rows]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;

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
        const int ld = c + r;                             // columns of W (<= CT)
        const bool in_lds = (int64_t)r * ld <= W_LDS_DOUBLES;
        double* W = in_lds ? wl : workspace + (int64_t)blockIdx.x * ws_stride;
        const double* src = tiles + toff;

        // ---- W = [A | I], row-major (A arrives column-major)
        for (int e = tid; e < r * c; e += CT) {
            const int i = e / c, j = e - i * c;
            W[(int64_t)i * ld + j] = src[(int64_t)j * r + i];
        }
        for (int e = tid; e < r * r; e += CT) {
            const int i = e / r, j = e - i * r;
            W[(int64_t)i * ld + c + j] = (i == j) ? 1.0 : 0.0;
        }
        if (tid < 2) flags[tid] = 0;
        __syncthreads();

        const bool mine = tid < ld;          // this thread owns column tid of W
        const bool isA = tid < c;            // ... a column of A
        double* wc = W + tid;                // wc[i * ld] = W(i, tid)
        bool live = isA;
        int pos = tid;                       // current position of an A column
        int kstep = -1;                      // step that chose it = final position
        double nu2 = -1.0, thr = 0.0;
        if (isA) {
            // squared column norms (ColPivHouseholderQR: m_colNormsUpdated^2, sqrt(eps) m_colNormsDirect^2)
            double s = 0.0;
            for (int i = 0; i < r; ++i) { const double v = wc[(int64_t)i * ld]; s = fma(v, v, s); }
            nu2 = s; thr = s * SQRT_EPS;
        }

        // ---- head of step 0: pivot, its column to LDS, dot products
        int P;                               // pivot thread of the current step
        {
            if (nb.pivoting) {
                Cand cd{live ? nu2 : -1.0, pos, tid, 0.0};
                cd = wave_best(cd);
                if (lane == 0) { cval[wave] = cd.val; cpos[wave] = cd.pos; ctid[wave] = cd.tidx; }
                __syncthreads();
                Cand b{cval[0], cpos[0], ctid[0], 0.0};
#pragma unroll
                for (int w = 1; w < NW; ++w) { Cand o{cval[w], cpos[w], ctid[w], 0.0}; if (better(o, b)) b = o; }
                P = b.tidx;
                const int ppos = b.pos;      // old position of the pivot column
                if (isA) { if (tid == P) pos = 0; else if (pos == 0) pos = ppos; }
            } else {
                P = 0;
            }
            if (tid == P) { live = false; kstep = 0; }
            for (int i = tid; i < r; i += CT) xv0[i] = W[(int64_t)i * ld + P];
            __syncthreads();
        }
        double d = 0.0, tsq = 0.0, ak = 0.0;   // tsq = |x_tail|^2, accumulated by every column thread itself
        if (mine) {
            ak = wc[0];
            for (int i = 1; i < r; ++i) { const double xi = xv0[i]; d = fma(xi, wc[(int64_t)i * ld], d); tsq = fma(xi, xi, tsq); }
        }

        for (int k = 0; k < c; ++k) {
            double* xc = (k & 1) ? xv1 : xv0;     // x of this step
            double* xn = (k & 1) ? xv0 : xv1;     // x of the next one
            // ---- makeHouseholder in the un-normalised form of bdqr_pair.hip:
            // nb_ = -beta = copysign(norm, x0), s = -w = nb_ + x0, ng = -1/(beta w); degenerate -> H = I
            const double xk = xc[k];
            double nb_, s, ng;
            bool degen = !(tsq > DBL_MIN);
            if (degen) { nb_ = -xk; s = 0.0; ng = 0.0; }
            else {
                const double nrm = sqrt(fma(xk, xk, tsq));
                nb_ = xk >= 0.0 ? nrm : -nrm;       // Eigen: if (c0 >= 0) beta = -beta  (-0.0 counts as >= 0)
                s = nb_ + xk;
                ng = -1.0 / (nb_ * s);
            }
            const double ngam = mine ? fma(s, ak, d) * ng : 0.0;      // -gamma of this column
            double an = fma(s, ngam, ak);
            if (tid == P && !degen) an = -nb_;                        // R(k,k) = beta
            if (mine) wc[(int64_t)k * ld] = an;                       // row k of W is final
            if (tid == P && hcoeffs) hcoeffs[cbase + k] = -(s * s) * ng;   // tau = w/beta

            if (k + 1 == c) {
                // last reflector: only Q^T still needs its update
                if (mine && !isA) {
                    for (int i = k + 1; i < r; ++i) wc[(int64_t)i * ld] = fma(ngam, xc[i], wc[(int64_t)i * ld]);
                }
                break;
            }

            // ---- LAWN-176 norm downdate (squared form, see bdqr_pair.hip) and the search for step k+1
            bool need = false;
            if (nb.pivoting) {
                if (live) {
                    const double nn = fma(-an, an, nu2);
                    nu2 = nn;
                    need = nn <= thr;
                }
                if (need) flags[k & 1] = 1;
                Cand cd{live ? nu2 : -1.0, pos, tid, ngam};
                cd = wave_best(cd);
                if (lane == 0) { cval[wave] = cd.val; cpos[wave] = cd.pos; ctid[wave] = cd.tidx; cngam[wave] = cd.ngam; }
                if (tid == 0) flags[(k + 1) & 1] = 0;
            }
            __syncthreads();
            const bool any_need = nb.pivoting && flags[k & 1] != 0;
            int Pn;                          // pivot thread of step k+1
            int ppos = k + 1;
            double ngP = 0.0;

            if (!any_need) {
                if (nb.pivoting) {
                    Cand b{cval[0], cpos[0], ctid[0], cngam[0]};
#pragma unroll
                    for (int w = 1; w < NW; ++w) { Cand o{cval[w], cpos[w], ctid[w], cngam[w]}; if (better(o, b)) b = o; }
                    Pn = b.tidx; ppos = b.pos; ngP = b.ngam;
                    // x' = column Pn after update k, built by the threads row-wise
                    for (int i = k + 1 + tid; i < r; i += CT) xn[i] = fma(ngP, xc[i], W[(int64_t)i * ld + Pn]);
                } else {
                    Pn = k + 1;
                    // -gamma of column k+1 is only known to its thread: publish it first
                    if (tid == Pn) cngam[0] = ngam;
                    __syncthreads();
                    ngP = cngam[0];
                    for (int i = k + 1 + tid; i < r; i += CT) xn[i] = fma(ngP, xc[i], W[(int64_t)i * ld + Pn]);
                }
                __syncthreads();
                // ---- fused sweep: apply update k, accumulate the dot products of step k+1
                d = 0.0; tsq = 0.0;
                if (mine) {
                    int i = k + 1;
                    {
                        const double w0 = fma(ngam, xc[i], wc[(int64_t)i * ld]);
                        wc[(int64_t)i * ld] = w0;
                        ak = w0;
                    }
                    for (i = k + 2; i + 3 < r; i += 4) {
                        double w0 = wc[(int64_t)i * ld], w1 = wc[(int64_t)(i + 1) * ld];
                        double w2 = wc[(int64_t)(i + 2) * ld], w3 = wc[(int64_t)(i + 3) * ld];
                        w0 = fma(ngam, xc[i], w0); w1 = fma(ngam, xc[i + 1], w1);
                        w2 = fma(ngam, xc[i + 2], w2); w3 = fma(ngam, xc[i + 3], w3);
                        wc[(int64_t)i * ld] = w0; wc[(int64_t)(i + 1) * ld] = w1;
                        wc[(int64_t)(i + 2) * ld] = w2; wc[(int64_t)(i + 3) * ld] = w3;
                        const double x0 = xn[i], x1 = xn[i + 1], x2 = xn[i + 2], x3 = xn[i + 3];
                        d = fma(x0, w0, d); d = fma(x1, w1, d); d = fma(x2, w2, d); d = fma(x3, w3, d);
                        tsq = fma(x0, x0, tsq); tsq = fma(x1, x1, tsq); tsq = fma(x2, x2, tsq); tsq = fma(x3, x3, tsq);
                    }
                    for (; i < r; ++i) {
                        const double w0 = fma(ngam, xc[i], wc[(int64_t)i * ld]);
                        wc[(int64_t)i * ld] = w0;
                        const double x0 = xn[i];
                        d = fma(x0, w0, d);
                        tsq = fma(x0, x0, tsq);
                    }
                }
            } else {
                // ---- rare: a column norm has to be recomputed from the updated column before the search
                double s2 = 0.0;
                if (mine) {
                    for (int i = k + 1; i < r; ++i) {
                        const double w0 = fma(ngam, xc[i], wc[(int64_t)i * ld]);
                        wc[(int64_t)i * ld] = w0;
                        s2 = fma(w0, w0, s2);
                    }
                }
                if (need) { nu2 = s2; thr = s2 * SQRT_EPS; }
                __syncthreads();
                Cand cd{live ? nu2 : -1.0, pos, tid, 0.0};
                cd = wave_best(cd);
                if (lane == 0) { cval[wave] = cd.val; cpos[wave] = cd.pos; ctid[wave] = cd.tidx; }
                __syncthreads();
                Cand b{cval[0], cpos[0], ctid[0], 0.0};
#pragma unroll
                for (int w = 1; w < NW; ++w) { Cand o{cval[w], cpos[w], ctid[w], 0.0}; if (better(o, b)) b = o; }
                Pn = b.tidx; ppos = b.pos;
                for (int i = k + 1 + tid; i < r; i += CT) xn[i] = W[(int64_t)i * ld + Pn];
                __syncthreads();
                d = 0.0; tsq = 0.0;
                if (mine) {
                    ak = wc[(int64_t)(k + 1) * ld];
                    for (int i = k + 2; i < r; ++i) { const double xi = xn[i]; d = fma(xi, wc[(int64_t)i * ld], d); tsq = fma(xi, xi, tsq); }
                }
            }
            // Eigen swaps columns k+1 and the pivot: the column that sat at k+1 takes the pivot's place
            if (isA && nb.pivoting) { if (tid == Pn) pos = k + 1; else if (pos == k + 1) pos = ppos; }
            if (tid == Pn) { live = false; kstep = k + 1; }
            P = Pn;
            __syncthreads();     // every column is up to date before the next x' is read across threads
        }
        __syncthreads();

        // ---- outputs.  Row i of R is row i of W; column at position p is the thread with kstep == p.
        if (isA) {
            col_of_pos[kstep] = tid;
            perm[cbase + kstep] = cbase + tid;     // m_outputPerm_c.indices()(base_col+j) (:519-521)
        }
        __syncthreads();
        for (int p = wave; p < c; p += NW) {       // packed upper triangle by columns = CSC value order of m_R
            const int tc = col_of_pos[p];
            for (int i = lane; i <= p; i += 64) r_vals[roff + (int64_t)p * (p + 1) / 2 + i] = W[(int64_t)i * ld + tc];
        }
        // Q_i row-major: Q(j, i) = Q^T(i, j) = W(i, c + j)
        for (int e = tid; e < r * r; e += CT) {
            const int j = e / r, i = e - j * r;
            q_vals[qoff + e] = W[(int64_t)i * ld + c + j];
        }
        __syncthreads();
    }
}

size_t bdqr_col_smem_bytes(int max_rows, int threads)
{
    const int nw = threads / 64;
    return (size_t)(col::W_LDS_DOUBLES + 2 * max_rows + 2 * nw) * sizeof(double) +
           (size_t)(2 * nw + 2 + max_rows) * sizeof(int) + 16;
}

hipError_t launch_bdqr_col(const WaveBatch& nb, const double* tiles, double* q_vals, double* r_vals, int32_t* perm,
                           double* hcoeffs, double* workspace, int64_t ws_stride, int num_wg, int max_rows,
                           int max_ld, hipStream_t stream)
{
    if (nb.num_tiles <= 0) return hipSuccess;
    const int64_t want = nb.num_tiles < (int64_t)num_wg ? nb.num_tiles : (int64_t)num_wg;
    const int threads = max_ld <= 128 ? 128 : (max_ld <= 256 ? 256 : 512);
    const size_t smem = bdqr_col_smem_bytes(max_rows, threads);
#define QRK_COL_LAUNCH(T)                                                                                   \
    do {                                                                                                    \
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(bdqr_col_kernel<T>),               \
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);          \
        if (e != hipSuccess) return e;                                                                      \
        hipLaunchKernelGGL((bdqr_col_kernel<T>), dim3((unsigned)want), dim3(T), smem, stream, nb, tiles, q_vals, \
                           r_vals, perm, hcoeffs, workspace, ws_stride, max_rows);                          \
    } while (0)
    if (threads == 128) QRK_COL_LAUNCH(128);
    else if (threads == 256) QRK_COL_LAUNCH(256);
    else QRK_COL_LAUNCH(512);
#undef QRK_COL_LAUNCH
    return hipGetLastError();
}

}  // namespace qrk
