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

        // ---- Y = unit-lower essentials (:471-475)
        double* Y = y_vals + p.y_off;
        for (int64_t e = tid; e < (int64_t)m * n; e += BB_THREADS) {
            const int i = (int)(e % m), j = (int)(e / m);
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

__global__ void __launch_bounds__(256)
bb_gather_r_kernel(const double* __restrict__ r_stage, const int64_t* __restrict__ r_src, int64_t nnz,
                   double* __restrict__ r_vals)
{
    const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (i < nnz) r_vals[i] = r_stage[r_src[i]];
}

// One workgroup per right-hand side: seg += Y (T^(T) (Y^T seg)) for every block in order.
__global__ void __launch_bounds__(BB_THREADS)
bb_apply_q_kernel(const BBPanel* __restrict__ panels, int num_panels, const double* __restrict__ y_vals,
                  const double* __restrict__ t_vals, int transpose, double* __restrict__ v, int64_t ldv, int64_t nrhs,
                  int max_act_rows, int max_ncols)
{
    extern __shared__ __attribute__((aligned(16))) double smem[];
    double* seg = smem;                      // [max_act_rows]
    double* w1 = seg + max_act_rows;         // [max_ncols]
    double* w2 = w1 + max_ncols;             // [max_ncols]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int64_t col = blockIdx.x; col < nrhs; col += gridDim.x) {
        double* x = v + col * ldv;
        for (int s = 0; s < num_panels; ++s) {
            const BBPanel p = panels[transpose ? s : num_panels - 1 - s];
            const int m = p.act_rows, n = p.ncols;
            const int seg2 = p.yrow + n + p.num_zeros;     // start of the second row segment
            const double* Y = y_vals + p.y_off;
            const double* T = t_vals + p.t_off;
            for (int i = tid; i < m; i += BB_THREADS) seg[i] = x[i < n ? p.yrow + i : seg2 + (i - n)];
            __syncthreads();
            for (int j = wave; j < n; j += BB_WAVES) {
                double d = 0.0;
                for (int i = lane; i < m; i += 64) d = fma(Y[(int64_t)j * m + i], seg[i], d);
                d = bb_wave_sum(d);
                if (lane == 0) w1[j] = d;
            }
            __syncthreads();
            for (int i = tid; i < n; i += BB_THREADS) {
                double d = 0.0;
                if (transpose) { for (int j = 0; j <= i; ++j) d = fma(T[(int64_t)i * n + j], w1[j], d); }   // T^T w
                else { for (int j = i; j < n; ++j) d = fma(T[(int64_t)j * n + i], w1[j], d); }               // T w
                w2[i] = d;
            }
            __syncthreads();
            for (int i = tid; i < m; i += BB_THREADS) {
                double d = seg[i];
                for (int j = 0; j < n; ++j) d = fma(Y[(int64_t)j * m + i], w2[j], d);
                x[i < n ? p.yrow + i : seg2 + (i - n)] = d;
            }
            __syncthreads();
        }
    }
}

size_t bb_chain_smem(int max_act_rows, int max_ncols) { return (size_t)(max_act_rows + 2 * max_ncols + BB_WAVES) * sizeof(double); }
size_t bb_apply_smem(int max_act_rows, int max_ncols) { return (size_t)(max_act_rows + 2 * max_ncols) * sizeof(double); }

hipError_t launch_bb_chain(const BBPanel* panels, int num_panels, const int32_t* prowptr, const int32_t* pcol,
                           const int64_t* pmap, const double* vals, double* W, double* lo, double* y_vals, double* t_vals,
                           double* r_stage, const int64_t* r_src, int64_t nnz_r, double* r_vals, int max_act_rows,
                           int max_ncols, hipStream_t stream)
{
    const size_t smem = bb_chain_smem(max_act_rows, max_ncols);
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(bb_chain_kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(bb_chain_kernel, dim3(1), dim3(BB_THREADS), smem, stream, panels, num_panels, prowptr, pcol, pmap, vals,
                       W, lo, y_vals, t_vals, r_stage, max_act_rows, max_ncols);
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
    hipLaunchKernelGGL(bb_apply_q_kernel, dim3(grid), dim3(BB_THREADS), smem, stream, panels, num_panels, y_vals, t_vals,
                       transpose, v, ldv, nrhs, max_act_rows, max_ncols);
    return hipGetLastError();
}

}  // namespace qrk
