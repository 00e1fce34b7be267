// dense_qr_cols.hip -- column-pivoted (or plain) Householder QR of a SQUARE-ISH dense matrix whose columns fit LDS
// (rows <= 8000), one kernel per reflector, a wavefront per column.
//
// Where it sits: second stage of the two-stage factorisation of a tall dense right block (caqr.hip;
// QRKit::BlockAngularSparseQR, rightSolver.compute(J2.bottomRows(...)), src/QRKit/BlockAngularSparseQR.h:361-369): the n x n
// factor R0 of the un-pivoted first stage is factorised with Eigen's ColPivHouseholderQR rule, R0 P = Q1 R.  The row-slab
// kernels of dense_qr_tall.hip are built for 40 000 rows: on 2 000 x 2 000 they occupy 16 of 256 CUs and need three kernels per
// reflector (141 ms).  Here the parallel axis is the COLUMN:
//   * every workgroup repeats the small serial part of the step itself -- first maximum of the bookkeeping norms, the pivot
//     column into LDS, |x_tail|^2, the reflector scalars (same data, same order of operations: bitwise the same everywhere) --
//     so there is no head kernel and no cross-workgroup reduction;
//   * then one wavefront per remaining column: dot with x, row k of R, rank-1 update, LAWN-176 downdate of its squared norm, and
//     when Eigen's recompute test fires the wave has the updated column at hand and recomputes the norm itself (no slow path);
//   * columns are never moved: positions map to physical columns through a ping-pong table (step parity), the pivot column is
//     read-only during its own step (its essential part and beta are written by the closing kernel, which also gathers the
//     columns into pivoted order = Eigen's packed format).
// Pivot bookkeeping, squared norms and the decision margins are those of dense_qr_tall.hip / bdqr_pair.hip: a decision inside
// its rounding margin sets `unclear` and the exact path redoes the factorisation in Eigen's operation order.
#include "qrk_device.h"

#include <float.h>
#include <cstdlib>

namespace qrk {
namespace cols {

constexpr int TT = 512;                  // 8 wavefronts: the register-resident column (64 VGPRs) needs more than the 128 VGPRs of a 1024-thread workgroup
constexpr int TW = TT / 64;
using namespace decide;

struct State {
    double a2;       // |A|^2: squared norm of the first pivot column (scale of the decision margins)
    int unclear;
};

struct Work {
    double* nu2[2];   // [cpad] m_colNormsUpdated^2 by POSITION, ping-pong on the parity of the step
    double* thr[2];   // [cpad] sqrt(eps) (1 + 2^-12) m_colNormsDirect^2
    int* cmap[2];     // [cpad] physical column at a position
    double* beta;     // [cpad] per reflector: beta (the diagonal of R)
    double* invs;     // [cpad] per reflector: 1 / (x0 - beta), 0 if H = I
    State* st;
};

// Sum over the 64 lanes, the same value in every lane: four DPP steps inside the rows of 16, then the four row sums through
// SGPRs (a fixed order; no LDS round trips -- a step of this kernel is a chain of such reductions).
__device__ __forceinline__ double wave_sum(double v)
{
    v += dpp_f64<0xB1>(v);    // quad_perm [1,0,3,2]
    v += dpp_f64<0x4E>(v);    // quad_perm [2,3,0,1]
    v += dpp_f64<0x141>(v);   // row_half_mirror
    v += dpp_f64<0x140>(v);   // row_mirror
    return (readlane_f64(v, 0) + readlane_f64(v, 16)) + (readlane_f64(v, 32) + readlane_f64(v, 48));
}

__global__ void __launch_bounds__(TT)
cols_init_kernel(const double* __restrict__ A, int64_t lda, int r, int c, Work w)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int jc = blockIdx.x * TW + wave;
    if (blockIdx.x == 0 && threadIdx.x == 0) { w.st->a2 = 0.0; w.st->unclear = 0; }
    if (jc >= c) return;
    double s = 0.0;
    for (int i = lane; i < r; i += 64) { const double v = A[(int64_t)jc * lda + i]; s = fma(v, v, s); }
    s = wave_sum(s);
    if (lane == 0) { w.nu2[0][jc] = s; w.thr[0][jc] = s * THR_HI; w.cmap[0][jc] = jc; }
}

// Step k.  Grid: ceil((c - k - 1) / 8) workgroups (at least one); dynamic LDS: (r - k) doubles.
// RPL: a lane keeps up to RPL entries of its column in registers between the dot and the update (32: columns of <= 2048 rows below the
// pivot, 64: <= 4096); longer columns are read twice.
template <int RPL, int STT>
__global__ void __launch_bounds__(STT) __attribute__((amdgpu_waves_per_eu(1, STT == 256 ? 1 : 2)))
cols_step_kernel(double* __restrict__ A, int64_t lda, int r, int c, int k, int pivoting, double* __restrict__ hcoeffs,
                 int32_t* __restrict__ perm, Work w)
{
    constexpr int STW = STT / 64;
    extern __shared__ double xs[];            // x = rows k .. r-1 of the pivot column
    __shared__ double red[STW];
    __shared__ int ired[STW];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int b = k & 1;
    // ---- this wave's column, assuming its position is not the one the pivot is swapped with (true for all waves but one):
    // the loads are issued before the serial part of the step, whose memory round trips they overlap
    const int pos = k + 1 + blockIdx.x * STW + wave;
    const int n = r - k - 1;                      // rows below the pivot row
    const bool cached = n <= 64 * RPL;
    double a[RPL];
    double ak = 0.0;
    int pc = 0;
    if (pos < c && cached) {
        pc = w.cmap[b][pos];
        const double* col0 = A + (int64_t)pc * lda;
#pragma unroll
        for (int q = 0; q < RPL; ++q) { const int i = q * 64 + lane; a[q] = i < n ? col0[k + 1 + i] : 0.0; }
        ak = col0[k];
    }
    // ---- the pivot: first maximum of the bookkeeping norms over positions k .. c-1 (larger wins, ties -> smaller position)
    int P = k;
    int pk = -1;                              // physical column of the pivot: travels with the candidate through the reduction, so that
                                              // no dependent look-up follows the search
    double best = 0.0;
    if (pivoting) {
        __shared__ int cred[STW];
        best = -1.0; int bi = c, bc = -1;
        for (int pos = k + tid; pos < c; pos += STT) {
            const double v = w.nu2[b][pos];
            const int cc = w.cmap[b][pos];
            if (v > best) { best = v; bi = pos; bc = cc; }
        }
        {   // the wave's first maximum without LDS round trips: the maximum by DPP steps and four readlanes, the smallest position among
            // the lanes that hold it the same way, its physical column from the lane a ballot names
            double m = row16_max(best);
            m = fmax(fmax(readlane_f64(m, 0), readlane_f64(m, 16)), fmax(readlane_f64(m, 32), readlane_f64(m, 48)));
            int p = best == m ? bi : 0x7fffffff;
            p = min(p, dpp_i32<0xB1>(p));
            p = min(p, dpp_i32<0x4E>(p));
            p = min(p, dpp_i32<0x141>(p));
            p = min(p, dpp_i32<0x140>(p));
            p = min(min(__builtin_amdgcn_readlane(p, 0), __builtin_amdgcn_readlane(p, 16)),
                    min(__builtin_amdgcn_readlane(p, 32), __builtin_amdgcn_readlane(p, 48)));
            const unsigned long long who = __ballot(best == m && bi == p);
            const int src = who ? (int)__builtin_ctzll(who) : 0;             // (no lane: NaN norms; any lane will do)
            bc = __builtin_amdgcn_readlane(bc, src);
            best = m; bi = p;
        }
        if (lane == 0) { red[wave] = best; ired[wave] = bi; cred[wave] = bc; }
        __syncthreads();
        best = red[0]; bi = ired[0]; bc = cred[0];
#pragma unroll
        for (int q = 1; q < STW; ++q) if (red[q] > best || (red[q] == best && ired[q] < bi)) { best = red[q]; bi = ired[q]; bc = cred[q]; }
        if (bi < c) { P = bi; pk = bc; }
        __syncthreads();
    }
    const double a2_in = w.st->a2;            // written by step 0 (a kernel boundary ago)
    if (pk < 0) pk = w.cmap[b][P];
    // ---- x into LDS, |x_tail|^2
    double t = 0.0;
    for (int i = k + tid; i < r; i += STT) {
        const double v = A[(int64_t)pk * lda + i];
        xs[i - k] = v;
        if (i > k) t = fma(v, v, t);
    }
    t = wave_sum(t);
    if (lane == 0) red[wave] = t;
    __syncthreads();
    double tsq = 0.0;
#pragma unroll
    for (int q = 0; q < STW; ++q) tsq += red[q];
    const double xk = xs[0];
    // makeHouseholder in the un-normalised form of bdqr_pair.hip: nb = -beta, s = x0 - beta, ng = -1/(beta w)
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
    const double a2 = pivoting ? (k == 0 ? best : a2_in) : (k == 0 ? fma(xk, xk, tsq) : a2_in);
    if (blockIdx.x == 0) {
        if (tid == 0) {
            if (k == 0) w.st->a2 = a2;
            w.beta[k] = -nb;
            w.invs[k] = degen ? 0.0 : 1.0 / s;
            hcoeffs[k] = tau;
            perm[k] = pk;                                        // colsPermutation().indices()(k)
            if (unclear_reflector(xk, tsq, k + 1 < r, pivoting != 0, a2, (pivoting & PIVOTING_SIGN_FREE) != 0))    // decisions (3), (4), (5)
                atomicOr(&w.st->unclear, 4 | (!(tsq > DBL_MIN) ? 8 : 0) | (xk * xk <= X0_TINY2 * a2 ? 16 : 0) | (fma(xk, xk, tsq) <= PIV_TINY2 * a2 ? 32 : 0));
        }
    }
    // ---- one wavefront per remaining position
    if (pos >= c) return;
    const int sp = pos == P ? k : pos;            // the swap of positions k and P
    if (sp != pos || !cached) pc = w.cmap[b][sp];
    double* col = A + (int64_t)pc * lda;
    double nn = 0.0, th = 0.0;
    if (cached) {
        // the column stays in registers between the dot and the update: one read, one write
        if (sp != pos) {
#pragma unroll
            for (int q = 0; q < RPL; ++q) { const int i = q * 64 + lane; a[q] = i < n ? col[k + 1 + i] : 0.0; }
            ak = col[k];
        }
        double d0 = 0.0, d1 = 0.0;
#pragma unroll
        for (int q = 0; q < RPL; q += 2) {
            const int i = q * 64 + lane;
            d0 = fma(i < n ? xs[1 + i] : 0.0, a[q], d0);
            d1 = fma(i + 64 < n ? xs[1 + i + 64] : 0.0, a[q + 1], d1);
        }
        const double d = wave_sum(d0 + d1);
        const double ngam = fma(s, ak, d) * ng;
        const double an = fma(s, ngam, ak);
        double q0 = 0.0, q1 = 0.0;
#pragma unroll
        for (int q = 0; q < RPL; q += 2) {
            const int i = q * 64 + lane;
            if (i < n) { const double v = fma(ngam, xs[1 + i], a[q]); col[k + 1 + i] = v; q0 = fma(v, v, q0); }
            if (i + 64 < n) { const double v = fma(ngam, xs[1 + i + 64], a[q + 1]); col[k + 1 + i + 64] = v; q1 = fma(v, v, q1); }
        }
        const double sq = wave_sum(q0 + q1);
        if (lane == 0) {
            col[k] = an;                                          // row k of R
            if (pivoting) {
                const double nu_old = w.nu2[b][sp];
                th = w.thr[b][sp];
                // decision (1): this column within the error margin of the chosen one (every remaining column has its wave here)
                if (near_best(nu_old, th, best, a2)) atomicOr(&w.st->unclear, 1);
                nn = fma(-an, an, nu_old);
                if (nn <= th) {                                   // LAWN-176: recompute from the updated column (which is right here)
                    if (in_recompute_band(nn, th, a2)) atomicOr(&w.st->unclear, 2);    // decision (2)
                    nn = sq; th = sq * THR_HI;
                }
            }
        }
    } else {
        ak = col[k];
        constexpr int UW = 8;              // loads in flight per lane
        double dacc[UW];
#pragma unroll
        for (int u = 0; u < UW; ++u) dacc[u] = 0.0;
        for (int i0 = lane; i0 < n; i0 += 64 * UW) {
            double av[UW];
#pragma unroll
            for (int u = 0; u < UW; ++u) av[u] = i0 + 64 * u < n ? col[k + 1 + i0 + 64 * u] : 0.0;
#pragma unroll
            for (int u = 0; u < UW; ++u) dacc[u] = fma(i0 + 64 * u < n ? xs[1 + i0 + 64 * u] : 0.0, av[u], dacc[u]);
        }
        const double d = wave_sum(((dacc[0] + dacc[1]) + (dacc[2] + dacc[3])) + ((dacc[4] + dacc[5]) + (dacc[6] + dacc[7])));
        const double ngam = fma(s, ak, d) * ng;
        const double an = fma(s, ngam, ak);
        double qacc[UW];
#pragma unroll
        for (int u = 0; u < UW; ++u) qacc[u] = 0.0;
        for (int i0 = lane; i0 < n; i0 += 64 * UW) {
            double av[UW];
#pragma unroll
            for (int u = 0; u < UW; ++u) av[u] = i0 + 64 * u < n ? col[k + 1 + i0 + 64 * u] : 0.0;
#pragma unroll
            for (int u = 0; u < UW; ++u)
                if (i0 + 64 * u < n) { const double v = fma(ngam, xs[1 + i0 + 64 * u], av[u]); col[k + 1 + i0 + 64 * u] = v; qacc[u] = fma(v, v, qacc[u]); }
        }
        const double sq = wave_sum(((qacc[0] + qacc[1]) + (qacc[2] + qacc[3])) + ((qacc[4] + qacc[5]) + (qacc[6] + qacc[7])));
        if (lane == 0) {
            col[k] = an;
            if (pivoting) {
                const double nu_old = w.nu2[b][sp];
                th = w.thr[b][sp];
                if (near_best(nu_old, th, best, a2)) atomicOr(&w.st->unclear, 1);      // decision (1)
                nn = fma(-an, an, nu_old);
                if (nn <= th) {
                    if (in_recompute_band(nn, th, a2)) atomicOr(&w.st->unclear, 2);
                    nn = sq; th = sq * THR_HI;
                }
            }
        }
    }
    if (lane == 0) { w.nu2[b ^ 1][pos] = nn; w.thr[b ^ 1][pos] = th; w.cmap[b ^ 1][pos] = pc; }
}

// Closing pass: Eigen's packed format in pivoted column order, out(:, pos) from the physical column perm[pos]: rows above the
// diagonal as they are (rows of R), beta on the diagonal, the essential part x_tail / (x0 - beta) below.
__global__ void __launch_bounds__(256)
cols_finish_kernel(const double* __restrict__ A, int64_t lda, int r, int c, int size, const int32_t* __restrict__ perm, Work w,
                   double* __restrict__ out, int64_t ldo)
{
    const int pos = blockIdx.x;
    const int pk = perm[pos];
    const bool refl = pos < size;
    const double beta = refl ? w.beta[pos] : 0.0, inv = refl ? w.invs[pos] : 0.0;
    for (int i = threadIdx.x; i < r; i += 256) {
        double v = A[(int64_t)pk * lda + i];
        if (refl) { if (i == pos) v = beta; else if (i > pos) v *= inv; }
        out[(int64_t)pos * ldo + i] = v;
    }
}
// positions size .. c-1 of a wide matrix keep their final bookkeeping order
__global__ void cols_tail_perm_kernel(int c, int size, Work w, int32_t* __restrict__ perm)
{
    const int pos = size + blockIdx.x * blockDim.x + threadIdx.x;
    if (pos < c) perm[pos] = w.cmap[size & 1][pos];
}

}  // namespace cols

// (the persistent form of dense_qr_pers.hip shares the workspace: its synchronisation words and slots follow the tables)
static size_t cols_tables_bytes(int cpad) { return ((size_t)cpad * (6 * sizeof(double) + 2 * sizeof(int)) + 256 + 255) / 256 * 256; }
// pers: the plan may run the persistent form of dense_qr_pers.hip (opt-in, decided ONCE at plan creation: dense_pers_supported with
// the handle's own CU count); only then does the workspace carry its 16.8 MB of synchronisation words and slots
size_t dense_cols_workspace_bytes(int c, int* cpad_out, bool pers)
{
    const int cpad = (c + 63) / 64 * 64;
    *cpad_out = cpad;
    return cols_tables_bytes(cpad) + (pers ? dense_pers_workspace_bytes() : 0);
}
bool dense_cols_supported(int r, int c) { return r >= 1 && r <= 8000 && c >= 1; }
int* dense_cols_unclear_ptr(void* workspace, int cpad)
{
    char* p = static_cast<char*>(workspace) + (size_t)cpad * (6 * sizeof(double) + 2 * sizeof(int));
    p = reinterpret_cast<char*>((reinterpret_cast<uintptr_t>(p) + 63) & ~(uintptr_t)63);
    return &reinterpret_cast<cols::State*>(p)->unclear;
}

// A (r x c, column-major, r <= 8000) is the input and the scratch of the factorisation; `out` (ldo >= r) receives Eigen's packed QR
// in pivoted column order, hcoeffs the tau, perm the column permutation.
hipError_t launch_dense_qr_cols(double* A, int64_t lda, int r, int c, int pivoting, double* hcoeffs, int32_t* perm, void* workspace,
                                int cpad, double* out, int64_t ldo, int pers_cus, hipStream_t stream)
{
    using namespace cols;
    // up to 2048 x 2048: ONE persistent launch with the matrix in registers (dense_qr_pers.hip) instead of a launch per reflector
    // (pers_cus > 0: the plan was created with the persistent form enabled and its workspace, for a device of that many CUs)
    if (pers_cus > 0 && dense_pers_supported(r, c, pers_cus))
        return launch_dense_qr_pers(A, lda, r, c, pivoting, hcoeffs, perm, dense_cols_unclear_ptr(workspace, cpad) - 2 /* State: {double a2; int unclear} */,
                                    static_cast<char*>(workspace) + cols_tables_bytes(cpad), pers_cus, out, ldo, stream);
    Work w;
    char* p = static_cast<char*>(workspace);
    for (int q = 0; q < 2; ++q) { w.nu2[q] = reinterpret_cast<double*>(p); p += (size_t)cpad * sizeof(double); }
    for (int q = 0; q < 2; ++q) { w.thr[q] = reinterpret_cast<double*>(p); p += (size_t)cpad * sizeof(double); }
    w.beta = reinterpret_cast<double*>(p); p += (size_t)cpad * sizeof(double);
    w.invs = reinterpret_cast<double*>(p); p += (size_t)cpad * sizeof(double);
    for (int q = 0; q < 2; ++q) { w.cmap[q] = reinterpret_cast<int*>(p); p += (size_t)cpad * sizeof(int); }
    p = reinterpret_cast<char*>((reinterpret_cast<uintptr_t>(p) + 63) & ~(uintptr_t)63);
    w.st = reinterpret_cast<State*>(p);
    const int size = r < c ? r : c;
    hipLaunchKernelGGL(cols_init_kernel, dim3((c + TW - 1) / TW), dim3(TT), 0, stream, A, lda, r, c, w);
    for (int k = 0; k < size; ++k) {
        // rows below the pivot: up to 2048 -> 8 waves per workgroup with 32 entries of the column per lane in registers; 4097 ... 6144 ->
        // 4 waves with 96 entries, rows up to 6144 (one wave per SIMD, 512 registers: the column is read once and written once instead of read twice; the
        // reference's own block-angular test size, a 5120 x 384 bottom block, is such a matrix)
        const bool longcol = r - k - 1 > 4096 && r - k - 1 <= 64 * 96 && !std::getenv("QRK_COLS_NO_LONG");
        const int tw = longcol ? 4 : TW;
        int nwg = (c - k - 1 + tw - 1) / tw;
        if (nwg < 1) nwg = 1;
        // (RPL = 64, columns of up to 4096 rows in registers, was measured slower: 256 VGPRs and spills: 3000 x 300 in 10.3 ms against 6)
        if (longcol)
            hipLaunchKernelGGL((cols_step_kernel<96, 256>), dim3(nwg), dim3(256), (size_t)(r - k) * sizeof(double), stream, A, lda, r, c, k,
                               pivoting, hcoeffs, perm, w);
        else
            hipLaunchKernelGGL((cols_step_kernel<32, TT>), dim3(nwg), dim3(TT), (size_t)(r - k) * sizeof(double), stream, A, lda, r, c, k,
                               pivoting, hcoeffs, perm, w);
    }
    if (c > size) hipLaunchKernelGGL(cols_tail_perm_kernel, dim3((c - size + 255) / 256), dim3(256), 0, stream, c, size, w, perm);
    hipLaunchKernelGGL(cols_finish_kernel, dim3(c), dim3(256), 0, stream, A, lda, r, c, size, perm, w, out, ldo);
    return hipGetLastError();
}

}  // namespace qrk
