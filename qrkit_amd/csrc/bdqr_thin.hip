// bdqr_thin.hip -- uniform batches of TALL-THIN small tiles (1 or 2 columns, at most 16 rows): one tile per LANE, for gfx950.
//
// The shapes the reference itself runs most: 7 x 2 (test/test-qrkit.cpp:49-51: ColPivHouseholderQRWrapper<Matrix<double, 7, 2>>) and
// the LM-damped 9 x 2 blocks (test/test-utils.cpp:254-274).  Same seam as bdqr_small.hip: the body of the hot loop of
// QRKit::BlockDiagonalSparseQR::factorize (src/QRKit/BlockDiagonalSparseQR.h:432-526) -- blockSolver.compute(block) (:437-438),
// Qi = blockSolver.matrixQ() (:446), the Q / R value assembly (:455-500), the permutation splice (:519-521).
//
// bdqr_small.hip gives such a tile a group of 8 or 16 lanes of which 2 hold columns of A: the batch runs at 20-36 % of the HBM roofline,
// bound by cross-lane traffic and by the staging sweeps through LDS.  With two columns nothing has to cross lanes at all: a lane reads its
// tile (r x 2 doubles, contiguous, 16-byte aligned), keeps both columns in registers, and Q = H0 H1 has the closed form
//     Q(j,k) = delta_jk - tau1 v1(j) v1(k) - tau0 v0(j) w(k),   w = v0 - tau1 (v0 . v1) v1,   v0 = [1; ess0], v1 = [0; 1; ess1]
// (HouseholderSequence::evalTo applies H1 to I and H0 to the result: the same products, summed in a slightly different order), so Q needs
// nothing but the two reflectors: no shuffles, no cross-lane reductions.  Global I/O is staged through LDS per workgroup (256 tiles =
// one contiguous run of `tiles` and of q_vals, row-major Q_i = CSR order of m_Q in both formats), see the kernel.
//
// Arithmetic and decisions as in the other fast kernels (bdqr_pair.hip, "Decisions and the exact path"): squared norms, un-normalised
// reflector; the pivot (2 columns: one comparison), a degenerate reflector on a non-empty tail, a first entry too small to fix the sign
// of beta and a pivot at the noise level are only taken when clear of rounding, else the tile goes to the redo list of the exact path.
// (The LAWN-176 downdate of the second column's norm cannot change any output of a two-column tile and is not evaluated.)
#include "qrk_device.h"

#include <float.h>

namespace qrk {
namespace thin {

using namespace decide;
constexpr int RM = 16;      // rows of a tile at most

// sqrt and 1 / x to <= 1 ulp for positive normal x / normal x, as in the 32 x 32 kernels (bdqr_pair.hip): v_rsq / v_rcp and two
// Newton / Goldschmidt steps instead of the library's sqrt and division sequences (four divisions and two square roots per tile were
// a quarter of the instructions of the arithmetic phase)
__device__ __forceinline__ double sqrt_pos(double x)
{
    const double y = __builtin_amdgcn_rsq(x);
    double g = x * y, h = 0.5 * y;
    const double e = fma(-h, g, 0.5);
    g = fma(g, e, g);
    h = fma(h, e, h);
    const double d = fma(-g, g, x);
    return fma(d, h, g);
}
__device__ __forceinline__ double recip(double x)
{
    double y = __builtin_amdgcn_rcp(x);
    double e = fma(-x, y, 1.0);
    y = fma(y, e, y);
    e = fma(-x, y, 1.0);
    y = fma(y, e, y);
    return y;
}

// One workgroup = 256 lanes = 256 consecutive tiles.  I/O goes through LDS so that every global access is a coalesced sweep over the
// workgroup's contiguous run of `tiles` / `q_vals` (a lane reading or writing its own tile directly touches 64 different lines per
// instruction: measured at 27 % of the roofline whatever the shape):
//   in    the 256 r c doubles of the tiles, copied in index order to LDS (tile t at an odd stride: conflict-free pick-up by its lane);
//   out   every lane leaves v0, v1, tau0, tau1 and tau1 (v0 . v1) of its tile in LDS (2 r + 3 doubles: 43 KB per workgroup at 9 rows, three
//         workgroups per CU; with al = tau0 v0, be = tau1 v1, v1, w precomputed -- 4 r doubles, 76 KB -- the 9 x 2 batch ran 8 % slower), and
//         the workgroup then produces the 256 r^2 entries of Q in OUTPUT order: entry e belongs to tile e / r^2, row (e % r^2) / r, column e % r.
template <bool PIVOT, bool HC>
__global__ void __launch_bounds__(256)
bdqr_thin_kernel(int64_t num_tiles, int r, int c, const double* __restrict__ tiles, double* __restrict__ q_vals, double* __restrict__ r_vals,
                 int32_t* __restrict__ perm, double* __restrict__ hcoeffs, int32_t* __restrict__ redo_count, int32_t* __restrict__ redo_ids)
{
    extern __shared__ double thin_lds[];
    const int tid = threadIdx.x;
    const int rc = r * c, rr = r * r;
    const int sin = rc | 1;              // LDS stride of a tile's input (odd)
    const int sout = (2 * r + 3) | 1;    // LDS stride of a tile's reflector data (odd): v0, v1, tau0, tau1, tau1 (v0 . v1)
    const float inv_rr = 1.0f / (float)rr, inv_r = 1.0f / (float)r;
    for (int64_t t0 = (int64_t)blockIdx.x * 256; t0 < num_tiles; t0 += (int64_t)gridDim.x * 256) {
        const int nt = (int)(num_tiles - t0 < 256 ? num_tiles - t0 : 256);
        // ---- in: coalesced copy of the workgroup's tiles.  All the loads of a thread are issued before the first one is waited for
        // (a loop of load -> wait -> LDS write, one 8-byte word at a time, cost 40 of the 132 us of a 7 x 2 batch: profiles/r05_thin.txt);
        // with an even number of words per tile a 16-byte word never straddles two tiles.  (Addresses are clamped, not the loads
        // predicated: a load under a condition gets its own wait.)
        {
            const double* src = tiles + t0 * (int64_t)rc;
            const int n = nt * rc;
            if ((rc & 1) == 0) {
                typedef double d2 __attribute__((ext_vector_type(2)));
                const d2* src2 = reinterpret_cast<const d2*>(src);             // (t0 is a multiple of 256: 16-byte aligned)
                const int n2 = n >> 1, h2 = rc >> 1;
                const float inv_h2 = 1.0f / (float)h2;
                for (int base = tid; base < n2; base += 256 * 8) {
                    d2 v[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u) { const int e = base + 256 * u; v[u] = QRK_TILE_LOAD(src2 + (e < n2 ? e : n2 - 1)); }
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        const int e = base + 256 * u;
                        if (e < n2) {
                            const int tl = (int)(((float)e + 0.5f) * inv_h2);
                            double* o = thin_lds + tl * sin + 2 * (e - tl * h2);
                            o[0] = v[u].x; o[1] = v[u].y;
                        }
                    }
                }
            } else {
                const float inv_rc = 1.0f / (float)rc;
                for (int base = tid; base < n; base += 256 * 8) {
                    double v[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u) { const int e = base + 256 * u; v[u] = QRK_TILE_LOAD(src + (e < n ? e : n - 1)); }
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        const int e = base + 256 * u;
                        if (e < n) {
                            const int tl = (int)(((float)e + 0.5f) * inv_rc);
                            thin_lds[tl * sin + (e - tl * rc)] = v[u];
                        }
                    }
                }
            }
        }
        __syncthreads();
        const int64_t t = t0 + tid;
        const bool have = tid < nt;
        double a0[RM], a1[RM];
#pragma unroll
        for (int i = 0; i < RM; ++i) {
            a0[i] = (have && i < r) ? thin_lds[tid * sin + i] : 0.0;
            a1[i] = (have && c > 1 && i < r) ? thin_lds[tid * sin + r + i] : 0.0;
        }
        __syncthreads();                 // (the input area is dead: the reflector data go over it)
        bool unclear = false;
        int P = 0;
        double a2 = 0.0;
        if (c > 1 && PIVOT) {
            // first maximum of the two squared column norms; a tie or a near tie is Eigen's to decide (exact path)
            double n0 = 0.0, n1 = 0.0;
#pragma unroll
            for (int i = 0; i < RM; ++i) { n0 = fma(a0[i], a0[i], n0); n1 = fma(a1[i], a1[i], n1); }
            P = n1 > n0 ? 1 : 0;
            const double best = P ? n1 : n0, other = P ? n0 : n1;
            a2 = best;
            if (near_best(other, other * THR_HI, best, a2)) unclear = true;
            if (P) {
#pragma unroll
                for (int i = 0; i < RM; ++i) { const double tmp = a0[i]; a0[i] = a1[i]; a1[i] = tmp; }
            }
        }
        // ---- reflector 0 on a0 (makeHouseholder, Eigen/src/Householder/Householder.h)
        double tsq = 0.0;
#pragma unroll
        for (int i = 1; i < RM; ++i) tsq = fma(a0[i], a0[i], tsq);
        const double x0 = a0[0];
        if (!(c > 1 && PIVOT)) a2 = fma(x0, x0, tsq);
        if (have && unclear_reflector(x0, tsq, r > 1, PIVOT, a2)) unclear = true;
        double tau0, beta0, inv0;
        if (!(tsq > DBL_MIN)) { tau0 = 0.0; beta0 = x0; inv0 = 0.0; }
        else {
            beta0 = sqrt_pos(fma(x0, x0, tsq));
            if (x0 >= 0.0) beta0 = -beta0;
            inv0 = recip(x0 - beta0);
            tau0 = (beta0 - x0) * recip(beta0);
        }
        a0[0] = 1.0;                                      // v0 = [1; essential] in place
#pragma unroll
        for (int i = 1; i < RM; ++i) a0[i] *= inv0;
        double tau1 = 0.0, beta1 = 0.0, r01 = 0.0, s01 = 0.0;
        if (c > 1) {
            // applyHouseholderOnTheLeft to the other column: tmp = y0 + ess^T y_tail; y0 -= tau tmp; y_tail -= tau tmp ess
            double tmp = a1[0];
#pragma unroll
            for (int i = 1; i < RM; ++i) tmp = fma(a0[i], a1[i], tmp);
            const double g = tau0 * tmp;
            r01 = a1[0] - g;
#pragma unroll
            for (int i = 1; i < RM; ++i) a1[i] = fma(-g, a0[i], a1[i]);
            // ---- reflector 1 on a1(1:)
            double tsq1 = 0.0;
#pragma unroll
            for (int i = 2; i < RM; ++i) tsq1 = fma(a1[i], a1[i], tsq1);
            const double y0 = a1[1];
            if (have && unclear_reflector(y0, tsq1, r > 2, PIVOT, a2)) unclear = true;
            double inv1;
            if (!(tsq1 > DBL_MIN)) { tau1 = 0.0; beta1 = y0; inv1 = 0.0; }
            else {
                beta1 = sqrt_pos(fma(y0, y0, tsq1));
                if (y0 >= 0.0) beta1 = -beta1;
                inv1 = recip(y0 - beta1);
                tau1 = (beta1 - y0) * recip(beta1);
            }
            a1[0] = 0.0; a1[1] = 1.0;                     // v1 = [0; 1; essential]
#pragma unroll
            for (int i = 2; i < RM; ++i) a1[i] *= inv1;
#pragma unroll
            for (int i = 1; i < RM; ++i) s01 = fma(a0[i], a1[i], s01);      // v0 . v1 (v1(0) = 0)
        }
        if (have) {
            // ---- a decision inside its error margin: the exact path redoes the tile (its outputs below are overwritten)
            if (unclear && redo_count) redo_ids[atomicAdd(redo_count, 1)] = (int32_t)t;
            // ---- R (packed upper triangle by columns), permutation, tau
            double* rv = r_vals + t * (int64_t)(c * (c + 1) / 2);
            rv[0] = beta0;
            if (c > 1) { rv[1] = r01; rv[2] = beta1; }
            const int cbase = (int)(t * c);
            perm[cbase] = cbase + P;
            if (c > 1) perm[cbase + 1] = cbase + 1 - P;
            if (HC && hcoeffs) { hcoeffs[cbase] = tau0; if (c > 1) hcoeffs[cbase + 1] = tau1; }
            // ---- the two reflectors and their scalars of this tile
            double* o = thin_lds + tid * sout;
#pragma unroll
            for (int i = 0; i < RM; ++i)
                if (i < r) { o[i] = a0[i]; o[r + i] = a1[i]; }
            o[2 * r] = tau0; o[2 * r + 1] = tau1; o[2 * r + 2] = tau1 * s01;
        }
        __syncthreads();
        // ---- Q in output order: Q(j,k) = delta_jk - be(j) v1(k) - al(j) w(k)
        {
            double* dst = q_vals + t0 * (int64_t)rr;
            const int n = nt * rr;
            for (int e = tid; e < n; e += 256) {
                const int tl = (int)(((float)e + 0.5f) * inv_rr);
                const int rem = e - tl * rr;
                const int j = (int)(((float)rem + 0.5f) * inv_r);
                const int k = rem - j * r;
                const double* o = thin_lds + tl * sout;
                const double v1k = o[r + k];
                const double wk = fma(-o[2 * r + 2], v1k, o[k]);                    // w(k) = v0(k) - tau1 (v0 . v1) v1(k)
                QRK_OUT_STORE(dst + e, fma(-(o[2 * r + 1] * o[r + j]), v1k, fma(-(o[2 * r] * o[j]), wk, j == k ? 1.0 : 0.0)));
            }
        }
        __syncthreads();
    }
}

}  // namespace thin

void launch_bdqr_thin(int64_t num_tiles, int r, int c, int pivoting, const double* tiles, double* q_vals, double* r_vals,
                      int32_t* perm, double* hcoeffs, int max_blocks, int32_t* redo_count, int32_t* redo_ids, hipStream_t stream)
{
    if (num_tiles <= 0) return;
    int64_t nwg = (num_tiles + 255) / 256;
    if (max_blocks > 0 && nwg > max_blocks) nwg = max_blocks;
    const dim3 grid((unsigned)nwg), block(256);
    const int sin = (r * c) | 1, sout = (2 * r + 3) | 1;
    const size_t smem = (size_t)256 * (sin > sout ? sin : sout) * sizeof(double);      // 43 KB at 9 x 2; <= 256 x 35 x 8 = 72 KB at 16 rows
    if (smem > 64 * 1024) {
        static hipError_t attr = [] {
            hipError_t e = hipSuccess;
            const void* fns[4] = {reinterpret_cast<const void*>(thin::bdqr_thin_kernel<true, true>), reinterpret_cast<const void*>(thin::bdqr_thin_kernel<true, false>),
                                  reinterpret_cast<const void*>(thin::bdqr_thin_kernel<false, true>), reinterpret_cast<const void*>(thin::bdqr_thin_kernel<false, false>)};
            for (const void* f : fns) if (e == hipSuccess) e = hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 4096);
            return e;
        }();
        (void)attr;
    }
#define QRK_THIN(P, H) \
    hipLaunchKernelGGL((thin::bdqr_thin_kernel<P, H>), grid, block, smem, stream, num_tiles, r, c, tiles, q_vals, r_vals, perm, hcoeffs, redo_count, redo_ids)
    if (pivoting) { if (hcoeffs) QRK_THIN(true, true); else QRK_THIN(true, false); }
    else { if (hcoeffs) QRK_THIN(false, true); else QRK_THIN(false, false); }
#undef QRK_THIN
}

}  // namespace qrk
