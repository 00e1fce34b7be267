// caqr.hip -- communication-avoiding blocked Householder QR (no pivoting) of a TALL dense column-major matrix on all
// CUs of a gfx950, with the trailing update on the matrix cores (v_mfma_f64_16x16x4_f64).
//
// Where it sits.  QRKit::BlockAngularSparseQR::factorize hands the bottom rows of Q1^T J2 to the right solver,
// rightSolver.compute(J2.bottomRows(...)) (src/QRKit/BlockAngularSparseQR.h:361-369); with the reference's default that is an
// Eigen::ColPivHouseholderQR of ONE tall dense matrix (BASELINE configs[3]: 40 000 x 2 000).  Column pivoting makes that a
// level-2 algorithm -- every reflector needs the whole trailing matrix -- and the device path of round 1 (dense_qr_tall.hip)
// streams 1.3 TB through HBM for it.  For a tall matrix the pivoted factorisation only depends on the Gram structure of the
// columns, which an un-pivoted QR preserves exactly:  A = Q0 R0  (this file, GEMM-shaped, each entry of A read and written once per
// 32 columns)  and then  R0 P = Q1 R  (the pivoted level-2 kernels on the n x n triangle: 1/20 of the bytes)  give
// A P = (Q0 Q1) R with the same pivots and the same R as the direct algorithm up to rounding and up to the sign of each row of R
// (the sign of a Householder beta follows the pivot entry, which the orthogonal change of basis alters; SURVEY.md section 7
// normalises diag(R) >= 0 when the elimination order differs).  Pivot decisions inside rounding reach the exact path as before.
//
// The algorithm is tall-skinny QR applied panel by panel (CAQR): the rows are cut into CHUNKS of 32, a SLAB is 8 chunks.
//   level 0   every slab factorises its 256 x 32 piece of the panel on its own (one workgroup, Householder, no communication):
//             R_s in the upper triangle of its top chunk, the reflectors below, their T factor to a side buffer;
//   level l   the stack of the R_s of 8 slabs of level l-1 -- 8 triangles, 256 virtual rows -- is factorised the same way.  Column j
//             of a stack of triangles only has entries in rows <= j of every triangle and the reflectors keep that shape, so the
//             level-l reflectors are stored IN the upper triangles they annihilate (the strictly lower parts keep the level-0
//             reflectors), and the top triangle receives the R of the level;
//   ...       until one slab is left: its R is rows 32 p .. 32 p + 31 of R0.
// Nothing but kernel boundaries synchronises: 2 launches per level and panel (factorise, apply to the trailing columns), 63 x 4
// levels for 40 000 x 2 000, against one launch sequence per REFLECTOR in the level-2 path.  The same apply kernel forms Q0^T b / Q0 b.
//
// Block reflector of a slab: Q_s = I - Y T Y^T (T upper triangular, LAPACK larft forward/columnwise = Eigen's
// make_block_householder_triangular_factor); Q_s^T C = C - Y T^T (Y^T C).  On the matrix cores, a wave per 16 columns of C:
//   W  = Y^T C     D[refl][col]   A = Y^T from LDS, B = C straight from memory (4 rows x 16 columns per step)
//   W' = -T^T W    the D registers of W are the B operand of this product, k-step by k-step (no data movement)
//   C += Y W'      as C^T += W'^T Y^T: the D registers of W' are the A operand, B = Y^T from LDS, D = 16 rows x 16 columns of C
//                  with the ROW on lane & 15 -- column-major C is read and written in 128-byte runs.
#include "qrk_device.h"

namespace qrk {
namespace caqr {

constexpr int NB = 32;          // panel width = rows of a chunk
constexpr int FAN = 8;          // chunks per slab
constexpr int SR = NB * FAN;    // virtual rows of a slab
constexpr int LS = NB + 1;      // LDS row stride (doubles): conflict-free by rows and by columns
constexpr size_t PANEL_LDS = (size_t)(SR * LS + 2 * FAN * NB + FAN + 2 * NB + 1) * sizeof(double);            // 72 KB: fits where an apply workgroup (76 KB) was
constexpr size_t APPLY_LDS = (size_t)(SR * LS + NB * LS) * sizeof(double);                                            // 76 KB
// LDS asked for by the apply launch that runs BESIDE the factorisation of the next panel when QRK_CAQR_BESIDE=1: more than half a CU's
// 160 KB, so that one apply workgroup per CU is resident and a panel workgroup (72 KB, 240 VGPRs) finds room at once instead of waiting
// for one of two apply workgroups to finish.  Helped the plain look-ahead (level-0 panel 133 -> 93 us, 49.75 -> 49.4 ms), costs the
// pipelined one, where the caller's stream is on the critical path too (47.4 vs 45.9 ms): off by default
constexpr size_t APPLY_LDS_BESIDE = 86 * 1024;

// Slab t of a level: chunk i of it is chunk  p + stride (FAN t + i)  of the matrix (rows 32 chunk .. 32 chunk + 31).
struct Slab {
    int p, stride, nchunks;     // first active chunk of the panel, chunk stride of the level (FAN^l), chunks of the level
};

__device__ __forceinline__ int64_t chunk_row0(const Slab& s, int t, int i)
{
    return (int64_t)NB * ((int64_t)s.p + (int64_t)s.stride * ((int64_t)FAN * t + i));
}
__device__ __forceinline__ int chunk_rows(int64_t row0, int m)
{
    const int64_t left = (int64_t)m - row0;
    return left <= 0 ? 0 : (left < NB ? (int)left : NB);
}

// sqrt(x) of a positive normal x by v_rsq_f64 + one Goldschmidt iteration + one residual correction (<= 1 ulp), and 1/x by v_rcp_f64 +
// two Newton steps (<= 1 ulp): the dependent chains of the IEEE sqrt / division sequences are three times as long, and with one
// wave per SIMD nothing hides them (same forms as bdqr_pair.hip).
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
    return fma(y, e, y);
}

struct PanelLds {
    double (*vb)[NB];
    double (*red)[NB];
    double (*zz)[LS];
    double* nrm;
    double* prow;
    double* taus;
    double* x0s;
#ifdef QRK_CAQR_STAMP
    unsigned long long* fine;      // diagnostic: s_memtime inside step 16
#endif
};

// Reflector J of the slab (every index into the register arrays is a compile-time constant: a rolled loop leaves them in scratch).
template <int J>
__device__ __forceinline__ void panel_step(double (&a)[NB], double& myinv, const PanelLds& L, int c, int i, int w)
{
#ifdef QRK_CAQR_STAMP
#define QRK_FINE(n) do { if (J == 16 && c == 0 && i == 0) L.fine[n] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define QRK_FINE(n) do { } while (0)
#endif
    QRK_FINE(0);
    __syncthreads();                                         // (A) vb holds column J
    QRK_FINE(1);
    double v[NB];
#pragma unroll
    for (int r2 = 0; r2 < NB; r2 += 2) {
        const double2 t2 = *reinterpret_cast<const double2*>(&L.vb[i][r2]);
        v[r2] = t2.x; v[r2 + 1] = t2.y;
    }
    const double xj = v[J];                                  // pivot entry (meaningful in chunk 0)
#pragma unroll
    for (int r2 = 0; r2 < NB; ++r2) if (r2 <= J) v[r2] = (i == 0) ? 0.0 : v[r2];   // rows above and at the pivot are not part of the tail
    double d0 = 0.0, d1 = 0.0, d2 = 0.0, d3 = 0.0, s0 = 0.0, s1 = 0.0;    // partial sums: no FMA waits for its predecessor
#pragma unroll
    for (int r2 = 0; r2 < NB; r2 += 4) {
        d0 = fma(v[r2], a[r2], d0); d1 = fma(v[r2 + 1], a[r2 + 1], d1); d2 = fma(v[r2 + 2], a[r2 + 2], d2); d3 = fma(v[r2 + 3], a[r2 + 3], d3);
        s0 = fma(v[r2], v[r2], s0); s1 = fma(v[r2 + 1], v[r2 + 1], s1); s0 = fma(v[r2 + 2], v[r2 + 2], s0); s1 = fma(v[r2 + 3], v[r2 + 3], s1);
    }
    L.red[i][c] = (d0 + d1) + (d2 + d3);
    if (c == 0) L.nrm[i] = s0 + s1;
    if (i == 0) { L.prow[c] = a[J]; if (c == 0) *L.x0s = xj; }
    QRK_FINE(2);
    __syncthreads();                                         // (B)
    QRK_FINE(3);
    double rr[FAN], nn[FAN];
#pragma unroll
    for (int ii = 0; ii < FAN; ++ii) { rr[ii] = L.red[ii][c]; nn[ii] = L.nrm[ii]; }
    const double D = ((rr[0] + rr[1]) + (rr[2] + rr[3])) + ((rr[4] + rr[5]) + (rr[6] + rr[7]));
    const double tailSq = ((nn[0] + nn[1]) + (nn[2] + nn[3])) + ((nn[4] + nn[5]) + (nn[6] + nn[7]));
    static_assert((SR * LS) % 2 == 0, "vb is read and written with 16-byte LDS accesses");
    static_assert(FAN == 8, "the sums above are written for 8 chunks");
    const double x0 = *L.x0s, a0c = L.prow[c];
    double tau, beta, inv;
    if (tailSq <= DBL_MIN) { tau = 0.0; beta = x0; inv = 0.0; }
    else {
        beta = fast_sqrt(fma(x0, x0, tailSq));
        if (x0 >= 0.0) beta = -beta;
        const double wv = x0 - beta;                 // |w| >= |beta|: no cancellation (beta and x0 have opposite signs)
        inv = fast_recip(wv);
        tau = -wv * fast_recip(beta);                // (beta - x0) / beta
    }
    const double tmp = fma(inv, D, a0c);                     // row0 + essential^T bottom
    QRK_FINE(4);
    // columns right of J: c_J -= tau tmp, tail -= tau tmp essential -- ONE sweep for every lane, coefficient 0 for the columns up to J
    // (until round 5 three divergent branches, which every wave ran one after the other: each has lanes with c < J, c == J and c > J;
    // profiles/r05_caqr_panel.txt).  Column J: beta on the diagonal now; its essential part x_tail / (x0 - beta) is a scaling of
    // registers that nothing reads again before the store: the thread keeps 1 / (x0 - beta) and scales at the end (same product, same bits).
    const double g = tau * tmp;
    const double coef = c > J ? -(g * inv) : 0.0;
    if (i == 0 && c > J) a[J] -= g;
#pragma unroll
    for (int r2 = 0; r2 < NB; ++r2) a[r2] = fma(coef, v[r2], a[r2]);
    QRK_FINE(5);
    // the next column goes out as soon as it is updated: everything below is off the chain
    if (J + 1 < w && c == J + 1) {
#pragma unroll
        for (int r2 = 0; r2 < NB; r2 += 2) *reinterpret_cast<double2*>(&L.vb[i][r2]) = make_double2(a[r2], a[r2 + 1]);
    }
    if (c == J) {
        myinv = inv;
        if (i == 0) { a[J] = beta; L.taus[J] = tau; }
    } else if (c < J && i == 0) {
        // y_c^T y_J = Y(J, c) + Y(tail, c)^T essential; column c is still x_tail, not yet x_tail / (x0 - beta): the factor goes on the sum
        L.zz[c][J] = tmp * myinv;
    }
    QRK_FINE(6);
}

// ---------------------------------------------------------------------------------------------------------------------------
// Panel factorisation of one slab: Householder QR of 256 (virtual) rows x w <= 32 columns.  Thread (c, i) = (tid & 31, tid >> 5)
// keeps column c of chunk i in 32 registers; the reflector column travels through 2 KB of LDS, the per-column dots are summed
// over the 8 chunks through LDS: two barriers per reflector, no other traffic.  Eigen's makeHouseholder /
// applyHouseholderOnTheLeft (Eigen/src/Householder/Householder.h), real square root and division.
// TRI: the rows are a stack of upper triangles (entries below the diagonal of a chunk are not data and are left alone).
template <bool TRI>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 2)))   // column + reflector in registers: ~150 VGPRs
caqr_panel_kernel(double* __restrict__ A, int64_t lda, int m, int pc, int w, Slab sl, double* __restrict__ Tout)
{
    extern __shared__ __attribute__((aligned(16))) double caqr_lds[];
    double* sm = caqr_lds;                                                        // [SR][LS] transposing buffer
    double (*vb)[NB] = reinterpret_cast<double (*)[NB]>(sm + SR * LS);            // [FAN][NB] reflector column (16-byte aligned: SR * LS is even)
    double (*red)[NB] = reinterpret_cast<double (*)[NB]>(&vb[FAN][0]);            // [FAN][NB] partial dots
    double (*zz)[LS] = reinterpret_cast<double (*)[LS]>(sm + 2 * NB * LS);        // [NB][LS] strictly upper part: y_c^T y_j, c < j.  Lives INSIDE the
                                                                                  // transposing buffer, which is idle between the load and the store (behind
                                                                                  // the two T scratch areas): 72 KB in all, the LDS slot of one apply workgroup
    double* nrm = &red[FAN][0];                                                   // [FAN] partial squared tail norms
    double* prow = nrm + FAN;                                                     // [NB] pivot row
    double* taus = prow + NB;                                                     // [NB]
    double& x0s = taus[NB];

    const int t = blockIdx.x;
    const int tid = threadIdx.x, c = tid & 31, i = tid >> 5;
    int cnt = sl.nchunks - FAN * t;
    cnt = cnt > FAN ? FAN : cnt;
    const int64_t row0 = chunk_row0(sl, t, i);
    const int nr = i < cnt ? chunk_rows(row0, m) : 0;

    // ---- load: lane = row of the chunk (column-major A: 256-byte runs), transposed through LDS to lane = column
    {
        const int x = c;
        const double* src = A + (int64_t)pc * lda + row0 + x;
        double ld[NB];                   // all 32 loads in flight at once
#pragma unroll
        for (int cc = 0; cc < NB; ++cc) ld[cc] = *((cc < w && x < nr && (!TRI || x <= cc)) ? src + (int64_t)cc * lda : A);   // (clamped address, not a load under a condition)
#pragma unroll
        for (int cc = 0; cc < NB; ++cc) sm[(i * NB + x) * LS + cc] = (cc < w && x < nr && (!TRI || x <= cc)) ? ld[cc] : 0.0;
    }
    __syncthreads();
    double a[NB];
#pragma unroll
    for (int r2 = 0; r2 < NB; ++r2) a[r2] = sm[(i * NB + r2) * LS + c];
    __syncthreads();                     // (the buffer has been read: its space now serves zz)
    for (int e = tid; e < NB * LS; e += 256) (&zz[0][0])[e] = 0.0;

    if (c == 0) {
#pragma unroll
        for (int r2 = 0; r2 < NB; ++r2) vb[i][r2] = a[r2];
    }
#ifdef QRK_CAQR_STAMP   // diagnostic only (tools/caqr_bench.hip): s_memtime at phase boundaries, parked in the (zero) strictly lower part of T
#define QRK_CAQR_STAMP_AT(n) do { if (tid == 0) stamps[n] = __builtin_amdgcn_s_memtime(); } while (0)
    __shared__ unsigned long long stamps[8];
#else
#define QRK_CAQR_STAMP_AT(n) do { } while (0)
#endif
    QRK_CAQR_STAMP_AT(0);
#ifdef QRK_CAQR_STAMP
    __shared__ unsigned long long fine[8];
    PanelLds L{vb, red, zz, nrm, prow, taus, &x0s, fine};
#else
    PanelLds L{vb, red, zz, nrm, prow, taus, &x0s};
#endif
    double myinv = 0.0;                  // 1 / (x0 - beta) of this thread's own column, once it has been the reflector
#define QRK_CAQR_STEP(J) if ((J) < w) panel_step<J>(a, myinv, L, c, i, w);
    QRK_CAQR_STEP(0) QRK_CAQR_STEP(1) QRK_CAQR_STEP(2) QRK_CAQR_STEP(3) QRK_CAQR_STEP(4) QRK_CAQR_STEP(5) QRK_CAQR_STEP(6) QRK_CAQR_STEP(7)
    QRK_CAQR_STAMP_AT(1);
    QRK_CAQR_STEP(8) QRK_CAQR_STEP(9) QRK_CAQR_STEP(10) QRK_CAQR_STEP(11) QRK_CAQR_STEP(12) QRK_CAQR_STEP(13) QRK_CAQR_STEP(14) QRK_CAQR_STEP(15)
    QRK_CAQR_STAMP_AT(2);
    QRK_CAQR_STEP(16) QRK_CAQR_STEP(17) QRK_CAQR_STEP(18) QRK_CAQR_STEP(19) QRK_CAQR_STEP(20) QRK_CAQR_STEP(21) QRK_CAQR_STEP(22) QRK_CAQR_STEP(23)
    QRK_CAQR_STAMP_AT(3);
    QRK_CAQR_STEP(24) QRK_CAQR_STEP(25) QRK_CAQR_STEP(26) QRK_CAQR_STEP(27) QRK_CAQR_STEP(28) QRK_CAQR_STEP(29) QRK_CAQR_STEP(30) QRK_CAQR_STEP(31)
#undef QRK_CAQR_STEP
    QRK_CAQR_STAMP_AT(4);
    // the essential parts: column c below its pivot (chunk 0 keeps its rows of R up to the diagonal), scaled by 1 / (x0 - beta)
    if (c < w) {
#pragma unroll
        for (int r2 = 0; r2 < NB; ++r2) a[r2] = (i > 0 || r2 > c) ? a[r2] * myinv : a[r2];
    }
    __syncthreads();
    // ---- T (LAPACK larft forward/columnwise = Eigen make_block_householder_triangular_factor): T(l,l) = tau_l,
    // T(0:l,l) = -tau_l T(0:l,0:l) U(0:l,l) with U = strictly upper part of Y^T Y (zz).  In the recursive form: the four 8 x 8 diagonal
    // blocks by the column recurrence (a thread per row, the blocks side by side), then T01 = -T00 (U01 T11), T23 likewise, then the
    // 16 x 16 corner T[01][23] = -T[01] (U[01][23] T[23]): 55 dependent FMAs instead of 496 (the serial form took 15 us of the kernel's 80).
    {
        double* tm = sm;                 // [NB][LS] T            (sm is free until the store below)
        double* mm = sm + NB * LS;       // [NB][LS] U * T of the merge in flight
        for (int e = tid; e < NB * LS; e += 256) tm[e] = 0.0;
        __syncthreads();
        if (tid < NB) {
            const int ar = tid, l0 = ar & ~7;
            for (int l = l0; l < l0 + 8; ++l) {
                double tv = 0.0;
                if (l < w) {
                    const double tau = taus[l];
                    if (ar == l) tv = tau;
                    else if (ar < l) {
                        double acc = 0.0;
                        for (int bb = ar; bb < l; ++bb) acc = fma(tm[ar * LS + bb], zz[bb][l], acc);
                        tv = -tau * acc;
                    }
                }
                tm[ar * LS + l] = tv;    // (row ar only depends on row ar)
            }
        }
        __syncthreads();
        // merges: rows [r0, r0 + h) x columns [r0 + h, r0 + 2 h)
#pragma unroll
        for (int h = 8; h <= 16; h *= 2) {
            const int per = h * h, nmerge = NB / (2 * h);
            if (tid < per * nmerge) {
                const int mg = tid / per, e = tid - mg * per, ii = e / h, jj = e - ii * h;
                const int r0 = mg * 2 * h, cb = r0 + h;
                double acc = 0.0;                                    // (U12 T22)(ii, jj)
                for (int q = 0; q <= jj; ++q) acc = fma(zz[r0 + ii][cb + q], tm[(cb + q) * LS + cb + jj], acc);
                mm[(r0 + ii) * LS + cb + jj] = acc;
            }
            __syncthreads();
            if (tid < per * nmerge) {
                const int mg = tid / per, e = tid - mg * per, ii = e / h, jj = e - ii * h;
                const int r0 = mg * 2 * h, cb = r0 + h;
                double acc = 0.0;                                    // -(T11 (U12 T22))(ii, jj)
                for (int q = ii; q < h; ++q) acc = fma(tm[(r0 + ii) * LS + r0 + q], mm[(r0 + q) * LS + cb + jj], acc);
                tm[(r0 + ii) * LS + cb + jj] = -acc;
            }
            __syncthreads();
        }
        double* dst = Tout + (int64_t)t * (NB * NB);
        for (int e = tid; e < NB * NB; e += 256) dst[e] = tm[(e >> 5) * LS + (e & 31)];
        __syncthreads();
    }
    QRK_CAQR_STAMP_AT(5);
    // ---- store: back through LDS to lane = row
#pragma unroll
    for (int r2 = 0; r2 < NB; ++r2) sm[(i * NB + r2) * LS + c] = a[r2];
    __syncthreads();
    {
        const int x = c;
        double* dst = A + (int64_t)pc * lda + row0 + x;
#pragma unroll 8
        for (int cc = 0; cc < NB; ++cc)
            if (cc < w && x < nr && (!TRI || x <= cc)) dst[(int64_t)cc * lda] = sm[(i * NB + x) * LS + cc];
    }
#ifdef QRK_CAQR_STAMP
    __syncthreads();
    if (tid == 0) {
        stamps[6] = __builtin_amdgcn_s_memtime();
        for (int q = 0; q < 7; ++q) reinterpret_cast<unsigned long long*>(Tout + (int64_t)t * (NB * NB))[NB * (q + 1)] = stamps[q];
        for (int q = 0; q < 7; ++q) reinterpret_cast<unsigned long long*>(Tout + (int64_t)t * (NB * NB))[NB * (q + 9)] = fine[q];
    }
#endif
}

// ---------------------------------------------------------------------------------------------------------------------------
// C <- Q_s^T C (transpose != 0) or Q_s C for the block reflector of every slab of a level, C column-major with the row indexing
// of A.  Grid (slabs, column groups); a wave takes 16 columns at a time.
typedef double d4 __attribute__((ext_vector_type(4)));

constexpr int APPLY_T = 512;     // 8 waves share the Y of a slab: two workgroups per CU (LDS) = 4 waves per SIMD behind the loads of C
template <bool TRI>
__global__ void __launch_bounds__(APPLY_T)
caqr_apply_kernel(const double* __restrict__ A, int64_t lda, int m, int pc, int w, Slab sl, const double* __restrict__ Tin,
                  int transpose, double* __restrict__ C, int64_t ldc, int ncols, int cols_per_wg)
{
    extern __shared__ __attribute__((aligned(16))) double caqr_lds[];
    double* ys = caqr_lds;              // [SR][LS]
    double* ts = ys + SR * LS;          // [NB][LS]
    const int t = blockIdx.x;
    const int tid = threadIdx.x;
    int cnt = sl.nchunks - FAN * t;
    cnt = cnt > FAN ? FAN : cnt;
    // ---- Y of the slab as a dense 256 x 32 matrix (unit diagonal and structural zeros written out)
    {
        const int x = tid & 31, y = (tid >> 5) & (FAN - 1), ch = tid >> 8;      // row of the chunk, chunk, half of the columns
        const int64_t row0 = chunk_row0(sl, t, y);
        const int nr = y < cnt ? chunk_rows(row0, m) : 0;
        const double* src = A + (int64_t)pc * lda + row0 + x;
        // (every load is issued before the first is waited for: the address of an entry that is not read from memory is clamped
        //  to a valid one and the value replaced afterwards -- loads under a condition each got their own wait, 16 round trips
        //  to L2 before the first MFMA of the workgroup)
        double raw[NB / 2];
#pragma unroll
        for (int q = 0; q < NB / 2; ++q) {
            const int cc = ch * (NB / 2) + q;
            const bool need = cc < w && x < nr && (y == 0 ? (!TRI && x > cc) : (!TRI || x <= cc));
            raw[q] = *(need ? src + (int64_t)cc * lda : A);
        }
#pragma unroll
        for (int q = 0; q < NB / 2; ++q) {
            const int cc = ch * (NB / 2) + q;
            const bool need = cc < w && x < nr && (y == 0 ? (!TRI && x > cc) : (!TRI || x <= cc));
            const double v = need ? raw[q] : ((cc < w && x < nr && y == 0 && x == cc) ? 1.0 : 0.0);
            ys[(y * NB + x) * LS + cc] = v;
        }
        for (int e = tid; e < NB * NB; e += APPLY_T) ts[(e >> 5) * LS + (e & 31)] = Tin[(int64_t)t * (NB * NB) + e];
    }
    __syncthreads();
    const int wave = tid >> 6, lane = tid & 63, kq = lane >> 4, l15 = lane & 15;
    const int col_base = blockIdx.y * cols_per_wg;
    const int ntile = (cols_per_wg + 15) >> 4;
    for (int tile = wave; tile < ntile; tile += APPLY_T / 64) {
        const int n0 = col_base + 16 * tile;
        if (n0 >= ncols) break;
        // ---- W = Y^T C  (the loads of the next chunk are in flight while the MFMAs of this one run).  A lane takes FOUR consecutive
        // rows of its column (two 16-byte loads; the 16 lanes of a k-group cover 16 rows x 16 columns in whole 128-byte lines) and
        // feeds them to four k-steps: slot k of a chunk is row 16 (k >> 2) + 4 kq + (k & 3) -- the order inside a k-step is free as
        // long as the Y operand is read with the same rows.
        d4 acc0 = d4{0.0, 0.0, 0.0, 0.0}, acc1 = d4{0.0, 0.0, 0.0, 0.0};
        {
            const int col = n0 + l15;
            const bool cok = col < ncols;
            const double* cb = C + (int64_t)col * ldc;
            auto load_chunk = [&](int ci, double (&bb)[8]) {
                const int64_t row0 = chunk_row0(sl, t, ci);
                const int nr = chunk_rows(row0, m);
#pragma unroll
                for (int blk = 0; blk < 2; ++blk) {
                    const int base = 16 * blk + 4 * kq;
                    const double* p = cb + row0 + base;
                    if (cok && base + 3 < nr && (reinterpret_cast<uintptr_t>(p) & 15) == 0) {
                        const double2 lo = *reinterpret_cast<const double2*>(p), hi = *reinterpret_cast<const double2*>(p + 2);
                        bb[4 * blk] = lo.x; bb[4 * blk + 1] = lo.y; bb[4 * blk + 2] = hi.x; bb[4 * blk + 3] = hi.y;
                    } else {
#pragma unroll
                        for (int jj = 0; jj < 4; ++jj) bb[4 * blk + jj] = (cok && base + jj < nr) ? p[jj] : 0.0;
                    }
                }
            };
            double bn[8];
            load_chunk(0, bn);
            for (int ci = 0; ci < cnt; ++ci) {
                double bv[8];
#pragma unroll
                for (int k = 0; k < 8; ++k) bv[k] = bn[k];
                if (ci + 1 < cnt) load_chunk(ci + 1, bn);
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const int vrow = ci * NB + 16 * (k >> 2) + 4 * kq + (k & 3);
                    acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(ys[vrow * LS + l15], bv[k], acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(ys[vrow * LS + 16 + l15], bv[k], acc1, 0, 0, 0);
                }
            }
        }
        // ---- W' = -(T^T or T) W
        d4 u0 = d4{0.0, 0.0, 0.0, 0.0}, u1 = d4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) {
            const double bw = ks < 4 ? acc0[ks & 3] : acc1[ks & 3];
            const int kk = 4 * ks + kq;
            const double t0 = transpose ? ts[kk * LS + l15] : ts[l15 * LS + kk];
            const double t1 = transpose ? ts[kk * LS + 16 + l15] : ts[(16 + l15) * LS + kk];
            u0 = __builtin_amdgcn_mfma_f64_16x16x4f64(t0, bw, u0, 0, 0, 0);
            u1 = __builtin_amdgcn_mfma_f64_16x16x4f64(t1, bw, u1, 0, 0, 0);
        }
        u0 = -u0; u1 = -u1;
        // ---- C += Y W'  (as C^T += W'^T Y^T: D[column kq + 4 z][row l15]); the tile after the current one is being loaded
        // while this one goes through the matrix cores
        {
            const int ntl = 2 * cnt;                 // 16-row tiles of the slab
            auto load_tile = [&](int tl, d4& dv) {
                const int ci = tl >> 1, rr = 16 * (tl & 1) + l15;
                const int64_t row0 = chunk_row0(sl, t, ci);
                const bool rok = rr < chunk_rows(row0, m);
                const double* cp = C + row0 + rr;
#pragma unroll
                for (int z = 0; z < 4; ++z) {
                    const int colz = n0 + kq + 4 * z;
                    dv[z] = (rok && colz < ncols) ? cp[(int64_t)colz * ldc] : 0.0;
                }
            };
            d4 dn;
            load_tile(0, dn);
            for (int tl = 0; tl < ntl; ++tl) {
                d4 dv = dn;
                if (tl + 1 < ntl) load_tile(tl + 1, dn);
                const int ci = tl >> 1, rr = 16 * (tl & 1) + l15;
                const int64_t row0 = chunk_row0(sl, t, ci);
                const bool rok = rr < chunk_rows(row0, m);
                const int vrow = ci * NB + rr;
#pragma unroll
                for (int ks = 0; ks < 8; ++ks) {
                    const double aw = ks < 4 ? u0[ks & 3] : u1[ks & 3];
                    dv = __builtin_amdgcn_mfma_f64_16x16x4f64(aw, ys[vrow * LS + 4 * ks + kq], dv, 0, 0, 0);
                }
                double* cp = C + row0 + rr;
#pragma unroll
                for (int z = 0; z < 4; ++z) {
                    const int colz = n0 + kq + 4 * z;
                    if (rok && colz < ncols) cp[(int64_t)colz * ldc] = dv[z];
                }
            }
        }
    }
}

// The same for a NARROW C (5 ... 32 columns: the columns of the next panel, which the look-ahead needs before anything else -- this launch
// sits on the loop-carried chain of the first stage, four of them per panel).  The kernel above gives a 16-column tile to ONE wave, which
// then walks the 8 chunks of the slab with one load in flight: 25 us per launch whatever the level.  Here wave w takes chunk w: the
// partial W of the chunks go through LDS and are added in a fixed order by every wave, and each wave updates the rows of its own chunk.
constexpr size_t NARROW_LDS = APPLY_LDS + (size_t)FAN * 2 * 8 * 64 * sizeof(double);       // + 64 KB of partial sums
template <bool TRI>
__global__ void __launch_bounds__(APPLY_T)
caqr_apply_narrow_kernel(const double* __restrict__ A, int64_t lda, int m, int pc, int w, Slab sl, const double* __restrict__ Tin,
                         int transpose, double* __restrict__ C, int64_t ldc, int ncols)
{
    extern __shared__ __attribute__((aligned(16))) double caqr_lds[];
    double* ys = caqr_lds;              // [SR][LS]
    double* ts = ys + SR * LS;          // [NB][LS]
    double* part = ts + NB * LS;        // [FAN][2][8][64]
    const int t = blockIdx.x;
    const int tid = threadIdx.x;
    int cnt = sl.nchunks - FAN * t;
    cnt = cnt > FAN ? FAN : cnt;
    {
        const int x = tid & 31, y = (tid >> 5) & (FAN - 1), ch = tid >> 8;
        const int64_t row0 = chunk_row0(sl, t, y);
        const int nr = y < cnt ? chunk_rows(row0, m) : 0;
        const double* src = A + (int64_t)pc * lda + row0 + x;
        double raw[NB / 2];
#pragma unroll
        for (int q = 0; q < NB / 2; ++q) {
            const int cc = ch * (NB / 2) + q;
            const bool need = cc < w && x < nr && (y == 0 ? (!TRI && x > cc) : (!TRI || x <= cc));
            raw[q] = *(need ? src + (int64_t)cc * lda : A);
        }
#pragma unroll
        for (int q = 0; q < NB / 2; ++q) {
            const int cc = ch * (NB / 2) + q;
            const bool need = cc < w && x < nr && (y == 0 ? (!TRI && x > cc) : (!TRI || x <= cc));
            const double v = need ? raw[q] : ((cc < w && x < nr && y == 0 && x == cc) ? 1.0 : 0.0);
            ys[(y * NB + x) * LS + cc] = v;
        }
        for (int e = tid; e < NB * NB; e += APPLY_T) ts[(e >> 5) * LS + (e & 31)] = Tin[(int64_t)t * (NB * NB) + e];
    }
    const int wave = tid >> 6, lane = tid & 63, kq = lane >> 4, l15 = lane & 15;
    const int ci = wave;                                  // this wave's chunk
    const bool active = ci < cnt;
    const int ntile = (ncols + 15) >> 4;                  // 1 or 2
    const int64_t crow0 = chunk_row0(sl, t, active ? ci : 0);
    const int cnr = active ? chunk_rows(crow0, m) : 0;
    // ---- the C entries of this chunk, for W (B-operand order: slot k = row 16 (k >> 2) + 4 kq + (k & 3)): in flight during the staging
    double bv[2][8];
#pragma unroll
    for (int tile = 0; tile < 2; ++tile) {
        const int col = 16 * tile + l15;
        const bool cok = tile < ntile && col < ncols;
        const double* cb = C + (int64_t)(cok ? col : 0) * ldc + crow0;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int rr = 16 * (k >> 2) + 4 * kq + (k & 3);
            const bool in = cok && rr < cnr;
            const double v = *(in ? cb + rr : C);
            bv[tile][k] = in ? v : 0.0;
        }
    }
    __syncthreads();
    // ---- partial W = Y_chunk^T C_chunk
#pragma unroll
    for (int tile = 0; tile < 2; ++tile) {
        d4 acc0 = d4{0.0, 0.0, 0.0, 0.0}, acc1 = d4{0.0, 0.0, 0.0, 0.0};
        if (active && tile < ntile) {
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const int vrow = ci * NB + 16 * (k >> 2) + 4 * kq + (k & 3);
                acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(ys[vrow * LS + l15], bv[tile][k], acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(ys[vrow * LS + 16 + l15], bv[tile][k], acc1, 0, 0, 0);
            }
        }
#pragma unroll
        for (int z = 0; z < 4; ++z) {
            part[((wave * 2 + tile) * 8 + z) * 64 + lane] = acc0[z];
            part[((wave * 2 + tile) * 8 + 4 + z) * 64 + lane] = acc1[z];
        }
    }
    __syncthreads();
    // ---- W (the chunks added in a fixed order), W' = -(T^T or T) W: by every wave for itself
    d4 u0[2], u1[2];
#pragma unroll
    for (int tile = 0; tile < 2; ++tile) {
        d4 acc0 = d4{0.0, 0.0, 0.0, 0.0}, acc1 = d4{0.0, 0.0, 0.0, 0.0};
        for (int cw = 0; cw < cnt; ++cw) {
#pragma unroll
            for (int z = 0; z < 4; ++z) {
                acc0[z] += part[((cw * 2 + tile) * 8 + z) * 64 + lane];
                acc1[z] += part[((cw * 2 + tile) * 8 + 4 + z) * 64 + lane];
            }
        }
        d4 a0 = d4{0.0, 0.0, 0.0, 0.0}, a1 = d4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) {
            const double bw = ks < 4 ? acc0[ks & 3] : acc1[ks & 3];
            const int kk = 4 * ks + kq;
            const double t0 = transpose ? ts[kk * LS + l15] : ts[l15 * LS + kk];
            const double t1 = transpose ? ts[kk * LS + 16 + l15] : ts[(16 + l15) * LS + kk];
            a0 = __builtin_amdgcn_mfma_f64_16x16x4f64(t0, bw, a0, 0, 0, 0);
            a1 = __builtin_amdgcn_mfma_f64_16x16x4f64(t1, bw, a1, 0, 0, 0);
        }
        u0[tile] = -a0; u1[tile] = -a1;
    }
    // ---- C_chunk += Y_chunk W' (as C^T += W'^T Y^T: D[column kq + 4 z][row l15]), the two 16-row tiles of this wave's chunk
    if (active) {
#pragma unroll
        for (int tile = 0; tile < 2; ++tile) {
            if (tile >= ntile) break;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int rr = 16 * h + l15;
                const bool rok = rr < cnr;
                double* cp = C + crow0 + rr;
                d4 dv;
#pragma unroll
                for (int z = 0; z < 4; ++z) {
                    const int colz = 16 * tile + kq + 4 * z;
                    const bool in = rok && colz < ncols;
                    const double v = *(in ? cp + (int64_t)colz * ldc : C);
                    dv[z] = in ? v : 0.0;
                }
                const int vrow = ci * NB + rr;
#pragma unroll
                for (int ks = 0; ks < 8; ++ks) {
                    const double aw = ks < 4 ? u0[tile][ks & 3] : u1[tile][ks & 3];
                    dv = __builtin_amdgcn_mfma_f64_16x16x4f64(aw, ys[vrow * LS + 4 * ks + kq], dv, 0, 0, 0);
                }
#pragma unroll
                for (int z = 0; z < 4; ++z) {
                    const int colz = 16 * tile + kq + 4 * z;
                    if (rok && colz < ncols) cp[(int64_t)colz * ldc] = dv[z];
                }
            }
        }
    }
}

// The same block reflectors applied to a few vectors (the right-hand side of a solve: C has 1 ... VEC_MAX columns).  The MFMA kernel
// above stages Y densely in LDS for tiles of 16 columns and takes 25-30 us per launch whatever the width; here thread (x, y) of the
// 256 keeps row x of chunk y of Y in 32 registers and one entry of each vector: w = Y^T c through LDS (two passes of 16 columns,
// fixed summation order), w' = -(T^T or T) w by 32 threads, c += Y w' from the registers.  252 launches of a solve with the
// 40 000 x 2 000 block: 6.7 -> 2.5 ms.
constexpr int VEC_MAX = 4;
template <bool TRI>
__global__ void __launch_bounds__(256)
caqr_apply_vec_kernel(const double* __restrict__ A, int64_t lda, int m, int pc, int w, Slab sl, const double* __restrict__ Tin,
                      int transpose, double* __restrict__ C, int64_t ldc, int ncols)
{
    __shared__ double prod[256 * 17];
    __shared__ double part[8 * 16];
    __shared__ double wv[NB], wp[NB];
    const int t = blockIdx.x, tid = threadIdx.x;
    int cnt = sl.nchunks - FAN * t;
    cnt = cnt > FAN ? FAN : cnt;
    const int x = tid & 31, y = tid >> 5;
    const int64_t row0 = chunk_row0(sl, t, y);
    const int nr = y < cnt ? chunk_rows(row0, m) : 0;
    const bool rok = x < nr;
    double yr[NB];
    {
        const double* src = A + (int64_t)pc * lda + row0 + x;
#pragma unroll
        for (int cc = 0; cc < NB; ++cc) {          // (all loads in flight at once: see caqr_apply_kernel)
            const bool need = cc < w && rok && (y == 0 ? (!TRI && x > cc) : (!TRI || x <= cc));
            yr[cc] = *(need ? src + (int64_t)cc * lda : A);
        }
#pragma unroll
        for (int cc = 0; cc < NB; ++cc) {
            const bool need = cc < w && rok && (y == 0 ? (!TRI && x > cc) : (!TRI || x <= cc));
            yr[cc] = need ? yr[cc] : ((cc < w && rok && y == 0 && x == cc) ? 1.0 : 0.0);
        }
    }
    const double* Tt = Tin + (int64_t)t * (NB * NB);
    // the row / column of T this thread multiplies with: fetched with the loads of Y, not after w is known (the kernel is a chain of
    // dependent steps of 1-2 us each, 252 of them per solve(): every one taken off it counts; round 5)
    double trow[NB];
    if (tid < NB) {
#pragma unroll
        for (int k = 0; k < NB; ++k) trow[k] = transpose ? Tt[k * NB + tid] : Tt[tid * NB + k];
    }
    double cfirst = 0.0;
    if (ncols > 0 && rok) cfirst = C[row0 + x];
    for (int v = 0; v < ncols; ++v) {
        double* cp = C + (int64_t)v * ldc + row0 + x;
        double c = v == 0 ? cfirst : (rok ? *cp : 0.0);
        // w = Y^T c
#pragma unroll
        for (int h = 0; h < 2; ++h) {
#pragma unroll
            for (int j = 0; j < 16; ++j) prod[tid * 17 + j] = yr[16 * h + j] * c;
            __syncthreads();
            if (tid < 128) {
                const int j = tid & 15, pp = tid >> 4;
                double sacc = 0.0;
                for (int q = 0; q < 32; ++q) sacc += prod[(pp * 32 + q) * 17 + j];
                part[pp * 16 + j] = sacc;
            }
            __syncthreads();
            if (tid < 16) {
                double sacc = 0.0;
#pragma unroll
                for (int pp = 0; pp < 8; ++pp) sacc += part[pp * 16 + tid];
                wv[16 * h + tid] = sacc;
            }
            __syncthreads();
        }
        // w' = -(T^T or T) w
        if (tid < NB) {
            double sacc = 0.0;
#pragma unroll
            for (int k = 0; k < NB; ++k) sacc = fma(trow[k], wv[k], sacc);
            wp[tid] = -sacc;
        }
        __syncthreads();
        // c += Y w'
#pragma unroll
        for (int j = 0; j < NB; ++j) c = fma(yr[j], wp[j], c);
        if (rok) *cp = c;
        __syncthreads();
    }
}

// R0 (upper triangle of the first n rows of A) as a dense n x n column-major matrix with zeros below the diagonal, and back
// (the final R of the pivoted second stage replaces R0 in the caller's array).
__global__ void caqr_copy_upper_kernel(const double* __restrict__ src, int64_t lds_, double* __restrict__ dst, int64_t ldd, int n,
                                       int zero_lower)
{
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= (int64_t)n * n) return;
    const int col = (int)(e / n), row = (int)(e - (int64_t)col * n);
    if (row <= col) dst[(int64_t)col * ldd + row] = src[(int64_t)col * lds_ + row];
    else if (zero_lower) dst[(int64_t)col * ldd + row] = 0.0;
}

}  // namespace caqr

// ---- host side ------------------------------------------------------------------------------------------------------------
namespace {
struct CaqrShape { int NC, NP, LMAX, S0; };
CaqrShape caqr_shape(int m, int n)
{
    CaqrShape s;
    s.NC = (m + caqr::NB - 1) / caqr::NB;
    s.NP = (n + caqr::NB - 1) / caqr::NB;
    s.S0 = (s.NC + caqr::FAN - 1) / caqr::FAN;
    s.LMAX = 1;
    for (int k = s.S0; k > 1; k = (k + caqr::FAN - 1) / caqr::FAN) ++s.LMAX;
    return s;
}
}  // namespace

size_t caqr_t_bytes(int m, int n)
{
    const CaqrShape s = caqr_shape(m, n);
    return (size_t)s.NP * s.LMAX * s.S0 * caqr::NB * caqr::NB * sizeof(double);
}

// One panel: factorise level by level, then (ncols > 0) apply the block reflectors of every level to C.
static hipError_t caqr_panel_levels(double* A, int64_t lda, int m, int p, int w, double* Tbuf, const CaqrShape& s, bool factorize,
                                    int transpose, double* C, int64_t ldc, int ncols, bool reverse, hipStream_t stream,
                                    size_t apply_lds = caqr::APPLY_LDS, int only_level = -1)
{
    using namespace caqr;
    {
        // dynamic LDS above 64 KB has to be asked for, once per kernel
        static hipError_t attr = [] {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(caqr_panel_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)PANEL_LDS);
            if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(caqr_panel_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)PANEL_LDS);
            if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(caqr_apply_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)APPLY_LDS_BESIDE);
            if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(caqr_apply_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)APPLY_LDS_BESIDE);
            if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(caqr_apply_narrow_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)NARROW_LDS);
            if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(caqr_apply_narrow_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)NARROW_LDS);
            return e;
        }();
        if (attr != hipSuccess) return attr;
    }
    const int pc = p * NB;
    int nlev = 0, Ks[16], strides[16];
    for (int K = s.NC - p, stride = 1;; ) {
        Ks[nlev] = K; strides[nlev] = stride; ++nlev;
        const int S = (K + FAN - 1) / FAN;
        if (S <= 1) break;
        K = S; stride *= FAN;
    }
    auto tptr = [&](int l) { return Tbuf + ((size_t)p * s.LMAX + l) * (size_t)s.S0 * (NB * NB); };
    if (factorize) {
        for (int l = 0; l < nlev; ++l) {
            if (only_level >= 0 && l != only_level) continue;
            const Slab sl{p, strides[l], Ks[l]};
            const int S = (Ks[l] + FAN - 1) / FAN;
            if (l == 0) hipLaunchKernelGGL((caqr_panel_kernel<false>), dim3(S), dim3(256), PANEL_LDS, stream, A, lda, m, pc, w, sl, tptr(l));
            else hipLaunchKernelGGL((caqr_panel_kernel<true>), dim3(S), dim3(256), PANEL_LDS, stream, A, lda, m, pc, w, sl, tptr(l));
        }
    }
    if (ncols > 0) {
        for (int li = 0; li < nlev; ++li) {
            const int l = reverse ? nlev - 1 - li : li;
            if (only_level >= 0 && l != only_level) continue;
            const Slab sl{p, strides[l], Ks[l]};
            const int S = (Ks[l] + FAN - 1) / FAN;
            // columns per workgroup: up to 128 (the Y of the slab is loaded once per workgroup), fewer when the level has few slabs,
            // so that the grid still covers the chip (>= ~512 workgroups)
            int cg_want = 512 / S; if (cg_want < 1) cg_want = 1;
            int cols_per_wg = ((ncols + cg_want - 1) / cg_want + 15) / 16 * 16;
            if (cols_per_wg > 128) cols_per_wg = 128;
            if (cols_per_wg < 16) cols_per_wg = 16;
            const int cg = (ncols + cols_per_wg - 1) / cols_per_wg;
            if (ncols <= VEC_MAX && !std::getenv("QRK_CAQR_NO_VEC")) {        // a few right-hand sides: the register kernel
                if (l == 0) hipLaunchKernelGGL((caqr_apply_vec_kernel<false>), dim3(S), dim3(256), 0, stream, A, lda, m, pc, w, sl, tptr(l),
                                               transpose, C, ldc, ncols);
                else hipLaunchKernelGGL((caqr_apply_vec_kernel<true>), dim3(S), dim3(256), 0, stream, A, lda, m, pc, w, sl, tptr(l),
                                        transpose, C, ldc, ncols);
                continue;
            }
            if (ncols <= 32 && !std::getenv("QRK_CAQR_NO_NARROW")) {       // the columns of the next panel: a wave per chunk
                if (l == 0) hipLaunchKernelGGL((caqr_apply_narrow_kernel<false>), dim3(S), dim3(APPLY_T), NARROW_LDS, stream, A, lda, m, pc, w,
                                               sl, tptr(l), transpose, C, ldc, ncols);
                else hipLaunchKernelGGL((caqr_apply_narrow_kernel<true>), dim3(S), dim3(APPLY_T), NARROW_LDS, stream, A, lda, m, pc, w, sl,
                                        tptr(l), transpose, C, ldc, ncols);
                continue;
            }
            if (l == 0) hipLaunchKernelGGL((caqr_apply_kernel<false>), dim3(S, cg), dim3(APPLY_T), apply_lds, stream, A, lda, m, pc, w, sl,
                                           tptr(l), transpose, C, ldc, ncols, cols_per_wg);
            else hipLaunchKernelGGL((caqr_apply_kernel<true>), dim3(S, cg), dim3(APPLY_T), apply_lds, stream, A, lda, m, pc, w, sl,
                                    tptr(l), transpose, C, ldc, ncols, cols_per_wg);
        }
    }
    return hipGetLastError();
}

// A (m x n, m >= n, column-major) <- R0 in the upper triangle, the reflectors of all levels below / inside the triangles; Tbuf
// (caqr_t_bytes) receives the T factors.
// With a side stream and two events (look-ahead): the reflectors of panel p are applied to the columns of panel p + 1 first; panel
// p + 1 is then factorised on the side stream (a few hundred workgroups at most, one wave per SIMD: latency-bound) WHILE the caller's
// stream applies panel p to the rest of the trailing matrix (disjoint columns, the Y / T of panel p only read).
// Levels of the reduction tree of panel p
static int caqr_num_levels(const CaqrShape& s, int p)
{
    int nlev = 0;
    for (int K = s.NC - p;;) {
        ++nlev;
        const int S = (K + caqr::FAN - 1) / caqr::FAN;
        if (S <= 1) break;
        K = S;
    }
    return nlev;
}

// The look-ahead pipelined by LEVELS (three streams).  The loop-carried chain of the plain look-ahead is: apply panel p to the columns
// of panel p + 1 (all levels) -> factorise panel p + 1 (all levels).  But level l of the apply only needs level l of the
// factorisation, so here:
//   side   : factorise panel q level by level, an event after each level but the last; then apply the LAST level to the columns of
//            panel q + 1 itself, so that the factorisation of panel q + 1 follows on the same stream;
//   urgent : as each earlier level of panel q is done, apply it to the columns of panel q + 1 -- beside the factorisation of the
//            next level (those columns must have seen panel q - 1 first: the caller's stream applies every panel to the columns
//            of the panel after the next FIRST and records an event);
//   stream : apply panel q - 1 to the columns of panel q + 1, then to the rest.
// The chain becomes: factorise the levels of q -> apply the last level to the columns of q + 1 -> factorise q + 1, all on one stream.
static hipError_t caqr_factorize_pipelined(double* A, int64_t lda, int m, int n, double* Tbuf, const CaqrShape& s, hipStream_t M,
                                           hipStream_t S, const CaqrPipe& pp, hipEvent_t ev_urgent)
{
    using namespace caqr;
    auto width = [&](int p) { const int pc = p * NB; return p < s.NP ? (n - pc < NB ? n - pc : NB) : 0; };
    auto col = [&](int p) { return A + (int64_t)p * NB * lda; };
    hipError_t e;
    static const bool beside = std::getenv("QRK_CAQR_BESIDE") && std::getenv("QRK_CAQR_BESIDE")[0] == '1';   // (off: see APPLY_LDS_BESIDE)
#define QRK_E(x) do { if ((e = (x)) != hipSuccess) return e; } while (0)
    // panel 0 on the caller's stream, and its reflectors on the columns of panel 1
    QRK_E(caqr_panel_levels(A, lda, m, 0, width(0), Tbuf, s, true, 1, nullptr, lda, 0, false, M));
    if (width(1) > 0) QRK_E(caqr_panel_levels(A, lda, m, 0, width(0), Tbuf, s, false, 1, col(1), lda, width(1), false, M));
    QRK_E(hipEventRecord(ev_urgent, M));
    for (int p = 0; p < s.NP; ++p) {
        const int w = width(p), w1 = width(p + 1), w2 = width(p + 2);
        const int nlev1 = w1 > 0 ? caqr_num_levels(s, p + 1) : 0;
        // (1) panel p is complete (p >= 1: its last level was recorded on the side stream in the previous iteration)
        if (p > 0) QRK_E(hipStreamWaitEvent(M, pp.ev_lvl[caqr_num_levels(s, p) - 1], 0));
        // (3) caller's stream: panel p on the columns of panel p + 2, first
        if (w2 > 0) QRK_E(caqr_panel_levels(A, lda, m, p, w, Tbuf, s, false, 1, col(p + 2), lda, w2, false, M));
        QRK_E(hipEventRecord(pp.ev_n2, M));
        if (w1 > 0) {
            // (2a) side: panel p + 1, level by level (its columns have seen every level of panel p: the previous iteration of this
            // stream, or the prologue); an event after every level but the last
            if (p == 0) QRK_E(hipStreamWaitEvent(S, ev_urgent, 0));
            for (int l = 0; l < nlev1; ++l) {
                QRK_E(caqr_panel_levels(A, lda, m, p + 1, w1, Tbuf, s, true, 1, nullptr, lda, 0, false, S, APPLY_LDS, l));
                if (l + 1 < nlev1) QRK_E(hipEventRecord(pp.ev_lvl[l], S));
            }
            if (w2 > 0) {
                // (4) urgent: every level but the last on the columns of panel p + 2, as soon as it exists (beside the next level)
                if (nlev1 > 1) {
                    QRK_E(hipStreamWaitEvent(pp.urgent, pp.ev_n2, 0));
                    for (int l = 0; l + 1 < nlev1; ++l) {
                        QRK_E(hipStreamWaitEvent(pp.urgent, pp.ev_lvl[l], 0));
                        QRK_E(caqr_panel_levels(A, lda, m, p + 1, w1, Tbuf, s, false, 1, col(p + 2), lda, w2, false, pp.urgent, APPLY_LDS, l));
                    }
                    QRK_E(hipEventRecord(pp.ev_u, pp.urgent));
                    QRK_E(hipStreamWaitEvent(S, pp.ev_u, 0));
                }
                // (2b) the last level on the side stream itself: the next panel's factorisation follows it without a hop between
                // streams (a hop costs 10-20 us, and this is the loop-carried chain)
                QRK_E(hipStreamWaitEvent(S, pp.ev_n2, 0));
                QRK_E(caqr_panel_levels(A, lda, m, p + 1, w1, Tbuf, s, false, 1, col(p + 2), lda, w2, false, S, APPLY_LDS, nlev1 - 1));
            }
            QRK_E(hipEventRecord(pp.ev_lvl[nlev1 - 1], S));       // panel p + 1 is complete (and applied to the next one)
        }
        // (5) caller's stream: panel p on the rest
        const int c3 = (p + 3) * NB, nrest = n - c3;
        if (nrest > 0)
            QRK_E(caqr_panel_levels(A, lda, m, p, w, Tbuf, s, false, 1, A + (int64_t)c3 * lda, lda, nrest, false, M,
                                    w1 > 0 && beside ? APPLY_LDS_BESIDE : APPLY_LDS));
    }
    // the caller's stream ends after everything: the last panel's levels (side) and the last urgent applies
    if (s.NP > 1) QRK_E(hipStreamWaitEvent(M, pp.ev_lvl[caqr_num_levels(s, s.NP - 1) - 1], 0));
#undef QRK_E
    return hipSuccess;
}

hipError_t launch_caqr_factorize(double* A, int64_t lda, int m, int n, double* Tbuf, hipStream_t stream, hipStream_t side,
                                 hipEvent_t ev_urgent, hipEvent_t ev_factored, const CaqrPipe* pipe)
{
    const CaqrShape s = caqr_shape(m, n);
    auto width = [&](int p) { const int pc = p * caqr::NB; return n - pc < caqr::NB ? n - pc : caqr::NB; };
    hipError_t e;
    if (side && ev_urgent && pipe && pipe->urgent && s.LMAX <= CaqrPipe::MAXL)
        return caqr_factorize_pipelined(A, lda, m, n, Tbuf, s, stream, side, *pipe, ev_urgent);
    if (!side || !ev_urgent || !ev_factored) {
        for (int p = 0; p < s.NP; ++p) {
            const int pc = p * caqr::NB, w = width(p);
            if ((e = caqr_panel_levels(A, lda, m, p, w, Tbuf, s, true, 1, A + (int64_t)(pc + w) * lda, lda, n - (pc + w), false, stream)) != hipSuccess)
                return e;
        }
        return hipSuccess;
    }
    if ((e = caqr_panel_levels(A, lda, m, 0, width(0), Tbuf, s, true, 1, nullptr, lda, 0, false, stream)) != hipSuccess) return e;
    for (int p = 0; p < s.NP; ++p) {
        const int pc = p * caqr::NB, w = width(p);
        const int pc1 = pc + w, w1 = p + 1 < s.NP ? width(p + 1) : 0, nrest = n - (pc1 + w1);
        if (w1 > 0) {
            if ((e = caqr_panel_levels(A, lda, m, p, w, Tbuf, s, false, 1, A + (int64_t)pc1 * lda, lda, w1, false, stream)) != hipSuccess) return e;
            if ((e = hipEventRecord(ev_urgent, stream)) != hipSuccess || (e = hipStreamWaitEvent(side, ev_urgent, 0)) != hipSuccess) return e;
            if ((e = caqr_panel_levels(A, lda, m, p + 1, w1, Tbuf, s, true, 1, nullptr, lda, 0, false, side)) != hipSuccess) return e;
            if ((e = hipEventRecord(ev_factored, side)) != hipSuccess) return e;
        }
        static const bool beside = std::getenv("QRK_CAQR_BESIDE") && std::getenv("QRK_CAQR_BESIDE")[0] == '1';   // (off: see APPLY_LDS_BESIDE)
        if (nrest > 0 &&
            (e = caqr_panel_levels(A, lda, m, p, w, Tbuf, s, false, 1, A + (int64_t)(pc1 + w1) * lda, lda, nrest, false, stream,
                                   w1 > 0 && beside ? caqr::APPLY_LDS_BESIDE : caqr::APPLY_LDS)) != hipSuccess)
            return e;
        if (w1 > 0 && (e = hipStreamWaitEvent(stream, ev_factored, 0)) != hipSuccess) return e;
    }
    return hipSuccess;
}

// B (m x nrhs, column-major) <- Q0^T B (transpose != 0) or Q0 B.
hipError_t launch_caqr_apply(const double* A, int64_t lda, int m, int n, const double* Tbuf, int transpose, double* B,
                             int64_t ldb, int64_t nrhs, hipStream_t stream)
{
    if (nrhs <= 0) return hipSuccess;
    const CaqrShape s = caqr_shape(m, n);
    for (int pi = 0; pi < s.NP; ++pi) {
        const int p = transpose ? pi : s.NP - 1 - pi;
        const int pc = p * caqr::NB;
        const int w = n - pc < caqr::NB ? n - pc : caqr::NB;
        const hipError_t e = caqr_panel_levels(const_cast<double*>(A), lda, m, p, w, const_cast<double*>(Tbuf), s, false,
                                               transpose, B, ldb, (int)nrhs, !transpose, stream);
        if (e != hipSuccess) return e;
    }
    return hipSuccess;
}

hipError_t launch_caqr_copy_upper(const double* src, int64_t lds_, double* dst, int64_t ldd, int n, int zero_lower, hipStream_t stream)
{
    const int64_t tot = (int64_t)n * n;
    hipLaunchKernelGGL(caqr::caqr_copy_upper_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, stream, src, lds_, dst, ldd, n,
                       zero_lower);
    return hipGetLastError();
}

}  // namespace qrk
