// qrk_device.h -- shared device-side declarations for the gfx950 kernels.
#ifndef QRK_DEVICE_H
#define QRK_DEVICE_H

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <float.h>

namespace qrk {

// Per-launch description of a batch of tiles that all fit one wave (rows, cols <= 32).
// Uniform batches leave the array pointers null and the kernel derives every
// offset from the tile index; mixed batches carry explicit per-tile arrays
// (prefix sums built on the host in qrk_bd_plan_create = the running offsets
// base_row/base_col of BlockDiagonalSparseQR.h:428-431,524-525).
struct WaveBatch {
    int64_t num_tiles;        // tiles in this launch
    int32_t rows, cols;       // uniform size (used when tile_ids == nullptr)
    const int32_t* tile_ids;  // [num_tiles] global tile index of each tile of this bin, or null
    const int32_t* t_rows;    // per global tile (mixed only)
    const int32_t* t_cols;
    const int64_t* t_off;     // offset of the tile in `tiles`
    const int64_t* q_off;     // offset of row 0 of Q_i in q_vals
    const int64_t* r_off;     // offset of R_i in r_vals
    const int32_t* c_off;     // base_col of the tile
    int32_t pivoting;         // 1 = ColPivHouseholderQR, 0 = HouseholderQR
};

// Per-tile geometry for the auxiliary kernels (uniform batches compute it from the index).
struct TileGeom {
    int64_t num_tiles;
    int32_t rows, cols;          // uniform size when t_rows == nullptr
    const int32_t* t_rows;
    const int32_t* t_cols;
    const int64_t* q_off;        // sum r^2 before the tile
    const int64_t* r_off;        // sum c(c+1)/2 before the tile
    const int32_t* c_off;        // base_col
    const int32_t* row_off;      // base_row
    int32_t mat_rows, mat_cols;
    int32_t sum_rows;            // rows covered by tiles; rows beyond get Q(i,i) = 1
    int64_t nnz_q_tiles;         // sum r^2
    int32_t q_format;            // 0 FullQ, 1 BlockDiagonalQ
};

// ---- host-side launchers (defined next to their kernels)
// redo_count / redo_ids: list of tiles whose decisions were not clear of rounding (redone by launch_bdqr_exact)
// bdqr_pair4.hip (QRK_PAIR_V2=0 disables): uniform 32 x 32 batches, two tiles per wavefront, four wavefronts per SIMD
int64_t bdqr_pair4_scratch_doubles(int num_wg);
bool bdqr_pair4_own_norm(int64_t num_tiles, int num_wg);
hipError_t launch_bdqr_pair4(int64_t num_tiles, int pivoting, const double* tiles, double* q_vals, double* r_vals, int32_t* perm,
                             double* hcoeffs, double* scratch, int num_wg, hipStream_t stream);
// bdqr_quad32.hip: uniform 32 x 32 batches, FOUR tiles per wavefront, two wavefronts per SIMD (num_wg: 8 per CU); direct < 0: by launch size
int64_t bdqr_quad32_scratch_doubles(int num_wg);
bool bdqr_quad32_preferred(int64_t num_tiles, int num_wg);
hipError_t launch_bdqr_quad32(int64_t num_tiles, int pivoting, const double* tiles, double* q_vals, double* r_vals, int32_t* perm,
                              double* hcoeffs, double* scratch, int num_wg, int direct, hipStream_t stream);
void launch_bdqr_pair(const WaveBatch& nb, bool full32, const double* tiles, double* q_vals,
                      double* r_vals, int32_t* perm, double* hcoeffs, int max_blocks,
                      int32_t* redo_count, int32_t* redo_ids, hipStream_t stream);
// Uniform batches of small tiles (rows <= 16, cols <= rows): 64/G tiles per wavefront (bdqr_small.hip).
// bdqr_quad.hip: uniform tiles with 9 .. 16 rows, four tiles per wavefront (two phases, DPP broadcast); flagged tiles -> redo list
bool bdqr_quad_supported(int r, int c);
int bdqr_quad_waves_per_cu(int r);
void launch_bdqr_quad(int64_t num_tiles, int r, int c, int pivoting, const double* tiles, double* q_vals, double* r_vals, int32_t* perm,
                      double* hcoeffs, int num_wg, int32_t* redo_count, int32_t* redo_ids, hipStream_t stream);
void launch_bdqr_small(int64_t num_tiles, int r, int c, int pivoting, const double* tiles, double* q_vals, double* r_vals,
                       int32_t* perm, double* hcoeffs, int max_blocks, int32_t* redo_count, int32_t* redo_ids, hipStream_t stream);
// Uniform batches of tall-thin small tiles (1 or 2 columns, rows <= 16): one tile per lane, no LDS (bdqr_thin.hip).
void launch_bdqr_thin(int64_t num_tiles, int r, int c, int pivoting, const double* tiles, double* q_vals, double* r_vals,
                      int32_t* perm, double* hcoeffs, int max_blocks, int32_t* redo_count, int32_t* redo_ids, hipStream_t stream);
// Mid-size tiles (32 < max(rows, cols) <= 256, rows >= cols): one thread per column of A, blocked Q (bdqr_col.hip).
// One launch serves one size class (columns <= 64, <= 128, <= 256): its LDS is carved for the largest tile of the class.
hipError_t launch_bdqr_col(const WaveBatch& nb, const double* tiles, double* q_vals, double* r_vals, int32_t* perm,
                           double* hcoeffs, double* workspace, int64_t ws_stride, int num_wg, int max_rows,
                           int max_cols, int w_lds, int32_t* redo_count, int32_t* redo_ids, int32_t* queue, hipStream_t stream);
// The tiles of that range with more than 64 columns, on chip: registers + LDS, 512 threads per tile, one workgroup per CU (bdqr_reg.hip).
hipError_t launch_bdqr_reg(const WaveBatch& nb, const double* tiles, double* q_vals, double* r_vals, int32_t* perm, double* hcoeffs,
                           double* workspace, int64_t ws_stride, int num_wg, int max_rows, int max_cols, int32_t* redo_count,
                           int32_t* redo_ids, int32_t* queue, hipStream_t stream);
// Tiles with 32 < rows <= 64, cols <= rows: one wavefront per tile, registers + LDS, no workspace (bdqr_w64.hip).
bool bdqr_w64_supported(int rows, int cols);
hipError_t launch_bdqr_w64(const WaveBatch& nb, const double* tiles, double* q_vals, double* r_vals, int32_t* perm, double* hcoeffs,
                           int num_cus, int max_rows, int32_t* redo_count, int32_t* redo_ids, int32_t* queue, hipStream_t stream);
int64_t bdqr_reg_ws_doubles();      // workspace of one workgroup of launch_bdqr_reg
bool bdqr_reg_small(int max_rows, int max_cols);   // the launch runs the 4-wave instantiation: two workgroups per CU
int bdqr_col_w_lds(int64_t max_rc, int64_t max_rc_fitting);
int bdqr_col_wgs_per_cu(int max_cols, int w_lds, int max_r);
constexpr int QRK_COL_W_LDS_MAX = 4352;     // doubles of LDS for A in the LDS-resident form (bdqr_col.hip)
void launch_bdqr_wg(const WaveBatch& nb, const double* tiles, double* q_vals, double* r_vals, int32_t* perm,
                    double* hcoeffs, double* workspace, int64_t ws_stride, int num_wg, int max_dim,
                    int32_t* redo_count, int32_t* redo_ids, hipStream_t stream);
// The exact-arithmetic path (bdqr_exact.hip): redoes the listed tiles with Eigen's operation order and rounding.
hipError_t launch_bdqr_exact(const WaveBatch& nb, const int32_t* ids, const int32_t* count, int32_t* next_count,
                             const double* tiles, double* q_vals, double* r_vals, int32_t* perm, double* hcoeffs,
                             double* workspace, int64_t ws_stride, int num_wg, int maxr, int maxc, hipStream_t stream);
bool bdqr_exact_needs_workspace(int maxr, int maxc);
size_t dense_qr_smem_bytes(int r, int c);
// Tall dense QR over all CUs (dense_qr_tall.hip): row slabs, one short kernel sequence per reflector.
size_t dense_tall_workspace_bytes(int r, int c, int num_cus, int* G, int* cpad, int* rows_per);
bool dense_tall_persistent_ok(int G, int rows_per, int num_cus);
hipError_t launch_dense_qr_tall(double* A, int64_t lda, int r, int c, int pivoting, double* hcoeffs, int32_t* perm,
                                void* workspace, int G, int cpad, int rows_per, bool persistent, hipStream_t stream);
hipError_t launch_dense_apply_q_tall(const double* QR, int64_t lda, int r, int nrefl, const double* hcoeffs, int transpose,
                                     double* B, int64_t ldb, int64_t nrhs, hipStream_t stream);
hipError_t launch_dense_qr(double* A, int64_t lda, int r, int c, int pivoting, double* hcoeffs, int32_t* perm, int* unclear,
                           hipStream_t stream);
int* dense_tall_unclear_ptr(void* workspace, int G, int cpad);
// The exact-arithmetic path of the dense solver (bdqr_exact.hip): when *unclear != 0 (or unclear == nullptr) the matrix is
// restored from `copy` (leading dimension r) and factorised again in Eigen's operation order, in place.
hipError_t launch_dense_exact(double* A, int64_t lda, int r, int c, int pivoting, const double* copy, double* hcoeffs,
                              int32_t* perm, const int* unclear, double* workspace, hipStream_t stream);
size_t dense_exact_workspace_bytes(int r, int c);
// [c] m_colNormsUpdated of the column chosen at every step of the LAST exact factorisation that used `workspace` (written by both exact
// forms; Eigen counts nonzeroPivots() from these)
const double* dense_exact_pivot_norms(const double* workspace, int r, int c);
// the exact path of a large dense block over the whole chip (A already restored; host-launched sequence, 2 launches per reflector)
hipError_t launch_dense_exact_wide(double* A, int64_t lda, int r, int c, int pivoting, double* hcoeffs, int32_t* perm,
                                   double* workspace, hipStream_t stream);
hipError_t launch_dense_apply_q(const double* QR, int64_t lda, int r, int nrefl, const double* hcoeffs,
                                int transpose, double* B, int64_t ldb, int64_t nrhs, hipStream_t stream);
// Un-pivoted communication-avoiding QR of a tall matrix on all CUs, MFMA trailing update (caqr.hip): first stage of the pivoted
// factorisation of tall dense right blocks.  Tbuf: caqr_t_bytes(m, n).
size_t caqr_t_bytes(int m, int n);
// third stream and events of the look-ahead pipelined by levels (caqr.hip, caqr_factorize_pipelined); all owned by the caller
struct CaqrPipe {
    static constexpr int MAXL = 8;
    hipStream_t urgent = nullptr;
    hipEvent_t ev_lvl[MAXL] = {};      // level l of the panel being factorised on the side stream is done
    hipEvent_t ev_n2 = nullptr;        // the caller's stream has applied panel p to the columns of panel p + 2
    hipEvent_t ev_u = nullptr;         // the urgent stream has applied every level of panel p + 1 to the columns of panel p + 2
};
hipError_t launch_caqr_factorize(double* A, int64_t lda, int m, int n, double* Tbuf, hipStream_t stream, hipStream_t side = nullptr,
                                 hipEvent_t ev_urgent = nullptr, hipEvent_t ev_factored = nullptr, const CaqrPipe* pipe = nullptr);
hipError_t launch_caqr_apply(const double* A, int64_t lda, int m, int n, const double* Tbuf, int transpose, double* B, int64_t ldb,
                             int64_t nrhs, hipStream_t stream);
hipError_t launch_caqr_copy_upper(const double* src, int64_t lds_, double* dst, int64_t ldd, int n, int zero_lower, hipStream_t stream);
// Column-parallel pivoted Householder QR of a square-ish matrix whose columns fit LDS (dense_qr_cols.hip): one kernel per reflector,
// a wavefront per column; second stage of the two-stage form.
// ... and the same as ONE persistent launch with the matrix in registers and a grid barrier per reflector (dense_qr_pers.hip): up to
// 2048 x 2048, rows >= cols; launch_dense_qr_cols dispatches to it (QRK_DENSE_PERS=0: never)
size_t dense_pers_workspace_bytes();
bool dense_pers_supported(int r, int c, int num_cus);
hipError_t launch_dense_qr_pers(const double* A, int64_t lda, int r, int c, int pivoting, double* hcoeffs, int32_t* perm, void* state,
                                void* workspace, int num_cus, double* out, int64_t ldo, hipStream_t stream);
size_t dense_cols_workspace_bytes(int c, int* cpad, bool pers);
bool dense_cols_supported(int r, int c);
int* dense_cols_unclear_ptr(void* workspace, int cpad);
hipError_t launch_dense_qr_cols(double* A, int64_t lda, int r, int c, int pivoting, double* hcoeffs, int32_t* perm, void* workspace,
                                int cpad, double* out, int64_t ldo, int pers_cus, hipStream_t stream);
// y(0:rows) -= sum_c S(:, colidx[c]) z[c] (bd_aux.hip): the strip term of the angular back substitution
hipError_t launch_gemv_sub(const double* S, int64_t lds, int64_t rows, int64_t cols, const int32_t* colidx, const double* z, double* y,
                           hipStream_t stream);
// dense column-major copy of rows [row0, row0 + nrows) of a sparse matrix on the device (bd_aux.hip)
hipError_t launch_sparse_window_to_dense(bool row_major, int64_t rows, int64_t cols, const int32_t* outer, const int32_t* inner,
                                         const double* vals, int64_t row0, int64_t nrows, const int32_t* row_map, double* out,
                                         int64_t ld, hipStream_t stream);
struct BBPanel;
hipError_t launch_bb_chain(const BBPanel* panels, int num_panels, const int32_t* prowptr, const int32_t* pcol,
                           const int64_t* pmap, const double* vals, double* W, double* lo, double* y_vals,
                           double* t_vals, double* r_stage, const int64_t* r_src, int64_t nnz_r, double* r_vals,
                           int max_act_rows, int max_ncols, hipStream_t stream);
hipError_t launch_bb_apply_q(const BBPanel* panels, int num_panels, const double* y_vals, const double* t_vals,
                             int transpose, double* v, int64_t ldv, int64_t nrhs, int max_act_rows, int max_ncols,
                             hipStream_t stream);
// blocked application of Q1^T of a two-stage factorisation (dense_qr.hip): T factors of blocks of 32 reflectors, then a launch per block over several workgroups
size_t dense_q_tfactors_doubles(int nrefl);
hipError_t launch_dense_q_tfactors(const double* QR, int64_t lda, int n, int nrefl, const double* tau, double* T, hipStream_t stream);
hipError_t launch_dense_apply_qt_blocks(const double* QR, int64_t lda, int n, int nrefl, const double* T, double* B, int64_t ldb, int64_t nrhs,
                                        double* work, hipStream_t stream);
hipError_t launch_dense_solve_r(const double* qr, int64_t lda, int n, double* b, int64_t ldb, int64_t nrhs, hipStream_t stream,
                                int* flags = nullptr, int flags_cap = 0);
hipError_t launch_bb_solve_r(const BBPanel* panels, int num_panels, const double* r_stage, int cols, double* v, int64_t ldv,
                             int64_t nrhs, hipStream_t stream);
size_t bb_chain_smem(int max_act_rows, int max_ncols);
size_t bb_apply_smem(int max_act_rows, int max_ncols);
// strips form of the banded factorisation (banded.hip): stage B on the triangles stage A left
hipError_t launch_bbs_chain(const BBPanel* panels, int num_panels, const double* r_packed, int64_t r_stride, int n, int lo,
                            int max_act_rows, double* lo_buf, double* y_vals, double* t_vals, double* r_stage,
                            const int* rlim_first, const int* rlim_rest, int* done, int single, int* piped_out, hipStream_t stream);
hipError_t launch_bbs_apply(const BBPanel* panels, int num_panels, const double* y_vals, const double* t_vals, int transpose,
                            double* ya, int64_t ya_ld, double* full, int64_t full_ld, int64_t nrhs, int ms, int n, int s, int lo,
                            int cols, int max_act_rows, hipStream_t stream);
// strips form through the carry maps (banded_maps.hip): the chains of Q^T b / Q x / R^-1 y reduced to one small matrix per strip and run
// in two levels (groups of K strips at once, then the group boundaries)
void bbs_maps_sizes(int num_panels, int lo, int* K, int64_t* cprod_len, int64_t* aprod_len, int64_t* gvec_len_per_rhs);
hipError_t launch_bbs_maps(const BBPanel* panels, int num_panels, const double* y_vals, const double* t_vals, const double* r_stage, int n,
                           int lo, double* cmap, double* amap, int K, double* cprod, double* aprod, hipStream_t stream);
hipError_t launch_bbs_apply_maps(const BBPanel* panels, int num_panels, const double* y_vals, const double* t_vals, const double* cmap,
                                 const double* cprod, int K, int transpose, double* ya, int64_t ya_ld, double* full, int64_t full_ld,
                                 int64_t nrhs, int ms, int n, int s, int lo, int cols, int max_act, double* carr, double* gvec, hipStream_t stream);
hipError_t launch_bbs_solve_r_maps(const BBPanel* panels, int num_panels, const double* r_stage, const double* amap, const double* aprod, int K,
                                   int s, int lo, int cols, double* v, int64_t ldv, int64_t nrhs, double* U, double* gvec, hipStream_t stream);
void launch_bd_pattern(const TileGeom& g, int64_t nnz_r, int32_t* q_rowptr, int32_t* q_colidx,
                       int32_t* r_colptr, int32_t* r_rowidx, hipStream_t stream);
void launch_bd_cut_tiles(const TileGeom& g, const int64_t* t_off, int row_major, const int32_t* outer_ptr,
                         const int32_t* inner_idx, const double* vals, int32_t nnz, double* tiles, hipStream_t stream);
void launch_bd_q_tail_ones(double* q_vals, int64_t start, int64_t count, hipStream_t stream);
void launch_bd_apply_qt(const TileGeom& g, const double* q_vals, const double* b, int64_t nrhs,
                        double* y, hipStream_t stream);
void launch_bd_apply_q(const TileGeom& g, int max_rows, const double* q_vals, const double* b, int64_t nrhs, double* y, hipStream_t stream);
void launch_bd_solve_r(const TileGeom& g, int max_cols, const double* r_vals, const double* y, int64_t nrhs, double* z,
                       hipStream_t stream);
void launch_bd_solve(const TileGeom& g, int max_cols, const double* q_vals, const double* r_vals,
                     const int32_t* perm, const double* b, int64_t nrhs, double* x,
                     hipStream_t stream);

// ---- decision margins of the fast kernels ------------------------------------------------
// A fast kernel (FMA chains, squared column norms, un-normalised reflector) takes a data-dependent decision of the reference
// algorithm only when it is clear of rounding, and otherwise sends the tile to the exact path (bdqr_exact.hip), which repeats
// it in Eigen's own operation order; see bdqr_pair.hip, "Decisions and the exact path", for the error model.  Here as plain
// double arithmetic for the thread-per-column kernels (bdqr_col, bdqr_wg, dense_qr*): a2 = |A|^2 is the squared norm of the
// first pivot column, which bounds every later column norm and scales the absolute error of the column entries.
namespace decide {
constexpr double MREL = 0.000244140625;               // 2^-12 = 2^14 eps / sqrt(eps)
constexpr double THR_HI = 1.4901161193847656e-08 * (1.0 + MREL);   // thr = sqrt(eps) (1 + 2^-12) normDirect^2: upper edge of the recompute band
constexpr double X0_TINY2 = 1.2924697071141057e-26;   // 2^-86: x0^2 <= (2^9 eps |A|)^2 leaves the sign of beta to rounding noise
constexpr double PIV_TINY2 = 8.673617379884035e-19;   // 2^-60: a pivot column below 2^-30 |A| is at the noise level of the tile
constexpr double ND_TINY2 = 5.820766091346741e-11;    // 2^-34: below 2^-17 |A| the error of a column norm passes the 2^-12 band

// (1) another live column within the error margin of the chosen one: nu2 of a column (thr its band edge) against the best
__device__ __forceinline__ bool near_best(double nu2, double thr, double best, double a2)
{
    const double margin = MREL * (thr + THR_HI * a2) + 4.547473508864641e-13 /* 2^-41 */ * sqrt(a2 * (best > 0.0 ? best : 0.0));
    return nu2 >= best - margin;
}
// (2) the downdated squared norm nn passed the recompute test (nn <= thr) inside the band around Eigen's threshold, or on a
// column so small against |A| that the band does not cover the error of its norm
__device__ __forceinline__ bool in_recompute_band(double nn, double thr, double a2)
{
    return nn > thr * (1.0 - 2.0 * MREL) || thr <= (THR_HI * ND_TINY2) * a2;
}
// (3) degenerate reflector on a non-empty tail, (4) |x0| too small to fix the sign of beta, (5) pivot at the noise level.
// sign_free: the caller's R is compared with Eigen's up to the signs of its rows anyway (second stage of the two-stage dense
// form, whose first stage changed the basis): (3) and (4) decide only the sign of beta - H = I against a reflector with the same
// |R_kk| - and are no reason for the exact path.  They are common there: structurally orthogonal columns (cameras of a bundle
// adjustment that share no point) leave entries at the noise level, or exact zeros, in the first stage's triangle.
__device__ __forceinline__ bool unclear_reflector(double xk, double tsq, bool tail, bool pivoting, double a2, bool sign_free = false)
{
    const double n2 = fma(xk, xk, tsq);
    return (!sign_free && tail && (!(tsq > DBL_MIN) || xk * xk <= X0_TINY2 * a2)) || (pivoting && n2 <= PIV_TINY2 * a2);
}
constexpr int PIVOTING_SIGN_FREE = 2;     // bit of the launchers' `pivoting` argument that sets sign_free
}  // namespace decide

// Tiles are read once per factorisation: the non-temporal hint keeps them from displacing the results that are being merged in L2
// (bdqr_pair4.hip: 78.0 -> 74.3 us per 10 000 tiles of 32 x 32; the small-tile kernels + 2-4 %; profiles/r04_p4_nt.txt).  Only where
// every byte of a line is taken by one load instruction: bdqr_w64 / bdqr_reg read a column per lane and live on the reuse of
// their lines (64 x 64: 16.5 -> 13 M tiles/s with the hint).  -DQRK_NT_TILES=0: plain loads.
#ifndef QRK_NT_TILES
#define QRK_NT_TILES 1
#endif
#if QRK_NT_TILES
#define QRK_TILE_LOAD(p) __builtin_nontemporal_load(p)
#else
#define QRK_TILE_LOAD(p) (*(p))
#endif
// Results that leave in coalesced sweeps of whole cache lines (the LDS-staged small-tile kernels).  -DQRK_NT_OUT=1: non-temporal stores
// (tools/ubench_stream_mix.hip: a 1 : 4 read : write stream of 1 GB runs at 0.65 of the nominal HBM rate with plain and at 0.85 with
// non-temporal stores).  NOT for stores that fill a line piecewise (bdqr_pair4's rows of Q and columns of R: 2 x slower, r04_p4_nt.txt).
#ifndef QRK_NT_OUT
#define QRK_NT_OUT 0
#endif
#if QRK_NT_OUT
#define QRK_OUT_STORE(p, v) __builtin_nontemporal_store((v), (p))
#else
#define QRK_OUT_STORE(p, v) (*(p) = (v))
#endif

// ---- cross-lane helpers (wave64) ---------------------------------------------------------

__device__ __forceinline__ double readlane_f64(double v, int lane)
{
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_readlane(lo, lane);
    hi = __builtin_amdgcn_readlane(hi, lane);
    return __hiloint2double(hi, lo);
}

template <int CTRL>
__device__ __forceinline__ double dpp_f64(double v)
{
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(lo, lo, CTRL, 0xF, 0xF, false);
    hi = __builtin_amdgcn_update_dpp(hi, hi, CTRL, 0xF, 0xF, false);
    return __hiloint2double(hi, lo);
}

// All-reduce max inside every row of 16 lanes: quad xor1, quad xor2, half mirror, row mirror.
__device__ __forceinline__ double row16_max(double v)
{
    v = fmax(v, dpp_f64<0xB1>(v));   // quad_perm [1,0,3,2]
    v = fmax(v, dpp_f64<0x4E>(v));   // quad_perm [2,3,0,1]
    v = fmax(v, dpp_f64<0x141>(v));  // row_half_mirror
    v = fmax(v, dpp_f64<0x140>(v));  // row_mirror
    return v;
}

template <int CTRL>
__device__ __forceinline__ int dpp_i32(int v)
{
    return __builtin_amdgcn_update_dpp(v, v, CTRL, 0xF, 0xF, false);
}

__device__ __forceinline__ int row16_min_i32(int v)
{
    v = min(v, dpp_i32<0xB1>(v));
    v = min(v, dpp_i32<0x4E>(v));
    v = min(v, dpp_i32<0x141>(v));
    v = min(v, dpp_i32<0x140>(v));
    return v;
}

// DPP-fused integer max (one VALU instruction per stage).  The s_nop covers the "VALU write ->
// DPP read of the same VGPR" hazard (2 wait states), which hipcc does not pad inside asm.
#define QRK_DPP_OP(NAME, OP, CTRL)                                                              \
    __device__ __forceinline__ int NAME(int v)                                                  \
    {                                                                                           \
        int r;                                                                                  \
        asm("s_nop 1\n\t" OP " %0, %1, %1 " CTRL " row_mask:0xf bank_mask:0xf"         \
                     : "=v"(r) : "v"(v));                                                       \
        return r;                                                                               \
    }
QRK_DPP_OP(max_i32_xor1, "v_max_i32_dpp", "quad_perm:[1,0,3,2]")
QRK_DPP_OP(max_i32_xor2, "v_max_i32_dpp", "quad_perm:[2,3,0,1]")
QRK_DPP_OP(max_i32_hmir, "v_max_i32_dpp", "row_half_mirror")
QRK_DPP_OP(max_i32_mir, "v_max_i32_dpp", "row_mirror")
#undef QRK_DPP_OP

// Max over each half of the wave (lanes 0..31 / 32..63), result in every lane of the half.
__device__ __forceinline__ int half32_max_i32_fast(int v)
{
    v = max_i32_mir(max_i32_hmir(max_i32_xor2(max_i32_xor1(v))));
    const auto r = __builtin_amdgcn_permlane16_swap((unsigned)v, (unsigned)v, false, false);
    return max((int)r[0], (int)r[1]);
}

// Same reduction with the four DPP stages in ONE asm statement (hipcc pads every asm statement that
// ends in a VALU write with another s_nop).
__device__ __forceinline__ int half32_max_i32_fused(int v)
{
    int m;
    asm("s_nop 1\n\t"
        "v_max_i32_dpp %0, %1, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\t"
        "v_max_i32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\t"
        "v_max_i32_dpp %0, %0, %0 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\t"
        "v_max_i32_dpp %0, %0, %0 row_mirror row_mask:0xf bank_mask:0xf"
        : "=&v"(m) : "v"(v));
    const auto r = __builtin_amdgcn_permlane16_swap((unsigned)m, (unsigned)m, false, false);
    return max((int)r[0], (int)r[1]);
}

// Max / min over each half of the wave (lanes 0..31 and 32..63 separately), result in every lane of
// the half: four DPP stages inside the rows of 16, then v_permlane16_swap to combine the two rows.
__device__ __forceinline__ int half32_max_i32(int v)
{
    v = max(v, dpp_i32<0xB1>(v));
    v = max(v, dpp_i32<0x4E>(v));
    v = max(v, dpp_i32<0x141>(v));
    v = max(v, dpp_i32<0x140>(v));
    const auto r = __builtin_amdgcn_permlane16_swap((unsigned)v, (unsigned)v, false, false);
    return max((int)r[0], (int)r[1]);
}

__device__ __forceinline__ unsigned half32_max_u32(unsigned v)
{
    v = max(v, (unsigned)dpp_i32<0xB1>((int)v));
    v = max(v, (unsigned)dpp_i32<0x4E>((int)v));
    v = max(v, (unsigned)dpp_i32<0x141>((int)v));
    v = max(v, (unsigned)dpp_i32<0x140>((int)v));
    const auto r = __builtin_amdgcn_permlane16_swap(v, v, false, false);
    return max((unsigned)r[0], (unsigned)r[1]);
}

__device__ __forceinline__ int half32_min_i32(int v)
{
    v = min(v, dpp_i32<0xB1>(v));
    v = min(v, dpp_i32<0x4E>(v));
    v = min(v, dpp_i32<0x141>(v));
    v = min(v, dpp_i32<0x140>(v));
    const auto r = __builtin_amdgcn_permlane16_swap((unsigned)v, (unsigned)v, false, false);
    return min((int)r[0], (int)r[1]);
}

}  // namespace qrk
#endif
