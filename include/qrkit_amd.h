/*
 * qrkit_amd.h -- C ABI of the MI355X (gfx950) structured sparse QR engine.
 *
 * This is the drop-in boundary for QRKit's block-diagonal hot path.  QRKit has
 * no FFI of its own (header-only C++ templates); the seam these entry points
 * replace is the body of
 *     QRKit::BlockDiagonalSparseQR<BlockQRSolver,QFormat>::analyzePattern
 *         (src/QRKit/BlockDiagonalSparseQR.h:392-405)
 *     QRKit::BlockDiagonalSparseQR<...>::factorize   (:415-547)
 *     QRKit::BlockDiagonalSparseQR<...>::_solve_impl (:257-280)
 * i.e. everything between "tiles of a SparseBlockDiagonal in" and
 * "values of m_Q / m_R / m_outputPerm_c out".  The C++ facade in
 * include/qrkit/ keeps the reference's class and method names on top of it;
 * INTEGRATION.md shows the binding a QRKit maintainer would add.
 *
 * Conventions
 *  - plain C types only; the caller owns every buffer; the library never frees
 *    or retains caller memory beyond the call (device work is enqueued on the
 *    handle's HIP stream and has completed when the stream has);
 *  - all matrices are double, tiles column-major (Eigen::Matrix default);
 *  - index arrays are int32 (QRKit's StorageIndex = int);
 *  - one handle per host thread (no internal locking);
 *  - every function returns qrk_status; qrk_last_error() gives the text;
 *  - there is NO CPU fallback: without a HIP device qrk_create() fails with
 *    QRK_STATUS_NO_DEVICE and nothing else can be called.
 */
#ifndef QRKIT_AMD_H
#define QRKIT_AMD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define QRK_VERSION_MAJOR 0
#define QRK_VERSION_MINOR 1

typedef enum qrk_status {
    QRK_STATUS_OK = 0,
    QRK_STATUS_INVALID_ARGUMENT = 1,
    QRK_STATUS_NO_DEVICE = 2,
    QRK_STATUS_HIP_ERROR = 3,
    QRK_STATUS_ALLOC_FAILED = 4,
    QRK_STATUS_UNSUPPORTED = 5,
    QRK_STATUS_NOT_FACTORIZED = 6
} qrk_status;

/* Eigen::ComputationInfo, as returned by QRKit's info() (BlockDiagonalSparseQR.h:309-313). */
typedef enum qrk_info {
    QRK_INFO_SUCCESS = 0,
    QRK_INFO_NUMERICAL_ISSUE = 1,
    QRK_INFO_NO_CONVERGENCE = 2,
    QRK_INFO_INVALID_INPUT = 3
} qrk_info;

/* BlockDiagonalSparseQR::MatrixQFormat (BlockDiagonalSparseQR.h:59-62). */
typedef enum qrk_q_format { QRK_FULL_Q = 0, QRK_BLOCK_DIAGONAL_Q = 1 } qrk_q_format;

/* Which Eigen dense solver stands behind the _BlockQRSolver template parameter
 * (BlockDiagonalSparseQR.h:37): ColPivHouseholderQR (test/test-qrkit.cpp:32-38,
 * 49-51) or HouseholderQR (examples/ellipse_fitting.cpp:153). */
typedef enum qrk_block_solver { QRK_COLPIV_HOUSEHOLDER = 0, QRK_HOUSEHOLDER = 1 } qrk_block_solver;

/* Where the caller's buffers live. */
typedef enum qrk_memspace { QRK_MEM_DEVICE = 0, QRK_MEM_HOST = 1 } qrk_memspace;

typedef struct qrk_context_s* qrk_handle;
typedef struct qrk_bd_plan_s* qrk_bd_plan;

/* ------------------------------------------------------------------ context */

/* Library version as major*1000+minor. */
int qrk_version(void);

/* Number of HIP devices visible (0 when there is none; never initialises a context). */
int qrk_device_count(void);

/* Create a context on HIP device `device`.  `stream` is a hipStream_t (or NULL
 * for the device's default stream) on which all work of this handle is enqueued. */
qrk_status qrk_create(qrk_handle* out, int device, void* stream);
qrk_status qrk_destroy(qrk_handle h);
qrk_status qrk_set_stream(qrk_handle h, void* stream);
/* Block until everything enqueued on the handle's stream has finished. */
qrk_status qrk_synchronize(qrk_handle h);
/* Text of the last error on this handle (h may be NULL for creation errors). */
const char* qrk_last_error(qrk_handle h);

/* ------------------------------------------------------- device memory helpers */

/* For callers without a HIP toolchain of their own (the C++ facade in include/qrkit/QRKit.hpp keeps the
 * factors on the device between factorize() and solve() with these): plain device memory on the
 * context's device, and copies ordered on the context's stream that return when the data has arrived.
 * direction: 0 = host -> device, 1 = device -> host. */
qrk_status qrk_device_alloc(qrk_handle h, int64_t bytes, void** out);
qrk_status qrk_device_free(qrk_handle h, void* ptr);
qrk_status qrk_memcpy(qrk_handle h, void* dst, const void* src, int64_t bytes, int direction);
/* `height` runs of `width_bytes` bytes, `dst_pitch` / `src_pitch` bytes apart (a block of columns of a column-major matrix, e.g.
 * the top or bottom rows of the dense right block J2, BlockAngularSparseQR.h:361-369).  direction: 0 = host -> device,
 * 1 = device -> host, 2 = device -> device. */
qrk_status qrk_memcpy_2d(qrk_handle h, void* dst, int64_t dst_pitch, const void* src, int64_t src_pitch, int64_t width_bytes,
                         int64_t height, int direction);

/* Dense column-major copy of the row window [row0, row0 + nrows) of a compressed sparse matrix whose arrays are on the device
 * (CSR when row_major != 0, else CSC; Eigen's outerIndexPtr / innerIndexPtr / valuePtr of a compressed matrix):
 * d_out[c * ld + (d_row_map ? d_row_map[r - row0] : r - row0)] = value of entry (r, c); every other element of the nrows x cols
 * window is zero.  d_row_map (device, nrows entries, a permutation of 0..nrows-1) or NULL.  Replaces the dense copy the
 * reference makes of a sparse right block - BlockedThinSparseQR::compute, BlockedThinSparseQR.h:131 "m_R = mat", reached from
 * BlockAngularSparseQR::solveRightBlock (BlockAngularSparseQR.h:361-369) - so that a sparse J2 crosses PCIe as its nonzeros. */
qrk_status qrk_sparse_window_to_dense(qrk_handle h, int row_major, int64_t rows, int64_t cols, const int32_t* d_outer,
                                      const int32_t* d_inner, const double* d_values, int64_t row0, int64_t nrows,
                                      const int32_t* d_row_map, double* d_out, int64_t ld);

/* -------------------------------------------- block-diagonal: analyzePattern */

/* The block structure of a QRKit::SparseBlockDiagonal (SparseBlockDiagonal.h:43-163):
 * num_blocks dense tiles on the diagonal, tile i being rows[i] x cols[i].  When
 * rows == cols == NULL every tile is block_rows x block_cols (the
 * fromBlockDiagonalPattern case, SparseBlockDiagonal.h:71-89).  mat_rows /
 * mat_cols are SparseBlockDiagonal::rows()/cols(); mat_rows may exceed the sum
 * of tile rows (trailing identity rows of Q, BlockDiagonalSparseQR.h:530-533);
 * mat_cols must equal the sum of tile cols.  rows/cols are HOST arrays. */
typedef struct qrk_bd_layout {
    int64_t num_blocks;
    int32_t block_rows;
    int32_t block_cols;
    const int32_t* rows;
    const int32_t* cols;
    int32_t mat_rows;
    int32_t mat_cols;
} qrk_bd_layout;

/* analyzePattern(): validates the layout, computes the prefix sums the hot loop
 * carries (base_row, base_col, m1; BlockDiagonalSparseQR.h:428-431,524-525),
 * uploads the per-tile descriptors and bins tiles by size class.  A landscape
 * tile (rows < cols) is accepted here and reported by factorize() through
 * info = QRK_INFO_INVALID_INPUT, as the reference does (:509-516). */
qrk_status qrk_bd_plan_create(qrk_handle h, const qrk_bd_layout* layout, qrk_q_format q_format,
                              qrk_block_solver solver, qrk_bd_plan* out);
qrk_status qrk_bd_plan_destroy(qrk_bd_plan plan);

/* Element counts of the caller-owned arrays:
 *   tiles_len = sum rows_i*cols_i          (input tiles, packed back to back)
 *   nnz_q     = sum rows_i^2 + (mat_rows - sum rows_i)   (m_Q values / column indices)
 *   nnz_r     = sum cols_i*(cols_i+1)/2    (m_R values / row indices)
 * perm and hcoeffs have mat_cols entries. */
qrk_status qrk_bd_plan_sizes(qrk_bd_plan plan, int64_t* tiles_len, int64_t* nnz_q, int64_t* nnz_r);

/* Sparse patterns exactly as factorize() assembles them: m_Q is RowMajor CSR
 * (mat_rows+1 row pointers, nnz_q column indices; BlockDiagonalSparseQR.h:455-492,
 * 530-536), m_R is ColMajor CSC (mat_cols+1 column pointers, nnz_r row indices;
 * :475-479,496-500,538-541).  Pure functions of the layout and q_format. */
qrk_status qrk_bd_pattern(qrk_bd_plan plan, int32_t* q_rowptr, int32_t* q_colidx, int32_t* r_colptr,
                          int32_t* r_rowidx, qrk_memspace space);

/* ------------------------------------------- block-diagonal: cutting the tiles */

/* SparseBlockDiagonal::fromBlockDiagonalPattern (SparseBlockDiagonal.h:71-89): tile i =
 * dense copy of mat.block(base_row_i, base_col_i, rows_i, cols_i), the block map being
 * BlockBandedMatrixInfo::fromBlockDiagonalPattern (SparseQRUtils.h:255-272; base_row / base_col
 * are the running sums of the plan's layout, = i*blockRows / i*blockCols for equal blocks).
 * `mat` is a compressed sparse matrix with int32 indices: column-major (CSC: mat_cols+1 outer
 * pointers, row indices) when row_major == 0, row-major (CSR: mat_rows+1 outer pointers, column
 * indices) otherwise.  Entries of `mat` outside the blocks are ignored, as block() ignores them.
 *   tiles [tiles_len] out: tile i column-major at the running offset (the input of qrk_bd_factorize) */
qrk_status qrk_bd_tiles_from_sparse(qrk_bd_plan plan, int row_major, const int32_t* outer_ptr,
                                    const int32_t* inner_idx, const double* vals, int64_t nnz, double* tiles,
                                    qrk_memspace space);

/* ------------------------------------------------- block-diagonal: factorize */

/* factorize(): per tile A_i P_i = Q_i R_i (BlockDiagonalSparseQR.h:432-526).
 *   tiles   [tiles_len]  in : tile i column-major at the running offset
 *   q_vals  [nnz_q]      out: values of m_Q in CSR order (row j of Q_i = [U_i(j,:), N_i(j,:)])
 *   r_vals  [nnz_r]      out: values of m_R in CSC order (upper triangle of R_i by columns)
 *   perm    [mat_cols]   out: m_outputPerm_c.indices() (:519-521)
 *   hcoeffs [mat_cols]   out, may be NULL: Householder coefficients tau of every tile
 * Asynchronous on the handle's stream for QRK_MEM_DEVICE; for QRK_MEM_HOST the
 * call stages through device buffers and returns after the results are on the host. */
qrk_status qrk_bd_factorize(qrk_bd_plan plan, const double* tiles, double* q_vals, double* r_vals,
                            int32_t* perm, double* hcoeffs, qrk_memspace space);

/* info() and rank() after factorize() (BlockDiagonalSparseQR.h:161-165,309-313):
 * rank = sum of tile cols (the solver is not rank revealing, :439-444). */
qrk_status qrk_bd_info(qrk_bd_plan plan, qrk_info* info, int64_t* rank);

/* ----------------------------------------------------- block-diagonal: solve */

/* y = Q^T b, the product the reference forms at BlockDiagonalSparseQR.h:266 and the
 * tests at test/test-qrkit.cpp:187.  b, y: mat_rows x nrhs column-major (ld = mat_rows). */
qrk_status qrk_bd_apply_qt(qrk_bd_plan plan, const double* q_vals, const double* b, int64_t nrhs,
                           double* y, qrk_memspace space);

/* y = Q b, matrixQ() * b with the explicit m_Q (BlockDiagonalSparseQR.h:235-237; the reference forms this
 * product as a sparse matrix-vector product, e.g. test/test-qrkit.cpp:201 and BlockAngularSparseQR.h:627-645).
 * b, y: nrhs columns of mat_rows entries; b and y must not alias. */
qrk_status qrk_bd_apply_q(qrk_bd_plan plan, const double* q_vals, const double* b, int64_t nrhs,
                          double* y, qrk_memspace space);

/* _solve_impl (BlockDiagonalSparseQR.h:257-280), FullQ only:
 * x = P * [ R(0:rank,0:rank)^-1 (Q^T b)(0:rank) ];  b: mat_rows x nrhs, x: mat_cols x nrhs. */
qrk_status qrk_bd_solve(qrk_bd_plan plan, const double* q_vals, const double* r_vals,
                        const int32_t* perm, const double* b, int64_t nrhs, double* x,
                        qrk_memspace space);

/* The triangular step of _solve_impl alone (BlockDiagonalSparseQR.h:271):
 * z = R(0:cols,0:cols).triangularView<Upper>().solve(y);  y, z: mat_cols x nrhs.  Used by the angular
 * composition, whose R = [R1, S; 0, R2] is solved block by block. */
qrk_status qrk_bd_solve_r(qrk_bd_plan plan, const double* r_vals, const double* y, int64_t nrhs, double* z,
                          qrk_memspace space);

/* --------------------------------------------- dense right-block solver (angular) */

/* A single dense Householder QR with implicit Q: the _BlockQRSolverRight of
 * QRKit::BlockAngularSparseQR (src/QRKit/BlockAngularSparseQR.h:79), which the reference tests
 * instantiate with Eigen::ColPivHouseholderQR<MatrixXd> (test/test-qrkit.cpp:46-48).  It is called at
 * BlockAngularSparseQR.h:361-369 (compute on the bottom rows of Q1^T J2), :488 (matrixR()), :498-503
 * (colsPermutation()) and :619-622 / :636-638 (matrixQ() products).  Q stays a sequence of reflectors. */
typedef struct qrk_dense_plan_s* qrk_dense_plan;

qrk_status qrk_dense_plan_create(qrk_handle h, int32_t rows, int32_t cols, qrk_block_solver solver,
                                 qrk_dense_plan* out);
qrk_status qrk_dense_plan_destroy(qrk_dense_plan plan);

/* compute(): a (rows x cols, column-major, leading dimension lda) is overwritten by the packed QR
 * (R in the upper triangle, essential Householder vectors below); hcoeffs[min(rows,cols)];
 * perm[cols] = colsPermutation().indices().
 *
 * Large tall pivoted problems (rows >= 4 cols, cols >= 128, rows * cols >= 2^22; QRK_DENSE_TWO_STAGE=0/1 overrides) are
 * factorised in two stages: A = Q0 R0 without pivoting on the matrix cores (communication-avoiding QR), then R0 P = Q1 R with
 * Eigen's pivot rule on the n x n triangle.  Same permutation; R equal to Eigen's up to the sign of each ROW (the sign of a
 * Householder beta follows the pivot entry, which the change of basis alters); Q = Q0 diag(Q1, I).  After such a factorisation the
 * upper triangle of `a` holds R as always, but what lies below are the reflectors of Q0, not Eigen's essential vectors, and
 * hcoeffs are those of Q1: use qrk_dense_apply_q / qrk_dense_solve_r with the same plan (the factors of Q1 stay in the plan),
 * do not interpret the lower part yourself.  A pivot decision inside its rounding margin still sends the whole matrix through
 * the exact path, which leaves Eigen's format; decisions that only fix the sign of a row of R (leading entry of a reflector at the
 * noise level, zero tail) do not, since that sign is open in this form anyway.
 *
 * Synchronisation: plans of at most 2^18 entries only enqueue (the exact path is queued behind the fast one and decides on the
 * device).  Two-stage plans and plans with rows * cols >= 2^18 SYNCHRONISE the handle's stream before they return: the host reads
 * one flag word (into a pinned buffer of the plan) to know which format the factors are in / whether the exact path has to be
 * launched over the whole chip.  Such a call cannot be captured into a hipGraph. */
qrk_status qrk_dense_factorize(qrk_dense_plan plan, double* a, int64_t lda, double* hcoeffs, int32_t* perm,
                               qrk_memspace space);

/* The two-stage format keeps part of Q (the T factors of Q0 and the packed Q1) IN THE PLAN, for the array factorised last:
 * qrk_dense_apply_q with any other array fails with QRK_STATUS_INVALID_ARGUMENT while that state is live (it would otherwise
 * apply the wrong Q without a sign of it).  A caller that factorises several matrices with one plan and applies their Qs later
 * (the panel chain of BlockedThinSparseQR, src/QRKit/BlockedThinSparseQR.h:105-165) switches the format off: every
 * factorisation then leaves Eigen's self-contained packed format in the caller's own arrays.  enable = 1 switches it back on
 * for a plan whose shape qualified at creation.  qrk_dense_plan_two_stage returns the current setting. */
qrk_status qrk_dense_plan_set_two_stage(qrk_dense_plan plan, int enable);
int qrk_dense_plan_two_stage(qrk_dense_plan plan);

/* b (rows x nrhs, leading dimension ldb) <- Q^T b (transpose != 0) or Q b. */
qrk_status qrk_dense_apply_q(qrk_dense_plan plan, const double* qr, int64_t lda, const double* hcoeffs,
                             int transpose, double* b, int64_t ldb, int64_t nrhs, qrk_memspace space);

/* b(0:cols, :) <- R^-1 b(0:cols, :) in place, R = the upper triangle of the packed QR (rows >= cols): the back
 * substitution of ColPivHouseholderQR::solve / of BlockAngularSparseQR::_solve_impl on its R2 block
 * (src/QRKit/BlockAngularSparseQR.h:202-227); the column permutation is the caller's. */
qrk_status qrk_dense_solve_r(qrk_dense_plan plan, const double* qr, int64_t lda, double* b, int64_t ldb, int64_t nrhs,
                             qrk_memspace space);

/* y(0:rows) -= sum_c S(:, colidx[c]) z[c]  for c = 0..cols-1 (colidx == NULL: column c), all device pointers, S column-major with
 * leading dimension lds: the strip term of the block back substitution of BlockAngularSparseQR::_solve_impl
 * (src/QRKit/BlockAngularSparseQR.h:202-227), z1 = R1^-1 (y1 - S z2) with S = (Q1^T J2)(0:m1, P2) kept on the device. */
qrk_status qrk_dense_gemv_sub(qrk_handle h, const double* S, int64_t lds, int64_t rows, int64_t cols, const int32_t* colidx,
                              const double* z, double* y);

/* ------------------------------------------------ thin sparse right solver (BlockedThinSparseQR) */

/* QRKit::BlockedThinSparseQR (src/QRKit/BlockedThinSparseQR.h:105-283): analyzePattern (:168-201: ColumnDensity column ordering,
 * SparseQROrdering.h:21-50, and as-banded-as-possible row ordering, :52-120) and compute (:105-165) of a thin sparse matrix given
 * in CSC on the host (colptr[cols + 1], rowidx / vals[nnz], row indices strictly increasing inside a column).  The permuted matrix
 * is made dense on the device (its nonzeros cross PCIe) and factorised panel by panel: a panel of block_cols columns takes the rows
 * its sparsity pattern says (updateBlockInfo, :203-238), is factorised by the column-pivoted dense solver (qrk_dense_factorize:
 * decisions inside rounding go through the exact path as everywhere), its reflectors are applied to the columns on the right, and
 * its columns of R are written (:271-279).  The number of nonzero pivots of a panel (:250-256, Eigen's nonzeroPivots(): counted on the
 * device from the downdated column norms -- the exact path's own table when a pivot is anywhere near the threshold) decides the rows
 * of the next one, so one word per panel travels to the host.  Both language mirrors (include/qrkit/QRKit.hpp, qrkit_amd/angular.py) call this. */
typedef struct qrk_thin_plan_s* qrk_thin_plan;
qrk_status qrk_thin_sparse_factorize(qrk_handle h, int32_t rows, int32_t cols, int32_t block_cols, const int32_t* colptr,
                                     const int32_t* rowidx, const double* vals, qrk_thin_plan* out);
qrk_status qrk_thin_destroy(qrk_thin_plan plan);
/* rank() = nonzero pivots; col_perm[cols] = colsPermutation().indices() (ColumnDensity permutation * Householder column
 * permutations, zero-pivot columns last, :151-159, :250-256); row_perm[rows] = rowsPermutation().indices().  Any pointer may be NULL. */
qrk_status qrk_thin_info(qrk_thin_plan plan, int32_t* rank, int32_t* col_perm, int32_t* row_perm);
/* matrixR(): the cols x cols upper triangle (rows rank.. are zero), column-major with leading dimension ldr */
qrk_status qrk_thin_matrix_r(qrk_thin_plan plan, double* r, int64_t ldr, qrk_memspace space);
/* v <- Q^T v (transpose != 0) or Q v, device pointer, nrhs columns of `rows` entries with leading dimension ldv >= 2 rows
 * (the panels are applied with zero rows appended; the entries rows..2 rows of a column must be zero and stay zero) */
qrk_status qrk_thin_apply_q(qrk_thin_plan plan, int transpose, double* v, int64_t ldv, int64_t nrhs);
/* BlockedThinQRBase::_solve_impl (BlockedThinQRBase.h:223-247) in place: v holds b (rows entries per column, ldv as above) and on
 * return x(0:cols) = [R(0:rank,0:rank)^-1 (Q^T b)(0:rank); 0] (the caller applies the permutations, as with the reference) */
qrk_status qrk_thin_solve(qrk_thin_plan plan, double* v, int64_t ldv, int64_t nrhs);

/* --------------------------------- banded matrix given as dense strips: two-stage factorisation */

/* BandedBlockedSparseQR::factorize (src/QRKit/BandedBlockedSparseQR.h:463-508) for a block-banded matrix whose block rows are
 * handed over as DENSE STRIPS: strip i is strip_rows x strip_cols and covers the rows [i strip_rows, (i+1) strip_rows) and the
 * columns [i col_step, i col_step + strip_cols) (BASELINE configs[2]: 50 000 strips of 256 x 192, step 64 -- 2.46e9 stored
 * entries, more than the int32 StorageIndex of the reference's SparseMatrix, and of qrk_bb_plan_create's CSR, can index; here
 * every offset is 64-bit).  The factorisation runs in two stages:
 *   A  every strip is triangularised on its own, A_i = Q_i [R_i; 0]: one workgroup per strip over all CUs (the block-diagonal
 *      solver's kernels, HouseholderQR);
 *   B  a chain over the strips merges the carried triangle (the strip_cols - col_step columns the next strip shares) with R_i:
 *      the two triangles are stacked with their rows interleaved, which makes the stack a staircase whose zeros the
 *      Householder sweep never visits; col_step rows of R leave per step, the rest is carried on.
 * The reference re-factorises (leftover + strip_rows) x strip_cols per step (:503-506).  R is the reference's up to the sign of
 * each ROW (R is unique up to row signs for a fixed column order; the signs follow the elimination order);
 * Q is kept as the two sequences of reflectors, not as the reference's Y / T blocks.  All pointers are device pointers.
 *
 * Layout of Q^T b / of the argument of Q (rows = num_strips strip_rows entries): first the cols = (num_strips - 1) col_step +
 * strip_cols entries that belong to the rows of R, then per strip i the components orthogonal to the columns: (strip_cols -
 * col_step) of the chain (strips 1..), then strip_rows - strip_cols of stage A. */
typedef struct qrk_bbs_plan_s* qrk_bbs_plan;
qrk_status qrk_bbs_plan_create(qrk_handle h, int64_t num_strips, int32_t strip_rows, int32_t strip_cols, int32_t col_step,
                               qrk_bbs_plan* out);
qrk_status qrk_bbs_plan_destroy(qrk_bbs_plan plan);
/* rows, cols of the matrix; r_len = doubles of R kept by the plan (col_step x strip_cols per strip, strip_cols^2 for the last) */
qrk_status qrk_bbs_plan_sizes(qrk_bbs_plan plan, int64_t* rows, int64_t* cols, int64_t* r_len);
/* strips: strip i column-major (leading dimension strip_rows) at strips + i strip_rows strip_cols.
 * Stage A is enqueued on the handle's stream.  When stage B takes the pipelined chain (three workgroups, more than two strips) the call
 * then SYNCHRONISES that stream: it reads the chain's abort word on the host and, if a hand-off wait ran out, runs stage B again on one
 * workgroup before it returns -- so the factors are complete on return in that case (the one-workgroup chain stays enqueue-only).
 * A caller that overlaps host work or other streams with this call should issue that work first. */
qrk_status qrk_bbs_factorize(qrk_bbs_plan plan, const double* strips);
/* the rows of R emitted by strip i: rows [i col_step, ..) x columns [i col_step, i col_step + strip_cols), column-major with
 * leading dimension = the number of rows (col_step; strip_cols for the last strip), upper trapezoidal */
qrk_status qrk_bbs_r_rows(qrk_bbs_plan plan, int64_t strip, double* r_rows);
/* out = Q^T v (transpose != 0) or Q v; v, out: rows x nrhs (ld = rows), distinct; work: rows x nrhs doubles.  v is not modified
 * logically (the Q v direction reads it only).
 * The chain from strip to strip carries strip_cols - col_step numbers and every strip acts on them linearly, so the first product
 * (or solve) after a factorisation forms one small matrix per strip on all CUs (qrkit_amd/csrc/banded_maps.hip: 2 x 8 (strip_cols -
 * col_step)^2 bytes per strip, kept by the plan until the next factorisation) and the products then run every strip at once plus a
 * two-level chain of those matrices: 0.8 us per strip instead of 70 at the BASELINE configs[2] shape.  QRK_BBS_MAPS=0, more than
 * 65 535 right-hand sides, or no memory for the matrices: one workgroup walks the strips as before (same result to rounding). */
qrk_status qrk_bbs_apply_q(qrk_bbs_plan plan, int transpose, const double* v, double* out, int64_t nrhs, double* work);
/* least squares: x (cols x nrhs, ld = cols) = R^-1 (Q^T b)(0:cols) (BandedBlockedSparseQR::_solve_impl, :290-311);
 * b: rows x nrhs; work: 2 rows nrhs doubles */
qrk_status qrk_bbs_solve(qrk_bbs_plan plan, const double* b, double* x, int64_t nrhs, double* work);

/* ------------------------------------- right block sharded over GPUs: local TSQR stage */

/* rightSolver.compute(J2.bottomRows(...)) (src/QRKit/BlockAngularSparseQR.h:361-369) when the rows of J2 live on several
 * GPUs (BASELINE configs[3], "8 x MI355X sharded"): every rank reduces ITS rows to an n x n triangle without pivoting,
 * A_rank = Q0 R0 (communication-avoiding QR on the matrix cores, the first stage of the two-stage form above), the root gathers
 * the triangles (n^2 doubles per rank instead of rows_rank * n), stacks them and runs qrk_dense_factorize on the stack; Q^T b
 * follows the same route with one n-vector per rank (qrkit_amd/sharding.py, ShardedBlockAngularQR).  Device memory only, rows >= cols.
 * factorize: a (rows x cols, column-major) <- R0 in the upper triangle of its first cols rows, reflectors of Q0 elsewhere.
 * apply_q:   b (rows x nrhs) <- Q0^T b (transpose != 0) or Q0 b. */
typedef struct qrk_tsqr_plan_s* qrk_tsqr_plan;
qrk_status qrk_tsqr_plan_create(qrk_handle h, int32_t rows, int32_t cols, qrk_tsqr_plan* out);
qrk_status qrk_tsqr_plan_destroy(qrk_tsqr_plan plan);
qrk_status qrk_tsqr_factorize(qrk_tsqr_plan plan, double* a, int64_t lda, qrk_memspace space);
qrk_status qrk_tsqr_apply_q(qrk_tsqr_plan plan, const double* a, int64_t lda, int transpose, double* b, int64_t ldb,
                            int64_t nrhs, qrk_memspace space);

/* ------------------------------------------------------------ block-banded solver */

/* QRKit::BandedBlockedSparseQR<SparseMatrix, HouseholderQR<MatrixXd>, Dynamic, SuggestedBlockCols>
 * (src/QRKit/BandedBlockedSparseQR.h:122-366), generic-pattern path.
 *
 * qrk_bb_plan_create = analyzePattern (:391-433): AsBandedAsPossible row ordering
 * (src/QRKit/SparseQROrdering.h:66-119), band detection and mergeBlocks
 * (src/QRKit/SparseQRUtils.h:186-253,308-385) on the CSR pattern of the matrix (HOST arrays), plus
 * the panel chain of factorize() (:457-508) as device descriptors.  Fails with
 * QRK_STATUS_INVALID_ARGUMENT where the reference itself is outside its domain (its mergeBlocks reads
 * back() of an empty vector for non-portrait strips, SparseQRUtils.h:375). */
typedef struct qrk_bb_plan_s* qrk_bb_plan;

qrk_status qrk_bb_plan_create(qrk_handle h, int32_t rows, int32_t cols, const int32_t* csr_rowptr,
                              const int32_t* csr_colidx, int32_t suggested_block_cols, qrk_bb_plan* out);
qrk_status qrk_bb_plan_destroy(qrk_bb_plan plan);

/* The fixed-pattern path of BandedBlockedSparseQR::analyzePattern (src/QRKit/BandedBlockedSparseQR.h:398-408, taken when the
 * block type is fixed-size and _BlockOverlap != Dynamic): identity row permutation and the block map of
 * BlockBandedMatrixInfo::fromBlockBandedPattern (src/QRKit/SparseQRUtils.h:274-302) + mergeBlocks (:308-385) instead of the
 * row ordering and band detection of qrk_bb_plan_create.  csr_*: pattern of the matrix, as there. */
qrk_status qrk_bb_plan_create_fixed(qrk_handle h, int32_t rows, int32_t cols, const int32_t* csr_rowptr,
                                    const int32_t* csr_colidx, int32_t block_rows, int32_t block_cols, int32_t block_overlap,
                                    int32_t suggested_block_cols, qrk_bb_plan* out);

/* Host-only: the merged block map of that fixed pattern (blocks: 4 ints each, idxRow idxCol numRows numCols; at most cap).
 * Reproduces the reference's known answer test/test-utils.cpp:228-241 (7x4 blocks, overlap 2 -> 255 blocks, the last 14x4). */
qrk_status qrk_bb_blocks_from_pattern(int32_t rows, int32_t cols, int32_t block_rows, int32_t block_cols, int32_t block_overlap,
                                      int32_t suggested_block_cols, int32_t cap, int32_t* num_blocks, int32_t* blocks);

/* The same structure analysis without a device (pure host integer logic): writes up to `cap` blocks as
 * (idxRow, idxCol, numRows, numCols) and, when row_perm != NULL, the row permutation indices;
 * *num_blocks receives the block count.  Lets the reference's known answers
 * (test/test-utils.cpp:199-205,228-241,264-271) be checked anywhere. */
qrk_status qrk_bb_analyze_host(int32_t rows, int32_t cols, const int32_t* csr_rowptr, const int32_t* csr_colidx,
                               int32_t suggested_block_cols, int32_t cap, int32_t* num_blocks, int32_t* blocks,
                               int32_t* row_perm, int32_t* has_row_perm);

/* num_blocks merged blocks; nnz_r entries of m_R (explicit zeros of the emitted rows included, :487-491);
 * y_len / t_len doubles of the implicit Q: per block the panel factorised in place (activeRows x numCols,
 * ROW-major; Y of the reference's BlockYTY, :471-475, is its unit-lower view - the diagonal and what lies
 * above it are to be ignored) and T (numCols x numCols, upper, column-major, NEGATED as the reference stores
 * it, :477); has_row_perm as
 * AsBandedAsPossible::hasPermutation. */
qrk_status qrk_bb_plan_info(qrk_bb_plan plan, int32_t* num_blocks, int64_t* nnz_r, int64_t* y_len,
                            int64_t* t_len, int32_t* has_row_perm);

/* Host outputs: blocks[4*num_blocks] = (idxRow, idxCol, numRows, numCols) of m_blockInfo in order;
 * row_perm[rows] = rowsPermutation().indices(): (P*M).row(row_perm[i]) = M.row(i);
 * yty[6*num_blocks] = per block (rowIndex, numZeros, rows(Y), cols(Y), ... ) see INTEGRATION.md. */
qrk_status qrk_bb_plan_blocks(qrk_bb_plan plan, int32_t* blocks, int32_t* row_perm, int64_t* yty);

/* CSC pattern of m_R (cols+1 pointers, nnz_r row indices). */
qrk_status qrk_bb_pattern(qrk_bb_plan plan, int32_t* r_colptr, int32_t* r_rowidx, qrk_memspace space);

/* factorize() (:443-519): csr_vals are the values of the UNPERMUTED matrix in the CSR order given to
 * qrk_bb_plan_create; r_vals [nnz_r] in CSC order; y_vals [y_len]; t_vals [t_len]. */
qrk_status qrk_bb_factorize(qrk_bb_plan plan, const double* csr_vals, int64_t nnz, double* r_vals,
                            double* y_vals, double* t_vals, qrk_memspace space);

/* v (rows x nrhs, ld = rows) <- Q^T v (transpose != 0; blocks ascending with T^T) or Q v (descending
 * with T): SparseBlockYTY_VecProduct (src/QRKit/SparseBlockYTY.h:100-139). */
qrk_status qrk_bb_apply_q(qrk_bb_plan plan, const double* y_vals, const double* t_vals, int transpose,
                          double* v, int64_t nrhs, qrk_memspace space);

/* v(0:cols, :) <- R(0:cols, 0:cols).triangularView<Upper>().solve(v(0:cols, :)) in place for nrhs columns (leading
 * dimension ldv >= cols): the back substitution that ends BandedBlockedSparseQR::_solve_impl
 * (src/QRKit/BandedBlockedSparseQR.h:290-311, after y = Q^T b).  R is the one of the last qrk_bb_factorize of this
 * plan (the plan keeps its rows on the device). */
qrk_status qrk_bb_solve_r(qrk_bb_plan plan, double* v, int64_t ldv, int64_t nrhs, qrk_memspace space);

/* ------------------------------------------------------------- multi-GPU: shards of the diagonal blocks */

/* One process per GPU.  The hot loop of BlockDiagonalSparseQR::factorize (src/QRKit/BlockDiagonalSparseQR.h:432-526) carries nothing
 * from block to block but the running offsets base_row / base_col (:428-431, :524-525), so the blocks shard as CONTIGUOUS ranges
 * with no data-path collective: rank g factorises blocks [first_block, first_block + num_blocks) with a plan of its own and its
 * Q / R / permutation shards are already in the global order.  No reference site (the reference is single-threaded): SURVEY.md 8(e).
 *
 * qrk_shard_ranges: ranges balanced by the Householder cost r c^2 per block (equal counts for uniform blocks), and the global
 * offsets of each range.  Host-only integer logic: no handle, no GPU.  Uniform layouts pass rows = cols = NULL and block_rows /
 * block_cols.  shards: world + 1 entries; entry world is the end sentinel (first_block = num_blocks, num_blocks = 0, the offsets
 * = the totals), so that the element counts of rank g are differences of consecutive entries. */
typedef struct qrk_shard {
    int64_t first_block, num_blocks;
    int64_t base_row, base_col;      /* rows / columns of the matrix before the range (BlockDiagonalSparseQR.h:428-431) */
    int64_t tiles_off, q_off, r_off; /* offsets of the range in the global tiles / q_vals (tile part) / r_vals arrays */
} qrk_shard;
qrk_status qrk_shard_ranges(int64_t num_blocks, int32_t block_rows, int32_t block_cols, const int32_t* rows, const int32_t* cols,
                            int32_t world, qrk_shard* shards);

/* The only exchange of the path: the composed R (packed CSC values) and the column permutation gathered on `root` with their TRUE
 * counts -- ncclGroupStart / one ncclRecv per peer on the root, one ncclSend on every other rank / ncclGroupEnd, on the handle's
 * stream (RCCL over xGMI; no padding to the largest shard).  nccl_comm: the caller's ncclComm_t of `world` ranks (passed as void*:
 * this header needs no RCCL header; the library resolves ncclSend / ncclRecv / ncclGroupStart / ncclGroupEnd from the RCCL that
 * is already loaded in the process, else from librccl.so); may be NULL when world = 1.  r_local / perm_local: this rank's shards
 * as qrk_bd_factorize of its own plan wrote them (device; permutation indices local to the shard).  r_all [r_off of the sentinel],
 * perm_all [base_col of the sentinel]: device, root only (ignored elsewhere); perm_all holds GLOBAL column indices
 * (m_outputPerm_c.indices(), :519-521: the shard's base_col added). */
qrk_status qrk_gather_r(qrk_handle h, void* nccl_comm, int32_t rank, int32_t world, int32_t root, const qrk_shard* shards,
                        const double* r_local, const int32_t* perm_local, double* r_all, int32_t* perm_all);

/* The sharded SOLVE: _solve_impl (BlockDiagonalSparseQR.h:257-280) is block-local -- x_g = P_g R_g^-1 (Q_g^T b_g)(0:cols_g) needs
 * only the rank's own Q, R and permutation (qrk_bd_solve on the rank's plan) -- so a least-squares consumer never needs the composed
 * R: only x travels, 8 bytes per column and right-hand side (256 B per 32 x 32 tile against 4 352 B of R and permutation).
 * x_local: this rank's solution as qrk_bd_solve wrote it (device; nrhs vectors of cols_g entries, vector k at k * cols_g).  x_all: device,
 * root only: nrhs vectors of the whole matrix's columns, vector k at k * total_cols, rank g's piece at base_col of its shard.  One
 * grouped ncclSend / ncclRecv per peer and right-hand side, true counts, on the handle's stream. */
qrk_status qrk_gather_x(qrk_handle h, void* nccl_comm, int32_t rank, int32_t world, int32_t root, const qrk_shard* shards,
                        const double* x_local, int64_t nrhs, double* x_all);

/* The two exchanges of the ANGULAR solver sharded by rows (the TSQR route of qrk_tsqr_* above; BlockAngularSparseQR.h:361-369, 459-514;
 * host side: qrkit::ShardedBlockAngularSparseQR, qrkit_amd/sharding.py): every rank's `count` doubles to `root` (recv: world * count
 * doubles on the root, piece g at g * count; one n x n triangle, or one n-vector, per rank), and `bytes` bytes at `buf` from `root` to
 * every rank (the permutation of the right block, z2).  Device memory, the handle's stream, grouped ncclSend / ncclRecv on the caller's
 * communicator exactly as qrk_gather_r.  Collective: every rank calls them with the same world / root / count / bytes. */
qrk_status qrk_gather_equal(qrk_handle h, void* nccl_comm, int32_t rank, int32_t world, int32_t root, const double* send, int64_t count,
                            double* recv);
qrk_status qrk_bcast(qrk_handle h, void* nccl_comm, int32_t rank, int32_t world, int32_t root, void* buf, int64_t bytes);

/* ------------------------------------------------------------- measurement */

/* Launch the factorisation kernel(s) of `plan` `iters` times back to back on the
 * handle's stream, rotating over `nsets` consecutive copies of the input/output
 * arrays (set s uses tiles + s*tiles_len, q_vals + s*nnz_q, ...), bracketed by
 * HIP events on that same stream.  *avg_ms = elapsed / iters.  Device memory only. */
qrk_status qrk_bd_time_factorize(qrk_bd_plan plan, const double* tiles, double* q_vals,
                                 double* r_vals, int32_t* perm, int nsets, int iters,
                                 float* avg_ms);

/* Name of the factorisation kernel `plan` launches for its tiles (which = 0: the kernel of the largest size class present),
 * taken from the same dispatch qrk_bd_factorize uses; for reports (bench.py's roofline.kernel).  The pointer is static. */
const char* qrk_bd_kernel_name(qrk_bd_plan plan, int which);


#ifdef __cplusplus
}
#endif
#endif /* QRKIT_AMD_H */
