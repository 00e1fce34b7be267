/*
 * qrk_oracle.h -- CPU ORACLE (test infrastructure, NOT product code).
 *
 * A plain-C restatement of the arithmetic and assembly semantics of the
 * reference hot path (jasvob/QRKit, BlockDiagonalSparseQR and the Eigen dense
 * QR it calls per tile).  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may link, load or call anything in oracle/.  The product
 * (qrkit_amd/, include/) never does, and fails loudly without its HIP library.
 *
 * PARITY PINNING STATUS
 *  - Integer structure logic (block maps, mergeBlocks) is pinned bit-exactly by
 *    the reference's own known-answer tests (test/test-utils.cpp:199-205,
 *    228-241,264-271), reproduced in tests/test_oracle_blockmap.py.
 *  - Floating-point VALUES (Q, R, tau, pivot order on given inputs): PARITY
 *    UNPINNED by the reference.  The per-tile arithmetic lives in Eigen >= 3.3
 *    (CMakeLists.txt:5; un-vendored, unpinned, absent from this image and from
 *    /root/reference), and the reference's tests hold no golden numbers, only
 *    invariants at 1e-6 (test/test-qrkit.cpp:201-203, test/test.h:31).  The
 *    oracle restates Eigen's published algorithms (Householder.h,
 *    ColPivHouseholderQR.h, HouseholderSequence.h, BlockHouseholder.h) and is
 *    cross-checked against LAPACK dgeqp3/dgeqrf (same reflector convention and
 *    LAWN-176 pivot rule) by tests/golden/make_golden.py, plus the reference's
 *    three invariants.
 *
 * All matrices are column-major doubles unless stated otherwise.
 */
#ifndef QRK_ORACLE_H
#define QRK_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Eigen::ComputationInfo values (Eigen/src/Core/util/Constants.h). */
enum { ORC_SUCCESS = 0, ORC_NUMERICAL_ISSUE = 1, ORC_NO_CONVERGENCE = 2, ORC_INVALID_INPUT = 3 };
/* BlockDiagonalSparseQR::MatrixQFormat, BlockDiagonalSparseQR.h:59-62 */
enum { ORC_FULL_Q = 0, ORC_BLOCK_DIAGONAL_Q = 1 };
/* which Eigen dense solver plays _BlockQRSolver */
enum { ORC_COLPIV = 0, ORC_NOPIV = 1 };

/* ---- Eigen dense kernels ------------------------------------------------ */

/* MatrixBase::makeHouseholderInPlace on x[0..len): on return x[0] is untouched
 * (caller stores beta there), x[1..] holds the essential part. */
void orc_make_householder_inplace(double* x, int len, double* tau, double* beta);

/* MatrixBase::applyHouseholderOnTheLeft on the m x n block M (ld = ldm) with
 * essential vector ess[0..m-1) and coefficient tau; work has n doubles. */
void orc_apply_householder_left(double* M, int m, int n, int ldm,
                                const double* ess, double tau, double* work);

/* ColPivHouseholderQR::computeInPlace.  A (m x n, ld = lda) becomes the packed
 * QR; hcoeffs[min(m,n)]; transpositions[min(m,n)] (Eigen's m_colsTranspositions);
 * perm[n] = colsPermutation().indices().  Returns m_nonzero_pivots. */
int orc_colpiv_qr(double* A, int m, int n, int lda, double* hcoeffs,
                  int32_t* transpositions, int32_t* perm, double* maxpivot);

/* HouseholderQR::compute (unblocked form; the blocked form used above 32/48
 * columns is mathematically identical).  perm-free. */
void orc_householder_qr(double* A, int m, int n, int lda, double* hcoeffs);

/* HouseholderSequence::evalTo: dense m x m Q = H_0 ... H_{k-1} from the packed
 * QR (essentials below the diagonal) and hcoeffs; nrefl reflectors. */
void orc_form_q(const double* QR, int m, int nrefl, int ldqr, const double* hcoeffs,
                double* Q, int ldq);

/* internal::make_block_householder_triangular_factor: T (n x n upper, ld = ldt)
 * from unit-lower V (m x n, strictly-lower part read, ld = ldv) and hcoeffs. */
void orc_block_triangular_factor(double* T, int ldt, const double* V, int m, int n,
                                 int ldv, const double* hcoeffs);

/* ---- Block maps (SparseQRUtils.h) --------------------------------------- */

typedef struct { int32_t idxRow, idxCol, numRows, numCols; } orc_block_info;

/* BlockBandedMatrixInfo::fromBlockDiagonalPattern, SparseQRUtils.h:255-272.
 * Returns the number of blocks written (matCols / blockCols). */
int orc_from_block_diagonal_pattern(int32_t matRows, int32_t matCols, int32_t blockRows,
                                    int32_t blockCols, orc_block_info* out, int cap);

/* BlockBandedMatrixInfo::mergeBlocks, SparseQRUtils.h:308-385, on an ordered
 * block list.  Returns the new count, or -1 where the reference would call
 * back() on an empty vector (undefined behaviour there). */
int orc_merge_blocks(const orc_block_info* in, int nin, int maxColStep, int suggestedBlockCols,
                     orc_block_info* out, int cap);

/* BlockBandedMatrixInfo::fromBlockBandedPattern, SparseQRUtils.h:274-302. */
int orc_from_block_banded_pattern(int32_t matRows, int32_t matCols, int32_t blockRows,
                                  int32_t blockCols, int32_t blockOverlap, int suggestedBlockCols,
                                  orc_block_info* out, int cap);

/* BlockBandedMatrixInfo::operator()(rowMajorMat), SparseQRUtils.h:186-253, from
 * CSR structure only (values are irrelevant).  Returns count or -1 (UB case). */
int orc_block_info_from_csr(int32_t rows, int32_t cols, const int32_t* rowptr,
                            const int32_t* colidx, int suggestedBlockCols,
                            orc_block_info* out, int cap);

/* SparseQROrdering::AsBandedAsPossible, SparseQROrdering.h:66-119: stable sort of
 * rows by first-nonzero column.  perm[orig_row] = new_row (Eigen convention:
 * (P*M).row(perm[i]) = M.row(i)).  Returns hasPermutation. */
int orc_as_banded_as_possible(int32_t rows, int32_t cols, const int32_t* rowptr,
                              const int32_t* colidx, int32_t* perm);

/* ---- BlockDiagonalSparseQR (BlockDiagonalSparseQR.h:392-547) ------------- */

/* Tile batch descriptor: B tiles; tile i is rows[i] x cols[i], column-major,
 * starting at tiles[tile_off[i]].  matRows >= sum(rows), matCols == sum(cols). */
typedef struct {
    int64_t B;
    const int32_t* rows;
    const int32_t* cols;
    const int64_t* tile_off;
    int32_t matRows, matCols;
    int q_format;     /* ORC_FULL_Q / ORC_BLOCK_DIAGONAL_Q */
    int block_solver; /* ORC_COLPIV / ORC_NOPIV */
} orc_bd_desc;

/* Sizes of the value arrays: nnzQ = sum r_i^2 + (matRows - sum r_i), nnzR = sum c_i(c_i+1)/2 */
void orc_bd_sizes(const orc_bd_desc* d, int64_t* nnzQ, int64_t* nnzR);

/* CSR pattern of Q (row-major, matRows x matRows) and CSC pattern of R
 * (col-major, matRows x matCols) exactly as factorize() assembles them. */
void orc_bd_pattern(const orc_bd_desc* d, int32_t* q_rowptr, int32_t* q_colidx,
                    int32_t* r_colptr, int32_t* r_rowidx);

/* factorize(): values in the CSR order of Q / CSC order of R, global column
 * permutation indices (int32), optional per-tile hcoeffs (sum c_i doubles, may
 * be NULL).  Returns info; *rank = sum cols (BlockDiagonalSparseQR.h:440). */
int orc_bd_factorize(const orc_bd_desc* d, const double* tiles, double* Q_vals, double* R_vals,
                     int32_t* perm, double* hcoeffs, int64_t* rank);

/* Same work, but assembling through per-element sparse insertion and a triplet
 * sort with the reference's B*r*c dummy triplets (BlockDiagonalSparseQR.h:424,
 * 457-479,536-541): the "faithful assembly" CPU baseline. */
int orc_bd_factorize_faithful(const orc_bd_desc* d, const double* tiles, double* Q_vals,
                              double* R_vals, int32_t* perm, int64_t* rank);

/* _solve_impl (BlockDiagonalSparseQR.h:257-280), FullQ only:
 * x = P * [R(0:rank,0:rank)^-1 (Q^T b)(0:rank)], nrhs right-hand sides,
 * b is matRows x nrhs (ld matRows), x is matCols x nrhs (ld matCols). */
int orc_bd_solve(const orc_bd_desc* d, const double* Q_vals, const double* R_vals,
                 const int32_t* perm, const double* b, int64_t nrhs, double* x);

/* ---- Reference test-input generator (test/test-qrkit.cpp:64-65,101-117) -- */

/* libstdc++ std::default_random_engine (= minstd_rand0, seed 1) +
 * std::uniform_real_distribution<double>(lo, hi). */
typedef struct { uint32_t state; } orc_minstd;
void orc_minstd_seed(orc_minstd* g, uint32_t seed);
uint32_t orc_minstd_next(orc_minstd* g);
double orc_uniform_real(orc_minstd* g, double lo, double hi);

/* generate_block_diagonal_matrix(numParams = 2*numVars, numResiduals = 7*numVars,
 * permuteRows = false) as packed 7x2 column-major tiles (numVars tiles). */
void orc_gen_reference_7x2(int numVars, double* tiles);

/* Fill n doubles with U(lo,hi) from a fresh default_random_engine with the given seed. */
void orc_gen_uniform(uint32_t seed, double lo, double hi, int64_t n, double* out);

#ifdef __cplusplus
}
#endif
#endif /* QRK_ORACLE_H */
