/*
 * qrk_oracle.c -- CPU ORACLE (test infrastructure, NOT product code).
 * See qrk_oracle.h for scope, allowed callers and the parity-pinning status
 * ("parity unpinned" for floating-point values; integer block maps pinned).
 *
 * Every function cites the reference site (relative to /root/reference) or the
 * Eigen routine whose published algorithm it restates.
 * Plain sequential loops, no FMA contraction assumed (build with
 * -ffp-contract=off): rounding is that of a scalar, left-to-right evaluation.
 */
#include "qrk_oracle.h"

#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

/* ======================================================================== */
/* Eigen dense kernels                                                      */
/* ======================================================================== */

/* Eigen/src/Householder/Householder.h, MatrixBase::makeHouseholder (real case)
 * as called by makeHouseholderInPlace from ColPivHouseholderQR.h / HouseholderQR.h
 * (reference call sites: BlockDiagonalSparseQR.h:438 via _BlockQRSolver,
 * BandedBlockedSparseQR.h:468). */
void orc_make_householder_inplace(double* x, int len, double* tau, double* beta)
{
    double tailSqNorm = 0.0;
    for (int i = 1; i < len; ++i) tailSqNorm += x[i] * x[i];
    const double c0 = x[0];
    const double tol = DBL_MIN;
    if (tailSqNorm <= tol) {
        *tau = 0.0;
        *beta = c0;
        for (int i = 1; i < len; ++i) x[i] = 0.0;
    } else {
        double b = sqrt(c0 * c0 + tailSqNorm);
        if (c0 >= 0.0) b = -b;
        const double denom = c0 - b;
        for (int i = 1; i < len; ++i) x[i] = x[i] / denom;
        *tau = (b - c0) / b;
        *beta = b;
    }
}

/* Eigen/src/Householder/Householder.h, MatrixBase::applyHouseholderOnTheLeft. */
void orc_apply_householder_left(double* M, int m, int n, int ldm, const double* ess, double tau,
                                double* work)
{
    if (m == 1) {
        for (int j = 0; j < n; ++j) M[(size_t)j * ldm] *= (1.0 - tau);
        return;
    }
    if (tau == 0.0) return;
    for (int j = 0; j < n; ++j) {
        double* col = M + (size_t)j * ldm;
        double t = 0.0;
        for (int i = 1; i < m; ++i) t += ess[i - 1] * col[i]; /* tmp = essential^T * bottom */
        t += col[0];                                          /* tmp += row(0)            */
        work[j] = t;
    }
    for (int j = 0; j < n; ++j) {
        double* col = M + (size_t)j * ldm;
        col[0] -= tau * work[j];                                         /* row(0) -= tau*tmp */
        for (int i = 1; i < m; ++i) col[i] -= (tau * ess[i - 1]) * work[j]; /* bottom -= tau*ess*tmp */
    }
}

static double col_norm(const double* x, int len)
{
    double s = 0.0;
    for (int i = 0; i < len; ++i) s += x[i] * x[i];
    return sqrt(s);
}

/* Eigen/src/QR/ColPivHouseholderQR.h, ColPivHouseholderQR::computeInPlace
 * (Eigen >= 3.3: norms, LAWN-176 downdate).  Reference call site:
 * blockSolver.compute(block), BlockDiagonalSparseQR.h:437-438 with
 * BlockQRSolver = ColPivHouseholderQRWrapper<...> (test/test-qrkit.cpp:32-38,49-51). */
int orc_colpiv_qr(double* A, int m, int n, int lda, double* hcoeffs, int32_t* transpositions,
                  int32_t* perm, double* maxpivot_out)
{
    const int size = m < n ? m : n;
    double* normUpd = (double*)malloc(sizeof(double) * (size_t)(n > 0 ? n : 1));
    double* normDir = (double*)malloc(sizeof(double) * (size_t)(n > 0 ? n : 1));
    double* work = (double*)malloc(sizeof(double) * (size_t)(n > 0 ? n : 1));
    double maxnorm = 0.0;
    for (int k = 0; k < n; ++k) {
        normDir[k] = col_norm(A + (size_t)k * lda, m);
        normUpd[k] = normDir[k];
        if (k == 0 || normUpd[k] > maxnorm) maxnorm = normUpd[k];
    }
    const double eps = DBL_EPSILON;
    const double threshold_helper = ((maxnorm * eps) * (maxnorm * eps)) / (double)m;
    const double norm_downdate_threshold = sqrt(eps);
    int nonzero_pivots = size;
    double maxpivot = 0.0;

    for (int k = 0; k < size; ++k) {
        /* first maximum of normUpd[k..n) */
        int b = k;
        double best = normUpd[k];
        for (int j = k + 1; j < n; ++j)
            if (normUpd[j] > best) { best = normUpd[j]; b = j; }
        const double biggest_sq = best * best;
        if (nonzero_pivots == size && biggest_sq < threshold_helper * (double)(m - k))
            nonzero_pivots = k;
        transpositions[k] = b;
        if (k != b) {
            double* ck = A + (size_t)k * lda;
            double* cb = A + (size_t)b * lda;
            for (int i = 0; i < m; ++i) { double t = ck[i]; ck[i] = cb[i]; cb[i] = t; }
            double t = normUpd[k]; normUpd[k] = normUpd[b]; normUpd[b] = t;
            t = normDir[k]; normDir[k] = normDir[b]; normDir[b] = t;
        }
        double beta;
        double* ck = A + (size_t)k * lda + k;
        orc_make_householder_inplace(ck, m - k, &hcoeffs[k], &beta);
        ck[0] = beta;
        if (fabs(beta) > maxpivot) maxpivot = fabs(beta);
        if (n - k - 1 > 0)
            orc_apply_householder_left(A + (size_t)(k + 1) * lda + k, m - k, n - k - 1, lda, ck + 1,
                                       hcoeffs[k], work);
        for (int j = k + 1; j < n; ++j) {
            if (normUpd[j] != 0.0) {
                double temp = fabs(A[(size_t)j * lda + k]) / normUpd[j];
                temp = (1.0 + temp) * (1.0 - temp);
                temp = temp < 0.0 ? 0.0 : temp;
                const double ratio = normUpd[j] / normDir[j];
                const double temp2 = temp * (ratio * ratio);
                if (temp2 <= norm_downdate_threshold) {
                    normDir[j] = col_norm(A + (size_t)j * lda + k + 1, m - k - 1);
                    normUpd[j] = normDir[j];
                } else {
                    normUpd[j] *= sqrt(temp);
                }
            }
        }
    }
    /* m_colsPermutation.setIdentity(); applyTranspositionOnTheRight(k, transpositions[k]) */
    for (int j = 0; j < n; ++j) perm[j] = j;
    for (int k = 0; k < size; ++k) {
        int32_t t = perm[k]; perm[k] = perm[transpositions[k]]; perm[transpositions[k]] = t;
    }
    if (maxpivot_out) *maxpivot_out = maxpivot;
    free(normUpd); free(normDir); free(work);
    return nonzero_pivots;
}

/* Eigen/src/QR/HouseholderQR.h, householder_qr_inplace_unblocked (the blocked
 * driver is algebraically the same).  Reference call site:
 * BandedBlockedSparseQR.h:453,468; ellipse_fitting.cpp:153. */
void orc_householder_qr(double* A, int m, int n, int lda, double* hcoeffs)
{
    const int size = m < n ? m : n;
    double* work = (double*)malloc(sizeof(double) * (size_t)(n > 0 ? n : 1));
    for (int k = 0; k < size; ++k) {
        double beta;
        double* ck = A + (size_t)k * lda + k;
        orc_make_householder_inplace(ck, m - k, &hcoeffs[k], &beta);
        ck[0] = beta;
        if (n - k - 1 > 0)
            orc_apply_householder_left(A + (size_t)(k + 1) * lda + k, m - k, n - k - 1, lda, ck + 1,
                                       hcoeffs[k], work);
    }
    free(work);
}

/* Eigen/src/Householder/HouseholderSequence.h, HouseholderSequence::evalTo
 * (OnTheLeft, unblocked): dst = I; for k = nrefl-1..0 apply H_k to the
 * bottom-right (m-k)x(m-k) corner.  Reference call site: Qi = blockSolver.matrixQ(),
 * BlockDiagonalSparseQR.h:446. */
void orc_form_q(const double* QR, int m, int nrefl, int ldqr, const double* hcoeffs, double* Q,
                int ldq)
{
    double* work = (double*)malloc(sizeof(double) * (size_t)(m > 0 ? m : 1));
    for (int j = 0; j < m; ++j)
        for (int i = 0; i < m; ++i) Q[(size_t)j * ldq + i] = (i == j) ? 1.0 : 0.0;
    for (int k = nrefl - 1; k >= 0; --k) {
        const int corner = m - k;
        orc_apply_householder_left(Q + (size_t)k * ldq + k, corner, corner, ldq,
                                   QR + (size_t)k * ldqr + k + 1, hcoeffs[k], work);
    }
    free(work);
}

/* Eigen/src/Householder/BlockHouseholder.h,
 * internal::make_block_householder_triangular_factor (Eigen >= 3.3 form).
 * Reference call site: BandedBlockedSparseQR.h:476, BlockedThinQRBase.h:331. */
void orc_block_triangular_factor(double* T, int ldt, const double* V, int m, int n, int ldv,
                                 const double* h)
{
    for (int j = 0; j < n; ++j)
        for (int i = 0; i < n; ++i) T[(size_t)j * ldt + i] = 0.0;
    for (int i = n - 1; i >= 0; --i) {
        const int rs = m - i - 1; /* rows below i   */
        const int rt = n - i - 1; /* columns after i */
        if (rt > 0) {
            /* T(i, i+1:) = -h_i * V(i+1:, i)^T * V(i+1:, i+1:)  (V unit-lower) */
            for (int c = 0; c < rt; ++c) {
                const int col = i + 1 + c;
                double s = 0.0;
                for (int r = 0; r < rs; ++r) {
                    const int row = i + 1 + r;
                    const double vi = V[(size_t)i * ldv + row];
                    double vc;
                    if (row == col) vc = 1.0;
                    else if (row > col) vc = V[(size_t)col * ldv + row];
                    else vc = 0.0;
                    s += vi * vc;
                }
                T[(size_t)col * ldt + i] = -h[i] * s;
            }
            /* T(i, i+1:) = T(i, i+1:) * T(i+1:, i+1:) (upper triangular), in place */
            for (int j = n - 1; j > i; --j) {
                const double z = T[(size_t)j * ldt + i];
                T[(size_t)j * ldt + i] = z * T[(size_t)j * ldt + j];
                const int nbelow = n - j - 1;
                for (int c = 0; c < nbelow; ++c)
                    T[(size_t)(j + 1 + c) * ldt + i] += z * T[(size_t)(j + 1 + c) * ldt + j];
            }
        }
        T[(size_t)i * ldt + i] = h[i];
    }
}

/* ======================================================================== */
/* Block maps                                                               */
/* ======================================================================== */

/* SparseQRUtils.h:255-272 */
int orc_from_block_diagonal_pattern(int32_t matRows, int32_t matCols, int32_t blockRows,
                                    int32_t blockCols, orc_block_info* out, int cap)
{
    (void)matRows;
    const int32_t numBlocks = matCols / blockCols;
    for (int i = 0; i < numBlocks && i < cap; ++i) {
        out[i].idxRow = i * blockRows;
        out[i].idxCol = i * blockCols;
        out[i].numRows = blockRows;
        out[i].numCols = blockCols;
    }
    return numBlocks;
}

/* SparseQRUtils.h:308-385 */
int orc_merge_blocks(const orc_block_info* in, int nin, int maxColStep, int suggestedBlockCols,
                     orc_block_info* out, int cap)
{
    int nout = 0;
    orc_block_info first = {0, 0, 0, 0};
    int currRows = 0, currCols = 0;
    for (int it = 0; it < nin; ++it) {
        const orc_block_info curr = in[it];
        if (nout > 0) {
            const orc_block_info last = out[nout - 1];
            if (curr.idxCol + curr.numCols <= last.idxCol + last.numCols) {
                out[nout - 1].numRows = last.numRows + curr.numRows; /* :331-333 */
                continue;
            }
        }
        if (first.numRows == 0) {
            first = curr;
            currRows = curr.numRows;
            currCols = curr.numCols;
        } else {
            currRows = curr.idxRow + curr.numRows - first.idxRow;
            currCols = curr.idxCol + curr.numCols - first.idxCol;
        }
        if (currRows > currCols && currCols >= maxColStep && currCols >= suggestedBlockCols) { /* :357 */
            if (nout >= cap) return -2;
            out[nout].idxRow = first.idxRow;
            out[nout].idxCol = first.idxCol;
            out[nout].numRows = currRows;
            out[nout].numCols = currCols;
            ++nout;
            first.idxRow = first.idxCol = first.numRows = first.numCols = 0;
        }
    }
    if (first.numRows != 0) {
        if (currRows > currCols && currCols >= maxColStep && currCols >= suggestedBlockCols) {
            if (nout >= cap) return -2;
            out[nout].idxRow = first.idxRow;
            out[nout].idxCol = first.idxCol;
            out[nout].numRows = currRows;
            out[nout].numCols = currCols;
            ++nout;
        } else {
            if (nout == 0) return -1; /* reference: newBlockOrder.back() on empty vector, :375 */
            const orc_block_info last = out[nout - 1];
            out[nout - 1].numRows = last.numRows + currRows;
            out[nout - 1].numCols = first.idxCol + currCols - last.idxCol;
        }
    }
    return nout;
}

/* SparseQRUtils.h:274-302 */
int orc_from_block_banded_pattern(int32_t matRows, int32_t matCols, int32_t blockRows,
                                  int32_t blockCols, int32_t blockOverlap, int suggestedBlockCols,
                                  orc_block_info* out, int cap)
{
    (void)matRows;
    const int32_t maxColStep = blockCols - blockOverlap;
    const int32_t numBlocks = matCols / maxColStep;
    orc_block_info* tmp = (orc_block_info*)malloc(sizeof(orc_block_info) * (size_t)(numBlocks > 0 ? numBlocks : 1));
    for (int i = 0; i < numBlocks; ++i) {
        tmp[i].idxRow = i * blockRows;
        tmp[i].idxCol = i * maxColStep;
        tmp[i].numRows = blockRows;
        tmp[i].numCols = (i < numBlocks - 1) ? blockCols : blockCols - blockOverlap;
    }
    const int n = orc_merge_blocks(tmp, numBlocks, maxColStep, suggestedBlockCols, out, cap);
    free(tmp);
    return n;
}

/* std::binary_search on an ascending int array */
static int bsearch_i32(const int32_t* a, int n, int32_t key)
{
    int lo = 0, hi = n;
    while (lo < hi) {
        int mid = lo + (hi - lo) / 2;
        if (a[mid] < key) lo = mid + 1; else hi = mid;
    }
    return lo < n && !(key < a[lo]);
}

/* SparseQRUtils.h:186-253 */
int orc_block_info_from_csr(int32_t rows, int32_t cols, const int32_t* rowptr,
                            const int32_t* colidx, int suggestedBlockCols, orc_block_info* out,
                            int cap)
{
    /* bandWidths / bandHeights keyed by start column: dense tables of size cols+1 */
    int32_t* bandW = (int32_t*)calloc((size_t)cols + 1, sizeof(int32_t));
    int32_t* bandH = (int32_t*)calloc((size_t)cols + 1, sizeof(int32_t));
    int32_t* start = (int32_t*)malloc(sizeof(int32_t) * (size_t)(rows > 0 ? rows : 1));
    for (int32_t j = 0; j < rows; ++j) {
        int32_t s = cols, e;
        if (rowptr[j + 1] > rowptr[j]) s = colidx[rowptr[j]];
        e = s;
        if (rowptr[j + 1] > rowptr[j]) e = colidx[rowptr[j + 1] - 1];
        start[j] = s;
        const int32_t bw = e - s + 1;
        if (bandH[s] == 0) bandW[s] = bw; else if (bandW[s] < bw) bandW[s] = bw;
        bandH[s] += 1;
    }
    int32_t maxColStep = 0;
    for (int32_t j = 0; j + 1 < rows; ++j)
        if (start[j + 1] - start[j] > maxColStep) maxColStep = start[j + 1] - start[j];

    orc_block_info* est = (orc_block_info*)malloc(sizeof(orc_block_info) * (size_t)(rows > 0 ? rows : 1));
    int32_t* order = (int32_t*)malloc(sizeof(int32_t) * (size_t)(rows > 0 ? rows : 1));
    int nest = 0;
    for (int32_t r = 0; r < rows; ++r) {
        if (!bsearch_i32(order, nest, start[r])) {
            if (start[r] < cols) {
                order[nest] = start[r];
                est[nest].idxRow = r;
                est[nest].idxCol = start[r];
                est[nest].numRows = bandH[start[r]];
                est[nest].numCols = bandW[start[r]];
                ++nest;
            }
        }
    }
    const int n = orc_merge_blocks(est, nest, maxColStep, suggestedBlockCols, out, cap);
    free(bandW); free(bandH); free(start); free(est); free(order);
    return n;
}

/* SparseQROrdering.h:66-119.  Stable insertion into buckets = std::stable_sort by start. */
int orc_as_banded_as_possible(int32_t rows, int32_t cols, const int32_t* rowptr,
                              const int32_t* colidx, int32_t* perm)
{
    int32_t* start = (int32_t*)malloc(sizeof(int32_t) * (size_t)(rows > 0 ? rows : 1));
    int sorted = 1;
    for (int32_t j = 0; j < rows; ++j) {
        start[j] = (rowptr[j + 1] > rowptr[j]) ? colidx[rowptr[j]] : cols;
        if (j > 0 && start[j] < start[j - 1]) sorted = 0;
    }
    if (sorted) {
        for (int32_t j = 0; j < rows; ++j) perm[j] = j;
    } else {
        /* counting sort by start (stable) */
        int32_t* cnt = (int32_t*)calloc((size_t)cols + 2, sizeof(int32_t));
        for (int32_t j = 0; j < rows; ++j) cnt[start[j] + 1]++;
        for (int32_t c = 0; c <= cols; ++c) cnt[c + 1] += cnt[c];
        for (int32_t j = 0; j < rows; ++j) perm[j] = cnt[start[j]]++;
        free(cnt);
    }
    free(start);
    return !sorted;
}

/* ======================================================================== */
/* BlockDiagonalSparseQR                                                    */
/* ======================================================================== */

void orc_bd_sizes(const orc_bd_desc* d, int64_t* nnzQ, int64_t* nnzR)
{
    int64_t q = 0, r = 0, sumRows = 0;
    for (int64_t i = 0; i < d->B; ++i) {
        q += (int64_t)d->rows[i] * d->rows[i];
        r += (int64_t)d->cols[i] * (d->cols[i] + 1) / 2;
        sumRows += d->rows[i];
    }
    q += (int64_t)d->matRows - sumRows; /* trailing identity rows, BlockDiagonalSparseQR.h:530-533 */
    if (nnzQ) *nnzQ = q;
    if (nnzR) *nnzR = r;
}

/* Patterns of m_Q (RowMajor) and m_R (ColMajor) as assembled at
 * BlockDiagonalSparseQR.h:455-500,530-541. */
void orc_bd_pattern(const orc_bd_desc* d, int32_t* q_rowptr, int32_t* q_colidx, int32_t* r_colptr,
                    int32_t* r_rowidx)
{
    int32_t base_row = 0, base_col = 0, m1 = 0;
    const int32_t N_start = d->matCols;
    int64_t qp = 0, rp = 0;
    for (int64_t i = 0; i < d->B; ++i) {
        const int32_t r = d->rows[i], c = d->cols[i];
        for (int32_t j = 0; j < r; ++j) {
            q_rowptr[base_row + j] = (int32_t)qp;
            if (d->q_format == ORC_FULL_Q) {
                for (int32_t k = 0; k < c; ++k) q_colidx[qp++] = base_col + k;
                for (int32_t k = 0; k < r - c; ++k) q_colidx[qp++] = N_start + m1 + k;
            } else {
                for (int32_t k = 0; k < r; ++k) q_colidx[qp++] = base_row + k;
            }
        }
        for (int32_t k = 0; k < c; ++k) {
            r_colptr[base_col + k] = (int32_t)rp;
            for (int32_t j = 0; j <= k; ++j)
                r_rowidx[rp++] = (d->q_format == ORC_FULL_Q ? base_col : base_row) + j;
        }
        m1 += r - c;
        base_row += r;
        base_col += c;
    }
    for (int32_t i = base_row; i < d->matRows; ++i) {
        q_rowptr[i] = (int32_t)qp;
        q_colidx[qp++] = i;
    }
    q_rowptr[d->matRows] = (int32_t)qp;
    r_colptr[d->matCols] = (int32_t)rp;
}

static int tile_qr(const orc_bd_desc* d, int64_t i, const double* tiles, double* qr, double* hc,
                   int32_t* tr, int32_t* p, double* Qi)
{
    const int32_t r = d->rows[i], c = d->cols[i];
    if (r < c) return ORC_INVALID_INPUT; /* BlockDiagonalSparseQR.h:509-516 */
    memcpy(qr, tiles + d->tile_off[i], sizeof(double) * (size_t)r * c); /* block = mat[i], :434 */
    if (d->block_solver == ORC_COLPIV) {
        orc_colpiv_qr(qr, r, c, r, hc, tr, p, NULL); /* :437-438 */
    } else {
        orc_householder_qr(qr, r, c, r, hc);
        for (int32_t j = 0; j < c; ++j) p[j] = j;
    }
    orc_form_q(qr, r, c, r, hc, Qi, r); /* :446 */
    return ORC_SUCCESS;
}

/* BlockDiagonalSparseQR::factorize, BlockDiagonalSparseQR.h:415-547 (lean assembly:
 * values written straight into their final CSR/CSC slots). */
int orc_bd_factorize(const orc_bd_desc* d, const double* tiles, double* Q_vals, double* R_vals,
                     int32_t* perm, double* hcoeffs, int64_t* rank_out)
{
    int32_t maxr = 1, maxc = 1;
    for (int64_t i = 0; i < d->B; ++i) {
        if (d->rows[i] > maxr) maxr = d->rows[i];
        if (d->cols[i] > maxc) maxc = d->cols[i];
    }
    double* qr = (double*)malloc(sizeof(double) * (size_t)maxr * maxc);
    double* Qi = (double*)malloc(sizeof(double) * (size_t)maxr * maxr);
    double* hc = (double*)malloc(sizeof(double) * (size_t)maxc);
    int32_t* tr = (int32_t*)malloc(sizeof(int32_t) * (size_t)maxc);
    int32_t* p = (int32_t*)malloc(sizeof(int32_t) * (size_t)maxc);

    for (int32_t j = 0; j < d->matCols; ++j) perm[j] = j; /* m_outputPerm_c.setIdentity, :417 */
    int64_t rank = 0, qp = 0, rp = 0;
    int32_t base_row = 0, base_col = 0;
    int info = ORC_SUCCESS;
    for (int64_t i = 0; i < d->B; ++i) {
        const int32_t r = d->rows[i], c = d->cols[i];
        info = tile_qr(d, i, tiles, qr, hc, tr, p, Qi);
        if (info != ORC_SUCCESS) break;
        rank += c; /* :440 */
        /* Q rows: [U | N] in FullQ, whole row in BlockDiagonalQ: both are row j of Qi (:455-492) */
        for (int32_t j = 0; j < r; ++j)
            for (int32_t k = 0; k < r; ++k) Q_vals[qp++] = Qi[(size_t)k * r + j];
        /* R upper triangle, CSC order (:475-479 / :496-500) */
        for (int32_t k = 0; k < c; ++k)
            for (int32_t j = 0; j <= k; ++j) R_vals[rp++] = qr[(size_t)k * r + j];
        for (int32_t j = 0; j < c; ++j) perm[base_col + j] = base_col + p[j]; /* :519-521 */
        if (hcoeffs) memcpy(hcoeffs + base_col, hc, sizeof(double) * (size_t)c);
        base_row += r;
        base_col += c;
    }
    if (info == ORC_SUCCESS)
        for (int32_t i = base_row; i < d->matRows; ++i) Q_vals[qp++] = 1.0; /* :530-533 */
    if (rank_out) *rank_out = rank;
    free(qr); free(Qi); free(hc); free(tr); free(p);
    return info;
}

/* The same factorisation assembled the way the reference does it: one
 * insertBack per Q entry into growing value/index vectors (:457-470), a triplet
 * list pre-sized with B*r*c zero triplets at (0,0) and then appended to (:424,
 * :477), and Eigen's setFromTriplets (count, scatter into the transposed
 * layout, collapse duplicates by summation, transpose back) + makeCompressed. */
typedef struct { int32_t row, col; double val; } orc_triplet;

int orc_bd_factorize_faithful(const orc_bd_desc* d, const double* tiles, double* Q_vals,
                              double* R_vals, int32_t* perm, int64_t* rank_out)
{
    int32_t maxr = 1, maxc = 1;
    for (int64_t i = 0; i < d->B; ++i) {
        if (d->rows[i] > maxr) maxr = d->rows[i];
        if (d->cols[i] > maxc) maxc = d->cols[i];
    }
    double* qr = (double*)malloc(sizeof(double) * (size_t)maxr * maxc);
    double* Qi = (double*)malloc(sizeof(double) * (size_t)maxr * maxr);
    double* hc = (double*)malloc(sizeof(double) * (size_t)maxc);
    int32_t* tr = (int32_t*)malloc(sizeof(int32_t) * (size_t)maxc);
    int32_t* p = (int32_t*)malloc(sizeof(int32_t) * (size_t)maxc);

    for (int32_t j = 0; j < d->matCols; ++j) perm[j] = j;
    /* tripletsR(numBlocks * block.rows() * block.cols()) with block = mat[0] (:423-424) */
    size_t ntrip = (size_t)d->B * (size_t)d->rows[0] * (size_t)d->cols[0], captrip = ntrip + 16;
    orc_triplet* trip = (orc_triplet*)calloc(captrip, sizeof(orc_triplet));
    /* m_Q value/index vectors grown by insertBack */
    size_t qn = 0, qcap = 1024;
    double* qv = (double*)malloc(sizeof(double) * qcap);
    int32_t* qi = (int32_t*)malloc(sizeof(int32_t) * qcap);
    int32_t* qouter = (int32_t*)calloc((size_t)d->matRows + 1, sizeof(int32_t));

    int64_t rank = 0;
    int32_t base_row = 0, base_col = 0, m1 = 0;
    const int32_t N_start = d->matCols;
    int info = ORC_SUCCESS;
    for (int64_t i = 0; i < d->B; ++i) {
        const int32_t r = d->rows[i], c = d->cols[i];
        info = tile_qr(d, i, tiles, qr, hc, tr, p, Qi);
        if (info != ORC_SUCCESS) break;
        rank += c;
        for (int32_t j = 0; j < r; ++j) {
            qouter[base_row + j] = (int32_t)qn; /* startVec */
            for (int32_t k = 0; k < r; ++k) {
                if (qn == qcap) {
                    qcap *= 2;
                    qv = (double*)realloc(qv, sizeof(double) * qcap);
                    qi = (int32_t*)realloc(qi, sizeof(int32_t) * qcap);
                }
                int32_t col;
                if (d->q_format == ORC_FULL_Q) col = k < c ? base_col + k : N_start + m1 + (k - c);
                else col = base_row + k;
                qv[qn] = Qi[(size_t)k * r + j];
                qi[qn] = col;
                ++qn;
            }
        }
        m1 += r - c;
        for (int32_t j = 0; j < c; ++j)
            for (int32_t k = j; k < c; ++k) {
                if (ntrip == captrip) {
                    captrip *= 2;
                    trip = (orc_triplet*)realloc(trip, sizeof(orc_triplet) * captrip);
                }
                trip[ntrip].row = (d->q_format == ORC_FULL_Q ? base_col : base_row) + j;
                trip[ntrip].col = base_col + k;
                trip[ntrip].val = qr[(size_t)k * r + j];
                ++ntrip;
            }
        for (int32_t j = 0; j < c; ++j) perm[base_col + j] = base_col + p[j];
        base_row += r;
        base_col += c;
    }
    if (info == ORC_SUCCESS) {
        for (int32_t i = base_row; i < d->matRows; ++i) {
            qouter[i] = (int32_t)qn;
            if (qn == qcap) {
                qcap *= 2;
                qv = (double*)realloc(qv, sizeof(double) * qcap);
                qi = (int32_t*)realloc(qi, sizeof(int32_t) * qcap);
            }
            qv[qn] = 1.0; qi[qn] = i; ++qn;
        }
        qouter[d->matRows] = (int32_t)qn; /* finalize */
        memcpy(Q_vals, qv, sizeof(double) * qn);

        /* Eigen::internal::set_from_triplets for a ColMajor destination:
         * pass 1 count per row of the RowMajor temporary, pass 2 insert
         * uncompressed, pass 3 collapse duplicates (sum), pass 4 transpose
         * into the ColMajor matrix (count per column + scatter). */
        const int32_t R = d->matRows, C = d->matCols;
        int32_t* wi = (int32_t*)calloc((size_t)R + 1, sizeof(int32_t));
        for (size_t t = 0; t < ntrip; ++t) wi[trip[t].row + 1]++;
        for (int32_t r_ = 0; r_ < R; ++r_) wi[r_ + 1] += wi[r_];
        int32_t* fill = (int32_t*)malloc(sizeof(int32_t) * ((size_t)R + 1));
        memcpy(fill, wi, sizeof(int32_t) * ((size_t)R + 1));
        int32_t* tcol = (int32_t*)malloc(sizeof(int32_t) * (ntrip ? ntrip : 1));
        double* tval = (double*)malloc(sizeof(double) * (ntrip ? ntrip : 1));
        for (size_t t = 0; t < ntrip; ++t) {
            const int32_t pos = fill[trip[t].row]++;
            tcol[pos] = trip[t].col;
            tval[pos] = trip[t].val;
        }
        /* collapseDuplicates */
        int32_t* wmark = (int32_t*)malloc(sizeof(int32_t) * (size_t)(C > 0 ? C : 1));
        for (int32_t c_ = 0; c_ < C; ++c_) wmark[c_] = -1;
        int32_t count = 0;
        int32_t* newouter = (int32_t*)malloc(sizeof(int32_t) * ((size_t)R + 1));
        for (int32_t r_ = 0; r_ < R; ++r_) {
            const int32_t st = count;
            for (int32_t k = wi[r_]; k < wi[r_ + 1]; ++k) {
                const int32_t c_ = tcol[k];
                if (wmark[c_] >= st) {
                    tval[wmark[c_]] += tval[k];
                } else {
                    tval[count] = tval[k];
                    tcol[count] = c_;
                    wmark[c_] = count;
                    ++count;
                }
            }
            newouter[r_] = st;
        }
        newouter[R] = count;
        /* transpose RowMajor temporary into ColMajor m_R */
        int32_t* cptr = (int32_t*)calloc((size_t)C + 1, sizeof(int32_t));
        for (int32_t k = 0; k < count; ++k) cptr[tcol[k] + 1]++;
        for (int32_t c_ = 0; c_ < C; ++c_) cptr[c_ + 1] += cptr[c_];
        int32_t* cfill = (int32_t*)malloc(sizeof(int32_t) * ((size_t)C + 1));
        memcpy(cfill, cptr, sizeof(int32_t) * ((size_t)C + 1));
        for (int32_t r_ = 0; r_ < R; ++r_)
            for (int32_t k = newouter[r_]; k < newouter[r_ + 1]; ++k)
                R_vals[cfill[tcol[k]]++] = tval[k];
        free(wi); free(fill); free(tcol); free(tval); free(wmark); free(newouter); free(cptr); free(cfill);
    }
    if (rank_out) *rank_out = rank;
    free(qr); free(Qi); free(hc); free(tr); free(p); free(trip); free(qv); free(qi); free(qouter);
    return info;
}

/* BlockDiagonalSparseQR::_solve_impl, BlockDiagonalSparseQR.h:257-280 (FullQ).
 * y = Q^T b (sparse Q in the FullQ [U|N] layout), back-substitution with the
 * block upper-triangular R(0:rank,0:rank), dest = colsPermutation * y.topRows(cols). */
int orc_bd_solve(const orc_bd_desc* d, const double* Q_vals, const double* R_vals,
                 const int32_t* perm, const double* b, int64_t nrhs, double* x)
{
    if (d->q_format != ORC_FULL_Q) return ORC_INVALID_INPUT;
    const int32_t M = d->matRows, N = d->matCols;
    double* y = (double*)malloc(sizeof(double) * (size_t)(M > N ? M : N));
    for (int64_t rhs = 0; rhs < nrhs; ++rhs) {
        const double* bb = b + (size_t)rhs * M;
        double* xx = x + (size_t)rhs * N;
        /* y = Q^T b: y[col] = sum_row Q(row,col) b[row] */
        for (int32_t i = 0; i < M; ++i) y[i] = 0.0;
        int64_t qp = 0;
        int32_t base_row = 0, base_col = 0, m1 = 0;
        for (int64_t i = 0; i < d->B; ++i) {
            const int32_t r = d->rows[i], c = d->cols[i];
            for (int32_t j = 0; j < r; ++j)
                for (int32_t k = 0; k < r; ++k) {
                    const int32_t col = k < c ? base_col + k : N + m1 + (k - c);
                    y[col] += Q_vals[qp++] * bb[base_row + j];
                }
            m1 += r - c; base_row += r; base_col += c;
        }
        for (int32_t i = base_row; i < M; ++i) y[i] += Q_vals[qp++] * bb[i];
        /* triangular solve, block by block (R is block upper-triangular) */
        int64_t rp = 0;
        base_col = 0;
        for (int64_t i = 0; i < d->B; ++i) {
            const int32_t c = d->cols[i];
            const double* Rt = R_vals + rp; /* packed upper by columns */
            for (int32_t k = c - 1; k >= 0; --k) {
                const double* colk = Rt + (size_t)k * (k + 1) / 2;
                y[base_col + k] /= colk[k];
                const double yk = y[base_col + k];
                for (int32_t j = 0; j < k; ++j) y[base_col + j] -= colk[j] * yk;
            }
            rp += (int64_t)c * (c + 1) / 2;
            base_col += c;
        }
        /* dest = P * y.topRows(cols): dest[perm[j]] = y[j] */
        for (int32_t j = 0; j < N; ++j) xx[perm[j]] = y[j];
    }
    free(y);
    return ORC_SUCCESS;
}

/* ======================================================================== */
/* Reference test-input generator                                           */
/* ======================================================================== */

/* libstdc++ std::minstd_rand0: x <- 16807 x mod (2^31 - 1); default seed 1. */
void orc_minstd_seed(orc_minstd* g, uint32_t seed)
{
    uint32_t s = seed % 2147483647u;
    g->state = s == 0 ? 1u : s;
}

uint32_t orc_minstd_next(orc_minstd* g)
{
    g->state = (uint32_t)(((uint64_t)g->state * 16807u) % 2147483647u);
    return g->state;
}

/* libstdc++ std::generate_canonical<double,53>(minstd_rand0): range
 * R = max - min + 1 = 2147483646, floor(log2 R) = 30, so m = 2 draws;
 * uniform_real_distribution: canonical * (hi - lo) + lo. */
double orc_uniform_real(orc_minstd* g, double lo, double hi)
{
    const double R = 2147483646.0;
    double sum = 0.0, tmp = 1.0;
    for (int k = 0; k < 2; ++k) {
        sum += (double)(orc_minstd_next(g) - 1u) * tmp;
        tmp *= R;
    }
    double ret = sum / tmp;
    if (ret >= 1.0) ret = nextafter(1.0, 0.0);
    return ret * (hi - lo) + lo;
}

/* test/test-qrkit.cpp:101-117: for i, for j in {2i, 2i+1}: rows 7i..7i+6 at column j,
 * one dist(gen) per entry in that order; tile i = rows 7i.. x cols 2i..2i+1. */
void orc_gen_reference_7x2(int numVars, double* tiles)
{
    orc_minstd g;
    orc_minstd_seed(&g, 1u);
    for (int i = 0; i < numVars; ++i)
        for (int j = 0; j < 2; ++j)
            for (int r = 0; r < 7; ++r) tiles[(size_t)i * 14 + (size_t)j * 7 + r] = orc_uniform_real(&g, 0.5, 5.0);
}

void orc_gen_uniform(uint32_t seed, double lo, double hi, int64_t n, double* out)
{
    orc_minstd g;
    orc_minstd_seed(&g, seed);
    for (int64_t i = 0; i < n; ++i) out[i] = orc_uniform_real(&g, lo, hi);
}
