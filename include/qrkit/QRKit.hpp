// QRKit.hpp -- header-only C++ facade over the C ABI (include/qrkit_amd.h).
//
// Keeps the reference's class and method names for the block-diagonal path so that code written
// against QRKit (src/QRKit/SparseBlockDiagonal.h, src/QRKit/BlockDiagonalSparseQR.h) can switch by
// changing the include and the namespace:
//     QRKit::SparseBlockDiagonal<Block>            ->  qrkit::SparseBlockDiagonal
//     QRKit::BlockDiagonalSparseQR<Solver,QFormat> ->  qrkit::BlockDiagonalSparseQR<SolverTag,QFormat>
// Eigen is not required (it is absent from the build image): the small value types below stand in
// for Eigen::Matrix / SparseMatrix / PermutationMatrix with the same accessor names and conventions.
// All numerical work happens in the HIP library; there is no host fallback.
#ifndef QRKIT_FACADE_HPP
#define QRKIT_FACADE_HPP

#include <algorithm>
#include <cassert>
#include <cmath>
#include <cstdint>
#include <stdexcept>
#include <string>
#include <vector>

#include "../qrkit_amd.h"

namespace qrkit {

typedef std::int64_t Index;
typedef int StorageIndex;

// Eigen::ComputationInfo
enum ComputationInfo { Success = 0, NumericalIssue = 1, NoConvergence = 2, InvalidInput = 3 };

// Column-major dense matrix (Eigen::Matrix<double,Dynamic,Dynamic>).
class Matrix {
  public:
    Matrix() : m_rows(0), m_cols(0) {}
    Matrix(Index r, Index c) : m_rows(r), m_cols(c), m_data((size_t)(r * c), 0.0) {}
    Index rows() const { return m_rows; }
    Index cols() const { return m_cols; }
    double& operator()(Index i, Index j) { return m_data[(size_t)(j * m_rows + i)]; }
    double operator()(Index i, Index j) const { return m_data[(size_t)(j * m_rows + i)]; }
    double* data() { return m_data.data(); }
    const double* data() const { return m_data.data(); }
  private:
    Index m_rows, m_cols;
    std::vector<double> m_data;
};
typedef std::vector<double> Vector;

// Eigen::PermutationMatrix<Dynamic,Dynamic,int>: P(indices[j], j) = 1.
class PermutationMatrix {
  public:
    PermutationMatrix() {}
    explicit PermutationMatrix(Index n) { setIdentity(n); }
    void setIdentity(Index n) { m_indices.resize((size_t)n); for (Index i = 0; i < n; ++i) m_indices[(size_t)i] = (int)i; }
    Index size() const { return (Index)m_indices.size(); }
    Index rows() const { return size(); }
    std::vector<int>& indices() { return m_indices; }
    const std::vector<int>& indices() const { return m_indices; }
    // (P * y)[indices[j]] = y[j]
    Vector operator*(const Vector& y) const {
        Vector out(y.size());
        for (size_t j = 0; j < m_indices.size(); ++j) out[(size_t)m_indices[j]] = y[j];
        return out;
    }
  private:
    std::vector<int> m_indices;
};

struct Triplet { int row, col; double value; Triplet(int r, int c, double v) : row(r), col(c), value(v) {} };

// Compressed sparse matrix, RowMajor (CSR) or ColMajor (CSC) like Eigen::SparseMatrix<double,Major,int>.
template <bool RowMajor>
class SparseMatrix {
  public:
    SparseMatrix() : m_rows(0), m_cols(0) {}
    SparseMatrix(Index r, Index c) : m_rows(r), m_cols(c), m_outer((size_t)((RowMajor ? r : c) + 1), 0) {}
    Index rows() const { return m_rows; }
    Index cols() const { return m_cols; }
    Index outerSize() const { return RowMajor ? m_rows : m_cols; }
    Index nonZeros() const { return (Index)m_values.size(); }
    std::vector<int>& outerIndex() { return m_outer; }
    std::vector<int>& innerIndex() { return m_inner; }
    std::vector<double>& values() { return m_values; }
    const std::vector<int>& outerIndex() const { return m_outer; }
    const std::vector<int>& innerIndex() const { return m_inner; }
    const std::vector<double>& values() const { return m_values; }
    void resize(Index r, Index c) { m_rows = r; m_cols = c; m_outer.assign((size_t)((RowMajor ? r : c) + 1), 0); m_inner.clear(); m_values.clear(); }

    // setFromTriplets: duplicates are summed, entries sorted by inner index.
    void setFromTriplets(const std::vector<Triplet>& t) {
        const Index no = outerSize();
        std::vector<int> cnt((size_t)no + 1, 0);
        for (const Triplet& e : t) cnt[(size_t)(RowMajor ? e.row : e.col) + 1]++;
        for (Index i = 0; i < no; ++i) cnt[(size_t)i + 1] += cnt[(size_t)i];
        std::vector<int> inner(t.size()), fill(cnt.begin(), cnt.end() - 1);
        std::vector<double> val(t.size());
        for (const Triplet& e : t) {
            const int o = RowMajor ? e.row : e.col, p = fill[(size_t)o]++;
            inner[(size_t)p] = RowMajor ? e.col : e.row; val[(size_t)p] = e.value;
        }
        m_outer.assign((size_t)no + 1, 0); m_inner.clear(); m_values.clear();
        for (Index o = 0; o < no; ++o) {
            std::vector<std::pair<int, double> > seg;
            for (int p = cnt[(size_t)o]; p < cnt[(size_t)o + 1]; ++p) seg.push_back(std::make_pair(inner[(size_t)p], val[(size_t)p]));
            std::stable_sort(seg.begin(), seg.end(), [](const std::pair<int, double>& a, const std::pair<int, double>& b) { return a.first < b.first; });
            for (size_t q = 0; q < seg.size(); ++q) {
                if (!m_inner.empty() && (Index)m_inner.size() > m_outer[(size_t)o] && m_inner.back() == seg[q].first) m_values.back() += seg[q].second;
                else { m_inner.push_back(seg[q].first); m_values.push_back(seg[q].second); }
            }
            m_outer[(size_t)o + 1] = (int)m_inner.size();
        }
    }
    double coeff(Index i, Index j) const {
        const Index o = RowMajor ? i : j, in = RowMajor ? j : i;
        for (int p = m_outer[(size_t)o]; p < m_outer[(size_t)o + 1]; ++p) if (m_inner[(size_t)p] == in) return m_values[(size_t)p];
        return 0.0;
    }
    // y = M x and y = M^T x
    Vector operator*(const Vector& x) const {
        Vector y((size_t)m_rows, 0.0);
        for (Index o = 0; o < outerSize(); ++o)
            for (int p = m_outer[(size_t)o]; p < m_outer[(size_t)o + 1]; ++p) {
                if (RowMajor) y[(size_t)o] += m_values[(size_t)p] * x[(size_t)m_inner[(size_t)p]];
                else y[(size_t)m_inner[(size_t)p]] += m_values[(size_t)p] * x[(size_t)o];
            }
        return y;
    }
    Vector transposeTimes(const Vector& x) const {
        Vector y((size_t)m_cols, 0.0);
        for (Index o = 0; o < outerSize(); ++o)
            for (int p = m_outer[(size_t)o]; p < m_outer[(size_t)o + 1]; ++p) {
                if (RowMajor) y[(size_t)m_inner[(size_t)p]] += m_values[(size_t)p] * x[(size_t)o];
                else y[(size_t)o] += m_values[(size_t)p] * x[(size_t)m_inner[(size_t)p]];
            }
        return y;
    }
    Matrix toDense() const {
        Matrix d(m_rows, m_cols);
        for (Index o = 0; o < outerSize(); ++o)
            for (int p = m_outer[(size_t)o]; p < m_outer[(size_t)o + 1]; ++p) {
                if (RowMajor) d(o, m_inner[(size_t)p]) = m_values[(size_t)p]; else d(m_inner[(size_t)p], o) = m_values[(size_t)p];
            }
        return d;
    }
  private:
    Index m_rows, m_cols;
    std::vector<int> m_outer, m_inner;
    std::vector<double> m_values;
};
typedef SparseMatrix<true> SparseMatrixRowMajor;
typedef SparseMatrix<false> SparseMatrixColMajor;

// solve() with a SPARSE right-hand side (the SparseMatrixBase overload of every reference solver, e.g. BlockDiagonalSparseQR.h:293-299;
// Eigen evaluates Solve<Dec, SparseRhs> with internal::solve_sparse_through_dense_panels): the columns of B go through the solver's
// dense path in panels of four, and the result keeps the entries that are not exactly zero (tmpX.sparseView()).  `Panel` = the
// solver's dense solve() takes several columns at once (rows x k, column-major); otherwise one column per call.
namespace detail {
template <bool Panel, typename Solver, bool RM>
inline SparseMatrix<false> solveSparseThroughDensePanels(const Solver& dec, Index decRows, Index decCols, const SparseMatrix<RM>& B) {
    assert(decRows == B.rows() && "SparseQR::solve() : invalid number of rows in the right hand side matrix");
    const Index rhsCols = B.cols(), NbColsAtOnce = 4;
    // column access to B whatever its storage order
    std::vector<std::vector<std::pair<int, double> > > colsOfB((size_t)rhsCols);
    for (Index o = 0; o < B.outerSize(); ++o)
        for (int p = B.outerIndex()[(size_t)o]; p < B.outerIndex()[(size_t)o + 1]; ++p) {
            const int in = B.innerIndex()[(size_t)p];
            if (RM) colsOfB[(size_t)in].push_back(std::make_pair((int)o, B.values()[(size_t)p]));
            else colsOfB[(size_t)o].push_back(std::make_pair(in, B.values()[(size_t)p]));
        }
    SparseMatrix<false> X(decCols, rhsCols);
    std::vector<int>& outer = X.outerIndex();
    for (Index k = 0; k < rhsCols; k += NbColsAtOnce) {
        const Index actualCols = std::min<Index>(rhsCols - k, NbColsAtOnce);
        Vector tmp((size_t)(decRows * actualCols), 0.0), tmpX;
        for (Index c = 0; c < actualCols; ++c)
            for (size_t q = 0; q < colsOfB[(size_t)(k + c)].size(); ++q)
                tmp[(size_t)(c * decRows + colsOfB[(size_t)(k + c)][q].first)] = colsOfB[(size_t)(k + c)][q].second;
        if (Panel) {
            tmpX = dec.solve(tmp);
        } else {
            tmpX.resize((size_t)(decCols * actualCols));
            for (Index c = 0; c < actualCols; ++c) {
                const Vector xc = dec.solve(Vector(tmp.begin() + c * decRows, tmp.begin() + (c + 1) * decRows));
                std::copy(xc.begin(), xc.begin() + decCols, tmpX.begin() + c * decCols);
            }
        }
        for (Index c = 0; c < actualCols; ++c) {
            for (Index i = 0; i < decCols; ++i) {
                const double v = tmpX[(size_t)(c * decCols + i)];
                if (v != 0.0) { X.innerIndex().push_back((int)i); X.values().push_back(v); }
            }
            outer[(size_t)(k + c + 1)] = (int)X.innerIndex().size();
        }
    }
    return X;
}
}  // namespace detail

// QRKit::SparseBlockDiagonal (SparseBlockDiagonal.h:43-163): the blocks are kept packed back to back
// (column-major), which is what std::vector<Matrix<double,r,c>> is for fixed-size blocks.
class SparseBlockDiagonal {
  public:
    SparseBlockDiagonal() : nRows(0), nCols(0) {}
    SparseBlockDiagonal(StorageIndex rows, StorageIndex cols) : nRows(rows), nCols(cols) {}

    // SparseBlockDiagonal.h:71-89 with the block map of SparseQRUtils.h:255-272:
    // numBlocks = matCols / blockCols blocks (i*blockRows, i*blockCols, blockRows, blockCols).
    // The tiles are cut on the device (qrk_bd_tiles_from_sparse); mat is a compressed SparseMatrix<RowMajor?>.
    template <bool RM>
    void fromBlockDiagonalPattern(const SparseMatrix<RM>& mat, StorageIndex blockRows, StorageIndex blockCols, int device = 0) {
        clear();
        nRows = (StorageIndex)mat.rows();
        nCols = (StorageIndex)mat.cols();
        const StorageIndex numBlocks = nCols / blockCols;
        m_rows.assign((size_t)numBlocks, (int32_t)blockRows);
        m_cols.assign((size_t)numBlocks, (int32_t)blockCols);
        m_tiles.assign((size_t)numBlocks * (size_t)blockRows * (size_t)blockCols, 0.0);
        if (numBlocks == 0) return;
        qrk_handle h = 0;
        if (qrk_create(&h, device, 0) != QRK_STATUS_OK) throw std::runtime_error(std::string("qrkit: ") + qrk_last_error(0));
        qrk_bd_layout lay;
        lay.num_blocks = numBlocks; lay.block_rows = blockRows; lay.block_cols = blockCols;
        lay.rows = lay.cols = 0;
        lay.mat_rows = nRows; lay.mat_cols = numBlocks * blockCols;
        qrk_bd_plan plan = 0;
        qrk_status st = qrk_bd_plan_create(h, &lay, QRK_FULL_Q, QRK_COLPIV_HOUSEHOLDER, &plan);
        if (st == QRK_STATUS_OK)
            st = qrk_bd_tiles_from_sparse(plan, RM ? 1 : 0, mat.outerIndex().data(), mat.innerIndex().data(), mat.values().data(),
                                          (int64_t)mat.nonZeros(), m_tiles.data(), QRK_MEM_HOST);
        const std::string msg = st == QRK_STATUS_OK ? std::string() : std::string("qrkit: ") + qrk_last_error(h);
        if (plan) qrk_bd_plan_destroy(plan);
        qrk_destroy(h);
        if (st != QRK_STATUS_OK) throw std::runtime_error(msg);
    }
    void insertBack(const Matrix& elem) {
        m_rows.push_back((int32_t)elem.rows());
        m_cols.push_back((int32_t)elem.cols());
        m_tiles.insert(m_tiles.end(), elem.data(), elem.data() + elem.rows() * elem.cols());
    }
    StorageIndex size() const { return (StorageIndex)m_rows.size(); }
    void clear() { m_rows.clear(); m_cols.clear(); m_tiles.clear(); }
    Matrix operator[](StorageIndex i) const {
        size_t off = 0;
        for (StorageIndex k = 0; k < i; ++k) off += (size_t)m_rows[(size_t)k] * m_cols[(size_t)k];
        Matrix b(m_rows[(size_t)i], m_cols[(size_t)i]);
        for (Index e = 0; e < b.rows() * b.cols(); ++e) b.data()[e] = m_tiles[off + (size_t)e];
        return b;
    }
    StorageIndex rows() const { return nRows; }
    StorageIndex cols() const { return nCols; }
    void setDims(StorageIndex r, StorageIndex c) { nRows = r; nCols = c; }
    const std::vector<int32_t>& blockRows() const { return m_rows; }
    const std::vector<int32_t>& blockCols() const { return m_cols; }
    const std::vector<double>& tiles() const { return m_tiles; }
    // blocks [first, first + count) as a block-diagonal matrix of their own (the shard of one rank: ShardedBlockDiagonalSparseQR)
    SparseBlockDiagonal blockRange(StorageIndex first, StorageIndex count) const {
        SparseBlockDiagonal out;
        size_t off = 0;
        for (StorageIndex k = 0; k < first; ++k) off += (size_t)m_rows[(size_t)k] * m_cols[(size_t)k];
        size_t len = 0; StorageIndex r = 0, c = 0;
        for (StorageIndex k = first; k < first + count; ++k) {
            len += (size_t)m_rows[(size_t)k] * m_cols[(size_t)k]; r += m_rows[(size_t)k]; c += m_cols[(size_t)k];
        }
        out.m_rows.assign(m_rows.begin() + first, m_rows.begin() + first + count);
        out.m_cols.assign(m_cols.begin() + first, m_cols.begin() + first + count);
        out.m_tiles.assign(m_tiles.begin() + (std::ptrdiff_t)off, m_tiles.begin() + (std::ptrdiff_t)(off + len));
        out.nRows = r; out.nCols = c;
        return out;
    }
  protected:
    std::vector<int32_t> m_rows, m_cols;
    std::vector<double> m_tiles;
    StorageIndex nRows, nCols;
};

// tags standing in for the _BlockQRSolver template argument (BlockDiagonalSparseQR.h:37).  Rows/ColsAtCompileTime are what
// BandedBlockedSparseQR consults to choose its fixed-pattern analysis (BandedBlockedSparseQR.h:398): Dynamic (-1) for the
// plain tags, the block shape for the *Fixed ones (the reference's ColPivHouseholderQR<Matrix<double, 7, 2>> etc.).
const int Dynamic = -1;
struct ColPivHouseholderQR { static const int kSolver = QRK_COLPIV_HOUSEHOLDER; enum { RowsAtCompileTime = Dynamic, ColsAtCompileTime = Dynamic }; };
struct HouseholderQR { static const int kSolver = QRK_HOUSEHOLDER; enum { RowsAtCompileTime = Dynamic, ColsAtCompileTime = Dynamic }; };
template <int R, int C> struct ColPivHouseholderQRFixed { static const int kSolver = QRK_COLPIV_HOUSEHOLDER; enum { RowsAtCompileTime = R, ColsAtCompileTime = C }; };
template <int R, int C> struct HouseholderQRFixed { static const int kSolver = QRK_HOUSEHOLDER; enum { RowsAtCompileTime = R, ColsAtCompileTime = C }; };

// QRKit::BlockDiagonalSparseQR<_BlockQRSolver,_QFormat> (BlockDiagonalSparseQR.h:37-335).
template <typename BlockQRSolver = ColPivHouseholderQR, int QFormat = 0>
class BlockDiagonalSparseQR {
  public:
    typedef SparseBlockDiagonal MatrixType;
    typedef SparseMatrixRowMajor MatrixQType;
    typedef SparseMatrixColMajor MatrixRType;
    typedef PermutationMatrix PermutationType;
    enum MatrixQFormat { FullQ = 0, BlockDiagonalQ = 1 };

    explicit BlockDiagonalSparseQR(int device = 0)
        : m_info(Success), m_nonzeropivots(0), m_isInitialized(false), m_analysisIsok(false), m_factorizationIsok(false),
          m_handle(0), m_plan(0) {
        if (qrk_create(&m_handle, device, 0) != QRK_STATUS_OK)
            throw std::runtime_error(std::string("qrkit: ") + qrk_last_error(0));
    }
    explicit BlockDiagonalSparseQR(const MatrixType& mat, int device = 0) : BlockDiagonalSparseQR(device) { compute(mat); }
    ~BlockDiagonalSparseQR() { releaseDevice(); if (m_plan) qrk_bd_plan_destroy(m_plan); if (m_handle) qrk_destroy(m_handle); }
    BlockDiagonalSparseQR(const BlockDiagonalSparseQR&) = delete;
    BlockDiagonalSparseQR& operator=(const BlockDiagonalSparseQR&) = delete;

    // BlockDiagonalSparseQR.h:94-102
    void compute(const MatrixType& mat, const PermutationType& rowPerm = PermutationType(), bool /*forcePatternAlaysis*/ = false) {
        analyzePattern(mat, rowPerm);
        m_isInitialized = false;
        m_factorizationIsok = false;
        factorize(mat);
    }
    // :392-405
    void analyzePattern(const MatrixType& mat, const PermutationType& rowPerm = PermutationType()) {
        if (rowPerm.rows() == 0) m_rowPerm.setIdentity(mat.rows()); else m_rowPerm = rowPerm;
        m_R.resize(mat.rows(), mat.cols());
        if (m_plan) { qrk_bd_plan_destroy(m_plan); m_plan = 0; }
        qrk_bd_layout lay;
        lay.num_blocks = mat.size();
        lay.block_rows = lay.block_cols = 0;
        lay.rows = mat.blockRows().data();
        lay.cols = mat.blockCols().data();
        bool uniform = mat.size() > 0;
        for (StorageIndex i = 1; i < mat.size() && uniform; ++i)
            uniform = mat.blockRows()[(size_t)i] == mat.blockRows()[0] && mat.blockCols()[(size_t)i] == mat.blockCols()[0];
        if (uniform) {   // fixed-size blocks (fromBlockDiagonalPattern): offsets are arithmetic on the device
            lay.block_rows = mat.blockRows()[0]; lay.block_cols = mat.blockCols()[0];
            lay.rows = lay.cols = 0;
        }
        lay.mat_rows = mat.rows();
        lay.mat_cols = mat.cols();
        check(qrk_bd_plan_create(m_handle, &lay, (qrk_q_format)QFormat, (qrk_block_solver)BlockQRSolver::kSolver, &m_plan));
        m_analysisIsok = true;
    }
    // :415-547.  The factors stay ON THE DEVICE (Q and R values, permutation); matrixQ() / matrixR() copy them to the host
    // when they are asked for, and solve() / matrixQ() products run on the resident copies: an LM loop that only
    // solves moves the tiles up and the solution down, nothing else.
    void factorize(const MatrixType& mat) {
        assert(m_analysisIsok && "analyzePattern() should be called first");
        int64_t tl, nq, nr;
        check(qrk_bd_plan_sizes(m_plan, &tl, &nq, &nr));
        const Index rows = mat.rows(), cols = mat.cols();
        m_nq = nq; m_nr = nr;
        m_Q.resize(rows, rows);
        m_R.resize(rows, cols);
        m_Qsynced = m_Rsynced = false;
        m_outputPerm_c.setIdentity(cols);
        reserve(m_dtiles, m_ctiles, tl * (int64_t)sizeof(double));
        reserve(m_dq, m_cq, nq * (int64_t)sizeof(double));
        reserve(m_dr, m_cr, nr * (int64_t)sizeof(double));
        reserve(m_dperm, m_cperm, cols * (int64_t)sizeof(int32_t));
        check(qrk_memcpy(m_handle, m_dtiles, mat.tiles().data(), tl * (int64_t)sizeof(double), 0));
        check(qrk_bd_factorize(m_plan, (const double*)m_dtiles, (double*)m_dq, (double*)m_dr, (int32_t*)m_dperm, 0, QRK_MEM_DEVICE));
        qrk_info info; int64_t rank;
        check(qrk_bd_info(m_plan, &info, &rank));
        m_info = (ComputationInfo)info;
        if (m_info != Success) return;   // :504-516: m_info = InvalidInput; return (nothing was written: the permutation stays identity)
        check(qrk_memcpy(m_handle, m_outputPerm_c.indices().data(), m_dperm, cols * (int64_t)sizeof(int32_t), 1));
        m_nonzeropivots = rank;
        m_isInitialized = true;
        m_factorizationIsok = true;
    }

    Index rows() const { return m_R.rows(); }
    Index cols() const { return m_R.cols(); }
    const MatrixRType& matrixR() const { syncR(); return m_R; }
    MatrixQType matrixQ() const { syncQ(); return m_Q; }   // by value, as the reference (:235-237)
    const PermutationType& colsPermutation() const { assert(m_isInitialized && "Decomposition is not initialized."); return m_outputPerm_c; }
    const PermutationType& rowsPermutation() const { assert(m_isInitialized && "Decomposition is not initialized."); return m_rowPerm; }
    Index rank() const { assert(m_isInitialized && "The factorization should be called first, use compute()"); return m_nonzeropivots; }
    ComputationInfo info() const { return m_info; }

    // :257-280 / :286-299 for a dense right-hand side (rows x nrhs, column-major)
    bool _solve_impl(const Vector& B, Vector& dest) const {
        assert(m_isInitialized && "The factorization should be called first, use compute()");
        assert((Index)B.size() % rows() == 0 && "SparseQR::solve() : invalid number of rows in the right hand side matrix");
        const int64_t nrhs = (int64_t)B.size() / rows();
        dest.assign((size_t)(nrhs * cols()), 0.0);
        DeviceVec b(*this, B), x(*this, dest.size());
        check(qrk_bd_solve(m_plan, (const double*)m_dq, (const double*)m_dr, (const int32_t*)m_dperm, b.ptr(), nrhs, x.ptr(), QRK_MEM_DEVICE));
        x.download(dest);
        m_info = Success;
        return true;
    }
    Vector solve(const Vector& B) const { Vector x; _solve_impl(B, x); return x; }
    // the SparseMatrixBase overload (:293-299)
    template <bool RM> SparseMatrix<false> solve(const SparseMatrix<RM>& B) const { return detail::solveSparseThroughDensePanels<true>(*this, rows(), cols(), B); }
    // the triangular step alone, on the device: z = R(0:cols,0:cols).triangularView<Upper>().solve(y) (:271); y: cols x nrhs
    Vector solveR(const Vector& y) const {
        assert(m_isInitialized && (Index)y.size() % cols() == 0);
        Vector z(y.size());
        DeviceVec dy(*this, y), dz(*this, z.size());
        check(qrk_bd_solve_r(m_plan, (const double*)m_dr, dy.ptr(), (int64_t)y.size() / cols(), dz.ptr(), QRK_MEM_DEVICE));
        dz.download(z);
        return z;
    }
    // matrixQ().transpose() * B on the device (test/test-qrkit.cpp:187)
    Vector applyQt(const Vector& B) const {
        const int64_t nrhs = (int64_t)B.size() / rows();
        Vector y(B.size());
        DeviceVec b(*this, B), dy(*this, y.size());
        check(qrk_bd_apply_qt(m_plan, (const double*)m_dq, b.ptr(), nrhs, dy.ptr(), QRK_MEM_DEVICE));
        dy.download(y);
        return y;
    }
    // matrixQ() * B on the device (the product with the explicit m_Q)
    Vector applyQ(const Vector& B) const {
        const int64_t nrhs = (int64_t)B.size() / rows();
        Vector y(B.size());
        DeviceVec b(*this, B), dy(*this, y.size());
        check(qrk_bd_apply_q(m_plan, (const double*)m_dq, b.ptr(), nrhs, dy.ptr(), QRK_MEM_DEVICE));
        dy.download(y);
        return y;
    }
    // The same three steps on device pointers (rows x nrhs / cols x nrhs, contiguous columns; in and out must not alias); they
    // return when the result is in memory.  BlockAngularSparseQR keeps Q1^T J2 on the device with them.
    void applyQDevice(const double* d_in, int64_t nrhs, double* d_out, bool transpose) const {
        check(transpose ? qrk_bd_apply_qt(m_plan, (const double*)m_dq, d_in, nrhs, d_out, QRK_MEM_DEVICE)
                        : qrk_bd_apply_q(m_plan, (const double*)m_dq, d_in, nrhs, d_out, QRK_MEM_DEVICE));
        check(qrk_synchronize(m_handle));
    }
    void solveRDevice(const double* d_y, int64_t nrhs, double* d_z) const {
        check(qrk_bd_solve_r(m_plan, (const double*)m_dr, d_y, nrhs, d_z, QRK_MEM_DEVICE));
        check(qrk_synchronize(m_handle));
    }

  protected:
    void check(qrk_status st) const {
        if (st != QRK_STATUS_OK) throw std::runtime_error(std::string("qrkit: ") + qrk_last_error(m_handle));
    }
    // scoped device vector of doubles for a right-hand side / a result
    struct DeviceVec {
        const BlockDiagonalSparseQR& s; void* p; size_t n;
        DeviceVec(const BlockDiagonalSparseQR& ss, size_t count) : s(ss), p(0), n(count) {
            s.check(qrk_device_alloc(s.m_handle, (int64_t)(n * sizeof(double)), &p));
        }
        DeviceVec(const BlockDiagonalSparseQR& ss, const Vector& host) : DeviceVec(ss, host.size()) {
            s.check(qrk_memcpy(s.m_handle, p, host.data(), (int64_t)(n * sizeof(double)), 0));
        }
        ~DeviceVec() { if (p) qrk_device_free(s.m_handle, p); }
        double* ptr() const { return (double*)p; }
        void download(Vector& host) const { s.check(qrk_memcpy(s.m_handle, host.data(), p, (int64_t)(n * sizeof(double)), 1)); }
        DeviceVec(const DeviceVec&) = delete;
        DeviceVec& operator=(const DeviceVec&) = delete;
    };
    void reserve(void*& buf, int64_t& cap, int64_t bytes) {
        if (bytes <= cap) return;
        if (buf) { qrk_device_free(m_handle, buf); buf = 0; cap = 0; }
        check(qrk_device_alloc(m_handle, bytes, &buf));
        cap = bytes;
    }
    void releaseDevice() {
        void** bufs[4] = {&m_dtiles, &m_dq, &m_dr, &m_dperm};
        for (void** b : bufs) { if (*b) qrk_device_free(m_handle, *b); *b = 0; }
        m_ctiles = m_cq = m_cr = m_cperm = 0;
    }
    // host copies of the factors, made when first asked for (values from the device, structure from qrk_bd_pattern)
    void syncQ() const {
        if (m_Qsynced || !m_isInitialized) return;
        m_Q.values().assign((size_t)m_nq, 0.0); m_Q.innerIndex().assign((size_t)m_nq, 0);
        std::vector<int> rp((size_t)m_R.cols() + 1, 0), ri((size_t)m_nr, 0);
        check(qrk_bd_pattern(m_plan, m_Q.outerIndex().data(), m_Q.innerIndex().data(), rp.data(), ri.data(), QRK_MEM_HOST));
        check(qrk_memcpy(m_handle, m_Q.values().data(), m_dq, m_nq * (int64_t)sizeof(double), 1));
        m_Qsynced = true;
    }
    void syncR() const {
        if (m_Rsynced || !m_isInitialized) return;
        m_R.values().assign((size_t)m_nr, 0.0); m_R.innerIndex().assign((size_t)m_nr, 0);
        std::vector<int> qp((size_t)m_R.rows() + 1, 0), qi((size_t)m_nq, 0);
        check(qrk_bd_pattern(m_plan, qp.data(), qi.data(), m_R.outerIndex().data(), m_R.innerIndex().data(), QRK_MEM_HOST));
        check(qrk_memcpy(m_handle, m_R.values().data(), m_dr, m_nr * (int64_t)sizeof(double), 1));
        m_Rsynced = true;
    }
    void* m_dtiles = 0; void* m_dq = 0; void* m_dr = 0; void* m_dperm = 0;     // device-resident tiles, factors, permutation
    int64_t m_ctiles = 0, m_cq = 0, m_cr = 0, m_cperm = 0, m_nq = 0, m_nr = 0;
    mutable bool m_Qsynced = false, m_Rsynced = false;
    mutable ComputationInfo m_info;
    mutable MatrixRType m_R;
    mutable MatrixQType m_Q;
    PermutationType m_outputPerm_c;
    PermutationType m_rowPerm;
    Index m_nonzeropivots;
    bool m_isInitialized, m_analysisIsok, m_factorizationIsok;
    qrk_handle m_handle;
    qrk_bd_plan m_plan;
};

// One process per GPU (the reference is single-threaded -- this is the sharded form of the hot loop
// BlockDiagonalSparseQR.h:432-526, which carries nothing from block to block but the running offsets): rank g of `world` factorises
// its contiguous range of the diagonal blocks (qrk_shard_ranges: balanced by r c^2) with the single-GPU solver above -- no
// data-path collective --, and gatherR() composes the packed R values and the global column permutation on the root
// (qrk_gather_r: grouped ncclSend / ncclRecv with the true counts, RCCL over xGMI).  ncclComm: the caller's ncclComm_t of `world`
// ranks (void*: no RCCL header needed here); may be null when world = 1.
template <typename BlockQRSolver = ColPivHouseholderQR, int QFormat = 0>
class ShardedBlockDiagonalSparseQR : public BlockDiagonalSparseQR<BlockQRSolver, QFormat> {
  public:
    typedef BlockDiagonalSparseQR<BlockQRSolver, QFormat> Base;
    ShardedBlockDiagonalSparseQR(int rank, int world, void* ncclComm, int device = 0)
        : Base(device), m_rank(rank), m_world(world), m_comm(ncclComm), m_shards((size_t)world + 1) {
        if (world <= 0 || rank < 0 || rank >= world) throw std::runtime_error("qrkit: ShardedBlockDiagonalSparseQR: bad rank / world");
    }
    // mat: the WHOLE block-diagonal matrix (every rank knows the layout; only the tiles of the rank's own range are read)
    void compute(const SparseBlockDiagonal& mat) {
        const qrk_status st = qrk_shard_ranges((int64_t)mat.size(), 0, 0, mat.blockRows().data(), mat.blockCols().data(), (int32_t)m_world,
                                               m_shards.data());
        if (st != QRK_STATUS_OK) throw std::runtime_error(std::string("qrkit: ") + qrk_last_error(0));
        m_local = mat.blockRange((StorageIndex)shard().first_block, (StorageIndex)shard().num_blocks);
        if (m_local.size() > 0) Base::compute(m_local);
    }
    const qrk_shard& shard() const { return m_shards[(size_t)m_rank]; }
    const std::vector<qrk_shard>& shards() const { return m_shards; }       // world + 1 entries, the last one the end sentinel
    const SparseBlockDiagonal& localMatrix() const { return m_local; }
    // On the root: rValues = the packed CSC values of the whole m_R (tile part), perm = m_outputPerm_c.indices() of the whole matrix
    // (:519-521).  Elsewhere both come back empty.  Collective: every rank calls it.
    void gatherR(int root, Vector& rValues, std::vector<int>& perm) const {
        const qrk_shard& end = m_shards[(size_t)m_world];
        void *dr = 0, *dp = 0;
        rValues.clear(); perm.clear();
        if (m_rank == root) {
            this->check(qrk_device_alloc(this->m_handle, std::max<int64_t>(end.r_off, 1) * (int64_t)sizeof(double), &dr));
            this->check(qrk_device_alloc(this->m_handle, std::max<int64_t>(end.base_col, 1) * (int64_t)sizeof(int32_t), &dp));
        }
        qrk_status st = qrk_gather_r(this->m_handle, m_comm, (int32_t)m_rank, (int32_t)m_world, (int32_t)root, m_shards.data(),
                                     (const double*)this->m_dr, (const int32_t*)this->m_dperm, (double*)dr, (int32_t*)dp);
        if (st == QRK_STATUS_OK && m_rank == root) {
            rValues.assign((size_t)end.r_off, 0.0); perm.assign((size_t)end.base_col, 0);
            if (end.r_off > 0) st = qrk_memcpy(this->m_handle, rValues.data(), dr, end.r_off * (int64_t)sizeof(double), 1);
            if (st == QRK_STATUS_OK && end.base_col > 0) st = qrk_memcpy(this->m_handle, perm.data(), dp, end.base_col * (int64_t)sizeof(int32_t), 1);
        }
        if (st == QRK_STATUS_OK) st = qrk_synchronize(this->m_handle);
        if (dr) qrk_device_free(this->m_handle, dr);
        if (dp) qrk_device_free(this->m_handle, dp);
        this->check(st);
    }
    // The least-squares solve of the whole matrix without the composed R: _solve_impl (BlockDiagonalSparseQR.h:257-280) is block-local,
    // so every rank solves its own range (bLocal: the rows of b that belong to it) and ONLY x is gathered -- 8 bytes per column against
    // the 4 224 + 128 bytes per 32 x 32 tile of gatherR.  On the root: x of the whole matrix; elsewhere empty.  Collective.
    Vector solve(const Vector& bLocal, int root) const {
        const qrk_shard& end = m_shards[(size_t)m_world];
        const int64_t nc = m_shards[(size_t)m_rank + 1].base_col - shard().base_col;
        Vector xl;
        if (m_local.size() > 0) xl = Base::solve(bLocal);
        void *dxl = 0, *dx = 0;
        qrk_status st = qrk_device_alloc(this->m_handle, std::max<int64_t>(nc, 1) * (int64_t)sizeof(double), &dxl);
        if (st == QRK_STATUS_OK && m_rank == root) st = qrk_device_alloc(this->m_handle, std::max<int64_t>(end.base_col, 1) * (int64_t)sizeof(double), &dx);
        if (st == QRK_STATUS_OK && nc > 0) st = qrk_memcpy(this->m_handle, dxl, xl.data(), nc * (int64_t)sizeof(double), 0);
        if (st == QRK_STATUS_OK)
            st = qrk_gather_x(this->m_handle, m_comm, (int32_t)m_rank, (int32_t)m_world, (int32_t)root, m_shards.data(), (const double*)dxl, 1, (double*)dx);
        Vector x;
        if (st == QRK_STATUS_OK && m_rank == root) {
            x.assign((size_t)end.base_col, 0.0);
            if (end.base_col > 0) st = qrk_memcpy(this->m_handle, x.data(), dx, end.base_col * (int64_t)sizeof(double), 1);
        }
        if (st == QRK_STATUS_OK) st = qrk_synchronize(this->m_handle);
        if (dxl) qrk_device_free(this->m_handle, dxl);
        if (dx) qrk_device_free(this->m_handle, dx);
        this->check(st);
        return x;
    }
  protected:
    int m_rank, m_world;
    void* m_comm;
    std::vector<qrk_shard> m_shards;
    SparseBlockDiagonal m_local;
};

// ---------------------------------------------------------------------------------------------
// Helpers shared by the compositions (host side, small).

// CSR copy (sorted column indices) of a sparse matrix in either storage order.
template <bool RowMajor>
inline void toCsr(const SparseMatrix<RowMajor>& m, std::vector<int>& rowptr, std::vector<int>& colidx, std::vector<double>& vals) {
    const Index rows = m.rows();
    rowptr.assign((size_t)rows + 1, 0);
    if (RowMajor) {
        rowptr = m.outerIndex(); colidx = m.innerIndex(); vals = m.values();
        return;
    }
    for (int r : m.innerIndex()) rowptr[(size_t)r + 1]++;
    for (Index i = 0; i < rows; ++i) rowptr[(size_t)i + 1] += rowptr[(size_t)i];
    colidx.assign(m.innerIndex().size(), 0); vals.assign(m.innerIndex().size(), 0.0);
    std::vector<int> fill(rowptr.begin(), rowptr.end() - 1);
    for (Index c = 0; c < m.cols(); ++c)      // columns ascending => column indices sorted inside every row
        for (int p = m.outerIndex()[(size_t)c]; p < m.outerIndex()[(size_t)c + 1]; ++p) {
            const int q = fill[(size_t)m.innerIndex()[(size_t)p]]++;
            colidx[(size_t)q] = (int)c; vals[(size_t)q] = m.values()[(size_t)p];
        }
}

// x = R(0:n,0:n).triangularView<Upper>().solve(y) for a CSC matrix with sorted row indices
// (the step every _solve_impl of the reference ends with).
inline Vector solveUpperCsc(const SparseMatrixColMajor& R, Index n, const Vector& y) {
    Vector x(y.begin(), y.begin() + n);
    for (Index k = n - 1; k >= 0; --k) {
        const int p0 = R.outerIndex()[(size_t)k], p1 = R.outerIndex()[(size_t)k + 1];
        double diag = 0.0;
        for (int p = p0; p < p1; ++p) if (R.innerIndex()[(size_t)p] == k) diag = R.values()[(size_t)p];
        x[(size_t)k] /= diag;
        for (int p = p0; p < p1; ++p) { const int i = R.innerIndex()[(size_t)p]; if (i < k) x[(size_t)i] -= R.values()[(size_t)p] * x[(size_t)k]; }
    }
    return x;
}

// matrixQ() of the compositions: an expression that only supports products, like the reference's
// SparseBlockYTY / BlockAngularSparseQRMatrixQReturnType.  v may hold several columns (rows x nrhs).
template <typename Solver>
class QProduct {
  public:
    QProduct(const Solver& s, bool transposed) : m_s(s), m_t(transposed) {}
    QProduct transpose() const { return QProduct(m_s, !m_t); }
    Vector operator*(const Vector& v) const { return m_t ? m_s.applyQt(v) : m_s.applyQ(v); }
    Matrix operator*(const Matrix& v) const {
        Vector in(v.data(), v.data() + v.rows() * v.cols());
        Vector out = (*this) * in;
        Matrix r(v.rows(), v.cols());
        std::copy(out.begin(), out.end(), r.data());
        return r;
    }
  private:
    const Solver& m_s;
    bool m_t;
};

// ---------------------------------------------------------------------------------------------
// QRKit::BandedBlockedSparseQR<SparseMatrix, HouseholderQR<MatrixXd>, Dynamic, SuggestedBlockCols>
// (BandedBlockedSparseQR.h:122-366), generic-pattern path: as-banded-as-possible row ordering, band
// detection + block merge, sequential chain of dense Householder panels with Q kept as (Y, T) blocks.
// BlockRows / BlockCols / BlockOverlap != Dynamic select the fixed-pattern analysis (:398-408: identity row permutation, block map
// of fromBlockBandedPattern, SparseQRUtils.h:274-302) through qrk_bb_plan_create_fixed.  The reference's parameter list is
// provided by QRKit::BandedBlockedSparseQR at the end of this header.
template <int SuggestedBlockCols = 2, int BlockRows = Dynamic, int BlockCols = Dynamic, int BlockOverlap = Dynamic>
class BandedBlockedSparseQR {
  public:
    typedef SparseMatrixColMajor MatrixRType;
    typedef PermutationMatrix PermutationType;
    typedef QProduct<BandedBlockedSparseQR> MatrixQType;

    explicit BandedBlockedSparseQR(int device = 0)
        : m_info(Success), m_nonzeropivots(0), m_isInitialized(false), m_analysisIsok(false), m_handle(0), m_plan(0), m_rows(0), m_cols(0),
          m_nnzR(0), m_hasRowPermutation(false) {
        if (qrk_create(&m_handle, device, 0) != QRK_STATUS_OK)
            throw std::runtime_error(std::string("qrkit: ") + qrk_last_error(0));
    }
    ~BandedBlockedSparseQR() { releaseDevice(); if (m_plan) qrk_bb_plan_destroy(m_plan); if (m_handle) qrk_destroy(m_handle); }
    BandedBlockedSparseQR(const BandedBlockedSparseQR&) = delete;
    BandedBlockedSparseQR& operator=(const BandedBlockedSparseQR&) = delete;

    // :170-182 -- the pattern analysis is cached unless forced
    template <bool RM>
    void compute(const SparseMatrix<RM>& mat, bool forcePatternAlaysis = false) {
        if (!m_analysisIsok || forcePatternAlaysis) analyzePattern(mat);
        factorize(mat);
    }
    // :391-433
    template <bool RM>
    void analyzePattern(const SparseMatrix<RM>& mat) {
        std::vector<double> vals;
        toCsr(mat, m_rowptr, m_colidx, vals);
        m_rows = mat.rows(); m_cols = mat.cols();
        if (m_plan) { qrk_bb_plan_destroy(m_plan); m_plan = 0; }
        if (BlockRows != Dynamic && BlockCols != Dynamic && BlockOverlap != Dynamic)
            check(qrk_bb_plan_create_fixed(m_handle, (int32_t)m_rows, (int32_t)m_cols, m_rowptr.data(), m_colidx.data(), BlockRows, BlockCols,
                                           BlockOverlap, SuggestedBlockCols, &m_plan));
        else
            check(qrk_bb_plan_create(m_handle, (int32_t)m_rows, (int32_t)m_cols, m_rowptr.data(), m_colidx.data(), SuggestedBlockCols, &m_plan));
        int32_t nb = 0, has = 0; int64_t yl = 0, tl = 0;
        check(qrk_bb_plan_info(m_plan, &nb, &m_nnzR, &yl, &tl, &has));
        m_hasRowPermutation = has != 0;
        m_blocks.assign((size_t)4 * nb, 0);
        m_yty.assign((size_t)6 * nb, 0);
        m_rowPerm.setIdentity(m_rows);
        check(qrk_bb_plan_blocks(m_plan, m_blocks.data(), m_rowPerm.indices().data(), m_yty.data()));
        m_yl = yl; m_tl = tl;
        m_outputPerm_c.setIdentity(m_cols);
        m_analysisIsok = true;
    }
    // :443-519
    template <bool RM>
    void factorize(const SparseMatrix<RM>& mat) {
        assert(m_analysisIsok && "analyzePattern() should be called first");
        std::vector<int> rp, ci; std::vector<double> vals;
        toCsr(mat, rp, ci, vals);
        assert(ci.size() == m_colidx.size() && "the sparsity pattern differs from the analysed one");
        m_R.resize(m_rows, m_cols);
        m_Rsynced = false;
        // the implicit Q (Y, T of every block) and the values of R stay on the device; products and solve() run there
        reserve(m_dvals, m_cvals, (int64_t)(vals.size() * sizeof(double)));
        reserve(m_dr, m_cr, std::max<int64_t>(m_nnzR, 1) * (int64_t)sizeof(double));
        reserve(m_dy, m_cy, std::max<int64_t>(m_yl, 1) * (int64_t)sizeof(double));
        reserve(m_dt, m_ct, std::max<int64_t>(m_tl, 1) * (int64_t)sizeof(double));
        check(qrk_memcpy(m_handle, m_dvals, vals.data(), (int64_t)(vals.size() * sizeof(double)), 0));
        check(qrk_bb_factorize(m_plan, (const double*)m_dvals, (int64_t)vals.size(), (double*)m_dr, (double*)m_dy, (double*)m_dt, QRK_MEM_DEVICE));
        check(qrk_synchronize(m_handle));
        m_nonzeropivots = m_cols;                 // :513 "assuming all cols are nonzero"
        m_isInitialized = true;
        m_info = Success;
    }

    Index rows() const { return m_rows; }
    Index cols() const { return m_cols; }
    Index rank() const { assert(m_isInitialized); return m_nonzeropivots; }
    ComputationInfo info() const { return m_info; }
    const MatrixRType& matrixR() const { syncR(); return m_R; }
    MatrixQType matrixQ() const { return MatrixQType(*this, false); }
    const PermutationType& colsPermutation() const { return m_outputPerm_c; }     // identity (:253-257)
    const PermutationType& rowsPermutation() const { return m_rowPerm; }
    bool hasRowPermutation() const { return m_hasRowPermutation; }
    Index numBlocks() const { return (Index)m_blocks.size() / 4; }
    // (idxRow, idxCol, numRows, numCols) of merged block k -- m_blockInfo
    const int32_t* blockInfo(Index k) const { return &m_blocks[(size_t)4 * k]; }

    // SparseBlockYTY products (SparseBlockYTY.h:100-139); v: rows x nrhs, column-major
    Vector applyQ(const Vector& v) const { return apply(v, 0); }
    Vector applyQt(const Vector& v) const { return apply(v, 1); }

    // _solve_impl (:290-311): y = Q^T B (B already row-permuted by the caller, as in the reference test),
    // x = R(0:rank,0:rank)^-1 y(0:rank); the column permutation is the identity.
    Vector solve(const Vector& B) const {
        assert(m_isInitialized && "The factorization should be called first, use compute()");
        // y = Q^T B, then the back substitution with R, both on the device (qrk_bb_apply_q, qrk_bb_solve_r)
        const int64_t nrhs = (int64_t)B.size() / m_rows;
        Vector y(B.size());
        {
            DeviceBuf d(*this, B);
            check(qrk_bb_apply_q(m_plan, (const double*)m_dy, (const double*)m_dt, 1, d.ptr(), nrhs, QRK_MEM_DEVICE));
            check(qrk_bb_solve_r(m_plan, d.ptr(), (int64_t)m_rows, nrhs, QRK_MEM_DEVICE));
            d.download(y);
        }
        Vector x((size_t)(m_cols * nrhs));
        for (int64_t c = 0; c < nrhs; ++c) std::copy(y.begin() + c * m_rows, y.begin() + c * m_rows + m_cols, x.begin() + c * m_cols);
        return x;
    }
    // the SparseMatrixBase overload (BandedBlockedSparseQR.h: solve(const SparseMatrixBase<Rhs>&))
    template <bool RM2> SparseMatrix<false> solve(const SparseMatrix<RM2>& B) const { return detail::solveSparseThroughDensePanels<true>(*this, rows(), cols(), B); }
    // Q^T v / Q v and the triangular step on device pointers (the ABI works in place: in is copied to out first)
    void applyQDevice(const double* d_in, int64_t nrhs, double* d_out, bool transpose) const {
        check(qrk_memcpy_2d(m_handle, d_out, m_rows * (int64_t)sizeof(double), d_in, m_rows * (int64_t)sizeof(double),
                            m_rows * (int64_t)sizeof(double), nrhs, 2));
        check(qrk_bb_apply_q(m_plan, (const double*)m_dy, (const double*)m_dt, transpose ? 1 : 0, d_out, nrhs, QRK_MEM_DEVICE));
        check(qrk_synchronize(m_handle));
    }
    void solveRDevice(const double* d_y, int64_t nrhs, double* d_z) const {
        check(qrk_memcpy_2d(m_handle, d_z, m_cols * (int64_t)sizeof(double), d_y, m_cols * (int64_t)sizeof(double),
                            m_cols * (int64_t)sizeof(double), nrhs, 2));
        check(qrk_bb_solve_r(m_plan, d_z, (int64_t)m_cols, nrhs, QRK_MEM_DEVICE));
        check(qrk_synchronize(m_handle));
    }
    // the triangular step alone, on the device (qrk_bb_solve_r); y: cols x nrhs
    Vector solveR(const Vector& y) const {
        assert(m_isInitialized && (Index)y.size() % m_cols == 0);
        Vector z(y.size());
        DeviceBuf d(*this, y);
        check(qrk_bb_solve_r(m_plan, d.ptr(), (int64_t)m_cols, (int64_t)y.size() / m_cols, QRK_MEM_DEVICE));
        d.download(z);
        return z;
    }

  protected:
    Vector apply(const Vector& v, int transpose) const {
        assert(m_isInitialized && (Index)v.size() % m_rows == 0);
        Vector out(v.size());
        DeviceBuf d(*this, v);
        check(qrk_bb_apply_q(m_plan, (const double*)m_dy, (const double*)m_dt, transpose, d.ptr(), (int64_t)v.size() / m_rows, QRK_MEM_DEVICE));
        d.download(out);
        return out;
    }
    void check(qrk_status st) const {
        if (st != QRK_STATUS_OK) throw std::runtime_error(std::string("qrkit: ") + qrk_last_error(m_handle));
    }
    struct DeviceBuf {           // scoped device copy of a host vector
        const BandedBlockedSparseQR& s; void* p; size_t n;
        DeviceBuf(const BandedBlockedSparseQR& ss, const Vector& host) : s(ss), p(0), n(host.size()) {
            s.check(qrk_device_alloc(s.m_handle, (int64_t)(n * sizeof(double)), &p));
            s.check(qrk_memcpy(s.m_handle, p, host.data(), (int64_t)(n * sizeof(double)), 0));
        }
        ~DeviceBuf() { if (p) qrk_device_free(s.m_handle, p); }
        double* ptr() const { return (double*)p; }
        void download(Vector& host) const { s.check(qrk_memcpy(s.m_handle, host.data(), p, (int64_t)(n * sizeof(double)), 1)); }
        DeviceBuf(const DeviceBuf&) = delete;
        DeviceBuf& operator=(const DeviceBuf&) = delete;
    };
    void reserve(void*& buf, int64_t& cap, int64_t bytes) {
        if (bytes <= cap) return;
        if (buf) { qrk_device_free(m_handle, buf); buf = 0; cap = 0; }
        check(qrk_device_alloc(m_handle, bytes, &buf));
        cap = bytes;
    }
    void releaseDevice() {
        void** bufs[4] = {&m_dvals, &m_dr, &m_dy, &m_dt};
        for (void** b : bufs) { if (*b) qrk_device_free(m_handle, *b); *b = 0; }
        m_cvals = m_cr = m_cy = m_ct = 0;
    }
    void syncR() const {         // host copy of R when it is asked for
        if (m_Rsynced || !m_isInitialized) return;
        m_R.values().assign((size_t)m_nnzR, 0.0); m_R.innerIndex().assign((size_t)m_nnzR, 0);
        check(qrk_bb_pattern(m_plan, m_R.outerIndex().data(), m_R.innerIndex().data(), QRK_MEM_HOST));
        check(qrk_memcpy(m_handle, m_R.values().data(), m_dr, m_nnzR * (int64_t)sizeof(double), 1));
        m_Rsynced = true;
    }
    void* m_dvals = 0; void* m_dr = 0; void* m_dy = 0; void* m_dt = 0;
    int64_t m_cvals = 0, m_cr = 0, m_cy = 0, m_ct = 0, m_yl = 0, m_tl = 0;
    mutable bool m_Rsynced = false;
    ComputationInfo m_info;
    Index m_nonzeropivots;
    bool m_isInitialized, m_analysisIsok;
    qrk_handle m_handle;
    qrk_bb_plan m_plan;
    Index m_rows, m_cols;
    int64_t m_nnzR;
    bool m_hasRowPermutation;
    std::vector<int> m_rowptr, m_colidx;
    std::vector<int32_t> m_blocks;
    std::vector<int64_t> m_yty;
    mutable MatrixRType m_R;
    PermutationType m_rowPerm, m_outputPerm_c;
};

// The STRIPS FORM of the banded solver behind the same interface (qrk_bbs_*, qrkit_amd/csrc/banded.hip + banded_maps.hip): a block-banded
// matrix whose block row i is a dense BlockRows x BlockCols strip at the columns [i (BlockCols - BlockOverlap), ..) -- the pattern
// BandedBlockedSparseQR's fixed template parameters describe (BandedBlockedSparseQR.h:398-408, fromBlockBandedPattern,
// SparseQRUtils.h:274-302) and BASELINE configs[2]'s.  Every strip is triangularised on all CUs, a chain merges the triangles, solve()
// and the products with Q run every strip at once plus a two-level chain of one small matrix per strip; 64-bit offsets throughout, so
// the 50 000-strip shape the CSR entry cannot index fits (compute() from a SparseMatrix is for matrices the host can hold; factorizeStrips()
// takes the strips directly).  R is the reference's up to the sign of each row (unique up to row signs for a fixed column order); the rows
// of Q^T v are ordered [the cols entries that meet R | per strip: chain residuals, stage-A residuals] (include/qrkit_amd.h).
// Constraints of the entry: BlockRows >= BlockCols, both and the column step multiples of 16, at most 256.
template <int BlockRows, int BlockCols, int BlockOverlap>
class BandedStripsSparseQR {
  public:
    typedef SparseMatrixColMajor MatrixRType;
    typedef PermutationMatrix PermutationType;
    typedef QProduct<BandedStripsSparseQR> MatrixQType;
    enum { ColStep = BlockCols - BlockOverlap };

    explicit BandedStripsSparseQR(int device = 0) : m_info(Success), m_isInitialized(false), m_handle(0), m_plan(0), m_strips(0), m_rows(0), m_cols(0) {
        if (qrk_create(&m_handle, device, 0) != QRK_STATUS_OK)
            throw std::runtime_error(std::string("qrkit: ") + qrk_last_error(0));
    }
    ~BandedStripsSparseQR() { if (m_plan) qrk_bbs_plan_destroy(m_plan); if (m_handle) qrk_destroy(m_handle); }
    BandedStripsSparseQR(const BandedStripsSparseQR&) = delete;
    BandedStripsSparseQR& operator=(const BandedStripsSparseQR&) = delete;

    template <bool RM> void compute(const SparseMatrix<RM>& mat) { analyzePattern(mat); factorize(mat); }
    // the geometry alone decides the plan: rows = N BlockRows, cols = (N - 1) ColStep + BlockCols
    template <bool RM> void analyzePattern(const SparseMatrix<RM>& mat) { plan(mat.rows() / BlockRows, mat.rows(), mat.cols()); }
    // every stored entry must lie inside its strip (InvalidInput otherwise, nothing is factorised)
    template <bool RM> void factorize(const SparseMatrix<RM>& mat) {
        assert(m_plan && mat.rows() == m_rows && mat.cols() == m_cols && "analyzePattern() should be called first");
        std::vector<double> strips((size_t)m_strips * BlockRows * BlockCols, 0.0);
        m_info = Success;
        for (Index o = 0; o < mat.outerSize(); ++o)
            for (int p = mat.outerIndex()[(size_t)o]; p < mat.outerIndex()[(size_t)o + 1]; ++p) {
                const Index r = RM ? o : mat.innerIndex()[(size_t)p], c = RM ? mat.innerIndex()[(size_t)p] : o;
                const Index i = r / BlockRows, lc = c - i * ColStep;
                if (lc < 0 || lc >= BlockCols) { m_info = InvalidInput; m_isInitialized = false; return; }
                strips[(size_t)((i * BlockCols + lc) * BlockRows + (r - i * BlockRows))] = mat.values()[(size_t)p];
            }
        factorizeStrips(strips.data(), m_strips);
    }
    // strips: strip i column-major (BlockRows x BlockCols) at strips + i BlockRows BlockCols, host memory
    void factorizeStrips(const double* strips, Index numStrips) {
        if (!m_plan || numStrips != m_strips) plan(numStrips, numStrips * BlockRows, (numStrips - 1) * ColStep + BlockCols);
        const int64_t bytes = (int64_t)m_strips * BlockRows * BlockCols * (int64_t)sizeof(double);
        Dev d(*this, bytes);
        check(qrk_memcpy(m_handle, d.p, strips, bytes, 0));
        check(qrk_bbs_factorize(m_plan, (const double*)d.p));
        check(qrk_synchronize(m_handle));
        m_Rsynced = false;
        m_isInitialized = true;
        m_info = Success;
    }

    Index rows() const { return m_rows; }
    Index cols() const { return m_cols; }
    Index rank() const { assert(m_isInitialized); return m_cols; }          // (BandedBlockedSparseQR.h:513 "assuming all cols are nonzero")
    ComputationInfo info() const { return m_info; }
    const PermutationType& colsPermutation() const { return m_perm_c; }      // identity, as the reference's
    const PermutationType& rowsPermutation() const { return m_perm_r; }      // identity: the strips are given in band order
    MatrixQType matrixQ() const { return MatrixQType(*this, false); }
    // R as a sparse matrix: strip i emits the rows [i ColStep, ..) over its own columns (dense inside the band, explicit zeros kept
    // as the reference keeps them, BandedBlockedSparseQR.h:484-491)
    const MatrixRType& matrixR() const {
        assert(m_isInitialized);
        if (m_Rsynced) return m_R;
        std::vector<Triplet> t;
        Dev d(*this, (int64_t)BlockCols * BlockCols * (int64_t)sizeof(double));
        std::vector<double> blk((size_t)BlockCols * BlockCols);
        for (Index i = 0; i < m_strips; ++i) {
            const Index solved = i + 1 < m_strips ? ColStep : BlockCols;
            check(qrk_bbs_r_rows(m_plan, (int64_t)i, (double*)d.p));
            check(qrk_memcpy(m_handle, blk.data(), d.p, (int64_t)(solved * BlockCols) * (int64_t)sizeof(double), 1));
            for (Index c = 0; c < BlockCols; ++c)
                for (Index r = 0; r < solved && r <= c; ++r) t.push_back(Triplet((int)(i * ColStep + r), (int)(i * ColStep + c), blk[(size_t)(c * solved + r)]));
        }
        m_R.resize(m_rows, m_cols);
        m_R.setFromTriplets(t);
        m_Rsynced = true;
        return m_R;
    }
    Vector applyQ(const Vector& v) const { return apply(v, 0); }
    Vector applyQt(const Vector& v) const { return apply(v, 1); }
    // _solve_impl (BandedBlockedSparseQR.h:290-311): x = R^-1 (Q^T B)(0:cols); B: rows x nrhs column-major
    Vector solve(const Vector& B) const {
        assert(m_isInitialized && "The factorization should be called first, use compute()");
        const int64_t nrhs = (int64_t)B.size() / m_rows, bytes = (int64_t)(B.size() * sizeof(double));
        Dev b(*this, bytes), x(*this, m_cols * nrhs * (int64_t)sizeof(double)), w(*this, 2 * bytes);
        check(qrk_memcpy(m_handle, b.p, B.data(), bytes, 0));
        check(qrk_bbs_solve(m_plan, (const double*)b.p, (double*)x.p, nrhs, (double*)w.p));
        Vector out((size_t)(m_cols * nrhs));
        check(qrk_memcpy(m_handle, out.data(), x.p, (int64_t)(out.size() * sizeof(double)), 1));
        return out;
    }
    template <bool RM2> SparseMatrix<false> solve(const SparseMatrix<RM2>& B) const { return detail::solveSparseThroughDensePanels<true>(*this, rows(), cols(), B); }

  protected:
    struct Dev {                 // scoped device scratch
        const BandedStripsSparseQR& s; void* p;
        Dev(const BandedStripsSparseQR& ss, int64_t bytes) : s(ss), p(0) { s.check(qrk_device_alloc(s.m_handle, bytes > 0 ? bytes : 8, &p)); }
        ~Dev() { if (p) qrk_device_free(s.m_handle, p); }
        Dev(const Dev&) = delete;
        Dev& operator=(const Dev&) = delete;
    };
    void plan(Index numStrips, Index rows, Index cols) {
        if (m_plan) { qrk_bbs_plan_destroy(m_plan); m_plan = 0; }
        m_isInitialized = false;
        if (numStrips < 1 || rows != numStrips * BlockRows || cols != (numStrips - 1) * ColStep + BlockCols)
            throw std::runtime_error("qrkit: BandedStripsSparseQR: the matrix is not N strips of BlockRows x BlockCols at column step BlockCols - BlockOverlap");
        check(qrk_bbs_plan_create(m_handle, (int64_t)numStrips, BlockRows, BlockCols, ColStep, &m_plan));
        m_strips = numStrips; m_rows = rows; m_cols = cols;
        m_perm_c.setIdentity(m_cols); m_perm_r.setIdentity(m_rows);
        m_Rsynced = false;
    }
    Vector apply(const Vector& v, int transpose) const {
        assert(m_isInitialized && (Index)v.size() % m_rows == 0);
        const int64_t nrhs = (int64_t)v.size() / m_rows, bytes = (int64_t)(v.size() * sizeof(double));
        Dev in(*this, bytes), out(*this, bytes), w(*this, bytes);
        check(qrk_memcpy(m_handle, in.p, v.data(), bytes, 0));
        check(qrk_bbs_apply_q(m_plan, transpose, (const double*)in.p, (double*)out.p, nrhs, (double*)w.p));
        Vector r(v.size());
        check(qrk_memcpy(m_handle, r.data(), out.p, bytes, 1));
        return r;
    }
    void check(qrk_status st) const {
        if (st != QRK_STATUS_OK) throw std::runtime_error(std::string("qrkit: ") + qrk_last_error(m_handle));
    }
    ComputationInfo m_info;
    bool m_isInitialized;
    qrk_handle m_handle;
    qrk_bbs_plan m_plan;
    Index m_strips, m_rows, m_cols;
    mutable MatrixRType m_R;
    mutable bool m_Rsynced = false;
    PermutationType m_perm_c, m_perm_r;
};

// Dense Householder QR kept on the device (qrk_dense_*): the packed QR and tau stay resident after factorize(), products
// with Q and the triangular solve move only their vectors.  Shared by the thin solvers and the angular right block.
class DenseDeviceQR {
  public:
    DenseDeviceQR() : m_h(0), m_plan(0), m_dqr(0), m_dhc(0), m_dperm(0), m_rows(0), m_cols(0), m_solver(-1) {}
    ~DenseDeviceQR() { release(); }
    DenseDeviceQR(const DenseDeviceQR&) = delete;
    DenseDeviceQR& operator=(const DenseDeviceQR&) = delete;
    // a: in the matrix, out the packed QR (R above the diagonal, essential parts below); hc: tau; perm: column permutation
    void factorize(qrk_handle h, Matrix& a, int solver, std::vector<double>& hc, std::vector<int32_t>& perm) {
        release();
        m_h = h; m_rows = a.rows(); m_cols = a.cols(); m_solver = solver;
        const Index k = std::min(m_rows, m_cols);
        check(qrk_dense_plan_create(m_h, (int32_t)m_rows, (int32_t)m_cols, (qrk_block_solver)solver, &m_plan));
        hc.assign((size_t)std::max<Index>(k, 1), 0.0);
        perm.assign((size_t)std::max<Index>(m_cols, 1), 0);
        void* dperm = 0;
        const int64_t nb = (int64_t)m_rows * m_cols * (int64_t)sizeof(double);
        check(qrk_device_alloc(m_h, std::max<int64_t>(nb, 8), &m_dqr));
        check(qrk_device_alloc(m_h, (int64_t)(hc.size() * sizeof(double)), &m_dhc));
        check(qrk_device_alloc(m_h, (int64_t)(perm.size() * sizeof(int32_t)), &dperm));
        check(qrk_memcpy(m_h, m_dqr, a.data(), nb, 0));
        qrk_status st = qrk_dense_factorize(m_plan, (double*)m_dqr, m_rows, (double*)m_dhc, (int32_t*)dperm, QRK_MEM_DEVICE);
        if (st == QRK_STATUS_OK) st = qrk_memcpy(m_h, a.data(), m_dqr, nb, 1);
        if (st == QRK_STATUS_OK) st = qrk_memcpy(m_h, hc.data(), m_dhc, (int64_t)(hc.size() * sizeof(double)), 1);
        if (st == QRK_STATUS_OK) st = qrk_memcpy(m_h, perm.data(), dperm, (int64_t)(perm.size() * sizeof(int32_t)), 1);
        qrk_device_free(m_h, dperm);
        check(st);
    }
    // the same for a matrix that is already on the device (rows x cols, leading dimension rows): the buffer becomes the solver's own
    // (freed by release()); only tau and the permutation travel to the host
    void factorizeDevice(qrk_handle h, void* d_a, Index rows, Index cols, int solver, std::vector<double>& hc, std::vector<int32_t>& perm) {
        // (the plan - workspaces of the size of the matrix, streams, events - and the small buffers are kept from one factorisation
        //  to the next when the shape is the same: an LM loop calls compute() with one shape)
        const bool same = m_plan && m_h == h && m_rows == rows && m_cols == cols && m_solver == solver;
        if (same) { if (m_dqr) qrk_device_free(m_h, m_dqr); }
        else release();
        m_h = h; m_rows = rows; m_cols = cols; m_dqr = d_a; m_solver = solver;
        const Index k = std::min(m_rows, m_cols);
        hc.assign((size_t)std::max<Index>(k, 1), 0.0);
        perm.assign((size_t)std::max<Index>(m_cols, 1), 0);
        if (!same) {
            check(qrk_dense_plan_create(m_h, (int32_t)m_rows, (int32_t)m_cols, (qrk_block_solver)solver, &m_plan));
            check(qrk_device_alloc(m_h, (int64_t)(hc.size() * sizeof(double)), &m_dhc));
            check(qrk_device_alloc(m_h, (int64_t)(perm.size() * sizeof(int32_t)), &m_dperm));
        }
        qrk_status st = qrk_dense_factorize(m_plan, (double*)m_dqr, m_rows, (double*)m_dhc, (int32_t*)m_dperm, QRK_MEM_DEVICE);
        if (st == QRK_STATUS_OK) st = qrk_memcpy(m_h, hc.data(), m_dhc, (int64_t)(hc.size() * sizeof(double)), 1);
        if (st == QRK_STATUS_OK) st = qrk_memcpy(m_h, perm.data(), m_dperm, (int64_t)(perm.size() * sizeof(int32_t)), 1);
        check(st);
    }
    // a device buffer for a rows x cols matrix to hand to factorizeDevice: the one of the last factorisation when the shape is the same
    void* acquireBuffer(qrk_handle h, Index rows, Index cols) {
        if (m_dqr && m_h == h && m_rows == rows && m_cols == cols) { void* p = m_dqr; m_dqr = 0; return p; }
        void* p = 0;
        if (qrk_device_alloc(h, std::max<int64_t>((int64_t)rows * cols * (int64_t)sizeof(double), 8), &p) != QRK_STATUS_OK)
            throw std::runtime_error(std::string("qrkit: ") + qrk_last_error(h));
        return p;
    }
    const double* packedDevice() const { return (const double*)m_dqr; }      // R in the upper triangle of its first cols rows
    const int32_t* permDevice() const { return (const int32_t*)m_dperm; }     // (after factorizeDevice)
    void applyQDevice(double* d_v, Index nrhs, bool transpose) const {
        check(qrk_dense_apply_q(m_plan, (const double*)m_dqr, m_rows, (const double*)m_dhc, transpose ? 1 : 0, d_v, m_rows, nrhs, QRK_MEM_DEVICE));
        check(qrk_synchronize(m_h));
    }
    void solveRDevice(double* d_z, Index ldb, Index nrhs) const {
        check(qrk_dense_solve_r(m_plan, (const double*)m_dqr, m_rows, d_z, ldb, nrhs, QRK_MEM_DEVICE));
        check(qrk_synchronize(m_h));
    }
    // v (rows x nrhs, column-major) <- Q^T v or Q v
    void applyQ(Vector& v, Index nrhs, bool transpose) const {
        Scoped d(*this, v);
        check(qrk_dense_apply_q(m_plan, (const double*)m_dqr, m_rows, (const double*)m_dhc, transpose ? 1 : 0, d.ptr(), m_rows, nrhs, QRK_MEM_DEVICE));
        d.download(v);
    }
    // z (ldb x nrhs; the first cols rows of every column) <- R^-1 z
    void solveR(Vector& z, Index ldb, Index nrhs) const {
        Scoped d(*this, z);
        check(qrk_dense_solve_r(m_plan, (const double*)m_dqr, m_rows, d.ptr(), ldb, nrhs, QRK_MEM_DEVICE));
        d.download(z);
    }
    bool ready() const { return m_plan != 0; }
    void release() {             // (before the handle it was created with is destroyed)
        if (m_dqr) qrk_device_free(m_h, m_dqr);
        if (m_dhc) qrk_device_free(m_h, m_dhc);
        if (m_dperm) qrk_device_free(m_h, m_dperm);
        if (m_plan) qrk_dense_plan_destroy(m_plan);
        m_dqr = m_dhc = m_dperm = 0; m_plan = 0;
    }
  private:
    struct Scoped {
        const DenseDeviceQR& s; void* p; size_t n;
        Scoped(const DenseDeviceQR& ss, const Vector& host) : s(ss), p(0), n(host.size()) {
            s.check(qrk_device_alloc(s.m_h, (int64_t)(std::max<size_t>(n, 1) * sizeof(double)), &p));
            s.check(qrk_memcpy(s.m_h, p, host.data(), (int64_t)(n * sizeof(double)), 0));
        }
        ~Scoped() { if (p) qrk_device_free(s.m_h, p); }
        double* ptr() const { return (double*)p; }
        void download(Vector& host) const { s.check(qrk_memcpy(s.m_h, host.data(), p, (int64_t)(n * sizeof(double)), 1)); }
        Scoped(const Scoped&) = delete;
        Scoped& operator=(const Scoped&) = delete;
    };
    void check(qrk_status st) const {
        if (st != QRK_STATUS_OK) throw std::runtime_error(std::string("qrkit: ") + qrk_last_error(m_h));
    }
    qrk_handle m_h;
    qrk_dense_plan m_plan;
    void* m_dqr; void* m_dhc; void* m_dperm;
    Index m_rows, m_cols;
    int m_solver;
};

// ---------------------------------------------------------------------------------------------
// QRKit::BlockedThinDenseQR<MatrixXd, HouseholderQR<MatrixXd>, SuggestedBlockCols> (BlockedThinDenseQR.h:53-176,
// base BlockedThinQRBase.h:47-333) and QRKit::BlockedThinSparseQR (BlockedThinSparseQR.h:105-283 -- the same
// factorisation fed from a sparse matrix): Householder QR of a thin dense matrix, no column pivoting, identity
// row permutation.  The reference walks panels of SuggestedBlockCols columns (HouseholderQR of the panel, Y/T of
// the panel, block-reflector update of the columns on the right, :152-173); the reflectors of that chain are
// those of one HouseholderQR of the whole matrix, which is what the device computes (qrk_dense_factorize with
// QRK_HOUSEHOLDER), Q stays implicit (qrk_dense_apply_q) as in the reference.  The class is also the
// RightSolver argument of BlockAngularSparseQR (test/test-qrkit.cpp:294-362).
template <int SuggestedBlockCols = 2>
class BlockedThinDenseQR {
  public:
    static const int kSolver = QRK_HOUSEHOLDER;
    typedef Matrix MatrixType;
    typedef Matrix MatrixRType;
    typedef PermutationMatrix PermutationType;
    typedef QProduct<BlockedThinDenseQR> MatrixQType;

    explicit BlockedThinDenseQR(int device = 0)
        : m_info(Success), m_nonzeroPivots(0), m_isInitialized(false), m_handle(0) {
        if (qrk_create(&m_handle, device, 0) != QRK_STATUS_OK)
            throw std::runtime_error(std::string("qrkit: ") + qrk_last_error(0));
    }
    explicit BlockedThinDenseQR(const Matrix& mat, int device = 0) : BlockedThinDenseQR(device) { compute(mat); }
    ~BlockedThinDenseQR() { m_dev.release(); if (m_handle) qrk_destroy(m_handle); }
    BlockedThinDenseQR(const BlockedThinDenseQR&) = delete;
    BlockedThinDenseQR& operator=(const BlockedThinDenseQR&) = delete;

    // BlockedThinDenseQR.h:104-136
    void compute(const Matrix& mat) {
        m_isInitialized = false;
        analyzePattern(mat);
        m_qr = mat;
        const Index rows = mat.rows(), cols = mat.cols(), k = std::min(rows, cols);
        std::vector<int32_t> p;
        m_dev.factorize(m_handle, m_qr, QRK_HOUSEHOLDER, m_hc, p);      // packed QR and tau stay on the device
        m_R = Matrix(rows, cols);
        for (Index c = 0; c < cols; ++c) for (Index r = 0; r <= std::min(c, k - 1); ++r) m_R(r, c) = m_qr(r, c);
        m_nonzeroPivots = cols;     // (:132)
        m_isInitialized = true;
        m_info = Success;
    }
    template <bool RM>
    void compute(const SparseMatrix<RM>& mat) { compute(mat.toDense()); }     // BlockedThinSparseQR: same chain on mat
    // :137-145: no column permutation, no row permutation
    void analyzePattern(const Matrix& mat) { m_outputPerm_c.setIdentity(mat.cols()); m_rowPerm.setIdentity(mat.rows()); }

    Index rows() const { return m_qr.rows(); }
    Index cols() const { return m_qr.cols(); }
    Index rank() const { assert(m_isInitialized && "The factorization should be called first, use compute()"); return m_nonzeroPivots; }
    ComputationInfo info() const { return m_info; }
    const MatrixRType& matrixR() const { return m_R; }
    MatrixQType matrixQ() const { return MatrixQType(*this, false); }
    const PermutationType& colsPermutation() const { return m_outputPerm_c; }
    const PermutationType& rowsPermutation() const { return m_rowPerm; }
    const std::vector<double>& hCoeffs() const { return m_hc; }

    Vector applyQt(const Vector& v) const { return applyQImpl(v, true); }
    Vector applyQ(const Vector& v) const { return applyQImpl(v, false); }
    // BlockedThinQRBase::_solve_impl (BlockedThinQRBase.h:223-247): x = R(0:rank,0:rank)^-1 (Q^T b)(0:rank)
    Vector solve(const Vector& b) const {
        assert(m_isInitialized && "The factorization should be called first, use compute()");
        Vector y = applyQt(b);
        Vector z(y.begin(), y.begin() + cols());
        m_dev.solveR(z, cols(), 1);
        return z;
    }
    // the SparseMatrixBase overload (BlockedThinQRBase.h: solve(const SparseMatrixBase<Rhs>&))
    template <bool RM2> SparseMatrix<false> solve(const SparseMatrix<RM2>& B) const { return detail::solveSparseThroughDensePanels<false>(*this, rows(), cols(), B); }

  protected:
    Vector applyQImpl(const Vector& v, bool transpose) const {
        assert(m_isInitialized && "The factorization should be called first, use compute()");
        const Index r = rows(), nrhs = (Index)v.size() / r;
        Vector out(v);
        m_dev.applyQ(out, nrhs, transpose);
        return out;
    }
    void check(qrk_status st) const {
        if (st != QRK_STATUS_OK) throw std::runtime_error(std::string("qrkit: ") + qrk_last_error(m_handle));
    }
    ComputationInfo m_info;
    Index m_nonzeroPivots;
    bool m_isInitialized;
    qrk_handle m_handle;
    DenseDeviceQR m_dev;
    Matrix m_qr, m_R;
    std::vector<double> m_hc;
    PermutationType m_outputPerm_c, m_rowPerm;
};
// ---------------------------------------------------------------------------------------------
// QRKit::BlockedThinSparseQR<MatrixType, SuggestedBlockCols> (BlockedThinSparseQR.h:105-283): column-pivoted QR of a thin SPARSE
// matrix, panel by panel, with the ColumnDensity column ordering and the as-banded-as-possible row ordering (analyzePattern,
// :168-201), per-panel ColPivHouseholderQR and the nonzero / zero pivot column bookkeeping (:250-256).  compute() hands the CSC
// arrays to qrk_thin_sparse_factorize (include/qrkit_amd.h), which runs that chain on the device; Q stays implicit.
// As the RightSolver tag of BlockAngularSparseQR (test/test-qrkit.cpp:335-362) it selects the un-pivoted dense solver for the
// bottom block, whose input there is the dense product Q1^T J2 (kSolver).
template <int SuggestedBlockCols = 2>
class BlockedThinSparseQR {
  public:
    static const int kSolver = QRK_HOUSEHOLDER;
    typedef SparseMatrixColMajor MatrixType;
    typedef Matrix MatrixRType;
    typedef PermutationMatrix PermutationType;
    typedef QProduct<BlockedThinSparseQR> MatrixQType;

    explicit BlockedThinSparseQR(int device = 0)
        : m_info(Success), m_nonzeroPivots(0), m_isInitialized(false), m_handle(0), m_plan(0), m_rows(0), m_cols(0), m_Rbuilt(false) {
        if (qrk_create(&m_handle, device, 0) != QRK_STATUS_OK)
            throw std::runtime_error(std::string("qrkit: ") + qrk_last_error(0));
    }
    template <bool RM>
    explicit BlockedThinSparseQR(const SparseMatrix<RM>& mat, int device = 0) : BlockedThinSparseQR(device) { compute(mat); }
    ~BlockedThinSparseQR() { if (m_plan) qrk_thin_destroy(m_plan); if (m_handle) qrk_destroy(m_handle); }
    BlockedThinSparseQR(const BlockedThinSparseQR&) = delete;
    BlockedThinSparseQR& operator=(const BlockedThinSparseQR&) = delete;

    // :105-165 (analyzePattern + factorize)
    void compute(const SparseMatrixColMajor& mat) {
        m_isInitialized = false; m_Rbuilt = false;
        if (m_plan) { qrk_thin_destroy(m_plan); m_plan = 0; }
        m_rows = mat.rows(); m_cols = mat.cols();
        check(qrk_thin_sparse_factorize(m_handle, (int32_t)m_rows, (int32_t)m_cols, SuggestedBlockCols, mat.outerIndex().data(),
                                        mat.innerIndex().data(), mat.values().data(), &m_plan));
        int32_t rk = 0;
        std::vector<int32_t> cp((size_t)m_cols), rp((size_t)m_rows);
        check(qrk_thin_info(m_plan, &rk, cp.data(), rp.data()));
        m_outputPerm_c.setIdentity(m_cols); m_rowPerm.setIdentity(m_rows);
        for (Index j = 0; j < m_cols; ++j) m_outputPerm_c.indices()[(size_t)j] = cp[(size_t)j];
        for (Index i = 0; i < m_rows; ++i) m_rowPerm.indices()[(size_t)i] = rp[(size_t)i];
        m_nonzeroPivots = rk;
        m_info = Success;
        m_isInitialized = true;
    }
    void compute(const SparseMatrixRowMajor& mat) {
        // (the reference's MatrixType is column-major; a row-major input is converted, as Eigen's assignment would)
        std::vector<Triplet> t;
        for (Index r = 0; r < mat.rows(); ++r)
            for (int e = mat.outerIndex()[(size_t)r]; e < mat.outerIndex()[(size_t)r + 1]; ++e) t.push_back(Triplet((int)r, mat.innerIndex()[(size_t)e], mat.values()[(size_t)e]));
        SparseMatrixColMajor cm(mat.rows(), mat.cols());
        cm.setFromTriplets(t);
        compute(cm);
    }

    Index rows() const { return m_rows; }
    Index cols() const { return m_cols; }
    Index rank() const { assert(m_isInitialized && "The factorization should be called first, use compute()"); return m_nonzeroPivots; }
    ComputationInfo info() const { return m_info; }
    // rows x cols with the upper triangle in its first cols rows (the reference keeps it sparse, column by column)
    const MatrixRType& matrixR() const {
        if (!m_Rbuilt) {
            std::vector<double> top((size_t)(m_cols * m_cols));
            check(qrk_thin_matrix_r(m_plan, top.data(), m_cols, QRK_MEM_HOST));
            m_R = Matrix(m_rows, m_cols);
            for (Index c = 0; c < m_cols; ++c) for (Index r = 0; r < m_cols && r < m_rows; ++r) m_R(r, c) = top[(size_t)(c * m_cols + r)];
            m_Rbuilt = true;
        }
        return m_R;
    }
    MatrixQType matrixQ() const { return MatrixQType(*this, false); }
    const PermutationType& colsPermutation() const { return m_outputPerm_c; }
    const PermutationType& rowsPermutation() const { return m_rowPerm; }

    Vector applyQt(const Vector& v) const { return applyAny(v, true); }
    Vector applyQ(const Vector& v) const { return applyAny(v, false); }
    // BlockedThinQRBase::_solve_impl (BlockedThinQRBase.h:223-247): x(0:rank) = R(0:rank,0:rank)^-1 (Q^T b)(0:rank), the rest zero
    Vector solve(const Vector& b) const {
        assert(m_isInitialized && "The factorization should be called first, use compute()");
        Padded d(*this, b, 1);
        check(qrk_thin_solve(m_plan, d.ptr(), 2 * m_rows, 1));
        Vector x((size_t)m_cols);
        check(qrk_memcpy(m_handle, x.data(), d.p, (int64_t)(m_cols * (Index)sizeof(double)), 1));
        return x;
    }
    // the SparseMatrixBase overload (BlockedThinQRBase.h: solve(const SparseMatrixBase<Rhs>&))
    template <bool RM2> SparseMatrix<false> solve(const SparseMatrix<RM2>& B) const { return detail::solveSparseThroughDensePanels<false>(*this, rows(), cols(), B); }

  protected:
    // device copy of nrhs columns with the zero rows the panels are applied with (leading dimension 2 rows)
    struct Padded {
        const BlockedThinSparseQR& s; void* p; Index nrhs;
        Padded(const BlockedThinSparseQR& ss, const Vector& host, Index n) : s(ss), p(0), nrhs(n) {
            const int64_t D = (int64_t)sizeof(double);
            Vector pad((size_t)(2 * s.m_rows * nrhs), 0.0);
            for (Index c = 0; c < nrhs; ++c) std::copy(host.begin() + c * s.m_rows, host.begin() + (c + 1) * s.m_rows, pad.begin() + c * 2 * s.m_rows);
            s.check(qrk_device_alloc(s.m_handle, 2 * s.m_rows * nrhs * D, &p));
            s.check(qrk_memcpy(s.m_handle, p, pad.data(), 2 * s.m_rows * nrhs * D, 0));
        }
        ~Padded() { if (p) qrk_device_free(s.m_handle, p); }
        double* ptr() const { return (double*)p; }
        Padded(const Padded&) = delete;
        Padded& operator=(const Padded&) = delete;
    };
    Vector applyAny(const Vector& v, bool transpose) const {
        assert(m_isInitialized && "The factorization should be called first, use compute()");
        const Index nrhs = (Index)v.size() / m_rows;
        Padded d(*this, v, nrhs);
        check(qrk_thin_apply_q(m_plan, transpose ? 1 : 0, d.ptr(), 2 * m_rows, nrhs));
        Vector pad((size_t)(2 * m_rows * nrhs));
        check(qrk_memcpy(m_handle, pad.data(), d.p, (int64_t)(pad.size() * sizeof(double)), 1));
        Vector out((size_t)(m_rows * nrhs));
        for (Index c = 0; c < nrhs; ++c) std::copy(pad.begin() + c * 2 * m_rows, pad.begin() + c * 2 * m_rows + m_rows, out.begin() + c * m_rows);
        return out;
    }
    void check(qrk_status st) const {
        if (st != QRK_STATUS_OK) throw std::runtime_error(std::string("qrkit: ") + qrk_last_error(m_handle));
    }
    ComputationInfo m_info;
    Index m_nonzeroPivots;
    bool m_isInitialized;
    qrk_handle m_handle;
    qrk_thin_plan m_plan;
    Index m_rows, m_cols;
    mutable Matrix m_R;
    mutable bool m_Rbuilt;
    PermutationType m_outputPerm_c, m_rowPerm;
};

// ---------------------------------------------------------------------------------------------
// QRKit::BlockMatrix1x2 (BlockMatrix1x2.h:31-67): non-owning [left | right].
template <typename LeftBlockType, typename RightBlockType = Matrix>
class BlockMatrix1x2 {
  public:
    BlockMatrix1x2(const LeftBlockType& l, const RightBlockType& r) : m_left(l), m_right(r) {}
    const LeftBlockType& leftBlock() const { return m_left; }
    const RightBlockType& rightBlock() const { return m_right; }
    Index rows() const { return (Index)m_right.rows(); }
    Index cols() const { return (Index)m_left.cols() + (Index)m_right.cols(); }
  private:
    const LeftBlockType& m_left;
    const RightBlockType& m_right;
};

// Uniform "Q^T v / Q v on several columns" access to the left solvers.
template <typename BS, int QF>
inline Vector leftApplyQ(const BlockDiagonalSparseQR<BS, QF>& s, const Vector& v, bool transpose) {
    return transpose ? s.applyQt(v) : s.applyQ(v);
}
template <int SBC, int BR, int BC, int BO>
inline Vector leftApplyQ(const BandedBlockedSparseQR<SBC, BR, BC, BO>& s, const Vector& v, bool transpose) {
    return transpose ? s.applyQt(v) : s.applyQ(v);
}

// QRKit::BlockAngularSparseQR<LeftSolver, RightSolver> (BlockAngularSparseQR.h:79-419): QR of [J1 | J2],
// J1 handled by LeftSolver (block diagonal or banded), J2 dense; the right solver is the dense
// ColPivHouseholderQR / HouseholderQR of the library (the reference tests use
// Eigen::ColPivHouseholderQR<MatrixXd>, test/test-qrkit.cpp:46-48).
template <typename LeftSolver, typename RightSolverTag = ColPivHouseholderQR>
class BlockAngularSparseQR {
  public:
    typedef SparseMatrixColMajor MatrixRType;
    typedef PermutationMatrix PermutationType;
    typedef QProduct<BlockAngularSparseQR> MatrixQType;

    explicit BlockAngularSparseQR(int device = 0)
        : m_leftSolver(device), m_info(Success), m_nonzeropivots(0), m_isInitialized(false), m_handle(0),
          m_rows(0), m_cols(0), m_m1(0), m_m2(0), m_n1(0), m_k2(0) {
        if (qrk_create(&m_handle, device, 0) != QRK_STATUS_OK)
            throw std::runtime_error(std::string("qrkit: ") + qrk_last_error(0));
    }
    ~BlockAngularSparseQR() {
        m_dense.release();
        if (m_dS) qrk_device_free(m_handle, m_dS);
        if (m_dTop.p) qrk_device_free(m_handle, m_dTop.p);
        if (m_dT.p) qrk_device_free(m_handle, m_dT.p);
        if (m_handle) qrk_destroy(m_handle);
    }
    BlockAngularSparseQR(const BlockAngularSparseQR&) = delete;
    BlockAngularSparseQR& operator=(const BlockAngularSparseQR&) = delete;

    template <typename LeftMat>
    void compute(const BlockMatrix1x2<LeftMat, Matrix>& mat) { analyzePattern(mat); factorize(mat); }
    // sparse right block (BlockMatrix1x2<JacobianType, JacobianType>, test/test-qrkit.cpp:335): the right solver
    // works on dense columns either way (BlockedThinSparseQR.h:131: m_R = mat)
    template <typename LeftMat, bool RM>
    void compute(const BlockMatrix1x2<LeftMat, SparseMatrix<RM> >& mat) { analyzePattern(mat); factorize(mat); }

    // :431-449
    template <typename LeftMat, typename RightMat>
    void analyzePattern(const BlockMatrix1x2<LeftMat, RightMat>& mat) {
        assert(mat.leftBlock().cols() > mat.rightBlock().cols() && "the left block should be the bigger one");
        m_rows = mat.rows(); m_cols = mat.cols();
        m_rowPerm.setIdentity(m_rows);
    }

    // :459-514.  J2 crosses PCIe once: Q1^T J2, the strip S = (Q1^T J2)(0:m1, :) and the packed QR of the bottom rows stay on the
    // device (the host copy of R is assembled when matrixR() is first asked for).
    template <typename LeftMat>
    void factorize(const BlockMatrix1x2<LeftMat, Matrix>& mat) {
        const Matrix& J2in = mat.rightBlock();
        const int64_t D = (int64_t)sizeof(double);
        factorizeWith(mat.leftBlock(), J2in.cols(),
            [&](double* dTop, const std::vector<int>& rp, bool identity) {           // rows [0, n1) of J2, row r -> rp[r]
                if (identity) {
                    check(qrk_memcpy_2d(m_handle, dTop, m_n1 * D, J2in.data(), m_rows * D, m_n1 * D, m_m2, 0));
                } else {
                    Vector top((size_t)(m_n1 * m_m2));
                    for (Index c = 0; c < m_m2; ++c)
                        for (Index r = 0; r < m_n1; ++r) top[(size_t)(c * m_n1 + rp[(size_t)r])] = J2in(r, c);
                    check(qrk_memcpy(m_handle, dTop, top.data(), m_n1 * m_m2 * D, 0));
                }
            },
            [&](double* dst, int64_t ld) {                                            // rows [n1, rows) of J2
                return qrk_memcpy_2d(m_handle, dst, ld * D, J2in.data() + m_n1, m_rows * D, (m_rows - m_n1) * D, m_m2, 0);
            });
    }
    // sparse right block (BlockMatrix1x2<JacobianType, JacobianType>, test/test-qrkit.cpp:335): its nonzeros cross PCIe and the
    // dense copy the right solver works on (BlockedThinSparseQR.h:131: m_R = mat) is written on the device
    template <typename LeftMat, bool RM>
    void factorize(const BlockMatrix1x2<LeftMat, SparseMatrix<RM> >& mat) {
        const SparseMatrix<RM>& J2in = mat.rightBlock();
        const int64_t nnz = J2in.nonZeros(), no = J2in.outerSize();
        void *dOuter = 0, *dInner = 0, *dVal = 0, *dMap = 0;
        auto release = [&]() { qrk_device_free(m_handle, dOuter); qrk_device_free(m_handle, dInner); qrk_device_free(m_handle, dVal); qrk_device_free(m_handle, dMap); };
        try {
            check(qrk_device_alloc(m_handle, (no + 1) * 4, &dOuter));
            check(qrk_device_alloc(m_handle, std::max<int64_t>(nnz, 1) * 4, &dInner));
            check(qrk_device_alloc(m_handle, std::max<int64_t>(nnz, 1) * 8, &dVal));
            check(qrk_memcpy(m_handle, dOuter, J2in.outerIndex().data(), (no + 1) * 4, 0));
            if (nnz > 0) {
                check(qrk_memcpy(m_handle, dInner, J2in.innerIndex().data(), nnz * 4, 0));
                check(qrk_memcpy(m_handle, dVal, J2in.values().data(), nnz * 8, 0));
            }
            factorizeWith(mat.leftBlock(), J2in.cols(),
                [&](double* dTop, const std::vector<int>& rp, bool identity) {
                    if (!identity) {
                        check(qrk_device_alloc(m_handle, std::max<int64_t>(m_n1, 1) * 4, &dMap));
                        check(qrk_memcpy(m_handle, dMap, rp.data(), m_n1 * 4, 0));
                    }
                    check(qrk_sparse_window_to_dense(m_handle, RM ? 1 : 0, m_rows, m_m2, (const int32_t*)dOuter, (const int32_t*)dInner,
                                                     (const double*)dVal, 0, m_n1, (const int32_t*)dMap, dTop, m_n1));
                },
                [&](double* dst, int64_t ld) {
                    return qrk_sparse_window_to_dense(m_handle, RM ? 1 : 0, m_rows, m_m2, (const int32_t*)dOuter, (const int32_t*)dInner,
                                                      (const double*)dVal, m_n1, m_rows - m_n1, 0, dst, ld);
                });
            check(qrk_synchronize(m_handle));
        } catch (...) { release(); throw; }
        release();
    }

    Index rows() const { return m_rows; }
    Index cols() const { return m_cols; }
    Index rank() const { assert(m_isInitialized); return m_nonzeropivots; }
    ComputationInfo info() const { return m_info; }
    const MatrixRType& matrixR() const { if (!m_Rbuilt) makeR(); return m_R; }   // (host copy assembled on first use)
    MatrixQType matrixQ() const { return MatrixQType(*this, false); }
    const PermutationType& colsPermutation() const { return m_outputPerm_c; }
    const PermutationType& rowsPermutation() const { return m_rowPerm; }
    const LeftSolver& leftSolver() const { return m_leftSolver; }

    // matrixQ().transpose() * v (:607-625): rows [0,n1) <- Q1^T, then rows [m1, rows) <- Q2^T; the vector crosses PCIe once each way
    Vector applyQt(const Vector& v) const { return applyAny(v, true); }
    // matrixQ() * v (:627-645): the same two factors in the opposite order
    Vector applyQ(const Vector& v) const { return applyAny(v, false); }
    // _solve_impl (:202-227): x = P [R(0:rank,0:rank)^-1 (Q^T b)(0:rank)], b already row-permuted
    Vector solve(const Vector& b) const {
        assert(m_isInitialized && "The factorization should be called first, use compute()");
        if (m_rows - m_m1 < m_m2) return m_outputPerm_c * solveUpperCsc(matrixR(), m_cols, applyQt(b));   // (fewer bottom rows than right columns)
        // R = [R1 S; 0 R2], block by block and on the device: y = Q^T b, z2 = R2^-1 y2 (qrk_dense_solve_r), y1 -= S(:, P2) z2
        // (qrk_dense_gemv_sub), z1 = R1^-1 y1 (the left solver's triangular step); only b and x cross PCIe
        const int64_t D = (int64_t)sizeof(double);
        DBuf y(*this, m_rows), z2(*this, m_m2), z1(*this, m_m1);
        check(qrk_memcpy(m_handle, y.p, b.data(), m_rows * D, 0));
        applyDevice(y.ptr(), 1, true);
        check(qrk_memcpy_2d(m_handle, z2.p, m_m2 * D, y.ptr() + m_m1, m_m2 * D, m_m2 * D, 1, 2));
        m_dense.solveRDevice(z2.ptr(), m_m2, 1);
        check(qrk_dense_gemv_sub(m_handle, (const double*)m_dS, m_m1, m_m1, m_m2, m_dense.permDevice(), z2.ptr(), y.ptr()));
        check(qrk_synchronize(m_handle));
        m_leftSolver.solveRDevice(y.ptr(), 1, z1.ptr());
        Vector z((size_t)(m_m1 + m_m2));
        check(qrk_memcpy(m_handle, z.data(), z1.p, m_m1 * D, 1));
        check(qrk_memcpy(m_handle, z.data() + m_m1, z2.p, m_m2 * D, 1));
        return m_outputPerm_c * z;
    }
    // the SparseMatrixBase overload (BlockAngularSparseQR.h: solve(const SparseMatrixBase<Rhs>&))
    template <bool RM2> SparseMatrix<false> solve(const SparseMatrix<RM2>& B) const { return detail::solveSparseThroughDensePanels<false>(*this, rows(), cols(), B); }

  protected:
    // the factorisation proper: `top(dTop, rp, identity)` puts rows [0, n1) of J2 on the device (n1 x m2, column-major, source row
    // r at row rp[r]), `bottom(dst, ld)` rows [n1, rows) at dst with leading dimension ld
    template <typename LeftMat, typename TopFn, typename BottomFn>
    void factorizeWith(const LeftMat& left, Index m2, TopFn top, BottomFn bottom) {
        m_m1 = left.cols(); m_n1 = left.rows(); m_m2 = m2;
        m_isInitialized = false; m_Rbuilt = false;
        // J1 = Q1 R1 (:472-475)
        m_leftSolver.compute(left);
        m_info = m_leftSolver.info();
        if (m_info != Success) return;
        const int64_t D = (int64_t)sizeof(double);
        const Index rb = m_rows - m_m1;
        // solveRightBlock (:361-369): J2.top(n1) <- Q1^T (rowPerm1 * J2.top(n1)); the rows below stay
        const std::vector<int>& rp = m_leftSolver.rowsPermutation().indices();
        bool identity = true;
        for (Index r = 0; r < m_n1 && identity; ++r) identity = rp[(size_t)r] == (int)r;
        for (Index r = 0; r < m_n1; ++r) m_rowPerm.indices()[(size_t)r] = rp[(size_t)r];
        void* dBottom = 0;
        {
            // (the two n1 x m2 work matrices are kept between calls: an LM loop calls compute() with one shape, and hipMalloc / hipFree
            //  of gigabytes cost milliseconds)
            Scratch &dTop = m_dTop, &dT = m_dT;
            dTop.reserve(*this, m_n1 * m_m2); dT.reserve(*this, m_n1 * m_m2);
            top(dTop.ptr(), rp, identity);
            m_leftSolver.applyQDevice(dTop.ptr(), m_m2, dT.ptr(), true);
            // rightSolver.compute(J2.bottomRows(rows - m1)): rows m1..n1 of Q1^T J2, then the rows of J2 below J1
            dBottom = m_dense.acquireBuffer(m_handle, rb, m_m2);
            qrk_status st = qrk_memcpy_2d(m_handle, dBottom, rb * D, dT.ptr() + m_m1, m_n1 * D, (m_n1 - m_m1) * D, m_m2, 2);
            if (st == QRK_STATUS_OK && m_rows > m_n1) st = bottom((double*)dBottom + (m_n1 - m_m1), rb);
            // the strip of R: S = (Q1^T J2)(0:m1, :)
            if (m_dS && m_dScount != m_m1 * m_m2) { qrk_device_free(m_handle, m_dS); m_dS = 0; }
            if (st == QRK_STATUS_OK && !m_dS) { st = qrk_device_alloc(m_handle, std::max<int64_t>(m_m1 * m_m2 * D, 8), &m_dS); m_dScount = m_m1 * m_m2; }
            if (st == QRK_STATUS_OK) st = qrk_memcpy_2d(m_handle, m_dS, m_m1 * D, dT.ptr(), m_n1 * D, m_m1 * D, m_m2, 2);
            if (st != QRK_STATUS_OK) { qrk_device_free(m_handle, dBottom); check(st); }
        }
        m_k2 = std::min(rb, m_m2);
        std::vector<int32_t> p2;
        m_dense.factorizeDevice(m_handle, dBottom, rb, m_m2, RightSolverTag::kSolver, m_hc, p2);   // (takes the buffer over)
        // column permutation (:498-503) and rank (:510)
        m_outputPerm_c.setIdentity(m_cols);
        for (Index j = 0; j < m_m1; ++j) m_outputPerm_c.indices()[(size_t)j] = m_leftSolver.colsPermutation().indices()[(size_t)j];
        for (Index j = 0; j < m_m2; ++j) m_outputPerm_c.indices()[(size_t)(m_m1 + j)] = (int)(m_m1 + p2[(size_t)j]);
        m_P2.assign(p2.begin(), p2.begin() + m_m2);
        m_nonzeropivots = m_leftSolver.rank() + m_k2;
        m_isInitialized = true;
    }
    // grow-only device buffer of doubles on this solver's handle (freed with the solver)
    struct Scratch {
        void* p; int64_t cap;
        Scratch() : p(0), cap(0) {}
        void reserve(const BlockAngularSparseQR& s, int64_t count) {
            if (count <= cap) return;
            if (p) { qrk_device_free(s.m_handle, p); p = 0; cap = 0; }
            s.check(qrk_device_alloc(s.m_handle, std::max<int64_t>(count, 1) * (int64_t)sizeof(double), &p));
            cap = count;
        }
        double* ptr() const { return (double*)p; }
    };
    // scoped device buffer of doubles on this solver's handle
    struct DBuf {
        const BlockAngularSparseQR& s; void* p;
        DBuf(const BlockAngularSparseQR& ss, int64_t count) : s(ss), p(0) {
            s.check(qrk_device_alloc(s.m_handle, std::max<int64_t>(count, 1) * (int64_t)sizeof(double), &p));
        }
        ~DBuf() { if (p) qrk_device_free(s.m_handle, p); }
        double* ptr() const { return (double*)p; }
        DBuf(const DBuf&) = delete;
        DBuf& operator=(const DBuf&) = delete;
    };
    Vector applyAny(const Vector& v, bool transpose) const {
        const Index nrhs = (Index)v.size() / m_rows;
        DBuf d(*this, m_rows * nrhs);
        check(qrk_memcpy(m_handle, d.p, v.data(), m_rows * nrhs * (int64_t)sizeof(double), 0));
        applyDevice(d.ptr(), nrhs, transpose);
        Vector out(v.size());
        check(qrk_memcpy(m_handle, out.data(), d.p, m_rows * nrhs * (int64_t)sizeof(double), 1));
        return out;
    }
    // d_v (rows x nrhs on the device, leading dimension rows) <- Q^T d_v or Q d_v
    void applyDevice(double* d_v, Index nrhs, bool transpose) const {
        if (transpose) { leftPart(d_v, nrhs, true); rightPart(d_v, nrhs, true); }
        else { rightPart(d_v, nrhs, false); leftPart(d_v, nrhs, false); }
    }
    void leftPart(double* d_v, Index nrhs, bool transpose) const {
        const int64_t D = (int64_t)sizeof(double);
        DBuf a(*this, m_n1 * nrhs), b(*this, m_n1 * nrhs);
        check(qrk_memcpy_2d(m_handle, a.p, m_n1 * D, d_v, m_rows * D, m_n1 * D, nrhs, 2));
        m_leftSolver.applyQDevice(a.ptr(), nrhs, b.ptr(), transpose);
        check(qrk_memcpy_2d(m_handle, d_v, m_rows * D, b.p, m_n1 * D, m_n1 * D, nrhs, 2));
    }
    void rightPart(double* d_v, Index nrhs, bool transpose) const {
        const int64_t D = (int64_t)sizeof(double);
        const Index rb = m_rows - m_m1;
        DBuf c(*this, rb * nrhs);
        check(qrk_memcpy_2d(m_handle, c.p, rb * D, d_v + m_m1, m_rows * D, rb * D, nrhs, 2));
        m_dense.applyQDevice(c.ptr(), nrhs, transpose);
        check(qrk_memcpy_2d(m_handle, d_v + m_m1, m_rows * D, c.p, rb * D, rb * D, nrhs, 2));
    }
    // makeR (:285-335): R = [R1(0:m1,:), (Q1^T J2)(0:m1, P2); 0, R2; 0, 0] as a host sparse matrix: the strip and R2 come from the device
    void makeR() const {
        const int64_t D = (int64_t)sizeof(double);
        const Index rb = m_rows - m_m1;
        Vector S((size_t)(m_m1 * m_m2)), R2((size_t)(m_k2 * m_m2));
        check(qrk_memcpy(m_handle, S.data(), m_dS, m_m1 * m_m2 * D, 1));
        check(qrk_memcpy_2d(m_handle, R2.data(), m_k2 * D, m_dense.packedDevice(), rb * D, m_k2 * D, m_m2, 1));
        std::vector<Triplet> t;
        const SparseMatrixColMajor& R1 = m_leftSolver.matrixR();
        for (Index c = 0; c < m_m1; ++c)
            for (int p = R1.outerIndex()[(size_t)c]; p < R1.outerIndex()[(size_t)c + 1]; ++p)
                if (R1.innerIndex()[(size_t)p] < m_m1) t.emplace_back(R1.innerIndex()[(size_t)p], (int)c, R1.values()[(size_t)p]);
        for (Index c = 0; c < m_m2; ++c)
            for (Index r = 0; r < m_m1; ++r) t.emplace_back((int)r, (int)(m_m1 + c), S[(size_t)(m_P2[(size_t)c] * m_m1 + r)]);
        for (Index c = 0; c < m_m2; ++c)
            for (Index r = 0; r <= std::min(c, m_k2 - 1); ++r) t.emplace_back((int)(m_m1 + r), (int)(m_m1 + c), R2[(size_t)(c * m_k2 + r)]);
        m_R.resize(m_rows, m_cols);
        m_R.setFromTriplets(t);
        m_Rbuilt = true;
    }
    void check(qrk_status st) const {
        if (st != QRK_STATUS_OK) throw std::runtime_error(std::string("qrkit: ") + qrk_last_error(m_handle));
    }
    LeftSolver m_leftSolver;
    ComputationInfo m_info;
    Index m_nonzeropivots;
    bool m_isInitialized;
    qrk_handle m_handle;
    DenseDeviceQR m_dense;
    Index m_rows, m_cols, m_m1, m_m2, m_n1, m_k2;
    void* m_dS = 0;                     // device: the strip S = (Q1^T J2)(0:m1, :), m1 x m2 (the packed QR of the rows below is m_dense's)
    int64_t m_dScount = 0;
    Scratch m_dTop, m_dT;               // device: J2.top(n1) and Q1^T of it, kept between factorisations
    std::vector<double> m_hc;
    std::vector<int> m_P2;
    mutable MatrixRType m_R;            // host copy of R, assembled by the first matrixR()
    mutable bool m_Rbuilt = false;
    PermutationType m_outputPerm_c, m_rowPerm;
};


// ---------------------------------------------------------------------------------------------
// QRKit::BlockAngularSparseQR with the ROWS sharded over the GPUs of a node (BASELINE configs[3], "8 x MI355X sharded"; the C++
// counterpart of qrkit_amd/sharding.py::ShardedBlockAngularQR; SURVEY.md 8(e) -- no reference site for the sharding itself).
// One process per GPU.  Rank g holds a contiguous range of the tiles of the block-diagonal left block J1 and the rows of the dense
// right block J2 that belong to them (any rows of J2 below J1 go to one rank).  compute (BlockAngularSparseQR.h:459-514):
//   * J1_g = Q1_g R1_g and T = Q1_g^T J2_g on the rank, no exchange (:472-475, solveRightBlock :361-369); the strip S_g = T(0:m1_g, :)
//     stays on the rank;
//   * the rows of T below it are reduced ON THE RANK to one m2 x m2 triangle (qrk_tsqr_*: un-pivoted CAQR on the matrix cores);
//   * the root gathers `world` triangles (m2^2 doubles each: qrk_gather_equal), stacks them and runs the pivoted right solver on
//     the stack -- tall-skinny QR across ranks: the Gram structure of the columns, hence Eigen's pivots, is that of the un-sharded block;
//   * the permutation P2 of the right block goes back to every rank (qrk_bcast).
// solve (_solve_impl :202-227) follows the same route with one m2-vector per rank up and z2 down; every rank gets the entries of x
// of its own tiles and the m2 entries of the right block.
// Failure contract: compute() / solve() are collectives.  An exception thrown on ONE rank before its qrk_gather_equal / qrk_bcast (a
// failed left solver, an allocation) leaves the other ranks inside the collective -- as with any RCCL program, the caller must treat an
// exception on any rank as fatal for the communicator (abort the job or ncclCommAbort); no status word is exchanged ahead of the data.
template <typename LeftSolver = BlockDiagonalSparseQR<> >
class ShardedBlockAngularSparseQR {
  public:
    ShardedBlockAngularSparseQR(int rank, int world, void* ncclComm, int device = 0, int root = 0)
        : m_rank(rank), m_world(world), m_root(root), m_comm(ncclComm), m_leftSolver(device), m_handle(0), m_tsqr(0) {
        if (world <= 0 || rank < 0 || rank >= world || root < 0 || root >= world) throw std::runtime_error("qrkit: ShardedBlockAngularSparseQR: bad rank / world / root");
        if (qrk_create(&m_handle, device, 0) != QRK_STATUS_OK) throw std::runtime_error("qrkit: ShardedBlockAngularSparseQR: no device");
    }
    ~ShardedBlockAngularSparseQR() {
        m_right.release();
        if (m_tsqr) qrk_tsqr_plan_destroy(m_tsqr);
        for (void* p : {m_dS, m_dBottom, m_dP2}) if (p) qrk_device_free(m_handle, p);
        if (m_handle) qrk_destroy(m_handle);
    }
    ShardedBlockAngularSparseQR(const ShardedBlockAngularSparseQR&) = delete;
    ShardedBlockAngularSparseQR& operator=(const ShardedBlockAngularSparseQR&) = delete;

    // localLeft: this rank's tiles of J1; localRight: this rank's rows of J2 (rows >= localLeft.rows(), m2 columns)
    void compute(const SparseBlockDiagonal& localLeft, const Matrix& localRight) {
        const int64_t D = (int64_t)sizeof(double);
        m_n1 = localLeft.rows(); m_m1 = localLeft.cols(); m_m2 = localRight.cols();
        const Index nloc = localRight.rows();
        if (nloc < m_n1) throw std::runtime_error("qrkit: ShardedBlockAngularSparseQR: the right block has fewer rows than the left one");
        m_leftSolver.compute(localLeft);
        m_info = m_leftSolver.info();
        if (m_info != Success) throw std::runtime_error("qrkit: ShardedBlockAngularSparseQR: the left solver failed (a collective would hang)");
        m_nb = nloc - m_m1;                                           // rows of this rank's bottom block
        freeBuf(m_dS); freeBuf(m_dBottom); freeBuf(m_dP2);            // (all three are sized by this call's right block)
        // T = Q1^T (rowPerm1 J2.top(n1)) on the device
        Buf dTop(m_handle, m_n1 * m_m2), dT(m_handle, m_n1 * m_m2);
        {
            const std::vector<int>& rp = m_leftSolver.rowsPermutation().indices();
            Matrix top(m_n1, m_m2);
            for (Index j = 0; j < m_m2; ++j) for (Index i = 0; i < m_n1; ++i) top(rp[(size_t)i], j) = localRight(i, j);
            check(qrk_memcpy(m_handle, dTop.p, top.data(), m_n1 * m_m2 * D, 0));
        }
        m_leftSolver.applyQDevice(dTop.ptr(), m_m2, dT.ptr(), true);
        check(qrk_device_alloc(m_handle, std::max<int64_t>(m_m1 * m_m2 * D, 8), &m_dS));
        check(qrk_memcpy_2d(m_handle, m_dS, m_m1 * D, dT.ptr(), m_n1 * D, m_m1 * D, m_m2, 2));
        // bottom_g = [T(m1:n1, :); J2 below J1], m_nb x m2, column-major
        const Index ldb = std::max<Index>(m_nb, m_m2);                 // (fewer rows than columns: zero rows below, see below)
        check(qrk_device_alloc(m_handle, std::max<int64_t>(ldb * m_m2 * D, 8), &m_dBottom));
        if (ldb > m_nb) {
            Vector zeros((size_t)(ldb * m_m2), 0.0);
            check(qrk_memcpy(m_handle, m_dBottom, zeros.data(), ldb * m_m2 * D, 0));
        }
        if (m_n1 > m_m1) check(qrk_memcpy_2d(m_handle, m_dBottom, ldb * D, dT.ptr() + m_m1, m_n1 * D, (m_n1 - m_m1) * D, m_m2, 2));
        if (nloc > m_n1) {
            Matrix below(nloc - m_n1, m_m2);
            for (Index j = 0; j < m_m2; ++j) for (Index i = m_n1; i < nloc; ++i) below(i - m_n1, j) = localRight(i, j);
            check(qrk_memcpy_2d(m_handle, (double*)m_dBottom + (m_n1 - m_m1), ldb * D, below.data(), (nloc - m_n1) * D, (nloc - m_n1) * D, m_m2, 0));
        }
        // its triangle: un-pivoted CAQR on the rank (with fewer rows than columns the rows themselves, zero-padded, are the "triangle")
        Buf tri(m_handle, m_m2 * m_m2);
        m_reduced = m_nb >= m_m2;
        if (m_reduced) {
            if (m_tsqr) { qrk_tsqr_plan_destroy(m_tsqr); m_tsqr = 0; }
            check(qrk_tsqr_plan_create(m_handle, (int32_t)m_nb, (int32_t)m_m2, &m_tsqr));
            check(qrk_tsqr_factorize(m_tsqr, (double*)m_dBottom, ldb, QRK_MEM_DEVICE));
            Vector host((size_t)(m_m2 * m_m2));
            check(qrk_memcpy_2d(m_handle, host.data(), m_m2 * D, m_dBottom, ldb * D, m_m2 * D, m_m2, 1));
            for (Index j = 0; j < m_m2; ++j) for (Index i = j + 1; i < m_m2; ++i) host[(size_t)(j * m_m2 + i)] = 0.0;     // (reflectors below R0)
            check(qrk_memcpy(m_handle, tri.p, host.data(), m_m2 * m_m2 * D, 0));
        } else {
            check(qrk_memcpy_2d(m_handle, tri.p, m_m2 * D, m_dBottom, ldb * D, m_m2 * D, m_m2, 2));
        }
        // the triangles to the root, the pivoted right solver on their stack there, P2 back
        Buf parts(m_handle, m_rank == m_root ? (Index)m_world * m_m2 * m_m2 : 1);
        check(qrk_gather_equal(m_handle, m_comm, m_rank, m_world, m_root, tri.ptr(), m_m2 * m_m2, parts.ptr()));
        check(qrk_synchronize(m_handle));
        if (!m_dP2) check(qrk_device_alloc(m_handle, std::max<int64_t>(m_m2 * (int64_t)sizeof(int32_t), 8), &m_dP2));
        m_P2.assign((size_t)m_m2, 0);
        if (m_rank == m_root) {
            const Index sr = (Index)m_world * m_m2;
            void* stack = m_right.acquireBuffer(m_handle, sr, m_m2);
            for (int g = 0; g < m_world; ++g)
                check(qrk_memcpy_2d(m_handle, (double*)stack + g * m_m2, sr * D, parts.ptr() + (Index)g * m_m2 * m_m2, m_m2 * D, m_m2 * D, m_m2, 2));
            std::vector<int32_t> p2;
            m_right.factorizeDevice(m_handle, stack, sr, m_m2, QRK_COLPIV_HOUSEHOLDER, m_hc, p2);
            for (Index j = 0; j < m_m2; ++j) m_P2[(size_t)j] = p2[(size_t)j];
            check(qrk_memcpy(m_handle, m_dP2, m_P2.data(), m_m2 * (int64_t)sizeof(int32_t), 0));
        }
        check(qrk_bcast(m_handle, m_comm, m_rank, m_world, m_root, m_dP2, m_m2 * (int64_t)sizeof(int32_t)));
        check(qrk_synchronize(m_handle));
        check(qrk_memcpy(m_handle, m_P2.data(), m_dP2, m_m2 * (int64_t)sizeof(int32_t), 1));
        m_isInitialized = true;
    }
    ComputationInfo info() const { return m_info; }
    const LeftSolver& leftSolver() const { return m_leftSolver; }
    // colsPermutation() of the right block: column j of (J2 P2) is column colsPermutationRight()[j] of J2 -- the same on every rank
    const std::vector<int>& colsPermutationRight() const { return m_P2; }

    // Least squares: b_local = this rank's rows of the right-hand side.  x1_local: the entries of x that belong to this rank's tiles
    // (in the order of its columns of J1), x2: the m2 entries of the right block (on every rank).
    void solve(const Vector& b_local, Vector& x1_local, Vector& x2) const {
        assert(m_isInitialized && "The factorization should be called first, use compute()");
        const int64_t D = (int64_t)sizeof(double);
        const Index nloc = m_m1 + m_nb, ldb = std::max<Index>(m_nb, m_m2);
        assert((Index)b_local.size() == nloc);
        Buf b(m_handle, m_n1), y(m_handle, m_n1), yb(m_handle, ldb), t(m_handle, m_m2);
        {
            const std::vector<int>& rp = m_leftSolver.rowsPermutation().indices();
            Vector top((size_t)m_n1);
            for (Index i = 0; i < m_n1; ++i) top[(size_t)rp[(size_t)i]] = b_local[(size_t)i];
            check(qrk_memcpy(m_handle, b.p, top.data(), m_n1 * D, 0));
        }
        m_leftSolver.applyQDevice(b.ptr(), 1, y.ptr(), true);
        {
            Vector host((size_t)ldb, 0.0), ytop((size_t)m_n1);
            check(qrk_memcpy(m_handle, ytop.data(), y.p, m_n1 * D, 1));
            for (Index i = m_m1; i < m_n1; ++i) host[(size_t)(i - m_m1)] = ytop[(size_t)i];
            for (Index i = m_n1; i < nloc; ++i) host[(size_t)(i - m_m1)] = b_local[(size_t)i];
            check(qrk_memcpy(m_handle, yb.p, host.data(), ldb * D, 0));
        }
        if (m_reduced) check(qrk_tsqr_apply_q(m_tsqr, (const double*)m_dBottom, ldb, 1, yb.ptr(), ldb, 1, QRK_MEM_DEVICE));
        check(qrk_memcpy_2d(m_handle, t.p, m_m2 * D, yb.p, ldb * D, m_m2 * D, 1, 2));
        Buf parts(m_handle, m_rank == m_root ? (Index)m_world * m_m2 : 1), z2(m_handle, m_m2);
        check(qrk_gather_equal(m_handle, m_comm, m_rank, m_world, m_root, t.ptr(), m_m2, parts.ptr()));
        check(qrk_synchronize(m_handle));
        if (m_rank == m_root) {
            m_right.applyQDevice(parts.ptr(), 1, true);               // (the stack of the m2-vectors, in the order of the triangles)
            check(qrk_memcpy_2d(m_handle, z2.p, m_m2 * D, parts.p, m_m2 * D, m_m2 * D, 1, 2));
            m_right.solveRDevice(z2.ptr(), m_m2, 1);
        }
        check(qrk_bcast(m_handle, m_comm, m_rank, m_world, m_root, z2.p, m_m2 * D));
        check(qrk_synchronize(m_handle));
        // z1 = R1^-1 (y1 - S(:, P2) z2) on the device; x1 = colsPermutation1 * z1, x2 = P2 * z2
        check(qrk_dense_gemv_sub(m_handle, (const double*)m_dS, m_m1, m_m1, m_m2, (const int32_t*)m_dP2, z2.ptr(), y.ptr()));
        check(qrk_synchronize(m_handle));
        Buf z1(m_handle, m_m1);
        m_leftSolver.solveRDevice(y.ptr(), 1, z1.ptr());
        Vector hz1((size_t)m_m1), hz2((size_t)m_m2);
        check(qrk_memcpy(m_handle, hz1.data(), z1.p, m_m1 * D, 1));
        check(qrk_memcpy(m_handle, hz2.data(), z2.p, m_m2 * D, 1));
        x1_local = m_leftSolver.colsPermutation() * hz1;
        x2.assign((size_t)m_m2, 0.0);
        for (Index j = 0; j < m_m2; ++j) x2[(size_t)m_P2[(size_t)j]] = hz2[(size_t)j];
    }

  protected:
    struct Buf {                 // scoped device buffer of doubles
        qrk_handle h; void* p;
        Buf(qrk_handle hh, Index count) : h(hh), p(0) {
            if (qrk_device_alloc(h, std::max<int64_t>(count, 1) * (int64_t)sizeof(double), &p) != QRK_STATUS_OK)
                throw std::runtime_error(std::string("qrkit: ") + qrk_last_error(h));
        }
        ~Buf() { if (p) qrk_device_free(h, p); }
        double* ptr() const { return (double*)p; }
        Buf(const Buf&) = delete;
        Buf& operator=(const Buf&) = delete;
    };
    void freeBuf(void*& p) { if (p) { qrk_device_free(m_handle, p); p = 0; } }
    void check(qrk_status st) const {
        if (st != QRK_STATUS_OK) throw std::runtime_error(std::string("qrkit: ") + qrk_last_error(m_handle));
    }
    int m_rank, m_world, m_root;
    void* m_comm;
    LeftSolver m_leftSolver;
    qrk_handle m_handle;
    qrk_tsqr_plan m_tsqr;
    mutable DenseDeviceQR m_right;      // root only: the pivoted QR of the stacked triangles
    ComputationInfo m_info = Success;
    bool m_isInitialized = false, m_reduced = false;
    Index m_n1 = 0, m_m1 = 0, m_m2 = 0, m_nb = 0;
    void *m_dS = 0, *m_dBottom = 0, *m_dP2 = 0;
    std::vector<double> m_hc;
    std::vector<int> m_P2;
};

}  // namespace qrkit

// The reference's namespace and template parameter lists (src/QRKit/*.h), so that code written against QRKit compiles against
// this header by changing the include: parameters the device engine does not need (_MatrixType, _StorageIndex) are accepted and
// ignored; the _BlockQRSolver argument is one of the tags above (Eigen's solver classes are not available without Eigen).
namespace QRKit {
using qrkit::Index;
using qrkit::Matrix;
using qrkit::Vector;
using qrkit::SparseMatrix;
using qrkit::SparseMatrixRowMajor;
using qrkit::SparseMatrixColMajor;
using qrkit::PermutationMatrix;
using qrkit::ComputationInfo;
using qrkit::Success;
using qrkit::InvalidInput;
using qrkit::Dynamic;
using qrkit::ColPivHouseholderQR;
using qrkit::HouseholderQR;
using qrkit::ColPivHouseholderQRFixed;
using qrkit::HouseholderQRFixed;
enum MatrixQFormat { FullQ = 0, BlockDiagonalQ = 1 };     // BlockDiagonalSparseQR.h:59-62
// SparseBlockDiagonal<BlockMatrixType, _StorageIndex> (SparseBlockDiagonal.h:43-44)
template <typename BlockMatrixType = Matrix, typename StorageIndex = int> using SparseBlockDiagonal = qrkit::SparseBlockDiagonal;
// BlockDiagonalSparseQR<_BlockQRSolver, _QFormat> (BlockDiagonalSparseQR.h:37)
template <typename BlockQRSolver = qrkit::ColPivHouseholderQR, int QFormat = 0>
using BlockDiagonalSparseQR = qrkit::BlockDiagonalSparseQR<BlockQRSolver, QFormat>;
// no reference counterpart: the multi-GPU form of the same solver (one process per GPU)
template <typename BlockQRSolver = qrkit::ColPivHouseholderQR, int QFormat = 0>
using ShardedBlockDiagonalSparseQR = qrkit::ShardedBlockDiagonalSparseQR<BlockQRSolver, QFormat>;
// BandedBlockedSparseQR<_MatrixType, _BlockQRSolver, _BlockOverlap = Dynamic, _SuggestedBlockCols = 2> (BandedBlockedSparseQR.h:122)
template <typename MatrixType, typename BlockQRSolver, int BlockOverlap = qrkit::Dynamic, int SuggestedBlockCols = 2>
using BandedBlockedSparseQR =
    qrkit::BandedBlockedSparseQR<SuggestedBlockCols, BlockQRSolver::RowsAtCompileTime, BlockQRSolver::ColsAtCompileTime, BlockOverlap>;
// BlockedThinDenseQR / BlockedThinSparseQR<_MatrixType, _SuggestedBlockCols = 2> (BlockedThinDenseQR.h:61, BlockedThinSparseQR.h:58)
template <typename MatrixType, int SuggestedBlockCols = 2> using BlockedThinDenseQR = qrkit::BlockedThinDenseQR<SuggestedBlockCols>;
template <typename MatrixType, int SuggestedBlockCols = 2> using BlockedThinSparseQR = qrkit::BlockedThinSparseQR<SuggestedBlockCols>;
// BlockMatrix1x2<LeftBlockMatrixType, RightBlockMatrixType> (BlockMatrix1x2.h:31), BlockAngularSparseQR<Left, Right> (BlockAngularSparseQR.h:79)
template <typename L, typename R> using BlockMatrix1x2 = qrkit::BlockMatrix1x2<L, R>;
template <typename L, typename R> using BlockAngularSparseQR = qrkit::BlockAngularSparseQR<L, R>;
}  // namespace QRKit

#endif  // QRKIT_FACADE_HPP
