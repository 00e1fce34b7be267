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

// QRKit::SparseBlockDiagonal (SparseBlockDiagonal.h:43-163): the blocks are kept packed back to back
// (column-major), which is what std::vector<Matrix<double,r,c>> is for fixed-size blocks.
class SparseBlockDiagonal {
  public:
    SparseBlockDiagonal() : nRows(0), nCols(0) {}
    SparseBlockDiagonal(StorageIndex rows, StorageIndex cols) : nRows(rows), nCols(cols) {}

    // SparseBlockDiagonal.h:71-89 with the block map of SparseQRUtils.h:255-272:
    // numBlocks = matCols / blockCols blocks (i*blockRows, i*blockCols, blockRows, blockCols).
    template <typename SparseMat>
    void fromBlockDiagonalPattern(const SparseMat& mat, StorageIndex blockRows, StorageIndex blockCols) {
        clear();
        nRows = (StorageIndex)mat.rows();
        nCols = (StorageIndex)mat.cols();
        const StorageIndex numBlocks = nCols / blockCols;
        for (StorageIndex i = 0; i < numBlocks; ++i) {
            Matrix b(blockRows, blockCols);
            for (StorageIndex c = 0; c < blockCols; ++c)
                for (StorageIndex r = 0; r < blockRows; ++r) b(r, c) = mat.coeff(i * blockRows + r, i * blockCols + c);
            insertBack(b);
        }
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
  protected:
    std::vector<int32_t> m_rows, m_cols;
    std::vector<double> m_tiles;
    StorageIndex nRows, nCols;
};

// tags standing in for the _BlockQRSolver template argument (BlockDiagonalSparseQR.h:37)
struct ColPivHouseholderQR { static const int kSolver = QRK_COLPIV_HOUSEHOLDER; };
struct HouseholderQR { static const int kSolver = QRK_HOUSEHOLDER; };

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
    ~BlockDiagonalSparseQR() { if (m_plan) qrk_bd_plan_destroy(m_plan); if (m_handle) qrk_destroy(m_handle); }
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
    // :415-547
    void factorize(const MatrixType& mat) {
        assert(m_analysisIsok && "analyzePattern() should be called first");
        int64_t tl, nq, nr;
        check(qrk_bd_plan_sizes(m_plan, &tl, &nq, &nr));
        const Index rows = mat.rows(), cols = mat.cols();
        m_Q.resize(rows, rows);
        m_R.resize(rows, cols);
        m_Q.values().assign((size_t)nq, 0.0); m_Q.innerIndex().assign((size_t)nq, 0);
        m_R.values().assign((size_t)nr, 0.0); m_R.innerIndex().assign((size_t)nr, 0);
        m_outputPerm_c.setIdentity(cols);
        check(qrk_bd_factorize(m_plan, mat.tiles().data(), m_Q.values().data(), m_R.values().data(),
                               m_outputPerm_c.indices().data(), 0, QRK_MEM_HOST));
        qrk_info info; int64_t rank;
        check(qrk_bd_info(m_plan, &info, &rank));
        m_info = (ComputationInfo)info;
        if (m_info != Success) return;   // :504-516: m_info = InvalidInput; return
        check(qrk_bd_pattern(m_plan, m_Q.outerIndex().data(), m_Q.innerIndex().data(), m_R.outerIndex().data(),
                             m_R.innerIndex().data(), QRK_MEM_HOST));
        m_nonzeropivots = rank;
        m_isInitialized = true;
        m_factorizationIsok = true;
    }

    Index rows() const { return m_R.rows(); }
    Index cols() const { return m_R.cols(); }
    const MatrixRType& matrixR() const { return m_R; }
    MatrixQType matrixQ() const { return m_Q; }   // by value, as the reference (:235-237)
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
        check(qrk_bd_solve(m_plan, m_Q.values().data(), m_R.values().data(), m_outputPerm_c.indices().data(), B.data(), nrhs,
                           dest.data(), QRK_MEM_HOST));
        m_info = Success;
        return true;
    }
    Vector solve(const Vector& B) const { Vector x; _solve_impl(B, x); return x; }
    // matrixQ().transpose() * B on the device (test/test-qrkit.cpp:187)
    Vector applyQt(const Vector& B) const {
        const int64_t nrhs = (int64_t)B.size() / rows();
        Vector y(B.size());
        check(qrk_bd_apply_qt(m_plan, m_Q.values().data(), B.data(), nrhs, y.data(), QRK_MEM_HOST));
        return y;
    }

  protected:
    void check(qrk_status st) const {
        if (st != QRK_STATUS_OK) throw std::runtime_error(std::string("qrkit: ") + qrk_last_error(m_handle));
    }
    mutable ComputationInfo m_info;
    MatrixRType m_R;
    MatrixQType m_Q;
    PermutationType m_outputPerm_c;
    PermutationType m_rowPerm;
    Index m_nonzeropivots;
    bool m_isInitialized, m_analysisIsok, m_factorizationIsok;
    qrk_handle m_handle;
    qrk_bd_plan m_plan;
};

}  // namespace qrkit

#endif  // QRKIT_FACADE_HPP
