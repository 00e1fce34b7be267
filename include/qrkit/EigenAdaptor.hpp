// qrkit/EigenAdaptor.hpp -- Eigen-typed front of the facade (the typedefs of
// src/QRKit/BlockDiagonalSparseQR.h:45-57 and the SparseSolverBase surface a QRKit user programs against).
//
// Compiled only with -DQRK_WITH_EIGEN and an Eigen >= 3.3 on the include path.  NOT COMPILED OR TESTED IN THIS REPOSITORY'S IMAGE: Eigen is
// not installed there (the reason the reference itself cannot be built); the header is kept to plain conversions so
// that there is little to get wrong, and INTEGRATION.md shows the two lines a reference user changes.  Everything numeric happens in
// QRKit.hpp's classes (device-resident factors behind the C ABI); this header only converts at the boundary:
//   Eigen::SparseMatrix<double, Major, int>  <->  qrkit::SparseMatrix<Major == RowMajor>
//   Eigen::MatrixXd / VectorXd               <->  qrkit::Matrix / Vector
//   Eigen::PermutationMatrix<Dynamic, Dynamic, int>  <-  qrkit::PermutationMatrix
#ifndef QRKIT_EIGEN_ADAPTOR_HPP
#define QRKIT_EIGEN_ADAPTOR_HPP
#ifdef QRK_WITH_EIGEN

#include <Eigen/Core>
#include <Eigen/SparseCore>

#include "qrkit/QRKit.hpp"

namespace qrkit {
namespace eigen {

typedef Eigen::PermutationMatrix<Eigen::Dynamic, Eigen::Dynamic, int> PermutationType;      // BlockDiagonalSparseQR.h:56

template <int Options>
inline SparseMatrix<(Options & Eigen::RowMajor) != 0> fromEigen(const Eigen::SparseMatrix<double, Options, int>& m) {
    Eigen::SparseMatrix<double, Options, int> c = m;
    c.makeCompressed();
    SparseMatrix<(Options & Eigen::RowMajor) != 0> out(c.rows(), c.cols());
    out.outerIndex().assign(c.outerIndexPtr(), c.outerIndexPtr() + c.outerSize() + 1);
    out.innerIndex().assign(c.innerIndexPtr(), c.innerIndexPtr() + c.nonZeros());
    out.values().assign(c.valuePtr(), c.valuePtr() + c.nonZeros());
    return out;
}
template <bool RowMajor>
inline Eigen::SparseMatrix<double, RowMajor ? Eigen::RowMajor : Eigen::ColMajor, int> toEigen(const SparseMatrix<RowMajor>& m) {
    typedef Eigen::SparseMatrix<double, RowMajor ? Eigen::RowMajor : Eigen::ColMajor, int> Out;
    return Out(Eigen::Map<const Out>(m.rows(), m.cols(), m.nonZeros(), m.outerIndex().data(), m.innerIndex().data(), m.values().data()));
}
inline Matrix fromEigen(const Eigen::MatrixXd& m) {
    Matrix out(m.rows(), m.cols());
    Eigen::Map<Eigen::MatrixXd>(out.data(), m.rows(), m.cols()) = m;
    return out;
}
inline Eigen::MatrixXd toEigen(const Matrix& m) { return Eigen::Map<const Eigen::MatrixXd>(m.data(), m.rows(), m.cols()); }
inline Vector fromEigen(const Eigen::VectorXd& v) { return Vector(v.data(), v.data() + v.size()); }
inline Eigen::VectorXd toEigen(const Vector& v) { return Eigen::Map<const Eigen::VectorXd>(v.data(), (Eigen::Index)v.size()); }
inline PermutationType toEigen(const PermutationMatrix& p) {
    PermutationType out((Eigen::Index)p.size());
    for (Index i = 0; i < p.size(); ++i) out.indices()(i) = p.indices()[(size_t)i];
    return out;
}

// The block-diagonal solver with Eigen types at its surface (BlockDiagonalSparseQR.h:45-57, 94-102, 161-165, 286-299): compute() from an
// Eigen sparse matrix with a block-diagonal pattern (fromBlockDiagonalPattern on the device, SparseBlockDiagonal.h:71-89) or from a
// qrkit::SparseBlockDiagonal; matrixQ() / matrixR() as Eigen sparse matrices, colsPermutation() / rowsPermutation(), rank(), info(),
// solve() for dense and sparse right-hand sides.
template <typename BlockQRSolver = ColPivHouseholderQR, int QFormat = 0>
class BlockDiagonalSparseQR {
  public:
    typedef Eigen::SparseMatrix<double, Eigen::RowMajor, int> MatrixQType;      // (:52)
    typedef Eigen::SparseMatrix<double, Eigen::ColMajor, int> MatrixRType;      // (:53)
    explicit BlockDiagonalSparseQR(int device = 0) : m_impl(device) {}
    void compute(const SparseBlockDiagonal& mat) { m_impl.compute(mat); }
    template <int Options>
    void compute(const Eigen::SparseMatrix<double, Options, int>& mat, StorageIndex blockRows, StorageIndex blockCols) {
        SparseBlockDiagonal bd;
        bd.fromBlockDiagonalPattern(fromEigen(mat), blockRows, blockCols);
        m_impl.compute(bd);
    }
    Index rows() const { return m_impl.rows(); }
    Index cols() const { return m_impl.cols(); }
    Index rank() const { return m_impl.rank(); }
    Eigen::ComputationInfo info() const { return m_impl.info() == Success ? Eigen::Success : Eigen::InvalidInput; }
    MatrixQType matrixQ() const { return toEigen(m_impl.matrixQ()); }
    MatrixRType matrixR() const { return toEigen(m_impl.matrixR()); }
    PermutationType colsPermutation() const { return toEigen(m_impl.colsPermutation()); }
    PermutationType rowsPermutation() const { return toEigen(m_impl.rowsPermutation()); }
    // rows x nrhs in, cols x nrhs out
    Eigen::MatrixXd solve(const Eigen::MatrixXd& B) const {
        Vector b(B.data(), B.data() + B.size());
        const Vector x = m_impl.solve(b);
        return Eigen::Map<const Eigen::MatrixXd>(x.data(), m_impl.cols(), B.cols());
    }
    template <int Options>
    Eigen::SparseMatrix<double, Eigen::ColMajor, int> solve(const Eigen::SparseMatrix<double, Options, int>& B) const {
        return toEigen(m_impl.solve(fromEigen(B)));
    }
    const qrkit::BlockDiagonalSparseQR<BlockQRSolver, QFormat>& impl() const { return m_impl; }
  private:
    qrkit::BlockDiagonalSparseQR<BlockQRSolver, QFormat> m_impl;
};

}  // namespace eigen
}  // namespace qrkit

#endif  // QRK_WITH_EIGEN
#endif  // QRKIT_EIGEN_ADAPTOR_HPP
