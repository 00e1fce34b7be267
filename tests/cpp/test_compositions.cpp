// C++ parity tests of the compositions through the facade (include/qrkit/QRKit.hpp):
//   test_banded_blocked  (test/test-qrkit.cpp:208-258) on the reference's three banded inputs
//                        (main(), :369-384: block diagonal, overlapping, overlapping + shuffled rows),
//   test_block_angular   (:260-293) on the reference's block-angular input (:386-395) with the banded
//                        left solver it uses (typedefs :43-48), and with a BlockDiagonalSparseQR left
//                        solver on a block-diagonal left part (BASELINE configs[3] shape).
// The reference checks at 1e-6 (test/test.h:31); the bars here are 1e-10 / 1e-8.
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <cstdlib>
#include <numeric>
#include <random>

#include "qrkit/QRKit.hpp"
#include "thin_sparse_fixture.h"

using namespace qrkit;

typedef BandedBlockedSparseQR<8> BandedBlockedQRSolver;          // SuggestedBlockCols = 8 (test-qrkit.cpp:44)

static double frob(const Matrix& a) { double s = 0; for (Index i = 0; i < a.rows() * a.cols(); ++i) s += a.data()[i] * a.data()[i]; return std::sqrt(s); }
static bool approx(const Matrix& a, const Matrix& b, double prec) {   // Eigen isApprox
    Matrix d(a.rows(), a.cols());
    for (Index i = 0; i < a.rows() * a.cols(); ++i) d.data()[i] = a.data()[i] - b.data()[i];
    return frob(d) <= prec * std::min(frob(a), frob(b));
}
static bool approxVec(const Vector& a, const Vector& b, double prec) {
    double d = 0, na = 0, nb = 0;
    for (size_t i = 0; i < a.size(); ++i) { d += (a[i] - b[i]) * (a[i] - b[i]); na += a[i] * a[i]; nb += b[i] * b[i]; }
    return std::sqrt(d) <= prec * std::sqrt(std::min(na, nb));
}
static Matrix matmul(const Matrix& a, const Matrix& b, bool transA) {
    const Index m = transA ? a.cols() : a.rows(), k = transA ? a.rows() : a.cols(), n = b.cols();
    Matrix c(m, n);
    for (Index j = 0; j < n; ++j) for (Index p = 0; p < k; ++p) { const double bv = b(p, j); if (bv == 0) continue;
        for (Index i = 0; i < m; ++i) c(i, j) += (transA ? a(p, i) : a(i, p)) * bv; }
    return c;
}
static Matrix identity(Index n) { Matrix I(n, n); for (Index i = 0; i < n; ++i) I(i, i) = 1.0; return I; }

// generate_block_diagonal_matrix / generate_overlapping_block_diagonal_matrix (test-qrkit.cpp:62-128)
static void generate_banded(Index numParams, Index numResiduals, bool overlap, int shuffleSeed, SparseMatrixColMajor& spJ) {
    std::default_random_engine gen;
    std::uniform_real_distribution<double> dist(0.5, 5.0);
    const int stride = 7;
    std::vector<Triplet> jvals;
    for (int i = 0; i < numParams; i++)
        for (int j = i * 2; j < (i * 2) + 2 && j < numParams; j++) {
            for (int r = 0; r < 7; ++r) jvals.emplace_back(i * stride + r, j, dist(gen));
            if (overlap && j < numParams - 2) jvals.emplace_back(i * stride + 6, j + 2, dist(gen));
        }
    if (shuffleSeed) {   // spJ = perm * spJ with a random row permutation
        std::vector<int> perm((size_t)numResiduals);
        std::iota(perm.begin(), perm.end(), 0);
        std::mt19937 rng((unsigned)shuffleSeed);
        std::shuffle(perm.begin(), perm.end(), rng);
        for (Triplet& t : jvals) t.row = perm[(size_t)t.row];
    }
    spJ.resize(numResiduals, numParams);
    spJ.setFromTriplets(jvals);
}

// generate_block_angular_matrix (:132-165)
static void generate_block_angular(Index numParams, Index numAngularParams, Index numResiduals, bool overlap, SparseMatrixColMajor& left, Matrix& right) {
    std::default_random_engine gen;
    std::uniform_real_distribution<double> dist(0.5, 5.0);
    const int stride = 7;
    std::vector<Triplet> jvals;
    for (int i = 0; i < numParams; i++)
        for (int j = i * 2; j < (i * 2) + 2 && j < numParams; j++) {
            for (int r = 0; r < 7; ++r) jvals.emplace_back(i * stride + r, j, dist(gen));
            if (overlap && j < numParams - 2) jvals.emplace_back(i * stride + 6, j + 2, dist(gen));
        }
    right = Matrix(numResiduals, numAngularParams);
    for (Index i = 0; i < numResiduals; i++) for (Index j = 0; j < numAngularParams; j++) right(i, j) = dist(gen);
    left.resize(numResiduals, numParams);
    left.setFromTriplets(jvals);
}

// solve() with a sparse right-hand side (the SparseMatrixBase overload of every reference solver): five columns -- b, -2 b, an empty
// one, b again, 0.5 b -- so the panels of four and the remainder are both exercised; the result must be the dense solve() of every
// column with the exact zeros dropped (an empty column stays empty).
template <typename Solver>
static int checkSparseRhs(const Solver& dec, const Vector& b, const char* what) {
    const Index rows = (Index)b.size();
    const double scale[5] = {1.0, -2.0, 0.0, 1.0, 0.5};
    std::vector<Triplet> trips;
    for (int c = 0; c < 5; ++c) for (Index i = 0; i < rows; ++i) if (scale[c] != 0.0 && b[(size_t)i] != 0.0) trips.push_back(Triplet((int)i, c, scale[c] * b[(size_t)i]));
    SparseMatrixColMajor Bc(rows, 5);
    Bc.setFromTriplets(trips);
    SparseMatrixRowMajor Br(rows, 5);
    Br.setFromTriplets(trips);
    const Vector xd = dec.solve(b);
    int fails = 0;
    for (int pass = 0; pass < 2; ++pass) {
        const SparseMatrixColMajor X = pass ? dec.solve(Br) : dec.solve(Bc);
        if (X.rows() != (Index)xd.size() || X.cols() != 5) { std::printf("  %s: sparse solve() has the wrong shape\n", what); return 1; }
        if (X.outerIndex()[3] != X.outerIndex()[2]) { std::printf("  %s: the empty column did not stay empty\n", what); ++fails; }
        for (int c = 0; c < 5; ++c) {
            Vector col(xd.size(), 0.0);
            for (int p = X.outerIndex()[(size_t)c]; p < X.outerIndex()[(size_t)c + 1]; ++p) {
                if (X.values()[(size_t)p] == 0.0) { std::printf("  %s: explicit zero kept\n", what); ++fails; }
                col[(size_t)X.innerIndex()[(size_t)p]] = X.values()[(size_t)p];
            }
            double num = 0.0, den = 0.0;
            for (size_t i = 0; i < xd.size(); ++i) { const double d = col[i] - scale[c] * xd[i]; num += d * d; den += xd[i] * xd[i]; }
            if (std::sqrt(num) > 1e-10 * std::sqrt(den)) { std::printf("  %s: column %d of the sparse solve() differs from the dense one\n", what, c); ++fails; }
        }
    }
    return fails;
}

static int test_banded_blocked(const SparseMatrixColMajor& spJ, const char* name) {
    int fails = 0;
    BandedBlockedQRSolver slvr;
    slvr.compute(spJ);
    const Index rows = spJ.rows(), cols = spJ.cols();
    const Matrix I = identity(rows);
    const Matrix slvrQ = slvr.matrixQ() * I;                          // Q * I      (:221-222)
    const Matrix slvrQt = slvr.matrixQ().transpose() * I;             // Q.T * I    (:224-225)

    std::mt19937_64 rng(7);
    std::uniform_real_distribution<double> ud(-1.0, 1.0);
    Vector x((size_t)cols);
    for (double& v : x) v = ud(rng);
    Vector b = slvr.rowsPermutation() * (spJ * x);                    // (:229-233)
    const Vector y = slvr.matrixQ().transpose() * b;
    const Vector solved = solveUpperCsc(slvr.matrixR(), cols, y);     // R.topLeftCorner.triangularView<Upper>().solve (:239)
    Vector backperm((size_t)cols, 0.0);
    for (Index i = 0; i < cols; ++i) backperm[(size_t)slvr.colsPermutation().indices()[(size_t)i]] = solved[(size_t)i];

    // spJRowPerm = rowsPermutation() * spJ (:246)
    const Matrix J = spJ.toDense();
    Matrix JP(rows, cols);
    for (Index i = 0; i < rows; ++i) for (Index j = 0; j < cols; ++j) JP(slvr.rowsPermutation().indices()[(size_t)i], j) = J(i, j);
    const Matrix Rd = slvr.matrixR().toDense();
    if (!approx(matmul(slvrQ, Rd, false), JP, 1e-10)) { std::printf("  Q*R != P*J\n"); ++fails; }                 // (:249)
    if (!approx(matmul(slvrQ, JP, true), Rd, 1e-10)) { std::printf("  Q^T*P*J != R\n"); ++fails; }               // (:250)
    if (!approx(matmul(slvrQt, Rd, true), JP, 1e-10)) { std::printf("  (Q^T)^T*R != P*J\n"); ++fails; }          // (:251)
    if (!approx(matmul(slvrQt, JP, false), Rd, 1e-10)) { std::printf("  (Q^T)*P*J != R\n"); ++fails; }           // (:252)
    if (!approxVec(x, backperm, 1e-8)) { std::printf("  LS recovery failed\n"); ++fails; }                        // (:253)
    if (!approxVec(x, slvr.solve(b), 1e-8)) { std::printf("  solve() recovery failed\n"); ++fails; }
    fails += checkSparseRhs(slvr, b, "BandedBlockedSparseQR");
    std::printf("test_banded_blocked [%s] %lld blocks: %s\n", name, (long long)slvr.numBlocks(), fails ? "Failed." : "Passed.");
    return fails;
}

// The strips form of the banded solver behind the facade (BandedStripsSparseQR<BlockRows, BlockCols, BlockOverlap>, qrk_bbs_*): the
// reference's own invariants of test_banded_blocked (test-qrkit.cpp:249-253) on a block-banded matrix of N dense strips, solve() with
// dense and sparse right-hand sides, and R against the CSR entry's (the reference's elimination order) after aligning the sign of
// each row -- for a fixed column order R is unique up to row signs.  N >= 64 runs the chains of solve() / Q products in two levels.
template <int BR, int BC, int BO>
static int test_banded_strips(Index N, const char* name) {
    int fails = 0;
    const Index step = BC - BO, rows = N * BR, cols = (N - 1) * step + BC;
    std::mt19937_64 rng(11 + (unsigned)N);
    std::uniform_real_distribution<double> ud(-1.0, 1.0);
    std::vector<Triplet> t;
    for (Index i = 0; i < N; ++i)
        for (Index c = 0; c < BC; ++c)
            for (Index r = 0; r < BR; ++r) { const double v = ud(rng); t.emplace_back((int)(i * BR + r), (int)(i * step + c), v + (v < 0 ? -0.25 : 0.25)); }
    SparseMatrixColMajor spJ(rows, cols);
    spJ.setFromTriplets(t);
    BandedStripsSparseQR<BR, BC, BO> slvr;
    slvr.compute(spJ);
    if (slvr.info() != Success) { std::printf("  info() != Success\n"); ++fails; }
    const Matrix J = spJ.toDense(), Rd = slvr.matrixR().toDense();
    Vector x((size_t)cols);
    for (double& v : x) v = ud(rng);
    const Vector b = spJ * x;
    if (rows <= 1200) {
        const Matrix I = identity(rows);
        const Matrix Q = slvr.matrixQ() * I, Qt = slvr.matrixQ().transpose() * I;
        if (!approx(matmul(Q, Rd, false), J, 1e-10)) { std::printf("  Q*R != J\n"); ++fails; }
        if (!approx(matmul(Q, J, true), Rd, 1e-10)) { std::printf("  Q^T*J != R\n"); ++fails; }
        if (!approx(matmul(Qt, Rd, true), J, 1e-10)) { std::printf("  (Q^T)^T*R != J\n"); ++fails; }
        if (!approx(matmul(Qt, J, false), Rd, 1e-10)) { std::printf("  (Q^T)*J != R\n"); ++fails; }
    } else {
        const Vector y = slvr.matrixQ().transpose() * b, back = slvr.matrixQ() * y;
        if (!approxVec(back, b, 1e-12)) { std::printf("  Q Q^T b != b\n"); ++fails; }
        double tail = 0, all = 0;
        for (size_t i = 0; i < y.size(); ++i) { all += y[i] * y[i]; if ((Index)i >= cols) tail += y[i] * y[i]; }
        if (std::sqrt(tail) > 1e-11 * std::sqrt(all)) { std::printf("  Q^T (J x) leaves the range of R\n"); ++fails; }
    }
    const Vector y = slvr.matrixQ().transpose() * b;
    if (!approxVec(x, solveUpperCsc(slvr.matrixR(), cols, y), 1e-8)) { std::printf("  LS recovery failed\n"); ++fails; }
    if (!approxVec(x, slvr.solve(b), 1e-8)) { std::printf("  solve() recovery failed\n"); ++fails; }
    fails += checkSparseRhs(slvr, b, "BandedStripsSparseQR");
    {   // the same matrix through the CSR entry in the reference's order
        BandedBlockedSparseQR<8> ref;
        ref.compute(spJ);
        const Matrix Rr = ref.matrixR().toDense();
        double worst = 0.0;
        for (Index i = 0; i < cols; ++i) {
            const double sg = (Rd(i, i) < 0) == (Rr(i, i) < 0) ? 1.0 : -1.0;
            double d = 0, nr = 0;
            for (Index j = i; j < cols; ++j) { const double e = sg * Rd(i, j) - Rr(i, j); d += e * e; nr += Rr(i, j) * Rr(i, j); }
            worst = std::max(worst, std::sqrt(d / nr));
        }
        if (!(worst <= 1e-10)) { std::printf("  R differs from the CSR entry's beyond row signs: %.2e\n", worst); ++fails; }
    }
    // an entry outside its strip: InvalidInput, nothing factorised
    {
        std::vector<Triplet> t2 = t;
        if (N >= 3) {
            t2.emplace_back((int)0, (int)(cols - 1), 1.0);
            SparseMatrixColMajor bad(rows, cols);
            bad.setFromTriplets(t2);
            BandedStripsSparseQR<BR, BC, BO> s2;
            s2.compute(bad);
            if (s2.info() != InvalidInput) { std::printf("  an entry outside the band was accepted\n"); ++fails; }
        }
    }
    std::printf("test_banded_strips [%s] %lld strips of %d x %d, overlap %d: %s\n", name, (long long)N, BR, BC, BO, fails ? "Failed." : "Passed.");
    return fails;
}

template <typename Solver, typename LeftMat, typename RightMat>
static int test_block_angular_as(const LeftMat& leftForSolver, const SparseMatrixColMajor& leftSparse, const Matrix& right,
                                 const RightMat& rightForSolver, const char* name);
template <typename Solver, typename LeftMat>
static int test_block_angular(const LeftMat& leftForSolver, const SparseMatrixColMajor& leftSparse, const Matrix& right, const char* name) {
    return test_block_angular_as<Solver>(leftForSolver, leftSparse, right, right, name);
}
template <typename Solver, typename LeftMat, typename RightMat>
static int test_block_angular_as(const LeftMat& leftForSolver, const SparseMatrixColMajor& leftSparse, const Matrix& right,
                                 const RightMat& rightForSolver, const char* name) {
    int fails = 0;
    Solver baqr;
    BlockMatrix1x2<LeftMat, RightMat> blkAngular(leftForSolver, rightForSolver);
    baqr.compute(blkAngular);
    const Index rows = right.rows(), m1 = leftSparse.cols(), m2 = right.cols(), cols = m1 + m2;
    if (baqr.info() != Success || baqr.rank() != cols) { std::printf("  info/rank wrong\n"); ++fails; }
    std::mt19937_64 rng(11);
    std::uniform_real_distribution<double> ud(-1.0, 1.0);
    Vector x((size_t)cols);
    for (double& v : x) v = ud(rng);
    // b = spJ * x with spJ = [left | right]
    Vector xl(x.begin(), x.begin() + m1);
    Vector b = leftSparse * xl;
    for (Index j = 0; j < m2; ++j) for (Index i = 0; i < rows; ++i) b[(size_t)i] += right(i, j) * x[(size_t)(m1 + j)];
    b = baqr.rowsPermutation() * b;                                                    // (:275)
    const Vector y = baqr.matrixQ().transpose() * b;                                   // (:279)
    const Vector solved = solveUpperCsc(baqr.matrixR(), cols, y);                      // (:283)
    Vector backperm((size_t)cols, 0.0);
    for (Index i = 0; i < cols; ++i) backperm[(size_t)baqr.colsPermutation().indices()[(size_t)i]] = solved[(size_t)i];   // (:286-288)
    if (!approxVec(x, backperm, 1e-8)) { std::printf("  LS recovery failed\n"); ++fails; }                    // (:290)
    if (!approxVec(x, baqr.solve(b), 1e-8)) { std::printf("  solve() recovery failed\n"); ++fails; }
    fails += checkSparseRhs(baqr, b, "BlockAngularSparseQR");
    // Q Q^T b = b and |Q^T b| = |b|
    const Vector back = baqr.matrixQ() * y;
    if (!approxVec(b, back, 1e-10)) { std::printf("  Q*(Q^T*b) != b\n"); ++fails; }
    // Q^T [J1 | J2] P = R on a few columns: column c of J*P
    const Matrix Rd = baqr.matrixR().toDense();
    for (Index c : {Index(0), m1 / 2, m1, cols - 1}) {
        const Index src = baqr.colsPermutation().indices()[(size_t)c];
        Vector col((size_t)rows, 0.0);
        if (src < m1) { Vector e((size_t)m1, 0.0); e[(size_t)src] = 1.0; col = leftSparse * e; }
        else for (Index i = 0; i < rows; ++i) col[(size_t)i] = right(i, src - m1);
        col = baqr.rowsPermutation() * col;
        const Vector qc = baqr.matrixQ().transpose() * col;
        Vector rc((size_t)rows);
        for (Index i = 0; i < rows; ++i) rc[(size_t)i] = Rd(i, c);
        double d = 0, n = 0;
        for (Index i = 0; i < rows; ++i) { d += (qc[(size_t)i] - rc[(size_t)i]) * (qc[(size_t)i] - rc[(size_t)i]); n += rc[(size_t)i] * rc[(size_t)i]; }
        if (std::sqrt(d) > 1e-10 * std::sqrt(n)) { std::printf("  Q^T (J P)(:,%lld) != R(:,%lld)\n", (long long)c, (long long)c); ++fails; }
    }
    std::printf("test_block_angular [%s] %lld + %lld columns: %s\n", name, (long long)m1, (long long)m2, fails ? "Failed." : "Passed.");
    return fails;
}

// BlockedThinDenseQR on its own: the solver concept of BlockedThinQRBase.h:100-222 (compute, matrixR, matrixQ products,
// identity permutations, rank, solve) checked with the invariants the reference checks on every solver (:201-203).
static int test_blocked_thin(const Matrix& A) {
    int fails = 0;
    BlockedThinDenseQR<2> slvr;
    slvr.compute(A);
    const Index rows = A.rows(), cols = A.cols();
    if (slvr.info() != Success || slvr.rank() != cols) { std::printf("  info/rank wrong\n"); ++fails; }
    for (Index j = 0; j < cols; ++j) if (slvr.colsPermutation().indices()[(size_t)j] != j) { std::printf("  column permutation is not the identity\n"); ++fails; break; }
    const Matrix& R = slvr.matrixR();
    for (Index j = 0; j < cols && !fails; ++j) for (Index i = j + 1; i < rows; ++i) if (R(i, j) != 0.0) { std::printf("  R is not upper triangular\n"); ++fails; break; }
    const Matrix QtA = slvr.matrixQ().transpose() * A;        // Q^T A = R
    if (!approx(QtA, R, 1e-12)) { std::printf("  Q^T*A != R\n"); ++fails; }
    const Matrix QR = slvr.matrixQ() * R;                     // Q R = A
    if (!approx(QR, A, 1e-12)) { std::printf("  Q*R != A\n"); ++fails; }
    std::mt19937_64 rng(5);
    std::uniform_real_distribution<double> ud(-1.0, 1.0);
    Vector x((size_t)cols);
    for (double& v : x) v = ud(rng);
    Vector b((size_t)rows, 0.0);
    for (Index j = 0; j < cols; ++j) for (Index i = 0; i < rows; ++i) b[(size_t)i] += A(i, j) * x[(size_t)j];
    if (!approxVec(x, slvr.solve(b), 1e-8)) { std::printf("  LS recovery failed\n"); ++fails; }
    fails += checkSparseRhs(slvr, b, "BlockedThin*QR");
    std::printf("test_blocked_thin %lld x %lld: %s\n", (long long)rows, (long long)cols, fails ? "Failed." : "Passed.");
    return fails;
}

// BlockedThinSparseQR on its own (BlockedThinSparseQR.h:105-283): ColumnDensity / as-banded-as-possible orderings, per-panel
// column pivoting, rank.  Invariants on any input: Q^T (Pr A Pc) = R upper triangular, Q (Q^T b) = b; with full rank also the
// least-squares solution through both permutations; with the committed rank-deficient fixture the oracle's permutations and rank.
static int test_blocked_thin_sparse(const SparseMatrixColMajor& A, const int* wantColPerm, const int* wantRowPerm, int wantRank, const char* name) {
    int fails = 0;
    BlockedThinSparseQR<2> slvr;
    slvr.compute(A);
    const Index rows = A.rows(), cols = A.cols();
    if (slvr.info() != Success) { std::printf("  info wrong\n"); ++fails; }
    if (wantRank >= 0 && slvr.rank() != wantRank) { std::printf("  rank %lld, want %d\n", (long long)slvr.rank(), wantRank); ++fails; }
    if (wantColPerm) for (Index j = 0; j < cols; ++j) if (slvr.colsPermutation().indices()[(size_t)j] != wantColPerm[j]) { std::printf("  column permutation differs from the oracle's at %lld\n", (long long)j); ++fails; break; }
    if (wantRowPerm) for (Index i = 0; i < rows; ++i) if (slvr.rowsPermutation().indices()[(size_t)i] != wantRowPerm[i]) { std::printf("  row permutation differs from the oracle's at %lld\n", (long long)i); ++fails; break; }
    // Pr A Pc, dense: row i of A goes to row rowPerm[i]; column j of the product is column colPerm[j] of A
    Matrix PAP(rows, cols);
    for (Index j = 0; j < cols; ++j) {
        const int src = slvr.colsPermutation().indices()[(size_t)j];
        for (int e = A.outerIndex()[(size_t)src]; e < A.outerIndex()[(size_t)src + 1]; ++e)
            PAP(slvr.rowsPermutation().indices()[(size_t)A.innerIndex()[(size_t)e]], j) = A.values()[(size_t)e];
    }
    const Matrix& R = slvr.matrixR();
    for (Index j = 0; j < cols && !fails; ++j) for (Index i = j + 1; i < rows; ++i) if (R(i, j) != 0.0) { std::printf("  R is not upper triangular\n"); ++fails; break; }
    // (the zero-pivot columns come last and carry no column of R: the leading rank() columns are the factorisation)
    const Index rk = slvr.rank();
    Matrix PAPr(rows, rk), Rr(rows, rk);
    for (Index j = 0; j < rk; ++j) for (Index i = 0; i < rows; ++i) { PAPr(i, j) = PAP(i, j); Rr(i, j) = R(i, j); }
    const Matrix QtA = slvr.matrixQ().transpose() * PAPr;
    if (!approx(QtA, Rr, 1e-12)) { std::printf("  Q^T*(Pr A Pc) != R\n"); ++fails; }
    const Matrix QR = slvr.matrixQ() * Rr;
    if (!approx(QR, PAPr, 1e-12)) { std::printf("  Q*R != Pr A Pc\n"); ++fails; }
    if (slvr.rank() == cols) {
        std::mt19937_64 rng(9);
        std::uniform_real_distribution<double> ud(-1.0, 1.0);
        Vector x((size_t)cols);
        for (double& v : x) v = ud(rng);
        Vector b = A * x, pb((size_t)rows);
        for (Index i = 0; i < rows; ++i) pb[(size_t)slvr.rowsPermutation().indices()[(size_t)i]] = b[(size_t)i];
        const Vector z = slvr.solve(pb);                  // solves (Pr A Pc) z = Pr b
        Vector xs((size_t)cols);
        for (Index j = 0; j < cols; ++j) xs[(size_t)slvr.colsPermutation().indices()[(size_t)j]] = z[(size_t)j];
        if (!approxVec(x, xs, 1e-8)) { std::printf("  LS recovery failed\n"); ++fails; }
    }
    std::printf("test_blocked_thin_sparse [%s] %lld x %lld, rank %lld: %s\n", name, (long long)rows, (long long)cols, (long long)slvr.rank(), fails ? "Failed." : "Passed.");
    return fails;
}

int main() {
    int fails = 0;
    {   // main(), test-qrkit.cpp:363-384
        const Index numVars = 256, numParams = numVars * 2, numResiduals = numVars * 3 + numVars + numVars * 3;
        SparseMatrixColMajor spJ;
        generate_banded(numParams, numResiduals, false, 0, spJ);
        fails += test_banded_blocked(spJ, "block diagonal");
        generate_banded(numParams, numResiduals, true, 0, spJ);
        fails += test_banded_blocked(spJ, "overlapping");
        generate_banded(numParams, numResiduals, true, 5, spJ);
        fails += test_banded_blocked(spJ, "overlapping, rows shuffled");
    }
    {   // the strips form behind the facade: few strips (one-level chains), 70 strips (two levels)
        fails += test_banded_strips<64, 48, 32>(9, "one level");
        fails += test_banded_strips<64, 48, 32>(70, "two levels");
    }
    {   // :386-395 at a quarter of the reference's size (numVars = 1024, 384 angular parameters there)
        const Index numVars = 256, numParams = numVars * 2, numResiduals = numVars * 3 + numVars + numVars * 3, numAngular = 96;
        SparseMatrixColMajor left; Matrix right;
        generate_block_angular(numParams, numAngular, numResiduals, true, left, right);
        fails += test_block_angular<BlockAngularSparseQR<BandedBlockedQRSolver, ColPivHouseholderQR> >(left, left, right, "banded left solver");
        // block-diagonal left part (7x2 tiles) through BlockDiagonalSparseQR
        generate_block_angular(numParams, numAngular, numResiduals, false, left, right);
        SparseBlockDiagonal blk;
        blk.fromBlockDiagonalPattern(left, 7, 2);
        fails += test_block_angular<BlockAngularSparseQR<BlockDiagonalSparseQR<ColPivHouseholderQR>, ColPivHouseholderQR> >(blk, left, right, "block-diagonal left solver");
    }
    {   // test_block_angular_denseblocked / _denseblocked_sparse (:294-362, typedefs :54-58): thin right solvers
        const Index numVars = 256, numParams = numVars * 2, numResiduals = numVars * 3 + numVars + numVars * 3, numAngular = 96;
        SparseMatrixColMajor left; Matrix right;
        generate_block_angular(numParams, numAngular, numResiduals, true, left, right);
        fails += test_block_angular<BlockAngularSparseQR<BandedBlockedQRSolver, BlockedThinDenseQR<2> > >(left, left, right, "banded left, BlockedThinDenseQR right");
        std::vector<Triplet> rt;
        for (Index j = 0; j < right.cols(); ++j) for (Index i = 0; i < right.rows(); ++i) rt.emplace_back((int)i, (int)j, right(i, j));
        SparseMatrixColMajor rightSparse(right.rows(), right.cols());
        rightSparse.setFromTriplets(rt);
        fails += test_block_angular_as<BlockAngularSparseQR<BandedBlockedQRSolver, BlockedThinSparseQR<2> > >(left, left, right, rightSparse, "banded left, BlockedThinSparseQR right (sparse right block)");
        {   // the same right block row-major, and one with two thirds of its entries absent (the window kernel's zero fill)
            SparseMatrixRowMajor rightRM(right.rows(), right.cols());
            rightRM.setFromTriplets(rt);
            fails += test_block_angular_as<BlockAngularSparseQR<BandedBlockedQRSolver, BlockedThinSparseQR<2> > >(left, left, right, rightRM, "banded left, row-major sparse right block");
            Matrix thin(right.rows(), right.cols());
            std::vector<Triplet> tt;
            for (Index j = 0; j < right.cols(); ++j)
                for (Index i = 0; i < right.rows(); ++i)
                    if ((i + 2 * j) % 3 == 0) { thin(i, j) = right(i, j); tt.emplace_back((int)i, (int)j, right(i, j)); }
            SparseMatrixColMajor thinCM(right.rows(), right.cols());
            SparseMatrixRowMajor thinRM(right.rows(), right.cols());
            thinCM.setFromTriplets(tt); thinRM.setFromTriplets(tt);
            fails += test_block_angular_as<BlockAngularSparseQR<BandedBlockedQRSolver, BlockedThinSparseQR<2> > >(left, left, thin, thinCM, "banded left, right block with 1/3 of its entries (CSC)");
            fails += test_block_angular_as<BlockAngularSparseQR<BandedBlockedQRSolver, BlockedThinSparseQR<2> > >(left, left, thin, thinRM, "banded left, right block with 1/3 of its entries (CSR)");
        }
        fails += test_blocked_thin(right);
        // the sparse thin solver as the reference defines it: full rank (a third of the right block's entries), then the committed
        // rank-deficient fixture with the oracle's permutations and rank
        {
            std::vector<Triplet> tt;
            for (Index j = 0; j < right.cols(); ++j)
                for (Index i = 0; i < right.rows(); ++i)
                    if ((i + 2 * j) % 3 == 0) tt.emplace_back((int)i, (int)j, right(i, j));
            SparseMatrixColMajor thinCM(right.rows(), right.cols());
            thinCM.setFromTriplets(tt);
            fails += test_blocked_thin_sparse(thinCM, 0, 0, (int)right.cols(), "a third of the right block");
            SparseMatrixColMajor fx(kThinRows, kThinCols);
            fx.outerIndex().assign(kThinColPtr, kThinColPtr + kThinCols + 1);
            fx.innerIndex().assign(kThinRowIdx, kThinRowIdx + kThinColPtr[kThinCols]);
            fx.values().assign(kThinVals, kThinVals + kThinColPtr[kThinCols]);
            fails += test_blocked_thin_sparse(fx, kThinColPerm, kThinRowPerm, kThinRank, "rank-deficient fixture (oracle.bt_sparse_qr)");
        }
    }
    if (std::getenv("QRK_BIG")) {
        // BASELINE configs[3] through the facade: 20000 tiles of 8x6 + 2000 dense columns, host matrices in, solution out.
        // J2 (2.56 GB) crosses PCIe once; Q1^T J2, the strip of R and the packed right factor stay on the device.
        const bool small = std::strcmp(std::getenv("QRK_BIG"), "small") == 0;        // (a quarter of the shape, for diagnostics)
        const Index B = small ? 5000 : 20000, r = 8, c = 6, m2 = small ? 504 : 2000, n1 = B * r, m1 = B * c;
        std::mt19937_64 rng(3);
        std::uniform_real_distribution<double> ud(0.5, 5.0);
        SparseBlockDiagonal blk(n1, m1);
        Matrix tile(r, c);
        for (Index b = 0; b < B; ++b) { for (Index e = 0; e < r * c; ++e) tile.data()[e] = ud(rng); blk.insertBack(tile); }
        Matrix right(n1, m2);
        for (Index e = 0; e < n1 * m2; ++e) right.data()[e] = ud(rng);
        Vector x((size_t)(m1 + m2));
        for (double& v : x) v = ud(rng) - 2.75;
        Vector bvec((size_t)n1, 0.0);
        {   // b = [J1 | J2] x
            size_t off = 0;
            for (Index b = 0; b < B; ++b, off += (size_t)(r * c))
                for (Index j = 0; j < c; ++j) for (Index i = 0; i < r; ++i) bvec[(size_t)(b * r + i)] += blk.tiles()[off + (size_t)(j * r + i)] * x[(size_t)(b * c + j)];
            for (Index j = 0; j < m2; ++j) { const double xv = x[(size_t)(m1 + j)]; const double* col = right.data() + j * n1; for (Index i = 0; i < n1; ++i) bvec[(size_t)i] += col[i] * xv; }
        }
        BlockAngularSparseQR<BlockDiagonalSparseQR<ColPivHouseholderQR>, ColPivHouseholderQR> baqr;
        BlockMatrix1x2<SparseBlockDiagonal, Matrix> mat(blk, right);
        baqr.compute(mat);                               // (first call: plans, allocations)
        const auto t0 = std::chrono::steady_clock::now();
        baqr.compute(mat);
        const auto t1 = std::chrono::steady_clock::now();
        const Vector xs = baqr.solve(bvec);
        const auto t2 = std::chrono::steady_clock::now();
        const bool ok = approxVec(x, xs, 1e-8);
        std::printf("configs[3] through the facade (%lld x (8x6) + %lld dense, host matrices in): compute %.1f ms, solve %.1f ms: %s\n",
                    (long long)B, (long long)m2, std::chrono::duration<double, std::milli>(t1 - t0).count(),
                    std::chrono::duration<double, std::milli>(t2 - t1).count(), ok ? "Passed." : "Failed.");
        if (!ok) ++fails;
        // the same shape with a sparse right block, bundle-adjustment style: every row sees one of 222 cameras, 9 nonzeros in
        // that camera's columns (1.44 M nonzeros instead of 320 M entries cross PCIe; the dense copy is written on the device)
        {
            std::vector<Triplet> tt;
            tt.reserve((size_t)n1 * 9);
            const Index ncam = m2 / 9;
            for (Index i = 0; i < n1; ++i) {
                const Index cam = (Index)((uint64_t)i * 2654435761ull % (uint64_t)ncam);
                for (Index q = 0; q < 9; ++q) tt.emplace_back((int)i, (int)(9 * cam + q), ud(rng));
            }
            for (Index j = 9 * ncam; j < m2; ++j) for (Index i = j; i < n1; i += 997) tt.emplace_back((int)i, (int)j, ud(rng));
            SparseMatrixRowMajor rs(n1, m2);
            rs.setFromTriplets(tt);
            Vector x2(x.begin() + m1, x.end());
            Vector b2 = rs * x2;
            {
                size_t off = 0;
                for (Index b = 0; b < B; ++b, off += (size_t)(r * c))
                    for (Index j = 0; j < c; ++j) for (Index i = 0; i < r; ++i) b2[(size_t)(b * r + i)] += blk.tiles()[off + (size_t)(j * r + i)] * x[(size_t)(b * c + j)];
            }
            BlockMatrix1x2<SparseBlockDiagonal, SparseMatrixRowMajor> smat(blk, rs);
            baqr.compute(smat);
            const auto s0 = std::chrono::steady_clock::now();
            baqr.compute(smat);
            const auto s1 = std::chrono::steady_clock::now();
            const Vector xs2 = baqr.solve(b2);
            const auto s2 = std::chrono::steady_clock::now();
            const bool ok2 = approxVec(x, xs2, 1e-7);
            std::printf("configs[3] through the facade, sparse right block (%lld nonzeros): compute %.1f ms, solve %.1f ms: %s\n",
                        (long long)rs.nonZeros(), std::chrono::duration<double, std::milli>(s1 - s0).count(),
                        std::chrono::duration<double, std::milli>(s2 - s1).count(), ok2 ? "Passed." : "Failed.");
            if (!ok2) ++fails;
        }
    }
    {   // the reference's own spelling: namespace QRKit, four template parameters, a fixed-size block type and a block overlap
        // (typedef at test-qrkit.cpp:43-44 with <Matrix<double,7,4>> / overlap 2 would take the fixed-pattern analysis,
        // BandedBlockedSparseQR.h:398-408).  On the un-shuffled overlapping matrix both analyses must give the same block map
        // (the reference's known answers, test-utils.cpp:228-241) and therefore the same R.
        const Index numVars = 256, numParams = numVars * 2, numResiduals = numVars * 3 + numVars + numVars * 3;
        SparseMatrixColMajor spJ;
        generate_banded(numParams, numResiduals, true, 0, spJ);
        QRKit::BandedBlockedSparseQR<QRKit::SparseMatrixColMajor, QRKit::HouseholderQRFixed<7, 4>, 2, 8> fixedSlvr;
        QRKit::BandedBlockedSparseQR<QRKit::SparseMatrixColMajor, QRKit::HouseholderQR, QRKit::Dynamic, 8> genericSlvr;
        fixedSlvr.compute(spJ);
        genericSlvr.compute(spJ);
        const SparseMatrixColMajor &Rf = fixedSlvr.matrixR(), &Rg = genericSlvr.matrixR();
        bool same = Rf.values().size() == Rg.values().size() && Rf.innerIndex() == Rg.innerIndex() && Rf.outerIndex() == Rg.outerIndex();
        double md = 0.0;
        if (same) for (size_t i = 0; i < Rf.values().size(); ++i) md = std::max(md, std::fabs(Rf.values()[i] - Rg.values()[i]));
        const bool ok = same && md == 0.0 && fixedSlvr.rowsPermutation().indices() == genericSlvr.rowsPermutation().indices();
        std::printf("fixed-pattern banded analysis (QRKit:: spelling, 7x4 blocks, overlap 2): %s\n", ok ? "Passed." : "Failed.");
        fails += ok ? 0 : 1;
    }
    return fails ? 1 : 0;
}
