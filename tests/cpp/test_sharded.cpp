// The multi-GPU entry of the C ABI and of the facade on ONE GPU: qrk_shard_ranges (host logic, any world size) and
// ShardedBlockDiagonalSparseQR / qrk_gather_r with world = 1 over a real one-rank RCCL communicator (ncclCommInitRank; a
// one-rank communicator is legal) -- the path bench.py's N > 1 legs and a C++ user on an 8-GPU node take, minus the peers.
// The gathered R and permutation must equal the un-sharded solver's, bit for bit (same kernels, same tiles).
#include <cstdio>
#include <cmath>
#include <cstring>
#include <random>

#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h>

#include "qrkit/QRKit.hpp"

using namespace qrkit;

static int check_ranges() {
    // mixed sizes, three ranks: contiguous cover, offsets = prefix sums, sentinel = totals, cost balanced
    std::mt19937 g(5);
    std::uniform_int_distribution<int> d(8, 256);
    const int B = 3000, W = 3;
    std::vector<int32_t> n((size_t)B);
    for (auto& v : n) v = d(g);
    std::vector<qrk_shard> sh((size_t)W + 1);
    if (qrk_shard_ranges(B, 0, 0, n.data(), n.data(), W, sh.data()) != QRK_STATUS_OK) return 1;
    if (sh[0].first_block != 0 || sh[W].first_block != B || sh[W].num_blocks != 0) return 1;
    double cost[3] = {0, 0, 0}, total = 0;
    int64_t br = 0, bc = 0, to = 0, qo = 0, ro = 0;
    for (int g2 = 0; g2 < W; ++g2) {
        if (sh[g2].first_block + sh[g2].num_blocks != sh[g2 + 1].first_block) return 1;
        if (sh[g2].base_row != br || sh[g2].base_col != bc || sh[g2].tiles_off != to || sh[g2].q_off != qo || sh[g2].r_off != ro) return 1;
        for (int64_t i = sh[g2].first_block; i < sh[g2 + 1].first_block; ++i) {
            const int64_t v = n[(size_t)i];
            br += v; bc += v; to += v * v; qo += v * v; ro += v * (v + 1) / 2;
            cost[g2] += (double)v * v * v;
        }
        total += cost[g2];
    }
    if (sh[W].base_row != br || sh[W].r_off != ro) return 1;
    for (int g2 = 0; g2 < W; ++g2) if (cost[g2] > 1.05 * total / W) return 1;
    // uniform layout, 8 ranks: equal counts
    std::vector<qrk_shard> su(9);
    if (qrk_shard_ranges(10000, 32, 32, 0, 0, 8, su.data()) != QRK_STATUS_OK) return 1;
    for (int g2 = 0; g2 < 8; ++g2) if (su[g2].num_blocks != 1250 || su[g2].r_off != (int64_t)g2 * 1250 * 528) return 1;
    return 0;
}

int main() {
    int fails = 0;
    if (check_ranges()) { std::printf("qrk_shard_ranges: Failed.\n"); ++fails; } else std::printf("qrk_shard_ranges: Passed.\n");

    if (hipSetDevice(0) != hipSuccess) { std::printf("no GPU\n"); return 2; }
    ncclUniqueId id;
    ncclComm_t comm = 0;
    if (ncclGetUniqueId(&id) != ncclSuccess || ncclCommInitRank(&comm, 1, id, 0) != ncclSuccess) { std::printf("RCCL communicator: Failed.\n"); return 1; }

    std::mt19937 g(11);
    std::uniform_int_distribution<int> d(4, 70);
    std::uniform_real_distribution<double> u(-1.0, 1.0);
    SparseBlockDiagonal mat;
    Index rows = 0, cols = 0;
    for (int i = 0; i < 200; ++i) {
        const int c = d(g), r = c + d(g) % 5;
        Matrix m(r, c);
        for (Index e = 0; e < (Index)r * c; ++e) m.data()[e] = u(g);
        mat.insertBack(m);
        rows += r; cols += c;
    }
    mat.setDims((int)rows, (int)cols);

    BlockDiagonalSparseQR<ColPivHouseholderQR> whole;
    whole.compute(mat);
    ShardedBlockDiagonalSparseQR<ColPivHouseholderQR> shard(0, 1, comm);
    shard.compute(mat);
    Vector rv; std::vector<int> perm;
    shard.gatherR(0, rv, perm);
    const SparseMatrixColMajor& R = whole.matrixR();
    bool ok = rv.size() == R.values().size() && std::memcmp(rv.data(), R.values().data(), rv.size() * sizeof(double)) == 0;
    ok = ok && perm.size() == (size_t)cols;
    for (size_t j = 0; ok && j < perm.size(); ++j) ok = perm[j] == whole.colsPermutation().indices()[j];
    std::printf("ShardedBlockDiagonalSparseQR world 1 over RCCL, %zu R values, %zu columns: %s\n", rv.size(), perm.size(), ok ? "Passed." : "Failed.");
    if (!ok) ++fails;
    {
        // the sharded solve: block-local solve + gather of x only (qrk_gather_x); world 1: bit for bit the un-sharded solver's x
        Vector b((size_t)rows);
        for (double& v : b) v = u(g);
        const Vector xs = shard.solve(b, 0), xw = whole.solve(b);
        const bool oks = xs.size() == xw.size() && xs.size() == (size_t)cols && std::memcmp(xs.data(), xw.data(), xs.size() * sizeof(double)) == 0;
        std::printf("ShardedBlockDiagonalSparseQR::solve (x only gathered), %zu columns: %s\n", xs.size(), oks ? "Passed." : "Failed.");
        if (!oks) ++fails;
    }

    // ShardedBlockAngularSparseQR with world = 1 over the same communicator: 400 tiles of 8 x 6 on the left, 96 dense columns on the
    // right, rows below the left block as well; the least-squares solution must be the un-sharded BlockAngularSparseQR's and the
    // generating x, the permutation of the right block must be a permutation
    {
        const int nt = 400, br = 8, bc = 6, m2 = 96, extra = 50;
        SparseBlockDiagonal left;
        std::vector<Triplet> lt;
        for (int i = 0; i < nt; ++i) {
            Matrix m(br, bc);
            for (Index e = 0; e < (Index)br * bc; ++e) m.data()[e] = u(g);
            left.insertBack(m);
            for (int c = 0; c < bc; ++c) for (int r = 0; r < br; ++r) lt.push_back(Triplet(i * br + r, i * bc + c, m(r, c)));
        }
        const Index n1 = (Index)nt * br, m1 = (Index)nt * bc, nrows = n1 + extra;
        left.setDims((int)n1, (int)m1);
        Matrix right(nrows, m2);
        for (Index e = 0; e < nrows * m2; ++e) right.data()[e] = u(g);
        Vector x((size_t)(m1 + m2));
        for (double& v : x) v = u(g);
        Vector b((size_t)nrows, 0.0);
        for (const Triplet& t : lt) b[(size_t)t.row] += t.value * x[(size_t)t.col];
        for (Index j = 0; j < m2; ++j) for (Index i = 0; i < nrows; ++i) b[(size_t)i] += right(i, j) * x[(size_t)(m1 + j)];
        ShardedBlockAngularSparseQR<> sharded(0, 1, comm);
        sharded.compute(left, right);
        Vector x1, x2;
        sharded.solve(b, x1, x2);
        double num = 0.0, den = 0.0;
        for (Index i = 0; i < m1; ++i) { const double dlt = x1[(size_t)i] - x[(size_t)i]; num += dlt * dlt; den += x[(size_t)i] * x[(size_t)i]; }
        for (Index j = 0; j < m2; ++j) { const double dlt = x2[(size_t)j] - x[(size_t)(m1 + j)]; num += dlt * dlt; den += x[(size_t)(m1 + j)] * x[(size_t)(m1 + j)]; }
        bool ok2 = std::sqrt(num) <= 1e-9 * std::sqrt(den);
        std::vector<int> seen((size_t)m2, 0);
        for (int v : sharded.colsPermutationRight()) { if (v < 0 || v >= m2 || seen[(size_t)v]++) ok2 = false; }
        // the un-sharded solver on the same matrix: same pivots of the right block, same solution
        BlockAngularSparseQR<BlockDiagonalSparseQR<>, ColPivHouseholderQR> whole2;
        BlockMatrix1x2<SparseBlockDiagonal, Matrix> blk(left, right);
        whole2.compute(blk);
        const Vector xs = whole2.solve(whole2.rowsPermutation() * b);
        double num2 = 0.0;
        for (Index i = 0; i < m1 + m2; ++i) { const double dlt = xs[(size_t)i] - (i < m1 ? x1[(size_t)i] : x2[(size_t)(i - m1)]); num2 += dlt * dlt; }
        ok2 = ok2 && std::sqrt(num2) <= 1e-9 * std::sqrt(den);
        for (Index j = 0; ok2 && j < m2; ++j) ok2 = whole2.colsPermutation().indices()[(size_t)(m1 + j)] == (int)m1 + sharded.colsPermutationRight()[(size_t)j];
        std::printf("ShardedBlockAngularSparseQR world 1 over RCCL, %d tiles + %d dense columns: %s\n", nt, m2, ok2 ? "Passed." : "Failed.");
        if (!ok2) ++fails;
    }
    ncclCommDestroy(comm);
    return fails;
}
