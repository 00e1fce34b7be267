// Sanitizer driver for the host-side structure analysis of the banded solver (qrkit_amd/csrc/banded_host.hip: pure C++ / STL integer
// logic, the part of the product that never runs on the GPU).  Built with g++ -fsanitize=address,undefined by `make san` and run on
// the reference's known answers (test/test-utils.cpp:182-241: 256 blocks (7i, 2i, 7, 2); 255 blocks (7i, 2i, 7, 4) with the last
// 14 x 4), on shuffled rows, on the BASELINE configs[2] strip shape and on malformed patterns (which must be rejected, not read).
#include "../../qrkit_amd/csrc/banded_host.h"

#include <algorithm>
#include <cstdio>
#include <numeric>
#include <random>
#include <vector>

using namespace qrk;

static int fails = 0;
#define CHECK(cond, msg) do { if (!(cond)) { std::printf("FAILED: %s\n", msg); ++fails; } } while (0)

int main()
{
    std::vector<int32_t> rp, ci;
    int rows, cols;
    std::string err;
    {   // generate_block_diagonal_matrix: the rows of variable pair i are 7 i .. 7 i + 6, columns 2 i, 2 i + 1 (numVars = 128 pairs here)
        const int nb = 256;
        rows = 7 * nb; cols = 2 * nb;
        for (unsigned seed : {0u, 3u}) {
            std::vector<std::vector<int32_t> > r((size_t)rows);
            for (int i = 0; i < nb; ++i) for (int q = 0; q < 7; ++q) { r[(size_t)(7 * i + q)].push_back(2 * i); r[(size_t)(7 * i + q)].push_back(2 * i + 1); }
            std::vector<int> order((size_t)rows); std::iota(order.begin(), order.end(), 0);
            if (seed) { std::mt19937 g(seed); std::shuffle(order.begin(), order.end(), g); }
            rp.assign(1, 0); ci.clear();
            for (int k = 0; k < rows; ++k) { for (int32_t c : r[(size_t)order[(size_t)k]]) ci.push_back(c); rp.push_back((int32_t)ci.size()); }
            BandedStructure st;
            CHECK(analyze_banded(rows, cols, rp.data(), ci.data(), 2, st, err), "block-diagonal pattern rejected");
            CHECK((int)st.blocks.size() == nb, "block-diagonal pattern: 256 blocks expected");
            for (int i = 0; i < (int)st.blocks.size() && i < nb; ++i)
                CHECK(st.blocks[(size_t)i].idxRow == 7 * i && st.blocks[(size_t)i].idxCol == 2 * i && st.blocks[(size_t)i].numRows == 7 && st.blocks[(size_t)i].numCols == 2, "block (7i, 2i, 7, 2)");
            CHECK(st.has_row_perm == (seed != 0), "row permutation flag");
        }
    }
    {   // overlapping pattern: 255 blocks (7i, 2i, 7, 4), the last 14 x 4 (test-utils.cpp:228-241), rows shuffled
        const int nb = 256;
        rows = 7 * nb; cols = 2 * nb;
        std::vector<std::vector<int32_t> > r((size_t)rows);
        for (int i = 0; i < nb; ++i)
            for (int q = 0; q < 7; ++q) {
                r[(size_t)(7 * i + q)].push_back(2 * i); r[(size_t)(7 * i + q)].push_back(2 * i + 1);
                if (i + 1 < nb) { r[(size_t)(7 * i + q)].push_back(2 * i + 2); r[(size_t)(7 * i + q)].push_back(2 * i + 3); }
            }
        std::vector<int> order((size_t)rows); std::iota(order.begin(), order.end(), 0);
        std::mt19937 g(11); std::shuffle(order.begin(), order.end(), g);
        rp.assign(1, 0); ci.clear();
        for (int k = 0; k < rows; ++k) { for (int32_t c : r[(size_t)order[(size_t)k]]) ci.push_back(c); rp.push_back((int32_t)ci.size()); }
        BandedStructure st;
        CHECK(analyze_banded(rows, cols, rp.data(), ci.data(), 2, st, err), "overlapping pattern rejected");
        CHECK((int)st.blocks.size() == nb - 1, "overlapping pattern: 255 blocks expected");
        if ((int)st.blocks.size() == nb - 1) {
            for (int i = 0; i < nb - 2; ++i)
                CHECK(st.blocks[(size_t)i].idxRow == 7 * i && st.blocks[(size_t)i].idxCol == 2 * i && st.blocks[(size_t)i].numRows == 7 && st.blocks[(size_t)i].numCols == 4, "block (7i, 2i, 7, 4)");
            CHECK(st.blocks.back().numRows == 14 && st.blocks.back().numCols == 4, "last block 14 x 4");
        }
        // panels: offsets increase, every panel is portrait
        int64_t yo = -1;
        for (const BBPanel& p : st.panels) { CHECK(p.y_off > yo, "panel offsets increase"); yo = p.y_off; CHECK(p.act_rows >= p.ncols, "portrait panel"); }
    }
    {   // BASELINE configs[2] strip shape: 64 strips of 256 x 192, step 64 (generic analysis; the fixed block map on its own)
        const int N = 64, ms = 256, n = 192, s = 64;
        rows = N * ms; cols = (N - 1) * s + n;
        rp.assign(1, 0); ci.clear();
        for (int i = 0; i < N; ++i) for (int q = 0; q < ms; ++q) { for (int c = 0; c < n; ++c) ci.push_back(i * s + c); rp.push_back((int32_t)ci.size()); }
        FixedBandedPattern fx{ms, n, n - s};
        BandedStructure st2;
        CHECK(analyze_banded(rows, cols, rp.data(), ci.data(), s, st2, err), "configs[2] strips rejected (generic pattern)");
        CHECK(!st2.has_row_perm, "strips need no row permutation");
        std::vector<BlockInfo> blocks;
        CHECK(banded_block_map_fixed(rows, cols, fx, s, blocks, err), "fixed block map");
    }
    {   // malformed CSR patterns must be rejected before anything is indexed with them
        BandedStructure st;
        int32_t bad_rp1[] = {0, 2, 1}, cidx[] = {0, 1};
        CHECK(!analyze_banded(2, 2, bad_rp1, cidx, 2, st, err), "decreasing row pointers accepted");
        int32_t rp2[] = {0, 2}, bad_ci[] = {1, 5};
        CHECK(!analyze_banded(1, 2, rp2, bad_ci, 2, st, err), "column index out of range accepted");
        int32_t bad_ci2[] = {1, 0};
        CHECK(!analyze_banded(1, 2, rp2, bad_ci2, 2, st, err), "unsorted column indices accepted");
        CHECK(!analyze_banded(0, 0, rp2, cidx, 2, st, err), "empty matrix accepted");
    }
    std::printf("banded_host under the sanitizers: %s\n", fails ? "Failed." : "Passed.");
    return fails ? 1 : 0;
}
