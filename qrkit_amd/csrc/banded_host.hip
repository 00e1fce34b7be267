// banded_host.hip -- host-side structure analysis of the block-banded solver (see banded_host.h).
// Pure integer logic; must reproduce the reference's block maps bit for bit
// (known answers: test/test-utils.cpp:199-205,228-241,264-271).
#include "banded_host.h"

#include <algorithm>
#include <map>
#include <numeric>

namespace qrk {

namespace {

struct RowRange {
    int32_t origIdx, start, end;
};

// Panels from the row blocks of a banded matrix: what BlockBandedMatrixInfo::mergeBlocks (SparseQRUtils.h:308-385) computes, as one
// pass of a panel builder over the blocks in order.  A block whose columns end inside the last finished panel only makes that panel
// taller; otherwise it extends the panel under construction, which is finished as soon as it is portrait (strictly more rows than
// columns), at least `min_step` columns wide and at least `suggested`; what is still under construction at the end is finished if it
// qualifies and otherwise folded into the last panel.  The reference's known answers (test/test-utils.cpp:228-241, 264-271) pin it.
struct PanelBuilder {
    std::vector<BlockInfo> done;
    bool open = false;
    int32_t row0 = 0, col0 = 0, rows = 0, cols = 0;
    int32_t min_step, suggested;
    PanelBuilder(int32_t step, int32_t sugg) : min_step(step), suggested(sugg) {}
    bool qualifies() const { return rows > cols && cols >= min_step && cols >= suggested; }
    void finish() {
        BlockInfo b;
        b.idxRow = row0; b.idxCol = col0; b.numRows = rows; b.numCols = cols;
        done.push_back(b);
        open = false;
    }
    void add(const BlockInfo& b) {
        if (!done.empty() && b.idxCol + b.numCols <= done.back().idxCol + done.back().numCols) { done.back().numRows += b.numRows; return; }
        if (!open) { open = true; row0 = b.idxRow; col0 = b.idxCol; }
        rows = b.idxRow + b.numRows - row0;
        cols = b.idxCol + b.numCols - col0;
        if (qualifies()) finish();
    }
    bool close(std::string& err) {
        if (!open) return true;
        if (qualifies()) { finish(); return true; }
        if (done.empty()) {
            err = "block structure cannot be merged into portrait panels (the reference reads back() of an empty vector here, "
                  "SparseQRUtils.h:375)";
            return false;
        }
        BlockInfo& last = done.back();
        last.numRows += rows;
        last.numCols = col0 + cols - last.idxCol;
        open = false;
        return true;
    }
};

bool merge_blocks(const std::vector<BlockInfo>& in, int maxColStep, int suggested, std::vector<BlockInfo>& out, std::string& err)
{
    PanelBuilder pb(maxColStep, suggested);
    for (const BlockInfo& b : in) pb.add(b);
    if (!pb.close(err)) return false;
    out.swap(pb.done);
    return true;
}

}  // namespace

// BlockBandedMatrixInfo::fromBlockBandedPattern (SparseQRUtils.h:274-302), before the merge: numBlocks = cols / (blockCols -
// overlap) blocks (i blockRows, i step, blockRows, blockCols), the last one blockCols - overlap wide
static bool fixed_block_map(int32_t cols, const FixedBandedPattern& fx, std::vector<BlockInfo>& raw, int32_t& maxColStep, std::string& err)
{
    maxColStep = fx.block_cols - fx.overlap;
    if (fx.block_rows <= 0 || fx.block_cols <= 0 || fx.overlap < 0 || maxColStep <= 0) { err = "bad fixed banded pattern"; return false; }
    const int32_t numBlocks = cols / maxColStep;
    for (int32_t i = 0; i < numBlocks; ++i) {
        BlockInfo b;
        b.idxRow = i * fx.block_rows; b.idxCol = i * maxColStep; b.numRows = fx.block_rows;
        b.numCols = i < numBlocks - 1 ? fx.block_cols : fx.block_cols - fx.overlap;
        raw.push_back(b);
    }
    return true;
}

// The merged block map of the fixed pattern alone (fromBlockBandedPattern + mergeBlocks): what the reference's known answers
// (test/test-utils.cpp:228-241) pin.
bool banded_block_map_fixed(int32_t rows, int32_t cols, const FixedBandedPattern& fx, int32_t suggested, std::vector<BlockInfo>& blocks,
                            std::string& err)
{
    (void)rows;
    std::vector<BlockInfo> raw;
    int32_t maxColStep = 0;
    if (!fixed_block_map(cols, fx, raw, maxColStep, err)) return false;
    blocks.clear();
    return merge_blocks(raw, maxColStep, suggested, blocks, err);
}

bool analyze_banded(int32_t rows, int32_t cols, const int32_t* rowptr, const int32_t* colidx, int32_t suggested,
                    BandedStructure& out, std::string& err, const FixedBandedPattern* fixed)
{
    out = BandedStructure();
    out.rows = rows; out.cols = cols;
    if (rows <= 0 || cols <= 0) { err = "empty matrix"; return false; }
    // The pattern comes straight from the public C ABI: a malformed CSR (row pointers out of order, column indices out of
    // range or unsorted within a row) would otherwise yield blocks that overrun the matrix and out-of-bounds panel descriptors.
    if (!rowptr || rowptr[0] < 0) { err = "invalid CSR pattern: rowptr[0] < 0"; return false; }
    for (int32_t j = 0; j < rows; ++j) {
        if (rowptr[j + 1] < rowptr[j]) { err = "invalid CSR pattern: row pointers decrease"; return false; }
        if (rowptr[j + 1] > rowptr[j] && !colidx) { err = "invalid CSR pattern: colidx is NULL"; return false; }
        for (int32_t e = rowptr[j]; e < rowptr[j + 1]; ++e) {
            if (colidx[e] < 0 || colidx[e] >= cols) { err = "invalid CSR pattern: column index out of range"; return false; }
            if (e > rowptr[j] && colidx[e] <= colidx[e - 1]) { err = "invalid CSR pattern: column indices of a row not strictly increasing"; return false; }
        }
    }

    // ---- AsBandedAsPossible (SparseQROrdering.h:66-119): stable sort of the rows by first nonzero column
    std::vector<RowRange> ranges((size_t)rows);
    for (int32_t j = 0; j < rows; ++j) {
        int32_t s = cols, e;
        if (rowptr[j + 1] > rowptr[j]) s = colidx[rowptr[j]];
        e = s;
        if (rowptr[j + 1] > rowptr[j]) e = colidx[rowptr[j + 1] - 1];
        ranges[(size_t)j] = RowRange{j, s, e};
    }
    auto less = [](const RowRange& a, const RowRange& b) { return a.start < b.start; };
    // (fixed pattern, BandedBlockedSparseQR.h:398-408: "rows are already sorted, no permutation needed")
    out.has_row_perm = !fixed && !std::is_sorted(ranges.begin(), ranges.end(), less);
    if (out.has_row_perm) std::stable_sort(ranges.begin(), ranges.end(), less);
    out.row_perm.assign((size_t)rows, 0);
    for (int32_t r = 0; r < rows; ++r) out.row_perm[(size_t)ranges[(size_t)r].origIdx] = r;

    // ---- permuted matrix (m_pmat = m_rowPerm * mat, BandedBlockedSparseQR.h:446) as a CSR view
    out.prowptr.assign((size_t)rows + 1, 0);
    for (int32_t r = 0; r < rows; ++r) {
        const int32_t o = ranges[(size_t)r].origIdx;
        out.prowptr[(size_t)r + 1] = out.prowptr[(size_t)r] + (rowptr[o + 1] - rowptr[o]);
    }
    const int64_t nnz = out.prowptr[(size_t)rows];
    out.pcol.resize((size_t)nnz);
    out.pmap.resize((size_t)nnz);
    for (int32_t r = 0; r < rows; ++r) {
        const int32_t o = ranges[(size_t)r].origIdx;
        int64_t dst = out.prowptr[(size_t)r];
        for (int32_t e = rowptr[o]; e < rowptr[o + 1]; ++e, ++dst) { out.pcol[(size_t)dst] = colidx[e]; out.pmap[(size_t)dst] = e; }
    }

    std::vector<BlockInfo> raw;                 // the row blocks before the merge, in row order
    int32_t maxColStep = 0;
    if (fixed) {
        if (!fixed_block_map(cols, *fixed, raw, maxColStep, err)) return false;
    } else {
    // ---- BlockBandedMatrixInfo::operator() (SparseQRUtils.h:186-253) on the sorted rows
    std::map<int32_t, int32_t> bandWidths, bandHeights;
    for (const RowRange& rr : ranges) {
        const int32_t bw = rr.end - rr.start + 1;
        auto it = bandWidths.find(rr.start);
        if (it == bandWidths.end()) bandWidths.insert(std::make_pair(rr.start, bw));
        else if (it->second < bw) it->second = bw;
        bandHeights[rr.start] += 1;
    }
    for (size_t j = 0; j + 1 < ranges.size(); ++j)
        maxColStep = std::max(maxColStep, ranges[j + 1].start - ranges[j].start);
    // one block per distinct first column, at the first row that starts there (the rows are sorted by it: a new start is a new block)
    int32_t rowIdx = 0;
    for (const RowRange& rr : ranges) {
        if (rr.start < cols && (raw.empty() || raw.back().idxCol != rr.start)) {
            BlockInfo b;
            b.idxRow = rowIdx; b.idxCol = rr.start; b.numRows = bandHeights.at(rr.start); b.numCols = bandWidths.at(rr.start);
            raw.push_back(b);
        }
        ++rowIdx;
    }
    }
    if (!merge_blocks(raw, maxColStep, suggested, out.blocks, err)) return false;
    if (out.blocks.empty()) { err = "no blocks found"; return false; }
    for (const BlockInfo& b : out.blocks)
        if (b.idxRow < 0 || b.idxCol < 0 || b.numRows <= 0 || b.numCols <= 0 || b.idxCol + b.numCols > cols || b.idxRow + b.numRows > rows) {
            err = "block map overruns the matrix";
            return false;
        }

    // ---- the panel chain of factorize() (BandedBlockedSparseQR.h:457-508) as descriptors
    const size_t nb = out.blocks.size();
    BlockInfo bi = out.blocks[0];
    int32_t activeRows = bi.numRows, numZeros = 0, row0 = bi.idxRow, ncolsJi = bi.numCols;
    int32_t lo_rows = 0, lo_cols = 0, lo_from = 0;
    for (size_t i = 0; i < nb; ++i) {
        bi = out.blocks[i];
        if (ncolsJi != bi.numCols) {
            err = "panel wider than its block (overlap larger than the next block): not supported";
            return false;
        }
        if (activeRows < bi.numCols) { err = "landscape panel in the banded chain"; return false; }
        BBPanel p;
        p.row0 = row0; p.col0 = bi.idxCol; p.act_rows = activeRows; p.ncols = bi.numCols;
        p.solved = (i == nb - 1) ? bi.numRows : out.blocks[i + 1].idxCol - bi.idxCol;
        if (p.solved > activeRows || p.col0 + p.solved > rows) { err = "panel emits more rows than it has"; return false; }
        p.lo_rows = lo_rows; p.lo_cols = lo_cols; p.lo_from = lo_from;
        p.yrow = bi.idxCol; p.num_zeros = numZeros;
        p.y_off = out.y_len; p.t_off = out.t_len; p.r_off = out.stage_len;
        out.y_len += (int64_t)activeRows * bi.numCols;
        out.t_len += (int64_t)bi.numCols * bi.numCols;
        out.stage_len += (int64_t)p.solved * bi.numCols;
        out.max_act_rows = std::max(out.max_act_rows, activeRows);
        out.max_ncols = std::max(out.max_ncols, bi.numCols);
        if (p.yrow + p.ncols + p.num_zeros + (p.act_rows - p.ncols) > rows) { err = "Q block exceeds the matrix rows"; return false; }
        out.panels.push_back(p);
        if (i + 1 < nb) {
            const BlockInfo nx = out.blocks[i + 1];
            const int32_t overlap = (bi.idxCol + bi.numCols) - nx.idxCol;
            const int32_t colInc = bi.numCols - overlap;
            activeRows = bi.numRows + nx.numRows - colInc;
            numZeros = std::max((nx.idxRow + nx.numRows) - activeRows - nx.idxCol, 0);
            ncolsJi = nx.numCols >= overlap ? nx.numCols : overlap;
            row0 = bi.idxRow + colInc;
            lo_rows = overlap > 0 ? activeRows - nx.numRows : 0;
            lo_cols = overlap > 0 ? overlap : 0;
            lo_from = colInc;
            if (row0 + activeRows > rows || lo_rows < 0 || lo_from + lo_rows > out.panels.back().act_rows) {
                err = "inconsistent panel chain";
                return false;
            }
        }
    }

    // ---- CSC pattern of R: panel i contributes the dense rows [col0, col0+solved) x [col0, col0+ncols)
    std::vector<int32_t> cnt((size_t)cols + 1, 0);
    for (const BBPanel& p : out.panels)
        for (int32_t c = 0; c < p.ncols; ++c) cnt[(size_t)(p.col0 + c) + 1] += p.solved;
    out.r_colptr.assign((size_t)cols + 1, 0);
    for (int32_t c = 0; c < cols; ++c) out.r_colptr[(size_t)c + 1] = out.r_colptr[(size_t)c] + cnt[(size_t)c + 1];
    out.nnz_r = out.r_colptr[(size_t)cols];
    out.r_rowidx.resize((size_t)out.nnz_r);
    out.r_src.resize((size_t)out.nnz_r);
    std::vector<int32_t> fill(out.r_colptr.begin(), out.r_colptr.end() - 1);
    for (const BBPanel& p : out.panels)     // panels in order => rows ascending inside every column
        for (int32_t c = 0; c < p.ncols; ++c)
            for (int32_t br = 0; br < p.solved; ++br) {
                const int32_t pos = fill[(size_t)(p.col0 + c)]++;
                out.r_rowidx[(size_t)pos] = p.col0 + br;
                out.r_src[(size_t)pos] = p.r_off + (int64_t)c * p.solved + br;
            }
    return true;
}

}  // namespace qrk
