// banded_host.h -- host-side structure analysis of the block-banded solver (product code, C++/STL).
//
// Counterpart of QRKit::BandedBlockedSparseQR::analyzePattern, generic path
// (src/QRKit/BandedBlockedSparseQR.h:409-427): SparseQROrdering::AsBandedAsPossible
// (src/QRKit/SparseQROrdering.h:66-119), BlockBandedMatrixInfo::operator() and mergeBlocks
// (src/QRKit/SparseQRUtils.h:186-253,308-385), plus the index arithmetic of the panel chain in
// factorize() (BandedBlockedSparseQR.h:457-508) turned into per-panel descriptors for the device.
#ifndef QRK_BANDED_HOST_H
#define QRK_BANDED_HOST_H

#include <cstdint>
#include <string>
#include <vector>

namespace qrk {

struct BlockInfo {
    int32_t idxRow = 0, idxCol = 0, numRows = 0, numCols = 0;
};

// One panel of the chain, as the device kernels need it.
struct BBPanel {
    int32_t row0;       // first row of the permuted matrix that enters the panel
    int32_t col0;       // idxCol of the block
    int32_t act_rows;   // activeRows
    int32_t ncols;      // numCols
    int32_t solved;     // rows of R emitted by this panel
    int32_t lo_rows;    // leftover block taken from the previous panel: rows ...
    int32_t lo_cols;    // ... and columns (0 for the first panel)
    int32_t lo_from;    // V.block(lo_from, lo_from, lo_rows, lo_cols) of the previous panel
    int32_t lo_stride = 1;   // leftover row i goes to panel row lo_stride * i (2: interleaved with the new rows, the strips form)
    int32_t yrow;       // BlockYTY row index (= idxCol)
    int32_t num_zeros;  // BlockYTY zero gap between its two row segments
    int64_t y_off;      // offset of the panel (act_rows x ncols, row-major; Y = its unit-lower view) in y_vals
    int64_t t_off;      // offset of T (ncols x ncols, column-major, negated) in t_vals
    int64_t r_off;      // offset of the emitted R rows (solved x ncols, column-major) in the staging array
};

struct BandedStructure {
    int32_t rows = 0, cols = 0;
    bool has_row_perm = false;
    std::vector<int32_t> row_perm;       // (P*M).row(row_perm[i]) = M.row(i)
    std::vector<BlockInfo> blocks;       // merged block map, in order
    std::vector<BBPanel> panels;
    // permuted matrix in CSR: entry e of the permuted matrix is entry pmap[e] of the caller's CSR
    std::vector<int32_t> prowptr, pcol;
    std::vector<int64_t> pmap;
    // R in CSC (explicit zeros of the emitted rows kept, BandedBlockedSparseQR.h:487-491)
    std::vector<int32_t> r_colptr, r_rowidx;
    std::vector<int64_t> r_src;          // CSC entry -> offset in the staging array
    int64_t nnz_r = 0, y_len = 0, t_len = 0, stage_len = 0;
    int32_t max_act_rows = 0, max_ncols = 0;
};

// Returns false and sets err when the reference itself would be outside its domain (e.g. mergeBlocks
// calling back() on an empty vector, SparseQRUtils.h:375) or a panel is not portrait.
// `fixed` (optional): the fixed-pattern path of BandedBlockedSparseQR::analyzePattern (BandedBlockedSparseQR.h:398-408): no row
// ordering, block map from BlockBandedMatrixInfo::fromBlockBandedPattern (SparseQRUtils.h:274-302) instead of band detection.
struct FixedBandedPattern { int32_t block_rows, block_cols, overlap; };
bool banded_block_map_fixed(int32_t rows, int32_t cols, const FixedBandedPattern& fx, int32_t suggested_block_cols,
                            std::vector<BlockInfo>& blocks, std::string& err);
bool analyze_banded(int32_t rows, int32_t cols, const int32_t* rowptr, const int32_t* colidx,
                    int32_t suggested_block_cols, BandedStructure& out, std::string& err,
                    const FixedBandedPattern* fixed = nullptr);

}  // namespace qrk
#endif
