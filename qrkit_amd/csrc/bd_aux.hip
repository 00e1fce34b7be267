// bd_aux.hip -- pattern generation and the block-local solve path of the block-diagonal solver.
//
//  * bd_pattern_*: CSR pattern of m_Q and CSC pattern of m_R exactly as
//    BlockDiagonalSparseQR::factorize inserts them (BlockDiagonalSparseQR.h:455-500,530-541);
//    pure functions of the tile sizes, so they are generated on the device with coalesced stores
//    instead of one insertBack() per entry.
//  * bd_apply_qt / bd_solve: y = Q^T b and x = P R^-1 (Q^T b)_top of _solve_impl (:257-280);
//    both are block-local: one wavefront per tile and right-hand side, or (solve, tiles of at most 32 columns, round 5) a group of
//    2 .. 32 lanes per tile.
#include "qrk_device.h"

#include <cstdlib>

namespace qrk {

// QRK_SOLVE_GROUPED=0: one wavefront per tile in solve / applyQ / solveR (the kernels of rounds 1-4).  A diagnostic switch, read on every
// call like the other ones (QRK_BBS_MAPS, QRK_SOLVE_R_COOP): tests toggle it inside one process.
static bool solve_grouped()
{
    const char* e = std::getenv("QRK_SOLVE_GROUPED");
    return !(e && e[0] == '0');
}


__device__ __forceinline__ void tile_geom(const TileGeom& g, int64_t t, int& r, int& c, int64_t& qoff,
                                          int64_t& roff, int& base_row, int& base_col)
{
    if (g.t_rows) {
        r = g.t_rows[t]; c = g.t_cols[t]; qoff = g.q_off[t]; roff = g.r_off[t];
        base_row = g.row_off[t]; base_col = g.c_off[t];
    } else {
        r = g.rows; c = g.cols;
        qoff = t * (int64_t)r * r; roff = t * (int64_t)(c * (c + 1) / 2);
        base_row = (int)(t * r); base_col = (int)(t * c);
    }
}

// One workgroup per tile.  m1 (the running sum of rows-cols, BlockDiagonalSparseQR.h:428,471)
// equals base_row - base_col.
__global__ void __launch_bounds__(256)
bd_pattern_kernel(TileGeom g, int32_t* __restrict__ q_rowptr, int32_t* __restrict__ q_colidx,
                  int32_t* __restrict__ r_colptr, int32_t* __restrict__ r_rowidx)
{
    for (int64_t t = blockIdx.x; t < g.num_tiles; t += gridDim.x) {
        int r, c, base_row, base_col;
        int64_t qoff, roff;
        tile_geom(g, t, r, c, qoff, roff, base_row, base_col);
        const int m1 = base_row - base_col;
        const int n_start = g.mat_cols;
        for (int j = threadIdx.x; j < r; j += blockDim.x) q_rowptr[base_row + j] = (int32_t)(qoff + (int64_t)j * r);
        for (int e = threadIdx.x; e < r * r; e += blockDim.x) {
            const int j = e / r, k = e - j * r;
            (void)j;
            int32_t colidx;
            if (g.q_format == 0) colidx = k < c ? base_col + k : n_start + m1 + (k - c);
            else colidx = base_row + k;
            q_colidx[qoff + e] = colidx;
        }
        for (int k = threadIdx.x; k < c; k += blockDim.x) r_colptr[base_col + k] = (int32_t)(roff + (int64_t)k * (k + 1) / 2);
        const int rbase = g.q_format == 0 ? base_col : base_row;
        for (int k = 0; k < c; ++k)
            for (int j = threadIdx.x; j <= k; j += blockDim.x) r_rowidx[roff + (int64_t)k * (k + 1) / 2 + j] = rbase + j;
    }
}

// Trailing identity rows of Q and the closing pointers.
__global__ void __launch_bounds__(256)
bd_pattern_tail_kernel(TileGeom g, int64_t nnz_r, int32_t* __restrict__ q_rowptr,
                       int32_t* __restrict__ q_colidx, int32_t* __restrict__ r_colptr)
{
    const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    const int64_t ntail = (int64_t)g.mat_rows - g.sum_rows;
    if (i < ntail) {
        q_rowptr[g.sum_rows + i] = (int32_t)(g.nnz_q_tiles + i);
        q_colidx[g.nnz_q_tiles + i] = (int32_t)(g.sum_rows + i);
    }
    if (i == 0) {
        q_rowptr[g.mat_rows] = (int32_t)(g.nnz_q_tiles + ntail);
        r_colptr[g.mat_cols] = (int32_t)nnz_r;
    }
}

// Fill the trailing identity VALUES of Q (BlockDiagonalSparseQR.h:530-533).
__global__ void __launch_bounds__(256)
bd_q_tail_ones_kernel(double* __restrict__ q_vals, int64_t start, int64_t count)
{
    const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (i < count) q_vals[start + i] = 1.0;
}

// y = Q^T b for one tile and one right-hand side per wavefront (tiles with rows <= 64 use the
// lanes as output index; larger tiles loop).  FullQ: y[base_col+k] for k < c, y[N + m1 + k - c]
// for the N part; BlockDiagonalQ: y[base_row + k].
__global__ void __launch_bounds__(64)
bd_apply_qt_kernel(TileGeom g, const double* __restrict__ q_vals, const double* __restrict__ b,
                   int64_t nrhs, double* __restrict__ y)
{
    const int lane = threadIdx.x;
    const int64_t total = g.num_tiles * nrhs;
    for (int64_t w = blockIdx.x; w < total; w += gridDim.x) {
        const int64_t t = w % g.num_tiles, rhs = w / g.num_tiles;
        int r, c, base_row, base_col;
        int64_t qoff, roff;
        tile_geom(g, t, r, c, qoff, roff, base_row, base_col);
        const double* bb = b + rhs * (int64_t)g.mat_rows + base_row;
        double* yy = y + rhs * (int64_t)g.mat_rows;
        const int m1 = base_row - base_col;
        for (int k = lane; k < r; k += 64) {
            double s = 0.0;
            for (int j = 0; j < r; ++j) s = fma(q_vals[qoff + (int64_t)j * r + k], bb[j], s);
            int idx;
            if (g.q_format == 0) idx = k < c ? base_col + k : g.mat_cols + m1 + (k - c);
            else idx = base_row + k;
            yy[idx] = s;
        }
    }
}

// The same for batches of equal tiles with at most 32 rows: 64 / RP (tile, right-hand side) pairs per wavefront
// (RP = rows rounded up to a power of two), so that all the lanes work and a wavefront reads 64 consecutive entries of
// b - one wave per pair left 56 of 64 lanes idle on the 8x6 tiles of the block-angular BASELINE shape (13.4 ms for
// Q1^T J2 with 2000 columns, 5 GB of traffic).
template <int RP>
__global__ void __launch_bounds__(64)
bd_apply_qt_small_kernel(TileGeom g, const double* __restrict__ q_vals, const double* __restrict__ b,
                         int64_t nrhs, double* __restrict__ y)
{
    constexpr int PER = 64 / RP;
    const int lane = threadIdx.x, sub = lane / RP, k = lane % RP;
    const int r = g.rows, c = g.cols;
    const int64_t total = g.num_tiles * nrhs;
    for (int64_t w0 = (int64_t)blockIdx.x * PER; w0 < total; w0 += (int64_t)gridDim.x * PER) {
        const int64_t w = w0 + sub;
        if (w >= total || k >= r) continue;
        const int64_t t = w % g.num_tiles, rhs = w / g.num_tiles;
        const int base_row = (int)(t * r), base_col = (int)(t * c);
        const double* q = q_vals + t * (int64_t)r * r;
        const double* bb = b + rhs * (int64_t)g.mat_rows + base_row;
        double* yy = y + rhs * (int64_t)g.mat_rows;
        double s = 0.0;
        {
            double qv[RP], bv[RP];                             // (every row in flight before the first use; round 5)
#pragma unroll
            for (int j = 0; j < RP; ++j) { const int jj = j < r ? j : 0; qv[j] = q[jj * r + k]; bv[j] = bb[jj]; }
#pragma unroll
            for (int j = 0; j < RP; ++j) if (j < r) s = fma(qv[j], bv[j], s);
        }
        int idx;
        if (g.q_format == 0) idx = k < c ? base_col + k : g.mat_cols + (base_row - base_col) + (k - c);
        else idx = base_row + k;
        yy[idx] = s;
    }
}

// The same for MANY right-hand sides (Q1^T J2 of the block-angular composition: 2000 columns): a lane keeps its column of Q_i in
// registers and walks a chunk of right-hand sides - per output one coalesced load of b, the tile's b through LDS (one write, RP / 2
// 16-byte reads), RP FMAs, one store.  The kernel above loads Q_i and b 2 RP times per output (1.4 TB/s on the 8 x 6 tiles).
constexpr int APQT_CHUNK = 64;      // right-hand sides per wave
template <int RP>
__global__ void __launch_bounds__(64)
bd_apply_qt_small_many_kernel(TileGeom g, const double* __restrict__ q_vals, const double* __restrict__ b, int64_t nrhs,
                              double* __restrict__ y)
{
    constexpr int PER = 64 / RP;
    __shared__ __attribute__((aligned(16))) double sb[2][64];
    const int lane = threadIdx.x, sub = lane / RP, k = lane % RP;
    const int r = g.rows, c = g.cols;
    const int64_t tgroups = (g.num_tiles + PER - 1) / PER;
    const int64_t tg = blockIdx.x % tgroups, chunk = blockIdx.x / tgroups;
    const int64_t t = tg * PER + sub;
    const bool on = t < g.num_tiles && k < r;
    const int base_row = (int)(t * r), base_col = (int)(t * c);
    double qk[RP];
    {
        const double* q = q_vals + (on ? t : 0) * (int64_t)r * r;
#pragma unroll
        for (int j = 0; j < RP; ++j) { const double v = q[(j < r ? j : 0) * r + (k < r ? k : 0)]; qk[j] = (on && j < r) ? v : 0.0; }
    }
    int idx;
    if (g.q_format == 0) idx = k < c ? base_col + k : g.mat_cols + (base_row - base_col) + (k - c);
    else idx = base_row + k;
    const int64_t r0 = chunk * APQT_CHUNK, r1 = r0 + APQT_CHUNK < nrhs ? r0 + APQT_CHUNK : nrhs;
    const double* bp = b + r0 * (int64_t)g.mat_rows + (on ? base_row + k : 0);
    double bn = r0 < r1 ? *bp : 0.0;
    for (int64_t rhs = r0; rhs < r1; ++rhs) {
        const int par = (int)(rhs & 1);
        sb[par][lane] = on ? bn : 0.0;
        bp += g.mat_rows;
        if (rhs + 1 < r1) bn = *bp;                          // next right-hand side in flight
        __builtin_amdgcn_wave_barrier();
        double s0 = 0.0, s1 = 0.0;
#pragma unroll
        for (int j = 0; j < RP; j += 2) {
            const double2 v2 = *reinterpret_cast<const double2*>(&sb[par][sub * RP + j]);
            s0 = fma(qk[j], v2.x, s0); s1 = fma(qk[j + 1], v2.y, s1);
        }
        if (on) y[rhs * (int64_t)g.mat_rows + idx] = s0 + s1;
        __builtin_amdgcn_wave_barrier();
    }
}

// y = Q b, the product matrixQ() * b of the explicit m_Q (BlockDiagonalSparseQR.h:235-237, row j of Q_i = [U_i(j,:), N_i(j,:)],
// :455-492): one lane per row of a tile, the rows of Q_i are contiguous in q_vals.  FullQ reads b at base_col + k (k < c)
// and at N + m1 + (k - c) for the N part; BlockDiagonalQ at base_row + k.  The trailing identity rows copy b.
__global__ void __launch_bounds__(64)
bd_apply_q_kernel(TileGeom g, const double* __restrict__ q_vals, const double* __restrict__ b,
                  int64_t nrhs, double* __restrict__ y)
{
    const int lane = threadIdx.x;
    const int64_t total = g.num_tiles * nrhs;
    for (int64_t w = blockIdx.x; w < total; w += gridDim.x) {
        const int64_t t = w % g.num_tiles, rhs = w / g.num_tiles;
        int r, c, base_row, base_col;
        int64_t qoff, roff;
        tile_geom(g, t, r, c, qoff, roff, base_row, base_col);
        const double* bb = b + rhs * (int64_t)g.mat_rows;
        double* yy = y + rhs * (int64_t)g.mat_rows + base_row;
        const int m1 = base_row - base_col;
        for (int j = lane; j < r; j += 64) {
            const double* qrow = q_vals + qoff + (int64_t)j * r;
            double s = 0.0;
            for (int k = 0; k < r; ++k) {
                int idx;
                if (g.q_format == 0) idx = k < c ? base_col + k : g.mat_cols + m1 + (k - c);
                else idx = base_row + k;
                s = fma(qrow[k], bb[idx], s);
            }
            yy[j] = s;
        }
    }
}

__global__ void __launch_bounds__(256)
bd_copy_tail_kernel(TileGeom g, const double* __restrict__ b, int64_t nrhs, double* __restrict__ y)
{
    const int64_t ntail = (int64_t)g.mat_rows - g.sum_rows;
    const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (i < ntail * nrhs) {
        const int64_t rhs = i / ntail, k = i - rhs * ntail;
        y[rhs * (int64_t)g.mat_rows + g.sum_rows + k] = b[rhs * (int64_t)g.mat_rows + g.sum_rows + k];
    }
}

// _solve_impl for FullQ, tiles with cols <= 64: per wavefront, y = (Q_i^T b_i)(0:c), column-oriented
// back substitution with the packed upper-triangular R_i, x[perm[base_col+k]] = y[k].
__global__ void __launch_bounds__(64)
bd_solve_kernel(TileGeom g, const double* __restrict__ q_vals, const double* __restrict__ r_vals,
                const int32_t* __restrict__ perm, const double* __restrict__ b, int64_t nrhs,
                double* __restrict__ x)
{
    const int lane = threadIdx.x;
    const int64_t total = g.num_tiles * nrhs;
    for (int64_t w = blockIdx.x; w < total; w += gridDim.x) {
        const int64_t t = w % g.num_tiles, rhs = w / g.num_tiles;
        int r, c, base_row, base_col;
        int64_t qoff, roff;
        tile_geom(g, t, r, c, qoff, roff, base_row, base_col);
        const double* bb = b + rhs * (int64_t)g.mat_rows + base_row;
        double yk = 0.0;
        {
            // (eight rows of Q in flight per lane, clamped addresses: one load per trip was a chain of memory latencies; round 5)
            const double* qk = q_vals + qoff + (lane < c ? lane : 0);
            for (int j0 = 0; j0 < r; j0 += 8) {
                double qv[8], bv[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) { const int j = (j0 + u < r) ? j0 + u : 0; qv[u] = qk[(int64_t)j * r]; bv[u] = bb[j]; }
#pragma unroll
                for (int u = 0; u < 8; ++u) if (lane < c && j0 + u < r) yk = fma(qv[u], bv[u], yk);
            }
        }
        for (int kk = c - 1; kk >= 0; --kk) {
            const double* colk = r_vals + roff + (int64_t)kk * (kk + 1) / 2;
            const double piv = readlane_f64(yk, kk) / colk[kk];
            if (lane == kk) yk = piv;
            else if (lane < kk) yk = fma(-colk[lane], piv, yk);
        }
        if (lane < c) x[rhs * (int64_t)g.mat_cols + perm[base_col + lane]] = yk;
    }
}

// The same with G lanes per tile (G = 2 .. 32, a power of two >= the widest tile of the launch), 64 / G tiles per wavefront (round 5):
// one wavefront per tile left 62 of 64 lanes idle on the reference's 7 x 2 blocks (0.11 of the HBM roofline at 10^6 tiles; 32 x 32: 0.27)
// and read Q one row per trip -- a chain of memory latencies.  Here lane k of a group owns entry k of y: eight rows of Q in flight per
// lane (clamped addresses, not predicated loads), the lane's row of R (entries (k, kk), kk >= k) prefetched into registers before the
// substitution, whose steps are unrolled (static register indices) and broadcast the solved entry by ds_bpermute.  The sums run in the
// same order as in bd_solve_kernel: the results are bitwise the same.
// WITH_Q = false: the back substitution alone (bd_solve_r_kernel's job: z = R^-1 y, both indexed by column, no permutation).
template <int G, bool WITH_Q>
__global__ void __launch_bounds__(64)
bd_solve_group_kernel(TileGeom g, const double* __restrict__ q_vals, const double* __restrict__ r_vals,
                      const int32_t* __restrict__ perm, const double* __restrict__ b, int64_t nrhs,
                      double* __restrict__ x)
{
    constexpr int TPW = 64 / G;
    const int lane = threadIdx.x, grp = lane / G, k = lane % G;
    const int64_t ntg = (g.num_tiles + TPW - 1) / TPW;
    const int64_t total = ntg * nrhs;
    for (int64_t w = blockIdx.x; w < total; w += gridDim.x) {
        const int64_t t = (w % ntg) * TPW + grp, rhs = w / ntg;
        const bool valid = t < g.num_tiles;
        int r, c, base_row, base_col;
        int64_t qoff, roff;
        tile_geom(g, valid ? t : 0, r, c, qoff, roff, base_row, base_col);
        if (!valid) { r = 0; c = 0; }
        const bool act = k < c;
        // the lane's row of R: entry (k, kk) of the packed upper triangle is at kk (kk + 1) / 2 + k
        double rrow[G];
#pragma unroll
        for (int kk = 0; kk < G; ++kk) {
            const double v = r_vals[roff + ((act && kk >= k && kk < c) ? (int64_t)kk * (kk + 1) / 2 + k : 0)];
            rrow[kk] = (act && kk >= k && kk < c) ? v : 1.0;
        }
        double yk = 0.0;
        if (WITH_Q) {
            // y_k = sum_j Q(j, k) b_j, j ascending (one accumulator: the order of bd_solve_kernel)
            const double* bb = b + rhs * (int64_t)g.mat_rows + base_row;
            const double* qk = q_vals + qoff + (act ? k : 0);
            int rmax = r;                                      // (ragged batches: the tallest tile of the wave bounds the loop)
            if (g.t_rows) {
#pragma unroll
                for (int o = 32; o >= 1; o >>= 1) { const int other = __shfl_xor(rmax, o, 64); rmax = other > rmax ? other : rmax; }
            }
            for (int j0 = 0; j0 < rmax; j0 += 8) {
                double qv[8], bv[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) { const int j = (j0 + u < r) ? j0 + u : 0; qv[u] = qk[(int64_t)j * r]; bv[u] = bb[j]; }
#pragma unroll
                for (int u = 0; u < 8; ++u) if (act && j0 + u < r) yk = fma(qv[u], bv[u], yk);
            }
        } else if (act) {
            yk = b[rhs * (int64_t)g.mat_cols + base_col + k];
        }
        // column-oriented back substitution (the triangularView<Upper>().solve of _solve_impl, :271)
#pragma unroll
        for (int kk = G - 1; kk >= 0; --kk) {
            const double pv = yk / rrow[kk];                   // (meaningful in lane kk of the group)
            const double piv = __shfl(pv, grp * G + kk, 64);
            if (kk < c) {
                if (k == kk) yk = piv;
                else if (k < kk) yk = fma(-rrow[kk], piv, yk);
            }
        }
        if (act) x[rhs * (int64_t)g.mat_cols + (WITH_Q ? perm[base_col + k] : base_col + k)] = yk;
    }
}

// y = Q b with RP lanes per tile (RP = 2 .. 64 >= the tallest tile), 64 / RP tiles per wavefront (round 5; bd_apply_q_kernel: one
// wavefront per tile and one entry of Q per loop trip): lane j of a group owns row j of Q_i, all of it in flight before the first use
// (clamped addresses); lane k also fetches the k-th entry of the tile's part of b, which reaches the others by ds_bpermute.  Same
// products in the same order.
template <int RP>
__global__ void __launch_bounds__(64)
bd_apply_q_group_kernel(TileGeom g, const double* __restrict__ q_vals, const double* __restrict__ b, int64_t nrhs,
                        double* __restrict__ y)
{
    constexpr int TPW = 64 / RP;
    const int lane = threadIdx.x, grp = lane / RP, j = lane % RP;
    const int64_t ntg = (g.num_tiles + TPW - 1) / TPW;
    const int64_t total = ntg * nrhs;
    for (int64_t w = blockIdx.x; w < total; w += gridDim.x) {
        const int64_t t = (w % ntg) * TPW + grp, rhs = w / ntg;
        const bool valid = t < g.num_tiles;
        int r, c, base_row, base_col;
        int64_t qoff, roff;
        tile_geom(g, valid ? t : 0, r, c, qoff, roff, base_row, base_col);
        if (!valid) r = 0;
        const bool act = j < r;
        double qv[RP];
        if (!g.t_rows) {
            // uniform tiles: the wave's 64 / RP tiles are one contiguous run of q_vals -- coalesced loads (eight per lane in flight) into
            // LDS, row stride RP + 1, then every lane its row (a lane reading its row straight from memory touches 64 lines per
            // instruction: 16 x 16 ran at 0.25 of the roofline that way)
            __shared__ double lq[64 * (RP + 1)];
            const int ru = g.rows, rr = ru * ru;               // (r is 0 in the lanes of a group beyond the batch: they still help to load)
            const int64_t first = (w % ntg) * TPW;
            const int ntw = (int)(g.num_tiles - first < TPW ? g.num_tiles - first : TPW);
            const int n = ntw * rr;
            const double* src = q_vals + first * (int64_t)rr;
            const float inv_rr = 1.0f / (float)rr, inv_r = 1.0f / (float)ru;
            __builtin_amdgcn_wave_barrier();                   // (the rows of the round before have been read)
            for (int e0 = lane; e0 < n; e0 += 64 * 8) {
                double ld[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) { const int e = e0 + 64 * u; ld[u] = src[e < n ? e : n - 1]; }
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int e = e0 + 64 * u;
                    if (e < n) {
                        const int tl = (int)(((float)e + 0.5f) * inv_rr), rem = e - tl * rr;
                        const int row = (int)(((float)rem + 0.5f) * inv_r), col = rem - row * ru;
                        lq[(tl * RP + row) * (RP + 1) + col] = ld[u];
                    }
                }
            }
            __builtin_amdgcn_wave_barrier();
            const double* lrow = lq + (grp * RP + (act ? j : 0)) * (RP + 1);
#pragma unroll
            for (int k = 0; k < RP; ++k) { const double v = lrow[k]; qv[k] = (act && k < r) ? v : 0.0; }
        } else {
            const double* qrow = q_vals + qoff + (int64_t)(act ? j : 0) * r;
#pragma unroll
            for (int k = 0; k < RP; ++k) { const double v = qrow[k < r ? k : 0]; qv[k] = (act && k < r) ? v : 0.0; }
        }
        // entry j of the tile's part of b: FullQ reads b at base_col + k (k < c) and at N + m1 + (k - c); BlockDiagonalQ at base_row + k
        int idx;
        if (g.q_format == 0) idx = j < c ? base_col + j : g.mat_cols + (base_row - base_col) + (j - c);
        else idx = base_row + j;
        const double bj = act ? b[rhs * (int64_t)g.mat_rows + idx] : 0.0;
        double s = 0.0;
#pragma unroll
        for (int k = 0; k < RP; ++k) {
            const double bk = __shfl(bj, grp * RP + k, 64);
            if (k < r) s = fma(qv[k], bk, s);
        }
        if (act) y[rhs * (int64_t)g.mat_rows + base_row + j] = s;
    }
}

// Block upper-triangular solve only: z = R(0:cols,0:cols)^-1 y with y, z: mat_cols x nrhs (the
// triangularView<Upper>().solve step of _solve_impl, BlockDiagonalSparseQR.h:271, on its own; the
// angular composition needs it with a modified right-hand side).  One workgroup per tile and RHS.
__global__ void __launch_bounds__(256)
bd_solve_r_kernel(TileGeom g, const double* __restrict__ r_vals, const double* __restrict__ y, int64_t nrhs,
                  double* __restrict__ z)
{
    extern __shared__ double ysm[];
    const int tid = threadIdx.x;
    const int64_t total = g.num_tiles * nrhs;
    for (int64_t w = blockIdx.x; w < total; w += gridDim.x) {
        const int64_t t = w % g.num_tiles, rhs = w / g.num_tiles;
        int r, c, base_row, base_col;
        int64_t qoff, roff;
        tile_geom(g, t, r, c, qoff, roff, base_row, base_col);
        for (int k = tid; k < c; k += blockDim.x) ysm[k] = y[rhs * (int64_t)g.mat_cols + base_col + k];
        __syncthreads();
        for (int kk = c - 1; kk >= 0; --kk) {
            const double* colk = r_vals + roff + (int64_t)kk * (kk + 1) / 2;
            const double piv = ysm[kk] / colk[kk];
            __syncthreads();
            if (tid == 0) ysm[kk] = piv;
            for (int j = tid; j < kk; j += blockDim.x) ysm[j] = fma(-colk[j], piv, ysm[j]);
            __syncthreads();
        }
        for (int k = tid; k < c; k += blockDim.x) z[rhs * (int64_t)g.mat_cols + base_col + k] = ysm[k];
        __syncthreads();
    }
}

// _solve_impl for tiles of any size: one workgroup per tile and right-hand side, y in LDS.
__global__ void __launch_bounds__(256)
bd_solve_wg_kernel(TileGeom g, const double* __restrict__ q_vals, const double* __restrict__ r_vals,
                   const int32_t* __restrict__ perm, const double* __restrict__ b, int64_t nrhs,
                   double* __restrict__ x)
{
    extern __shared__ double ysm[];
    const int tid = threadIdx.x;
    const int64_t total = g.num_tiles * nrhs;
    for (int64_t w = blockIdx.x; w < total; w += gridDim.x) {
        const int64_t t = w % g.num_tiles, rhs = w / g.num_tiles;
        int r, c, base_row, base_col;
        int64_t qoff, roff;
        tile_geom(g, t, r, c, qoff, roff, base_row, base_col);
        const double* bb = b + rhs * (int64_t)g.mat_rows + base_row;
        for (int k = tid; k < c; k += blockDim.x) {
            double s = 0.0;
            const double* qk = q_vals + qoff + k;
            for (int j0 = 0; j0 < r; j0 += 8) {
                double qv[8], bv[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) { const int j = (j0 + u < r) ? j0 + u : 0; qv[u] = qk[(int64_t)j * r]; bv[u] = bb[j]; }
#pragma unroll
                for (int u = 0; u < 8; ++u) if (j0 + u < r) s = fma(qv[u], bv[u], s);
            }
            ysm[k] = s;
        }
        __syncthreads();
        for (int kk = c - 1; kk >= 0; --kk) {
            const double* colk = r_vals + roff + (int64_t)kk * (kk + 1) / 2;
            const double piv = ysm[kk] / colk[kk];
            __syncthreads();
            if (tid == 0) ysm[kk] = piv;
            for (int j = tid; j < kk; j += blockDim.x) ysm[j] = fma(-colk[j], piv, ysm[j]);
            __syncthreads();
        }
        for (int k = tid; k < c; k += blockDim.x) x[rhs * (int64_t)g.mat_cols + perm[base_col + k]] = ysm[k];
        __syncthreads();
    }
}

// Tiles cut out of a sparse matrix on the device: SparseBlockDiagonal::fromBlockDiagonalPattern
// (SparseBlockDiagonal.h:71-89), i.e. BlockMatrixType(mat.block(idxRow, idxCol, numRows, numCols)) for the blocks
// (base_row, base_col, rows, cols) of BlockBandedMatrixInfo::fromBlockDiagonalPattern (SparseQRUtils.h:255-272).
// One workgroup per tile.  The tile is assembled piece by piece in LDS (a piece = PR rows x PC columns, at most
// CUT_BUF doubles) and leaves as coalesced column-major stores, so HBM sees each tile element written once.
// The entries of a piece's outer range (columns for CSC, rows for CSR) are one contiguous run of the value
// array: the threads walk it with stride 256 and find the outer index of an entry by bisection of the
// outer pointers kept in LDS.  Entries outside the block (the reference's block() ignores them) are skipped.
constexpr int CUT_BUF = 4096;      // doubles of LDS per piece
constexpr int CUT_OUTER = 1024;    // outer indices per piece at most

template <bool CSR>
__global__ void __launch_bounds__(256)
bd_cut_tiles_kernel(TileGeom g, const int64_t* __restrict__ t_off, const int32_t* __restrict__ outer_ptr,
                    const int32_t* __restrict__ inner_idx, const double* __restrict__ vals, int32_t nnz,
                    double* __restrict__ tiles)
{
    __shared__ double buf[CUT_BUF];
    __shared__ int32_t optr[CUT_OUTER + 1];
    const int tid = threadIdx.x;
    for (int64_t t = blockIdx.x; t < g.num_tiles; t += gridDim.x) {
        int r, c, base_row, base_col;
        int64_t qoff, roff;
        tile_geom(g, t, r, c, qoff, roff, base_row, base_col);
        const int64_t toff = t_off ? t_off[t] : t * (int64_t)r * c;
        if (r <= 0 || c <= 0) continue;
        // piece shape: as many whole columns as fit (CSC walks columns, CSR walks rows of the piece)
        const int pr_max = r < CUT_BUF ? r : CUT_BUF;
        int pc_max = CUT_BUF / pr_max;
        if (pc_max > c) pc_max = c;
        const int o_max = CSR ? (pr_max < CUT_OUTER ? pr_max : CUT_OUTER) : (pc_max < CUT_OUTER ? pc_max : CUT_OUTER);
        const int prr = CSR ? o_max : pr_max;   // rows per piece
        const int pcc = CSR ? pc_max : o_max;   // columns per piece
        for (int c0 = 0; c0 < c; c0 += pcc) {
            const int nc = c - c0 < pcc ? c - c0 : pcc;
            for (int r0 = 0; r0 < r; r0 += prr) {
                const int nr = r - r0 < prr ? r - r0 : prr;
                const int n_outer = CSR ? nr : nc;
                const int o_base = CSR ? base_row + r0 : base_col + c0;
                for (int e = tid; e < nr * nc; e += 256) buf[e] = 0.0;
                for (int o = tid; o <= n_outer; o += 256) optr[o] = outer_ptr[o_base + o];
                __syncthreads();
                // (clamped to the arrays: malformed outer pointers must not turn into out-of-bounds reads)
                const int e_begin = optr[0] > 0 ? optr[0] : 0, e_end = optr[n_outer] < nnz ? optr[n_outer] : nnz;
                for (int e = e_begin + tid; e < e_end; e += 256) {
                    int lo = 0, hi = n_outer;          // outer index o with optr[o] <= e < optr[o+1]
                    while (hi - lo > 1) {
                        const int mid = (lo + hi) >> 1;
                        if (optr[mid] <= e) lo = mid; else hi = mid;
                    }
                    const int in = inner_idx[e] - (CSR ? base_col + c0 : base_row + r0);
                    if (in >= 0 && in < (CSR ? nc : nr)) {
                        const int row = CSR ? lo : in, col = CSR ? in : lo;
                        buf[col * nr + row] = vals[e];
                    }
                }
                __syncthreads();
                for (int e = tid; e < nr * nc; e += 256) {
                    const int col = e / nr, row = e - col * nr;
                    tiles[toff + (int64_t)(c0 + col) * r + r0 + row] = buf[e];
                }
                __syncthreads();
            }
        }
    }
}

void launch_bd_pattern(const TileGeom& g, int64_t nnz_r, int32_t* q_rowptr, int32_t* q_colidx,
                       int32_t* r_colptr, int32_t* r_rowidx, hipStream_t stream)
{
    if (g.num_tiles > 0) {
        const unsigned grid = (unsigned)(g.num_tiles < 65536 ? g.num_tiles : 65536);
        hipLaunchKernelGGL(bd_pattern_kernel, dim3(grid), dim3(256), 0, stream, g, q_rowptr, q_colidx,
                           r_colptr, r_rowidx);
    }
    const int64_t ntail = (int64_t)g.mat_rows - g.sum_rows;
    const unsigned tgrid = (unsigned)((ntail > 0 ? ntail : 1) + 255) / 256;
    hipLaunchKernelGGL(bd_pattern_tail_kernel, dim3(tgrid), dim3(256), 0, stream, g, nnz_r, q_rowptr,
                       q_colidx, r_colptr);
}

void launch_bd_q_tail_ones(double* q_vals, int64_t start, int64_t count, hipStream_t stream)
{
    if (count <= 0) return;
    hipLaunchKernelGGL(bd_q_tail_ones_kernel, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, stream,
                       q_vals, start, count);
}

void launch_bd_apply_qt(const TileGeom& g, const double* q_vals, const double* b, int64_t nrhs, double* y,
                        hipStream_t stream)
{
    const int64_t total = g.num_tiles * nrhs;
    if (total > 0) {
        if (g.t_rows == nullptr && g.rows <= 32) {
            const int rp = g.rows <= 4 ? 4 : (g.rows <= 8 ? 8 : (g.rows <= 16 ? 16 : 32));
            if (nrhs >= 16) {                                  // many right-hand sides: Q_i in registers, a chunk of columns per wave
                const int64_t tgroups = (g.num_tiles + 64 / rp - 1) / (64 / rp), chunks = (nrhs + APQT_CHUNK - 1) / APQT_CHUNK;
                if (tgroups * chunks <= 0x7fffffff) {
                    const unsigned grid = (unsigned)(tgroups * chunks);
                    if (rp == 4) hipLaunchKernelGGL(bd_apply_qt_small_many_kernel<4>, dim3(grid), dim3(64), 0, stream, g, q_vals, b, nrhs, y);
                    else if (rp == 8) hipLaunchKernelGGL(bd_apply_qt_small_many_kernel<8>, dim3(grid), dim3(64), 0, stream, g, q_vals, b, nrhs, y);
                    else if (rp == 16) hipLaunchKernelGGL(bd_apply_qt_small_many_kernel<16>, dim3(grid), dim3(64), 0, stream, g, q_vals, b, nrhs, y);
                    else hipLaunchKernelGGL(bd_apply_qt_small_many_kernel<32>, dim3(grid), dim3(64), 0, stream, g, q_vals, b, nrhs, y);
                    goto tail;
                }
            }
            const int64_t waves = (total + 64 / rp - 1) / (64 / rp);
            const unsigned grid = (unsigned)(waves < 262144 ? waves : 262144);
            if (rp == 4) hipLaunchKernelGGL(bd_apply_qt_small_kernel<4>, dim3(grid), dim3(64), 0, stream, g, q_vals, b, nrhs, y);
            else if (rp == 8) hipLaunchKernelGGL(bd_apply_qt_small_kernel<8>, dim3(grid), dim3(64), 0, stream, g, q_vals, b, nrhs, y);
            else if (rp == 16) hipLaunchKernelGGL(bd_apply_qt_small_kernel<16>, dim3(grid), dim3(64), 0, stream, g, q_vals, b, nrhs, y);
            else hipLaunchKernelGGL(bd_apply_qt_small_kernel<32>, dim3(grid), dim3(64), 0, stream, g, q_vals, b, nrhs, y);
        } else {
            const unsigned grid = (unsigned)(total < 262144 ? total : 262144);
            hipLaunchKernelGGL(bd_apply_qt_kernel, dim3(grid), dim3(64), 0, stream, g, q_vals, b, nrhs, y);
        }
    }
tail:
    const int64_t ntail = ((int64_t)g.mat_rows - g.sum_rows) * nrhs;
    if (ntail > 0)
        hipLaunchKernelGGL(bd_copy_tail_kernel, dim3((unsigned)((ntail + 255) / 256)), dim3(256), 0, stream, g,
                           b, nrhs, y);
}

void launch_bd_apply_q(const TileGeom& g, int max_rows, const double* q_vals, const double* b, int64_t nrhs, double* y, hipStream_t stream)
{
    const int64_t total = g.num_tiles * nrhs;
    const bool grouped = solve_grouped();
    if (total > 0 && grouped && max_rows <= 64) {
        int RP = 2;
        while (RP < max_rows) RP *= 2;
        const int64_t waves = ((g.num_tiles + 64 / RP - 1) / (64 / RP)) * nrhs;
        const unsigned gg = (unsigned)(waves < 262144 ? waves : 262144);
#define QRK_APQ_G(GG) case GG: hipLaunchKernelGGL((bd_apply_q_group_kernel<GG>), dim3(gg), dim3(64), 0, stream, g, q_vals, b, nrhs, y); break;
        switch (RP) { QRK_APQ_G(2) QRK_APQ_G(4) QRK_APQ_G(8) QRK_APQ_G(16) QRK_APQ_G(32) QRK_APQ_G(64) default: break; }
#undef QRK_APQ_G
    } else if (total > 0) {
        const unsigned grid = (unsigned)(total < 262144 ? total : 262144);
        hipLaunchKernelGGL(bd_apply_q_kernel, dim3(grid), dim3(64), 0, stream, g, q_vals, b, nrhs, y);
    }
    const int64_t ntail = ((int64_t)g.mat_rows - g.sum_rows) * nrhs;
    if (ntail > 0)
        hipLaunchKernelGGL(bd_copy_tail_kernel, dim3((unsigned)((ntail + 255) / 256)), dim3(256), 0, stream, g, b, nrhs, y);
}

void launch_bd_solve_r(const TileGeom& g, int max_cols, const double* r_vals, const double* y, int64_t nrhs, double* z,
                       hipStream_t stream)
{
    const int64_t total = g.num_tiles * nrhs;
    if (total <= 0) return;
    const bool grouped = solve_grouped();
    if (grouped && max_cols <= 32) {
        int G = 2;
        while (G < max_cols) G *= 2;
        const int64_t waves = ((g.num_tiles + 64 / G - 1) / (64 / G)) * nrhs;
        const unsigned gg = (unsigned)(waves < 262144 ? waves : 262144);
#define QRK_SOLVER_G(GG) case GG: hipLaunchKernelGGL((bd_solve_group_kernel<GG, false>), dim3(gg), dim3(64), 0, stream, g, nullptr, r_vals, nullptr, y, nrhs, z); break;
        switch (G) { QRK_SOLVER_G(2) QRK_SOLVER_G(4) QRK_SOLVER_G(8) QRK_SOLVER_G(16) QRK_SOLVER_G(32) default: break; }
#undef QRK_SOLVER_G
        return;
    }
    const unsigned grid = (unsigned)(total < 262144 ? total : 262144);
    const unsigned threads = max_cols <= 64 ? 64 : 256;
    hipLaunchKernelGGL(bd_solve_r_kernel, dim3(grid), dim3(threads), (size_t)max_cols * sizeof(double), stream, g, r_vals, y,
                       nrhs, z);
}

void launch_bd_solve(const TileGeom& g, int max_cols, const double* q_vals, const double* r_vals, const int32_t* perm,
                     const double* b, int64_t nrhs, double* x, hipStream_t stream)
{
    const int64_t total = g.num_tiles * nrhs;
    if (total <= 0) return;
    const unsigned grid = (unsigned)(total < 262144 ? total : 262144);
    const bool grouped = solve_grouped();   // (0: one wavefront per tile, rounds 1-4)
    // (up to 32 columns; at 33 .. 64 a group is the whole wavefront and the 64 row registers + 64 unrolled steps were slower than the
    //  loop of bd_solve_kernel: 64 x 64, 20 000 tiles: 393 against 286 us)
    if (max_cols <= 32 && grouped) {
        int G = 2;
        while (G < max_cols) G *= 2;
        const int64_t waves = ((g.num_tiles + 64 / G - 1) / (64 / G)) * nrhs;
        const unsigned gg = (unsigned)(waves < 262144 ? waves : 262144);
#define QRK_SOLVE_G(GG) case GG: hipLaunchKernelGGL((bd_solve_group_kernel<GG, true>), dim3(gg), dim3(64), 0, stream, g, q_vals, r_vals, perm, b, nrhs, x); break;
        switch (G) { QRK_SOLVE_G(2) QRK_SOLVE_G(4) QRK_SOLVE_G(8) QRK_SOLVE_G(16) QRK_SOLVE_G(32) default: break; }
#undef QRK_SOLVE_G
    } else if (max_cols <= 64)
        hipLaunchKernelGGL(bd_solve_kernel, dim3(grid), dim3(64), 0, stream, g, q_vals, r_vals, perm, b, nrhs, x);
    else
        hipLaunchKernelGGL(bd_solve_wg_kernel, dim3(grid), dim3(256), (size_t)max_cols * sizeof(double), stream, g, q_vals,
                           r_vals, perm, b, nrhs, x);
}

// The same for uniform SMALL blocks (at most 256 entries; round 5): a workgroup per tile spends a launch slot and 256 threads on the 14
// entries of the reference's 7 x 2 blocks (200 000 of them: 334 us, 2 % of the HBM roofline).  Here the tiles are zeroed by a memset and a
// THREAD per outer index (a column of the CSC matrix, a row of the CSR one) scatters its entries into the tile they belong to; entries
// outside the block are skipped as block() skips them.
template <bool CSR>
__global__ void __launch_bounds__(256)
bd_cut_tiles_small_kernel(TileGeom g, const int32_t* __restrict__ outer_ptr, const int32_t* __restrict__ inner_idx,
                          const double* __restrict__ vals, double* __restrict__ tiles)
{
    const int r = g.rows, c = g.cols;
    const int64_t n_outer = g.num_tiles * (int64_t)(CSR ? r : c);
    for (int64_t o = (int64_t)blockIdx.x * 256 + threadIdx.x; o < n_outer; o += (int64_t)gridDim.x * 256) {
        const int64_t t = o / (CSR ? r : c);
        const int lo = (int)(o - t * (CSR ? r : c));          // local row (CSR) / local column (CSC)
        const int64_t base_in = t * (int64_t)(CSR ? c : r);   // first inner index of the block
        double* dst = tiles + t * (int64_t)r * c + (CSR ? lo : lo * r);
        const int e0 = outer_ptr[o], e1 = outer_ptr[o + 1];
        for (int e = e0; e < e1; ++e) {
            const int64_t li = (int64_t)inner_idx[e] - base_in;
            if (li >= 0 && li < (CSR ? c : r)) dst[CSR ? li * r : li] = vals[e];
        }
    }
}

void launch_bd_cut_tiles(const TileGeom& g, const int64_t* t_off, int row_major, const int32_t* outer_ptr,
                         const int32_t* inner_idx, const double* vals, int32_t nnz, double* tiles, hipStream_t stream)
{
    if (g.num_tiles <= 0) return;
    // (a few hundred tiles are one short launch either way: 256 tiles of 7 x 2 take 7-8 us with the workgroup per tile, 10 with memset + scatter)
    if (!g.t_rows && g.rows * g.cols <= 256 && g.num_tiles > 1024) {
        (void)hipMemsetAsync(tiles, 0, (size_t)g.num_tiles * g.rows * g.cols * sizeof(double), stream);
        const int64_t n_outer = g.num_tiles * (int64_t)(row_major ? g.rows : g.cols);
        const int64_t wg = (n_outer + 255) / 256;
        const unsigned gs = (unsigned)(wg < 65536 ? wg : 65536);
        if (row_major) hipLaunchKernelGGL(bd_cut_tiles_small_kernel<true>, dim3(gs), dim3(256), 0, stream, g, outer_ptr, inner_idx, vals, tiles);
        else hipLaunchKernelGGL(bd_cut_tiles_small_kernel<false>, dim3(gs), dim3(256), 0, stream, g, outer_ptr, inner_idx, vals, tiles);
        return;
    }
    const unsigned grid = (unsigned)(g.num_tiles < 262144 ? g.num_tiles : 262144);
    if (row_major)
        hipLaunchKernelGGL(bd_cut_tiles_kernel<true>, dim3(grid), dim3(256), 0, stream, g, t_off, outer_ptr, inner_idx, vals, nnz, tiles);
    else
        hipLaunchKernelGGL(bd_cut_tiles_kernel<false>, dim3(grid), dim3(256), 0, stream, g, t_off, outer_ptr, inner_idx, vals, nnz, tiles);
}

// y -= S(:, colidx) z: a thread per row (column-major S: a wave reads 512 contiguous bytes per column), the z of 256 columns at a
// time through LDS.  HBM-bound: the strip is read once.
__global__ void __launch_bounds__(256)
gemv_sub_kernel(const double* __restrict__ S, int64_t lds, int64_t rows, int64_t cols, const int32_t* __restrict__ colidx,
                const double* __restrict__ z, double* __restrict__ y)
{
    __shared__ double zs[256];
    __shared__ int cs[256];
    const int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x;
    double acc = 0.0;
    for (int64_t c0 = 0; c0 < cols; c0 += 256) {
        const int64_t c = c0 + threadIdx.x;
        __syncthreads();
        if (c < cols) { zs[threadIdx.x] = z[c]; cs[threadIdx.x] = colidx ? colidx[c] : (int)c; }
        __syncthreads();
        const int nc = (int)(cols - c0 < 256 ? cols - c0 : 256);
        if (r < rows) {
#pragma unroll 8
            for (int q = 0; q < nc; ++q) acc = fma(S[(int64_t)cs[q] * lds + r], zs[q], acc);
        }
    }
    if (r < rows) y[r] -= acc;
}

hipError_t launch_gemv_sub(const double* S, int64_t lds, int64_t rows, int64_t cols, const int32_t* colidx, const double* z, double* y,
                           hipStream_t stream)
{
    hipLaunchKernelGGL(gemv_sub_kernel, dim3((unsigned)((rows + 255) / 256)), dim3(256), 0, stream, S, lds, rows, cols, colidx, z, y);
    return hipGetLastError();
}

// Dense column-major copy of a row window of a compressed sparse matrix (the right block J2 of a block-angular matrix handed over
// sparse: BlockedThinSparseQR.h:131 "m_R = mat" densifies it inside the right solver either way).  The window is zeroed by the
// launcher; here one wave walks one outer vector (a row of a CSR matrix, a column of a CSC one), its entries over the lanes.
template <bool CSR>
__global__ void __launch_bounds__(256)
sparse_window_kernel(int64_t outer_size, const int32_t* __restrict__ outer, const int32_t* __restrict__ inner,
                     const double* __restrict__ vals, int64_t row0, int64_t nrows, const int32_t* __restrict__ row_map,
                     double* __restrict__ out, int64_t ld)
{
    const int lane = threadIdx.x & 63;
    const int64_t wave = ((int64_t)blockIdx.x * 256 + threadIdx.x) >> 6, nwaves = ((int64_t)gridDim.x * 256) >> 6;
    const int64_t o_lo = CSR ? row0 : 0, o_hi = CSR ? row0 + nrows : outer_size;
    for (int64_t o = o_lo + wave; o < o_hi; o += nwaves) {
        const int e0 = outer[o], e1 = outer[o + 1];
        for (int e = e0 + lane; e < e1; e += 64) {
            const int64_t r = CSR ? o : inner[e], c = CSR ? inner[e] : o;
            if (r < row0 || r >= row0 + nrows) continue;
            const int64_t dr = row_map ? row_map[r - row0] : r - row0;
            out[c * ld + dr] = vals[e];
        }
    }
}

hipError_t launch_sparse_window_to_dense(bool row_major, int64_t rows, int64_t cols, const int32_t* outer, const int32_t* inner,
                                         const double* vals, int64_t row0, int64_t nrows, const int32_t* row_map, double* out,
                                         int64_t ld, hipStream_t stream)
{
    if (nrows <= 0 || cols <= 0) return hipSuccess;
    if (hipError_t e = hipMemset2DAsync(out, (size_t)ld * sizeof(double), 0, (size_t)nrows * sizeof(double), (size_t)cols, stream)) return e;
    const int64_t work = row_major ? nrows : cols;          // outer vectors to walk, one wave each
    int64_t grid = (work + 3) / 4;
    if (grid > 65536) grid = 65536;
    if (row_major)
        hipLaunchKernelGGL(sparse_window_kernel<true>, dim3((unsigned)grid), dim3(256), 0, stream, rows, outer, inner, vals, row0, nrows,
                           row_map, out, ld);
    else
        hipLaunchKernelGGL(sparse_window_kernel<false>, dim3((unsigned)grid), dim3(256), 0, stream, cols, outer, inner, vals, row0, nrows,
                           row_map, out, ld);
    return hipGetLastError();
}

}  // namespace qrk
