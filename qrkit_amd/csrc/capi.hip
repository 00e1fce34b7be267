// capi.hip -- implementation of the C ABI declared in include/qrkit_amd.h.
//
// Host-side counterpart of QRKit::BlockDiagonalSparseQR (src/QRKit/BlockDiagonalSparseQR.h):
// qrk_bd_plan_create = analyzePattern (:392-405) plus the running offsets of the hot loop
// (:428-431,524-525) as prefix sums; qrk_bd_factorize = factorize (:415-547);
// qrk_bd_solve = _solve_impl (:257-280).  There is no CPU fallback anywhere in this file.
#include "../../include/qrkit_amd.h"
#include "qrk_device.h"
// The few RCCL / NCCL types the exchanges need, declared here so that the library builds on hosts without RCCL headers (the entry
// points are resolved at run time, rccl_api() below; the values are the ABI of nccl.h / rccl.h: ncclSuccess = 0, ncclInt8 = 0,
// ncclInt32 = 2, ncclFloat64 = 8)
typedef struct ncclComm* ncclComm_t;
typedef int ncclResult_t;
typedef int ncclDataType_t;
enum : int { ncclSuccess = 0, ncclInt8 = 0, ncclInt32 = 2, ncclFloat64 = 8 };
#include <dlfcn.h>
#include "banded_host.h"

#include <cstdio>
#include <cstdlib>
#include <algorithm>
#include <cmath>
#include <cstring>
#include <new>
#include <string>
#include <vector>

#define QRK_WG_MAX_DIM 2048
#define QRK_COL_MAX_DIM 256     // bdqr_col.hip: one thread per column of [A | Q^T], at most 512 columns

namespace {

thread_local std::string g_create_error;

}  // namespace

struct qrk_context_s {
    int device = 0;
    hipStream_t stream = nullptr;
    int num_cus = 256;
    int pair_wgs_per_cu = 8;       // resident pair-kernel workgroups per CU: 2 waves per SIMD (232 VGPRs, 20 KB LDS each)
    bool force_exact = false;      // QRK_EXACT=1: every tile through the exact-arithmetic path (bdqr_exact.hip): results bit-identical
                                   // to a scalar evaluation of Eigen's algorithm, ~30x slower; the fast kernels send only the tiles
                                   // whose decisions are not clear of rounding there
    bool use_thin_kernel = true;   // ... and of those the tiles with 1 or 2 columns: one tile per lane (bdqr_thin.hip); QRK_THIN=0 disables
    bool use_quad_kernel = true;   // uniform tiles with 5 .. 16 rows: four or eight tiles per wavefront (bdqr_quad.hip); QRK_QUAD=0: bdqr_small.hip's groups of 8 / 16 lanes
    int quad_min_rows = 5;         // (QRK_QUAD_MIN_ROWS: diagnostic -- shorter tiles stay on bdqr_small.hip)
    bool use_small_kernel = true;  // uniform tiles with at most 16 rows: 64/G tiles per wavefront (bdqr_small.hip); QRK_SMALL=0 disables
    // side streams for the size classes of a mixed batch (fork after / join into `stream`), created on first use
    hipStream_t side[3] = {nullptr, nullptr, nullptr};
    hipEvent_t ev_fork = nullptr, ev_join[3] = {nullptr, nullptr, nullptr};
    int* d_solve_flags = nullptr;  // scratch of the many-workgroup triangular solve (launch_dense_solve_r), allocated on first use
    static constexpr int SOLVE_FLAGS = 8192;
    std::string error;
};

struct qrk_bd_plan_s {
    qrk_handle h = nullptr;
    int64_t B = 0;
    bool uniform = true;
    int32_t r = 0, c = 0;
    int32_t mat_rows = 0, mat_cols = 0, sum_rows = 0;
    int q_format = 0, solver = 0;
    int64_t tiles_len = 0, nnz_q_tiles = 0, nnz_q = 0, nnz_r = 0;
    bool landscape = false;       // some tile has rows < cols -> InvalidInput
    int32_t max_dim = 0;          // largest tile dimension
    int32_t max_cols = 0;         // widest tile (the lanes per tile of the grouped solve kernel)
    bool factorized = false;
    // device-resident per-tile descriptors (mixed batches only)
    int32_t *d_rows = nullptr, *d_cols = nullptr, *d_coff = nullptr, *d_rowoff = nullptr;
    int64_t *d_toff = nullptr, *d_qoff = nullptr, *d_roff = nullptr;
    int32_t* d_wave_ids = nullptr;   // tiles with rows, cols <= 32: half-wave kernels
    int64_t n_wave = 0;
    int32_t* d_wg_ids = nullptr;     // larger tiles: one workgroup each (bdqr_wg.hip)
    int64_t n_wg = 0;
    double* d_workspace = nullptr;   // per-workgroup working copies of the large tiles
    int64_t ws_stride = 0;
    int num_wg = 0;
    // mid-size tiles (32 < max dim <= QRK_COL_MAX_DIM): one thread per column of [A | Q^T] (bdqr_col.hip)
    // in three size classes (columns <= 64, <= 128, <= 256), one launch each: the LDS of a launch is carved for the
    // largest tile of its class, so that small tiles run with many workgroups per CU
    struct ColClass {
        int64_t off = 0, n = 0;          // range of d_col_ids (mixed batches), largest tiles first
        int64_t ws_stride = 0;           // max rows * cols
        int64_t ws_off = 0;              // this class's part of d_col_workspace (the classes run concurrently)
        int64_t max_rc_fitting = 0;      // largest rows * cols that fits the LDS-resident form
        int max_rows = 0, max_cols = 0, w_lds = 0, num_wg = 0;
        bool onchip = false;             // every tile of the class has more than 64 columns: bdqr_reg.hip (registers + LDS) instead
    } col_cls[3];
    int32_t* d_col_ids = nullptr;
    int64_t n_col = 0;
    // tiles with 32 < rows <= 64 (cols <= rows): one wavefront each, on chip (bdqr_w64.hip); mixed batches: largest first
    int32_t* d_w64_ids = nullptr;
    int32_t w64_maxr = 0;          // tallest of them (sizes the LDS of the launch)
    int64_t n_w64 = 0;
    bool w64_uniform = false;            // a uniform batch of such tiles
    double* d_col_workspace = nullptr;   // one part per class
    // redo list of the exact path: [0], [1] = counters of this / the next factorisation (ping-pong: the exact kernel zeroes the
    // other one, so no memset sits on the stream), [2..2+B) = global tile ids
    int32_t* d_redo = nullptr;
    double* d_p4_scratch = nullptr;   // bdqr_pair4.hip (uniform 32 x 32): working copies of its exact path
    int p4_wgs = 0;
    bool k1_quad32 = false;           // bdqr_quad32.hip (four tiles per wavefront) instead of bdqr_pair4.hip; chosen in qrk_bd_plan_create
    int q32_wgs = 0;
    int redo_parity = 0;
    double* d_exact_ws = nullptr;        // working copies of tiles too large for the exact kernel's LDS
    int64_t exact_ws_stride = 0;
    int exact_num_wg = 0, exact_maxr = 0, exact_maxc = 0;
};

// One size class of the mid-size tiles: on chip (bdqr_reg.hip) when all of its tiles are wider than 64 columns, else bdqr_col.hip
static hipError_t launch_col_class(const qrk_bd_plan_s::ColClass& k, const qrk::WaveBatch& cb, const double* tiles, double* q, double* r,
                                   int32_t* perm, double* hc, double* workspace, int32_t* redo_cnt, int32_t* redo_ids, int32_t* queue,
                                   hipStream_t stream)
{
    if (k.onchip)
        return qrk::launch_bdqr_reg(cb, tiles, q, r, perm, hc, workspace, k.ws_stride, k.num_wg, k.max_rows, k.max_cols, redo_cnt, redo_ids,
                                    queue, stream);
    return qrk::launch_bdqr_col(cb, tiles, q, r, perm, hc, workspace, k.ws_stride, k.num_wg, k.max_rows, k.max_cols, k.w_lds, redo_cnt,
                                redo_ids, queue, stream);
}

struct qrk_bb_plan_s {
    qrk_handle h = nullptr;
    qrk::BandedStructure st;
    qrk::BBPanel* d_panels = nullptr;
    int32_t *d_prowptr = nullptr, *d_pcol = nullptr, *d_rcolptr = nullptr, *d_rrowidx = nullptr;
    int64_t *d_pmap = nullptr, *d_rsrc = nullptr;
    double *d_W = nullptr, *d_lo = nullptr, *d_stage = nullptr;
    bool factorized = false;       // d_stage holds the rows of R of the last factorize
};

// Banded matrix given as dense strips (include/qrkit_amd.h, qrk_bbs_*): two-stage factorisation, see banded.hip (strips form).
struct qrk_bbs_plan_s {
    qrk_handle h = nullptr;
    int64_t N = 0;                      // strips
    int32_t ms = 0, n = 0, s = 0, lo = 0;   // strip rows / columns, column step, carry size n - s
    int64_t rows = 0, cols = 0;
    qrk_bd_plan bd = nullptr;           // stage A: the strips as a block-diagonal matrix (HouseholderQR, BlockDiagonalQ)
    std::vector<qrk::BBPanel> panels;
    qrk::BBPanel* d_panels = nullptr;
    int32_t* d_rlim = nullptr;          // [2][n / 16]: staircase row limits of panel 0 / of the other panels
    int32_t* d_done = nullptr;          // [2 N + 2] rows-final words of the pipelined chain, its abort word at [N], the panels' second words (banded.hip, BBPipe)
    int64_t chain_reruns = 0;           // factorisations whose pipelined chain was given up and run again on one workgroup
    double *d_q = nullptr, *d_ra = nullptr;     // stage A: explicit Q_i (ms x ms each), packed R_i
    int32_t* d_perm = nullptr;
    double *d_y = nullptr, *d_t = nullptr, *d_stage = nullptr, *d_lo = nullptr;   // stage B: panels (Y below the diagonal), T, rows of R, carry
    int64_t y_len = 0, t_len = 0, stage_len = 0;
    int32_t max_act = 0;
    bool factorized = false;
    // the chains of Q^T b / Q x / R^-1 y through one small matrix per strip (banded_maps.hip): built on the first product after a
    // factorisation, kept until the next one; QRK_BBS_MAPS=0, or an allocation that fails, leaves the one-workgroup chains in charge
    double *d_cmap = nullptr, *d_amap = nullptr, *d_carry = nullptr;   // [N][lo][lo] each (carries / back substitution), [carry_cap][N][lo]
    double *d_cprod = nullptr, *d_aprod = nullptr, *d_gvec = nullptr;   // group products of the two-level chains, [carry_cap] boundary vectors
    int32_t maps_K = 0;                 // group size (0: one level)
    int64_t carry_cap = 0, gvec_per_rhs = 0;
    bool maps_ready = false, maps_off = false;
};

// ---- QRKit::BlockedThinSparseQR on the device (include/qrkit_amd.h, qrk_thin_*) -----------------------------------------------
namespace qrk {
// the identity permutation of an un-pivoted factorisation (no host vector, no synchronisation)
__global__ void __launch_bounds__(256) identity_perm_kernel(int32_t* __restrict__ p, int n)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) p[i] = i;
}
// columns of R that a panel finishes (BlockedThinSparseQR.h:271-279): column nzp + bc of R takes the rows above the panel from
// column c0 + p[bc] of the working matrix and the panel's own upper triangle below them
__global__ void __launch_bounds__(256)
thin_r_columns_kernel(const double* __restrict__ D, int64_t ldd, const double* __restrict__ Ji, int64_t ldj, const int32_t* __restrict__ pp,
                      int nzp, int c0, int k, int nnew, double* __restrict__ R, int64_t ldr)
{
    const int bc = blockIdx.x;
    if (bc >= nnew) return;
    const double* src = D + (int64_t)(c0 + pp[bc]) * ldd;
    double* dst = R + (int64_t)(nzp + bc) * ldr;
    for (int i = threadIdx.x; i < nzp; i += 256) dst[i] = src[i];
    for (int i = threadIdx.x; i < k; i += 256) dst[nzp + i] = i <= bc ? Ji[(int64_t)bc * ldj + i] : 0.0;
}
// [nonzeroPivots() | perm as doubles (0:nnew)] in one buffer: one small copy to the host per panel.  nonzeroPivots() as Eigen counts
// it (ColPivHouseholderQR::computeInPlace): the first step q whose biggest updated column norm, squared, is below
// threshold_helper (rows - q), threshold_helper = (largest initial norm * eps)^2 / rows, `rows` = the panel's own height.  The fast
// kernels only finish a factorisation themselves when every pivot is far above that (a pivot below 2^-30 |A| sends the block to the
// exact path, decision (5)): then the count is k.  When the exact path ran, its own norm table (m_colNormsUpdated in Eigen's
// rounding, pivn) decides -- not |R_qq|, which equals the updated norm only up to rounding.
__global__ void thin_pack_kernel(const int32_t* __restrict__ pp, int k, int nnew, int nrows, const int* __restrict__ flag, int exact_known,
                                 const double* __restrict__ pivn, double* __restrict__ out)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < nnew) out[1 + i] = (double)pp[i];
    if (i == 0) {
        const bool exact = exact_known >= 0 ? exact_known != 0 : (flag ? *flag != 0 : true);
        int nz = k;
        if (exact && k > 0) {
            const double eps = 2.220446049250313e-16, maxn = pivn[0];
            const double thr = ((maxn * eps) * (maxn * eps)) / (double)nrows;
            for (int q = 0; q < k; ++q)
                if (pivn[q] * pivn[q] < thr * (double)(nrows - q)) { nz = q; break; }
        }
        out[0] = (double)nz;
    }
}
}  // namespace qrk

struct qrk_thin_plan_s {
    qrk_handle h = nullptr;
    int32_t rows = 0, cols = 0, block_cols = 0;
    int32_t rank = 0, maxrows = 0;
    std::vector<int32_t> col_perm, row_perm;       // m_outputPerm_c.indices(), m_rowPerm.indices()
    // A panel is factorised at its OWN height (updateBlockInfo, BlockedThinSparseQR.h:203-238), rounded up to a bucket (a power of two
    // from 64 up, capped at `rows`: zero rows appended change neither the reflectors nor Eigen's sums) so that a few dense plans serve
    // every panel: hb = the bucket = leading dimension of the panel's packed QR; plan = index into dplans.
    struct Panel { int32_t r0, nrows, nnew, hb, plan; double* d_ji; double* d_hc; };
    std::vector<Panel> panels;                     // packed QR + tau of every panel (hb x nnew)
    struct BucketPlan { int32_t hb, nnew; qrk_dense_plan plan; };
    std::vector<BucketPlan> dplans;                // one dense plan per (bucket height, panel width) that occurred
    std::vector<double*> arena;                    // pooled storage of the panels (chunks; a panel never straddles two)
    size_t arena_used = 0, arena_cap = 0;          // doubles used / capacity of the last chunk
    double* d_R = nullptr;                         // cols x cols, column-major: R(0:rank, :) upper trapezoidal
    int64_t ldd = 0;
};

struct qrk_dense_plan_s {
    qrk_handle h = nullptr;
    int32_t rows = 0, cols = 0;
    int solver = 0;
    // multi-workgroup row-slab path (dense_qr_tall.hip): anything but small matrices
    bool tall = false;
    void* d_ws = nullptr;
    int G = 0, cpad = 0, rows_per = 0;
    int pers_cus = 0;          // > 0: the column-parallel stage may run as ONE persistent launch (dense_qr_pers.hip, QRK_DENSE_PERS=1 at plan creation) on that many CUs
    bool persistent = false;   // the whole factorisation as ONE cooperative kernel (all slabs resident); QRK_DENSE_PERSISTENT=1 enables
    // exact path: copy of the input (rows x cols, restored when a decision of the fast kernels was not clear of rounding),
    // the flag word of the single-workgroup kernel, workspace of the exact kernel
    double* d_copy = nullptr;
    int* d_unclear = nullptr;
    int* h_unclear = nullptr;      // pinned: where the host reads the flag word (two-stage and exact_wide plans synchronise the stream)
    double* d_exact_ws = nullptr;
    const int* last_flag = nullptr;   // device word that says whether the last factorisation went through the exact path (null: see last_exact)
    int last_exact = -1;              // 1 / 0: known on the host (forced, or read back by an exact_wide plan); -1: the device word decides
    // two-stage form of the pivoted factorisation of a tall matrix (caqr.hip): A = Q0 R0 without pivoting on the matrix cores,
    // then R0 P = Q1 R by the level-2 kernels on the n x n triangle.  d_r0 keeps the packed QR of the second stage, d_t the T
    // factors of the first; the caller's array holds the reflectors of Q0 below / inside its top triangles and R above.
    bool two_stage = false;
    bool caqr_only = false;    // un-pivoted solver (HouseholderQR) on a tall matrix: the first stage IS the factorisation, A = Q0 R0 on the matrix
                               // cores (the block-reflector update of BlockedThinQRBase::updateMat, BlockedThinQRBase.h:309-333, as MFMA tiles)
    int fmt_capable = 0;       // 0: Eigen's format only; 1: two-stage workspaces exist; 2: CAQR-only workspaces exist
    bool ts_active = false;    // the last factorisation ended in the two-stage format (false: Eigen's packed format, also after the exact path)
    const void* ts_owner = nullptr;   // ... and this is the caller's array it was computed in: the T factors and Q1 kept in the plan belong to
                                      // that factorisation only (qrk_dense_apply_q refuses any other array while ts_active)
    double* d_t = nullptr;
    double* d_r0 = nullptr;    // R0 (n x n), scratch of the second stage
    double* d_q1 = nullptr;    // packed QR of the second stage in Eigen's format (== d_r0 when the row-slab kernels factorise in place)
    double* d_t1 = nullptr;    // T factors of blocks of 32 reflectors of the second stage (dense_qr.hip: the blocked application of Q1^T)
    double* d_xw = nullptr;    // the slabs' partial products of that application, 2 x 16 x 32 per right-hand side (allocated on first use)
    int64_t xw_cap = 0;
    void* d_ws2 = nullptr;
    hipStream_t la_stream = nullptr;               // look-ahead of the first stage: the next panel is factorised beside the trailing update
    hipEvent_t la_urgent = nullptr, la_factored = nullptr;
    qrk::CaqrPipe la_pipe;                         // third stream + events of the look-ahead pipelined by levels (QRK_CAQR_PIPE=0: off)
    int G2 = 0, cpad2 = 0, rows_per2 = 0;
    bool tall2 = false, cols2 = false;
    bool exact_wide = false;   // large block: the exact path runs as a host-launched sequence over all CUs (the host reads the unclear word)
    // column-parallel kernel as the DIRECT algorithm (same reflectors, signs and format as Eigen) when a column fits LDS: d_q1 then
    // receives the packed result (rows x cols), copied back into the caller's array
    bool cols_direct = false;
};

namespace {

qrk_status fail(qrk_handle h, qrk_status st, const std::string& msg)
{
    if (h) h->error = msg; else g_create_error = msg;
    return st;
}

#define QRK_HIP(h, expr)                                                                       \
    do {                                                                                       \
        hipError_t e_ = (expr);                                                                \
        if (e_ != hipSuccess)                                                                  \
            return fail((h), QRK_STATUS_HIP_ERROR,                                             \
                        std::string(#expr) + ": " + hipGetErrorString(e_));                    \
    } while (0)

// The handle's pool of side streams (fork after / join into the caller's stream), created ONCE, when the handle is created: the classes of a
// mixed batch run on them side by side, and the look-ahead of the dense solver's first stage uses the two high-priority ones.  Why a
// pool and why eagerly: ROCm maps streams onto a few hardware queues in creation order, and streams that share a queue serialise.
// With per-plan streams created on first use, the SAME 40 000 x 2 000 factorisation took 47 ms in a fresh process and 59 ms after a
// mixed batch had created its three streams first (the look-ahead stream then shared a queue with the caller's: profiles/
// r04_stream_pool.txt); with the pool the assignment is fixed at qrk_create, before the caller's workload has created anything.
qrk_status ensure_pool(qrk_handle h)
{
    if (h->ev_fork) return QRK_STATUS_OK;
    QRK_HIP(h, hipEventCreateWithFlags(&h->ev_fork, hipEventDisableTiming));
    int prio_least = 0, prio_greatest = 0;
    QRK_HIP(h, hipDeviceGetStreamPriorityRange(&prio_least, &prio_greatest));
    const bool use_prio = !(std::getenv("QRK_COL_PRIO") && std::atoi(std::getenv("QRK_COL_PRIO")) == 0);
    for (int z = 2; z >= 0; --z) {
        // side[2], side[1]: highest priority (the class of the largest tiles of a mixed batch -- the critical path of the batch -- and the
        // next one; the look-ahead panel and the urgent applies of the dense solver: a panel workgroup needs the LDS of one apply
        // workgroup and should get the next slot that frees up); side[0]: lowest
        const int prio = !use_prio ? prio_least : (z >= 1 ? prio_greatest : prio_least);
        QRK_HIP(h, hipStreamCreateWithPriority(&h->side[z], hipStreamNonBlocking, prio));
        QRK_HIP(h, hipEventCreateWithFlags(&h->ev_join[z], hipEventDisableTiming));
    }
    return QRK_STATUS_OK;
}

template <typename T>
qrk_status upload(qrk_handle h, const std::vector<T>& v, T** out)
{
    *out = nullptr;
    if (v.empty()) return QRK_STATUS_OK;
    QRK_HIP(h, hipMalloc((void**)out, v.size() * sizeof(T)));
    QRK_HIP(h, hipMemcpyAsync(*out, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice, h->stream));
    QRK_HIP(h, hipStreamSynchronize(h->stream));
    return QRK_STATUS_OK;
}

qrk::TileGeom make_geom(const qrk_bd_plan_s* p)
{
    qrk::TileGeom g{};
    g.num_tiles = p->B;
    g.rows = p->r; g.cols = p->c;
    g.t_rows = p->uniform ? nullptr : p->d_rows;
    g.t_cols = p->d_cols; g.q_off = p->d_qoff; g.r_off = p->d_roff;
    g.c_off = p->d_coff; g.row_off = p->d_rowoff;
    g.mat_rows = p->mat_rows; g.mat_cols = p->mat_cols; g.sum_rows = p->sum_rows;
    g.nnz_q_tiles = p->nnz_q_tiles; g.q_format = p->q_format;
    return g;
}

// Scoped device staging of host buffers for the QRK_MEM_HOST entry points.
struct Staging {
    qrk_handle h;
    std::vector<void*> bufs;
    explicit Staging(qrk_handle hh) : h(hh) {}
    ~Staging() { for (void* b : bufs) (void)hipFree(b); }
    template <typename T>
    qrk_status in(const T* host, int64_t n, T** dev)
    {
        *dev = nullptr;
        if (n <= 0 || !host) return QRK_STATUS_OK;
        QRK_HIP(h, hipMalloc((void**)dev, (size_t)n * sizeof(T)));
        bufs.push_back(*dev);
        QRK_HIP(h, hipMemcpyAsync(*dev, host, (size_t)n * sizeof(T), hipMemcpyHostToDevice, h->stream));
        return QRK_STATUS_OK;
    }
    template <typename T>
    qrk_status out(int64_t n, T** dev)
    {
        *dev = nullptr;
        if (n <= 0) return QRK_STATUS_OK;
        QRK_HIP(h, hipMalloc((void**)dev, (size_t)n * sizeof(T)));
        bufs.push_back(*dev);
        return QRK_STATUS_OK;
    }
    template <typename T>
    qrk_status back(T* host, const T* dev, int64_t n)
    {
        if (n <= 0 || !host || !dev) return QRK_STATUS_OK;
        QRK_HIP(h, hipMemcpyAsync(host, dev, (size_t)n * sizeof(T), hipMemcpyDeviceToHost, h->stream));
        return QRK_STATUS_OK;
    }
};

qrk_status enqueue_factorize(qrk_bd_plan_s* p, const double* tiles, double* q, double* r, int32_t* perm,
                             double* hc)
{
    qrk_handle h = p->h;
    qrk::WaveBatch nb{};
    nb.pivoting = p->solver == QRK_COLPIV_HOUSEHOLDER ? 1 : 0;
    // every tile of the plan, as the exact path addresses them (global tile index)
    qrk::WaveBatch all = nb;
    all.num_tiles = p->B; all.rows = p->r; all.cols = p->c;
    if (!p->uniform) {
        all.t_rows = p->d_rows; all.t_cols = p->d_cols; all.t_off = p->d_toff;
        all.q_off = p->d_qoff; all.r_off = p->d_roff; all.c_off = p->d_coff;
    }
    if (h->force_exact) {
        QRK_HIP(h, qrk::launch_bdqr_exact(all, nullptr, nullptr, nullptr, tiles, q, r, perm, hc, p->d_exact_ws, p->exact_ws_stride,
                                          p->exact_num_wg, p->exact_maxr, p->exact_maxc, h->stream));
        qrk::launch_bd_q_tail_ones(q, p->nnz_q_tiles, p->nnz_q - p->nnz_q_tiles, h->stream);
        QRK_HIP(h, hipGetLastError());
        return QRK_STATUS_OK;
    }
    int32_t* const redo_cnt = p->d_redo + p->redo_parity;
    int32_t* const redo_next = p->d_redo + (p->redo_parity ^ 1);
    int32_t* const redo_ids = p->d_redo + 2;
    bool redo_pass = true;     // a launch of the exact kernel follows (it reads the redo list the fast kernels filled)
    if (p->uniform) {
        nb.num_tiles = p->B; nb.rows = p->r; nb.cols = p->c;
        const bool full32 = p->r == 32 && p->c == 32 &&
                            ((reinterpret_cast<uintptr_t>(tiles) | reinterpret_cast<uintptr_t>(q) |
                              reinterpret_cast<uintptr_t>(r)) & 15u) == 0;
        if (p->w64_uniform)
            QRK_HIP(h, qrk::launch_bdqr_w64(nb, tiles, q, r, perm, hc, h->num_cus, p->r, redo_cnt, redo_ids, p->d_redo + 2 + p->B + 3, h->stream));
        else if (p->max_dim > 32 && p->max_dim <= QRK_COL_MAX_DIM) {
            const auto& k = p->col_cls[0];
            QRK_HIP(h, launch_col_class(k, nb, tiles, q, r, perm, hc, p->d_col_workspace, redo_cnt, redo_ids, p->d_redo + 2 + p->B, h->stream));
        } else if (p->max_dim > 32)
            qrk::launch_bdqr_wg(nb, tiles, q, r, perm, hc, p->d_workspace, p->ws_stride, p->num_wg, p->max_dim, redo_cnt, redo_ids, h->stream);
        else if (p->max_dim <= 16 && p->r >= p->c && p->c <= 2 && h->use_small_kernel && h->use_thin_kernel)   // a tile per lane (bdqr_thin.hip)
            qrk::launch_bdqr_thin(p->B, p->r, p->c, nb.pivoting, tiles, q, r, perm, hc, h->num_cus * 32, redo_cnt, redo_ids, h->stream);
        else if (p->max_dim <= 16 && h->use_small_kernel && h->use_quad_kernel && p->r >= h->quad_min_rows && qrk::bdqr_quad_supported(p->r, p->c))   // four or eight tiles per wavefront (bdqr_quad.hip)
        {
            // (four or eight tiles per wavefront; it redoes its flagged tiles itself: no launch of the exact kernel behind it)
            qrk::launch_bdqr_quad(p->B, p->r, p->c, nb.pivoting, tiles, q, r, perm, hc, h->num_cus * qrk::bdqr_quad_waves_per_cu(p->r), redo_cnt, redo_ids, h->stream);
            redo_pass = false;
        }
        else if (p->max_dim <= 16 && p->r >= p->c && h->use_small_kernel)   // 64/G tiles per wavefront (bdqr_small.hip)
            qrk::launch_bdqr_small(p->B, p->r, p->c, nb.pivoting, tiles, q, r, perm, hc, h->num_cus * 32, redo_cnt, redo_ids, h->stream);
        else if (p->d_p4_scratch) {
            // the two-phase 32 x 32 kernels: bdqr_quad32.hip (four tiles per wavefront, two waves per SIMD) or bdqr_pair4.hip (two tiles,
            // four waves per SIMD); both redo their flagged tiles themselves
            if (p->k1_quad32)
                QRK_HIP(h, qrk::launch_bdqr_quad32(p->B, nb.pivoting, tiles, q, r, perm, hc, p->d_p4_scratch, p->q32_wgs, -1, h->stream));
            else
                QRK_HIP(h, qrk::launch_bdqr_pair4(p->B, nb.pivoting, tiles, q, r, perm, hc, p->d_p4_scratch, p->p4_wgs, h->stream));
            redo_pass = false;
        } else {
            int wgs = h->num_cus * h->pair_wgs_per_cu;
            if (const char* e = std::getenv("QRK_PAIR_WGS")) { const int v = std::atoi(e); if (v > 0) wgs = v; }   // (experiments)
            qrk::launch_bdqr_pair(nb, full32, tiles, q, r, perm, hc, wgs, redo_cnt, redo_ids, h->stream);
            // the persistent 32 x 32 kernel redoes its flagged tiles itself (bdqr_pair.hip, redo_exact32): nothing queued behind it
            if (full32) redo_pass = false;
        }
    } else {
        nb.num_tiles = p->n_wave; nb.tile_ids = p->d_wave_ids;
        nb.t_rows = p->d_rows; nb.t_cols = p->d_cols; nb.t_off = p->d_toff;
        nb.q_off = p->d_qoff; nb.r_off = p->d_roff; nb.c_off = p->d_coff;
        qrk::launch_bdqr_pair(nb, false, tiles, q, r, perm, hc, h->num_cus * h->pair_wgs_per_cu, redo_cnt, redo_ids, h->stream);
        if (p->n_w64 > 0) {
            qrk::WaveBatch wb = nb;
            wb.num_tiles = p->n_w64; wb.tile_ids = p->d_w64_ids;
            QRK_HIP(h, qrk::launch_bdqr_w64(wb, tiles, q, r, perm, hc, h->num_cus, p->w64_maxr, redo_cnt, redo_ids, p->d_redo + 2 + p->B + 3, h->stream));
        }
        // the size classes are independent of each other and of the small tiles above: each on its own side stream
        // (forked after what is already queued on the caller's stream, joined back below), so that the tail of one
        // launch overlaps the others
        // Side by side by default since the large classes run on chip (round 3: they no longer cost each other cache, and the small
        // workgroups of the other classes fill the CUs that the tail of the 512-thread launch leaves idle: 4 000 mixed tiles 8.2 ->
        // 7.3 ms); QRK_COL_CONCURRENT=0 / 1 forces one way, and with QRK_COL_ONCHIP=0 the round-2 finding stands (serial).
        static const bool col_serial = [] {
            if (const char* e = std::getenv("QRK_COL_CONCURRENT")) return std::atoi(e) != 1;
            return std::getenv("QRK_COL_ONCHIP") && std::atoi(std::getenv("QRK_COL_ONCHIP")) == 0;
        }();
        if (p->n_col > 0 && col_serial) {
            // one class after the other on the caller's stream, the largest tiles first.  Measured alone the classes of 4 000 mixed
            // 8...256 tiles take 12.7 + 3.65 + 0.53 = 16.9 ms; launched side by side on three streams (QRK_COL_CONCURRENT=1) the
            // batch takes 18 or 22 ms from run to run: what the classes cost each other in L2 / Infinity Cache is more than the
            // tails they hide
            for (int z = 2; z >= 0; --z) {
                const auto& k = p->col_cls[z];
                if (k.n <= 0) continue;
                qrk::WaveBatch cb = nb;
                cb.num_tiles = k.n; cb.tile_ids = p->d_col_ids + k.off;
                QRK_HIP(h, launch_col_class(k, cb, tiles, q, r, perm, hc, p->d_col_workspace + k.ws_off, redo_cnt, redo_ids, p->d_redo + 2 + p->B + z, h->stream));
            }
        } else if (p->n_col > 0) {
            if (qrk_status stp = ensure_pool(h)) return stp;
            QRK_HIP(h, hipEventRecord(h->ev_fork, h->stream));
            for (int z = 2; z >= 0; --z) {            // (largest class first)
                const auto& k = p->col_cls[z];
                if (k.n <= 0) continue;
                qrk::WaveBatch cb = nb;
                cb.num_tiles = k.n; cb.tile_ids = p->d_col_ids + k.off;
                QRK_HIP(h, hipStreamWaitEvent(h->side[z], h->ev_fork, 0));
                QRK_HIP(h, launch_col_class(k, cb, tiles, q, r, perm, hc, p->d_col_workspace + k.ws_off, redo_cnt, redo_ids, p->d_redo + 2 + p->B + z, h->side[z]));
                QRK_HIP(h, hipEventRecord(h->ev_join[z], h->side[z]));
                QRK_HIP(h, hipStreamWaitEvent(h->stream, h->ev_join[z], 0));
            }
        }
        if (p->n_wg > 0) {
            qrk::WaveBatch lb = nb;
            lb.num_tiles = p->n_wg; lb.tile_ids = p->d_wg_ids;
            qrk::launch_bdqr_wg(lb, tiles, q, r, perm, hc, p->d_workspace, p->ws_stride, p->num_wg, p->max_dim, redo_cnt, redo_ids, h->stream);
        }
    }
    // the tiles whose decisions were not clear of rounding, again, with the reference's own operation order (bdqr_exact.hip);
    // the list is empty on generic data and the workgroups return at once
    if (redo_pass) {
        QRK_HIP(h, qrk::launch_bdqr_exact(all, redo_ids, redo_cnt, redo_next, tiles, q, r, perm, hc, p->d_exact_ws, p->exact_ws_stride,
                                          p->exact_num_wg, p->exact_maxr, p->exact_maxc, h->stream));
        p->redo_parity ^= 1;
    }
    qrk::launch_bd_q_tail_ones(q, p->nnz_q_tiles, p->nnz_q - p->nnz_q_tiles, h->stream);
    QRK_HIP(h, hipGetLastError());
    return QRK_STATUS_OK;
}

}  // namespace

extern "C" {

int qrk_version(void) { return QRK_VERSION_MAJOR * 1000 + QRK_VERSION_MINOR; }

int qrk_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

qrk_status qrk_create(qrk_handle* out, int device, void* stream)
{
    if (!out) return fail(nullptr, QRK_STATUS_INVALID_ARGUMENT, "qrk_create: out is NULL");
    *out = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0)
        return fail(nullptr, QRK_STATUS_NO_DEVICE,
                    "qrk_create: no HIP device visible; this library has no CPU fallback");
    if (device < 0 || device >= n)
        return fail(nullptr, QRK_STATUS_INVALID_ARGUMENT, "qrk_create: device index out of range");
    qrk_context_s* h = new (std::nothrow) qrk_context_s();
    if (!h) return fail(nullptr, QRK_STATUS_ALLOC_FAILED, "qrk_create: out of host memory");
    h->device = device;
    h->stream = static_cast<hipStream_t>(stream);
    if (const char* k = std::getenv("QRK_EXACT")) h->force_exact = k[0] == '1';
    if (const char* k = std::getenv("QRK_SMALL")) h->use_small_kernel = k[0] != '0';
    if (const char* k = std::getenv("QRK_QUAD")) h->use_quad_kernel = k[0] != '0';
    if (const char* k = std::getenv("QRK_QUAD_MIN_ROWS")) h->quad_min_rows = std::atoi(k);
    if (const char* k = std::getenv("QRK_THIN")) h->use_thin_kernel = k[0] != '0';
    if (const char* k = std::getenv("QRK_PAIR_WGS_PER_CU")) { const int v = std::atoi(k); if (v > 0) h->pair_wgs_per_cu = v; }
    if (hipSetDevice(device) != hipSuccess) {
        delete h;
        return fail(nullptr, QRK_STATUS_HIP_ERROR, "qrk_create: hipSetDevice failed");
    }
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) == hipSuccess) {
        h->num_cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
        if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
            std::string msg = std::string("qrk_create: device is ") + prop.gcnArchName +
                              ", this library is built for gfx950 (MI355X) only";
            delete h;
            return fail(nullptr, QRK_STATUS_UNSUPPORTED, msg);
        }
    }
    if (qrk_status stp = ensure_pool(h)) { const std::string msg = h->error; qrk_destroy(h); return fail(nullptr, stp, msg); }
    *out = h;
    return QRK_STATUS_OK;
}

qrk_status qrk_destroy(qrk_handle h)
{
    if (h) {
        for (int z = 0; z < 3; ++z) {
            if (h->side[z]) (void)hipStreamDestroy(h->side[z]);
            if (h->ev_join[z]) (void)hipEventDestroy(h->ev_join[z]);
        }
        if (h->ev_fork) (void)hipEventDestroy(h->ev_fork);
        if (h->d_solve_flags) (void)hipFree(h->d_solve_flags);
    }
    delete h;
    return QRK_STATUS_OK;
}

qrk_status qrk_set_stream(qrk_handle h, void* stream)
{
    if (!h) return QRK_STATUS_INVALID_ARGUMENT;
    h->stream = static_cast<hipStream_t>(stream);
    return QRK_STATUS_OK;
}

qrk_status qrk_synchronize(qrk_handle h)
{
    if (!h) return QRK_STATUS_INVALID_ARGUMENT;
    QRK_HIP(h, hipStreamSynchronize(h->stream));
    return QRK_STATUS_OK;
}

const char* qrk_last_error(qrk_handle h) { return h ? h->error.c_str() : g_create_error.c_str(); }

qrk_status qrk_device_alloc(qrk_handle h, int64_t bytes, void** out)
{
    if (!h || !out || bytes < 0) return fail(h, QRK_STATUS_INVALID_ARGUMENT, "qrk_device_alloc: bad argument");
    *out = nullptr;
    if (bytes == 0) return QRK_STATUS_OK;
    QRK_HIP(h, hipSetDevice(h->device));
    if (hipMalloc(out, (size_t)bytes) != hipSuccess) { *out = nullptr; return fail(h, QRK_STATUS_ALLOC_FAILED, "qrk_device_alloc: out of device memory"); }
    return QRK_STATUS_OK;
}

qrk_status qrk_device_free(qrk_handle h, void* ptr)
{
    if (!h) return QRK_STATUS_INVALID_ARGUMENT;
    if (ptr) { QRK_HIP(h, hipSetDevice(h->device)); QRK_HIP(h, hipStreamSynchronize(h->stream)); QRK_HIP(h, hipFree(ptr)); }
    return QRK_STATUS_OK;
}

qrk_status qrk_memcpy(qrk_handle h, void* dst, const void* src, int64_t bytes, int direction)
{
    if (!h || bytes < 0 || (bytes > 0 && (!dst || !src)) || (direction != 0 && direction != 1))
        return fail(h, QRK_STATUS_INVALID_ARGUMENT, "qrk_memcpy: bad argument");
    if (bytes == 0) return QRK_STATUS_OK;
    QRK_HIP(h, hipSetDevice(h->device));
    QRK_HIP(h, hipMemcpyAsync(dst, src, (size_t)bytes, direction == 0 ? hipMemcpyHostToDevice : hipMemcpyDeviceToHost, h->stream));
    QRK_HIP(h, hipStreamSynchronize(h->stream));
    return QRK_STATUS_OK;
}

qrk_status qrk_memcpy_2d(qrk_handle h, void* dst, int64_t dst_pitch, const void* src, int64_t src_pitch, int64_t width_bytes,
                         int64_t height, int direction)
{
    if (!h || width_bytes < 0 || height < 0 || dst_pitch < width_bytes || src_pitch < width_bytes || direction < 0 || direction > 2 ||
        (width_bytes > 0 && height > 0 && (!dst || !src)))
        return fail(h, QRK_STATUS_INVALID_ARGUMENT, "qrk_memcpy_2d: bad argument");
    if (width_bytes == 0 || height == 0) return QRK_STATUS_OK;
    QRK_HIP(h, hipSetDevice(h->device));
    const hipMemcpyKind kind = direction == 0 ? hipMemcpyHostToDevice : direction == 1 ? hipMemcpyDeviceToHost : hipMemcpyDeviceToDevice;
    QRK_HIP(h, hipMemcpy2DAsync(dst, (size_t)dst_pitch, src, (size_t)src_pitch, (size_t)width_bytes, (size_t)height, kind, h->stream));
    QRK_HIP(h, hipStreamSynchronize(h->stream));
    return QRK_STATUS_OK;
}

qrk_status qrk_dense_gemv_sub(qrk_handle h, const double* S, int64_t lds, int64_t rows, int64_t cols, const int32_t* colidx,
                              const double* z, double* y)
{
    if (!h || rows < 0 || cols < 0 || lds < rows || (rows > 0 && cols > 0 && (!S || !z || !y)))
        return fail(h, QRK_STATUS_INVALID_ARGUMENT, "qrk_dense_gemv_sub: bad argument");
    if (rows == 0 || cols == 0) return QRK_STATUS_OK;
    QRK_HIP(h, hipSetDevice(h->device));
    QRK_HIP(h, qrk::launch_gemv_sub(S, lds, rows, cols, colidx, z, y, h->stream));
    return QRK_STATUS_OK;
}

qrk_status qrk_sparse_window_to_dense(qrk_handle h, int row_major, int64_t rows, int64_t cols, const int32_t* d_outer,
                                      const int32_t* d_inner, const double* d_values, int64_t row0, int64_t nrows,
                                      const int32_t* d_row_map, double* d_out, int64_t ld)
{
    if (!h || rows < 0 || cols < 0 || row0 < 0 || nrows < 0 || row0 + nrows > rows || ld < nrows ||
        (nrows > 0 && cols > 0 && (!d_outer || !d_out)))
        return fail(h, QRK_STATUS_INVALID_ARGUMENT, "qrk_sparse_window_to_dense: bad argument");
    if (nrows == 0 || cols == 0) return QRK_STATUS_OK;
    QRK_HIP(h, hipSetDevice(h->device));
    QRK_HIP(h, qrk::launch_sparse_window_to_dense(row_major != 0, rows, cols, d_outer, d_inner, d_values, row0, nrows, d_row_map, d_out,
                                                  ld, h->stream));
    return QRK_STATUS_OK;
}

qrk_status qrk_bd_plan_create(qrk_handle h, const qrk_bd_layout* L, qrk_q_format q_format,
                              qrk_block_solver solver, qrk_bd_plan* out)
{
    if (!h || !L || !out) return fail(h, QRK_STATUS_INVALID_ARGUMENT, "qrk_bd_plan_create: NULL argument");
    *out = nullptr;
    if (L->num_blocks < 0 || (L->rows == nullptr) != (L->cols == nullptr))
        return fail(h, QRK_STATUS_INVALID_ARGUMENT, "qrk_bd_plan_create: bad layout");
    if (q_format != QRK_FULL_Q && q_format != QRK_BLOCK_DIAGONAL_Q)
        return fail(h, QRK_STATUS_INVALID_ARGUMENT, "qrk_bd_plan_create: unknown Q format");
    if (solver != QRK_COLPIV_HOUSEHOLDER && solver != QRK_HOUSEHOLDER)
        return fail(h, QRK_STATUS_INVALID_ARGUMENT, "qrk_bd_plan_create: unknown block solver");
    QRK_HIP(h, hipSetDevice(h->device));

    qrk_bd_plan_s* p = new (std::nothrow) qrk_bd_plan_s();
    if (!p) return fail(h, QRK_STATUS_ALLOC_FAILED, "qrk_bd_plan_create: out of host memory");
    p->h = h; p->B = L->num_blocks; p->uniform = L->rows == nullptr;
    p->q_format = q_format; p->solver = solver;
    p->mat_rows = L->mat_rows; p->mat_cols = L->mat_cols;

    const int64_t B = p->B;
    int64_t sum_rows = 0, sum_cols = 0;
    std::vector<int32_t> coff, rowoff, wave_ids, wg_ids, col_ids, col_bin[3], w64_ids;
    // 32 < rows <= 64: one wavefront per tile, on chip (bdqr_w64.hip); QRK_W64=0 keeps bdqr_col.hip's LDS-resident form (cross-check)
    static const bool use_w64 = !(std::getenv("QRK_W64") && std::atoi(std::getenv("QRK_W64")) == 0);
    int64_t ws_stride = 0;
    std::vector<int64_t> toff, qoff, roff;
    if (p->uniform) {
        p->r = L->block_rows; p->c = L->block_cols;
        if (p->r <= 0 || p->c <= 0) { delete p; return fail(h, QRK_STATUS_INVALID_ARGUMENT, "qrk_bd_plan_create: non-positive tile size"); }
        sum_rows = B * p->r; sum_cols = B * p->c;
        p->tiles_len = B * (int64_t)p->r * p->c;
        p->nnz_q_tiles = B * (int64_t)p->r * p->r;
        p->nnz_r = B * (int64_t)(p->c * (p->c + 1) / 2);
        p->landscape = p->r < p->c;
        p->max_dim = p->r > p->c ? p->r : p->c;
        p->max_cols = p->c;
        if (p->max_dim > QRK_COL_MAX_DIM) ws_stride = (int64_t)p->r * p->c;
        else if (use_w64 && !p->landscape && p->max_dim > 32 && qrk::bdqr_w64_supported(p->r, p->c)) p->w64_uniform = true;
        else if (p->max_dim > 32 && !p->landscape) {
            auto& k = p->col_cls[0];
            k.n = B; k.ws_stride = (int64_t)p->r * p->c; k.max_rows = p->r; k.max_cols = p->c;
            if (k.ws_stride <= qrk::QRK_COL_W_LDS_MAX) k.max_rc_fitting = k.ws_stride;
        }
    } else {
        coff.resize(B); rowoff.resize(B); toff.resize(B); qoff.resize(B); roff.resize(B);
        for (int64_t i = 0; i < B; ++i) {
            const int32_t r = L->rows[i], c = L->cols[i];
            if (r <= 0 || c <= 0) { delete p; return fail(h, QRK_STATUS_INVALID_ARGUMENT, "qrk_bd_plan_create: non-positive tile size"); }
            toff[i] = p->tiles_len; qoff[i] = p->nnz_q_tiles; roff[i] = p->nnz_r;
            coff[i] = (int32_t)sum_cols; rowoff[i] = (int32_t)sum_rows;
            p->tiles_len += (int64_t)r * c;
            p->nnz_q_tiles += (int64_t)r * r;
            p->nnz_r += (int64_t)c * (c + 1) / 2;
            sum_rows += r; sum_cols += c;
            if (r < c) p->landscape = true;
            const int32_t md = r > c ? r : c;
            if (md > p->max_dim) p->max_dim = md;
            if (c > p->max_cols) p->max_cols = c;
            if (md <= 32) wave_ids.push_back((int32_t)i);
            else if (use_w64 && qrk::bdqr_w64_supported(r, c)) { w64_ids.push_back((int32_t)i); p->w64_maxr = std::max(p->w64_maxr, r); }
            else if (md <= QRK_COL_MAX_DIM && r >= c) {
                // classes: up to 64 columns (one wave per tile, many workgroups per CU) and the rest (bdqr_col.hip: own instantiation
                // and LDS layout each)
                // (with the tiles of a launch handed out largest-first through a queue, one launch for everything wider than 64 columns
                //  balances itself: 4 000 mixed 8...256 tiles 17.5 ms, against 19.1 with the tiles of up to 160 rows in a launch of
                //  their own (QRK_COL_MERGE=0) -- the split dates from the grid-stride assignment)
                static const bool merge12 = !(std::getenv("QRK_COL_MERGE") && std::atoi(std::getenv("QRK_COL_MERGE")) == 0);
                // on chip (bdqr_reg.hip) the tiles of at most 128 columns and 192 rows form a class of their own: two workgroups of 256
                // threads per CU instead of one of 512
                static const bool onchip_classes = !(std::getenv("QRK_COL_ONCHIP") && std::atoi(std::getenv("QRK_COL_ONCHIP")) == 0);
                const int z = c <= 64 ? 0 : (onchip_classes ? (qrk::bdqr_reg_small(r, c) ? 1 : 2) : ((r <= 160 && !merge12) ? 1 : 2));
                auto& k = p->col_cls[z];
                col_bin[z].push_back((int32_t)i);
                const int64_t rc = (int64_t)r * c;
                if (rc > k.ws_stride) k.ws_stride = rc;
                if (rc <= qrk::QRK_COL_W_LDS_MAX && rc > k.max_rc_fitting) k.max_rc_fitting = rc;
                if (r > k.max_rows) k.max_rows = r;
                if (c > k.max_cols) k.max_cols = c;
            }
            else { wg_ids.push_back((int32_t)i); if ((int64_t)r * c > ws_stride) ws_stride = (int64_t)r * c; }
        }
    }
    if (sum_cols != p->mat_cols || sum_rows > p->mat_rows || sum_rows > INT32_MAX || sum_cols > INT32_MAX) {
        delete p;
        return fail(h, QRK_STATUS_INVALID_ARGUMENT,
                    "qrk_bd_plan_create: mat_cols must equal the sum of tile cols and mat_rows must cover the tile rows");
    }
    p->sum_rows = (int32_t)sum_rows;
    p->nnz_q = p->nnz_q_tiles + (p->mat_rows - sum_rows);
    if (p->max_dim > QRK_WG_MAX_DIM && !p->landscape) {
        delete p;
        return fail(h, QRK_STATUS_UNSUPPORTED, "qrk_bd_plan_create: tile dimension above 2048 is not supported");
    }
    if (!p->landscape) {
        size_t ws_doubles = 0;
        for (int z = 0; z < 3; ++z) {
            auto& k = p->col_cls[z];
            if (!p->uniform) {
                // largest tiles first: the grid-stride assignment then ends with the small ones
                std::stable_sort(col_bin[z].begin(), col_bin[z].end(), [&](int32_t a, int32_t b) {
                    return (int64_t)L->rows[a] * L->rows[a] * L->cols[a] > (int64_t)L->rows[b] * L->rows[b] * L->cols[b];
                });
                k.off = (int64_t)col_ids.size(); k.n = (int64_t)col_bin[z].size();
                col_ids.insert(col_ids.end(), col_bin[z].begin(), col_bin[z].end());
            }
            if (k.n <= 0 || k.ws_stride <= 0) { k.n = p->uniform ? k.n : 0; continue; }
            k.w_lds = qrk::bdqr_col_w_lds(k.ws_stride, k.max_rc_fitting);
            int64_t wgs = (int64_t)h->num_cus * qrk::bdqr_col_wgs_per_cu(k.max_cols, k.w_lds, k.max_rows);
            // tiles wider than 64 columns: factorised on chip, one workgroup of 512 threads per CU (QRK_COL_ONCHIP=0: bdqr_col.hip's
            // global-workspace form, for comparison)
            static const bool use_onchip = !(std::getenv("QRK_COL_ONCHIP") && std::atoi(std::getenv("QRK_COL_ONCHIP")) == 0);
            k.onchip = use_onchip && (p->uniform ? p->c > 64 : z >= 1);
            if (k.onchip) {
                wgs = (int64_t)h->num_cus * (qrk::bdqr_reg_small(k.max_rows, k.max_cols) ? 2 : 1);
                k.w_lds = 0; k.ws_stride = qrk::bdqr_reg_ws_doubles();
            }
            if (const char* e = std::getenv("QRK_COL_WGS")) { const long v = std::atol(e); if (v > 0) wgs = v; }
            k.num_wg = (int)(k.n < wgs ? k.n : wgs);
            if (k.ws_stride > k.w_lds) {       // some tile of the class works in (or, on chip, is dumped to) global memory
                k.ws_off = (int64_t)ws_doubles;
                ws_doubles += (size_t)k.num_wg * (size_t)k.ws_stride;
            }
        }
        if (ws_doubles > 0 && hipMalloc((void**)&p->d_col_workspace, ws_doubles * sizeof(double)) != hipSuccess) {
            delete p;
            return fail(h, QRK_STATUS_ALLOC_FAILED, "qrk_bd_plan_create: cannot allocate the mid-size-tile workspace");
        }
    }
    if (ws_stride > 0 && !p->landscape) {
        const int64_t n_large = p->uniform ? B : (int64_t)wg_ids.size();
        p->num_wg = (int)(n_large < 2 * (int64_t)h->num_cus ? n_large : 2 * (int64_t)h->num_cus);
        p->ws_stride = ws_stride;
        if (hipMalloc((void**)&p->d_workspace, (size_t)p->num_wg * (size_t)ws_stride * sizeof(double)) != hipSuccess) {
            delete p;
            return fail(h, QRK_STATUS_ALLOC_FAILED, "qrk_bd_plan_create: cannot allocate the large-tile workspace");
        }
    }
    if (!p->landscape && B > 0) {
        // exact path: redo list + workspace for the tiles that do not fit its LDS
        int32_t maxr = p->r, maxc = p->c;
        if (!p->uniform) { maxr = 0; maxc = 0; for (int64_t i = 0; i < B; ++i) { maxr = std::max(maxr, L->rows[i]); maxc = std::max(maxc, L->cols[i]); } }
        p->exact_maxr = maxr; p->exact_maxc = maxc;
        int64_t wgs = 2 * (int64_t)h->num_cus;
        if (qrk::bdqr_exact_needs_workspace(maxr, maxc)) {
            p->exact_ws_stride = (int64_t)maxr * maxc;
            const int64_t cap = ((int64_t)1 << 31) / (p->exact_ws_stride * (int64_t)sizeof(double));   // at most 2 GiB of workspace
            wgs = std::max<int64_t>(1, std::min(wgs, cap));
        }
        p->exact_num_wg = (int)std::min<int64_t>(wgs, B);
        if (hipMalloc((void**)&p->d_redo, (size_t)(B + 2 + 4) * sizeof(int32_t)) != hipSuccess ||
            hipMemset(p->d_redo, 0, 2 * sizeof(int32_t)) != hipSuccess ||   /* synchronous: the first factorisation may run on another stream */
            (p->exact_ws_stride > 0 &&
             hipMalloc((void**)&p->d_exact_ws, (size_t)p->exact_num_wg * (size_t)p->exact_ws_stride * sizeof(double)) != hipSuccess)) {
            qrk_bd_plan_destroy(p);
            return fail(h, QRK_STATUS_ALLOC_FAILED, "qrk_bd_plan_create: cannot allocate the redo list / workspace of the exact path");
        }
    }
    // uniform 32 x 32 batches: the second-generation kernel (bdqr_pair4.hip); QRK_PAIR_V2=0 keeps the first (bdqr_pair.hip)
    if (p->uniform && p->r == 32 && p->c == 32 && B > 0 && !(std::getenv("QRK_PAIR_V2") && std::atoi(std::getenv("QRK_PAIR_V2")) == 0)) {
        p->p4_wgs = h->num_cus * 16;
        if (const char* e = std::getenv("QRK_PAIR_WGS")) { const int v = std::atoi(e); if (v > 0) p->p4_wgs = v; }
        // (one working copy per workgroup that can exist: the launch has min(pairs, p4_wgs) of them)
        const int64_t p4_live = std::min<int64_t>((B + 1) / 2, p->p4_wgs);
        // QRK_K1_FORM: quad32 / pair4 force the kernel (diagnostic); otherwise by the launch size
        p->q32_wgs = h->num_cus * 8;
        if (const char* e = std::getenv("QRK_Q32_WGS")) { const int v = std::atoi(e); if (v > 0) p->q32_wgs = v; }
        p->k1_quad32 = qrk::bdqr_quad32_preferred(B, p->q32_wgs);
        if (const char* e = std::getenv("QRK_K1_FORM")) p->k1_quad32 = std::strcmp(e, "quad32") == 0 ? true : (std::strcmp(e, "pair4") == 0 ? false : p->k1_quad32);
        const int64_t q32_live = std::min<int64_t>((B + 3) / 4, p->q32_wgs);
        const int64_t scratch_doubles = std::max(qrk::bdqr_pair4_scratch_doubles((int)p4_live), qrk::bdqr_quad32_scratch_doubles((int)q32_live));
        if (hipMalloc((void**)&p->d_p4_scratch, (size_t)scratch_doubles * sizeof(double)) != hipSuccess) {
            qrk_bd_plan_destroy(p);
            return fail(h, QRK_STATUS_ALLOC_FAILED, "qrk_bd_plan_create: cannot allocate the scratch of the 32 x 32 kernel");
        }
    }
    if (!p->uniform) {
        // (largest first: the workgroups take the tiles of a launch through a queue)
        std::stable_sort(w64_ids.begin(), w64_ids.end(), [&](int32_t a, int32_t b) {
            return (int64_t)L->rows[a] * L->rows[a] * L->cols[a] > (int64_t)L->rows[b] * L->rows[b] * L->cols[b];
        });
        std::vector<int32_t> rows(L->rows, L->rows + B), cols(L->cols, L->cols + B);
        qrk_status st;
        if ((st = upload(h, rows, &p->d_rows)) || (st = upload(h, cols, &p->d_cols)) ||
            (st = upload(h, coff, &p->d_coff)) || (st = upload(h, rowoff, &p->d_rowoff)) ||
            (st = upload(h, toff, &p->d_toff)) || (st = upload(h, qoff, &p->d_qoff)) ||
            (st = upload(h, roff, &p->d_roff)) || (st = upload(h, wave_ids, &p->d_wave_ids)) ||
            (st = upload(h, wg_ids, &p->d_wg_ids)) || (st = upload(h, col_ids, &p->d_col_ids)) ||
            (st = upload(h, w64_ids, &p->d_w64_ids))) {
            qrk_bd_plan_destroy(p);
            return st;
        }
        p->n_wave = (int64_t)wave_ids.size();
        p->n_w64 = (int64_t)w64_ids.size();
        p->n_wg = (int64_t)wg_ids.size();
        p->n_col = (int64_t)col_ids.size();
    }
    *out = p;
    return QRK_STATUS_OK;
}

qrk_status qrk_bd_plan_destroy(qrk_bd_plan p)
{
    if (!p) return QRK_STATUS_OK;
    (void)hipFree(p->d_rows); (void)hipFree(p->d_cols); (void)hipFree(p->d_coff); (void)hipFree(p->d_rowoff);
    (void)hipFree(p->d_toff); (void)hipFree(p->d_qoff); (void)hipFree(p->d_roff); (void)hipFree(p->d_wave_ids); (void)hipFree(p->d_w64_ids);
    (void)hipFree(p->d_wg_ids); (void)hipFree(p->d_workspace);
    (void)hipFree(p->d_col_ids); (void)hipFree(p->d_col_workspace); (void)hipFree(p->d_redo); (void)hipFree(p->d_exact_ws);
    (void)hipFree(p->d_p4_scratch);
    delete p;
    return QRK_STATUS_OK;
}

qrk_status qrk_bd_plan_sizes(qrk_bd_plan p, int64_t* tiles_len, int64_t* nnz_q, int64_t* nnz_r)
{
    if (!p) return QRK_STATUS_INVALID_ARGUMENT;
    if (tiles_len) *tiles_len = p->tiles_len;
    if (nnz_q) *nnz_q = p->nnz_q;
    if (nnz_r) *nnz_r = p->nnz_r;
    return QRK_STATUS_OK;
}

qrk_status qrk_bd_pattern(qrk_bd_plan p, int32_t* q_rowptr, int32_t* q_colidx, int32_t* r_colptr,
                          int32_t* r_rowidx, qrk_memspace space)
{
    if (!p || !q_rowptr || !q_colidx || !r_colptr || !r_rowidx)
        return fail(p ? p->h : nullptr, QRK_STATUS_INVALID_ARGUMENT, "qrk_bd_pattern: NULL argument");
    qrk_handle h = p->h;
    if (p->nnz_q > INT32_MAX || p->nnz_r > INT32_MAX)
        return fail(h, QRK_STATUS_UNSUPPORTED, "qrk_bd_pattern: nnz exceeds the int32 StorageIndex of the reference");
    QRK_HIP(h, hipSetDevice(h->device));
    const qrk::TileGeom g = make_geom(p);
    if (space == QRK_MEM_DEVICE) {
        qrk::launch_bd_pattern(g, p->nnz_r, q_rowptr, q_colidx, r_colptr, r_rowidx, h->stream);
        QRK_HIP(h, hipGetLastError());
        return QRK_STATUS_OK;
    }
    Staging s(h);
    int32_t *d_qp, *d_qi, *d_rp, *d_ri;
    qrk_status st;
    if ((st = s.out(p->mat_rows + 1, &d_qp)) || (st = s.out(p->nnz_q, &d_qi)) ||
        (st = s.out(p->mat_cols + 1, &d_rp)) || (st = s.out(p->nnz_r, &d_ri)))
        return st;
    qrk::launch_bd_pattern(g, p->nnz_r, d_qp, d_qi, d_rp, d_ri, h->stream);
    QRK_HIP(h, hipGetLastError());
    if ((st = s.back(q_rowptr, d_qp, p->mat_rows + 1)) || (st = s.back(q_colidx, d_qi, p->nnz_q)) ||
        (st = s.back(r_colptr, d_rp, p->mat_cols + 1)) || (st = s.back(r_rowidx, d_ri, p->nnz_r)))
        return st;
    QRK_HIP(h, hipStreamSynchronize(h->stream));
    return QRK_STATUS_OK;
}

qrk_status qrk_bd_tiles_from_sparse(qrk_bd_plan p, int row_major, const int32_t* outer_ptr, const int32_t* inner_idx,
                                    const double* vals, int64_t nnz, double* tiles, qrk_memspace space)
{
    if (!p || !outer_ptr || !tiles || nnz < 0 || (nnz > 0 && (!inner_idx || !vals)))
        return fail(p ? p->h : nullptr, QRK_STATUS_INVALID_ARGUMENT, "qrk_bd_tiles_from_sparse: NULL argument");
    qrk_handle h = p->h;
    if (nnz > INT32_MAX)
        return fail(h, QRK_STATUS_UNSUPPORTED, "qrk_bd_tiles_from_sparse: nnz exceeds the int32 StorageIndex of the reference");
    QRK_HIP(h, hipSetDevice(h->device));
    const qrk::TileGeom g = make_geom(p);
    const int64_t* t_off = p->uniform ? nullptr : p->d_toff;
    if (space == QRK_MEM_DEVICE) {
        qrk::launch_bd_cut_tiles(g, t_off, row_major, outer_ptr, inner_idx, vals, (int32_t)nnz, tiles, h->stream);
        QRK_HIP(h, hipGetLastError());
        return QRK_STATUS_OK;
    }
    const int64_t n_outer = (row_major ? p->mat_rows : p->mat_cols) + 1;
    for (int64_t o = 0; o + 1 < n_outer; ++o)
        if (outer_ptr[o] < 0 || outer_ptr[o] > outer_ptr[o + 1] || outer_ptr[o + 1] > nnz)
            return fail(h, QRK_STATUS_INVALID_ARGUMENT, "qrk_bd_tiles_from_sparse: outer pointers are not a compressed layout of nnz entries");
    Staging s(h);
    int32_t *d_ptr, *d_idx;
    double *d_vals, *d_tiles;
    qrk_status st;
    if ((st = s.in(outer_ptr, n_outer, &d_ptr)) || (st = s.in(inner_idx, nnz, &d_idx)) || (st = s.in(vals, nnz, &d_vals)) ||
        (st = s.out(p->tiles_len, &d_tiles)))
        return st;
    qrk::launch_bd_cut_tiles(g, t_off, row_major, d_ptr, d_idx, d_vals, (int32_t)nnz, d_tiles, h->stream);
    QRK_HIP(h, hipGetLastError());
    if ((st = s.back(tiles, d_tiles, p->tiles_len))) return st;
    QRK_HIP(h, hipStreamSynchronize(h->stream));
    return QRK_STATUS_OK;
}

qrk_status qrk_bd_factorize(qrk_bd_plan p, const double* tiles, double* q_vals, double* r_vals,
                            int32_t* perm, double* hcoeffs, qrk_memspace space)
{
    if (!p) return QRK_STATUS_INVALID_ARGUMENT;
    qrk_handle h = p->h;
    p->factorized = false;
    if (p->landscape) {   // BlockDiagonalSparseQR.h:509-516: m_info = InvalidInput; return
        p->factorized = true;
        return QRK_STATUS_OK;
    }
    if (p->B > 0 && (!tiles || !q_vals || !r_vals || !perm))
        return fail(h, QRK_STATUS_INVALID_ARGUMENT, "qrk_bd_factorize: NULL buffer");
    QRK_HIP(h, hipSetDevice(h->device));
    qrk_status st;
    if (space == QRK_MEM_DEVICE) {
        if ((st = enqueue_factorize(p, tiles, q_vals, r_vals, perm, hcoeffs))) return st;
    } else {
        Staging s(h);
        double *d_t, *d_q, *d_r, *d_hc = nullptr;
        int32_t* d_p;
        if ((st = s.in(tiles, p->tiles_len, &d_t)) || (st = s.out(p->nnz_q, &d_q)) ||
            (st = s.out(p->nnz_r, &d_r)) || (st = s.out((int64_t)p->mat_cols, &d_p)))
            return st;
        if (hcoeffs && (st = s.out((int64_t)p->mat_cols, &d_hc))) return st;
        if ((st = enqueue_factorize(p, d_t, d_q, d_r, d_p, d_hc))) return st;
        if ((st = s.back(q_vals, d_q, p->nnz_q)) || (st = s.back(r_vals, d_r, p->nnz_r)) ||
            (st = s.back(perm, d_p, (int64_t)p->mat_cols)) || (st = s.back(hcoeffs, d_hc, (int64_t)p->mat_cols)))
            return st;
        QRK_HIP(h, hipStreamSynchronize(h->stream));
    }
    p->factorized = true;
    return QRK_STATUS_OK;
}

qrk_status qrk_bd_info(qrk_bd_plan p, qrk_info* info, int64_t* rank)
{
    if (!p) return QRK_STATUS_INVALID_ARGUMENT;
    if (!p->factorized) return fail(p->h, QRK_STATUS_NOT_FACTORIZED, "qrk_bd_info: factorize() has not been called");
    if (info) *info = p->landscape ? QRK_INFO_INVALID_INPUT : QRK_INFO_SUCCESS;
    if (rank) *rank = p->mat_cols;   // rank += blockSolver.cols(), BlockDiagonalSparseQR.h:440
    return QRK_STATUS_OK;
}

qrk_status qrk_bd_apply_qt(qrk_bd_plan p, const double* q_vals, const double* b, int64_t nrhs, double* y,
                           qrk_memspace space)
{
    if (!p || !q_vals || !b || !y || nrhs < 0)
        return fail(p ? p->h : nullptr, QRK_STATUS_INVALID_ARGUMENT, "qrk_bd_apply_qt: bad argument");
    qrk_handle h = p->h;
    QRK_HIP(h, hipSetDevice(h->device));
    const qrk::TileGeom g = make_geom(p);
    if (space == QRK_MEM_DEVICE) {
        qrk::launch_bd_apply_qt(g, q_vals, b, nrhs, y, h->stream);
        QRK_HIP(h, hipGetLastError());
        return QRK_STATUS_OK;
    }
    Staging s(h);
    double *d_q, *d_b, *d_y;
    qrk_status st;
    if ((st = s.in(q_vals, p->nnz_q, &d_q)) || (st = s.in(b, nrhs * p->mat_rows, &d_b)) ||
        (st = s.out(nrhs * p->mat_rows, &d_y)))
        return st;
    qrk::launch_bd_apply_qt(g, d_q, d_b, nrhs, d_y, h->stream);
    QRK_HIP(h, hipGetLastError());
    if ((st = s.back(y, d_y, nrhs * p->mat_rows))) return st;
    QRK_HIP(h, hipStreamSynchronize(h->stream));
    return QRK_STATUS_OK;
}

qrk_status qrk_bd_apply_q(qrk_bd_plan p, const double* q_vals, const double* b, int64_t nrhs, double* y,
                          qrk_memspace space)
{
    if (!p || !q_vals || !b || !y || nrhs < 0 || b == y)
        return fail(p ? p->h : nullptr, QRK_STATUS_INVALID_ARGUMENT, "qrk_bd_apply_q: bad argument");
    qrk_handle h = p->h;
    QRK_HIP(h, hipSetDevice(h->device));
    const qrk::TileGeom g = make_geom(p);
    if (space == QRK_MEM_DEVICE) {
        qrk::launch_bd_apply_q(g, p->max_dim, q_vals, b, nrhs, y, h->stream);
        QRK_HIP(h, hipGetLastError());
        return QRK_STATUS_OK;
    }
    Staging s(h);
    double *d_q, *d_b, *d_y;
    qrk_status st;
    if ((st = s.in(q_vals, p->nnz_q, &d_q)) || (st = s.in(b, nrhs * p->mat_rows, &d_b)) ||
        (st = s.out(nrhs * p->mat_rows, &d_y)))
        return st;
    qrk::launch_bd_apply_q(g, p->max_dim, d_q, d_b, nrhs, d_y, h->stream);
    QRK_HIP(h, hipGetLastError());
    if ((st = s.back(y, d_y, nrhs * p->mat_rows))) return st;
    QRK_HIP(h, hipStreamSynchronize(h->stream));
    return QRK_STATUS_OK;
}

qrk_status qrk_bd_solve(qrk_bd_plan p, const double* q_vals, const double* r_vals, const int32_t* perm,
                        const double* b, int64_t nrhs, double* x, qrk_memspace space)
{
    if (!p || !q_vals || !r_vals || !perm || !b || !x || nrhs < 0)
        return fail(p ? p->h : nullptr, QRK_STATUS_INVALID_ARGUMENT, "qrk_bd_solve: bad argument");
    qrk_handle h = p->h;
    if (p->q_format != QRK_FULL_Q)
        return fail(h, QRK_STATUS_UNSUPPORTED,
                    "qrk_bd_solve: R is upper triangular only in the FullQ format (BlockDiagonalSparseQR.h:134-154)");
    QRK_HIP(h, hipSetDevice(h->device));
    const qrk::TileGeom g = make_geom(p);
    if (space == QRK_MEM_DEVICE) {
        qrk::launch_bd_solve(g, p->max_cols, q_vals, r_vals, perm, b, nrhs, x, h->stream);
        QRK_HIP(h, hipGetLastError());
        return QRK_STATUS_OK;
    }
    Staging s(h);
    double *d_q, *d_r, *d_b, *d_x;
    int32_t* d_p;
    qrk_status st;
    if ((st = s.in(q_vals, p->nnz_q, &d_q)) || (st = s.in(r_vals, p->nnz_r, &d_r)) ||
        (st = s.in(perm, (int64_t)p->mat_cols, &d_p)) || (st = s.in(b, nrhs * p->mat_rows, &d_b)) ||
        (st = s.out(nrhs * p->mat_cols, &d_x)))
        return st;
    qrk::launch_bd_solve(g, p->max_cols, d_q, d_r, d_p, d_b, nrhs, d_x, h->stream);
    QRK_HIP(h, hipGetLastError());
    if ((st = s.back(x, d_x, nrhs * p->mat_cols))) return st;
    QRK_HIP(h, hipStreamSynchronize(h->stream));
    return QRK_STATUS_OK;
}

qrk_status qrk_bd_solve_r(qrk_bd_plan p, const double* r_vals, const double* y, int64_t nrhs, double* z,
                          qrk_memspace space)
{
    if (!p || !r_vals || !y || !z || nrhs < 0)
        return fail(p ? p->h : nullptr, QRK_STATUS_INVALID_ARGUMENT, "qrk_bd_solve_r: bad argument");
    qrk_handle h = p->h;
    if (p->q_format != QRK_FULL_Q)
        return fail(h, QRK_STATUS_UNSUPPORTED, "qrk_bd_solve_r: R is upper triangular only in the FullQ format");
    QRK_HIP(h, hipSetDevice(h->device));
    const qrk::TileGeom g = make_geom(p);
    if (space == QRK_MEM_DEVICE) {
        qrk::launch_bd_solve_r(g, p->max_cols, r_vals, y, nrhs, z, h->stream);
        QRK_HIP(h, hipGetLastError());
        return QRK_STATUS_OK;
    }
    Staging s(h);
    double *d_r, *d_y, *d_z;
    qrk_status st;
    if ((st = s.in(r_vals, p->nnz_r, &d_r)) || (st = s.in(y, nrhs * p->mat_cols, &d_y)) ||
        (st = s.out(nrhs * p->mat_cols, &d_z)))
        return st;
    qrk::launch_bd_solve_r(g, p->max_cols, d_r, d_y, nrhs, d_z, h->stream);
    QRK_HIP(h, hipGetLastError());
    if ((st = s.back(z, d_z, nrhs * p->mat_cols))) return st;
    QRK_HIP(h, hipStreamSynchronize(h->stream));
    return QRK_STATUS_OK;
}

qrk_status qrk_dense_plan_create(qrk_handle h, int32_t rows, int32_t cols, qrk_block_solver solver,
                                 qrk_dense_plan* out)
{
    if (!h || !out || rows <= 0 || cols <= 0)
        return fail(h, QRK_STATUS_INVALID_ARGUMENT, "qrk_dense_plan_create: bad argument");
    *out = nullptr;
    QRK_HIP(h, hipSetDevice(h->device));
    qrk_dense_plan_s* p = new (std::nothrow) qrk_dense_plan_s();
    if (!p) return fail(h, QRK_STATUS_ALLOC_FAILED, "qrk_dense_plan_create: out of host memory");
    p->h = h; p->rows = rows; p->cols = cols; p->solver = solver;
    // The single-workgroup kernel keeps a column in LDS and sweeps with one CU; from 256x256 elements on (and
    // always when it does not fit) the row-slab path over all CUs is used.  QRK_DENSE_PATH=single|tall overrides.
    const bool fits = qrk::dense_qr_smem_bytes(rows, cols) <= 150 * 1024;
    p->tall = !fits || (int64_t)rows * cols >= 65536;
    if (const char* e = std::getenv("QRK_DENSE_PATH")) {
        if (!std::strcmp(e, "tall") || !std::strcmp(e, "slabs") || !std::strcmp(e, "cols")) p->tall = true;   // "tall": the multi-workgroup
        // paths, whichever fits; "slabs" / "cols": the row-slab kernels (dense_qr_tall.hip) / the column-parallel kernel (dense_qr_cols.hip)
        else if (!std::strcmp(e, "single") && fits) p->tall = false;
    }
    if (p->tall) {
        const size_t bytes = qrk::dense_tall_workspace_bytes(rows, cols, h->num_cus, &p->G, &p->cpad, &p->rows_per);
        if ((size_t)2 * p->rows_per * sizeof(double) > 150 * 1024) {
            delete p;
            return fail(h, QRK_STATUS_UNSUPPORTED, "qrk_dense_plan_create: more than ~2.4 M rows are not supported");
        }
        if (hipMalloc(&p->d_ws, bytes) != hipSuccess) {
            delete p;
            return fail(h, QRK_STATUS_ALLOC_FAILED, "qrk_dense_plan_create: cannot allocate the workspace");
        }
        // (measured slower than the kernel sequence - 472 vs 363 ms at 40000 x 2000, 14.0 vs 12.9 ms at 5120 x 384: every
        // workgroup's agent-scope fences write back / invalidate the XCD's L2 at each barrier - so it is opt-in)
        if (const char* e = std::getenv("QRK_DENSE_PERSISTENT")) {
            if (e[0] == '1')
                p->persistent = (size_t)2 * p->rows_per * sizeof(double) <= 64 * 1024 &&
                                qrk::dense_tall_persistent_ok(p->G, p->rows_per, h->num_cus);
        }
    }
    // Two-stage form: pivoted, tall (rows >= 4 cols) and large enough that the level-2 sweeps over the whole matrix dominate.
    // QRK_DENSE_TWO_STAGE=0/1 overrides (1 needs rows >= cols, pivoting).
    const bool big_tall = p->tall && cols >= 128 && (int64_t)rows >= 4 * (int64_t)cols && (int64_t)rows * cols >= (int64_t)1 << 22;
    p->two_stage = big_tall && solver == QRK_COLPIV_HOUSEHOLDER;
    p->caqr_only = big_tall && solver == QRK_HOUSEHOLDER;
    if (const char* e = std::getenv("QRK_DENSE_TWO_STAGE")) {
        if (e[0] == '0') { p->two_stage = false; p->caqr_only = false; }
        else if (e[0] == '1' && rows >= cols) { p->two_stage = solver == QRK_COLPIV_HOUSEHOLDER; p->caqr_only = solver == QRK_HOUSEHOLDER; }
    }
    p->fmt_capable = p->two_stage ? 1 : (p->caqr_only ? 2 : 0);
    p->exact_wide = (int64_t)rows * cols >= (int64_t)1 << 18;
    if (const char* e = std::getenv("QRK_EXACT_WIDE")) p->exact_wide = e[0] == '1' || (e[0] != '0' && p->exact_wide);
    p->cols_direct = p->tall && !p->two_stage && !p->caqr_only && !p->persistent && qrk::dense_cols_supported(rows, cols);
    if (const char* e = std::getenv("QRK_DENSE_PATH")) { if (!std::strcmp(e, "slabs")) p->cols_direct = false; }
    if (p->cols_direct) {
        p->pers_cus = qrk::dense_pers_supported(rows, cols, h->num_cus) ? h->num_cus : 0;
        const size_t bytes2 = qrk::dense_cols_workspace_bytes(cols, &p->cpad2, p->pers_cus > 0);
        if (hipMalloc((void**)&p->d_q1, (size_t)rows * (size_t)cols * sizeof(double)) != hipSuccess || hipMalloc(&p->d_ws2, bytes2) != hipSuccess) {
            qrk_dense_plan_destroy(p);
            return fail(h, QRK_STATUS_ALLOC_FAILED, "qrk_dense_plan_create: cannot allocate the workspaces of the column-parallel kernel");
        }
    }
    if (p->caqr_only && hipMalloc((void**)&p->d_t, qrk::caqr_t_bytes(rows, cols)) != hipSuccess) {
        qrk_dense_plan_destroy(p);
        return fail(h, QRK_STATUS_ALLOC_FAILED, "qrk_dense_plan_create: cannot allocate the T factors of the communication-avoiding QR");
    }
    if (p->two_stage) {
        // second stage: the column-parallel kernel (one launch per reflector) when a column fits LDS, else the row-slab kernels
        const bool fits2 = qrk::dense_qr_smem_bytes(cols, cols) <= 150 * 1024;
        p->cols2 = qrk::dense_cols_supported(cols, cols) && (int64_t)cols * cols >= 65536;
        if (const char* e = std::getenv("QRK_DENSE_STAGE2")) p->cols2 = p->cols2 && std::strcmp(e, "slabs") != 0;
        p->tall2 = !p->cols2 && (!fits2 || (int64_t)cols * cols >= 65536);
        size_t bytes2 = 0;
        if (p->cols2) {
            p->pers_cus = qrk::dense_pers_supported(cols, cols, h->num_cus) ? h->num_cus : 0;
            bytes2 = qrk::dense_cols_workspace_bytes(cols, &p->cpad2, p->pers_cus > 0);
        }
        else if (p->tall2) bytes2 = qrk::dense_tall_workspace_bytes(cols, cols, h->num_cus, &p->G2, &p->cpad2, &p->rows_per2);
        if (hipMalloc((void**)&p->d_t, qrk::caqr_t_bytes(rows, cols)) != hipSuccess ||
            hipMalloc((void**)&p->d_r0, (size_t)cols * (size_t)cols * sizeof(double)) != hipSuccess ||
            hipMalloc((void**)&p->d_t1, qrk::dense_q_tfactors_doubles(cols) * sizeof(double)) != hipSuccess ||
            (p->cols2 && hipMalloc((void**)&p->d_q1, (size_t)cols * (size_t)cols * sizeof(double)) != hipSuccess) ||
            (bytes2 && hipMalloc(&p->d_ws2, bytes2) != hipSuccess)) {
            qrk_dense_plan_destroy(p);
            return fail(h, QRK_STATUS_ALLOC_FAILED, "qrk_dense_plan_create: cannot allocate the two-stage workspaces");
        }
        if (!p->cols2) p->d_q1 = p->d_r0;
    }
    if (p->two_stage || p->caqr_only) {
        const char* la = std::getenv("QRK_CAQR_LOOKAHEAD");
        // (highest priority: a panel workgroup needs the LDS of one apply workgroup and should get the next slot that frees up)
        // (the streams are the handle's: ensure_pool)
        if (!(la && la[0] == '0')) {
            if (ensure_pool(h) != QRK_STATUS_OK ||
                hipEventCreateWithFlags(&p->la_urgent, hipEventDisableTiming) != hipSuccess ||
                hipEventCreateWithFlags(&p->la_factored, hipEventDisableTiming) != hipSuccess) {
                qrk_dense_plan_destroy(p);
                return fail(h, QRK_STATUS_ALLOC_FAILED, "qrk_dense_plan_create: cannot create the look-ahead events");
            }
            p->la_stream = h->side[2];
        }
        const char* pe = std::getenv("QRK_CAQR_PIPE");
        if (p->la_stream && !(pe && pe[0] == '0')) {
            p->la_pipe.urgent = h->side[1];
            bool ok = hipEventCreateWithFlags(&p->la_pipe.ev_n2, hipEventDisableTiming) == hipSuccess &&
                      hipEventCreateWithFlags(&p->la_pipe.ev_u, hipEventDisableTiming) == hipSuccess;
            for (int l = 0; ok && l < qrk::CaqrPipe::MAXL; ++l) ok = hipEventCreateWithFlags(&p->la_pipe.ev_lvl[l], hipEventDisableTiming) == hipSuccess;
            if (!ok) {
                qrk_dense_plan_destroy(p);
                return fail(h, QRK_STATUS_ALLOC_FAILED, "qrk_dense_plan_create: cannot create the streams of the pipelined look-ahead");
            }
        }
    }
    // exact path: a copy of the input, the flag of the single-workgroup kernel, the exact kernel's workspace
    if (hipMalloc((void**)&p->d_copy, (size_t)rows * (size_t)cols * sizeof(double)) != hipSuccess ||
        hipMalloc((void**)&p->d_unclear, sizeof(int)) != hipSuccess ||
        hipHostMalloc((void**)&p->h_unclear, sizeof(int), hipHostMallocDefault) != hipSuccess ||
        hipMalloc((void**)&p->d_exact_ws, qrk::dense_exact_workspace_bytes(rows, cols)) != hipSuccess) {
        qrk_dense_plan_destroy(p);
        return fail(h, QRK_STATUS_ALLOC_FAILED, "qrk_dense_plan_create: cannot allocate the input copy of the exact path");
    }
    *out = p;
    return QRK_STATUS_OK;
}

qrk_status qrk_dense_plan_destroy(qrk_dense_plan p)
{
    if (p) {
        (void)hipFree(p->d_ws); (void)hipFree(p->d_copy); (void)hipFree(p->d_unclear); (void)hipFree(p->d_exact_ws);
        if (p->h_unclear) (void)hipHostFree(p->h_unclear);
        if (p->la_urgent) (void)hipEventDestroy(p->la_urgent);
        if (p->la_factored) (void)hipEventDestroy(p->la_factored);
        if (p->la_pipe.ev_n2) (void)hipEventDestroy(p->la_pipe.ev_n2);
        if (p->la_pipe.ev_u) (void)hipEventDestroy(p->la_pipe.ev_u);
        for (int l = 0; l < qrk::CaqrPipe::MAXL; ++l) if (p->la_pipe.ev_lvl[l]) (void)hipEventDestroy(p->la_pipe.ev_lvl[l]);
        (void)hipFree(p->d_t); if (p->d_q1 != p->d_r0) (void)hipFree(p->d_q1); (void)hipFree(p->d_r0); (void)hipFree(p->d_ws2); (void)hipFree(p->d_t1); (void)hipFree(p->d_xw);
    }
    delete p;
    return QRK_STATUS_OK;
}

qrk_status qrk_dense_plan_set_two_stage(qrk_dense_plan p, int enable)
{
    if (!p) return fail(nullptr, QRK_STATUS_INVALID_ARGUMENT, "qrk_dense_plan_set_two_stage: null plan");
    if (enable && p->fmt_capable == 0)
        return fail(p->h, QRK_STATUS_UNSUPPORTED, "qrk_dense_plan_set_two_stage: this plan was created without the two-stage workspaces");
    p->two_stage = enable != 0 && p->fmt_capable == 1;
    p->caqr_only = enable != 0 && p->fmt_capable == 2;
    if (!enable) { p->ts_active = false; p->ts_owner = nullptr; }
    return QRK_STATUS_OK;
}

int qrk_dense_plan_two_stage(qrk_dense_plan p) { return p && (p->two_stage || p->caqr_only) ? 1 : 0; }

qrk_status qrk_dense_factorize(qrk_dense_plan p, double* a, int64_t lda, double* hcoeffs, int32_t* perm,
                               qrk_memspace space)
{
    if (!p || !a || !hcoeffs || !perm || lda < p->rows)
        return fail(p ? p->h : nullptr, QRK_STATUS_INVALID_ARGUMENT, "qrk_dense_factorize: bad argument");
    qrk_handle h = p->h;
    QRK_HIP(h, hipSetDevice(h->device));
    const int piv = p->solver == QRK_COLPIV_HOUSEHOLDER ? 1 : 0;
    const int size = p->rows < p->cols ? p->rows : p->cols;
    // the exact path of a large block: input restored from the plan's copy, then Eigen's operation order column by column, the
    // columns spread over the chip (bdqr_exact.hip, launch_dense_exact_wide)
    auto exact_wide = [&](double* da, double* dhc, int32_t* dp, int unclear) -> qrk_status {
        if (std::getenv("QRK_DEBUG_UNCLEAR"))
            std::fprintf(stderr, "qrk_dense_factorize: %d x %d to the exact path, flags 0x%x (1 pivot margin, 2 recompute band, 4 reflector: 8 degenerate, 16 |x0| tiny, 32 pivot tiny)\n",
                         p->rows, p->cols, unclear);
        QRK_HIP(h, hipMemcpy2DAsync(da, (size_t)lda * sizeof(double), p->d_copy, (size_t)p->rows * sizeof(double),
                                    (size_t)p->rows * sizeof(double), (size_t)p->cols, hipMemcpyDeviceToDevice, h->stream));
        QRK_HIP(h, qrk::launch_dense_exact_wide(da, lda, p->rows, p->cols, piv, dhc, dp, p->d_exact_ws, h->stream));
        return QRK_STATUS_OK;
    };
    auto run = [&](double* da, double* dhc, int32_t* dp) -> qrk_status {
        // keep the input: the exact path restores it when a decision was not clear of rounding (a D2D copy, < 1 % of the work)
        QRK_HIP(h, hipMemcpy2DAsync(p->d_copy, (size_t)p->rows * sizeof(double), da, (size_t)lda * sizeof(double),
                                    (size_t)p->rows * sizeof(double), (size_t)p->cols, hipMemcpyDeviceToDevice, h->stream));
        const int* flag = nullptr;
        p->ts_active = false;
        p->ts_owner = nullptr;
        if (!h->force_exact && p->caqr_only) {
            // HouseholderQR of a tall matrix as communication-avoiding QR: R0 in the upper triangle of the first cols rows, the reflectors
            // of the tree and their T factors elsewhere / in the plan; nothing is decided by the data, so nothing goes to the exact path
            QRK_HIP(h, qrk::launch_caqr_factorize(da, lda, p->rows, p->cols, p->d_t, h->stream, p->la_stream, p->la_urgent, p->la_factored,
                                                  p->la_pipe.urgent ? &p->la_pipe : nullptr));
            QRK_HIP(h, hipMemsetAsync(dhc, 0, (size_t)size * sizeof(double), h->stream));        // (no Householder coefficients in this format)
            hipLaunchKernelGGL(qrk::identity_perm_kernel, dim3((unsigned)((p->cols + 255) / 256)), dim3(256), 0, h->stream, dp, p->cols);
            QRK_HIP(h, hipGetLastError());                                                       // (no host vector, no synchronisation)
            p->ts_active = true;
            p->ts_owner = static_cast<const void*>(a);
            return QRK_STATUS_OK;
        }
        if (!h->force_exact && p->two_stage) {
            // stage 1: A = Q0 R0 (no pivoting, MFMA trailing updates); stage 2: R0 P = Q1 R on the n x n triangle
            const int n = p->cols;
            const int piv2 = piv | qrk::decide::PIVOTING_SIGN_FREE;      // R is Eigen's up to the signs of its rows either way
            QRK_HIP(h, qrk::launch_caqr_factorize(da, lda, p->rows, n, p->d_t, h->stream, p->la_stream, p->la_urgent, p->la_factored,
                                                  p->la_pipe.urgent ? &p->la_pipe : nullptr));
            QRK_HIP(h, qrk::launch_caqr_copy_upper(da, lda, p->d_r0, n, n, 1, h->stream));
            if (p->cols2) {
                QRK_HIP(h, qrk::launch_dense_qr_cols(p->d_r0, n, n, n, piv2, dhc, dp, p->d_ws2, p->cpad2, p->d_q1, n, p->pers_cus, h->stream));
                flag = qrk::dense_cols_unclear_ptr(p->d_ws2, p->cpad2);
            } else if (p->tall2) {
                QRK_HIP(h, qrk::launch_dense_qr_tall(p->d_r0, n, n, n, piv2, dhc, dp, p->d_ws2, p->G2, p->cpad2, p->rows_per2, false, h->stream));
                flag = qrk::dense_tall_unclear_ptr(p->d_ws2, p->G2, p->cpad2);
            } else {
                QRK_HIP(h, hipMemsetAsync(p->d_unclear, 0, sizeof(int), h->stream));
                QRK_HIP(h, qrk::launch_dense_qr(p->d_r0, n, n, n, piv2, dhc, dp, p->d_unclear, h->stream));
                flag = p->d_unclear;
            }
            // R replaces R0 in the caller's array (consumers read R from its upper triangle, as in Eigen's packed format)
            QRK_HIP(h, qrk::launch_caqr_copy_upper(p->d_q1, n, da, lda, n, 0, h->stream));
            // the T factors of Q1 by blocks of 32 reflectors: what qrk_dense_apply_q applies it with (three matrix-vector products per
            // block instead of 32 dependent reflectors)
            if (p->d_t1) QRK_HIP(h, qrk::launch_dense_q_tfactors(p->d_q1, n, n, n, dhc, p->d_t1, h->stream));
            // a decision of the second stage inside rounding: the exact path redoes the whole matrix in Eigen's operation order
            // and leaves Eigen's packed format; the host has to know which format the factors are in
            QRK_HIP(h, hipMemcpyAsync(p->h_unclear, flag, sizeof(int), hipMemcpyDeviceToHost, h->stream));
            QRK_HIP(h, hipStreamSynchronize(h->stream));
            const int unclear = *p->h_unclear;
            p->ts_active = unclear == 0;
            p->ts_owner = p->ts_active ? static_cast<const void*>(a) : nullptr;
            if (unclear) return exact_wide(da, dhc, dp, unclear);
            return QRK_STATUS_OK;
        }
        if (!h->force_exact) {
            if (p->cols_direct) {
                QRK_HIP(h, qrk::launch_dense_qr_cols(da, lda, p->rows, p->cols, piv, dhc, dp, p->d_ws2, p->cpad2, p->d_q1, p->rows, p->pers_cus, h->stream));
                QRK_HIP(h, hipMemcpy2DAsync(da, (size_t)lda * sizeof(double), p->d_q1, (size_t)p->rows * sizeof(double),
                                            (size_t)p->rows * sizeof(double), (size_t)p->cols, hipMemcpyDeviceToDevice, h->stream));
                flag = qrk::dense_cols_unclear_ptr(p->d_ws2, p->cpad2);
            } else if (p->tall) {
                QRK_HIP(h, qrk::launch_dense_qr_tall(da, lda, p->rows, p->cols, piv, dhc, dp, p->d_ws, p->G, p->cpad, p->rows_per,
                                                     p->persistent, h->stream));
                flag = qrk::dense_tall_unclear_ptr(p->d_ws, p->G, p->cpad);
            } else {
                QRK_HIP(h, hipMemsetAsync(p->d_unclear, 0, sizeof(int), h->stream));
                QRK_HIP(h, qrk::launch_dense_qr(da, lda, p->rows, p->cols, piv, dhc, dp, p->d_unclear, h->stream));
                flag = p->d_unclear;
            }
        }
        p->last_flag = flag; p->last_exact = flag ? -1 : 1;
        if (p->exact_wide) {
            // large blocks: one workgroup would take minutes, so the host reads the word and runs the exact path over the whole chip
            int unclear = 1;                               // (no flag: the exact path was asked for)
            if (flag) {
                QRK_HIP(h, hipMemcpyAsync(p->h_unclear, flag, sizeof(int), hipMemcpyDeviceToHost, h->stream));
                QRK_HIP(h, hipStreamSynchronize(h->stream));
                unclear = *p->h_unclear;
            }
            p->last_exact = unclear ? 1 : 0;
            if (unclear) return exact_wide(da, dhc, dp, unclear);
            return QRK_STATUS_OK;
        }
        QRK_HIP(h, qrk::launch_dense_exact(da, lda, p->rows, p->cols, piv, p->d_copy, dhc, dp, flag, p->d_exact_ws, h->stream));
        return QRK_STATUS_OK;
    };
    if (space == QRK_MEM_DEVICE) return run(a, hcoeffs, perm);
    Staging s(h);
    double *d_a, *d_hc;
    int32_t* d_p;
    qrk_status st;
    if ((st = s.in(a, lda * p->cols, &d_a)) || (st = s.out((int64_t)size, &d_hc)) || (st = s.out((int64_t)p->cols, &d_p)))
        return st;
    if ((st = run(d_a, d_hc, d_p))) return st;
    if ((st = s.back(a, d_a, lda * p->cols)) || (st = s.back(hcoeffs, d_hc, (int64_t)size)) ||
        (st = s.back(perm, d_p, (int64_t)p->cols)))
        return st;
    QRK_HIP(h, hipStreamSynchronize(h->stream));
    return QRK_STATUS_OK;
}

qrk_status qrk_dense_apply_q(qrk_dense_plan p, const double* qr, int64_t lda, const double* hcoeffs, int transpose,
                             double* b, int64_t ldb, int64_t nrhs, qrk_memspace space)
{
    if (!p || !qr || !hcoeffs || !b || nrhs < 0 || lda < p->rows || ldb < p->rows)
        return fail(p ? p->h : nullptr, QRK_STATUS_INVALID_ARGUMENT, "qrk_dense_apply_q: bad argument");
    qrk_handle h = p->h;
    QRK_HIP(h, hipSetDevice(h->device));
    const int size = p->rows < p->cols ? p->rows : p->cols;
    // the two-stage factors kept in the plan (T of Q0, packed Q1) are those of the LAST factorisation only: applying them to another
    // array would silently compute with the wrong Q
    if (p->ts_active && qr != p->ts_owner)
        return fail(h, QRK_STATUS_INVALID_ARGUMENT,
                    "qrk_dense_apply_q: this plan last factorised a different array in the two-stage format, whose Q lives in the plan; "
                    "factorise with one plan per matrix, or switch the format off with qrk_dense_plan_set_two_stage(plan, 0)");
    // Q = Q0 diag(Q1, I) after a two-stage factorisation (the factors of Q1 and the T factors of Q0 live in the plan)
    auto apply = [&](const double* dqr, const double* dhc, double* db) -> hipError_t {
        auto eigen_form = [&](const double* packed, int64_t ld, int rows, int nrefl) -> hipError_t {
            // Q1^T of a two-stage factorisation: block by block of 32 reflectors with the T factors the factorisation left in the plan, a
            // launch per block over several workgroups (the reflector-by-reflector kernel pulls all of V through one CU: 1.8 ms of
            // configs[3]'s solve()); QRK_DENSE_APPLY_BLOCKS=0 keeps that kernel.  (Q1 v, the other direction, stays on it.)
            const char* sw = std::getenv("QRK_DENSE_APPLY_BLOCKS");
            if (transpose && packed == p->d_q1 && p->d_t1 && p->ts_active && rows >= 256 && rows <= 8192 && nrhs <= 65535 && !(sw && sw[0] == '0')) {      // (at most 32 slabs of 256 rows)
                const int64_t need = 2 * 32 * 32 * nrhs;  // the slabs' shares of a block's w (at most 32 slabs), two sets
                if (need > p->xw_cap) {
                    (void)hipStreamSynchronize(h->stream);
                    (void)hipFree(p->d_xw); p->d_xw = nullptr; p->xw_cap = 0;
                    if (hipMalloc((void**)&p->d_xw, (size_t)need * sizeof(double)) == hipSuccess) p->xw_cap = need;
                    else (void)hipGetLastError();
                }
                if (p->d_xw) return qrk::launch_dense_apply_qt_blocks(packed, ld, rows, nrefl, p->d_t1, db, ldb, nrhs, p->d_xw, h->stream);
            }
            if ((size_t)(rows + 4) * sizeof(double) > 150 * 1024)
                return qrk::launch_dense_apply_q_tall(packed, ld, rows, nrefl, dhc, transpose, db, ldb, nrhs, h->stream);
            return qrk::launch_dense_apply_q(packed, ld, rows, nrefl, dhc, transpose, db, ldb, nrhs, h->stream);
        };
        if (!p->ts_active) return eigen_form(dqr, lda, p->rows, size);
        hipError_t e = hipSuccess;
        const bool q1 = !p->caqr_only;        // (two-stage: Q = Q0 diag(Q1, I); CAQR alone: Q = Q0)
        if (transpose) {
            e = qrk::launch_caqr_apply(dqr, lda, p->rows, p->cols, p->d_t, 1, db, ldb, nrhs, h->stream);
            if (e == hipSuccess && q1) e = eigen_form(p->d_q1, p->cols, p->cols, p->cols);
        } else {
            if (q1) e = eigen_form(p->d_q1, p->cols, p->cols, p->cols);
            if (e == hipSuccess) e = qrk::launch_caqr_apply(dqr, lda, p->rows, p->cols, p->d_t, 0, db, ldb, nrhs, h->stream);
        }
        return e;
    };
    if (space == QRK_MEM_DEVICE) {
        QRK_HIP(h, apply(qr, hcoeffs, b));
        return QRK_STATUS_OK;
    }
    Staging s(h);
    double *d_qr, *d_hc, *d_b;
    qrk_status st;
    if ((st = s.in(qr, lda * p->cols, &d_qr)) || (st = s.in(hcoeffs, (int64_t)size, &d_hc)) ||
        (st = s.in((const double*)b, ldb * nrhs, &d_b)))
        return st;
    QRK_HIP(h, apply(d_qr, d_hc, d_b));
    if ((st = s.back(b, d_b, ldb * nrhs))) return st;
    QRK_HIP(h, hipStreamSynchronize(h->stream));
    return QRK_STATUS_OK;
}

// (null when the allocation fails: the one-workgroup kernel serves then)
static int* solve_flags(qrk_handle h)
{
    if (!h->d_solve_flags && hipMalloc((void**)&h->d_solve_flags, qrk_context_s::SOLVE_FLAGS * sizeof(int)) != hipSuccess) h->d_solve_flags = nullptr;
    return h->d_solve_flags;
}

qrk_status qrk_dense_solve_r(qrk_dense_plan p, const double* qr, int64_t lda, double* b, int64_t ldb, int64_t nrhs,
                             qrk_memspace space)
{
    if (!p || !qr || !b || nrhs < 0 || lda < p->rows || ldb < p->cols || p->rows < p->cols)
        return fail(p ? p->h : nullptr, QRK_STATUS_INVALID_ARGUMENT, "qrk_dense_solve_r: bad argument");
    qrk_handle h = p->h;
    QRK_HIP(h, hipSetDevice(h->device));
    if (space == QRK_MEM_DEVICE) {
        QRK_HIP(h, qrk::launch_dense_solve_r(qr, lda, p->cols, b, ldb, nrhs, h->stream, solve_flags(h), qrk_context_s::SOLVE_FLAGS));
        return QRK_STATUS_OK;
    }
    Staging s(h);
    double *d_qr, *d_b;
    qrk_status st;
    if ((st = s.in(qr, lda * p->cols, &d_qr)) || (st = s.in((const double*)b, ldb * nrhs, &d_b))) return st;
    QRK_HIP(h, qrk::launch_dense_solve_r(d_qr, lda, p->cols, d_b, ldb, nrhs, h->stream, solve_flags(h), qrk_context_s::SOLVE_FLAGS));
    if ((st = s.back(b, d_b, ldb * nrhs))) return st;
    QRK_HIP(h, hipStreamSynchronize(h->stream));
    return QRK_STATUS_OK;
}

struct qrk_tsqr_plan_s {
    qrk_handle h = nullptr;
    int32_t rows = 0, cols = 0;
    double* d_t = nullptr;
};

qrk_status qrk_tsqr_plan_create(qrk_handle h, int32_t rows, int32_t cols, qrk_tsqr_plan* out)
{
    if (!h || !out || cols <= 0 || rows < cols) return fail(h, QRK_STATUS_INVALID_ARGUMENT, "qrk_tsqr_plan_create: bad argument (rows >= cols > 0)");
    *out = nullptr;
    QRK_HIP(h, hipSetDevice(h->device));
    qrk_tsqr_plan_s* p = new (std::nothrow) qrk_tsqr_plan_s();
    if (!p) return fail(h, QRK_STATUS_ALLOC_FAILED, "qrk_tsqr_plan_create: out of host memory");
    p->h = h; p->rows = rows; p->cols = cols;
    if (hipMalloc((void**)&p->d_t, qrk::caqr_t_bytes(rows, cols)) != hipSuccess) {
        delete p;
        return fail(h, QRK_STATUS_ALLOC_FAILED, "qrk_tsqr_plan_create: cannot allocate the T factors");
    }
    *out = p;
    return QRK_STATUS_OK;
}

qrk_status qrk_tsqr_plan_destroy(qrk_tsqr_plan p)
{
    if (p) (void)hipFree(p->d_t);
    delete p;
    return QRK_STATUS_OK;
}

qrk_status qrk_tsqr_factorize(qrk_tsqr_plan p, double* a, int64_t lda, qrk_memspace space)
{
    if (!p || !a || lda < p->rows) return fail(p ? p->h : nullptr, QRK_STATUS_INVALID_ARGUMENT, "qrk_tsqr_factorize: bad argument");
    if (space != QRK_MEM_DEVICE) return fail(p->h, QRK_STATUS_UNSUPPORTED, "qrk_tsqr_factorize: device memory only");
    qrk_handle h = p->h;
    QRK_HIP(h, hipSetDevice(h->device));
    QRK_HIP(h, qrk::launch_caqr_factorize(a, lda, p->rows, p->cols, p->d_t, h->stream));
    return QRK_STATUS_OK;
}

qrk_status qrk_tsqr_apply_q(qrk_tsqr_plan p, const double* a, int64_t lda, int transpose, double* b, int64_t ldb, int64_t nrhs,
                            qrk_memspace space)
{
    if (!p || !a || !b || nrhs < 0 || lda < p->rows || ldb < p->rows)
        return fail(p ? p->h : nullptr, QRK_STATUS_INVALID_ARGUMENT, "qrk_tsqr_apply_q: bad argument");
    if (space != QRK_MEM_DEVICE) return fail(p->h, QRK_STATUS_UNSUPPORTED, "qrk_tsqr_apply_q: device memory only");
    qrk_handle h = p->h;
    QRK_HIP(h, hipSetDevice(h->device));
    QRK_HIP(h, qrk::launch_caqr_apply(a, lda, p->rows, p->cols, p->d_t, transpose, b, ldb, nrhs, h->stream));
    return QRK_STATUS_OK;
}

static qrk_status bb_plan_create_impl(qrk_handle h, int32_t rows, int32_t cols, const int32_t* csr_rowptr,
                                      const int32_t* csr_colidx, int32_t suggested_block_cols,
                                      const qrk::FixedBandedPattern* fixed, qrk_bb_plan* out)
{
    if (!h || !out || !csr_rowptr || !csr_colidx || rows <= 0 || cols <= 0 || suggested_block_cols <= 0)
        return fail(h, QRK_STATUS_INVALID_ARGUMENT, "qrk_bb_plan_create: bad argument");
    *out = nullptr;
    QRK_HIP(h, hipSetDevice(h->device));
    qrk_bb_plan_s* p = new (std::nothrow) qrk_bb_plan_s();
    if (!p) return fail(h, QRK_STATUS_ALLOC_FAILED, "qrk_bb_plan_create: out of host memory");
    p->h = h;
    std::string err;
    if (!qrk::analyze_banded(rows, cols, csr_rowptr, csr_colidx, suggested_block_cols, p->st, err, fixed)) {
        delete p;
        return fail(h, QRK_STATUS_INVALID_ARGUMENT, "qrk_bb_plan_create: " + err);
    }
    const qrk::BandedStructure& st = p->st;
    if (qrk::bb_chain_smem(st.max_act_rows, st.max_ncols) > 150 * 1024) {
        delete p;
        return fail(h, QRK_STATUS_UNSUPPORTED, "qrk_bb_plan_create: panel too large for the single-workgroup chain kernel");
    }
    qrk_status s;
    if ((s = upload(h, st.panels, &p->d_panels)) || (s = upload(h, st.prowptr, &p->d_prowptr)) ||
        (s = upload(h, st.pcol, &p->d_pcol)) || (s = upload(h, st.pmap, &p->d_pmap)) ||
        (s = upload(h, st.r_src, &p->d_rsrc)) || (s = upload(h, st.r_colptr, &p->d_rcolptr)) ||
        (s = upload(h, st.r_rowidx, &p->d_rrowidx))) {
        qrk_bb_plan_destroy(p);
        return s;
    }
    const size_t wlen = (size_t)st.max_act_rows * (size_t)st.max_ncols;
    if (hipMalloc((void**)&p->d_W, wlen * sizeof(double)) != hipSuccess ||
        hipMalloc((void**)&p->d_lo, wlen * sizeof(double)) != hipSuccess ||
        hipMalloc((void**)&p->d_stage, (size_t)(st.stage_len > 0 ? st.stage_len : 1) * sizeof(double)) != hipSuccess) {
        qrk_bb_plan_destroy(p);
        return fail(h, QRK_STATUS_ALLOC_FAILED, "qrk_bb_plan_create: cannot allocate workspaces");
    }
    *out = p;
    return QRK_STATUS_OK;
}

qrk_status qrk_bb_plan_create(qrk_handle h, int32_t rows, int32_t cols, const int32_t* csr_rowptr,
                              const int32_t* csr_colidx, int32_t suggested_block_cols, qrk_bb_plan* out)
{
    return bb_plan_create_impl(h, rows, cols, csr_rowptr, csr_colidx, suggested_block_cols, nullptr, out);
}

qrk_status qrk_bb_plan_create_fixed(qrk_handle h, int32_t rows, int32_t cols, const int32_t* csr_rowptr,
                                    const int32_t* csr_colidx, int32_t block_rows, int32_t block_cols, int32_t block_overlap,
                                    int32_t suggested_block_cols, qrk_bb_plan* out)
{
    const qrk::FixedBandedPattern fx{block_rows, block_cols, block_overlap};
    return bb_plan_create_impl(h, rows, cols, csr_rowptr, csr_colidx, suggested_block_cols, &fx, out);
}

qrk_status qrk_bb_blocks_from_pattern(int32_t rows, int32_t cols, int32_t block_rows, int32_t block_cols, int32_t block_overlap,
                                      int32_t suggested_block_cols, int32_t cap, int32_t* num_blocks, int32_t* blocks)
{
    if (!num_blocks || rows <= 0 || cols <= 0 || suggested_block_cols <= 0)
        return fail(nullptr, QRK_STATUS_INVALID_ARGUMENT, "qrk_bb_blocks_from_pattern: bad argument");
    // the block map alone (no values, no chain): an empty pattern of the right shape is enough for analyze_banded's front half,
    // but the chain descriptors need real rows; so the map is rebuilt here from the same two routines
    std::vector<int32_t> rp((size_t)rows + 1, 0), ci;
    qrk::BandedStructure st;
    std::string err;
    const qrk::FixedBandedPattern fx{block_rows, block_cols, block_overlap};
    if (!qrk::banded_block_map_fixed(rows, cols, fx, suggested_block_cols, st.blocks, err))
        return fail(nullptr, QRK_STATUS_INVALID_ARGUMENT, "qrk_bb_blocks_from_pattern: " + err);
    *num_blocks = (int32_t)st.blocks.size();
    if (blocks)
        for (size_t i = 0; i < st.blocks.size() && (int32_t)i < cap; ++i) {
            blocks[4 * i] = st.blocks[i].idxRow; blocks[4 * i + 1] = st.blocks[i].idxCol;
            blocks[4 * i + 2] = st.blocks[i].numRows; blocks[4 * i + 3] = st.blocks[i].numCols;
        }
    return QRK_STATUS_OK;
}

qrk_status qrk_bb_analyze_host(int32_t rows, int32_t cols, const int32_t* csr_rowptr, const int32_t* csr_colidx,
                               int32_t suggested_block_cols, int32_t cap, int32_t* num_blocks, int32_t* blocks,
                               int32_t* row_perm, int32_t* has_row_perm)
{
    if (!csr_rowptr || !csr_colidx || !num_blocks || rows <= 0 || cols <= 0 || suggested_block_cols <= 0)
        return fail(nullptr, QRK_STATUS_INVALID_ARGUMENT, "qrk_bb_analyze_host: bad argument");
    qrk::BandedStructure st;
    std::string err;
    if (!qrk::analyze_banded(rows, cols, csr_rowptr, csr_colidx, suggested_block_cols, st, err))
        return fail(nullptr, QRK_STATUS_INVALID_ARGUMENT, "qrk_bb_analyze_host: " + err);
    *num_blocks = (int32_t)st.blocks.size();
    if (blocks)
        for (size_t i = 0; i < st.blocks.size() && (int32_t)i < cap; ++i) {
            blocks[4 * i] = st.blocks[i].idxRow; blocks[4 * i + 1] = st.blocks[i].idxCol;
            blocks[4 * i + 2] = st.blocks[i].numRows; blocks[4 * i + 3] = st.blocks[i].numCols;
        }
    if (row_perm) std::memcpy(row_perm, st.row_perm.data(), st.row_perm.size() * sizeof(int32_t));
    if (has_row_perm) *has_row_perm = st.has_row_perm ? 1 : 0;
    return QRK_STATUS_OK;
}

qrk_status qrk_bb_plan_destroy(qrk_bb_plan p)
{
    if (!p) return QRK_STATUS_OK;
    (void)hipFree(p->d_panels); (void)hipFree(p->d_prowptr); (void)hipFree(p->d_pcol); (void)hipFree(p->d_pmap);
    (void)hipFree(p->d_rsrc); (void)hipFree(p->d_rcolptr); (void)hipFree(p->d_rrowidx);
    (void)hipFree(p->d_W); (void)hipFree(p->d_lo); (void)hipFree(p->d_stage);
    delete p;
    return QRK_STATUS_OK;
}

/* ---- BlockedThinSparseQR (src/QRKit/BlockedThinSparseQR.h:105-283) ---------------------------------------------------------- */

qrk_status qrk_thin_destroy(qrk_thin_plan p)
{
    if (!p) return QRK_STATUS_OK;
    for (double* c : p->arena) (void)hipFree(c);
    for (auto& b : p->dplans) if (b.plan) (void)qrk_dense_plan_destroy(b.plan);
    (void)hipFree(p->d_R);
    delete p;
    return QRK_STATUS_OK;
}

qrk_status qrk_thin_sparse_factorize(qrk_handle h, int32_t rows, int32_t cols, int32_t block_cols, const int32_t* colptr,
                                     const int32_t* rowidx, const double* vals, qrk_thin_plan* out)
{
    if (!h || !out || rows <= 0 || cols <= 0 || block_cols <= 0 || !colptr || rows < cols)
        return fail(h, QRK_STATUS_INVALID_ARGUMENT, "qrk_thin_sparse_factorize: bad argument (a thin matrix, rows >= cols, in CSC)");
    *out = nullptr;
    const int64_t nnz = colptr[cols];
    if (nnz < 0 || (nnz > 0 && (!rowidx || !vals))) return fail(h, QRK_STATUS_INVALID_ARGUMENT, "qrk_thin_sparse_factorize: bad CSC arrays");
    for (int32_t j = 0; j < cols; ++j) {
        if (colptr[j + 1] < colptr[j]) return fail(h, QRK_STATUS_INVALID_ARGUMENT, "qrk_thin_sparse_factorize: column pointers decrease");
        for (int32_t e = colptr[j]; e < colptr[j + 1]; ++e)
            if (rowidx[e] < 0 || rowidx[e] >= rows || (e > colptr[j] && rowidx[e] <= rowidx[e - 1]))
                return fail(h, QRK_STATUS_INVALID_ARGUMENT, "qrk_thin_sparse_factorize: row indices out of range or not strictly increasing in a column");
    }
    QRK_HIP(h, hipSetDevice(h->device));
    qrk_thin_plan_s* p = new (std::nothrow) qrk_thin_plan_s();
    if (!p) return fail(h, QRK_STATUS_ALLOC_FAILED, "qrk_thin_sparse_factorize: out of host memory");
    p->h = h; p->rows = rows; p->cols = cols; p->block_cols = block_cols;

    // ---- analyzePattern (:168-201).  ColumnDensity (SparseQROrdering.h:21-50): columns stable-sorted by their number of nonzeros;
    // cperm[original column] = sorted rank, and (A P)(:, j) = A(:, cperm[j]) as the reference applies it
    std::vector<int32_t> order((size_t)cols), cperm((size_t)cols);
    for (int32_t j = 0; j < cols; ++j) order[(size_t)j] = j;
    std::stable_sort(order.begin(), order.end(), [&](int32_t a, int32_t b) { return colptr[a + 1] - colptr[a] < colptr[b + 1] - colptr[b]; });
    for (int32_t k = 0; k < cols; ++k) cperm[(size_t)order[(size_t)k]] = k;
    // permuted CSC, then AsBandedAsPossible (SparseQROrdering.h:52-120): rows stable-sorted by the column of their first nonzero
    std::vector<int32_t> pcp((size_t)cols + 1, 0), pri((size_t)nnz);
    std::vector<double> pv((size_t)nnz);
    for (int32_t j = 0; j < cols; ++j) pcp[(size_t)j + 1] = pcp[(size_t)j] + (colptr[cperm[(size_t)j] + 1] - colptr[cperm[(size_t)j]]);
    std::vector<int32_t> start((size_t)rows, cols);
    for (int32_t j = 0; j < cols; ++j) {
        const int32_t src = cperm[(size_t)j];
        int32_t w = pcp[(size_t)j];
        for (int32_t e = colptr[src]; e < colptr[src + 1]; ++e, ++w) {
            pri[(size_t)w] = rowidx[e]; pv[(size_t)w] = vals[e];
            if (start[(size_t)rowidx[e]] == cols) start[(size_t)rowidx[e]] = j;      // (columns ascend: the first hit is the smallest)
        }
    }
    bool has = false;
    for (int32_t r = 1; r < rows && !has; ++r) has = start[(size_t)r] < start[(size_t)r - 1];
    std::vector<int32_t> rorder((size_t)rows), rperm((size_t)rows);
    for (int32_t r = 0; r < rows; ++r) rorder[(size_t)r] = r;
    if (has) std::stable_sort(rorder.begin(), rorder.end(), [&](int32_t a, int32_t b) { return start[(size_t)a] < start[(size_t)b]; });
    for (int32_t k = 0; k < rows; ++k) rperm[(size_t)rorder[(size_t)k]] = k;              // old row r -> new row rperm[r]
    // last (new) row of every permuted column: the height of a panel (updateBlockInfo, :203-238)
    std::vector<int32_t> lastrow((size_t)cols, 0);
    for (int32_t j = 0; j < cols; ++j)
        for (int32_t e = pcp[(size_t)j]; e < pcp[(size_t)j + 1]; ++e) lastrow[(size_t)j] = std::max(lastrow[(size_t)j], rperm[(size_t)pri[(size_t)e]]);

    // ---- the working matrix m_pmatDense on the device: the nonzeros cross PCIe; `rows` zero rows are appended so that a panel
    // can be handed to a dense plan of its bucket height wherever it starts (bucket <= rows)
    p->maxrows = rows;
    const int64_t ldd = (int64_t)rows + p->maxrows;
    p->ldd = ldd;
    int32_t *d_cp = nullptr, *d_ri = nullptr, *d_map = nullptr, *d_pp = nullptr;
    double *d_pv = nullptr, *d_D = nullptr, *d_pack = nullptr;
    auto cleanup = [&]() { (void)hipFree(d_cp); (void)hipFree(d_ri); (void)hipFree(d_map); (void)hipFree(d_pv); (void)hipFree(d_D); (void)hipFree(d_pp); (void)hipFree(d_pack); };
    qrk_status st;
    if ((st = upload(h, pcp, &d_cp)) || (st = upload(h, pri, &d_ri)) || (st = upload(h, pv, &d_pv)) || (st = upload(h, rperm, &d_map))) { cleanup(); qrk_thin_destroy(p); return st; }
    if (hipMalloc((void**)&d_D, (size_t)ldd * cols * sizeof(double)) != hipSuccess ||
        hipMalloc((void**)&p->d_R, (size_t)cols * cols * sizeof(double)) != hipSuccess ||
        hipMalloc((void**)&d_pp, (size_t)block_cols * sizeof(int32_t)) != hipSuccess ||
        hipMalloc((void**)&d_pack, ((size_t)1 + block_cols) * sizeof(double)) != hipSuccess) {
        cleanup(); qrk_thin_destroy(p);
        return fail(h, QRK_STATUS_ALLOC_FAILED, "qrk_thin_sparse_factorize: cannot allocate the dense working matrix");
    }
    auto bail = [&](qrk_status code) { cleanup(); qrk_thin_destroy(p); return code; };
#define QRK_THIN_HIP(expr) do { if ((expr) != hipSuccess) { (void)fail(h, QRK_STATUS_HIP_ERROR, std::string("qrk_thin_sparse_factorize: ") + hipGetErrorString(hipGetLastError())); return bail(QRK_STATUS_HIP_ERROR); } } while (0)
    QRK_THIN_HIP(hipMemsetAsync(d_D, 0, (size_t)ldd * cols * sizeof(double), h->stream));
    QRK_THIN_HIP(hipMemsetAsync(p->d_R, 0, (size_t)cols * cols * sizeof(double), h->stream));
    if (nnz > 0)
        QRK_THIN_HIP(qrk::launch_sparse_window_to_dense(false, rows, cols, d_cp, d_ri, d_pv, 0, rows, d_map, d_D, ldd, h->stream));

    // ---- compute (:105-165): panel by panel
    std::vector<int32_t> nnz_idx, zero_idx;
    std::vector<double> pack((size_t)1 + block_cols);
    int32_t nzp = 0, solved = 0, new_piv = 0, prev_rows = 0;
    while (solved < cols) {
        int32_t nnew = block_cols, nrows;
        if (solved + nnew >= cols) { nnew = cols - solved; nrows = rows - nzp; }
        else {
            int32_t biggest = 0;
            for (int32_t c = 0; c < nnew; ++c) if (pcp[(size_t)solved + c + 1] > pcp[(size_t)solved + c]) biggest = std::max(biggest, lastrow[(size_t)solved + c]);
            nrows = biggest - nzp + 1;
            if (nrows < prev_rows - new_piv) nrows = prev_rows - new_piv;
        }
        if (nrows < 0) nrows = 0;
        const int32_t r0 = nzp, c0 = solved, k = std::min(nrows, nnew);
        // the panel's bucket and its plan
        int32_t hb = 64;
        while (hb < nrows) hb *= 2;
        hb = std::min(hb, rows);
        hb = std::max(hb, nnew);                    // (a dense plan needs rows >= cols)
        int32_t pi = -1;
        for (size_t b = 0; b < p->dplans.size(); ++b) if (p->dplans[b].hb == hb && p->dplans[b].nnew == nnew) pi = (int32_t)b;
        if (pi < 0) {
            qrk_dense_plan np_ = nullptr;
            if ((st = qrk_dense_plan_create(h, hb, nnew, QRK_COLPIV_HOUSEHOLDER, &np_)) != QRK_STATUS_OK) return bail(st);
            (void)qrk_dense_plan_set_two_stage(np_, 0);     // one plan, many panels that are re-applied later: every Q in the panel's own arrays
            p->dplans.push_back({hb, nnew, np_});
            pi = (int32_t)p->dplans.size() - 1;
        }
        qrk_dense_plan dp = p->dplans[(size_t)pi].plan;
        // storage from the arena: hb x nnew for the packed QR + nnew for tau
        const size_t nji = ((size_t)hb * nnew + 31) / 32 * 32;                  // (256-byte granules: the panels keep hipMalloc's alignment)
        const size_t need = nji + ((size_t)std::max(nnew, 1) + 31) / 32 * 32;
        if (p->arena.empty() || p->arena_used + need > p->arena_cap) {
            const size_t chunk = std::max<size_t>(need, std::min<size_t>((size_t)rows * cols + (size_t)cols, (size_t)1 << 24));
            double* c = nullptr;
            if (hipMalloc((void**)&c, chunk * sizeof(double)) != hipSuccess) {
                (void)fail(h, QRK_STATUS_ALLOC_FAILED, "qrk_thin_sparse_factorize: cannot allocate panel storage");
                return bail(QRK_STATUS_ALLOC_FAILED);
            }
            p->arena.push_back(c); p->arena_used = 0; p->arena_cap = chunk;
        }
        qrk_thin_plan_s::Panel pn{r0, nrows, nnew, hb, pi, p->arena.back() + p->arena_used, p->arena.back() + p->arena_used + nji};
        p->arena_used += need;
        p->panels.push_back(pn);
        // Ji = copy of the block (the reference factorises a copy), zero rows below it
        if (hb > nrows) QRK_THIN_HIP(hipMemsetAsync(pn.d_ji, 0, (size_t)hb * nnew * sizeof(double), h->stream));
        if (nrows > 0)
            QRK_THIN_HIP(hipMemcpy2DAsync(pn.d_ji, (size_t)hb * sizeof(double), d_D + (int64_t)c0 * ldd + r0, (size_t)ldd * sizeof(double),
                                          (size_t)nrows * sizeof(double), (size_t)nnew, hipMemcpyDeviceToDevice, h->stream));
        if ((st = qrk_dense_factorize(dp, pn.d_ji, hb, pn.d_hc, d_pp, QRK_MEM_DEVICE)) != QRK_STATUS_OK) return bail(st);
        // pivots and permutation of the panel to the host: nonzeroPivots() decides the geometry of the next panel
        hipLaunchKernelGGL(qrk::thin_pack_kernel, dim3((unsigned)((nnew + 255) / 256)), dim3(256), 0, h->stream, d_pp, k, nnew, std::max(nrows, 1),
                           dp->last_flag, dp->last_exact, qrk::dense_exact_pivot_norms(dp->d_exact_ws, hb, nnew), d_pack);
        QRK_THIN_HIP(hipMemcpyAsync(pack.data(), d_pack, (size_t)(1 + nnew) * sizeof(double), hipMemcpyDeviceToHost, h->stream));
        QRK_THIN_HIP(hipStreamSynchronize(h->stream));
        const int32_t nz = (int32_t)pack[0];               // nonzeroPivots() of the panel (BlockedThinSparseQR.h:250-256), counted on the device
        if (k < nnew) {
            // A panel with fewer rows than columns is factorised here with zero rows appended, for nnew steps; Eigen's
            // ColPivHouseholderQR stops after k = min(rows, cols) steps and leaves the other columns where its k transpositions put
            // them.  The steps beyond k pivot among columns that are zero up to the rounding of the norm downdates, so the order of
            // the tail is restored from the first k choices: idx after the transpositions (q, position of the q-th chosen column).
            // (The reference itself is outside its domain for such a panel: it reads matrixQR()(br, bc) for br up to bc, :275-277.)
            std::vector<int32_t> idx((size_t)nnew), posof((size_t)nnew);
            for (int32_t c = 0; c < nnew; ++c) idx[(size_t)c] = posof[(size_t)c] = c;
            for (int32_t q = 0; q < k; ++q) {
                const int32_t c = (int32_t)pack[(size_t)(1 + q)], b = posof[(size_t)c], o = idx[(size_t)q];
                idx[(size_t)q] = c; idx[(size_t)b] = o; posof[(size_t)c] = q; posof[(size_t)o] = b;
            }
            for (int32_t c = k; c < nnew; ++c) pack[(size_t)(1 + c)] = (double)idx[(size_t)c];
        }
        for (int32_t c = 0; c < nz; ++c) nnz_idx.push_back(c0 + (int32_t)pack[(size_t)(1 + c)]);
        for (int32_t c = nz; c < nnew; ++c) zero_idx.push_back(c0 + (int32_t)pack[(size_t)(1 + c)]);
        // update of the columns to the right (the rows of the panel and the zero rows below them): Q_panel^T in reflector form
        const int32_t ntrail = cols - (c0 + nnew);
        if (ntrail > 0 && k > 0)
            if ((st = qrk_dense_apply_q(dp, pn.d_ji, hb, pn.d_hc, 1, d_D + (int64_t)(c0 + nnew) * ldd + r0, ldd, ntrail, QRK_MEM_DEVICE)) != QRK_STATUS_OK) return bail(st);
        hipLaunchKernelGGL(qrk::thin_r_columns_kernel, dim3((unsigned)nnew), dim3(256), 0, h->stream, d_D, ldd, pn.d_ji, (int64_t)hb, d_pp,
                           nzp, c0, k, nnew, p->d_R, (int64_t)cols);
        QRK_THIN_HIP(hipGetLastError());
        new_piv = nz; nzp += nz; prev_rows = nrows; solved += nnew;
    }
#undef QRK_THIN_HIP
    if (hipStreamSynchronize(h->stream) != hipSuccess) return bail(fail(h, QRK_STATUS_HIP_ERROR, "qrk_thin_sparse_factorize: device error"));
    p->rank = nzp;
    p->col_perm.resize((size_t)cols);
    size_t w = 0;
    for (int32_t c : nnz_idx) p->col_perm[w++] = cperm[(size_t)c];
    for (int32_t c : zero_idx) p->col_perm[w++] = cperm[(size_t)c];
    p->row_perm = rperm;
    cleanup();
    *out = p;
    return QRK_STATUS_OK;
}

qrk_status qrk_thin_info(qrk_thin_plan p, int32_t* rank, int32_t* col_perm, int32_t* row_perm)
{
    if (!p) return QRK_STATUS_INVALID_ARGUMENT;
    if (rank) *rank = p->rank;
    if (col_perm) std::copy(p->col_perm.begin(), p->col_perm.end(), col_perm);
    if (row_perm) std::copy(p->row_perm.begin(), p->row_perm.end(), row_perm);
    return QRK_STATUS_OK;
}

qrk_status qrk_thin_matrix_r(qrk_thin_plan p, double* r, int64_t ldr, qrk_memspace space)
{
    if (!p || !r || ldr < p->cols) return fail(p ? p->h : nullptr, QRK_STATUS_INVALID_ARGUMENT, "qrk_thin_matrix_r: bad argument");
    qrk_handle h = p->h;
    QRK_HIP(h, hipMemcpy2DAsync(r, (size_t)ldr * sizeof(double), p->d_R, (size_t)p->cols * sizeof(double), (size_t)p->cols * sizeof(double),
                                (size_t)p->cols, space == QRK_MEM_DEVICE ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost, h->stream));
    if (space != QRK_MEM_DEVICE) QRK_HIP(h, hipStreamSynchronize(h->stream));
    return QRK_STATUS_OK;
}

qrk_status qrk_thin_apply_q(qrk_thin_plan p, int transpose, double* v, int64_t ldv, int64_t nrhs)
{
    if (!p || !v || nrhs < 0 || ldv < (int64_t)p->rows + p->maxrows)
        return fail(p ? p->h : nullptr, QRK_STATUS_INVALID_ARGUMENT, "qrk_thin_apply_q: v needs a leading dimension of at least 2 rows (the panels are applied with zero rows appended)");
    // SparseBlockYTY sequence (SparseBlockYTY.h:111-138): Q^T v = the panels in order, Q v = in reverse; a panel acts on its row range
    const size_t np = p->panels.size();
    for (size_t s = 0; s < np; ++s) {
        const auto& q = p->panels[transpose ? s : np - 1 - s];
        if (std::min(q.nrows, q.nnew) == 0) continue;
        qrk_dense_plan dp = p->dplans[(size_t)q.plan].plan;
        qrk_status st = qrk_dense_apply_q(dp, q.d_ji, q.hb, q.d_hc, transpose ? 1 : 0, v + q.r0, ldv, nrhs, QRK_MEM_DEVICE);
        if (st != QRK_STATUS_OK) return st;
    }
    return QRK_STATUS_OK;
}

qrk_status qrk_thin_solve(qrk_thin_plan p, double* v, int64_t ldv, int64_t nrhs)
{
    // BlockedThinQRBase::_solve_impl (BlockedThinQRBase.h:223-247): y = Q^T b; x(0:rank) = R(0:rank,0:rank)^-1 y(0:rank), the rest zero;
    // in place in v (the first cols entries of every column on return)
    qrk_status st = qrk_thin_apply_q(p, 1, v, ldv, nrhs);
    if (st != QRK_STATUS_OK) return st;
    qrk_handle h = p->h;
    if (p->rank > 0) QRK_HIP(h, qrk::launch_dense_solve_r(p->d_R, p->cols, p->rank, v, ldv, nrhs, h->stream, solve_flags(h), qrk_context_s::SOLVE_FLAGS));
    if (p->rank < p->cols)
        QRK_HIP(h, hipMemset2DAsync(v + p->rank, (size_t)ldv * sizeof(double), 0, (size_t)(p->cols - p->rank) * sizeof(double), (size_t)nrhs, h->stream));
    return QRK_STATUS_OK;
}

/* ---- banded matrix as dense strips: two-stage factorisation (strips form, banded.hip) ------------------------------------ */

qrk_status qrk_bbs_plan_create(qrk_handle h, int64_t num_strips, int32_t strip_rows, int32_t strip_cols, int32_t col_step,
                               qrk_bbs_plan* out)
{
    if (!h || !out) return QRK_STATUS_INVALID_ARGUMENT;
    *out = nullptr;
    const int lo = strip_cols - col_step;
    if (num_strips < 1 || strip_cols < 16 || strip_cols > 256 || strip_cols % 16 || col_step < 16 || col_step % 16 || lo < 0 ||
        strip_rows < strip_cols || strip_rows > 256 || num_strips * (int64_t)strip_rows > INT32_MAX)
        return fail(h, QRK_STATUS_INVALID_ARGUMENT,
                    "qrk_bbs_plan_create: strips of rows >= cols, cols and col_step multiples of 16, cols <= 256, rows <= 256, "
                    "col_step <= cols");
    QRK_HIP(h, hipSetDevice(h->device));
    qrk_bbs_plan_s* p = new (std::nothrow) qrk_bbs_plan_s();
    if (!p) return fail(h, QRK_STATUS_ALLOC_FAILED, "qrk_bbs_plan_create: out of host memory");
    p->h = h; p->N = num_strips; p->ms = strip_rows; p->n = strip_cols; p->s = col_step; p->lo = lo;
    p->rows = num_strips * (int64_t)strip_rows;
    p->cols = (num_strips - 1) * (int64_t)col_step + strip_cols;
    const int n = p->n, s = p->s;
    // stage A: N tiles of ms x n, un-pivoted Householder QR, Q_i explicit and block diagonal (Q^T b of strip i = rows i ms .. of the product)
    qrk_bd_layout lay{};
    lay.num_blocks = num_strips; lay.block_rows = strip_rows; lay.block_cols = strip_cols;
    lay.mat_rows = (int32_t)p->rows; lay.mat_cols = (int32_t)(num_strips * (int64_t)strip_cols);
    if (num_strips * (int64_t)strip_cols > INT32_MAX) { delete p; return fail(h, QRK_STATUS_INVALID_ARGUMENT, "qrk_bbs_plan_create: too many columns"); }
    qrk_status st = qrk_bd_plan_create(h, &lay, QRK_BLOCK_DIAGONAL_Q, QRK_HOUSEHOLDER, &p->bd);
    if (st != QRK_STATUS_OK) { delete p; return st; }
    // stage B: one panel per strip.  Panel 0 is R_0 itself (n rows); panel i >= 1 stacks the carry (lo rows, interleaved) and R_i
    p->panels.resize((size_t)num_strips);
    for (int64_t i = 0; i < num_strips; ++i) {
        qrk::BBPanel& q = p->panels[(size_t)i];
        q = qrk::BBPanel{};
        q.row0 = 0; q.col0 = (int32_t)(i * s);
        q.act_rows = i == 0 ? n : lo + n;
        q.ncols = n;
        q.solved = i + 1 < num_strips ? s : n;
        q.lo_rows = i == 0 ? 0 : lo; q.lo_cols = q.lo_rows; q.lo_from = s; q.lo_stride = 2;
        q.yrow = q.col0; q.num_zeros = 0;
        q.y_off = p->y_len; q.t_off = p->t_len; q.r_off = p->stage_len;
        p->y_len += (int64_t)q.act_rows * n; p->t_len += (int64_t)n * n; p->stage_len += (int64_t)q.solved * n;
        if (q.act_rows > p->max_act) p->max_act = q.act_rows;
    }
    // staircase limits at 16-column granularity: panel 0 is upper triangular; in the interleaved stack row r starts in column r / 2
    // (r < 2 lo) or r - lo
    std::vector<int32_t> rlim((size_t)2 * (n / 16));
    for (int g = 0; g < n / 16; ++g) {
        const int last = 16 * g + 15;
        rlim[(size_t)g] = last + 1;
        rlim[(size_t)(n / 16 + g)] = last < lo ? 2 * last + 2 : lo + last + 1;
    }
    const int64_t ntri = (int64_t)n * (n + 1) / 2;
    if ((st = upload(h, p->panels, &p->d_panels)) || (st = upload(h, rlim, &p->d_rlim))) { qrk_bbs_plan_destroy(p); return st; }
    if (hipMalloc((void**)&p->d_q, (size_t)num_strips * strip_rows * strip_rows * sizeof(double)) != hipSuccess ||
        hipMalloc((void**)&p->d_ra, (size_t)num_strips * ntri * sizeof(double)) != hipSuccess ||
        hipMalloc((void**)&p->d_perm, (size_t)num_strips * n * sizeof(int32_t)) != hipSuccess ||
        hipMalloc((void**)&p->d_y, (size_t)p->y_len * sizeof(double)) != hipSuccess ||
        hipMalloc((void**)&p->d_t, (size_t)p->t_len * sizeof(double)) != hipSuccess ||
        hipMalloc((void**)&p->d_stage, (size_t)p->stage_len * sizeof(double)) != hipSuccess ||
        hipMalloc((void**)&p->d_lo, (size_t)(lo > 0 ? lo * lo : 1) * sizeof(double)) != hipSuccess ||
        hipMalloc((void**)&p->d_done, (size_t)(2 * num_strips + 2) * sizeof(int32_t)) != hipSuccess) {
        qrk_bbs_plan_destroy(p);
        return fail(h, QRK_STATUS_ALLOC_FAILED, "qrk_bbs_plan_create: cannot allocate the factors (Q of stage A, panels and T of stage B)");
    }
    *out = p;
    return QRK_STATUS_OK;
}

qrk_status qrk_bbs_plan_destroy(qrk_bbs_plan p)
{
    if (!p) return QRK_STATUS_OK;
    if (p->bd) (void)qrk_bd_plan_destroy(p->bd);
    (void)hipFree(p->d_panels); (void)hipFree(p->d_rlim); (void)hipFree(p->d_q); (void)hipFree(p->d_ra); (void)hipFree(p->d_perm);
    (void)hipFree(p->d_y); (void)hipFree(p->d_t); (void)hipFree(p->d_stage); (void)hipFree(p->d_lo); (void)hipFree(p->d_done);
    (void)hipFree(p->d_cmap); (void)hipFree(p->d_amap); (void)hipFree(p->d_carry); (void)hipFree(p->d_cprod); (void)hipFree(p->d_aprod);
    (void)hipFree(p->d_gvec);
    delete p;
    return QRK_STATUS_OK;
}

qrk_status qrk_bbs_plan_sizes(qrk_bbs_plan p, int64_t* rows, int64_t* cols, int64_t* r_len)
{
    if (!p) return QRK_STATUS_INVALID_ARGUMENT;
    if (rows) *rows = p->rows;
    if (cols) *cols = p->cols;
    if (r_len) *r_len = p->stage_len;
    return QRK_STATUS_OK;
}

qrk_status qrk_bbs_factorize(qrk_bbs_plan p, const double* strips)
{
    if (!p || !strips) return fail(p ? p->h : nullptr, QRK_STATUS_INVALID_ARGUMENT, "qrk_bbs_factorize: bad argument");
    qrk_handle h = p->h;
    QRK_HIP(h, hipSetDevice(h->device));
    p->factorized = false;
    p->maps_ready = false;
    // stage A on all CUs: every strip triangularised on its own
    qrk_status st = qrk_bd_factorize(p->bd, strips, p->d_q, p->d_ra, p->d_perm, nullptr, QRK_MEM_DEVICE);
    if (st != QRK_STATUS_OK) return st;
    // stage B: the chain merges the carried triangle with the strip's
    const int ng = p->n / 16;
    int piped = 0;
    QRK_HIP(h, qrk::launch_bbs_chain(p->d_panels, (int)p->N, p->d_ra, (int64_t)p->n * (p->n + 1) / 2, p->n, p->lo, p->max_act, p->d_lo,
                                     p->d_y, p->d_t, p->d_stage, p->d_rlim, p->d_rlim + ng, p->d_done, 0, &piped, h->stream));
    if (piped) {
        // the workgroups of the pipelined chain are an ordinary launch, not guaranteed to be co-resident: a wait that ran out raised
        // the chain's abort word (banded.hip, bb_pipe_wait) and the factors are unfinished -- run stage B again on ONE workgroup
        int32_t aborted = 0;
        QRK_HIP(h, hipMemcpyAsync(&aborted, p->d_done + p->N, sizeof(int32_t), hipMemcpyDeviceToHost, h->stream));
        QRK_HIP(h, hipStreamSynchronize(h->stream));
        if (aborted) {
            ++p->chain_reruns;
            QRK_HIP(h, qrk::launch_bbs_chain(p->d_panels, (int)p->N, p->d_ra, (int64_t)p->n * (p->n + 1) / 2, p->n, p->lo, p->max_act,
                                             p->d_lo, p->d_y, p->d_t, p->d_stage, p->d_rlim, p->d_rlim + ng, p->d_done, 1, nullptr,
                                             h->stream));
            if (std::getenv("QRK_BBS_PIPE_VERBOSE"))
                std::fprintf(stderr, "qrk_bbs_factorize: the pipelined chain was given up (a wait ran out); stage B ran again on one workgroup\n");
        }
    }
    p->factorized = true;
    if (std::getenv("QRK_BBS_PROF_DUMP")) {
        // diagnostic builds of banded.hip (-DQRK_BB_PROF) leave the chain's tick counts in the T of the last panel
        double t[20];
        QRK_HIP(h, hipStreamSynchronize(h->stream));
        QRK_HIP(h, hipMemcpy(t, p->d_t + p->panels.back().t_off, sizeof(t), hipMemcpyDeviceToHost));
        std::fprintf(stderr, "bb_chain2 ticks: carry-in %.0f  panel QR %.0f  R rows + carry out %.0f  hc %.0f | block->regs %.0f  reflectors %.0f  T recurrences %.0f  "
                             "update %.0f (V^T W %.0f  partial sums %.0f  T^T w %.0f  W -= V u %.0f) | block store %.0f  unit-lower V %.0f  Gram %.0f | of the reflectors: thread 0 waiting at the barrier as a non-owner %.0f | pipeline: publishing rows-final %.0f  waiting for the panel before %.0f\n",
                     t[0], t[1], t[2], t[3], t[6], t[7], t[8], t[9], t[10], t[11], t[12], t[13], t[14], t[15], t[16], t[17], t[18], t[19]);
    }
    return QRK_STATUS_OK;
}

qrk_status qrk_bbs_r_rows(qrk_bbs_plan p, int64_t strip, double* r_rows)
{
    if (!p || !r_rows || strip < 0 || strip >= p->N) return fail(p ? p->h : nullptr, QRK_STATUS_INVALID_ARGUMENT, "qrk_bbs_r_rows: bad argument");
    if (!p->factorized) return fail(p->h, QRK_STATUS_NOT_FACTORIZED, "qrk_bbs_r_rows: qrk_bbs_factorize has not run on this plan");
    qrk_handle h = p->h;
    const qrk::BBPanel& q = p->panels[(size_t)strip];
    QRK_HIP(h, hipMemcpyAsync(r_rows, p->d_stage + q.r_off, (size_t)q.solved * q.ncols * sizeof(double), hipMemcpyDeviceToDevice, h->stream));
    return QRK_STATUS_OK;
}

// The maps of the current factorisation and the scratch for nrhs right-hand sides (banded_maps.hip).  False: the one-workgroup chains run
// (lo = 0 or a single strip: there is no chain; QRK_BBS_MAPS=0; more than 65 535 right-hand sides; no memory for the maps: 2 x 131 KB per
// strip at the BASELINE configs[2] shape).
static bool bbs_maps(qrk_bbs_plan p, int64_t nrhs)
{
    const char* sw = std::getenv("QRK_BBS_MAPS");          // (read per call: the tests switch between the two forms in one process)
    const bool off = sw && sw[0] == '0';
    if (off || p->maps_off || p->lo <= 0 || p->N < 2 || nrhs > 65535 || nrhs <= 0) return false;
    qrk_handle h = p->h;
    auto give_up = [&]() { (void)hipGetLastError(); p->maps_off = true; return false; };
    const size_t l2 = (size_t)p->lo * p->lo * sizeof(double);
    if (std::getenv("QRK_BBS_MAPS_FAIL")) return give_up();      // (tests: an allocation of the maps that fails -- the plan stays on the old chains)
    if (!p->d_cmap) {
        int64_t cl = 0, al = 0;
        int K = 0;
        qrk::bbs_maps_sizes((int)p->N, p->lo, &K, &cl, &al, &p->gvec_per_rhs);
        if (const char* e = std::getenv("QRK_BBS_MAPS_K")) {      // (group size by hand; 0: one level)
            K = std::atoi(e);
            if (K < 0 || p->lo > 128) K = 0;
            if (K > 0) { cl = ((p->N - 2 + K - 1) / K) * (int64_t)p->lo * p->lo; al = ((p->N - 1 + K - 1) / K) * (int64_t)p->lo * p->lo;
                         p->gvec_per_rhs = ((p->N - 1 + K - 1) / K + 1) * (int64_t)p->lo; }
        }
        p->maps_K = K;
        if (hipMalloc((void**)&p->d_cmap, (size_t)p->N * l2) != hipSuccess) return give_up();
        // (the map form of the triangular solve carries lo numbers from panel to panel and takes its s new rows out of them: it needs
        //  lo >= s -- an overlap shorter than the column step stays with the one-workgroup chain, bb_solve_r_kernel)
        if (p->s <= 64 && p->lo >= p->s && hipMalloc((void**)&p->d_amap, (size_t)p->N * l2) != hipSuccess) return give_up();
        if (K > 0 && (hipMalloc((void**)&p->d_cprod, (size_t)(cl > 0 ? cl : 1) * sizeof(double)) != hipSuccess ||
                      hipMalloc((void**)&p->d_aprod, (size_t)(al > 0 ? al : 1) * sizeof(double)) != hipSuccess)) return give_up();
    }
    if (nrhs > p->carry_cap) {
        (void)hipStreamSynchronize(h->stream);      // (a product that still reads the old scratch)
        (void)hipFree(p->d_carry); (void)hipFree(p->d_gvec); p->d_carry = p->d_gvec = nullptr; p->carry_cap = 0;
        if (hipMalloc((void**)&p->d_carry, (size_t)nrhs * p->N * p->lo * sizeof(double)) != hipSuccess ||
            hipMalloc((void**)&p->d_gvec, (size_t)nrhs * p->gvec_per_rhs * sizeof(double)) != hipSuccess) return give_up();
        p->carry_cap = nrhs;
    }
    if (!p->maps_ready) {
        if (qrk::launch_bbs_maps(p->d_panels, (int)p->N, p->d_y, p->d_t, p->d_stage, p->n, p->lo, p->d_cmap, p->d_amap, p->maps_K, p->d_cprod,
                                 p->d_aprod, h->stream) != hipSuccess)
            return give_up();
        p->maps_ready = true;
    }
    return true;
}

qrk_status qrk_bbs_apply_q(qrk_bbs_plan p, int transpose, const double* v, double* out, int64_t nrhs, double* work)
{
    if (!p || !v || !out || !work || nrhs < 0 || v == out)
        return fail(p ? p->h : nullptr, QRK_STATUS_INVALID_ARGUMENT, "qrk_bbs_apply_q: bad argument");
    if (!p->factorized) return fail(p->h, QRK_STATUS_NOT_FACTORIZED, "qrk_bbs_apply_q: qrk_bbs_factorize has not run on this plan");
    qrk_handle h = p->h;
    QRK_HIP(h, hipSetDevice(h->device));
    qrk_status st;
    const bool maps = bbs_maps(p, nrhs);
    if (transpose) {
        // work = per strip Q_i^T v_i, then the chain: out = Q^T v in the layout of the header
        if ((st = qrk_bd_apply_qt(p->bd, p->d_q, v, nrhs, work, QRK_MEM_DEVICE)) != QRK_STATUS_OK) return st;
        if (maps)
            QRK_HIP(h, qrk::launch_bbs_apply_maps(p->d_panels, (int)p->N, p->d_y, p->d_t, p->d_cmap, p->d_cprod, p->maps_K, 1, work, p->rows, out,
                                                  p->rows, nrhs, p->ms, p->n, p->s, p->lo, (int)p->cols, p->max_act, p->d_carry, p->d_gvec, h->stream));
        else
            QRK_HIP(h, qrk::launch_bbs_apply(p->d_panels, (int)p->N, p->d_y, p->d_t, 1, work, p->rows, out, p->rows, nrhs, p->ms, p->n, p->s,
                                             p->lo, (int)p->cols, p->max_act, h->stream));
    } else {
        if (maps)
            QRK_HIP(h, qrk::launch_bbs_apply_maps(p->d_panels, (int)p->N, p->d_y, p->d_t, p->d_cmap, p->d_cprod, p->maps_K, 0, work, p->rows,
                                                  const_cast<double*>(v), p->rows, nrhs, p->ms, p->n, p->s, p->lo, (int)p->cols, p->max_act,
                                                  p->d_carry, p->d_gvec, h->stream));
        else
            QRK_HIP(h, qrk::launch_bbs_apply(p->d_panels, (int)p->N, p->d_y, p->d_t, 0, work, p->rows, const_cast<double*>(v), p->rows, nrhs,
                                             p->ms, p->n, p->s, p->lo, (int)p->cols, p->max_act, h->stream));
        if ((st = qrk_bd_apply_q(p->bd, p->d_q, work, nrhs, out, QRK_MEM_DEVICE)) != QRK_STATUS_OK) return st;
    }
    return QRK_STATUS_OK;
}

qrk_status qrk_bbs_solve(qrk_bbs_plan p, const double* b, double* x, int64_t nrhs, double* work)
{
    if (!p || !b || !x || !work || nrhs < 0)
        return fail(p ? p->h : nullptr, QRK_STATUS_INVALID_ARGUMENT, "qrk_bbs_solve: bad argument");
    if (!p->factorized) return fail(p->h, QRK_STATUS_NOT_FACTORIZED, "qrk_bbs_solve: qrk_bbs_factorize has not run on this plan");
    qrk_handle h = p->h;
    // work: 2 * rows * nrhs doubles: [Q_i^T b_i of stage A | Q^T b]; x(0:cols) = R^-1 (Q^T b)(0:cols)
    double* qtb = work + p->rows * nrhs;
    qrk_status st = qrk_bbs_apply_q(p, 1, b, qtb, nrhs, work);
    if (st != QRK_STATUS_OK) return st;
    if (bbs_maps(p, nrhs) && p->d_amap)
        QRK_HIP(h, qrk::launch_bbs_solve_r_maps(p->d_panels, (int)p->N, p->d_stage, p->d_amap, p->d_aprod, p->maps_K, p->s, p->lo, (int)p->cols,
                                                qtb, p->rows, nrhs, p->d_carry, p->d_gvec, h->stream));
    else
        QRK_HIP(h, qrk::launch_bb_solve_r(p->d_panels, (int)p->N, p->d_stage, (int)p->cols, qtb, p->rows, nrhs, h->stream));
    QRK_HIP(h, hipMemcpy2DAsync(x, (size_t)p->cols * sizeof(double), qtb, (size_t)p->rows * sizeof(double), (size_t)p->cols * sizeof(double),
                                (size_t)nrhs, hipMemcpyDeviceToDevice, h->stream));
    return QRK_STATUS_OK;
}

qrk_status qrk_bb_plan_info(qrk_bb_plan p, int32_t* num_blocks, int64_t* nnz_r, int64_t* y_len, int64_t* t_len,
                            int32_t* has_row_perm)
{
    if (!p) return QRK_STATUS_INVALID_ARGUMENT;
    if (num_blocks) *num_blocks = (int32_t)p->st.blocks.size();
    if (nnz_r) *nnz_r = p->st.nnz_r;
    if (y_len) *y_len = p->st.y_len;
    if (t_len) *t_len = p->st.t_len;
    if (has_row_perm) *has_row_perm = p->st.has_row_perm ? 1 : 0;
    return QRK_STATUS_OK;
}

qrk_status qrk_bb_plan_blocks(qrk_bb_plan p, int32_t* blocks, int32_t* row_perm, int64_t* yty)
{
    if (!p) return QRK_STATUS_INVALID_ARGUMENT;
    const qrk::BandedStructure& st = p->st;
    if (blocks)
        for (size_t i = 0; i < st.blocks.size(); ++i) {
            blocks[4 * i] = st.blocks[i].idxRow; blocks[4 * i + 1] = st.blocks[i].idxCol;
            blocks[4 * i + 2] = st.blocks[i].numRows; blocks[4 * i + 3] = st.blocks[i].numCols;
        }
    if (row_perm) std::memcpy(row_perm, st.row_perm.data(), st.row_perm.size() * sizeof(int32_t));
    if (yty)
        for (size_t i = 0; i < st.panels.size(); ++i) {
            const qrk::BBPanel& q = st.panels[i];
            yty[6 * i] = q.yrow; yty[6 * i + 1] = q.num_zeros; yty[6 * i + 2] = q.act_rows; yty[6 * i + 3] = q.ncols;
            yty[6 * i + 4] = q.y_off; yty[6 * i + 5] = q.t_off;
        }
    return QRK_STATUS_OK;
}

qrk_status qrk_bb_pattern(qrk_bb_plan p, int32_t* r_colptr, int32_t* r_rowidx, qrk_memspace space)
{
    if (!p || !r_colptr || !r_rowidx) return fail(p ? p->h : nullptr, QRK_STATUS_INVALID_ARGUMENT, "qrk_bb_pattern: NULL argument");
    qrk_handle h = p->h;
    const qrk::BandedStructure& st = p->st;
    if (space == QRK_MEM_HOST) {
        std::memcpy(r_colptr, st.r_colptr.data(), st.r_colptr.size() * sizeof(int32_t));
        std::memcpy(r_rowidx, st.r_rowidx.data(), st.r_rowidx.size() * sizeof(int32_t));
        return QRK_STATUS_OK;
    }
    QRK_HIP(h, hipMemcpyAsync(r_colptr, p->d_rcolptr, st.r_colptr.size() * sizeof(int32_t), hipMemcpyDeviceToDevice, h->stream));
    QRK_HIP(h, hipMemcpyAsync(r_rowidx, p->d_rrowidx, st.r_rowidx.size() * sizeof(int32_t), hipMemcpyDeviceToDevice, h->stream));
    return QRK_STATUS_OK;
}

qrk_status qrk_bb_factorize(qrk_bb_plan p, const double* csr_vals, int64_t nnz, double* r_vals, double* y_vals,
                            double* t_vals, qrk_memspace space)
{
    if (!p || !csr_vals || !r_vals || !y_vals || !t_vals)
        return fail(p ? p->h : nullptr, QRK_STATUS_INVALID_ARGUMENT, "qrk_bb_factorize: NULL argument");
    qrk_handle h = p->h;
    const qrk::BandedStructure& st = p->st;
    if (nnz != (int64_t)st.pmap.size()) return fail(h, QRK_STATUS_INVALID_ARGUMENT, "qrk_bb_factorize: nnz differs from the analysed pattern");
    QRK_HIP(h, hipSetDevice(h->device));
    if (space == QRK_MEM_DEVICE) {
        QRK_HIP(h, qrk::launch_bb_chain(p->d_panels, (int)st.panels.size(), p->d_prowptr, p->d_pcol, p->d_pmap, csr_vals, p->d_W,
                                        p->d_lo, y_vals, t_vals, p->d_stage, p->d_rsrc, st.nnz_r, r_vals, st.max_act_rows,
                                        st.max_ncols, h->stream));
        p->factorized = true;
        return QRK_STATUS_OK;
    }
    Staging s(h);
    double *d_v, *d_r, *d_y, *d_t;
    qrk_status stt;
    if ((stt = s.in(csr_vals, nnz, &d_v)) || (stt = s.out(st.nnz_r, &d_r)) || (stt = s.out(st.y_len, &d_y)) ||
        (stt = s.out(st.t_len, &d_t)))
        return stt;
    QRK_HIP(h, qrk::launch_bb_chain(p->d_panels, (int)st.panels.size(), p->d_prowptr, p->d_pcol, p->d_pmap, d_v, p->d_W, p->d_lo,
                                    d_y, d_t, p->d_stage, p->d_rsrc, st.nnz_r, d_r, st.max_act_rows, st.max_ncols, h->stream));
    if ((stt = s.back(r_vals, d_r, st.nnz_r)) || (stt = s.back(y_vals, d_y, st.y_len)) || (stt = s.back(t_vals, d_t, st.t_len)))
        return stt;
    QRK_HIP(h, hipStreamSynchronize(h->stream));
    p->factorized = true;
    return QRK_STATUS_OK;
}

qrk_status qrk_bb_solve_r(qrk_bb_plan p, double* v, int64_t ldv, int64_t nrhs, qrk_memspace space)
{
    if (!p || !v || nrhs < 0) return fail(p ? p->h : nullptr, QRK_STATUS_INVALID_ARGUMENT, "qrk_bb_solve_r: bad argument");
    qrk_handle h = p->h;
    const qrk::BandedStructure& st = p->st;
    if (ldv < st.cols) return fail(h, QRK_STATUS_INVALID_ARGUMENT, "qrk_bb_solve_r: ldv is smaller than the number of columns");
    if (!p->factorized) return fail(h, QRK_STATUS_INVALID_ARGUMENT, "qrk_bb_solve_r: qrk_bb_factorize has not run on this plan");
    QRK_HIP(h, hipSetDevice(h->device));
    if (space == QRK_MEM_DEVICE) {
        QRK_HIP(h, qrk::launch_bb_solve_r(p->d_panels, (int)st.panels.size(), p->d_stage, st.cols, v, ldv, nrhs, h->stream));
        return QRK_STATUS_OK;
    }
    Staging s(h);
    double* d_v;
    qrk_status stt;
    if ((stt = s.in((const double*)v, nrhs * ldv, &d_v))) return stt;
    QRK_HIP(h, qrk::launch_bb_solve_r(p->d_panels, (int)st.panels.size(), p->d_stage, st.cols, d_v, ldv, nrhs, h->stream));
    if ((stt = s.back(v, d_v, nrhs * ldv))) return stt;
    QRK_HIP(h, hipStreamSynchronize(h->stream));
    return QRK_STATUS_OK;
}

qrk_status qrk_bb_apply_q(qrk_bb_plan p, const double* y_vals, const double* t_vals, int transpose, double* v,
                          int64_t nrhs, qrk_memspace space)
{
    if (!p || !y_vals || !t_vals || !v || nrhs < 0)
        return fail(p ? p->h : nullptr, QRK_STATUS_INVALID_ARGUMENT, "qrk_bb_apply_q: bad argument");
    qrk_handle h = p->h;
    const qrk::BandedStructure& st = p->st;
    QRK_HIP(h, hipSetDevice(h->device));
    if (space == QRK_MEM_DEVICE) {
        QRK_HIP(h, qrk::launch_bb_apply_q(p->d_panels, (int)st.panels.size(), y_vals, t_vals, transpose, v, st.rows, nrhs,
                                          st.max_act_rows, st.max_ncols, h->stream));
        return QRK_STATUS_OK;
    }
    Staging s(h);
    double *d_y, *d_t, *d_v;
    qrk_status stt;
    if ((stt = s.in(y_vals, st.y_len, &d_y)) || (stt = s.in(t_vals, st.t_len, &d_t)) ||
        (stt = s.in((const double*)v, nrhs * st.rows, &d_v)))
        return stt;
    QRK_HIP(h, qrk::launch_bb_apply_q(p->d_panels, (int)st.panels.size(), d_y, d_t, transpose, d_v, st.rows, nrhs,
                                      st.max_act_rows, st.max_ncols, h->stream));
    if ((stt = s.back(v, d_v, nrhs * st.rows))) return stt;
    QRK_HIP(h, hipStreamSynchronize(h->stream));
    return QRK_STATUS_OK;
}

qrk_status qrk_bd_time_factorize(qrk_bd_plan p, const double* tiles, double* q_vals, double* r_vals,
                                 int32_t* perm, int nsets, int iters, float* avg_ms)
{
    if (!p || !avg_ms || nsets <= 0 || iters <= 0)
        return fail(p ? p->h : nullptr, QRK_STATUS_INVALID_ARGUMENT, "qrk_bd_time_factorize: bad argument");
    qrk_handle h = p->h;
    if (p->landscape) return fail(h, QRK_STATUS_INVALID_ARGUMENT, "qrk_bd_time_factorize: landscape tile");
    QRK_HIP(h, hipSetDevice(h->device));
    // (the events are destroyed on every path: a guard instead of QRK_HIP's early return past them)
    struct Events {
        hipEvent_t e0 = nullptr, e1 = nullptr;
        ~Events() { if (e0) (void)hipEventDestroy(e0); if (e1) (void)hipEventDestroy(e1); }
    } ev;
    QRK_HIP(h, hipEventCreate(&ev.e0));
    QRK_HIP(h, hipEventCreate(&ev.e1));
    QRK_HIP(h, hipEventRecord(ev.e0, h->stream));
    qrk_status st = QRK_STATUS_OK;
    for (int it = 0; it < iters && st == QRK_STATUS_OK; ++it) {
        const int64_t s = it % nsets;
        st = enqueue_factorize(p, tiles + s * p->tiles_len, q_vals + s * p->nnz_q, r_vals + s * p->nnz_r,
                               perm + s * (int64_t)p->mat_cols, nullptr);
    }
    QRK_HIP(h, hipEventRecord(ev.e1, h->stream));
    QRK_HIP(h, hipEventSynchronize(ev.e1));
    float ms = 0.f;
    QRK_HIP(h, hipEventElapsedTime(&ms, ev.e0, ev.e1));
    *avg_ms = ms / (float)iters;
    p->factorized = st == QRK_STATUS_OK;
    return st;
}

// ---- multi-GPU: contiguous shards of the diagonal blocks -------------------------------------------------------
qrk_status qrk_shard_ranges(int64_t B, int32_t block_rows, int32_t block_cols, const int32_t* rows, const int32_t* cols, int32_t world,
                            qrk_shard* shards)
{
    if (B < 0 || world <= 0 || !shards || ((rows == nullptr) != (cols == nullptr)) || (!rows && B > 0 && (block_rows <= 0 || block_cols <= 0)))
        return fail(nullptr, QRK_STATUS_INVALID_ARGUMENT, "qrk_shard_ranges: bad argument");
    auto br = [&](int64_t i) -> int64_t { return rows ? rows[i] : block_rows; };
    auto bc = [&](int64_t i) -> int64_t { return cols ? cols[i] : block_cols; };
    // bounds: the cut closest to k/world of the total Householder cost r c^2 (running sums in double, as the Python mirror's
    // numpy.cumsum: qrkit_amd/sharding.py::shard_ranges gives the same ranges)
    std::vector<double> cost((size_t)B);
    double run = 0.0;
    for (int64_t i = 0; i < B; ++i) { run += (double)br(i) * (double)bc(i) * (double)bc(i); cost[(size_t)i] = run; }
    std::vector<int64_t> bounds((size_t)world + 1, B);
    bounds[0] = 0;
    for (int32_t k = 1; k < world && B > 0; ++k) {
        const double target = run * (double)k / (double)world;
        int64_t idx = (int64_t)(std::lower_bound(cost.begin(), cost.end(), target) - cost.begin()) + 1;   // blocks before the cut
        if (idx - 1 > bounds[(size_t)k - 1] && std::fabs(cost[(size_t)idx - 2] - target) <= std::fabs(cost[(size_t)idx - 1] - target)) --idx;
        bounds[(size_t)k] = std::min(std::max(idx, bounds[(size_t)k - 1]), B);
    }
    if (B == 0) for (int32_t k = 1; k < world; ++k) bounds[(size_t)k] = 0;
    // running offsets (BlockDiagonalSparseQR.h:428-431, 524-525) at every bound
    qrk_shard acc{};
    int32_t k = 0;
    for (int64_t i = 0; i <= B; ++i) {
        while (k <= world && bounds[(size_t)k] == i) {
            shards[k] = acc;
            shards[k].first_block = i;
            shards[k].num_blocks = k < world ? bounds[(size_t)k + 1] - i : 0;
            ++k;
        }
        if (i == B) break;
        const int64_t r = br(i), c = bc(i);
        if (r <= 0 || c <= 0) return fail(nullptr, QRK_STATUS_INVALID_ARGUMENT, "qrk_shard_ranges: non-positive tile size");
        acc.base_row += r; acc.base_col += c;
        acc.tiles_off += r * c; acc.q_off += r * r; acc.r_off += c * (c + 1) / 2;
    }
    return QRK_STATUS_OK;
}

}  // extern "C"

namespace {
__global__ void __launch_bounds__(256) add_base_kernel(int32_t* __restrict__ v, int64_t n, int32_t base)
{
    for (int64_t i = blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) v[i] += base;
}

// The four RCCL entry points the gather needs, taken from the RCCL that the process already has (the one that made the caller's
// communicator -- torch ships its own copy), else from librccl.so.
struct RcclApi {
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    bool ok = false;
};
const RcclApi& rccl_api()
{
    static const RcclApi api = [] {
        RcclApi a;
        // the RCCL that made the caller's communicator: global symbols first (torch links its own copy), then a librccl that is
        // ALREADY loaded under either name (RTLD_NOLOAD: never a second instance beside it), and only then a fresh load -- a process
        // whose communicator came from a statically linked or differently named RCCL must export ncclSend itself
        void* lib = RTLD_DEFAULT;
        if (!dlsym(lib, "ncclSend")) {
            lib = dlopen("librccl.so.1", RTLD_NOW | RTLD_NOLOAD);
            if (!lib) lib = dlopen("librccl.so", RTLD_NOW | RTLD_NOLOAD);
            if (!lib) lib = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
            if (!lib) lib = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
            if (!lib) return a;
        }
        a.GroupStart = reinterpret_cast<decltype(a.GroupStart)>(dlsym(lib, "ncclGroupStart"));
        a.GroupEnd = reinterpret_cast<decltype(a.GroupEnd)>(dlsym(lib, "ncclGroupEnd"));
        a.Send = reinterpret_cast<decltype(a.Send)>(dlsym(lib, "ncclSend"));
        a.Recv = reinterpret_cast<decltype(a.Recv)>(dlsym(lib, "ncclRecv"));
        a.GetErrorString = reinterpret_cast<decltype(a.GetErrorString)>(dlsym(lib, "ncclGetErrorString"));
        a.ok = a.GroupStart && a.GroupEnd && a.Send && a.Recv;
        return a;
    }();
    return api;
}
}  // namespace

extern "C" {

// Equal pieces to the root / one buffer from the root: the two exchanges of the angular solver sharded by rows (one n x n triangle per
// rank up, the permutation of the right block down; BlockAngularSparseQR.h:361-369, 459-514 -- include/qrkit_amd.h).  Grouped
// ncclSend / ncclRecv on the handle's stream like qrk_gather_r, whose caveat applies: the checks below depend only on arguments that
// are the same on every rank, so that all ranks enter the group or none does.
qrk_status qrk_gather_equal(qrk_handle h, void* nccl_comm, int32_t rank, int32_t world, int32_t root, const double* send, int64_t count,
                            double* recv)
{
    if (!h || world <= 0 || rank < 0 || rank >= world || root < 0 || root >= world || count < 0 || (world > 1 && !nccl_comm))
        return fail(h, QRK_STATUS_INVALID_ARGUMENT, "qrk_gather_equal: bad argument");
    if (world > 1 && !rccl_api().ok) return fail(h, QRK_STATUS_UNSUPPORTED, "qrk_gather_equal: no RCCL in this process and librccl.so cannot be loaded");
    QRK_HIP(h, hipSetDevice(h->device));
    const bool local_ok = (count == 0 || send) && (rank != root || count == 0 || recv);
    if (rank == root && local_ok && count > 0)
        QRK_HIP(h, hipMemcpyAsync(recv + (int64_t)rank * count, send, (size_t)count * sizeof(double), hipMemcpyDeviceToDevice, h->stream));
    if (world > 1 && count > 0) {
        const RcclApi& api = rccl_api();
        ncclComm_t comm = static_cast<ncclComm_t>(nccl_comm);
        ncclResult_t e = api.GroupStart();
        if (e == ncclSuccess && local_ok) {
            if (rank == root) {
                for (int32_t peer = 0; peer < world && e == ncclSuccess; ++peer)
                    if (peer != root) e = api.Recv(recv + (int64_t)peer * count, (size_t)count, ncclFloat64, peer, comm, h->stream);
            } else {
                e = api.Send(send, (size_t)count, ncclFloat64, root, comm, h->stream);
            }
        }
        const ncclResult_t e2 = api.GroupEnd();
        if (e != ncclSuccess || e2 != ncclSuccess)
            return fail(h, QRK_STATUS_HIP_ERROR, std::string("qrk_gather_equal: RCCL: ") + (api.GetErrorString ? api.GetErrorString(e != ncclSuccess ? e : e2) : "error"));
    }
    if (!local_ok) return fail(h, QRK_STATUS_INVALID_ARGUMENT, "qrk_gather_equal: NULL buffer (this rank sent / received nothing: its peers wait)");
    return QRK_STATUS_OK;
}

qrk_status qrk_bcast(qrk_handle h, void* nccl_comm, int32_t rank, int32_t world, int32_t root, void* buf, int64_t bytes)
{
    if (!h || world <= 0 || rank < 0 || rank >= world || root < 0 || root >= world || bytes < 0 || (world > 1 && !nccl_comm))
        return fail(h, QRK_STATUS_INVALID_ARGUMENT, "qrk_bcast: bad argument");
    if (world == 1 || bytes == 0) return QRK_STATUS_OK;
    const RcclApi& api = rccl_api();
    if (!api.ok) return fail(h, QRK_STATUS_UNSUPPORTED, "qrk_bcast: no RCCL in this process and librccl.so cannot be loaded");
    QRK_HIP(h, hipSetDevice(h->device));
    ncclComm_t comm = static_cast<ncclComm_t>(nccl_comm);
    ncclResult_t e = api.GroupStart();
    if (e == ncclSuccess && buf) {
        if (rank == root) {
            for (int32_t peer = 0; peer < world && e == ncclSuccess; ++peer)
                if (peer != root) e = api.Send(buf, (size_t)bytes, ncclInt8, peer, comm, h->stream);
        } else {
            e = api.Recv(buf, (size_t)bytes, ncclInt8, root, comm, h->stream);
        }
    }
    const ncclResult_t e2 = api.GroupEnd();
    if (e != ncclSuccess || e2 != ncclSuccess)
        return fail(h, QRK_STATUS_HIP_ERROR, std::string("qrk_bcast: RCCL: ") + (api.GetErrorString ? api.GetErrorString(e != ncclSuccess ? e : e2) : "error"));
    if (!buf) return fail(h, QRK_STATUS_INVALID_ARGUMENT, "qrk_bcast: NULL buffer");
    return QRK_STATUS_OK;
}

qrk_status qrk_gather_r(qrk_handle h, void* nccl_comm, int32_t rank, int32_t world, int32_t root, const qrk_shard* sh,
                        const double* r_local, const int32_t* perm_local, double* r_all, int32_t* perm_all)
{
    if (!h || !sh || world <= 0 || rank < 0 || rank >= world || root < 0 || root >= world || (world > 1 && !nccl_comm) ||
        (rank == root && (!r_all || !perm_all)))
        return fail(h, QRK_STATUS_INVALID_ARGUMENT, "qrk_gather_r: bad argument");
    QRK_HIP(h, hipSetDevice(h->device));
    const int64_t nr = sh[rank + 1].r_off - sh[rank].r_off, nc = sh[rank + 1].base_col - sh[rank].base_col;
    if ((nr > 0 && !r_local) || (nc > 0 && !perm_local)) return fail(h, QRK_STATUS_INVALID_ARGUMENT, "qrk_gather_r: NULL shard");
    if (rank == root) {
        if (nr > 0) QRK_HIP(h, hipMemcpyAsync(r_all + sh[rank].r_off, r_local, (size_t)nr * sizeof(double), hipMemcpyDeviceToDevice, h->stream));
        if (nc > 0) QRK_HIP(h, hipMemcpyAsync(perm_all + sh[rank].base_col, perm_local, (size_t)nc * sizeof(int32_t), hipMemcpyDeviceToDevice, h->stream));
    }
    if (world > 1) {
        const RcclApi& api = rccl_api();
        if (!api.ok) return fail(h, QRK_STATUS_UNSUPPORTED, "qrk_gather_r: no RCCL in this process and librccl.so cannot be loaded");
        ncclComm_t comm = static_cast<ncclComm_t>(nccl_comm);
        auto nccl_fail = [&](ncclResult_t e, const char* what) {
            return fail(h, QRK_STATUS_HIP_ERROR, std::string("qrk_gather_r: ") + what + ": " + (api.GetErrorString ? api.GetErrorString(e) : "RCCL error"));
        };
        ncclResult_t e = api.GroupStart();
        if (e != ncclSuccess) return nccl_fail(e, "ncclGroupStart");
        if (rank == root) {
            for (int32_t peer = 0; peer < world && e == ncclSuccess; ++peer) {
                if (peer == root) continue;
                const int64_t pr = sh[peer + 1].r_off - sh[peer].r_off, pc = sh[peer + 1].base_col - sh[peer].base_col;
                if (pr > 0) e = api.Recv(r_all + sh[peer].r_off, (size_t)pr, ncclFloat64, peer, comm, h->stream);
                if (pc > 0 && e == ncclSuccess) e = api.Recv(perm_all + sh[peer].base_col, (size_t)pc, ncclInt32, peer, comm, h->stream);
            }
        } else {
            if (nr > 0) e = api.Send(r_local, (size_t)nr, ncclFloat64, root, comm, h->stream);
            if (nc > 0 && e == ncclSuccess) e = api.Send(perm_local, (size_t)nc, ncclInt32, root, comm, h->stream);
        }
        const ncclResult_t e2 = api.GroupEnd();
        if (e != ncclSuccess) return nccl_fail(e, "ncclSend / ncclRecv");
        if (e2 != ncclSuccess) return nccl_fail(e2, "ncclGroupEnd");
    }
    if (rank == root) {
        // local column indices -> m_outputPerm_c.indices() of the whole matrix (BlockDiagonalSparseQR.h:519-521)
        for (int32_t g = 0; g < world; ++g) {
            const int64_t pc = sh[g + 1].base_col - sh[g].base_col;
            if (pc <= 0 || sh[g].base_col == 0) continue;
            if (sh[g + 1].base_col > INT32_MAX) return fail(h, QRK_STATUS_UNSUPPORTED, "qrk_gather_r: more than 2^31 columns");
            const unsigned grid = (unsigned)std::min<int64_t>((pc + 255) / 256, 4096);
            hipLaunchKernelGGL(add_base_kernel, dim3(grid), dim3(256), 0, h->stream, perm_all + sh[g].base_col, pc, (int32_t)sh[g].base_col);
        }
        QRK_HIP(h, hipGetLastError());
    }
    return QRK_STATUS_OK;
}

qrk_status qrk_gather_x(qrk_handle h, void* nccl_comm, int32_t rank, int32_t world, int32_t root, const qrk_shard* sh,
                        const double* x_local, int64_t nrhs, double* x_all)
{
    if (!h || !sh || world <= 0 || rank < 0 || rank >= world || root < 0 || root >= world || nrhs < 0 || (world > 1 && !nccl_comm) ||
        (rank == root && !x_all && nrhs > 0))
        return fail(h, QRK_STATUS_INVALID_ARGUMENT, "qrk_gather_x: bad argument");
    QRK_HIP(h, hipSetDevice(h->device));
    const int64_t total = sh[world].base_col, nc = sh[rank + 1].base_col - sh[rank].base_col;
    if (nc > 0 && nrhs > 0 && !x_local) return fail(h, QRK_STATUS_INVALID_ARGUMENT, "qrk_gather_x: NULL shard");
    if (nrhs == 0) return QRK_STATUS_OK;
    if (rank == root && nc > 0)
        QRK_HIP(h, hipMemcpy2DAsync(x_all + sh[rank].base_col, (size_t)total * sizeof(double), x_local, (size_t)nc * sizeof(double),
                                    (size_t)nc * sizeof(double), (size_t)nrhs, hipMemcpyDeviceToDevice, h->stream));
    if (world > 1) {
        const RcclApi& api = rccl_api();
        if (!api.ok) return fail(h, QRK_STATUS_UNSUPPORTED, "qrk_gather_x: no RCCL in this process and librccl.so cannot be loaded");
        ncclComm_t comm = static_cast<ncclComm_t>(nccl_comm);
        ncclResult_t e = api.GroupStart();
        if (e != ncclSuccess) return fail(h, QRK_STATUS_HIP_ERROR, "qrk_gather_x: ncclGroupStart failed");
        for (int64_t k = 0; k < nrhs && e == ncclSuccess; ++k) {
            if (rank == root) {
                for (int32_t peer = 0; peer < world && e == ncclSuccess; ++peer) {
                    const int64_t pc = sh[peer + 1].base_col - sh[peer].base_col;
                    if (peer != root && pc > 0) e = api.Recv(x_all + k * total + sh[peer].base_col, (size_t)pc, ncclFloat64, peer, comm, h->stream);
                }
            } else if (nc > 0) {
                e = api.Send(x_local + k * nc, (size_t)nc, ncclFloat64, root, comm, h->stream);
            }
        }
        const ncclResult_t e2 = api.GroupEnd();
        if (e != ncclSuccess || e2 != ncclSuccess)
            return fail(h, QRK_STATUS_HIP_ERROR, std::string("qrk_gather_x: ncclSend / ncclRecv: ") +
                                                     (api.GetErrorString ? api.GetErrorString(e != ncclSuccess ? e : e2) : "RCCL error"));
    }
    return QRK_STATUS_OK;
}

const char* qrk_bd_kernel_name(qrk_bd_plan p, int which)
{
    (void)which;
    if (!p) return "";
    qrk_handle h = p->h;
    const bool piv = p->solver == QRK_COLPIV_HOUSEHOLDER;
    if (h->force_exact) return piv ? "qrk::bdqr_exact_kernel<true>" : "qrk::bdqr_exact_kernel<false>";
    if (p->max_dim > QRK_COL_MAX_DIM) return "qrk::bdqr_wg_kernel";
    if (p->w64_uniform) return piv ? "qrk::bdqr_w64_kernel<true>" : "qrk::bdqr_w64_kernel<false>";
    if (p->max_dim > 32) return "qrk::bdqr_col_kernel";
    if (p->uniform && p->max_dim <= 16 && p->r >= p->c && p->c <= 2 && h->use_small_kernel && h->use_thin_kernel)
        return piv ? "qrk::thin::bdqr_thin_kernel<true, HC>" : "qrk::thin::bdqr_thin_kernel<false, HC>";
    if (p->uniform && p->max_dim <= 16 && h->use_small_kernel && h->use_quad_kernel && p->r >= h->quad_min_rows && qrk::bdqr_quad_supported(p->r, p->c))
        return piv ? "qrk::bdqr_quad_kernel<WR, true, HC>" : "qrk::bdqr_quad_kernel<WR, false, HC>";
    if (p->uniform && p->max_dim <= 16 && p->r >= p->c && h->use_small_kernel)
        return piv ? "qrk::bdqr_small_kernel<G, true>" : "qrk::bdqr_small_kernel<G, false>";
    // (tau is not stored by the measurement entry point and by callers that pass hcoeffs = NULL: the <.., false> instantiation)
    if (p->d_p4_scratch && p->k1_quad32) return piv ? "qrk::bdqr_quad32_kernel<true, false>" : "qrk::bdqr_quad32_kernel<false, false>";
    if (p->d_p4_scratch) {
        // (third parameter: the own-norm step, which launch_bdqr_pair4 picks for more than one round of the resident waves)
        const bool own = qrk::bdqr_pair4_own_norm(p->B, p->p4_wgs);
        if (piv) return own ? "qrk::bdqr_pair4_kernel<true, false, true>" : "qrk::bdqr_pair4_kernel<true, false, false>";
        return own ? "qrk::bdqr_pair4_kernel<false, false, true>" : "qrk::bdqr_pair4_kernel<false, false, false>";
    }
    if (p->uniform && p->r == 32 && p->c == 32) return piv ? "qrk::bdqr_pair32_kernel<true, false>" : "qrk::bdqr_pair32_kernel<false, false>";
    return piv ? "qrk::bdqr_pair_kernel<false, true, true>" : "qrk::bdqr_pair_kernel<false, false, true>";
}

}  // extern "C"
