// bdqr_split.hip -- EXPERIMENT (opt-in, QRK_SPLIT=1; the pair kernel stays the product path): uniform 32x32 batches in
// TWO kernels, the column-pivoted factorisation of A alone with more tiles in flight than the pair kernel can hold,
// then Q formed from the stored reflectors by a chain-free kernel.
//
// Same seam as bdqr_pair.hip: the body of the hot loop of QRKit::BlockDiagonalSparseQR::factorize
// (src/QRKit/BlockDiagonalSparseQR.h:432-526): blockSolver.compute(block) (:437-438, Eigen ColPivHouseholderQR) is
// `bdqr_factor32_kernel`, Qi = blockSolver.matrixQ() (:446) is `bdqr_formq32_kernel`; the value assembly (:455-500)
// and the column-permutation splice (:519-521) are their stores.
//
// The idea: in the pair kernel a step is a chain of dependent LDS round trips, the register file (A and Q^T: 128 VGPRs
// of data per lane) and LDS (an 8.7 KB image of A per tile) both cap the tiles in flight at 16 per CU, and 10 000 tiles
// need three rounds of 2048 wavefronts.  Without Q^T the factorisation needs half the registers
// (168 VGPRs: three waves on a SIMD), and with an image of rows 6..31 only -- the six steps that need rows 0..5 of a
// pivot column take them from the pivot lane's registers by ds_bpermute -- a pair needs <= 16 KB of LDS: ten waves per CU,
// 2560 slots, TWO rounds for 5000 pairs (rows 8..31 in the end, see below).  The reflectors leave as they are produced (the published pivot column x_k,
// un-normalised, and the two scalars s_k, ng_k of the step), and the second kernel applies them to the identity with
// the pair kernel's operations in the pair kernel's order, so Q is bitwise the pair kernel's Q.
//
// MEASURED (rocprofv3, 10 000 tiles, MI355X; the pair kernel takes 88-90 us on the same box): the factorisation kernel
// takes 88 us with eight workgroups per CU (three rounds), 84 us with nine, 82.5 us with ten (two rounds; R0 = 8 so that a
// workgroup's LDS is 12 allocation granules of 1280 B - at R0 = 6, 16 KB, only nine were resident and the tenth ran as a
// late extra round: 98 us).  Two rounds of ten waves take as long as three rounds of eight: a CU does NOT process more
// pairs per microsecond with more waves (0.276 pairs/us at eight, 0.244 at ten), i.e. the steps are bound by a resource
// the waves of a CU share - the LDS pipe that serves the image reads, the publish/broadcast pairs and the refreshes -
// not by occupancy.  The reflector stores cost ~10 us (78 us without them at eight workgroups), the missing prefetch
// another ~13 (the pair kernel with its Q work removed: 65 us).  The Q kernel as written takes 50 us (210 VGPRs, two
// waves per SIMD, one dependent FMA chain per dot).  Sum 133-143 us against 88-90: the split does not pay, and raising
// the occupancy of the factorisation is not the lever.  Parity-green and bitwise equal to the pair kernel
// (tests/test_split_gpu.py); kept opt-in as a record of the experiment.
#include "qrk_device.h"

#include <float.h>
#include <cstdlib>

#ifndef QRK_SPLIT_WAVES
#define QRK_SPLIT_WAVES 3      // waves per SIMD the factorisation kernel is compiled for
#endif

namespace qrk {

namespace split {

constexpr int WR = 32;               // rows = cols = row registers per column
#ifndef QRK_SPLIT_R0
#define QRK_SPLIT_R0 8
#endif
constexpr int R0 = QRK_SPLIT_R0;     // first row kept in the LDS image
constexpr int LDI = WR - R0 + 1;     // image column stride in doubles (odd: conflict-free 8-byte accesses)
constexpr int RB = 4;                // the image is refreshed every RB steps
constexpr int L_IMG = 0;             // [32][LDI] column-major image of rows R0..31; R staging in the epilogue
constexpr int L_XBUF = WR * LDI;     // [32] current pivot column
constexpr int L_WBUF = L_XBUF + WR;  // [RB][32] update coefficients of the last RB steps, per A column
constexpr int L_HALF = L_WBUF + RB * WR;   // R0 = 8: 960 doubles = 7680 B per half, 15360 B per wave = 12 allocation granules of
                                           // 1280 B -> 10 waves per CU (at R0 = 6, 16384 B, only nine were resident)
static_assert(L_HALF <= 1024, "two tiles must fit 16 KB of LDS");

constexpr int V_TILE = WR * WR;      // doubles of reflector vectors per tile: x_k at [k][0..31]
constexpr int S_TILE = 2 * WR;       // (s_k, ng_k) per step

constexpr double SQRT_EPS = 1.4901161193847656e-08;   // sqrt(DBL_EPSILON), Eigen's norm_downdate_threshold

#define QRK_0_31(M)                                                                              \
    M(0) M(1) M(2) M(3) M(4) M(5) M(6) M(7) M(8) M(9) M(10) M(11) M(12) M(13) M(14) M(15) M(16)  \
    M(17) M(18) M(19) M(20) M(21) M(22) M(23) M(24) M(25) M(26) M(27) M(28) M(29) M(30) M(31)

__device__ __forceinline__ double sqrt_pos(double x)      // as in bdqr_pair.hip: <= 1 ulp, no division
{
    const double y = __builtin_amdgcn_rsq(x);
    double g = x * y, h = 0.5 * y;
    const double e = fma(-h, g, 0.5);
    g = fma(g, e, g);
    h = fma(h, e, h);
    const double d = fma(-g, g, x);
    return fma(d, h, g);
}
__device__ __forceinline__ double recip(double x)
{
    double y = __builtin_amdgcn_rcp(x);
    double e = fma(-x, y, 1.0);
    y = fma(y, e, y);
    e = fma(-x, y, 1.0);
    y = fma(y, e, y);
    return y;
}
__device__ __forceinline__ double bpermute_f64(int byte_addr, double v)
{
    const int lo = __builtin_amdgcn_ds_bpermute(byte_addr, __double2loint(v));
    const int hi = __builtin_amdgcn_ds_bpermute(byte_addr, __double2hiint(v));
    return __hiloint2double(hi, lo);
}

struct LaneState {
    int lane, j;
    int sh8, hb4;
    bool live, ispiv;
    int lbl;
    unsigned long long livemask;
    int kstep;
    double nu2, thr_nd2;
    double h[RB];
};

// Rare path of the pivot search (exact tie of the leading words of the largest squared norm): as in bdqr_pair.hip,
// Eigen's first maximum = smallest CURRENT position, positions rebuilt by replaying the transpositions.
__device__ __forceinline__ bool resolve_ties(int K, int lane, int kstep, unsigned klo, bool cand)
{
    const int half = lane >> 5;
    const unsigned ml = half32_max_u32(cand ? klo : 0u);
    cand = cand && klo == ml;
    int p = lane & 31;
#pragma unroll 1
    for (int k = 0; k < K; ++k) {
        const unsigned long long m = __builtin_amdgcn_ballot_w64(kstep == k);
        const unsigned mh = half ? (unsigned)(m >> 32) : (unsigned)m;
        const int l = (mh ? __ffs((int)mh) - 1 : 0) + 32 * half;
        const int pl = __builtin_amdgcn_ds_bpermute(l * 4, p);
        if (mh) p = (lane == l) ? k : (p == k ? pl : p);
    }
    const int pc = cand ? p : 64;
    const int pmin = half32_min_i32(pc);
    return pc == pmin && pc != 64;
}

// Head of step K: pivot search, then entry j of the pivot column -- from the image (rows >= R0: exact through step KR-1,
// plus the rank-1 corrections of steps KR..K-1) or, for the rows the image does not hold, from the pivot lane's
// registers (exact through step K-2: this runs before the update of step K-1 is applied, so one correction) --
// published for the broadcast reads of the step and stored as reflector vector K of the tile.
template <int K, bool PIVOT>
__device__ __forceinline__ void search_fetch(const double (&a)[WR], double* hl, LaneState& st, double* __restrict__ vtile,
                                             bool regs_current)
{
    const int j = st.j;
    bool ispiv;
    int lbl;
    if (PIVOT) {
        const int khi = __double2hiint(st.nu2);
        const int mh = half32_max_i32_fused(khi);
        unsigned long long pm = __builtin_amdgcn_uicmp(khi, mh, 32 /* ICMP_EQ */);
        ispiv = khi == mh;
        unsigned tlo = (unsigned)pm, thi = (unsigned)(pm >> 32);
        if (((tlo & (tlo - 1u)) | (thi & (thi - 1u))) != 0u) {
            ispiv = resolve_ties(K, st.lane, st.kstep, (unsigned)__double2loint(st.nu2), ispiv);
            pm = __builtin_amdgcn_ballot_w64(ispiv);
            tlo = (unsigned)pm; thi = (unsigned)(pm >> 32);
        }
        const int lA = __builtin_ctz(tlo), lB = __builtin_ctz(thi);
        lbl = (int)__builtin_amdgcn_ubfe((unsigned)(lA | (lB << 8)), (unsigned)st.sh8, 5u);
        st.livemask &= ~pm;
        st.nu2 = __hiloint2double(ispiv ? (int)0xBF800000 : khi, __double2loint(st.nu2));   // chosen: leaves the search
    } else {
        ispiv = j == K;
        lbl = K;
    }
    if (ispiv) { st.live = false; st.kstep = K; }
    st.ispiv = ispiv;
    st.lbl = lbl;

    constexpr int KR = K == 0 ? 0 : ((K - 1) / RB) * RB;
    double xi = hl[L_IMG + lbl * LDI + (j >= R0 ? j - R0 : 0)];
#pragma unroll
    for (int m = KR; m < K; ++m) xi = fma(hl[L_WBUF + (m % RB) * WR + lbl], st.h[m % RB], xi);
    if (K < R0) {
        const int src = (lbl << 2) + st.hb4;
        double xr = 0.0;
#pragma unroll
        for (int i = K; i < R0; ++i) {
            const double tv = bpermute_f64(src, a[i]);
            if (j == i) xr = tv;
        }
        // (regs_current: the rare norm-recompute path of step K-1 has applied its update already)
        if (K >= 1 && !regs_current) xr = fma(hl[L_WBUF + ((K + RB - 1) % RB) * WR + lbl], st.h[(K + RB - 1) % RB], xr);
        if (j < R0) xi = xr;
    }
    st.h[K % RB] = xi;
    hl[L_XBUF + j] = xi;
    if (vtile) vtile[K * WR + j] = xi;
}

template <int K, bool PIVOT>
__device__ __forceinline__ void factor_step(double (&a)[WR], double* hl, LaneState& st, double* __restrict__ vtile,
                                            double* __restrict__ stile, double* __restrict__ hcoeffs_tile)
{
    const int j = st.j;
    const bool ispiv = st.ispiv;
    const int lbl = st.lbl;
    const double ak = a[K];
    const double xk = hl[L_XBUF + K];
    double x[WR];
#pragma unroll
    for (int i = K + 1; i < WR; ++i) x[i] = hl[L_XBUF + i];
    double dA = 0.0;
#pragma unroll
    for (int i = K + 1; i < WR; ++i) {
        if (i == K + 1) dA = x[i] * a[i];
        else dA = fma(x[i], a[i], dA);
    }
    // makeHouseholder + applyHouseholderOnTheLeft, un-normalised (see bdqr_pair.hip): nb = -beta, s = x0 - beta = -w,
    // ng = -1/(beta w); tau = w/beta; for a column c with tail dot d: gamma = (d - w c_k)/(beta w).
    const double tailSq = bpermute_f64((lbl << 2) + st.hb4, dA);
    const double nrm = sqrt_pos(fma(xk, xk, tailSq));
    double nb = __hiloint2double((__double2hiint(nrm) & 0x7fffffff) | (__double2hiint(xk + 0.0) & (int)0x80000000),
                                 __double2loint(nrm));
    double s = nb + xk;
    double ng = -recip(nb * s);
    const unsigned long long dm = __builtin_amdgcn_fcmp(tailSq, DBL_MIN, 13 /* FCMP_ULE */);
    bool setdiag = ispiv;
    if (__builtin_expect(dm != 0ull, 0)) {
        asm volatile("");
        if (!(tailSq > DBL_MIN)) { ng = 0.0; s = 0.0; setdiag = false; }   // tau = 0, beta = x0, H = I
    }
    if (hcoeffs_tile && ispiv) hcoeffs_tile[K] = -(s * s) * ng;
    if (stile && j == 0) { stile[2 * K] = s; stile[2 * K + 1] = ng; }
    const double ngA = fma(s, ak, dA) * ng;
    double an = fma(s, ngA, ak);
    if (setdiag) an = -nb;                       // R(k,k) = beta
    a[K] = an;                                   // row K of R is final; it stays in the register
    hl[L_WBUF + (K % RB) * WR + j] = ngA;

    bool updated = false;
    if (PIVOT && K + 1 < WR) {
        const double nn = fma(-an, an, st.nu2);  // LAWN-176 downdate, squared form
        st.nu2 = nn;
        const unsigned long long nm = __builtin_amdgcn_fcmp(nn, st.thr_nd2, 5 /* FCMP_OLE */) & st.livemask;
        if (__builtin_expect(nm != 0ull, 0)) {
            asm volatile("");
#pragma unroll
            for (int i = K + 1; i < WR; ++i) a[i] = fma(ngA, x[i], a[i]);
            updated = true;
            const bool need = st.live && nn <= st.thr_nd2;
            double sq = 0.0;
#pragma unroll
            for (int i = K + 1; i < WR; ++i) sq = fma(a[i], a[i], sq);
            if (need) { st.nu2 = sq; st.thr_nd2 = sq * SQRT_EPS; }
        }
    }
    if (K + 1 < WR) search_fetch<(K + 1 < WR ? K + 1 : K), PIVOT>(a, hl, st, vtile, updated);
    if (!updated) {
#pragma unroll
        for (int i = K + 1; i < WR; ++i) a[i] = fma(ngA, x[i], a[i]);
    }
    if (K % RB == RB - 1 && K + 1 < WR) {
        if (st.live) {
#pragma unroll
            for (int i = (K + 1 > R0 ? K + 1 : R0); i < WR; ++i) hl[L_IMG + j * LDI + (i - R0)] = a[i];
        }
    }
}

}  // namespace split

// Factorisation of A: persistent workgroups of one wave, two tiles per wave.
template <bool PIVOT>
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(QRK_SPLIT_WAVES, QRK_SPLIT_WAVES)))
bdqr_factor32_kernel(int64_t num_tiles, const double* __restrict__ tiles, double* __restrict__ vbuf,
                     double* __restrict__ sbuf, double* __restrict__ r_vals, int32_t* __restrict__ perm,
                     double* __restrict__ hcoeffs)
{
    using namespace split;
    __shared__ __attribute__((aligned(16))) double lds[2 * L_HALF];
    const int64_t npairs = (num_tiles + 1) / 2;
    for (int64_t pi = blockIdx.x; pi < npairs; pi += gridDim.x) {
        int lane = threadIdx.x;
        asm volatile("" : "+v"(lane));
        const int half = lane >> 5, j = lane & 31;
        double* hl = lds + half * L_HALF;
        const int64_t t = 2 * pi + half;
        const bool valid = t < num_tiles;
        double a[WR];
        {
            // coalesced: lane j takes ROW j (8 bytes per column); the image area transposes in two passes because it
            // only has room for 26 rows: rows 0..25 first, then rows 6..31, which is the image the steps use
            double row[WR];
            if (valid) {
                const double* src = tiles + t * 1024 + j;
#pragma unroll
                for (int m = 0; m < WR; ++m) row[m] = src[32 * m];
            } else {
                // missing partner of an odd last tile: diag(64..33), nothing of it is stored
#pragma unroll
                for (int m = 0; m < WR; ++m) row[m] = (m == j) ? (double)(64 - j) : 0.0;
            }
            if (j < WR - R0) {
#pragma unroll
                for (int m = 0; m < WR; ++m) hl[L_IMG + m * LDI + j] = row[m];
            }
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int i = 0; i < WR - R0; ++i) a[i] = hl[L_IMG + j * LDI + i];
            __builtin_amdgcn_wave_barrier();
            if (j >= R0) {
#pragma unroll
                for (int m = 0; m < WR; ++m) hl[L_IMG + m * LDI + (j - R0)] = row[m];
            }
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int i = WR - R0; i < WR; ++i) a[i] = hl[L_IMG + j * LDI + (i - R0)];
        }

        LaneState st;
        st.lane = lane; st.j = j; st.kstep = 64;
        st.sh8 = half * 8; st.hb4 = half * 128;
        st.live = true;
        st.livemask = ~0ull;
#pragma unroll
        for (int m = 0; m < RB; ++m) st.h[m] = 0.0;
        {
            double s = 0.0;
#pragma unroll
            for (int i = 0; i < WR; ++i) s = fma(a[i], a[i], s);
            st.nu2 = s;
            st.thr_nd2 = s * SQRT_EPS;
        }
#ifdef QRK_SPLIT_NOV     // diagnostic: no reflector stores (results of the Q kernel are then garbage)
        double* vtile = nullptr;
        double* stile = nullptr;
#else
        double* vtile = valid ? vbuf + t * V_TILE : nullptr;
        double* stile = valid ? sbuf + t * S_TILE : nullptr;
#endif
        double* hc_tile = (hcoeffs && valid) ? hcoeffs + t * 32 : nullptr;

        search_fetch<0, PIVOT>(a, hl, st, vtile, false);
#define QRK_STEP(K) factor_step<K, PIVOT>(a, hl, st, vtile, stile, hc_tile);
        QRK_0_31(QRK_STEP)
#undef QRK_STEP

        // ---- R: the column chosen at step k ends at position k and holds R(0..k, k) in its registers; packed upper
        // triangle by columns = CSC value order of m_R (BlockDiagonalSparseQR.h:475-479), staged through the image area
        {
            int ln = threadIdx.x;
            asm volatile("" : "+v"(ln));
            const int jj = ln & 31;
            double* h2 = lds + (ln >> 5) * L_HALF;
            const int64_t tt = 2 * pi + (ln >> 5);
            const bool ok = tt < num_tiles;
            const int ks = st.kstep;
            __builtin_amdgcn_wave_barrier();
            const int base = ks * (ks + 1) / 2;
#pragma unroll
            for (int i = 0; i < WR; ++i)
                if (i <= ks) h2[base + i] = a[i];
            if (ok) perm[tt * 32 + ks] = (int32_t)(tt * 32 + jj);      // m_outputPerm_c.indices() (:519-521)
            __builtin_amdgcn_wave_barrier();
            if (ok) {
                double2* dst = reinterpret_cast<double2*>(r_vals + tt * 528);
#pragma unroll
                for (int qq = 0; qq < 9; ++qq) {
                    const int e2 = jj + 32 * qq;
                    if (e2 < 264) dst[e2] = make_double2(h2[2 * e2], h2[2 * e2 + 1]);
                }
            }
            __builtin_amdgcn_wave_barrier();
        }
    }
}

// Q_i = H_0 ... H_31 applied to the identity, row j of Q_i (= column j of Q^T) in lane j, with the operations and the
// order of bdqr_pair.hip (d = x_tail . c_tail as a product followed by FMAs; gamma' = (s c_k + d) ng; c_k += s gamma';
// c_i += gamma' x_i).  The reflectors of a tile are staged in LDS (coalesced read), the steps only do broadcast reads.
__global__ void __launch_bounds__(64, 2)
bdqr_formq32_kernel(int64_t num_tiles, const double* __restrict__ vbuf, const double* __restrict__ sbuf,
                    double* __restrict__ q_vals)
{
    using namespace split;
    constexpr int LH = V_TILE + S_TILE;      // 1088 doubles per half
    __shared__ __attribute__((aligned(16))) double lds[2 * LH];
    const int64_t npairs = (num_tiles + 1) / 2;
    for (int64_t pi = blockIdx.x; pi < npairs; pi += gridDim.x) {
        int lane = threadIdx.x;
        asm volatile("" : "+v"(lane));
        const int half = lane >> 5, j = lane & 31;
        double* hl = lds + half * LH;
        const int64_t t = 2 * pi + half;
        const bool valid = t < num_tiles;
        __builtin_amdgcn_wave_barrier();
        if (valid) {
            const double2* src = reinterpret_cast<const double2*>(vbuf + t * V_TILE);
#pragma unroll
            for (int qq = 0; qq < 16; ++qq)
                *reinterpret_cast<double2*>(&hl[2 * (j + 32 * qq)]) = src[j + 32 * qq];
            const double2 sc = reinterpret_cast<const double2*>(sbuf + t * S_TILE)[j];
            *reinterpret_cast<double2*>(&hl[V_TILE + 2 * j]) = sc;
        } else {
#pragma unroll
            for (int qq = 0; qq < 16; ++qq) *reinterpret_cast<double2*>(&hl[2 * (j + 32 * qq)]) = make_double2(0.0, 0.0);
            *reinterpret_cast<double2*>(&hl[V_TILE + 2 * j]) = make_double2(0.0, 0.0);
        }
        __builtin_amdgcn_wave_barrier();
        double q[WR];
#pragma unroll
        for (int i = 0; i < WR; ++i) q[i] = (i == j) ? 1.0 : 0.0;
#define QRK_QSTEP(K)                                                                             \
        {                                                                                        \
            const double s = hl[V_TILE + 2 * K], ng = hl[V_TILE + 2 * K + 1];                    \
            double dQ = 0.0;                                                                     \
            _Pragma("unroll")                                                                    \
            for (int i = K + 1; i < WR; ++i) {                                                   \
                const double xi = hl[K * WR + i];                                                \
                if (i == K + 1) dQ = xi * q[i]; else dQ = fma(xi, q[i], dQ);                     \
            }                                                                                    \
            const double ngQ = fma(s, q[K], dQ) * ng;                                            \
            q[K] = fma(s, ngQ, q[K]);                                                            \
            _Pragma("unroll")                                                                    \
            for (int i = K + 1; i < WR; ++i) q[i] = fma(ngQ, hl[K * WR + i], q[i]);              \
        }
        QRK_0_31(QRK_QSTEP)
#undef QRK_QSTEP
        if (valid) {
            double2* dst = reinterpret_cast<double2*>(q_vals + t * 1024 + j * 32);
#pragma unroll
            for (int m = 0; m < 16; ++m) dst[m] = make_double2(q[2 * m], q[2 * m + 1]);
        }
    }
}

void launch_bdqr_split32(int64_t num_tiles, int pivoting, const double* tiles, double* vbuf, double* sbuf, double* q_vals,
                         double* r_vals, int32_t* perm, double* hcoeffs, int num_cus, hipStream_t stream)
{
    if (num_tiles <= 0) return;
    const int64_t npairs = (num_tiles + 1) / 2;
    int wgs_a = QRK_SPLIT_WAVES >= 3 ? 10 : 8;
    if (const char* e = std::getenv("QRK_SPLIT_WGS")) { const int v = std::atoi(e); if (v > 0) wgs_a = v; }
    const int64_t slotsA = (int64_t)num_cus * wgs_a, slotsB = (int64_t)num_cus * 9;
    const dim3 block(64);
    const dim3 gridA((unsigned)(npairs < slotsA ? npairs : slotsA)), gridB((unsigned)(npairs < slotsB ? npairs : slotsB));
    if (pivoting)
        hipLaunchKernelGGL(bdqr_factor32_kernel<true>, gridA, block, 0, stream, num_tiles, tiles, vbuf, sbuf, r_vals, perm, hcoeffs);
    else
        hipLaunchKernelGGL(bdqr_factor32_kernel<false>, gridA, block, 0, stream, num_tiles, tiles, vbuf, sbuf, r_vals, perm, hcoeffs);
    hipLaunchKernelGGL(bdqr_formq32_kernel, gridB, block, 0, stream, num_tiles, vbuf, sbuf, q_vals);
}

}  // namespace qrk
