// bdqr_pair.hip -- one wavefront factorises TWO tiles (rows, cols <= 32) of a block-diagonal
// matrix at once: A_i P_i = Q_i R_i with explicit Q_i, for gfx950.
//
// Replaces the body of the hot loop of QRKit::BlockDiagonalSparseQR::factorize
// (src/QRKit/BlockDiagonalSparseQR.h:432-526): blockSolver.compute(block) (:437-438,
// Eigen ColPivHouseholderQR / HouseholderQR), Qi = blockSolver.matrixQ() (:446), the
// Q / R value assembly (:455-500) and the column-permutation splice (:519-521).
//
// Why two tiles per wave: the per-step work that is uniform per tile (pivot search, the reflector
// scalars with their sqrt and reciprocal, the norm downdate) is executed by every lane of a SIMD
// instruction anyway; the single-tile kernel (bdqr_wave.hip) spends ~80 % of its VALU cycles there.
// Giving each half of the wave its own tile makes every one of those instructions serve two tiles.
//
// Mapping (wave64): lane = 32*h + j.  Half h owns tile 2*pair + h; lane j of the half owns column j
// of A (32 row registers a[], zero padded) AND column j of Q^T (= row j of Q, registers q[]),
// starting from the identity.  Reflector k is the same operation on both, c <- c - gamma (x^T c ..),
// so A -> R and I -> Q^T advance together as FP64 FMA chains over the row registers.
//
// The pivot column has to reach all 32 lanes of its half.  v_readlane costs ~7 SIMD cycles per dword
// on gfx950 and ds_bpermute ~8 ns, but an LDS read of one address per half is almost free
// (tools/ubench2.hip).  So each half keeps a column-major image of its A in LDS (the staging image
// of the load, refreshed by the owning lanes every RB-th step with full-wave stores); lane j fetches
// element (j, pivot) of the image -- consecutive lanes, consecutive addresses -- applies the < RB
// rank-1 corrections since the last refresh from its own registers and publishes the result as a
// 32-double vector that the half then reads by broadcast.  No barrier inside the factorisation
// (one wave, in-order LDS queue).
//
// Columns are never physically swapped: each lane tracks the current position of its A column
// (Eigen's m_colsTranspositions bookkeeping), so the "first maximum" tie rule and the final
// permutation are those of Eigen's ColPivHouseholderQR.  Row k of R is final after step k and is
// parked in the LDS slot of the pivot column (dead from then on); the epilogue gathers the packed
// upper triangle through the permutation.  All global accesses are coalesced 16-B accesses via LDS.
#include "qrk_device.h"

#include <float.h>

namespace qrk {

namespace pair {

constexpr int WR = 32;               // row registers per column
constexpr int LDP = WR + 2;          // LDS column stride in doubles: 272 B, conflict-free b64/b128 access
constexpr int RB = 4;                // the LDS image of A is refreshed every RB steps
// LDS carve-up per HALF (doubles)
constexpr int L_IMG = 0;             // [32][LDP] column-major image of A / staging for Q; R rows parked here
constexpr int L_XBUF = WR * LDP;     // [32] current pivot column
constexpr int L_WBUF = L_XBUF + WR;  // [RB][32] update coefficients of the last RB steps, per A column
constexpr int L_POS = L_WBUF + RB * WR;   // [32] int: lane_of_pos
constexpr int L_HALF = L_POS + WR / 2;    // 1264 doubles = 10112 B per half, 20224 B per wave -> 8 waves per CU

constexpr double SQRT_EPS = 1.4901161193847656e-08;   // sqrt(DBL_EPSILON), Eigen's norm_downdate_threshold

#define QRK_0_31(M)                                                                              \
    M(0) M(1) M(2) M(3) M(4) M(5) M(6) M(7) M(8) M(9) M(10) M(11) M(12) M(13) M(14) M(15) M(16)  \
    M(17) M(18) M(19) M(20) M(21) M(22) M(23) M(24) M(25) M(26) M(27) M(28) M(29) M(30) M(31)

// sqrt(x) for a positive normal x: v_rsq_f64 seed (2^-24), one Goldschmidt iteration and one residual
// correction: <= 1 ulp (tools/ubench3.hip), no FP64 division.
__device__ __forceinline__ double sqrt_pos(double x)
{
    const double y = __builtin_amdgcn_rsq(x);
    double g = x * y, h = 0.5 * y;
    const double e = fma(-h, g, 0.5);
    g = fma(g, e, g);
    h = fma(h, e, h);
    const double d = fma(-g, g, x);
    return fma(d, h, g);
}

// 1/x by v_rcp_f64 + two Newton steps (<= 1 ulp), no v_div_* sequence.
__device__ __forceinline__ double recip(double x)
{
    double y = __builtin_amdgcn_rcp(x);
    double e = fma(-x, y, 1.0);
    y = fma(y, e, y);
    e = fma(-x, y, 1.0);
    y = fma(y, e, y);
    return y;
}

// Two FMAs that share one operand, issued back to back.  Written as one asm statement so that
// hipcc cannot separate them (it otherwise defers one accumulation chain and parks the shared
// pivot-column values in scratch).  d0 += x*c0; d1 += x*c1.
__device__ __forceinline__ void fmac2_shared_a(double& d0, double& d1, double x, double c0, double c1)
{
    asm("v_fmac_f64_e32 %0, %2, %3\n\tv_fmac_f64_e32 %1, %2, %4" : "+v"(d0), "+v"(d1) : "v"(x), "v"(c0), "v"(c1));
}
// d0 = x*c0; d1 = x*c1 (starts the two chains without zeroing accumulators).
__device__ __forceinline__ void mul2_shared_a(double& d0, double& d1, double x, double c0, double c1)
{
    asm("v_mul_f64 %0, %2, %3\n\tv_mul_f64 %1, %2, %4" : "=&v"(d0), "=&v"(d1) : "v"(x), "v"(c0), "v"(c1));
}
// c0 += n0*x; c1 += n1*x.
__device__ __forceinline__ void fmac2_shared_b(double& c0, double& c1, double n0, double n1, double x)
{
    asm("v_fmac_f64_e32 %0, %2, %4\n\tv_fmac_f64_e32 %1, %3, %4" : "+v"(c0), "+v"(c1) : "v"(n0), "v"(n1), "v"(x));
}

// Inverse of e = p(p+1)/2 + i (0 <= i <= p): position in the packed upper triangle by columns.
__device__ __forceinline__ void tri_unpack(int e, int& p, int& i)
{
    int q = (int)((sqrtf(8.0f * (float)e + 1.0f) - 1.0f) * 0.5f);
    if ((q + 1) * (q + 2) / 2 <= e) ++q;
    if (q * (q + 1) / 2 > e) --q;
    p = q;
    i = e - q * (q + 1) / 2;
}

// Per-lane state that lives across the steps.
struct LaneState {
    int lane, j, half;
    int sh8;         // 8 * half: bit offset of this half's field in packed wave-uniform words
    int hb4;         // 128 * half: ds_bpermute byte address of lane 0 of this half
    bool live;       // this lane's A column is not yet chosen as a pivot
    unsigned long long livemask;   // the same as a wave-uniform lane mask
    int kstep;       // step at which this column was chosen (= its final position), 64 = not yet
    int rows, cols;  // tile shape of this half (rows/cols beyond are zero padding)
    double nu2;      // m_colNormsUpdated^2
    double thr_nd2;  // sqrt(eps) * m_colNormsDirect^2
    double h[RB];    // entry j of the pivot columns of the last RB steps
};

__device__ __forceinline__ unsigned long long ballot64(bool p) { return __builtin_amdgcn_ballot_w64(p); }

__device__ __forceinline__ double bpermute_f64(int byte_addr, double v)
{
    const int lo = __builtin_amdgcn_ds_bpermute(byte_addr, __double2loint(v));
    const int hi = __builtin_amdgcn_ds_bpermute(byte_addr, __double2hiint(v));
    return __hiloint2double(hi, lo);
}

// Rare path of the pivot search: several live columns share the high word of the largest squared
// norm.  Compare the low words, then apply Eigen's first-maximum rule = smallest CURRENT position
// among exact ties.  Positions are not tracked in the hot path (the final position of a column is
// simply the step that chose it); here they are rebuilt by replaying the column transpositions of
// steps 0..K-1 (ColPivHouseholderQR.h: m_qr.col(k).swap(m_qr.col(biggest_col_index))).
// Returns a one-hot (per half) pivot flag.
__device__ __forceinline__ bool resolve_ties(int K, int lane, int kstep, unsigned klo, bool cand)
{
    const int half = lane >> 5;
    const unsigned ml = half32_max_u32(cand ? klo : 0u);
    cand = cand && klo == ml;
    int p = lane & 31;
#pragma unroll 1
    for (int k = 0; k < K; ++k) {
        const unsigned long long m = ballot64(kstep == k);
        const unsigned mh = half ? (unsigned)(m >> 32) : (unsigned)m;
        const int l = (mh ? __ffs((int)mh) - 1 : 0) + 32 * half;
        const int pl = __builtin_amdgcn_ds_bpermute(l * 4, p);
        if (mh) p = (lane == l) ? k : (p == k ? pl : p);
    }
    const int pc = cand ? p : 64;
    const int pmin = half32_min_i32(pc);
    return pc == pmin && pc != 64;
}

// One step of ColPivHouseholderQR::computeInPlace (Eigen/src/QR/ColPivHouseholderQR.h) on both
// wave-resident tiles: pivot search, reflector, trailing update of A and of Q^T, norm downdate.
//
// Column norms are tracked SQUARED: Eigen's  temp = (1+t)(1-t), t = |a_kj|/normUpd;
// normUpd *= sqrt(temp)  is  nu2 <- max(nu2 - a_kj^2, 0), and its recompute test
// temp (normUpd/normDir)^2 <= sqrt(eps)  is  nu2_new <= sqrt(eps) normDir^2: the same quantities
// without two FP64 divisions and a square root per step; the first maximum is the same column
// because squaring is monotone.
//
// The kernel is bound by the number of VALU instructions a wave issues (two waves per SIMD), so the
// step is written to keep everything that is uniform per half out of VGPR selects: the pivot lane is
// a lane mask, its index reaches the lanes as one bit-field extract of a packed scalar, and |tail|^2
// comes from the pivot lane by ds_bpermute.  Columns that were already chosen are NOT masked out of
// the arithmetic: nothing below the diagonal of R is ever read, so their lanes compute garbage.
template <int K, bool FULL32, bool PIVOT, bool HC>
__device__ __forceinline__ void pair_step(double (&a)[WR], double (&q)[WR], double* hl /* this half's LDS */,
                                          LaneState& st, double* __restrict__ hcoeffs_tile)
{
    const int j = st.j;
    // this half still has a column to eliminate at step K (always, for 32x32 tiles)
    const bool act = FULL32 ? true : K < st.cols;
    bool ispiv;      // this lane's column is the pivot of step K
    int lbl;         // pivot lane of this lane's half, 0..31
    if (PIVOT) {
        // first maximum of the updated norms over the live columns, per half.  Non-negative doubles
        // order like their bit patterns, so the max is an integer max on the high word (chosen
        // columns carry a negative norm, see below) ...
        const int khi = __double2hiint(st.nu2);
        const int mh = half32_max_i32_fused(khi);
        unsigned long long pm;
        if (FULL32) {
            pm = __builtin_amdgcn_uicmp(khi, mh, 32 /* ICMP_EQ */);   // a live column exists: mh >= 0
            ispiv = khi == mh;
        } else {
            ispiv = st.live && khi == mh;
            pm = ballot64(ispiv);
        }
        unsigned tlo = (unsigned)pm, thi = (unsigned)(pm >> 32);
        if (((tlo & (tlo - 1u)) | (thi & (thi - 1u))) != 0u) {
            // ... unless several columns share it
            ispiv = resolve_ties(K, st.lane, st.kstep, (unsigned)__double2loint(st.nu2), ispiv);
            pm = ballot64(ispiv);
            tlo = (unsigned)pm; thi = (unsigned)(pm >> 32);
        }
        const int lA = FULL32 ? __builtin_ctz(tlo) : (tlo ? __builtin_ctz(tlo) : 0);
        const int lB = FULL32 ? __builtin_ctz(thi) : (thi ? __builtin_ctz(thi) : 0);
        lbl = (int)__builtin_amdgcn_ubfe((unsigned)(lA | (lB << 8)), (unsigned)st.sh8, 5u);
        st.livemask &= ~pm;
        // a chosen column leaves the search: negative "norm" (it only decreases from here on)
        st.nu2 = __hiloint2double(ispiv ? (int)0xBF800000 : khi, __double2loint(st.nu2));
    } else {
        ispiv = act && j == K;   // HouseholderQR: column K
        lbl = K;
    }
    if (ispiv) { st.live = false; st.kstep = K; }

    // ---- pivot column: element (j, lbl) of the LDS image (exact through step K0-1) plus the rank-1
    // corrections of steps K0..K-1, then published for broadcast reads.
    constexpr int K0 = (K / RB) * RB;
    {
        double xi = hl[L_IMG + lbl * LDP + j];
#pragma unroll
        for (int m = K0; m < K; ++m) xi = fma(hl[L_WBUF + (m % RB) * WR + lbl], st.h[m % RB], xi);
        if (!FULL32) xi = (act && j < st.rows) ? xi : 0.0;
        st.h[K % RB] = xi;
        hl[L_XBUF + j] = xi;
    }

    // ---- d = x_tail^T c_tail for the A column and the Q^T column (pivot lane: dA = |x_tail|^2)
    const double ak = a[K], qk = q[K];
    const double xk = hl[L_XBUF + K];
    double dA = 0.0, dQ = 0.0;
    if (K + 1 < WR) mul2_shared_a(dA, dQ, hl[L_XBUF + K + 1], a[K + 1 < WR ? K + 1 : 0], q[K + 1 < WR ? K + 1 : 0]);
#pragma unroll
    for (int i = K + 2; i < WR; ++i) fmac2_shared_a(dA, dQ, hl[L_XBUF + i], a[i], q[i]);   // broadcast reads

    // ---- makeHouseholder + applyHouseholderOnTheLeft (Eigen/src/Householder/Householder.h) in the
    // un-normalised form: with beta = -sign(x0) sqrt(x0^2 + |tail|^2) and w = beta - x0,
    //   tau = w/beta, essential = tail/(x0 - beta) = -tail/w, and for a column c with tail dot d
    //   gamma = (d - w c_k) / (beta w):   c_k <- c_k + w gamma  (= c_k - tau tmp),
    //                                     c_i <- c_i - gamma x_i (= c_i - tau ess_i tmp),
    // which needs one square root and one reciprocal (of beta*w > 0) per step and no division.
    // Kept here: nb = -beta = copysign(norm, x0), s = -w = nb + x0, ng = -1/(beta w).
    const double tailSq = bpermute_f64((lbl << 2) + st.hb4, dA);
    const double nrm = sqrt_pos(fma(xk, xk, tailSq));
    // Eigen: if (c0 >= 0) beta = -beta; -0.0 counts as >= 0, hence the + 0.0
    double nb = __hiloint2double((__double2hiint(nrm) & 0x7fffffff) | (__double2hiint(xk + 0.0) & (int)0x80000000),
                                 __double2loint(nrm));
    double s = nb + xk;
    double ng = -recip(nb * s);
    // Eigen: tailSqNorm <= min() gives tau = 0, beta = x0, H = I.  Rare, so a real branch (the empty
    // asm keeps hipcc from flattening it into selects); s = 0 leaves c_k = x0 in the pivot lane.
    const bool degen = !act || !(tailSq > DBL_MIN);
    const unsigned long long dm = FULL32 ? __builtin_amdgcn_fcmp(tailSq, DBL_MIN, 13 /* FCMP_ULE */) : ballot64(degen);
    bool setdiag = ispiv;
    if (__builtin_expect(dm != 0ull, 0)) {
        asm volatile("");
        if (degen) { ng = 0.0; s = 0.0; setdiag = false; }   // (nrm may be NaN here: rsq(0) = inf)
    }
    if (HC) {
        if (hcoeffs_tile && ispiv) hcoeffs_tile[K] = -(s * s) * ng;   // tau = w/beta = w^2/(beta w)
    }

    const double ngA = fma(s, ak, dA) * ng;      // -gamma for the A column
    double an = fma(s, ngA, ak);
    if (setdiag) an = -nb;                       // R(k,k) = beta
    const double ngQ = fma(s, qk, dQ) * ng;
    q[K] = fma(s, ngQ, qk);
    hl[L_WBUF + (K % RB) * WR + j] = ngA;
#pragma unroll
    for (int i = K + 1; i < WR; ++i) fmac2_shared_b(a[i], q[i], ngA, ngQ, hl[L_XBUF + i]);

    // Row K of R is final: park it in the LDS slot of the pivot column (never read as a column again).
    if (FULL32 || act) hl[L_IMG + lbl * LDP + j] = an;

    // ---- refresh the LDS image of the live columns after every RB-th step
    if (K % RB == RB - 1 && K + 1 < WR) {
        if (st.live) {
#pragma unroll
            for (int i = K + 1; i < WR; ++i) hl[L_IMG + j * LDP + i] = a[i];
        }
    }

    // ---- LAWN-176 norm downdate for the remaining columns (squared form, see above)
    // No clamp at zero: a negative value is <= the threshold and is recomputed exactly.
    if (PIVOT && K + 1 < WR) {
        const double nn = fma(-an, an, st.nu2);
        st.nu2 = nn;
        const unsigned long long nm = __builtin_amdgcn_fcmp(nn, st.thr_nd2, 5 /* FCMP_OLE */) & st.livemask;
        if (__builtin_expect(nm != 0ull, 0)) {
            asm volatile("");
            const bool need = st.live && nn <= st.thr_nd2;
            double sq = 0.0;
#pragma unroll
            for (int i = K + 1; i < WR; ++i) sq = fma(a[i], a[i], sq);
            if (need) { st.nu2 = sq; st.thr_nd2 = sq * SQRT_EPS; }
        }
    }
}

}  // namespace pair

// FULL32: every tile is 32x32 and all arrays are 16-byte aligned (uniform batch).
// PIVOT: ColPivHouseholderQR (else HouseholderQR).  HC: also emit the Householder coefficients.
template <bool FULL32, bool PIVOT, bool HC>
__global__ void __launch_bounds__(64, 2)
bdqr_pair_kernel(WaveBatch nb, const double* __restrict__ tiles, double* __restrict__ q_vals,
                 double* __restrict__ r_vals, int32_t* __restrict__ perm,
                 double* __restrict__ hcoeffs)
{
    using namespace pair;
    __shared__ __attribute__((aligned(16))) double lds[2 * L_HALF];
    // One pair per workgroup (no grid-stride loop: a loop makes hipcc hoist per-lane addresses out of
    // it and keep them in scratch across the whole factorisation).
    {
        const int64_t pi = blockIdx.x;
        const int lane = threadIdx.x;
        const int half = lane >> 5, j = lane & 31;
        double* hl = lds + half * L_HALF;
        const int64_t t = 2 * pi + half;
        const bool valid = t < nb.num_tiles;
        int r, c, cbase;
        int64_t toff, qoff, roff;
        if (FULL32) {
            r = 32; c = 32;
            toff = t * 1024; qoff = t * 1024; roff = t * 528; cbase = (int)(t * 32);
        } else if (nb.tile_ids) {
            const int gidx = nb.tile_ids[valid ? t : nb.num_tiles - 1];
            r = nb.t_rows[gidx]; c = nb.t_cols[gidx];
            toff = nb.t_off[gidx]; qoff = nb.q_off[gidx]; roff = nb.r_off[gidx]; cbase = nb.c_off[gidx];
        } else {
            r = nb.rows; c = nb.cols;
            toff = t * r * c; qoff = t * (int64_t)r * r; roff = t * (int64_t)(c * (c + 1) / 2);
            cbase = (int)(t * c);
        }
        if (!valid) { r = 0; c = 0; }

        // ---- stage the tiles: coalesced global read (each half its own tile), padded LDS columns
        if (FULL32) {
            if (valid) {
                const double2* src = reinterpret_cast<const double2*>(tiles + toff);
                for (int q0 = 0; q0 < 16; q0 += 8) {   // two batches of eight 16-B loads per lane
#pragma unroll
                    for (int qq = 0; qq < 8; ++qq) {
                        const int e2 = j + 32 * (q0 + qq);    // double2 index, 16 per column
                        const double2 v = src[e2];
                        *reinterpret_cast<double2*>(&hl[L_IMG + (e2 >> 4) * LDP + ((e2 & 15) << 1)]) = v;
                    }
                }
            } else {
                // missing partner of an odd last tile: diag(64..33) -- distinct norms, no tie-breaking;
                // nothing of it is stored
#pragma unroll 4
                for (int i = 0; i < WR; ++i) hl[L_IMG + j * LDP + i] = (i == j) ? (double)(64 - j) : 0.0;
            }
        } else {
            const double* src = tiles + toff;
            const int n_in = r * c;
            for (int e = j; e < n_in; e += 32) {
                const int cc = e / r;
                hl[L_IMG + cc * LDP + (e - cc * r)] = src[e];
            }
        }
        __syncthreads();

        double a[WR], q[WR];
#pragma unroll
        for (int i = 0; i < WR; ++i) {
            if (FULL32) a[i] = hl[L_IMG + j * LDP + i];
            else a[i] = (j < c && i < r) ? hl[L_IMG + j * LDP + i] : 0.0;
            q[i] = (i == j && j < r) ? 1.0 : 0.0;
        }
        // (no barrier: the image stays valid, it is the source of the pivot columns)

        LaneState st;
        st.lane = lane; st.j = j; st.half = half; st.kstep = 64; st.rows = r; st.cols = c;
        st.sh8 = half * 8; st.hb4 = half * 128;
        st.live = FULL32 ? true : j < c;
        st.livemask = FULL32 ? ~0ull : __builtin_amdgcn_ballot_w64(j < c);
#pragma unroll
        for (int m = 0; m < RB; ++m) st.h[m] = 0.0;
        {
            // squared column norms (ColPivHouseholderQR: m_colNormsUpdated^2, sqrt(eps) m_colNormsDirect^2)
            double s = 0.0;
#pragma unroll
            for (int i = 0; i < WR; ++i) s = fma(a[i], a[i], s);
            st.nu2 = s;
            st.thr_nd2 = s * SQRT_EPS;
        }
        double* hc_tile = (hcoeffs && valid) ? hcoeffs + cbase : nullptr;
        // number of steps = the larger column count of the two tiles
        const int c0 = __builtin_amdgcn_readlane(c, 0), c1 = __builtin_amdgcn_readlane(c, 32);
        const int cmax = c0 > c1 ? c0 : c1;

        // The k loop is expanded by the preprocessor: every row-register index is a compile-time
        // constant.  (A rolled loop dispatching through a uniform switch makes hipcc's CFG
        // structurizer copy the whole register tile at every merge point.)
#define QRK_STEP(K) if (FULL32 || K < cmax) pair_step<K, FULL32, PIVOT, HC>(a, q, hl, st, hc_tile);
        QRK_0_31(QRK_STEP)
#undef QRK_STEP

        // ---- R: row i of R sits in the LDS slot of the column chosen at step i, indexed by ORIGINAL
        // column.  The packed upper triangle by columns is exactly the CSC value order of m_R
        // (BlockDiagonalSparseQR.h:475-479): element e -> (column p, row i), gathered through lane_of_pos.
        int* lane_of_pos = reinterpret_cast<int*>(&hl[L_POS]);
        if (j < c) {
            // the column chosen at step k ends at position k
            lane_of_pos[st.kstep] = j;
            perm[cbase + st.kstep] = cbase + j;   // m_outputPerm_c.indices()(base_col+j) (:519-521)
        }
        __syncthreads();
        if (FULL32) {
            if (valid) {
                double2* dst = reinterpret_cast<double2*>(r_vals + roff);
#pragma unroll
                for (int qq = 0; qq < 9; ++qq) {
                    const int e2 = j + 32 * qq;
                    if (e2 < 264) {
                        int p0, i0, p1, i1;
                        tri_unpack(2 * e2, p0, i0);
                        tri_unpack(2 * e2 + 1, p1, i1);
                        dst[e2] = make_double2(hl[L_IMG + lane_of_pos[i0] * LDP + lane_of_pos[p0]],
                                               hl[L_IMG + lane_of_pos[i1] * LDP + lane_of_pos[p1]]);
                    }
                }
            }
        } else {
            const int n_r = c * (c + 1) / 2;
            for (int e = j; e < n_r; e += 32) {
                int p0, i0;
                tri_unpack(e, p0, i0);
                r_vals[roff + e] = hl[L_IMG + lane_of_pos[i0] * LDP + lane_of_pos[p0]];
            }
        }
        __syncthreads();

        // ---- Q: lane j holds row j of Q_i; row-major rows are the CSR value order of m_Q in both
        // FullQ ([U|N] split, :455-471) and BlockDiagonalQ (:480-492) layouts.
        if (j < r) {
            if (FULL32) {
#pragma unroll
                for (int i = 0; i < WR; i += 2)
                    *reinterpret_cast<double2*>(&hl[L_IMG + j * LDP + i]) = make_double2(q[i], q[i + 1]);
            } else {
#pragma unroll
                for (int i = 0; i < WR; ++i)
                    if (i < r) hl[L_IMG + j * LDP + i] = q[i];
            }
        }
        __syncthreads();
        if (FULL32) {
            if (valid) {
                double2* dst = reinterpret_cast<double2*>(q_vals + qoff);
                for (int q0 = 0; q0 < 16; q0 += 8) {
#pragma unroll
                    for (int qq = 0; qq < 8; ++qq) {
                        const int e2 = j + 32 * (q0 + qq);
                        dst[e2] = *reinterpret_cast<const double2*>(&hl[L_IMG + (e2 >> 4) * LDP + ((e2 & 15) << 1)]);
                    }
                }
            }
        } else {
            const int n_q = r * r;
            for (int e = j; e < n_q; e += 32) {
                const int jj = e / r;
                q_vals[qoff + e] = hl[L_IMG + jj * LDP + (e - jj * r)];
            }
        }
        __syncthreads();
    }
}

void launch_bdqr_pair(const WaveBatch& nb, bool full32, const double* tiles, double* q_vals,
                      double* r_vals, int32_t* perm, double* hcoeffs, int max_blocks,
                      hipStream_t stream)
{
    if (nb.num_tiles <= 0) return;
    (void)max_blocks;
    const int64_t npairs = (nb.num_tiles + 1) / 2;
    const dim3 grid((unsigned)npairs), block(64);
#define QRK_LAUNCH(F, P, H)                                                                        \
    hipLaunchKernelGGL((bdqr_pair_kernel<F, P, H>), grid, block, 0, stream, nb, tiles, q_vals, r_vals, perm, hcoeffs)
    const bool piv = nb.pivoting != 0, hc = hcoeffs != nullptr;
    if (full32) {
        if (piv) { if (hc) QRK_LAUNCH(true, true, true); else QRK_LAUNCH(true, true, false); }
        else { if (hc) QRK_LAUNCH(true, false, true); else QRK_LAUNCH(true, false, false); }
    } else {
        if (piv) QRK_LAUNCH(false, true, true); else QRK_LAUNCH(false, false, true);
    }
#undef QRK_LAUNCH
}

}  // namespace qrk
