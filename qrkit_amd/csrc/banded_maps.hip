// banded_maps.hip -- the strips form of the banded solver (banded.hip, qrk_bbs_*): Q^T b, Q x and R^-1 y WITHOUT one workgroup walking
// every strip's factors, for gfx950.
//
// Reference seam: BandedBlockedSparseQR::_solve_impl (src/QRKit/BandedBlockedSparseQR.h:290-311) -- y = Q^T b as the product of the
// block reflectors in order (SparseBlockYTY_VecProduct, src/QRKit/SparseBlockYTY.h:100-139, with BlockYTY.h:152-172), then the
// triangular solve with R.  Both are chains from strip to strip, and bb_apply_q_kernel / bb_solve_r_kernel (banded.hip) walk them on ONE
// workgroup, which pulls every strip's Y and T (786 KB at the BASELINE configs[2] shape) and its rows of R through one CU, three
// dependent passes per strip: 52 + 19 us per strip, 3.5 s for the 50 000 strips of configs[2] on a chip whose other 255 CUs idle.
//
// What is sequential in those chains is small.  Panel i of stage B applies A_i = I + Y_i T_i^T Y_i^T to the stack
// [carry of panel i-1 interleaved with the first rows of the strip's own vector; the rest of it]; the only thing panel i + 1 needs from
// panel i is the carry -- lo = n - s numbers (128 at the configs[2] shape) -- and A_i is LINEAR:
//
//      carry_{i+1} = S_o A_i (E_c carry_i + E_y ya_i) = M_i carry_i + c_i,      M_i = S_o A_i E_c  (lo x lo),
//
// with E_c / E_y the placement of the carry / of the strip's vector in the stack and S_o the rows that leave as the next carry.  So:
//   1. bbs_carry_map_kernel    M_i of every panel, once per factorisation, all CUs (two small products per panel, 16 Mflop);
//   2. bbs_panel_apply_kernel  phase 1: every panel at once, carry = 0: c_i                                   (all CUs)
//   3. bbs_carry_chain_kernel  carry_{i+1} = M_i carry_i + c_i: ONE workgroup per right-hand side, 131 KB per strip instead of 786 KB,
//                              one pass instead of three, the next strip's M in flight while this one is summed;
//   4. bbs_panel_apply_kernel  phase 2: every panel at once with its true carry: everything but the carry rows  (all CUs)
// Q x runs the same maps transposed, from the last panel to the first (the carry a panel hands BACK is E_c^T A_i^T S_o^T = M_i^T).
//
// The triangular solve has the same shape: strip i owns rows [i s, i s + s) of R, x_i = D_i^-1 (y_i - U_i x_right) with x_right the next
// lo entries of x.  bbs_backsub_map_kernel forms G_i = D_i^-1 U_i (s x lo) once per factorisation; per solve h_i = D_i^-1 y_i for every
// strip at once (bbs_backsub_diag_kernel), then x_i = h_i - G_i x_right on one workgroup (bbs_backsub_chain_kernel: 65 KB per strip and
// no dependent divisions).  The last strip (all n rows of its triangle) goes through bb_solve_r_kernel as before.
//
// The results differ from the one-workgroup chains by rounding only (the same operators, associated differently); both forms are kept
// (QRK_BBS_MAPS=0 selects the old one) and tests/test_banded_strips_gpu.py compares them.
#include "banded_host.h"
#include "qrk_device.h"

#include <cstdlib>

namespace qrk {

namespace bbm {

// sum over the 64 lanes, the same value in every lane (banded.hip, bb_wave_sum_dpp)
__device__ __forceinline__ double wsum(double v)
{
    v += dpp_f64<0xB1>(v);
    v += dpp_f64<0x4E>(v);
    v += dpp_f64<0x141>(v);
    v += dpp_f64<0x140>(v);
    return (readlane_f64(v, 0) + readlane_f64(v, 16)) + (readlane_f64(v, 32) + readlane_f64(v, 48));
}

// orders LDS only: the global loads of the next step stay in flight across it (a __syncthreads() waits for them)
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

constexpr int PA_THREADS = 512;
constexpr int PA_WAVES = PA_THREADS / 64;
constexpr int CH_THREADS = 1024;

}  // namespace bbm

// ---- 2 / 4: one panel of the chain per workgroup -------------------------------------------------------------------------------------
// blockIdx.x = panel, blockIdx.y = right-hand side.  Layouts as in banded.hip (BBStrips): ya = [N][ms] stage-A vectors, full = the
// vector in the order [R part, cols | per strip: lo chain residuals (strips 1..) | ms - n stage-A residuals].
// carr: [nrhs][N][lo]; carr[p] is the carry that ENTERS panel p -- Q^T: from panel p - 1 (even stack rows below 2 lo), Q: from panel
// p + 1 (stack rows [solved, n)).  phase 1: the entering carry is taken as zero and only the leaving carry is written (added to by the
// chain kernel afterwards); phase 2: the entering carry is read and everything but the leaving carry is written.
__global__ void __launch_bounds__(bbm::PA_THREADS)
bbs_panel_apply_kernel(const BBPanel* __restrict__ panels, int num_panels, const double* __restrict__ y_vals, const double* __restrict__ t_vals,
                       int transpose, int phase, double* __restrict__ ya_all, int64_t ya_ld, double* __restrict__ full_all, int64_t full_ld,
                       double* __restrict__ carr_all, int ms, int s, int lo, int cols, int max_act, int max_n)
{
    using namespace bbm;
    extern __shared__ __attribute__((aligned(16))) double smem[];
    double* seg = smem;                      // [max_act]
    double* w1 = seg + max_act;              // [max_n]
    double* w2 = w1 + max_n;                 // [max_n]
    double* part = w2 + max_n;               // [PA_THREADS]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int pidx = blockIdx.x;
    const int64_t col = blockIdx.y;
    const BBPanel p = panels[pidx];
    const int m = p.act_rows, n = p.ncols, solved = p.solved;
    // panels that hand no carry on have nothing to do in phase 1
    if (phase == 1 && (transpose ? solved >= n : pidx == 0)) return;
    const double* Y = y_vals + p.y_off;
    const double* T = t_vals + p.t_off;
    double* yi = ya_all + col * ya_ld + (int64_t)pidx * ms;
    double* full = full_all + col * full_ld;
    double* carr = carr_all + col * (int64_t)num_panels * lo;
    const int64_t ro = (int64_t)cols + (int64_t)pidx * (ms - n) + (pidx >= 1 ? (int64_t)(pidx - 1) * lo : 0);
    const double* cin = carr + (int64_t)pidx * lo;

    // ---- the stack vector
    if (transpose) {
        for (int i = tid; i < m; i += PA_THREADS) {
            double val;
            if (pidx == 0) val = yi[i];
            else if (i < 2 * lo) val = (i & 1) ? yi[i >> 1] : (phase == 2 ? cin[i >> 1] : 0.0);
            else val = yi[i - lo];
            seg[i] = val;
        }
    } else {
        for (int i = tid; i < m; i += PA_THREADS) {
            double val;
            if (i < solved) val = full[(int64_t)s * pidx + i];
            else if (i < n) val = phase == 2 ? cin[i - solved] : 0.0;
            else val = full[ro + (i - n)];
            seg[i] = val;
        }
    }
    __syncthreads();
    // ---- w1 = Y^T seg (Y = unit-lower view of the panel, row-major m x n): thread = column x row group, coalesced rows
    const int CW = ((n + 63) / 64) * 64, RG = PA_THREADS / CW;
    {
        const int jl = tid % CW, g = tid / CW;
        double acc = 0.0;
        if (g < RG && jl < n) {
            constexpr int U = 8;
            for (int i = jl + 1 + g; i < m; i += U * RG) {
                double yv[U];
#pragma unroll
                for (int u = 0; u < U; ++u) { const int ii = i + u * RG; yv[u] = Y[(int64_t)(ii < m ? ii : m - 1) * n + jl]; }
#pragma unroll
                for (int u = 0; u < U; ++u) { const int ii = i + u * RG; if (ii < m) acc = fma(yv[u], seg[ii], acc); }
            }
        }
        if (g < RG) part[g * CW + jl] = acc;
        __syncthreads();
        if (tid < n) {
            double d = seg[tid];                 // the unit diagonal (m >= n)
            for (int q = 0; q < RG; ++q) d += part[q * CW + tid];
            w1[tid] = d;
        }
        __syncthreads();
    }
    // ---- w2 = T^T w1 (Q^T) or T w1 (Q); T upper triangular, column-major, negated by the factorisation
    if (transpose) {
        constexpr int RB = 4, LB = 4;            // four rows of T^T per wave in flight, n <= 256
        for (int i0 = wave; i0 < n; i0 += RB * PA_WAVES) {
            double tv[RB][LB];
#pragma unroll
            for (int rb = 0; rb < RB; ++rb) {
                const int i = i0 + rb * PA_WAVES;
#pragma unroll
                for (int q = 0; q < LB; ++q) { const int j = lane + 64 * q; tv[rb][q] = (i < n && j <= i) ? T[(int64_t)i * n + j] : 0.0; }
            }
#pragma unroll
            for (int rb = 0; rb < RB; ++rb) {
                const int i = i0 + rb * PA_WAVES;
                double d = 0.0;
#pragma unroll
                for (int q = 0; q < LB; ++q) { const int j = lane + 64 * q; if (j < n) d = fma(tv[rb][q], w1[j], d); }
                d = wsum(d);
                if (lane == 0 && i < n) w2[i] = d;
            }
        }
    } else {
        const int il = tid % CW, g = tid / CW;
        double acc = 0.0;
        if (g < RG && il < n) {
            constexpr int U = 8;
            for (int j = il + g; j < n; j += U * RG) {
                double tv[U];
#pragma unroll
                for (int u = 0; u < U; ++u) { const int jj = j + u * RG; tv[u] = T[(int64_t)(jj < n ? jj : n - 1) * n + il]; }
#pragma unroll
                for (int u = 0; u < U; ++u) { const int jj = j + u * RG; if (jj < n) acc = fma(tv[u], w1[jj], acc); }
            }
        }
        if (g < RG) part[g * CW + il] = acc;
        __syncthreads();
        if (tid < n) {
            double d = 0.0;
            for (int q = 0; q < RG; ++q) d += part[q * CW + tid];
            w2[tid] = d;
        }
    }
    __syncthreads();
    // ---- seg + Y w2 on the rows this phase writes: a wave per row, four rows in flight
    auto carry_row = [&](int i) -> bool {        // stack rows that leave as the carry of the next panel of the chain
        return transpose ? (i >= solved && i < n) : (pidx >= 1 && i < 2 * lo && !(i & 1));
    };
    constexpr int RB3 = 4, LB3 = 4;
    for (int i0 = wave; i0 < m; i0 += RB3 * PA_WAVES) {
        double yv[RB3][LB3];
        bool want[RB3];
#pragma unroll
        for (int rb = 0; rb < RB3; ++rb) {
            const int i = i0 + rb * PA_WAVES;
            want[rb] = i < m && (carry_row(i) == (phase == 1));
            const int je = i < n ? i : n;
#pragma unroll
            for (int q = 0; q < LB3; ++q) { const int j = lane + 64 * q; yv[rb][q] = (want[rb] && j < je) ? Y[(int64_t)i * n + j] : 0.0; }
        }
#pragma unroll
        for (int rb = 0; rb < RB3; ++rb) {
            const int i = i0 + rb * PA_WAVES;
            if (!want[rb]) continue;             // (wave-uniform)
            double d = 0.0;
#pragma unroll
            for (int q = 0; q < LB3; ++q) { const int j = lane + 64 * q; if (j < n) d = fma(yv[rb][q], w2[j], d); }
            d = wsum(d);
            if (lane == 0) {
                const double val = seg[i] + d + (i < n ? w2[i] : 0.0);
                if (transpose) {
                    if (i < solved) full[(int64_t)s * pidx + i] = val;
                    else if (i < n) carr[(int64_t)(pidx + 1) * lo + (i - solved)] = val;
                    else full[ro + (i - n)] = val;
                } else {
                    if (pidx == 0) yi[i] = val;
                    else if (i < 2 * lo) { if (i & 1) yi[i >> 1] = val; else carr[(int64_t)(pidx - 1) * lo + (i >> 1)] = val; }
                    else yi[i - lo] = val;
                }
            }
        }
    }
    if (phase == 2) {
        // the components stage A left out of the chain (rows n.. of the strip's vector) pass through
        const int64_t rp = ro + (pidx >= 1 ? lo : 0);
        for (int q = tid; q < ms - n; q += PA_THREADS) { if (transpose) full[rp + q] = yi[n + q]; else yi[n + q] = full[rp + q]; }
    }
}

// ---- 3: an affine chain  v <- Op(M_p) v + add_p  over the maps p of a range ------------------------------------------------------------------
// One kernel for the three chains of the strips form (d = lo, maps row-major d x d at maps + p d^2, vectors at vecs + index * d):
//   Q^T b      p = 1 .. N - 2 ascending,   Op = M_p,    in = vecs[p],     out = vecs[p + 1]   (the carries)
//   Q x        p = N - 2 .. 1 descending,  Op = M_p^T,  in = vecs[p],     out = vecs[p - 1]
//   R^-1 y     p = N - 2 .. 0 descending,  Op = A_p,    in = vecs[p + 1], out = vecs[p]       (the state x[p s .. p s + lo))
// Step: out <- Op(M_p) in + out (out holds the additive term on entry); the out of a step is the in of the next.  One workgroup pulls a
// map of 131 KB out of HBM in about 6 us even with the next step's entries in flight (21 GB/s into one CU, profiles/
// r05_banded_solve_maps.txt), so a chain of N steps on one workgroup costs N x 6 us.  Hence TWO LEVELS: the range is cut into groups of K maps (blockIdx.y = group),
//   mode 1  every group at once from a ZERO vector, nothing stored but the group's result e_j          (gvec: [ngroups + 1][d])
//   mode 0  on the group products P_j (bbs_group_product_kernel) and the e_j: the true vector at every group boundary -- ngroups steps
//   mode 3  every group at once again from its true entering vector, storing every step
// i.e. 2 K + N / K sequential steps instead of N.  mode 0 with K >= the whole range is the plain one-workgroup chain (short chains, d > 128).
// blockIdx.x = right-hand side.  gvec index of the vector ENTERING group j: j (ascending) or j + 1 (descending).
template <bool TR, bool FAST>
__global__ void __launch_bounds__(bbm::CH_THREADS)
bbs_affine_chain_kernel(const double* __restrict__ maps, double* __restrict__ vecs_all, int64_t vecs_stride, int d, int p_lo, int p_hi, int K,
                        int dir, int in_off, int out_off, int mode, double* __restrict__ gvec_all, int64_t gvec_stride)
{
    using namespace bbm;
    __shared__ double cv[2][CH_THREADS];      // the vector of the step, double-buffered; zero beyond d
    __shared__ double part[CH_THREADS];
    const int tid = threadIdx.x, j = blockIdx.y, ng = gridDim.y;
    double* vecs = vecs_all + (int64_t)blockIdx.x * vecs_stride;
    double* gvec = gvec_all ? gvec_all + (int64_t)blockIdx.x * gvec_stride : nullptr;
    const int ga = p_lo + j * K, gb = (ga + K - 1 < p_hi) ? ga + K - 1 : p_hi;
    if (ga > gb) return;
    const int nsteps = gb - ga + 1, pfirst = dir > 0 ? ga : gb;
    const int64_t l2 = (int64_t)d * d;
    if (tid >= d) { cv[0][tid] = 0.0; cv[1][tid] = 0.0; }      // (the Op = M^T form reads entries beyond d against rows it does not have)
    if (tid < d) {
        double v0 = 0.0;
        if (mode == 0) v0 = vecs[(int64_t)(pfirst + in_off) * d + tid];
        else if (mode == 3) v0 = gvec[(int64_t)(dir > 0 ? j : j + 1) * d + tid];
        else if (dir > 0 ? j == 0 : j == ng - 1) gvec[(int64_t)(dir > 0 ? 0 : ng) * d + tid] = vecs[(int64_t)(pfirst + in_off) * d + tid];   // the chain's start
        cv[0][tid] = v0;
    }
    int buf = 0;
    const int Q = d / 16;
    if (FAST) {
        // d <= 128: every thread owns 16 entries of a map -- TR: 16 consecutive of row a (a partial sum of out[a]); else column b of the
        // rows g, g + G, .. (a partial sum of out[b]) -- and the entries of the next step are in flight while this one is summed
        typedef double d2 __attribute__((ext_vector_type(2)));
        const int G = CH_THREADS / d;
        const int a = TR ? tid / Q : 0, q = TR ? tid % Q : 0, b = TR ? 0 : tid % d, g = TR ? 0 : tid / d;
        const bool act = TR ? tid < d * Q : g < G;
        // (a uniform base and a 32-bit offset per thread: one address register per load instead of two)
        const unsigned mine = act ? (TR ? (unsigned)(a * d + 16 * q) : (unsigned)(g * d + b)) : 0u;
        double cur[16], nxt[16];
        double addc = 0.0, addn = 0.0;
        auto load = [&](int p, double (&dst)[16], double& add) {
            const double* M = maps + (int64_t)p * l2;
            if (TR) {
                const d2* src = reinterpret_cast<const d2*>(M + mine);
#pragma unroll
                for (int u = 0; u < 8; ++u) { const d2 v = src[u]; dst[2 * u] = v.x; dst[2 * u + 1] = v.y; }
            } else {
                // (rows beyond d read the thread's first row and meet a zero below)
#pragma unroll
                for (int u = 0; u < 16; ++u) dst[u] = M[mine + ((g + G * u < d) ? (unsigned)(G * u * d) : 0u)];
            }
            add = tid < d ? vecs[(int64_t)(p + out_off) * d + tid] : 0.0;
        };
        load(pfirst, cur, addc);
        __syncthreads();
        int p = pfirst;
        for (int k = 0; k < nsteps; ++k, p += dir) {
            const bool more = k + 1 < nsteps;
            if (more) load(p + dir, nxt, addn);
            double a0 = 0.0, a1 = 0.0;
            if (TR) {
                const double* c = &cv[buf][16 * q];
#pragma unroll
                for (int u = 0; u < 16; u += 2) { a0 = fma(cur[u], c[u], a0); a1 = fma(cur[u + 1], c[u + 1], a1); }
            } else {
#pragma unroll
                for (int u = 0; u < 16; u += 2) {      // (rows g + G u >= d meet a zero: cv[d .. 1023] = 0 and g + 15 G < 16 * 1024 / d <= 1024)
                    a0 = fma(cur[u], cv[buf][g + G * u], a0);
                    a1 = fma(cur[u + 1], cv[buf][g + G * (u + 1)], a1);
                }
            }
            if (act) part[tid] = a0 + a1;
            lds_barrier();
            if (tid < d) {
                double sum = addc;
                if (TR) { for (int z = 0; z < Q; ++z) sum += part[tid * Q + z]; }
                else { for (int z = 0; z < G; ++z) sum += part[z * d + tid]; }
                cv[buf ^ 1][tid] = sum;
                if (mode != 1) vecs[(int64_t)(p + out_off) * d + tid] = sum;
            }
            lds_barrier();
            buf ^= 1;
#pragma unroll
            for (int u = 0; u < 16; ++u) cur[u] = nxt[u];
            addc = addn;
        }
    } else {
        const int G = CH_THREADS / d, b = tid % d, g = tid / d;      // (not TR: column b of the rows g, g + G, ..)
        __syncthreads();
        int p = pfirst;
        for (int k = 0; k < nsteps; ++k, p += dir) {
            const double* M = maps + (int64_t)p * l2;
            double acc = 0.0;
            if (TR) {
                // 16-entry pieces of the rows in passes of CH_THREADS pieces (d <= 240: at most four); the partial sums of a row meet in LDS
                const int nch = d * Q;
                for (int base = 0; base < nch; base += CH_THREADS) {
                    const int e = base + tid;
                    if (e < nch) {
                        const int ra = e / Q, cq = e % Q;
                        const double* src = M + (int64_t)ra * d + 16 * cq;
                        double v[16];
#pragma unroll
                        for (int u = 0; u < 16; ++u) v[u] = src[u];
                        double sacc = 0.0;
#pragma unroll
                        for (int u = 0; u < 16; ++u) sacc = fma(v[u], cv[buf][16 * cq + u], sacc);
                        part[tid] = sacc;
                    }
                    lds_barrier();
                    if (tid < d) {
                        int z0 = tid * Q, z1 = z0 + Q;
                        if (z0 < base) z0 = base;
                        if (z1 > base + CH_THREADS) z1 = base + CH_THREADS;
                        if (z1 > nch) z1 = nch;
                        for (int z = z0; z < z1; ++z) acc += part[z - base];
                    }
                    lds_barrier();
                }
            } else {
                if (g < G) {
                    constexpr int U = 8;
                    for (int r = g; r < d; r += U * G) {
                        double v[U];
#pragma unroll
                        for (int u = 0; u < U; ++u) { const int rr = r + G * u; v[u] = M[(int64_t)(rr < d ? rr : 0) * d + b]; }
#pragma unroll
                        for (int u = 0; u < U; ++u) { const int rr = r + G * u; if (rr < d) acc = fma(v[u], cv[buf][rr], acc); }
                    }
                    part[tid] = acc;
                }
                lds_barrier();
                acc = 0.0;
                if (tid < d) for (int z = 0; z < G; ++z) acc += part[z * d + tid];
            }
            if (tid < d) {
                const double sum = acc + vecs[(int64_t)(p + out_off) * d + tid];
                cv[buf ^ 1][tid] = sum;
                if (mode != 1) vecs[(int64_t)(p + out_off) * d + tid] = sum;
            }
            __syncthreads();
            buf ^= 1;
        }
    }
    if (mode == 1 && tid < d) gvec[(int64_t)(dir > 0 ? j + 1 : j) * d + tid] = cv[buf][tid];
}

// P_j = M_last .. M_first over the maps of group j in the order the chain takes them (dir > 0: ascending p), d <= 128: C <- M_p C in LDS,
// 256 threads as 16 x 16 with an 8 x 8 tile each, M_p through LDS in slabs of 16 columns.  (Op = M^T chains use the same products: the
// descending chain over a group applies M_first^T .. -- the transpose of the ascending product.)
__global__ void __launch_bounds__(256)
bbs_group_product_kernel(const double* __restrict__ maps, int d, int p_lo, int p_hi, int K, int dir, double* __restrict__ pmap)
{
    extern __shared__ __attribute__((aligned(16))) double smem[];
    constexpr int MS = 132;                  // stride of a slab row (doubles)
    double* C = smem;                        // [d][d]
    double* Ms = C + d * d;                  // [16][MS]: Ms[kk][r] = M(r, k0 + kk)
    const int tid = threadIdx.x, tx = tid & 15, ty = tid >> 4, j = blockIdx.x;
    const int ga = p_lo + j * K, gb = (ga + K - 1 < p_hi) ? ga + K - 1 : p_hi;
    if (ga > gb) return;
    const int nsteps = gb - ga + 1, pfirst = dir > 0 ? ga : gb;
    const int64_t l2 = (int64_t)d * d;
    for (int e = tid; e < d * d; e += 256) C[e] = maps[(int64_t)pfirst * l2 + e];
    __syncthreads();
    const bool mine = 8 * ty < d && 8 * tx < d;
    int p = pfirst + dir;
    for (int k = 1; k < nsteps; ++k, p += dir) {
        const double* M = maps + (int64_t)p * l2;
        double acc[8][8] = {};
        for (int k0 = 0; k0 < d; k0 += 16) {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int e = tid + 256 * u, r = e >> 4, kk = e & 15;
                Ms[kk * MS + r] = r < d ? M[(int64_t)r * d + k0 + kk] : 0.0;
            }
            __syncthreads();
            if (mine) {
#pragma unroll
                for (int kk = 0; kk < 16; ++kk) {
                    double av[8], bv[8];
#pragma unroll
                    for (int z = 0; z < 8; ++z) { av[z] = Ms[kk * MS + 8 * ty + z]; bv[z] = C[(k0 + kk) * d + 8 * tx + z]; }
#pragma unroll
                    for (int r = 0; r < 8; ++r)
#pragma unroll
                        for (int c = 0; c < 8; ++c) acc[r][c] = fma(av[r], bv[c], acc[r][c]);
                }
            }
            __syncthreads();
        }
        if (mine) {
#pragma unroll
            for (int r = 0; r < 8; ++r)
#pragma unroll
                for (int c = 0; c < 8; ++c) C[(8 * ty + r) * d + 8 * tx + c] = acc[r][c];
        }
        __syncthreads();
    }
    for (int e = tid; e < d * d; e += 256) pmap[(int64_t)j * l2 + e] = C[e];
}

// ---- 1: M_p = S_o (I + Y' T^T Y'^T) E_c of the panels 1 .. N - 2 -------------------------------------------------------------------------
// Two products per 64 columns of M: Z = T^T B (n x 64, kept in LDS) with B[k][b] = Y'(2 b, k), then M = [sel] + Y'(solved + a, :) Z.
// 256 threads as 16 x 16, a 4 x 4 tile of the 64 x 64 block each, operands through LDS in slabs of 16.
__global__ void __launch_bounds__(256)
bbs_carry_map_kernel(const BBPanel* __restrict__ panels, const double* __restrict__ y_vals, const double* __restrict__ t_vals, int lo,
                     double* __restrict__ cmap)
{
    extern __shared__ __attribute__((aligned(16))) double smem[];
    constexpr int LS = 68;                   // row stride of the operand slabs (doubles)
    double* As = smem;                       // [16][LS]
    double* Bs = As + 16 * LS;               // [16][LS]
    double* Zs = Bs + 16 * LS;               // [n][64]
    const int tid = threadIdx.x, tx = tid & 15, ty = tid >> 4;
    const int pi = blockIdx.x + 1;
    const BBPanel p = panels[pi];
    const int n = p.ncols, sv = p.solved;
    const double* Y = y_vals + p.y_off;
    const double* T = t_vals + p.t_off;
    double* M = cmap + (int64_t)pi * lo * lo;
    auto yp = [&](int r, int k) -> double { return k < r ? Y[(int64_t)r * n + k] : (k == r ? 1.0 : 0.0); };      // unit-lower view, k < n
    for (int b0 = 0; b0 < lo; b0 += 64) {
        // ---- Z[j][bb] = sum_{k <= j} T(k, j) Y'(2 (b0 + bb), k)
        for (int j0 = 0; j0 < n; j0 += 64) {
            double acc[4][4] = {};
            int kend = j0 + 64 < n ? j0 + 64 : n;
            if (2 * (b0 + 63) + 1 < kend) kend = 2 * (b0 + 63) + 1;        // (Y'(2 b, k) = 0 for k > 2 b)
            for (int k0 = 0; k0 < kend; k0 += 16) {
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int e = tid + 256 * u, xx = e >> 4, kk = e & 15, k = k0 + kk;
                    const int j = j0 + xx, bq = b0 + xx;
                    As[kk * LS + xx] = (j < n && k <= j) ? T[(int64_t)j * n + k] : 0.0;
                    Bs[kk * LS + xx] = (bq < lo && k < n) ? yp(2 * bq, k) : 0.0;
                }
                __syncthreads();
#pragma unroll
                for (int kk = 0; kk < 16; ++kk) {
                    double av[4], bv[4];
#pragma unroll
                    for (int z = 0; z < 4; ++z) { av[z] = As[kk * LS + 4 * ty + z]; bv[z] = Bs[kk * LS + 4 * tx + z]; }
#pragma unroll
                    for (int r = 0; r < 4; ++r)
#pragma unroll
                        for (int c = 0; c < 4; ++c) acc[r][c] = fma(av[r], bv[c], acc[r][c]);
                }
                __syncthreads();
            }
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int c = 0; c < 4; ++c) { const int j = j0 + 4 * ty + r; if (j < n) Zs[j * 64 + 4 * tx + c] = acc[r][c]; }
        }
        __syncthreads();
        // ---- M[a][b0 + bb] = [solved + a == 2 (b0 + bb)] + sum_j Y'(solved + a, j) Z[j][bb]
        for (int a0 = 0; a0 < lo; a0 += 64) {
            double acc[4][4] = {};
            const int jend = sv + a0 + 64 < n ? sv + a0 + 64 : n;          // (Y'(r, j) = 0 for j > r)
            for (int k0 = 0; k0 < jend; k0 += 16) {
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int e = tid + 256 * u, xx = e >> 4, kk = e & 15, k = k0 + kk;
                    const int aq = a0 + xx;
                    As[kk * LS + xx] = (aq < lo && k < n) ? yp(sv + aq, k) : 0.0;
                }
                __syncthreads();
#pragma unroll
                for (int kk = 0; kk < 16; ++kk) {
                    if (k0 + kk >= n) break;
                    double av[4], bv[4];
#pragma unroll
                    for (int z = 0; z < 4; ++z) { av[z] = As[kk * LS + 4 * ty + z]; bv[z] = Zs[(k0 + kk) * 64 + 4 * tx + z]; }
#pragma unroll
                    for (int r = 0; r < 4; ++r)
#pragma unroll
                        for (int c = 0; c < 4; ++c) acc[r][c] = fma(av[r], bv[c], acc[r][c]);
                }
                __syncthreads();
            }
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const int aq = a0 + 4 * ty + r, bq = b0 + 4 * tx + c;
                    if (aq < lo && bq < lo) M[(int64_t)aq * lo + bq] = acc[r][c] + ((sv + aq == 2 * bq) ? 1.0 : 0.0);
                }
        }
        __syncthreads();
    }
}

// ---- the triangular solve ------------------------------------------------------------------------------------------------------------------
// Panel p <= N - 2 (solved = s <= 64 rows): x_p = D_p^-1 (y_p - U_p x_right), D = R(0:s, 0:s), U = R(0:s, s:n), R(i, j) = R[j s + i] in the
// staging array.  With the state u_p = x[p s .. p s + lo): u_p = A_p u_{p+1} + [D_p^-1 y_p; 0],
//      A_p = [ -D_p^-1 U_p ]   s rows
//            [  I   0       ]   lo - s rows (the entries of u_{p+1} that are still in the window)
// -- an affine chain like the carries'.  A thread per column of U, the column in registers, the triangle padded to 64 x 64 with the
// identity.  amap: [N][lo][lo], panels 0 .. N - 2 written.
__global__ void __launch_bounds__(256)
bbs_backsub_map_kernel(const BBPanel* __restrict__ panels, const double* __restrict__ r_stage, int lo, double* __restrict__ amap)
{
    __shared__ double blk[64 * 65];          // blk[j * 65 + i] = D(i, j)
    __shared__ double rd[64];
    const int tid = threadIdx.x;
    const BBPanel p = panels[blockIdx.x];
    const int sv = p.solved;
    const double* R = r_stage + p.r_off;
    for (int e = tid; e < 64 * 64; e += 256) {
        const int j = e >> 6, i = e & 63;
        blk[j * 65 + i] = (i < sv && j < sv) ? (i <= j ? R[(int64_t)j * sv + i] : 0.0) : (i == j ? 1.0 : 0.0);
    }
    __syncthreads();
    if (tid < 64) rd[tid] = 1.0 / blk[tid * 65 + tid];
    __syncthreads();
    double* Ap = amap + (int64_t)blockIdx.x * lo * lo;
    for (int c = tid; c < lo; c += 256) {
        double x[64];
#pragma unroll
        for (int i = 0; i < 64; ++i) x[i] = i < sv ? R[(int64_t)(sv + c) * sv + i] : 0.0;
#pragma unroll
        for (int i = 63; i >= 0; --i) {
            const double dd = blk[i * 65 + i], r = rd[i];
            double xi = x[i] * r;
            xi = fma(fma(-xi, dd, x[i]), r, xi);         // the quotient to the last bit but for rare ties (banded.hip, dense_solve_r_coop_kernel)
            x[i] = xi;
#pragma unroll
            for (int k = 0; k < i; ++k) x[k] = fma(-blk[i * 65 + k], xi, x[k]);
        }
#pragma unroll
        for (int i = 0; i < 64; ++i) if (i < sv) Ap[(int64_t)i * lo + c] = -x[i];
    }
    for (int e = tid; e < (lo - sv) * lo; e += 256) { const int i = sv + e / lo, c = e % lo; Ap[(int64_t)i * lo + c] = (c == i - sv) ? 1.0 : 0.0; }
}

// The additive terms of that chain: U[p] = [D_p^-1 y_p; 0] for the panels p <= N - 2 (blockIdx.x = panel, a wave per right-hand side, four
// to a workgroup) and U[N - 1] = the first lo entries of x of the last panel, which bb_solve_r_kernel has solved by now.  U: [nrhs][N][lo].
__global__ void __launch_bounds__(256)
bbs_backsub_diag_kernel(const BBPanel* __restrict__ panels, int num_panels, const double* __restrict__ r_stage, const double* __restrict__ v,
                        int64_t ldv, int64_t nrhs, int lo, double* __restrict__ U)
{
    __shared__ double blk[64 * 65];
    const int tid = threadIdx.x, ln = tid & 63, grp = tid >> 6;
    const int pidx = blockIdx.x;
    const BBPanel p = panels[pidx];
    const int64_t col = (int64_t)blockIdx.y * 4 + grp;
    if (pidx == num_panels - 1) {
        if (col < nrhs) for (int i = ln; i < lo; i += 64) U[(col * num_panels + pidx) * lo + i] = v[col * ldv + p.col0 + i];
        return;
    }
    const int sv = p.solved;
    const double* R = r_stage + p.r_off;
    for (int e = tid; e < sv * sv; e += 256) {
        const int j = e / sv, i = e - j * sv;
        blk[j * 65 + i] = R[(int64_t)j * sv + i];
    }
    __syncthreads();
    if (col >= nrhs) return;
    const double* x = v + col * ldv + p.col0;
    double t = ln < sv ? x[ln] : 0.0;
    const double dl = blk[(ln < sv ? ln : 0) * 65 + (ln < sv ? ln : 0)];
    const double rdl = 1.0 / dl;
    for (int i = sv - 1; i >= 0; --i) {
        const double ti = readlane_f64(t, i), di = readlane_f64(dl, i), ri = readlane_f64(rdl, i);
        double xi = ti * ri;
        xi = fma(fma(-xi, di, ti), ri, xi);
        if (ln < i) t = fma(-blk[i * 65 + ln], xi, t);
        else if (ln == i) t = xi;
    }
    double* u = U + (col * num_panels + pidx) * lo;
    for (int i = ln; i < lo; i += 64) u[i] = i < sv ? t : 0.0;     // (sv <= 64: lane i holds x_i)
}

// x[p s + i] = U[p][i], i < s, for the panels p <= N - 2
__global__ void __launch_bounds__(64)
bbs_backsub_scatter_kernel(int num_panels, int s, int lo, const double* __restrict__ U, double* __restrict__ v, int64_t ldv)
{
    const int pidx = blockIdx.x;
    const int64_t col = blockIdx.y;
    if (threadIdx.x < s) v[col * ldv + (int64_t)pidx * s + threadIdx.x] = U[(col * num_panels + pidx) * lo + threadIdx.x];
}

// ---- launchers ----------------------------------------------------------------------------------------------------------------------------
namespace bbm {
// group size of the two-level chains: 2 K + range / K sequential steps are least at K = sqrt(range / 2); 0: one level (short chains, and
// d > 128, for which there is no product kernel)
int group_size(int num_panels, int lo)
{
    if (lo > 128 || num_panels < 64) return 0;
    int K = 8;
    while ((int64_t)2 * (K + 1) * (K + 1) <= num_panels) ++K;
    return K;
}
int groups(int range, int K) { return K > 0 ? (range + K - 1) / K : 1; }

template <bool TR>
void launch_chain(bool fast, dim3 grid, hipStream_t stream, const double* maps, double* vecs, int64_t vstride, int d, int p_lo, int p_hi, int K,
                  int dir, int in_off, int out_off, int mode, double* gvec, int64_t gstride)
{
    if (fast) hipLaunchKernelGGL((bbs_affine_chain_kernel<TR, true>), grid, dim3(CH_THREADS), 0, stream, maps, vecs, vstride, d, p_lo, p_hi, K, dir,
                                 in_off, out_off, mode, gvec, gstride);
    else hipLaunchKernelGGL((bbs_affine_chain_kernel<TR, false>), grid, dim3(CH_THREADS), 0, stream, maps, vecs, vstride, d, p_lo, p_hi, K, dir,
                            in_off, out_off, mode, gvec, gstride);
}

// the chain over the maps [p_lo, p_hi] in one or two levels; prods: the group products of that range (K > 0); gvec: [nrhs][groups + 1][d]
template <bool TR>
void run_chain(hipStream_t stream, const double* maps, const double* prods, int K, double* vecs, int64_t vstride, int d, int p_lo, int p_hi,
               int dir, int in_off, int out_off, int64_t nrhs, double* gvec)
{
    if (p_hi < p_lo) return;
    const bool fast = d <= 128;
    const int range = p_hi - p_lo + 1;
    if (K <= 0 || range <= K) {
        launch_chain<TR>(fast, dim3((unsigned)nrhs, 1), stream, maps, vecs, vstride, d, p_lo, p_hi, range, dir, in_off, out_off, 0, nullptr, 0);
        return;
    }
    const int ng = groups(range, K);
    const int64_t gstride = (int64_t)(ng + 1) * d;
    launch_chain<TR>(fast, dim3((unsigned)nrhs, (unsigned)ng), stream, maps, vecs, vstride, d, p_lo, p_hi, K, dir, in_off, out_off, 1, gvec, gstride);
    // the boundaries: vectors gvec[0 .. ng], maps prods[0 .. ng - 1]; ascending: in = gvec[j], out = gvec[j + 1]; descending: in = gvec[j + 1], out = gvec[j]
    launch_chain<TR>(fast, dim3((unsigned)nrhs, 1), stream, prods, gvec, gstride, d, 0, ng - 1, ng, dir, dir > 0 ? 0 : 1, dir > 0 ? 1 : 0, 0, nullptr, 0);
    launch_chain<TR>(fast, dim3((unsigned)nrhs, (unsigned)ng), stream, maps, vecs, vstride, d, p_lo, p_hi, K, dir, in_off, out_off, 3, gvec, gstride);
}
}  // namespace bbm

// sizes of the scratch a plan keeps for these products (doubles): products of the carry chain / of the back substitution, boundary vectors per rhs
void bbs_maps_sizes(int num_panels, int lo, int* K, int64_t* cprod_len, int64_t* aprod_len, int64_t* gvec_len_per_rhs)
{
    const int k = bbm::group_size(num_panels, lo);
    const int ngc = bbm::groups(num_panels - 2, k), nga = bbm::groups(num_panels - 1, k);
    if (K) *K = k;
    if (cprod_len) *cprod_len = k > 0 ? (int64_t)ngc * lo * lo : 0;
    if (aprod_len) *aprod_len = k > 0 ? (int64_t)nga * lo * lo : 0;
    if (gvec_len_per_rhs) *gvec_len_per_rhs = (int64_t)((nga > ngc ? nga : ngc) + 1) * lo;
}

// the maps of a factorisation: cmap [N][lo][lo] (panels 1 .. N - 2 written), amap [N][lo][lo] or null (s > 64 or s > lo: the triangular solve stays
// on bb_solve_r_kernel), and with K > 0 the group products cprod / aprod (bbs_maps_sizes)
hipError_t launch_bbs_maps(const BBPanel* panels, int num_panels, const double* y_vals, const double* t_vals, const double* r_stage, int n,
                           int lo, double* cmap, double* amap, int K, double* cprod, double* aprod, hipStream_t stream)
{
    if (lo <= 0 || num_panels < 2) return hipSuccess;
    const size_t psmem = (size_t)(lo * lo + 16 * 132) * sizeof(double);
    if (K > 0) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(bbs_group_product_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)psmem);
        if (e != hipSuccess) return e;
    }
    if (num_panels >= 3) {
        const size_t smem = (size_t)(2 * 16 * 68 + n * 64) * sizeof(double);
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(bbs_carry_map_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(bbs_carry_map_kernel, dim3((unsigned)(num_panels - 2)), dim3(256), smem, stream, panels, y_vals, t_vals, lo, cmap);
        if (K > 0 && num_panels - 2 > K)
            hipLaunchKernelGGL(bbs_group_product_kernel, dim3((unsigned)bbm::groups(num_panels - 2, K)), dim3(256), psmem, stream, cmap, lo, 1,
                               num_panels - 2, K, 1, cprod);
    }
    if (amap) {
        hipLaunchKernelGGL(bbs_backsub_map_kernel, dim3((unsigned)(num_panels - 1)), dim3(256), 0, stream, panels, r_stage, lo, amap);
        if (K > 0 && num_panels - 1 > K)
            hipLaunchKernelGGL(bbs_group_product_kernel, dim3((unsigned)bbm::groups(num_panels - 1, K)), dim3(256), psmem, stream, amap, lo, 0,
                               num_panels - 2, K, -1, aprod);
    }
    return hipGetLastError();
}

// Q^T (transpose) or Q through the carry maps; carr: [nrhs][N][lo] scratch, gvec: [nrhs][bbs_maps_sizes] scratch
hipError_t launch_bbs_apply_maps(const BBPanel* panels, int num_panels, const double* y_vals, const double* t_vals, const double* cmap,
                                 const double* cprod, int K, int transpose, double* ya, int64_t ya_ld, double* full, int64_t full_ld,
                                 int64_t nrhs, int ms, int n, int s, int lo, int cols, int max_act, double* carr, double* gvec, hipStream_t stream)
{
    using namespace bbm;
    if (nrhs <= 0) return hipSuccess;
    if (nrhs > 65535 || lo <= 0 || lo % 16 || num_panels < 2) return hipErrorInvalidValue;
    const size_t smem = (size_t)(max_act + 2 * n + PA_THREADS) * sizeof(double);
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(bbs_panel_apply_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    if (e != hipSuccess) return e;
    const dim3 grid((unsigned)num_panels, (unsigned)nrhs);
    hipLaunchKernelGGL(bbs_panel_apply_kernel, grid, dim3(PA_THREADS), smem, stream, panels, num_panels, y_vals, t_vals, transpose, 1, ya, ya_ld,
                       full, full_ld, carr, ms, s, lo, cols, max_act, n);
    const int64_t vstride = (int64_t)num_panels * lo;
    if (transpose) run_chain<true>(stream, cmap, cprod, K, carr, vstride, lo, 1, num_panels - 2, 1, 0, 1, nrhs, gvec);
    else run_chain<false>(stream, cmap, cprod, K, carr, vstride, lo, 1, num_panels - 2, -1, 0, -1, nrhs, gvec);
    hipLaunchKernelGGL(bbs_panel_apply_kernel, grid, dim3(PA_THREADS), smem, stream, panels, num_panels, y_vals, t_vals, transpose, 2, ya, ya_ld,
                       full, full_ld, carr, ms, s, lo, cols, max_act, n);
    return hipGetLastError();
}

// x(0:cols) <- R^-1 x(0:cols) through A_p (s <= 64): the last panel as before, the additive terms of every other panel at once, the chain
// of the states, x out of the states.  U: [nrhs][N][lo] scratch (the carry scratch), gvec as above.
hipError_t launch_bbs_solve_r_maps(const BBPanel* panels, int num_panels, const double* r_stage, const double* amap, const double* aprod, int K,
                                   int s, int lo, int cols, double* v, int64_t ldv, int64_t nrhs, double* U, double* gvec, hipStream_t stream)
{
    using namespace bbm;
    if (nrhs <= 0) return hipSuccess;
    // (s > lo: the map of a panel would hold s rows of lo entries in an lo x lo slot, and the scatter would read past a panel's state)
    if (nrhs > 65535 || s > 64 || lo <= 0 || lo % 16 || s > lo || num_panels < 2 || !amap) return hipErrorInvalidValue;
    hipError_t e = launch_bb_solve_r(panels + (num_panels - 1), 1, r_stage, cols, v, ldv, nrhs, stream);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(bbs_backsub_diag_kernel, dim3((unsigned)num_panels, (unsigned)((nrhs + 3) / 4)), dim3(256), 0, stream, panels, num_panels,
                       r_stage, v, ldv, nrhs, lo, U);
    run_chain<true>(stream, amap, aprod, K, U, (int64_t)num_panels * lo, lo, 0, num_panels - 2, -1, 1, 0, nrhs, gvec);
    hipLaunchKernelGGL(bbs_backsub_scatter_kernel, dim3((unsigned)(num_panels - 1), (unsigned)nrhs), dim3(64), 0, stream, num_panels, s, lo, U, v, ldv);
    return hipGetLastError();
}

}  // namespace qrk
