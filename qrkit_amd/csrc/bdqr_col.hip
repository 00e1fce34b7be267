// bdqr_col.hip -- one workgroup factorises one tile with 32 < max(rows, cols) <= 256 (rows >= cols) of a
// block-diagonal matrix: A_i P_i = Q_i R_i with explicit Q_i, for gfx950.
//
// Same reference seam as bdqr_pair.hip (the hot loop of BlockDiagonalSparseQR::factorize,
// src/QRKit/BlockDiagonalSparseQR.h:432-526, Eigen ColPivHouseholderQR / HouseholderQR behind
// blockSolver.compute and HouseholderSequence behind matrixQ()), for the mid-size tiles of mixed
// batches (BASELINE configs[4]: sizes 8..256).
//
// Phase 1 -- R and the reflectors.  One THREAD per column of A, the matrix row-major in LDS (tiles up to
// 36 KB, e.g. 64x64) or in a per-workgroup global workspace, so that the threads of a wave touch consecutive
// addresses and no cross-lane reduction exists: every thread walks down its own column.  The two
// sweeps of the level-2 step (dot products, update) are fused across steps into ONE read-modify-write
// sweep: the dot products of step k give row k of the updated matrix (c_k + w gamma), that row is all
// the LAWN-176 norm downdate needs, so the pivot of step k+1 is known BEFORE the update of step k is
// applied; its column is brought up to date on the fly (x' = A(:,p') - gamma_p' x) and the sweep that
// applies update k also accumulates the dot products with x' (and |x'_tail|^2).  Only when Eigen's
// recompute test fires (rare) the step falls back to separate sweeps, because then the next pivot
// depends on the recomputed norms.  A chosen column stops sweeping: its thread stores the essential
// part of the reflector (x_tail / (x0 - beta), Eigen's packed form) in its place.
// Columns are never swapped: the column chosen at step k ends at position k; the "first maximum" tie
// rule compares current positions, which every thread tracks (Eigen's transpositions).
//
// Phase 2 -- Q_i = H_0 H_1 ... H_{c-1} (HouseholderSequence::evalTo) by BLOCKED backward accumulation
// directly in the output array (row-major Q_i is the CSR value order of m_Q): panels of NB reflectors
// as I - V T V^T (T by the larft recurrence from V^T V), applied to Q(kp:, kp:) with one thread per
// column of Q: w = V^T q, u = T w, q -= V u.  Q is read and written once per PANEL instead of once per
// reflector, which is what makes tiles above 64x64 affordable: carrying Q^T through the level-2
// sweeps (the first version of this kernel) moved 3x the bytes and ran at Infinity-Cache speed.
#include "qrk_device.h"
#include "bdqr_col_finish.h"

#include <float.h>
#include <cstdlib>
#include <type_traits>

#ifndef QRK_COL_UF
#define QRK_COL_UF 16          // loads in flight per thread in the read-only pass of the panel-blocked phase 1
#endif
#ifndef QRK_COL_UT
#define QRK_COL_UT 8           // ... in the trailing update of a panel
#endif
#ifndef QRK_COL_MFMA_TRAIL
#define QRK_COL_MFMA_TRAIL 1     // large class: the trailing update of a panel on the matrix cores (0 = one thread per column, scalar FMAs)
#endif
#ifndef QRK_COL_BIGW
#define QRK_COL_BIGW 2           // waves per SIMD the large-class instantiation is compiled for
#endif
#ifndef QRK_COL_REGROWS
#define QRK_COL_REGROWS 48     // rows of its column that a thread of the large class keeps in registers (0 = none)
#endif
#ifndef QRK_COL_BLOCKED
#define QRK_COL_BLOCKED 1      // tiles in global memory: panel-blocked phase 1 (0 = fused level-2 sweeps)
#endif

namespace qrk {

namespace col {

using namespace decide;   // decision margins of the fast kernels (qrk_device.h)
constexpr int W_LDS_DOUBLES = 4352;                   // 34 KB of LDS for A when the tile fits (two workgroups of the fixed layout per CU)
constexpr int NB = colfin::NB;                        // reflectors per block in the formation of Q (bdqr_col_finish.h)
constexpr int MAXR = 256;                             // largest tile dimension of this kernel

struct Cand {          // candidate of the pivot search
    double val;        // squared updated norm, < 0 = none
    int pos;           // current position (tie rule: smallest)
    int tidx;          // owning thread = original column
    double ngam;       // -gamma of that column in the step being applied
};

__device__ __forceinline__ bool better(const Cand& a, const Cand& b)   // a beats b
{
    return a.val > b.val || (a.val == b.val && a.pos < b.pos);
}

__device__ __forceinline__ double wave_sum_d(double v)
{
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off);
    return v;
}

__device__ __forceinline__ Cand wave_best(Cand c)
{
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        Cand o;
        o.val = __shfl_xor(c.val, off);
        o.pos = __shfl_xor(c.pos, off);
        o.tidx = __shfl_xor(c.tidx, off);
        o.ngam = __shfl_xor(c.ngam, off);
        if (better(o, c)) c = o;
    }
    return c;
}

}  // namespace col

// Everything of one tile.  Called once with W in LDS and once with W in global memory, so that after
// inlining hipcc knows the address space of every access: through one generic pointer it has to assume
// that a store to W may alias the LDS vectors and serialises the sweeps on the store latency.
// REG (the large class, panel-blocked form): every column thread keeps the LAST RR rows of its column in registers for as long
// as the panels stay above them.  The read-only pass of a step covers rows k.. of every live column, so rows near the bottom are
// read in (almost) every step: with RR = 64 of 256 rows 37 % of the bytes of all passes never leave the registers (48 % at 192
// rows).  The registers are written back once, when the first panel reaches them.
template <int CT, bool BLOCKED, bool DYN, bool REG = false>
__device__ __forceinline__ void factor_tile(double* __restrict__ W, double* smem, int r, int c, int cbase, int pivoting,
                                            const double* __restrict__ src, double* __restrict__ Q,
                                            double* __restrict__ rv, int32_t* __restrict__ perm,
                                            double* __restrict__ hcoeffs, const int w_lds_dyn, const int max_r_dyn,
                                            int32_t* __restrict__ redo_count, int32_t* __restrict__ redo_ids, int gid)
{
    // LDS carve-up.  DYN (launches of tiles with at most 64 columns): w_lds doubles for A of the tiles that fit and
    // vectors of max_r rows, the largest tile of the launch - small tiles then leave room for many workgroups per
    // CU.  Otherwise the fixed layout (two workgroups per CU): the offsets are then compile-time constants, which
    // is worth 5-8 % on the large tiles.
    using namespace col;
    const int w_lds = DYN ? w_lds_dyn : W_LDS_DOUBLES, max_r = DYN ? max_r_dyn : MAXR;
    constexpr int NW = CT / 64;
    double* vs = smem + w_lds;                           // [max_r * (NB + 1)] V panel: phase 1 stride NB, phase 2 NB + 1
    double* xv0 = vs + max_r * (NB + 1);                 // [max_r] pivot column, even steps
    double* xv1 = xv0 + max_r;                           // [max_r] odd steps
    double* taus = xv1 + max_r;                          // [max_r] Householder coefficients
    double* gm = taus + max_r;                           // [NB * NB] V^T V of a panel
    double* tm = gm + NB * NB;                           // [NB * NB] T of a panel
    double* cval = tm + NB * NB;                         // [NW] candidates of the waves
    double* cngam = cval + NW;                           // [NW]
    int* cpos = reinterpret_cast<int*>(cngam + NW);      // [NW]
    int* ctid = cpos + NW;                               // [NW]
    int* flags = ctid + NW;                              // [2] any-need flags (double buffered) + [1] redo flag (+ pad)
    int* col_of_pos = flags + 4;                         // [max_r] column chosen at step k
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int ld = c;                                    // A row-major: W(i, j) = W[i * ld + j]
    {
        {
        for (int e = tid; e < r * c; e += CT) {           // (A arrives column-major)
            const int i = e / c, j = e - i * c;
            W[(int64_t)i * ld + j] = src[(int64_t)j * r + i];
        }
        if (tid < 2) flags[tid] = 0;
        if (tid == 0) flags[2] = 0;          // some decision of this tile was not clear of rounding
        __syncthreads();

        // ================= phase 1: R, reflectors, permutation =================
        const bool isA = tid < c;            // this thread owns column tid of A  (c <= CT)
        double* wc = W + tid;                // wc[i * ld] = A(i, tid)
        bool live = isA;
        int pos = tid;                       // current position of the column
        double nu2 = -1.0, thr = 0.0;
        double a2 = 0.0;                     // |A|^2: squared norm of the first pivot column
        bool unclear = false;
        if (isA) {
            // squared column norms (ColPivHouseholderQR: m_colNormsUpdated^2, sqrt(eps) m_colNormsDirect^2)
            double s = 0.0;
            for (int i = 0; i < r; ++i) { const double v = wc[(int64_t)i * ld]; s = fma(v, v, s); }
            nu2 = s; thr = s * THR_HI;
        }

        if (BLOCKED) {
        // ---- panel-blocked form (LAPACK dlaqps) for tiles whose matrix lives in global memory: inside a panel of
        // NBP columns the trailing matrix is only READ -- per column one pass F(:, j) = tau A^T v, corrected for
        // the reflectors of the panel not yet applied -- while the chosen column and row k of R are brought up to
        // date on demand from V (LDS) and F (registers); the trailing matrix is updated once per panel,
        // A -= V F^T.  Half the bytes of the fused level-2 sweeps, and mostly reads.
        constexpr int NBP = NB;
#ifdef QRK_COL_PROF
        unsigned long long pt[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, pt0 = __builtin_amdgcn_s_memtime();
#define COL_TICK(z) do { const unsigned long long t1 = __builtin_amdgcn_s_memtime(); pt[z] += t1 - pt0; pt0 = t1; } while (0)
#else
#define COL_TICK(z) do { } while (0)
#endif
        double* vp = vs;                      // [rows k0.. x NBP] V of the current panel, row-major
        double* xv = xv0;                     // [r] x, then v, of the current reflector
        double* fP = gm;                      // [NBP] F row of the pivot column
        double* cvec = gm + NBP;              // [NBP] V^T v
        double* red = tm;                     // [NW] block reduction
        double F[NBP];
        constexpr int RR = REG ? QRK_COL_REGROWS : 1;
        double ar[RR];                        // rows rbase.. of this thread's column while inreg
        const int rbase = r - RR;
        const bool use_reg = REG && rbase >= 2 * NBP;
        double* xr = smem;                    // [RR] register rows of the pivot column, published by its thread (the LDS that holds
                                              // A in the LDS-resident form is free here)
        // panels [0, kflush) run with the register rows, the rest without: two instantiations of the panel loop, so that inside each
        // the register array is read and written unconditionally (a register array defined under a run-time flag ends up in scratch)
        int kflush = 0;
        if (use_reg) { kflush = (rbase / NBP) * NBP; const int cend = ((c + NBP - 1) / NBP) * NBP; if (kflush > cend) kflush = cend; }
#pragma unroll
        for (int u = 0; u < RR; ++u) { const int ii = (use_reg && isA) ? rbase + u : 0; ar[u] = wc[(int64_t)ii * ld]; }
        auto run_panels = [&](auto tag, const int k0_begin, const int k0_end) {
        constexpr bool inreg = decltype(tag)::value;
        constexpr bool MFMA_TRAIL = !DYN && CT == 256 && QRK_COL_MFMA_TRAIL != 0;
        for (int k0 = k0_begin; k0 < k0_end && k0 < c; k0 += NBP) {
            const int kb = (c - k0) < NBP ? (c - k0) : NBP;
            const int rm = inreg ? rbase : r;  // rows [k, rm) of the live columns are in memory
#pragma unroll
            for (int l = 0; l < NBP; ++l) F[l] = 0.0;
            for (int j = 0; j < kb; ++j) {
                const int k = k0 + j;
                int P = k, ppos = k;
                if (pivoting) {
                    Cand cd{live ? nu2 : -1.0, pos, tid, 0.0};
                    cd = wave_best(cd);
                    if (lane == 0) { cval[wave] = cd.val; cpos[wave] = cd.pos; ctid[wave] = cd.tidx; }
                    __syncthreads();
                    Cand bb{cval[0], cpos[0], ctid[0], 0.0};
#pragma unroll
                    for (int w = 1; w < NW; ++w) { Cand o{cval[w], cpos[w], ctid[w], 0.0}; if (better(o, bb)) bb = o; }
                    P = bb.tidx; ppos = bb.pos;
                    if (k == 0) a2 = bb.val;
                    if (live && tid != P && near_best(nu2, thr, bb.val, a2)) unclear = true;
                    if (isA) { if (tid == P) pos = k; else if (pos == k) pos = ppos; }
                }
                COL_TICK(0);
                if (tid == P) {
                    live = false; col_of_pos[k] = tid;
#pragma unroll
                    for (int l = 0; l < NBP; ++l) fP[l] = F[l];
                    if (inreg) {
#pragma unroll
                        for (int u = 0; u < RR; ++u) xr[u] = ar[u];
                    }
                }
                __syncthreads();
                // x = column P brought up to date (rows k..): A(k:, P) - V(k:, 0:j) F(P, 0:j)^T
                double part = 0.0;
                for (int i = k + tid; i < rm; i += CT) {
                    double x = W[(int64_t)i * ld + P];
#pragma unroll
                    for (int l = 0; l < NBP; ++l) if (l < j) x = fma(-vp[(i - k0) * NBP + l], fP[l], x);
                    xv[i] = x;
                    if (i > k) part = fma(x, x, part);
                }
                if (inreg) {
                    // (the rows the pivot thread held in registers; separate loop: one loop with a pointer chosen per row makes hipcc
                    //  emit a generic-address select that its own verifier rejects)
                    for (int i = rbase + tid; i < r; i += CT) {
                        double x = xr[i - rbase];
#pragma unroll
                        for (int l = 0; l < NBP; ++l) if (l < j) x = fma(-vp[(i - k0) * NBP + l], fP[l], x);
                        xv[i] = x;
                        part = fma(x, x, part);
                    }
                }
                part = wave_sum_d(part);
                if (lane == 0) red[wave] = part;
                __syncthreads();
                COL_TICK(1);
                double tsq = 0.0;
#pragma unroll
                for (int w = 0; w < NW; ++w) tsq += red[w];
                const double xk = xv[k];
                double beta, tau, inv_s;
                if (k == 0 && !pivoting) a2 = fma(xk, xk, tsq);
                if (unclear_reflector(xk, tsq, k + 1 < r, pivoting != 0, a2)) unclear = true;
                if (!(tsq > DBL_MIN)) { beta = xk; tau = 0.0; inv_s = 0.0; }     // makeHouseholder: H = I
                else {
                    const double nrm = sqrt(fma(xk, xk, tsq));
                    beta = xk >= 0.0 ? -nrm : nrm;
                    inv_s = 1.0 / (xk - beta);
                    tau = (beta - xk) / beta;
                }
                __syncthreads();        // every x has been read (xk, the reduction) before it becomes v
                for (int i = k0 + tid; i < r; i += CT) {
                    double v = 0.0;
                    if (i == k) v = 1.0; else if (i > k) v = xv[i] * inv_s;
                    vp[(i - k0) * NBP + j] = v;
                    if (i >= k) { xv[i] = v; W[(int64_t)i * ld + P] = (i == k) ? beta : v; }   // packed QR: beta, essential part
                }
                if (tid == 0) { taus[k] = tau; if (hcoeffs) hcoeffs[cbase + k] = tau; }
                __syncthreads();
                COL_TICK(2);
                // c_l = V(:, l)^T v for the earlier reflectors of the panel (one wave per l)
                for (int l = wave; l < j; l += NW) {
                    double cp = 0.0;
                    for (int i = k + lane; i < r; i += 64) cp = fma(vp[(i - k0) * NBP + l], xv[i], cp);
                    cp = wave_sum_d(cp);
                    if (lane == 0) cvec[l] = cp;
                }
                __syncthreads();
                COL_TICK(3);
                if (live) {
                    // F(t, j) = tau (A(k:, t)^T v - F(t, 0:j) c): read-only pass over this thread's column
                    double f = 0.0;
                    {
                        // (U loads in flight, no scalar tail: rows past the end are clamped and meet a zero of v, so the sum
                        //  is the same sequence of FMAs - the pass is bound by the round trips to L2 / Infinity Cache)
                        constexpr int U = QRK_COL_UF;
                        for (int i = k; i < rm; i += U) {
                            double wv[U];
#pragma unroll
                            for (int u = 0; u < U; ++u) { int ii = i + u; ii = ii < rm ? ii : rm - 1; wv[u] = wc[(int64_t)ii * ld]; }
#pragma unroll
                            for (int u = 0; u < U; ++u) { const int ii = i + u; f = fma(wv[u], ii < rm ? xv[ii < rm ? ii : rm - 1] : 0.0, f); }
                        }
                        if (inreg) {
                            // (eight rows at a time: left alone, the scheduler hoists all RR reads of v ahead of the FMAs and spills)
#pragma unroll
                            for (int u = 0; u < RR; ++u) {
                                f = fma(ar[u], xv[rbase + u], f);
                                if ((u & 7) == 7) __builtin_amdgcn_sched_barrier(0);
                            }
                        }
                    }
#pragma unroll
                    for (int l = 0; l < NBP; ++l) if (l < j) f = fma(-F[l], cvec[l], f);
                    f *= tau;
#ifdef QRK_COL_PROF
                    if (f == 1.2345e300) pt[9]++;      // (the loads have to land before the tick)
                    COL_TICK(4);
#endif
#pragma unroll
                    for (int l = 0; l < NBP; ++l) if (l == j) F[l] = f;
                    // row k of R: A(k, t) - V(k, 0:j+1) F(t, 0:j+1)^T
                    double an = wc[(int64_t)k * ld];
#pragma unroll
                    for (int l = 0; l < NBP; ++l) if (l <= j) an = fma(-vp[(k - k0) * NBP + l], F[l], an);
                    wc[(int64_t)k * ld] = an;
                    if (pivoting) {
                        // LAWN-176 downdate (squared form); Eigen's recompute uses the up-to-date column
                        const double nn = fma(-an, an, nu2);
                        nu2 = nn;
                        if (nn <= thr) {
                            if (in_recompute_band(nn, thr, a2)) unclear = true;    // (2) inside the band around Eigen's threshold
                            double s2 = 0.0;
                            for (int i = k + 1; i < rm; ++i) {
                                double a = wc[(int64_t)i * ld];
#pragma unroll
                                for (int l = 0; l < NBP; ++l) if (l <= j) a = fma(-vp[(i - k0) * NBP + l], F[l], a);
                                s2 = fma(a, a, s2);
                            }
                            if (inreg) {
#pragma unroll
                                for (int u = 0; u < RR; ++u) {
                                    double a = ar[u];
#pragma unroll
                                    for (int l = 0; l < NBP; ++l) if (l <= j) a = fma(-vp[(rbase + u - k0) * NBP + l], F[l], a);
                                    s2 = fma(a, a, s2);
                                    if ((u & 1) == 1) __builtin_amdgcn_sched_barrier(0);
                                }
                            }
                            nu2 = s2; thr = s2 * THR_HI;
                        }
                    }
                }
            }
            COL_TICK(5);
            // trailing update of the live columns: A(k0+kb:, t) -= V(k0+kb:, :) F(t, :)^T
            if (MFMA_TRAIL) {
                // The same update on the matrix cores (the large class: fixed LDS layout, whose A region is free here):
                // A(row0:rm, :) += V(row0:rm, 0:16) (-F)^T as 16 x 16 tiles with v_mfma_f64_16x16x4_f64, a wave per strip of 16
                // columns.  -F^T goes through LDS (the B operand wants F(col, 4 ks + lane / 16) of column lane % 16: other
                // threads' registers); chosen columns carry zeros, so their places (reflectors) are rewritten unchanged.  The
                // products of one entry are accumulated in the order l = 0..15 of the scalar form.
                typedef double d4 __attribute__((ext_vector_type(4)));
                double* fT = smem + 64;                     // [NBP][CT]
#pragma unroll
                for (int l = 0; l < NBP; ++l) fT[l * CT + tid] = live ? -F[l] : 0.0;
                __syncthreads();
                const int row0 = k0 + kb;
                const int kq = lane >> 4, l15 = lane & 15;
                const int nrt = (rm - row0 + 15) >> 4, nct = (c + 15) >> 4;
                for (int st = wave; st < nct; st += NW) {
                    const int col = 16 * st + l15;
                    const bool cok = col < c;
                    double bop[4];
#pragma unroll
                    for (int ks = 0; ks < 4; ++ks) bop[ks] = fT[(4 * ks + kq) * CT + (cok ? col : 0)];
                    if (!cok) { bop[0] = 0.0; bop[1] = 0.0; bop[2] = 0.0; bop[3] = 0.0; }
                    double* wcol = W + (cok ? col : 0);
                    constexpr int UT = 2;
                    for (int rt = 0; rt < nrt; rt += UT) {
                        d4 dv[UT];
#pragma unroll
                        for (int u2 = 0; u2 < UT; ++u2)
#pragma unroll
                            for (int z = 0; z < 4; ++z) {
                                int row = row0 + 16 * (rt + u2) + kq + 4 * z; row = row < rm ? row : rm - 1;
                                dv[u2][z] = wcol[(int64_t)row * ld];
                            }
#pragma unroll
                        for (int u2 = 0; u2 < UT; ++u2) {
                            int arow = row0 + 16 * (rt + u2) + l15; arow = arow < rm ? arow : rm - 1;
#pragma unroll
                            for (int ks = 0; ks < 4; ++ks)
                                dv[u2] = __builtin_amdgcn_mfma_f64_16x16x4f64(vp[(arow - k0) * NBP + 4 * ks + kq], bop[ks], dv[u2], 0, 0, 0);
#pragma unroll
                            for (int z = 0; z < 4; ++z) {
                                const int row = row0 + 16 * (rt + u2) + kq + 4 * z;
                                if (row < rm && cok) wcol[(int64_t)row * ld] = dv[u2][z];
                            }
                        }
                    }
                }
            } else if (live) {
                constexpr int U = QRK_COL_UT;
                for (int i = k0 + kb; i < rm; i += U) {
                    double wv[U];
#pragma unroll
                    for (int u = 0; u < U; ++u) { int ii = i + u; ii = ii < rm ? ii : rm - 1; wv[u] = wc[(int64_t)ii * ld]; }
#pragma unroll
                    for (int u = 0; u < U; ++u) {
                        int ii = i + u; const bool in = ii < rm; ii = in ? ii : rm - 1;
#pragma unroll
                        for (int l = 0; l < NBP; ++l) wv[u] = fma(-vp[(ii - k0) * NBP + l], F[l], wv[u]);
                        if (in) wc[(int64_t)ii * ld] = wv[u];
                    }
                }
            }
            if (inreg && live) {
#pragma unroll
                for (int u = 0; u < RR; ++u) {
#pragma unroll
                    for (int l = 0; l < NBP; ++l) ar[u] = fma(-vp[(rbase + u - k0) * NBP + l], F[l], ar[u]);
                    if ((u & 1) == 1) __builtin_amdgcn_sched_barrier(0);
                }
            }
            __syncthreads();
            COL_TICK(6);
        }
        };
        run_panels(std::integral_constant<bool, REG>(), 0, kflush);
        if (REG && kflush > 0) {
            // the next panel reaches the register rows: back to memory (live columns only: a chosen column's place holds its reflector)
            if (live) {
#pragma unroll
                for (int u = 0; u < RR; ++u) wc[(int64_t)(rbase + u) * ld] = ar[u];
            }
            __syncthreads();
        }
        run_panels(std::integral_constant<bool, false>(), kflush, c);
#ifdef QRK_COL_PROF
        if (blockIdx.x == 0 && (tid == 0 || tid == c - 1 || tid == c / 2))
            printf("col prof (last column's thread, 100 MHz ticks): pivot %llu  x %llu  v %llu  VtV %llu  Fpass %llu  downdate %llu  trailing %llu\n",
                   pt[0], pt[1], pt[2], pt[3], pt[4], pt[5], pt[6]);
#endif
        } else {
        // ---- head of step 0: pivot, its column to LDS, dot products
        int P;                               // pivot thread of the current step
        {
            if (pivoting) {
                Cand cd{live ? nu2 : -1.0, pos, tid, 0.0};
                cd = wave_best(cd);
                if (lane == 0) { cval[wave] = cd.val; cpos[wave] = cd.pos; ctid[wave] = cd.tidx; }
                __syncthreads();
                Cand b{cval[0], cpos[0], ctid[0], 0.0};
#pragma unroll
                for (int w = 1; w < NW; ++w) { Cand o{cval[w], cpos[w], ctid[w], 0.0}; if (better(o, b)) b = o; }
                P = b.tidx;
                const int ppos = b.pos;      // old position of the pivot column
                a2 = b.val;
                if (live && tid != P && near_best(nu2, thr, b.val, a2)) unclear = true;
                if (isA) { if (tid == P) pos = 0; else if (pos == 0) pos = ppos; }
            } else {
                P = 0;
            }
            if (tid == P) { live = false; col_of_pos[0] = tid; }
            for (int i = tid; i < r; i += CT) xv0[i] = W[(int64_t)i * ld + P];
            __syncthreads();
        }
        double d = 0.0, tsq = 0.0, ak = 0.0;   // tsq = |x_tail|^2, accumulated by every column thread itself
        if (isA) {
            ak = wc[0];
            for (int i = 1; i < r; ++i) { const double xi = xv0[i]; d = fma(xi, wc[(int64_t)i * ld], d); tsq = fma(xi, xi, tsq); }
        }

        for (int k = 0; k < c; ++k) {
            double* xc = (k & 1) ? xv1 : xv0;     // x of this step
            double* xn = (k & 1) ? xv0 : xv1;     // x of the next one
            const bool active = live || tid == P; // columns that still take part in step k
            // ---- makeHouseholder in the un-normalised form of bdqr_pair.hip:
            // nb_ = -beta = copysign(norm, x0), s = -w = nb_ + x0 (= x0 - beta), ng = -1/(beta w); degenerate -> H = I
            const double xk = xc[k];
            double nb_, s, ng;
            const bool degen = !(tsq > DBL_MIN);
            if (isA && active) { // (every column thread that took part in the last sweep accumulated the same |x_tail|^2; the threads
                                 // of columns chosen earlier carry tsq = 0 and must not read it as a degenerate reflector)
                if (k == 0 && !pivoting) a2 = fma(xk, xk, tsq);
                if (unclear_reflector(xk, tsq, k + 1 < r, pivoting != 0, a2)) unclear = true;
            }
            if (degen) { nb_ = -xk; s = 0.0; ng = 0.0; }
            else {
                const double nrm = sqrt(fma(xk, xk, tsq));
                nb_ = xk >= 0.0 ? nrm : -nrm;       // Eigen: if (c0 >= 0) beta = -beta  (-0.0 counts as >= 0)
                s = nb_ + xk;
                ng = -1.0 / (nb_ * s);
            }
            const double ngam = active ? fma(s, ak, d) * ng : 0.0;    // -gamma of this column
            double an = fma(s, ngam, ak);
            if (tid == P && !degen) an = -nb_;                        // R(k,k) = beta
            if (active) wc[(int64_t)k * ld] = an;                     // row k of R
            if (tid == P) {
                const double tau = -(s * s) * ng;                     // w / beta
                taus[k] = tau;
                if (hcoeffs) hcoeffs[cbase + k] = tau;
            }
            const double inv_s = degen ? 0.0 : 1.0 / s;               // essential part = x_tail / (x0 - beta), 0 if H = I

            if (k + 1 == c) {
                if (tid == P) for (int i = k + 1; i < r; ++i) wc[(int64_t)i * ld] = xc[i] * inv_s;
                break;
            }

            // ---- LAWN-176 norm downdate (squared form, see bdqr_pair.hip) and the search for step k+1
            bool need = false;
            if (pivoting) {
                if (live) {
                    const double nn = fma(-an, an, nu2);
                    nu2 = nn;
                    need = nn <= thr;
                    if (need && in_recompute_band(nn, thr, a2)) unclear = true;    // (2) inside the band around Eigen's threshold
                }
                if (need) flags[k & 1] = 1;
                Cand cd{live ? nu2 : -1.0, pos, tid, ngam};
                cd = wave_best(cd);
                if (lane == 0) { cval[wave] = cd.val; cpos[wave] = cd.pos; ctid[wave] = cd.tidx; cngam[wave] = cd.ngam; }
                if (tid == 0) flags[(k + 1) & 1] = 0;
            }
            __syncthreads();
            const bool any_need = pivoting && flags[k & 1] != 0;
            int Pn;                          // pivot thread of step k+1
            int ppos = k + 1;
            double ngP = 0.0;

            if (!any_need) {
                if (pivoting) {
                    Cand b{cval[0], cpos[0], ctid[0], cngam[0]};
#pragma unroll
                    for (int w = 1; w < NW; ++w) { Cand o{cval[w], cpos[w], ctid[w], cngam[w]}; if (better(o, b)) b = o; }
                    Pn = b.tidx; ppos = b.pos; ngP = b.ngam;
                    if (live && tid != Pn && near_best(nu2, thr, b.val, a2)) unclear = true;
                } else {
                    Pn = k + 1;
                    if (tid == Pn) cngam[0] = ngam;   // -gamma of column k+1 is only known to its thread
                    __syncthreads();
                    ngP = cngam[0];
                }
                // x' = column Pn after update k, built by the threads row-wise
                for (int i = k + 1 + tid; i < r; i += CT) xn[i] = fma(ngP, xc[i], W[(int64_t)i * ld + Pn]);
                __syncthreads();
                // ---- fused sweep: apply update k, accumulate the dot products of step k+1
                d = 0.0; tsq = 0.0;
                if (tid == P) {
                    for (int i = k + 1; i < r; ++i) wc[(int64_t)i * ld] = xc[i] * inv_s;   // the reflector replaces the column
                } else if (live) {
                    int i = k + 1;
                    {
                        const double w0 = fma(ngam, xc[i], wc[(int64_t)i * ld]);
                        wc[(int64_t)i * ld] = w0;
                        ak = w0;
                    }
                    // rows in chunks of U with all U loads issued first: the sweep is bound by the latency of
                    // the loads (LDS or L2/Infinity Cache), not by bandwidth
                    constexpr int U = 16;
                    for (i = k + 2; i + U <= r; i += U) {
                        double wv[U];
#pragma unroll
                        for (int u = 0; u < U; ++u) wv[u] = wc[(int64_t)(i + u) * ld];
#pragma unroll
                        for (int u = 0; u < U; ++u) {
                            const double w0 = fma(ngam, xc[i + u], wv[u]);
                            wc[(int64_t)(i + u) * ld] = w0;
                            const double x0 = xn[i + u];
                            d = fma(x0, w0, d);
                            tsq = fma(x0, x0, tsq);
                        }
                    }
                    for (; i < r; ++i) {
                        const double w0 = fma(ngam, xc[i], wc[(int64_t)i * ld]);
                        wc[(int64_t)i * ld] = w0;
                        const double x0 = xn[i];
                        d = fma(x0, w0, d);
                        tsq = fma(x0, x0, tsq);
                    }
                }
            } else {
                // ---- rare: a column norm has to be recomputed from the updated column before the search
                double s2 = 0.0;
                if (tid == P) {
                    for (int i = k + 1; i < r; ++i) wc[(int64_t)i * ld] = xc[i] * inv_s;
                } else if (live) {
                    for (int i = k + 1; i < r; ++i) {
                        const double w0 = fma(ngam, xc[i], wc[(int64_t)i * ld]);
                        wc[(int64_t)i * ld] = w0;
                        s2 = fma(w0, w0, s2);
                    }
                }
                if (need) { nu2 = s2; thr = s2 * THR_HI; }
                __syncthreads();
                Cand cd{live ? nu2 : -1.0, pos, tid, 0.0};
                cd = wave_best(cd);
                if (lane == 0) { cval[wave] = cd.val; cpos[wave] = cd.pos; ctid[wave] = cd.tidx; }
                __syncthreads();
                Cand b{cval[0], cpos[0], ctid[0], 0.0};
#pragma unroll
                for (int w = 1; w < NW; ++w) { Cand o{cval[w], cpos[w], ctid[w], 0.0}; if (better(o, b)) b = o; }
                Pn = b.tidx; ppos = b.pos;
                if (live && tid != Pn && near_best(nu2, thr, b.val, a2)) unclear = true;
                for (int i = k + 1 + tid; i < r; i += CT) xn[i] = W[(int64_t)i * ld + Pn];
                __syncthreads();
                d = 0.0; tsq = 0.0;
                if (live) {
                    ak = wc[(int64_t)(k + 1) * ld];
                    for (int i = k + 2; i < r; ++i) { const double xi = xn[i]; d = fma(xi, wc[(int64_t)i * ld], d); tsq = fma(xi, xi, tsq); }
                }
            }
            // Eigen swaps columns k+1 and the pivot: the column that sat at k+1 takes the pivot's place
            if (isA && pivoting) { if (tid == Pn) pos = k + 1; else if (pos == k + 1) pos = ppos; }
            if (tid == Pn) { live = false; col_of_pos[k + 1] = tid; }
            P = Pn;
            __syncthreads();     // every column is up to date before the next x' is read across threads
        }
        __syncthreads();

        }
        // ---- a decision inside its error margin: the tile is redone by the exact path
        if (unclear) flags[2] = 1;
        __syncthreads();
        if (tid == 0 && flags[2] != 0 && redo_count) redo_ids[atomicAdd(redo_count, 1)] = gid;
        colfin::finish_tile<CT>(W, ld, r, c, cbase, col_of_pos, taus, vs, gm, tm, Q, rv, perm);
        }
    }
}

// Two instantiations per thread count.  BIG (launches whose largest tile has more than 160 rows): the register
// allocator's own choice (~210 VGPRs, two waves per SIMD) and the fixed LDS layout with compile-time offsets - two
// workgroups of 256 threads per CU is all their 80 KB allow anyway.  Otherwise 168 VGPRs (three waves per SIMD, ~20
// spilled) and LDS carved for the largest tile of the launch: 96x96 533k -> 757k tiles/s, 128x128 326k -> 443k, while
// the same settings cost the 192...256 tiles 3 % and a mixed 8...256 batch 6 %.
template <int CT, bool BIG>
__global__ void __launch_bounds__(CT) __attribute__((amdgpu_waves_per_eu(BIG ? QRK_COL_BIGW : 3)))
bdqr_col_kernel(WaveBatch nb, const double* __restrict__ tiles, double* __restrict__ q_vals,
                double* __restrict__ r_vals, int32_t* __restrict__ perm, double* __restrict__ hcoeffs,
                double* __restrict__ workspace, int64_t ws_stride, int w_lds, int max_r,
                int32_t* __restrict__ redo_count, int32_t* __restrict__ redo_ids, int32_t* __restrict__ queue)
{
    extern __shared__ __attribute__((aligned(16))) double smem[];
    // Tiles are handed out through a counter (zeroed by the launcher): with the list sorted largest first that is the
    // longest-processing-time rule.  (A grid-stride assignment gave workgroup 0 the largest tile of every round.)
    __shared__ int next_tile;
    for (int64_t t = blockIdx.x; t < nb.num_tiles;) {
        const int gidx = nb.tile_ids ? nb.tile_ids[t] : (int)t;
        int r, c, cbase;
        int64_t toff, qoff, roff;
        if (nb.t_rows) {
            r = nb.t_rows[gidx]; c = nb.t_cols[gidx];
            toff = nb.t_off[gidx]; qoff = nb.q_off[gidx]; roff = nb.r_off[gidx]; cbase = nb.c_off[gidx];
        } else {
            r = nb.rows; c = nb.cols;
            toff = t * (int64_t)r * c; qoff = t * (int64_t)r * r; roff = t * (int64_t)(c * (c + 1) / 2);
            cbase = (int)(t * c);
        }
        constexpr bool DYN = !BIG;
        if (r * c <= (DYN ? w_lds : col::W_LDS_DOUBLES))
            factor_tile<CT, false, DYN>(smem, smem, r, c, cbase, nb.pivoting, tiles + toff, q_vals + qoff, r_vals + roff, perm, hcoeffs,
                                        w_lds, max_r, redo_count, redo_ids, gidx);
        else
            factor_tile<CT, QRK_COL_BLOCKED != 0, DYN, (BIG && QRK_COL_BLOCKED != 0 && QRK_COL_REGROWS > 0)>(workspace + (int64_t)blockIdx.x * ws_stride, smem, r, c, cbase, nb.pivoting,
                                                       tiles + toff, q_vals + qoff, r_vals + roff, perm, hcoeffs, w_lds, max_r,
                                                       redo_count, redo_ids, gidx);
        __syncthreads();
        if (threadIdx.x == 0) next_tile = (int)gridDim.x + atomicAdd(queue, 1);
        __syncthreads();
        t = next_tile;
    }
}

bool bdqr_col_big(int max_rows) { return max_rows > 160; }

size_t bdqr_col_smem_bytes(int threads, int w_lds, int max_r)
{
    const int nw = threads / 64;
    if (bdqr_col_big(max_r)) { w_lds = col::W_LDS_DOUBLES; max_r = col::MAXR; }   // fixed layout (see bdqr_col_kernel)
    return (size_t)(w_lds + max_r * (col::NB + 1) + 3 * max_r + 2 * col::NB * col::NB + 2 * nw) * sizeof(double) +
           (size_t)(2 * nw + 4 + max_r) * sizeof(int) + 16;
}

bool bdqr_col_big(int max_rows);

int bdqr_col_threads(int max_cols, int max_rows)
{
    if (bdqr_col_big(max_rows)) return 256;        // (only that instantiation exists for the large class)
    if (const char* e = std::getenv("QRK_COL_THREADS")) { const int v = std::atoi(e); if (v == 64 || v == 128 || v == 256) return v >= max_cols ? v : 256; }
    return max_cols <= 64 ? 64 : (max_cols <= 128 ? 128 : 256);
}

// LDS doubles for A in a launch whose largest tile holds max_rc entries: the tile itself when it fits the budget of
// the LDS-resident form, else nothing at all unless smaller tiles of the launch can use it (mixed launches)
int bdqr_col_w_lds(int64_t max_rc, int64_t max_rc_fitting)
{
    if (max_rc <= col::W_LDS_DOUBLES) return (int)max_rc;
    return (int)max_rc_fitting;     // largest r*c <= W_LDS_DOUBLES present in the launch (0 if none)
}

// Workgroups of one launch that can be resident per CU (LDS and the 16 waves a CU holds at this kernel's register count)
int bdqr_col_wgs_per_cu(int max_cols, int w_lds, int max_r)
{
    const int threads = bdqr_col_threads(max_cols, max_r);
    const size_t smem = bdqr_col_smem_bytes(threads, w_lds, max_r);
    // waves a CU holds at the register count of the instantiation (see bdqr_col_kernel)
    const int cu_waves = bdqr_col_big(max_r) ? 8 : 12;
    int by_lds = (int)((size_t)160 * 1024 / smem), by_waves = cu_waves / (threads / 64);
    int n = by_lds < by_waves ? by_lds : by_waves;
    // tiles wider than 64 columns work mostly in global memory: more than two workgroups per CU only makes them
    // slower (measured: 256x256 50.0k -> 46.3k tiles/s, 96x96 475k -> 450k at four); the small ones gain (33x33 2.9M -> 5.2M)
    if (const char* e = std::getenv("QRK_COL_WGS_PER_CU")) { const int v = std::atoi(e); if (v > 0 && v < n) n = v; }
    return n < 1 ? 1 : n;
}

hipError_t launch_bdqr_col(const WaveBatch& nb, const double* tiles, double* q_vals, double* r_vals, int32_t* perm,
                           double* hcoeffs, double* workspace, int64_t ws_stride, int num_wg, int max_rows,
                           int max_cols, int w_lds, int32_t* redo_count, int32_t* redo_ids, int32_t* queue, hipStream_t stream)
{
    if (nb.num_tiles <= 0) return hipSuccess;
    if (hipError_t e = hipMemsetAsync(queue, 0, sizeof(int32_t), stream)) return e;
    if (max_rows > col::MAXR || max_cols > max_rows || w_lds > col::W_LDS_DOUBLES) return hipErrorInvalidValue;
    const int64_t want = nb.num_tiles < (int64_t)num_wg ? nb.num_tiles : (int64_t)num_wg;
    const int threads = bdqr_col_threads(max_cols, max_rows);
    const size_t smem = bdqr_col_smem_bytes(threads, w_lds, max_rows);
#define QRK_COL_LAUNCH(T, G)                                                                                \
    do {                                                                                                    \
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(bdqr_col_kernel<T, G>),            \
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);          \
        if (e != hipSuccess) return e;                                                                      \
        hipLaunchKernelGGL((bdqr_col_kernel<T, G>), dim3((unsigned)want), dim3(T), smem, stream, nb, tiles, q_vals, \
                           r_vals, perm, hcoeffs, workspace, ws_stride, w_lds, max_rows, redo_count, redo_ids, queue); \
    } while (0)
    if (bdqr_col_big(max_rows)) QRK_COL_LAUNCH(256, true);      // (more than 160 rows >= cols: 256 threads unless the tiles are narrow)
    else if (threads == 64) QRK_COL_LAUNCH(64, false);
    else if (threads == 128) QRK_COL_LAUNCH(128, false);
    else QRK_COL_LAUNCH(256, false);
#undef QRK_COL_LAUNCH
    return hipGetLastError();
}

}  // namespace qrk
