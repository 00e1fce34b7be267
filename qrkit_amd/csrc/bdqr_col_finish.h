// bdqr_col_finish.h -- the second half of a mid-size tile (32 < max dim <= 256), shared by the kernels whose phase 1 leaves the
// packed factorisation in a row-major working copy W (bdqr_col.hip: LDS or global workspace; bdqr_reg.hip: registers + LDS, dumped
// to the workspace): R in the packed CSC value order of m_R, the permutation splice, and Q = H_0 ... H_{c-1}
// (HouseholderSequence::evalTo; src/QRKit/BlockDiagonalSparseQR.h:455-492, 519-521) by blocked backward accumulation on the matrix cores.
#pragma once
#include "qrk_device.h"

namespace qrk {
namespace colfin {
constexpr int NB = 16;                                // reflectors per block in the formation of Q

// W(i, j) = W[i * ld + j]: rows 0..p of the column chosen at step p hold R(0:p, p), the rows below the essential part of reflector p.
// col_of_pos [c], taus [c]: LDS.  vs [r * (NB + 1)], gm / tm [NB * NB]: LDS scratch.  CT threads, all of them call.
template <int CT>
__device__ __forceinline__ void finish_tile(const double* __restrict__ W, const int ld, const int r, const int c, const int cbase,
                                            const int* col_of_pos, const double* taus, double* vs, double* gm, double* tm,
                                            double* __restrict__ Q, double* __restrict__ rv, int32_t* __restrict__ perm)
{
    constexpr int NW = CT / 64;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    {
        {
        // ---- R (packed upper triangle by columns = CSC value order of m_R) and the permutation splice:
        // row i of R is row i of W; the column at position p is col_of_pos[p].
        for (int p = tid; p < c; p += CT) perm[cbase + p] = cbase + col_of_pos[p];   // m_outputPerm_c.indices() (:519-521)
        for (int p = wave; p < c; p += NW) {
            const int tc = col_of_pos[p];
            for (int i = lane; i <= p; i += 64) rv[(int64_t)p * (p + 1) / 2 + i] = W[(int64_t)i * ld + tc];
        }

        // ================= phase 2: Q = H_0 ... H_{c-1}, blocked backward accumulation =================
        for (int e = tid; e < r * r; e += CT) { const int i = e / r; Q[e] = (e - i * r == i) ? 1.0 : 0.0; }
        __syncthreads();
#ifdef QRK_COL_SKIP_Q
        if (r > 0) return;   // diagnostic: time phase 1 alone
#endif
        for (int kp = ((c - 1) / NB) * NB; kp >= 0; kp -= NB) {
            const int kb = (c - kp) < NB ? (c - kp) : NB;
            const int m = r - kp;
            // V panel (m x kb, unit lower trapezoidal) to LDS, row-major with stride NB + 1 (the MFMA operand reads below
            // walk it both by rows and by columns)
            constexpr int VS = NB + 1;
            for (int e = tid; e < m * NB; e += CT) {
                const int i = e / NB, l = e - i * NB;
                double v = 0.0;
                if (l < kb) {
                    if (i == l) v = 1.0;
                    else if (i > l) v = W[(int64_t)(kp + i) * ld + col_of_pos[kp + l]];
                }
                vs[i * VS + l] = v;
            }
            __syncthreads();
            // G = V^T V (upper part), one pair per thread
            for (int e = tid; e < NB * NB; e += CT) {
                const int a = e / NB, b = e - a * NB;
                double g = 0.0;
                if (a <= b && b < kb) for (int i = b; i < m; ++i) g = fma(vs[i * VS + a], vs[i * VS + b], g);
                gm[e] = g;
            }
            __syncthreads();
            // T (forward, columnwise -- LAPACK larft): T(l,l) = tau_l, T(0:l,l) = -tau_l T(0:l,0:l) (V(:,0:l)^T v_l)
            if (tid < NB) {
                const int a = tid;
                for (int l = 0; l < NB; ++l) tm[a * NB + l] = 0.0;
                for (int l = 0; l < kb; ++l) {
                    const double tau = taus[kp + l];
                    double tv = 0.0;
                    if (a == l) tv = tau;
                    else if (a < l) {
                        double acc = 0.0;
                        for (int b = a; b < l; ++b) acc = fma(tm[a * NB + b], gm[b * NB + l], acc);
                        tv = -tau * acc;
                    }
                    tm[a * NB + l] = tv;     // row a only depends on row a: no synchronisation needed
                }
            }
            __syncthreads();
            // Q(kp:, kp:) <- (I - V T V^T) Q(kp:, kp:) with v_mfma_f64_16x16x4_f64, a wave per strip of 16 columns of Q:
            //   w = V^T q   A[row = lane & 15][k = lane >> 4] = V(4 k' + k, row) from LDS, B = Q(4 k' + k, col) from memory
            //   u = -T w    the result registers D[row = (lane >> 4) + 4 z][col] of w are the B operand of k-step z
            //   q += V u    A = V(16 t + row, 4 k' + k), B = u, D = the 16 x 16 tile of Q, read-modify-write
            // (one thread per column with scalar FMAs spent 3.8 ms of a 256 x 256 tile's 10 ms here)
            {
                typedef double d4 __attribute__((ext_vector_type(4)));
                const int kq = lane >> 4, l15 = lane & 15;
                const int S = (m + 15) >> 4, K = (m + 3) >> 2;
                for (int sidx = wave; sidx < S; sidx += NW) {
                    const int colq = kp + 16 * sidx + l15;
                    const bool cok = colq < r;
                    double* qc = Q + (int64_t)kp * r + colq;              // qc[i * r] = Q(kp + i, colq)
                    d4 acc = d4{0.0, 0.0, 0.0, 0.0};
                    constexpr int U = 8;
                    for (int k = 0; k < K; k += U) {
                        double bv[U];
#pragma unroll
                        for (int u2 = 0; u2 < U; ++u2) {
                            const int row = 4 * (k + u2) + kq;
                            bv[u2] = (row < m && cok) ? qc[(int64_t)row * r] : 0.0;
                        }
#pragma unroll
                        for (int u2 = 0; u2 < U; ++u2) {
                            int row = 4 * (k + u2) + kq; if (row > m - 1) row = m - 1;      // (bv is zero beyond m)
                            acc = __builtin_amdgcn_mfma_f64_16x16x4f64(vs[row * VS + l15], bv[u2], acc, 0, 0, 0);
                        }
                    }
                    d4 uu = d4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
                    for (int ks = 0; ks < 4; ++ks)
                        uu = __builtin_amdgcn_mfma_f64_16x16x4f64(tm[l15 * NB + 4 * ks + kq], acc[ks], uu, 0, 0, 0);
                    uu = -uu;
                    const int RT = (m + 15) >> 4;
                    constexpr int UT = 2;
                    for (int rt = 0; rt < RT; rt += UT) {
                        d4 dv[UT];
#pragma unroll
                        for (int u2 = 0; u2 < UT; ++u2)
#pragma unroll
                            for (int z = 0; z < 4; ++z) {
                                const int row = 16 * (rt + u2) + kq + 4 * z;
                                dv[u2][z] = (row < m && cok) ? qc[(int64_t)row * r] : 0.0;
                            }
#pragma unroll
                        for (int u2 = 0; u2 < UT; ++u2) {
                            int arow = 16 * (rt + u2) + l15; if (arow > m - 1) arow = m - 1;
#pragma unroll
                            for (int ks = 0; ks < 4; ++ks)
                                dv[u2] = __builtin_amdgcn_mfma_f64_16x16x4f64(vs[arow * VS + 4 * ks + kq], uu[ks], dv[u2], 0, 0, 0);
#pragma unroll
                            for (int z = 0; z < 4; ++z) {
                                const int row = 16 * (rt + u2) + kq + 4 * z;
                                if (row < m && cok) qc[(int64_t)row * r] = dv[u2][z];
                            }
                        }
                    }
                }
            }
            __syncthreads();
        }
        __syncthreads();
        }
    }
}

// The same for the on-chip kernel (bdqr_reg.hip: rows <= 256, 512 threads, no other register load at this point).  The round trips to
// memory are what the accumulation above spends its time on (eight Q loads in flight, twice per strip and panel: 0.85 ms of a
// 256 x 256 tile), and guards per element are what hipcc turns into one branch (and one wait) per load, so here
//   * Q is accumulated in a PADDED 256 x 256 frame of the workgroup's workspace (row stride 256: every address is a scalar base plus
//     one lane offset, no guard anywhere; rows and columns past the tile only ever hold identity) and copied out once at the end;
//   * a strip of 16 columns of Q(kp:, kp:) stays in REGISTERS from its first use to its store (<= 64 doubles per lane, in the
//     accumulator layout of v_mfma_f64_16x16x4_f64, which is also the B-operand layout of w = V^T q): one batch of loads per strip
//     and panel, no second read;
//   * the frame is never initialised: the part of Q(kp:, kp:) that no later panel has written yet is the identity, generated in the
//     registers (first panel: everything; then the leading 16 rows / columns) -- the last panel (kp = 0) writes every entry;
//   * V^T V comes from the matrix cores too (A and B operand are the same register: V(4 k' + lane / 16, lane % 16)), the k-steps
//     dealt over the waves and summed through LDS; T with its row in registers.
// Inputs, as bdqr_reg.hip leaves them in the workspace: Rw[k * 256 + j] = row k of R in ORIGINAL column order; Vb[k * 256 + i] =
// essential part of reflector k at row i of the tile (i > k; anything elsewhere).  Qp: the frame.
// vs [256 * (NB + 1)], gm [NB * NB], tm [NB * NB], gp [CT / 64][NB * NB]: LDS scratch -- TWICE each with PAIR (two panels of 16
// reflectors per pass over Q: half the loads and stores of the frame, half the barriers; 512 threads).
template <int CT, bool PAIR>
__device__ __forceinline__ void finish_tile_strips(const double* __restrict__ Rw, const double* __restrict__ Vb, double* __restrict__ Qp,
                                                   const int r, const int c, const int cbase, const int* col_of_pos, const double* taus,
                                                   double* vs, double* gm, double* tm, double* gp, double* __restrict__ Q,
                                                   double* __restrict__ rv, int32_t* __restrict__ perm)
{
    typedef double d4 __attribute__((ext_vector_type(4)));
    constexpr int NW = CT / 64, VS = NB + 1, MAXT = 16, LD = 256;
    const int tid0 = threadIdx.x, wave = tid0 >> 6;
    // (per-lane values are re-derived from an opaque thread id inside every loop body: hipcc otherwise hoists the loop-invariant
    //  address arithmetic of all unrolled accesses out of the loops -- and of the tile loop around this call -- and spills it)
#define QRK_FIN_LANE() int tid = threadIdx.x; asm volatile("" : "+v"(tid)); const int lane = tid & 63, kq = lane >> 4, l15 = lane & 15; (void)lane; (void)kq; (void)l15
#ifdef QRK_REG_PROF
    unsigned long long ft[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, ft0 = __builtin_amdgcn_s_memtime();
#define FIN_TICK(z) do { const unsigned long long t1 = __builtin_amdgcn_s_memtime(); ft[z] += t1 - ft0; ft0 = t1; } while (0)
#else
#define FIN_TICK(z) do { } while (0)
#endif
    for (int p = tid0; p < c; p += CT) perm[cbase + p] = cbase + col_of_pos[p];   // m_outputPerm_c.indices() (:519-521)
    // R: column p of the packed triangle is column col_of_pos[p] of Rw, rows 0..p.  Read by rows (coalesced), sixteen rows at a time
    // through LDS (vs: [256][17]), written by columns in runs of sixteen (a column of Rw straight from memory is one cache line --
    // and, at a stride of 2 KB, one L2 channel -- per element: 0.1 ms of a 256 x 256 tile)
    for (int ib = 0; 16 * ib < c; ++ib) {
        QRK_FIN_LANE();
        constexpr int PT = 16 * LD / CT;            // elements per thread of a block of 16 rows
        double rowv[PT];
#pragma unroll
        for (int u = 0; u < PT; ++u) { const int e = tid + CT * u, ii = e >> 8, jj = e & 255; rowv[u] = Rw[(16 * ib + ii) * LD + jj]; }
#pragma unroll
        for (int u = 0; u < PT; ++u) { const int e = tid + CT * u, ii = e >> 8, jj = e & 255; vs[jj * VS + ii] = rowv[u]; }
        __syncthreads();
        {
            const int pp = tid >> 4, ii = tid & 15, i = 16 * ib + ii;
#pragma unroll
            for (int u = 0; u < PT; ++u) {
                const int p = 16 * ib + pp + (CT / 16) * u;
                if (p < c && i <= p) rv[(int64_t)p * (p + 1) / 2 + i] = vs[col_of_pos[p] * VS + ii];
            }
        }
        __syncthreads();
    }
    FIN_TICK(0);
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);      // (uniform for the compiler too: scalar bases below)
    // LDS scratch per panel of a pass: V [256][VS], G, T, and the waves' partial G's
    constexpr int NP = PAIR ? 2 : 1;
    double* const vsp[2] = {vs, vs + 256 * VS};
    double* const gmp[2] = {gm, gm + NB * NB};
    double* const tmp_[2] = {tm, tm + NB * NB};
    double* const gpp[2] = {gp, gp + NW * NB * NB};
    const int npanels = (c + NB - 1) / NB;
    bool first = true;
    for (int ptop = npanels - 1; ptop >= 0;) {
        // a pass takes TWO panels (PAIR, and an even number of them left) or one: the strip of Q is loaded once, the upper panel's
        // block reflector (rows kp + 16 ..) and then the lower one's (rows kp ..) are applied to it in registers, and it is stored once
        const int npan = (PAIR && (ptop & 1)) ? 2 : 1;
        const int kp = (ptop - (npan - 1)) * NB;   // first row / column of the pass (the lower panel)
        const int m = r - kp;
        const int mt = (m + 15) >> 4;              // row tiles of a strip; V is zero-filled up to whole tiles
        const int wb = kp + NB * npan;             // Q(wb:, wb:) is what earlier passes have written (unless this is the first)
        QRK_FIN_LANE();
        // ---- V of the panels: panel x = 0 is the LOWER one (rows kp ..), x = 1 the upper (rows kp + 16 ..)
#pragma unroll
        for (int x = 0; x < NP; ++x)
            if (x < npan) {
                const int kx = kp + NB * x, kb = (c - kx) < NB ? (c - kx) : NB, mx = r - kx, mtx = (mx + 15) >> 4;
                // thread -> reflector l = tid / RPT, rows i = tid % RPT + RPT u: independent loads (a loop with one load per trip is a
                // chain of memory latencies)
                constexpr int RPT = CT / NB, NU = LD / RPT;
                const int l = tid / RPT, i0 = tid % RPT;
                double vv[NU];
#pragma unroll
                for (int u = 0; u < NU; ++u) {
                    const int i = i0 + RPT * u;
                    vv[u] = (l < kb && i > l && i < mx) ? Vb[(kx + l) * LD + kx + i] : (i == l && l < kb ? 1.0 : 0.0);
                }
#pragma unroll
                for (int u = 0; u < NU; ++u) { const int i = i0 + RPT * u; if (i < 16 * mtx) vsp[x][i * VS + l] = vv[u]; }
            }
        __syncthreads();
        FIN_TICK(1);
#pragma unroll
        for (int x = 0; x < NP; ++x)
            if (x < npan) {
                const int mtx = mt - x;
                d4 g = d4{0.0, 0.0, 0.0, 0.0};
                for (int k = wave_u; k < 4 * mtx; k += NW) {
                    const double v = vsp[x][(4 * k + kq) * VS + l15];
                    g = __builtin_amdgcn_mfma_f64_16x16x4f64(v, v, g, 0, 0, 0);
                }
#pragma unroll
                for (int z = 0; z < 4; ++z) gpp[x][wave * NB * NB + (kq + 4 * z) * NB + l15] = g[z];   // G(row = kq + 4 z, col = l15)
            }
        __syncthreads();
        // the first 256 threads sum the partial G's of the lower panel, the next 256 (if there are any) those of the upper one
        {
            const int x = PAIR ? (tid >> 8) : 0;
            if (tid < NB * NB * NP && x < npan) {
                const int e = tid & 255;
                double g = 0.0;
#pragma unroll
                for (int w = 0; w < NW; ++w) g += gpp[x][w * NB * NB + e];
                gmp[x][e] = g;
            }
        }
        __syncthreads();
        FIN_TICK(2);
        // T (forward, columnwise -- LAPACK larft): T(l,l) = tau_l, T(0:l,l) = -tau_l T(0:l,0:l) (V(:,0:l)^T v_l).  Lane (a, b) of four
        // waves holds T(a, b) in a register; column l is sixteen products T(a, b) G(b, l) summed over b inside the row of 16 lanes by
        // DPP: no LDS inside the recurrence (one thread per row, reading T and G from LDS, spent 12 000 cycles per panel).  With two
        // panels the second group of four waves does the upper one at the same time.
        {
            const int x = PAIR ? (tid >> 8) : 0;
            if (tid < NB * NB * NP && x < npan) {           // (whole waves: the DPP sums need every lane of their rows)
                const int kx = kp + NB * x, kb = (c - kx) < NB ? (c - kx) : NB;
                const int ta = (tid & 255) >> 4, tb = tid & 15;
                double g[NB];
#pragma unroll
                for (int l = 0; l < NB; ++l) g[l] = gmp[x][tb * NB + l];
                double tval = 0.0;
#pragma unroll
                for (int l = 0; l < NB; ++l) {
                    const double tau = l < kb ? taus[kx + l] : 0.0;
                    double sum = tb < l ? tval * g[l] : 0.0;
                    sum += dpp_f64<0xB1>(sum);
                    sum += dpp_f64<0x4E>(sum);
                    sum += dpp_f64<0x141>(sum);
                    sum += dpp_f64<0x140>(sum);
                    if (tb == l) tval = ta == l ? tau : (ta < l ? -tau * sum : 0.0);
                }
                tmp_[x][ta * NB + tb] = tval;
            }
        }
        __syncthreads();
        FIN_TICK(3);
        for (int sidx = wave_u; sidx < mt; sidx += NW) {
            QRK_FIN_LANE();
            double* qs = Qp + (kp * LD + kp + 16 * sidx);          // uniform: Q(kp + i, kp + 16 sidx + l15) = qs[i * LD + l15]
            const int loff = kq * LD + l15;
            d4 dv[MAXT];
            // what no earlier pass has written -- rows or columns below wb; everything, in the first pass -- is the identity
            if (!first && kp + 16 * sidx >= wb) {
#pragma unroll
                for (int rt = 0; rt < MAXT; ++rt)
                    if (rt < mt) {
                        if (rt < npan) {
#pragma unroll
                            for (int z = 0; z < 4; ++z) dv[rt][z] = 0.0;
                        } else {
#pragma unroll
                            for (int z = 0; z < 4; ++z) dv[rt][z] = (qs + (16 * rt + 4 * z) * LD)[loff];
                        }
                    }
            } else {
#pragma unroll
                for (int rt = 0; rt < MAXT; ++rt)
#pragma unroll
                    for (int z = 0; z < 4; ++z) dv[rt][z] = (16 * rt + kq + 4 * z == 16 * sidx + l15) ? 1.0 : 0.0;
            }
            FIN_TICK(6);
            // the upper panel first (Q <- H_lower (H_upper Q)): rows kp + 16 .. = tiles 1 .., columns kp + 16 .. = strips 1 ..
            if (PAIR && npan == 2 && sidx >= 1) {
                const double* v2 = vsp[1];
                d4 acc = d4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
                for (int rt = 1; rt < MAXT; ++rt)
                    if (rt < mt) {
#pragma unroll
                        for (int z = 0; z < 4; ++z)
                            acc = __builtin_amdgcn_mfma_f64_16x16x4f64(v2[(16 * (rt - 1) + 4 * z + kq) * VS + l15], dv[rt][z], acc, 0, 0, 0);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                d4 uu = d4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
                for (int ks = 0; ks < 4; ++ks)
                    uu = __builtin_amdgcn_mfma_f64_16x16x4f64(tmp_[1][l15 * NB + 4 * ks + kq], acc[ks], uu, 0, 0, 0);
                uu = -uu;
#pragma unroll
                for (int rt = 1; rt < MAXT; ++rt)
                    if (rt < mt) {
#pragma unroll
                        for (int ks = 0; ks < 4; ++ks)
                            dv[rt] = __builtin_amdgcn_mfma_f64_16x16x4f64(v2[(16 * (rt - 1) + l15) * VS + 4 * ks + kq], uu[ks], dv[rt], 0, 0, 0);
                        __builtin_amdgcn_sched_barrier(0);
                    }
            }
            const double* v1 = vsp[0];
            d4 acc = d4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int rt = 0; rt < MAXT; ++rt)
                if (rt < mt) {
#pragma unroll
                    for (int z = 0; z < 4; ++z)
                        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(v1[(16 * rt + 4 * z + kq) * VS + l15], dv[rt][z], acc, 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);     // (left alone, the scheduler hoists all 64 operand reads ahead of the MFMAs and spills)
                }
            FIN_TICK(7);
            d4 uu = d4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int ks = 0; ks < 4; ++ks)
                uu = __builtin_amdgcn_mfma_f64_16x16x4f64(tmp_[0][l15 * NB + 4 * ks + kq], acc[ks], uu, 0, 0, 0);
            uu = -uu;
            FIN_TICK(8);
#pragma unroll
            for (int rt = 0; rt < MAXT; ++rt)
                if (rt < mt) {
#pragma unroll
                    for (int ks = 0; ks < 4; ++ks)
                        dv[rt] = __builtin_amdgcn_mfma_f64_16x16x4f64(v1[(16 * rt + l15) * VS + 4 * ks + kq], uu[ks], dv[rt], 0, 0, 0);
                    if (kp > 0) {
#pragma unroll
                        for (int z = 0; z < 4; ++z) (qs + (16 * rt + 4 * z) * LD)[loff] = dv[rt][z];
                    } else {
                        // the last pass writes every entry of Q: straight to m_Q's values (row-major Q_i; guarded stores cost no wait)
#pragma unroll
                        for (int z = 0; z < 4; ++z) {
                            const int row = 16 * rt + 4 * z + kq, col = 16 * sidx + l15;
                            if (row < r && col < r) Q[(int64_t)row * r + col] = dv[rt][z];
                        }
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
        }
        FIN_TICK(9);
        first = false;
        ptop -= npan;
        __syncthreads();
        FIN_TICK(4);
    }
    FIN_TICK(5);
#ifdef QRK_REG_PROF
    if (blockIdx.x == 0 && threadIdx.x == 0)
        printf("   finish: R + perm %llu  V panels %llu  V^T V %llu  T %llu  apply %llu (wave 0: loads issued %llu  w = V^T q %llu  u = -T w %llu  q += V u and stores %llu  waiting for the others %llu)\n", ft[0], ft[1], ft[2], ft[3], ft[4] + ft[6] + ft[7] + ft[8] + ft[9], ft[6], ft[7], ft[8], ft[9], ft[4]);
#endif
#undef FIN_TICK
#undef QRK_FIN_LANE
}
}  // namespace colfin
}  // namespace qrk
