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
}  // namespace colfin
}  // namespace qrk
