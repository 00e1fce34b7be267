// dense_qr_pers.hip -- column-pivoted (or plain) Householder QR of a square-ish dense matrix of up to 2048 x 2048 as ONE persistent
// launch over the whole chip: the matrix lives in the REGISTERS of 256 workgroups (one wavefront per column, 32 rows per lane) and
// the workgroups meet once per reflector at an XCD-hierarchical grid barrier that also elects the pivot.
//
// Where it sits: second stage of the two-stage factorisation of a tall dense right block (caqr.hip; QRKit::BlockAngularSparseQR,
// rightSolver.compute(J2.bottomRows(...)), src/QRKit/BlockAngularSparseQR.h:361-369) -- R0 P = Q1 R with Eigen's ColPivHouseholderQR
// rule on the n x n factor of the un-pivoted first stage -- and any direct call of that shape.  The alternative to the launch-per-
// reflector form of dense_qr_cols.hip (10.3 us per reflector at n = 2000, of which 1.7 are the launch boundary and the rest three
// dependent trips through memory -- norm table -> pivot column -> own column -- plus the read-modify-write of the trailing matrix).
// Round 3 priced it with a flat-counter barrier (16-22 us) and did not build it; the XCD-hierarchical barrier costs 4.1 us, 5.7 with
// the exchange of this step (profiles/r04_grid_barrier_probe.txt), so round 4 built it -- and measured 11.4 us per reflector
// (profiles/r04_k3_stage2.txt): parity-green, slower than the launches, therefore OPT-IN (QRK_DENSE_PERS=1).
//
// One step (reflector k):
//   1. every wave offers its column if it is still live (squared updated norm); the workgroup's best wave PUBLISHES its column (rows
//      k..) and a 16-byte record into the workgroup's slot -- speculatively, before the pivot is known;
//   2. the barrier: arrival on the counter of the workgroup's XCD (HW_REG_XCC_ID); the XCD's last arriver reduces the records of its
//      XCD (they sit in the L2 the XCD shares), writes the XCD's best, does the ONE agent-scope release fence of the XCD, arrives on
//      the top counter, waits for the other XCDs, reduces the eight XCD records and opens its XCD's gate with the winner's workgroup
//      id in the generation word; the others poll that word;
//   3. everybody reads the winner's column into LDS (one trip), forms |x_tail|^2 and the reflector scalars redundantly (same data,
//      same order: bitwise the same everywhere), and every wave applies the reflector to its column in registers: dot, update,
//      LAWN-176 downdate of its squared norm, recompute from the registers when Eigen's test fires.  Row k of R simply stays where
//      it is; the pivot's owner turns its column into the essential part in place.
// At the end every wave writes its column once, at its pivoted position, in Eigen's packed format.  Squared norms, un-normalised
// reflector and the decision margins are those of dense_qr_cols.hip (qrk_device.h, decide::): a decision inside its rounding margin
// sets `unclear` and the exact path redoes the factorisation.  Spins are bounded: a workgroup that waits too long sets the abort
// word, everybody leaves, and `unclear` sends the matrix through the exact path.
#include "qrk_device.h"

#include <float.h>
#include <cstdlib>

namespace qrk {
namespace pers {

using namespace decide;
constexpr int PT = 512, PW = PT / 64;          // threads, waves (= columns) per workgroup
constexpr int RPL = 32, MAXR = 64 * RPL;       // rows per lane, rows at most
constexpr int SLOT = 8 + MAXR;                 // doubles per slot: the record (norm, column | workgroup), then the column
constexpr int MAXG = 512;                      // workgroups at most (slots, lists)

struct alignas(128) Line { unsigned v; unsigned pad[31]; };
struct alignas(128) XRec { double nu2; int pc; int wg; double pad[13]; };
struct Sync {                                  // zeroed by the launcher before every launch
    Line members[8];                           // workgroups per XCD
    Line setup;                                // flat arrival counter of the setup phase
    Line cnt[8];                               // per-XCD arrival counters (monotonic)
    Line gen[8];                               // per-XCD generation words: (step + 1) << 10 | winner's workgroup
    Line top;                                  // one arrival per XCD per step
    Line abort_word;
    XRec xrec[2][8];                           // the XCDs' best of the step, by parity
    int wgs_of_xcd[8][MAXG];                   // workgroup ids by XCD (setup)
};
struct State { double a2; int unclear; };       // (same place and meaning as dense_qr_cols.hip's: the host reads `unclear`)

__device__ __forceinline__ unsigned ld_agent(const unsigned* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ bool spin_until(const unsigned* p, unsigned target, unsigned* abort_word)
{
    unsigned spins = 0;
    while ((int)(ld_agent(p) - target) < 0) {
        __builtin_amdgcn_s_sleep(1);
        if (++spins > (1u << 22) || ld_agent(abort_word)) { atomicExch(abort_word, 1u); return false; }
    }
    return true;
}
// the same sum in every lane: four DPP steps inside the rows of 16, then the four row sums through SGPRs (dense_qr_cols.hip)
__device__ __forceinline__ double wave_sum(double v)
{
    v += dpp_f64<0xB1>(v);
    v += dpp_f64<0x4E>(v);
    v += dpp_f64<0x141>(v);
    v += dpp_f64<0x140>(v);
    return (readlane_f64(v, 0) + readlane_f64(v, 16)) + (readlane_f64(v, 32) + readlane_f64(v, 48));
}
__device__ __forceinline__ bool better(double v, int pc, double bv, int bpc) { return v > bv || (v == bv && pc < bpc); }

__global__ void __launch_bounds__(PT)
dense_pers_kernel(const double* __restrict__ A, int64_t lda, int r, int c, int pivoting, double* __restrict__ hcoeffs,
                  int32_t* __restrict__ perm, State* __restrict__ st, Sync* __restrict__ sy, double* __restrict__ slots,
                  double* __restrict__ out, int64_t ldo)
{
    const int G = gridDim.x, wg = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    __shared__ double xs[MAXR];
    __shared__ double red[PW];
    __shared__ int ired[PW];
    __shared__ int s_info[5];                  // [0] workgroups of my XCD, [1] XCDs, [2] my XCD, [3] winner workgroup / -1 = abort, [4] XCDs present (mask)
    const bool piv = (pivoting & 1) != 0, sign_free = (pivoting & PIVOTING_SIGN_FREE) != 0;

    // ---- setup (once): the workgroups of my XCD, the number of XCDs that take part
    if (tid == 0) {
        unsigned xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        xcc &= 7u;
        const unsigned idx = atomicAdd(&sy->members[xcc].v, 1u);
        sy->wgs_of_xcd[xcc][idx] = wg;
        __threadfence();
        atomicAdd(&sy->setup.v, 1u);
        const bool ok = spin_until(&sy->setup.v, (unsigned)G, &sy->abort_word.v);
        __threadfence();
        unsigned nx = 0, mask = 0;
        for (int x = 0; x < 8; ++x) if (ld_agent(&sy->members[x].v) != 0u) { ++nx; mask |= 1u << x; }
        const unsigned mine = ld_agent(&sy->members[xcc].v);
        s_info[0] = (int)mine; s_info[1] = (int)nx; s_info[2] = (int)xcc; s_info[3] = (ok && mine <= 64u) ? 0 : -1; s_info[4] = (int)mask;
    }
    __syncthreads();
    const int n_x = s_info[0], n_xcd = s_info[1], xcc = s_info[2], xmask = s_info[4];
    bool aborted = s_info[3] < 0;

    // ---- this wave's column: row q * 64 + lane in a[q]
    const int jc = wg * PW + wave;
    const bool has = jc < c;
    double a[RPL];
    {
        const double* col = A + (int64_t)(has ? jc : 0) * lda;
#pragma unroll
        for (int q = 0; q < RPL; ++q) { const int i = q * 64 + lane; const double v = col[i < r ? i : 0]; a[q] = (has && i < r) ? v : 0.0; }
    }
    bool live = has;
    int kstep = -1;
    double nu2 = -1.0, thr = 0.0, a2 = 0.0;
    int unclear = 0;
    if (piv) {
        double s = 0.0;
#pragma unroll
        for (int q = 0; q < RPL; ++q) s = fma(a[q], a[q], s);
        s = wave_sum(s);
        if (has) { nu2 = s; thr = s * THR_HI; }
    }
    const int size = r < c ? r : c;
    for (int k = 0; k < size && !aborted; ++k) {
        const int par = k & 1;
        // ---- 1. the workgroup's candidate, and its column into the workgroup's slot
        double cand = piv ? (live ? nu2 : -1.0) : (jc == k ? 1.0 : -1.0);
        if (lane == 0) { red[wave] = cand; ired[wave] = jc; }
        __syncthreads();
        double bv = red[0]; int bj = ired[0], bw = 0;
#pragma unroll
        for (int q = 1; q < PW; ++q) if (better(red[q], ired[q], bv, bj)) { bv = red[q]; bj = ired[q]; bw = q; }
        double* slot = slots + ((size_t)par * G + wg) * SLOT;
        if (wave == bw) {
            if (bv >= 0.0) {
#pragma unroll
                for (int q = 0; q < RPL; ++q) { const int i = q * 64 + lane; if (i >= k && i < r) slot[8 + i] = a[q]; }
            }
            if (lane == 0) { slot[0] = bv; reinterpret_cast<int*>(slot + 1)[0] = bj; reinterpret_cast<int*>(slot + 1)[1] = wg; }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __syncthreads();
        // ---- 2. the barrier, which also elects the pivot: run by the whole first wave (the records are read one per lane)
        if (wave == 0) {
            int W = 0, ok = 1;
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            unsigned old = 0;
            if (lane == 0) old = __hip_atomic_fetch_add(&sy->cnt[xcc].v, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            old = (unsigned)__builtin_amdgcn_readfirstlane((int)old);
            if (old == (unsigned)n_x * (unsigned)(k + 1) - 1u) {
                // last arriver of this XCD: the XCD's best (its records are in the L2 the XCD shares; this CU's L1 may hold the lines of two
                // steps ago)
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                double v = -2.0; int p2 = 0x7fffffff, w2 = 0;
                if (lane < n_x) {
                    w2 = sy->wgs_of_xcd[xcc][lane];
                    const double* s2 = slots + ((size_t)par * G + w2) * SLOT;
                    v = s2[0]; p2 = reinterpret_cast<const int*>(s2 + 1)[0];
                }
#pragma unroll
                for (int o = 32; o >= 1; o >>= 1) {
                    const double ov = __shfl_xor(v, o); const int op = __shfl_xor(p2, o), ow = __shfl_xor(w2, o);
                    if (better(ov, op, v, p2)) { v = ov; p2 = op; w2 = ow; }
                }
                if (lane == 0) { XRec* mine = &sy->xrec[par][xcc]; mine->nu2 = v; mine->pc = p2; mine->wg = w2; }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");          // ONE write-back of the XCD's L2 per step
                if (lane == 0) {
                    __hip_atomic_fetch_add(&sy->top.v, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    ok = spin_until(&sy->top.v, (unsigned)n_xcd * (unsigned)(k + 1), &sy->abort_word.v) ? 1 : 0;
                }
                ok = __builtin_amdgcn_readfirstlane(ok);
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                v = -2.0; p2 = 0x7fffffff; w2 = 0;
                if (lane < 8 && ((xmask >> lane) & 1)) { const XRec* o = &sy->xrec[par][lane]; v = o->nu2; p2 = o->pc; w2 = o->wg; }
#pragma unroll
                for (int o = 4; o >= 1; o >>= 1) {
                    const double ov = __shfl_xor(v, o); const int op = __shfl_xor(p2, o), ow = __shfl_xor(w2, o);
                    if (better(ov, op, v, p2)) { v = ov; p2 = op; w2 = ow; }
                }
                W = __builtin_amdgcn_readfirstlane(w2);
                if (lane == 0)
                    __hip_atomic_store(&sy->gen[xcc].v, ((unsigned)(k + 1) << 10) | (unsigned)W, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            } else {
                if (lane == 0) {
                    ok = spin_until(&sy->gen[xcc].v, (unsigned)(k + 1) << 10, &sy->abort_word.v) ? 1 : 0;
                    W = (int)(ld_agent(&sy->gen[xcc].v) & 1023u);
                }
                ok = __builtin_amdgcn_readfirstlane(ok); W = __builtin_amdgcn_readfirstlane(W);
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            }
            if (lane == 0) s_info[3] = ok ? W : -1;
        }
        __syncthreads();
        const int W = s_info[3];
        if (W < 0) { aborted = true; break; }
        // ---- 3. the winner's column into LDS, |x_tail|^2, the reflector
        const double* win = slots + ((size_t)par * G + W) * SLOT;
        const double best = win[0];
        const int pk = reinterpret_cast<const int*>(win + 1)[0];
        double t = 0.0;
        for (int i = k + tid; i < r; i += PT) {
            const double v = win[8 + i];
            xs[i - k] = v;
            if (i > k) t = fma(v, v, t);
        }
        t = wave_sum(t);
        if (lane == 0) red[wave] = t;
        __syncthreads();
        double tsq = 0.0;
#pragma unroll
        for (int q = 0; q < PW; ++q) tsq += red[q];
        const double xk = xs[0];
        double nb, s, ng, tau;
        const bool degen = !(tsq > DBL_MIN);
        if (degen) { nb = -xk; s = 0.0; ng = 0.0; tau = 0.0; }
        else {
            const double nrm = sqrt(fma(xk, xk, tsq));
            nb = xk >= 0.0 ? nrm : -nrm;
            s = nb + xk;
            ng = -1.0 / (nb * s);
            tau = -(s * s) * ng;
        }
        if (k == 0) a2 = piv ? best : fma(xk, xk, tsq);
        if (unclear_reflector(xk, tsq, k + 1 < r, piv, a2, sign_free))                   // decisions (3), (4), (5)
            unclear |= 4 | (!(tsq > DBL_MIN) ? 8 : 0) | (xk * xk <= X0_TINY2 * a2 ? 16 : 0) | (fma(xk, xk, tsq) <= PIV_TINY2 * a2 ? 32 : 0);
        // ---- 4. every wave its column
        if (jc == pk) {
            // the pivot's owner: beta on the diagonal, the essential part x_tail / (x0 - beta) below, in place
            live = false; kstep = k; nu2 = -1.0;
            const double inv = degen ? 0.0 : 1.0 / s;
#pragma unroll
            for (int q = 0; q < RPL; ++q) { const int i = q * 64 + lane; if (i == k) a[q] = -nb; else if (i > k) a[q] *= inv; }
            if (lane == 0) { hcoeffs[k] = tau; perm[k] = pk; }            // colsPermutation().indices()(k)
        } else if (live) {
            // d = x'^T a with x' = (0 .. 0, s, x_tail): the row of the diagonal rides along (no extraction of a_k from the registers)
            double d0 = 0.0, d1 = 0.0;
            double xr[RPL];
#pragma unroll
            for (int q = 0; q < RPL; ++q) { const int i = q * 64 + lane; xr[q] = (i > k && i < r) ? xs[i - k] : (i == k ? s : 0.0); }
#pragma unroll
            for (int q = 0; q < RPL; q += 2) { d0 = fma(xr[q], a[q], d0); d1 = fma(xr[q + 1], a[q + 1], d1); }
            const double ngam = wave_sum(d0 + d1) * ng;
            double q0 = 0.0, q1 = 0.0, anl = 0.0;
#pragma unroll
            for (int q = 0; q < RPL; q += 2) {
                const int i = q * 64 + lane;
                a[q] = fma(ngam, xr[q], a[q]); a[q + 1] = fma(ngam, xr[q + 1], a[q + 1]);
                if (i == k) anl = a[q]; else if (i > k) q0 = fma(a[q], a[q], q0);
                if (i + 64 == k) anl = a[q + 1]; else if (i + 64 > k) q1 = fma(a[q + 1], a[q + 1], q1);
            }
            if (piv) {
                const double an = wave_sum(anl);                          // row k of R of this column (one lane holds it)
                // decision (1): this column within the error margin of the chosen one (every remaining column has its wave here)
                if (near_best(nu2, thr, best, a2)) unclear |= 1;
                double nn = fma(-an, an, nu2);
                if (nn <= thr) {                                          // LAWN-176: recompute from the updated column, which is right here
                    if (in_recompute_band(nn, thr, a2)) unclear |= 2;     // decision (2)
                    const double sq = wave_sum(q0 + q1);
                    nn = sq; thr = sq * THR_HI;
                }
                nu2 = nn;
            }
        }
    }
    if (aborted) unclear |= 64;                // a bounded wait ran out: nothing of this launch is used, the exact path redoes the matrix
    if (lane == 0 && unclear) atomicOr(&st->unclear, unclear);
    if (wg == 0 && tid == 0) st->a2 = a2;
    // ---- Eigen's packed format in pivoted column order: the column chosen at step p is column p of `out`
    if (kstep >= 0) {
        double* dst = out + (int64_t)kstep * ldo;
#pragma unroll
        for (int q = 0; q < RPL; ++q) { const int i = q * 64 + lane; if (i < r) dst[i] = a[q]; }
    }
}

}  // namespace pers

size_t dense_pers_workspace_bytes() { return sizeof(pers::Sync) + (size_t)2 * pers::MAXG * pers::SLOT * sizeof(double) + 256; }
// rows >= cols (every column gets chosen), a column per wave of one workgroup per CU.  OPT-IN (QRK_DENSE_PERS=1): measured at
// 2000 x 2000 the persistent step costs 11.4 us against 10.4 us of a launch per reflector (profiles/r04_k3_stage2.txt) -- the barrier
// is 4.1 us, but the step around it has as many dependent trips through memory as the launch form (arrival atomic, the XCD's records,
// the XCDs' records, the generation word, the winner's column) and every XCD writes back 32 speculative columns per step.
bool dense_pers_supported(int r, int c, int num_cus)
{
    const char* e = std::getenv("QRK_DENSE_PERS");
    const bool on = e && std::atoi(e) == 1;
    const int G = num_cus < pers::MAXG ? num_cus : pers::MAXG;
    return on && r >= c && r <= pers::MAXR && c >= 256 && c <= G * pers::PW && G <= 1023;
}

// Same contract as launch_dense_qr_cols (A: input, untouched here; out: Eigen's packed QR in pivoted column order; hcoeffs, perm;
// state: the {a2, unclear} record the host reads, zeroed here).  workspace: dense_pers_workspace_bytes().
hipError_t launch_dense_qr_pers(const double* A, int64_t lda, int r, int c, int pivoting, double* hcoeffs, int32_t* perm, void* state,
                                void* workspace, int num_cus, double* out, int64_t ldo, hipStream_t stream)
{
    using namespace pers;
    Sync* sy = reinterpret_cast<Sync*>((reinterpret_cast<uintptr_t>(workspace) + 127) & ~(uintptr_t)127);
    double* slots = reinterpret_cast<double*>(sy + 1);
    State* st = static_cast<State*>(state);
    if (hipError_t e = hipMemsetAsync(sy, 0, sizeof(Sync), stream)) return e;
    if (hipError_t e = hipMemsetAsync(st, 0, sizeof(State), stream)) return e;
    int G = (c + PW - 1) / PW;
    const int cap = num_cus < MAXG ? num_cus : MAXG;
    if (G > cap) return hipErrorInvalidValue;
    void* args[] = {(void*)&A, (void*)&lda, (void*)&r, (void*)&c, (void*)&pivoting, (void*)&hcoeffs, (void*)&perm, (void*)&st, (void*)&sy,
                    (void*)&slots, (void*)&out, (void*)&ldo};
    return hipLaunchCooperativeKernel(reinterpret_cast<const void*>(dense_pers_kernel), dim3((unsigned)G), dim3(PT), args, 0, stream);
}

}  // namespace qrk
