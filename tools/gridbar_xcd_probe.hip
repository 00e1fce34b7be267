// Grid barrier cost on MI355X, XCD-hierarchical form (round 4; the flat counter of tools/gridbar_probe.hip is kept as FORM 0).
//
// FORM 1 "barrier-xcd" (MI355X_MICROARCH.md price list): every workgroup arrives on the counter of ITS XCD (the id is read from
// HW_REG_XCC_ID, never inferred from blockIdx); the last arriver of an XCD is that XCD's leader for the step: ONE agent-scope
// release fence per XCD (it writes back the whole L2 the XCD's workgroups share: their stores are in it once their own
// vmcnt has drained), one add on the top counter, a poll of the top counter, an agent-scope acquire fence, then the XCD's
// generation word; the other workgroups poll their XCD's generation word and finish with an agent-scope acquire fence.
// FORM 2: a barrier among the workgroups of ONE XCD only (counter + generation word of that XCD, no top level): the price of
// synchronising 32 CUs that share an L2.
//
// Modes as in gridbar_probe.hip: 0 barrier only; 1 every workgroup publishes `slot` bytes before the barrier; 2 publish and, after
// the barrier, read one other workgroup's slot (the exchange of a persistent pivoted-QR step: everybody reads the winner's column).
// Waits are bounded: a workgroup that spins too long sets an abort word and everybody leaves.
// hipcc -O3 --offload-arch=gfx950 tools/gridbar_xcd_probe.hip -o build/gridbar_xcd_probe && ./build/gridbar_xcd_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

struct alignas(128) Line { unsigned v; unsigned pad[31]; };
struct BarState {
    Line members[8];   // workgroups per XCD (setup)
    Line setup;        // flat arrival counter of the setup phase
    Line cnt[8];       // per-XCD arrival counters (monotonic)
    Line gen[8];       // per-XCD generation words
    Line top;          // top-level counter: one arrival per XCD per step
    Line flat;         // FORM 0
    Line abort_word;
};

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

template <int FORM>
__global__ void __launch_bounds__(512) bar_kernel(BarState* st, double* slots, int slot_doubles, int steps, int mode, double* sink,
                                                  unsigned* xcc_of_wg)
{
    const int G = gridDim.x, wg = blockIdx.x, tid = threadIdx.x;
    __shared__ double xs[2048];
    __shared__ unsigned s_nx, s_nxcd, s_part;
    unsigned xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    xcc &= 7u;
    // ---- setup (once): how many workgroups does my XCD hold, how many XCDs take part
    if (tid == 0) {
        xcc_of_wg[wg] = xcc;
        atomicAdd(&st->members[xcc].v, 1u);
        __threadfence();
        atomicAdd(&st->setup.v, 1u);
        spin_until(&st->setup.v, (unsigned)G, &st->abort_word.v);
        __threadfence();
        unsigned nx = 0;
        for (int x = 0; x < 8; ++x) nx += ld_agent(&st->members[x].v) != 0u;
        s_nx = ld_agent(&st->members[xcc].v);
        s_nxcd = nx;
        s_part = FORM == 2 ? (xcc == 0u) : 1u;
    }
    __syncthreads();
    if (!s_part) return;            // FORM 2: only the workgroups of XCD 0 play
    const unsigned n_x = s_nx, n_xcd = s_nxcd;
    double acc = 0.0;
    for (int k = 0; k < steps; ++k) {
        double* mine = slots + ((size_t)(k & 1) * G + wg) * slot_doubles;
        if (mode >= 1) for (int i = tid; i < slot_doubles; i += 512) mine[i] = (double)(k + wg) + acc * 1e-300;
        __syncthreads();            // every thread's stores have left the CU (vmcnt drained by the workgroup-scope release of the barrier)
        if (tid == 0) {
            bool ok = true;
            if (FORM == 0) {
                __threadfence();
                atomicAdd(&st->flat.v, 1u);
                ok = spin_until(&st->flat.v, (unsigned)G * (unsigned)(k + 1), &st->abort_word.v);
                __threadfence();
            } else {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                const unsigned old = __hip_atomic_fetch_add(&st->cnt[xcc].v, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (old == n_x * (unsigned)(k + 1) - 1u) {
                    // last arriver of this XCD: publish the XCD's L2, meet the other XCDs, open the XCD's gate
                    if (FORM == 1) {
                        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
                        __hip_atomic_fetch_add(&st->top.v, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        ok = spin_until(&st->top.v, n_xcd * (unsigned)(k + 1), &st->abort_word.v);
                    }
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                    __hip_atomic_store(&st->gen[xcc].v, (unsigned)(k + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                } else {
                    ok = spin_until(&st->gen[xcc].v, (unsigned)(k + 1), &st->abort_word.v);
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                }
            }
            (void)ok;
        }
        __syncthreads();
        if (ld_agent(&st->abort_word.v)) break;
        if (mode >= 2) {
            int other = (k * 37) % G;                                       // "the winner's column"
            if (FORM == 2) { for (int t = 0; t < G && xcc_of_wg[other] != 0u; ++t) other = (other + 1) % G; }
            const double* win = slots + ((size_t)(k & 1) * G + other) * slot_doubles;
            for (int i = tid; i < slot_doubles; i += 512) xs[i & 2047] = win[i];
            __syncthreads();
            if (win[tid % slot_doubles] != (double)(k + other)) atomicExch(&st->abort_word.v, 2u);   // stale read: the barrier leaked
            acc += xs[(tid * 7) & 2047] * 1e-300;
        }
    }
    if (acc == 12345.678) sink[0] = acc;
}

template <int FORM>
int run(int G, int steps, BarState* st, double* slots, double* sink, unsigned* xcc_of_wg)
{
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    for (int mode = 0; mode <= 2; ++mode)
        for (int sd : {16, 256, 2048}) {
            if (mode == 0 && sd != 16) continue;
            CHECK(hipMemset(st, 0, sizeof(BarState)));
            int stp = steps, md = mode, sdd = sd;
            void* args[] = {&st, &slots, &sdd, &stp, &md, &sink, &xcc_of_wg};
            CHECK(hipEventRecord(e0));
            CHECK(hipLaunchCooperativeKernel((const void*)bar_kernel<FORM>, dim3(G), dim3(512), args, 0, 0));
            CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
            float ms = 0; CHECK(hipEventElapsedTime(&ms, e0, e1));
            BarState h; CHECK(hipMemcpy(&h, st, sizeof(h), hipMemcpyDeviceToHost));
            printf("form=%d (0 flat counter, 1 barrier-xcd, 2 one XCD only) G=%d mode=%d slot=%6d B: %8.3f ms for %d steps = %6.2f us per step  members=[",
                   FORM, G, mode, sd * 8, ms, steps, ms * 1e3 / steps);
            for (int x = 0; x < 8; ++x) printf("%u%s", h.members[x].v, x == 7 ? "]" : " ");
            printf("%s\n", h.abort_word.v == 1 ? "  ABORTED (bounded wait)" : h.abort_word.v == 2 ? "  STALE READ" : "");
        }
    return 0;
}

int main(int argc, char** argv)
{
    const int G = argc > 1 ? atoi(argv[1]) : 256, steps = 2000;
    BarState* st; double *slots, *sink; unsigned* xcc_of_wg;
    const int maxd = 2048;
    CHECK(hipMalloc(&st, sizeof(BarState))); CHECK(hipMalloc(&sink, 8)); CHECK(hipMalloc(&xcc_of_wg, G * 4));
    CHECK(hipMalloc(&slots, (size_t)2 * G * maxd * 8));
    if (run<0>(G, steps, st, slots, sink, xcc_of_wg)) return 1;
    if (run<1>(G, steps, st, slots, sink, xcc_of_wg)) return 1;
    if (run<2>(G, steps, st, slots, sink, xcc_of_wg)) return 1;
    return 0;
}
