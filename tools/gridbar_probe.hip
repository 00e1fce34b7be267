// Grid barrier cost on MI355X (a persistent kernel with one workgroup per CU): N barriers back to back, optionally with a
// publish (every workgroup writes BYTES to its slot before the barrier) and a read of one slot after it -- the exchange pattern of a
// persistent pivoted QR step.  Bounded waits: a workgroup that spins too long sets an abort word and everybody leaves.
// hipcc -O3 --offload-arch=gfx950 tools/gridbar_probe.hip -o build/gridbar_probe && ./build/gridbar_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ void __launch_bounds__(512) bar_kernel(unsigned* counter, unsigned* abort_word, double* slots, int slot_doubles, int steps, int mode,
                                                  double* sink)
{
    const int G = gridDim.x, wg = blockIdx.x, tid = threadIdx.x;
    __shared__ double xs[2048];
    double acc = 0.0;
    for (int k = 0; k < steps; ++k) {
        double* mine = slots + ((size_t)(k & 1) * G + wg) * slot_doubles;
        if (mode >= 1) for (int i = tid; i < slot_doubles; i += 512) mine[i] = (double)(k + wg) + acc * 1e-300;
        __syncthreads();
        if (tid == 0) {
            __threadfence();                                             // release: the slot is visible device-wide
            atomicAdd(counter, 1u);
            const unsigned target = (unsigned)G * (unsigned)(k + 1);
            unsigned spins = 0;
            while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
                __builtin_amdgcn_s_sleep(1);
                if (++spins > (1u << 22) || __hip_atomic_load(abort_word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) { atomicExch(abort_word, 1u); break; }
            }
            __threadfence();                                             // acquire
        }
        __syncthreads();
        if (__hip_atomic_load(abort_word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) break;
        if (mode >= 2) {
            const double* win = slots + ((size_t)(k & 1) * G + (k * 37) % G) * slot_doubles;      // "the winner's column"
            for (int i = tid; i < slot_doubles; i += 512) xs[i & 2047] = __builtin_nontemporal_load(win + i);
            __syncthreads();
            acc += xs[(tid * 7) & 2047];
        }
    }
    if (acc == 12345.678) sink[0] = acc;
}

int main(int argc, char** argv)
{
    int dev = 0; hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, dev));
    const int G = argc > 1 ? atoi(argv[1]) : 250, steps = 2000;
    unsigned *counter, *abort_word; double *slots, *sink;
    const int maxd = 2048;
    CHECK(hipMalloc(&counter, 4)); CHECK(hipMalloc(&abort_word, 4)); CHECK(hipMalloc(&sink, 8));
    CHECK(hipMalloc(&slots, (size_t)2 * G * maxd * 8));
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    for (int mode = 0; mode <= 2; ++mode)
        for (int sd : {256, 1024, 2048}) {
            if (mode == 0 && sd != 256) continue;
            CHECK(hipMemset(counter, 0, 4)); CHECK(hipMemset(abort_word, 0, 4));
            int st = steps, md = mode, sdd = sd;
            void* args[] = {&counter, &abort_word, &slots, &sdd, &st, &md, &sink};
            CHECK(hipEventRecord(e0));
            CHECK(hipLaunchCooperativeKernel((const void*)bar_kernel, dim3(G), dim3(512), args, 0, 0));
            CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
            float ms = 0; CHECK(hipEventElapsedTime(&ms, e0, e1));
            unsigned ab = 0; CHECK(hipMemcpy(&ab, abort_word, 4, hipMemcpyDeviceToHost));
            printf("G=%d mode=%d (0 barrier only, 1 + publish, 2 + publish and read) slot=%5d B: %8.3f ms for %d steps = %6.2f us per step%s\n", G, mode,
                   sd * 8, ms, steps, ms * 1e3 / steps, ab ? "  ABORTED (bounded wait)" : "");
        }
    return 0;
}
