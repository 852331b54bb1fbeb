// Error string, ABI version and device query of librgrg_hip.so.
#include <stdarg.h>
#include <string.h>

#include "common.h"

namespace rgrg {
static thread_local char g_err[512] = "";
void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
}  // namespace rgrg

extern "C" const char* rgrg_last_error(void) { return rgrg::g_err; }
extern "C" int rgrg_abi_version(void) { return 9; }  // keep in step with rgrg_amd/_hip.py ABI_VERSION
extern "C" int rgrg_device_arch(int dev, char* buf, int buflen) {
    RGRG_CHECK_ARG(buf && buflen > 1);
    hipDeviceProp_t prop;
    RGRG_HIP(hipGetDeviceProperties(&prop, dev));
    strncpy(buf, prop.gcnArchName, buflen - 1);
    buf[buflen - 1] = 0;
    return RGRG_OK;
}

// ---------------------------------------------------------------------------------
// Debug / measurement helper (not on the product path): cost of a dependent chain of
// n trivial kernels launched (mode 0) eagerly on a private non-blocking stream,
// (mode 1) as one captured hipGraph replay, (mode 2) eagerly on the null stream.
// Returns host wall microseconds per kernel including the final synchronise.
// ---------------------------------------------------------------------------------
#include <chrono>
namespace rgrg {
__global__ void chain_kernel(int* p, int n) {
    if (threadIdx.x == 0 && blockIdx.x == 0) p[0] += n;
}
}  // namespace rgrg

extern "C" int rgrg_debug_chain(int n, int mode, int blocks, float* us_per_kernel) {
    RGRG_CHECK_ARG(n > 0 && us_per_kernel && blocks > 0);
    int* d = nullptr;
    RGRG_HIP(hipMalloc((void**)&d, 64));
    RGRG_HIP(hipMemset(d, 0, 64));
    hipStream_t st = nullptr;
    if (mode != 2) RGRG_HIP(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    hipGraphExec_t exec = nullptr;
    if (mode == 1) {
        hipGraph_t g = nullptr;
        RGRG_HIP(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
        for (int i = 0; i < n; ++i) hipLaunchKernelGGL(rgrg::chain_kernel, dim3(blocks), dim3(64), 0, st, d, 1);
        RGRG_HIP(hipStreamEndCapture(st, &g));
        RGRG_HIP(hipGraphInstantiate(&exec, g, nullptr, nullptr, 0));
        (void)hipGraphDestroy(g);
        RGRG_HIP(hipGraphLaunch(exec, st));  // warm-up replay
        RGRG_HIP(hipStreamSynchronize(st));
    } else {
        for (int i = 0; i < 16; ++i) hipLaunchKernelGGL(rgrg::chain_kernel, dim3(blocks), dim3(64), 0, st, d, 1);
        RGRG_HIP(hipStreamSynchronize(st));
    }
    auto t0 = std::chrono::steady_clock::now();
    if (mode == 1) {
        RGRG_HIP(hipGraphLaunch(exec, st));
    } else {
        for (int i = 0; i < n; ++i) hipLaunchKernelGGL(rgrg::chain_kernel, dim3(blocks), dim3(64), 0, st, d, 1);
    }
    RGRG_HIP(hipStreamSynchronize(st));
    auto t1 = std::chrono::steady_clock::now();
    *us_per_kernel = (float)(std::chrono::duration<double, std::micro>(t1 - t0).count() / n);
    if (exec) (void)hipGraphExecDestroy(exec);
    if (st) (void)hipStreamDestroy(st);
    (void)hipFree(d);
    return RGRG_OK;
}

// ---------------------------------------------------------------------------------
// Debug / measurement helper (not on the product path): latency of a grid-wide barrier
// among 256 resident workgroups of 512 threads (the shape of the fused decode kernels),
// per variant.  Every iteration each workgroup writes `payload` floats of its own slice,
// passes the barrier and reads the slice of workgroup (b + 97) % grid - the data hand-off
// a fused GEMM pair needs.  Spins are bounded; stale[0] counts hand-offs that read old data.
// Measured (profiles/r03_grid_barrier_bench.log): the two agent-scope fences a correct hand-off
// needs cost ~10 us per round with 0.5 MB of payload - more than a kernel boundary (~4 us),
// which is why the decode plan keeps one launch per GEMM.
//   0 one atomic counter, agent-scope fences around it (cooperative-groups style)
//   1 one atomic counter, no fences
//   2 per-workgroup arrival flags (plain coherent stores, wave 0 polls all of them), fences
//   3 flags, no fences (payload through agent-scope stores / loads instead)
//   4 two-level counters: 8 groups of 32 workgroups, then one counter of 8; fences
//   5 the two fences alone (no synchronisation; stale reads expected)
// ---------------------------------------------------------------------------------
namespace rgrg {
__device__ __forceinline__ bool gb_spin_counter(unsigned* c, unsigned target) {
    int spins = 0;
    while ((int)(__hip_atomic_load(c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - target) < 0) {
        __builtin_amdgcn_s_sleep(1);
        if (++spins > (1 << 16)) return false;
    }
    return true;
}

__global__ __launch_bounds__(512) void grid_barrier_bench_kernel(int variant, int iters, int payload, unsigned* ctr, unsigned* flags,
                                                                 float* data, unsigned* stale) {
    const int b = blockIdx.x, n = gridDim.x, tid = threadIdx.x;
    const bool fences = variant == 0 || variant == 2 || variant == 4 || variant == 5;
    const bool coherent_payload = variant == 3;
    unsigned bad = 0;
    for (int it = 1; it <= iters; ++it) {
        float* mine = data + (size_t)b * payload;
        for (int i = tid; i < payload; i += 512) {
            const float v = (float)(it * 1000 + b);
            if (coherent_payload) __hip_atomic_store(mine + i, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            else mine[i] = v;
        }
        __syncthreads();
        if (variant == 0 || variant == 1) {
            if (tid == 0) {
                if (fences) __threadfence();
                atomicAdd(ctr, 1u);
                if (!gb_spin_counter(ctr, (unsigned)(it * n))) stale[1] = 1;
                if (fences) __threadfence();
            }
        } else if (variant == 2 || variant == 3) {
            if (tid < 64) {
                if (tid == 0) {
                    if (fences) __threadfence();
                    else __builtin_amdgcn_s_waitcnt(0);  // the coherent payload stores of this wave are issued in order before the flag
                    __hip_atomic_store(flags + b * 16, (unsigned)it, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
                }
                int spins = 0;
                for (;;) {
                    bool ok = true;
                    for (int j = tid; j < n; j += 64)
                        ok = ok && (int)(__hip_atomic_load(flags + j * 16, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - (unsigned)it) >= 0;
                    if (__all(ok)) break;
                    if (++spins > (1 << 16)) { if (tid == 0) stale[1] = 1; break; }
                }
                if (fences && tid == 0) __threadfence();
            }
        } else if (variant == 4) {
            if (tid == 0) {
                __threadfence();
                unsigned* grp = ctr + 64 * (1 + (b & 7));
                const unsigned old = atomicAdd(grp, 1u);
                const unsigned per = (unsigned)(n / 8);
                if (old % per == per - 1) atomicAdd(ctr, 1u);  // last of its group
                if (!gb_spin_counter(ctr, (unsigned)(it * 8))) stale[1] = 1;
                __threadfence();
            }
        } else {
            if (tid == 0) { __threadfence(); __threadfence(); }
        }
        __syncthreads();
        const float* other = data + (size_t)((b + 97) % n) * payload;
        const float want = (float)(it * 1000 + (b + 97) % n);
        for (int i = tid; i < payload; i += 512) {
            const float v = coherent_payload ? __hip_atomic_load(other + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : other[i];
            if (v != want) ++bad;
        }
        __syncthreads();
        // (a slice is rewritten in round it + 1 while a slow reader of round it may still be at it: such reads count as
        //  stale too - the count is an upper bound of the visibility failures, the timing is unaffected)
    }
    if (bad) atomicAdd(stale, bad);
}
}  // namespace rgrg

extern "C" int rgrg_debug_grid_barrier(int variant, int iters, int payload_floats, float* us_per_barrier, unsigned* stale_out) {
    RGRG_CHECK_ARG(variant >= 0 && variant <= 5 && iters > 0 && payload_floats >= 0 && us_per_barrier && stale_out);
    const int grid = 256;
    unsigned *ctr = nullptr, *flags = nullptr, *stale = nullptr;
    float* data = nullptr;
    RGRG_HIP(hipMalloc((void**)&ctr, 64 * 16 * sizeof(unsigned)));
    RGRG_HIP(hipMalloc((void**)&flags, (size_t)grid * 16 * sizeof(unsigned)));
    RGRG_HIP(hipMalloc((void**)&stale, 2 * sizeof(unsigned)));
    RGRG_HIP(hipMalloc((void**)&data, (size_t)grid * (payload_floats + 1) * sizeof(float)));
    hipEvent_t e0, e1;
    RGRG_HIP(hipEventCreate(&e0));
    RGRG_HIP(hipEventCreate(&e1));
    float ms = 0.f;
    for (int rep = 0; rep < 2; ++rep) {  // the second launch is the measured one
        RGRG_HIP(hipMemset(ctr, 0, 64 * 16 * sizeof(unsigned)));
        RGRG_HIP(hipMemset(flags, 0, (size_t)grid * 16 * sizeof(unsigned)));
        RGRG_HIP(hipMemset(stale, 0, 2 * sizeof(unsigned)));
        RGRG_HIP(hipDeviceSynchronize());
        RGRG_HIP(hipEventRecord(e0, nullptr));
        hipLaunchKernelGGL(rgrg::grid_barrier_bench_kernel, dim3(grid), dim3(512), 0, nullptr, variant, iters, payload_floats, ctr, flags,
                           data, stale);
        RGRG_HIP(hipEventRecord(e1, nullptr));
        RGRG_HIP(hipEventSynchronize(e1));
        RGRG_HIP(hipEventElapsedTime(&ms, e0, e1));
    }
    RGRG_HIP(hipMemcpy(stale_out, stale, 2 * sizeof(unsigned), hipMemcpyDeviceToHost));
    *us_per_barrier = ms * 1e3f / (float)iters;
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    (void)hipFree(ctr); (void)hipFree(flags); (void)hipFree(stale); (void)hipFree(data);
    return RGRG_OK;
}
