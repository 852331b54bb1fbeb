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
extern "C" int rgrg_abi_version(void) { return 20; }  // keep in step with rgrg_amd/_hip.py ABI_VERSION
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
// passes the barrier and reads other workgroups' slices - the hand-off a fused GEMM chain needs.
// Spins are bounded; stale[0] counts hand-offs that read old data, stale[1] flags a timeout.
// Round-3 variants (plain payload stores, one reader slice of workgroup (b + 97) % grid):
//   0 one atomic counter, agent-scope fences around it (cooperative-groups style)
//   1 one atomic counter, no fences
//   2 per-workgroup arrival flags (plain coherent stores, wave 0 polls all of them), fences
//   3 flags, no fences (payload through agent-scope stores / loads instead)
//   4 two-level counters: 8 groups of 32 workgroups, then one counter of 8; fences
//   5 the two fences alone (no synchronisation; stale reads expected)
// Round-4 variants (MI355X_MICROARCH.md price list rows barrier-xcd / publish-large): the payload is the real epilogue
// pattern - every thread publishes its floats with WRITE-THROUGH (sc1) stores, every storing wave drains vmcnt(0), no
// release fence - and every workgroup then reads the slices of 64 producers (128 KiB at 512 floats per workgroup: the A
// operand of the next GEMM) with 16-byte loads:
//   6 XCD-hierarchical barrier (per-group counter -> last arriver of the group bumps the top counter -> last group
//     writes the 8 per-group generation words; groups = b % 8 = the XCD under round-robin placement, correctness does
//     not depend on it), relaxed sc1 polls + s_sleep, ONE agent-scope acquire per workgroup after the poll, plain loads
//   7 the same barrier, no acquire: the consumer's loads are sc1 (L1-bypassing) instead
//   8 one flat relaxed counter, sc1 payload stores, sc1 loads, no fence at all
// `variant | 0x100`: uneven load (a pseudo-random sixth of the workgroups is delayed ~2 us before publishing) - the
// regime in which missing acquires / releases show; use it for the stale-read check, not for the timing.
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

// Arrival + wait of ONE thread per workgroup (tid 0, after __syncthreads() and after every storing wave drained its
// write-through stores).  st: [8 group counters | top counter | 8 group generation words], 64 bytes apart, zeroed
// before the launch.  `it` = phase index + 1 (never 0).  Placement independent: the group is blockIdx % 8.
__device__ __forceinline__ bool gb_xcd_barrier(unsigned* st, unsigned it, int b, int n) {
    const int g = b & 7;
    const unsigned per = (unsigned)((n - g + 7) >> 3);   // workgroups of this group
    unsigned* grp = st + 16 * g;
    unsigned* top = st + 16 * 8;
    const unsigned old = __hip_atomic_fetch_add(grp, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (old == it * per - 1) {                              // last arriver of the group in this phase
        const unsigned ngroups = (unsigned)(n < 8 ? n : 8);
        const unsigned o2 = __hip_atomic_fetch_add(top, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (o2 == it * ngroups - 1)
            for (unsigned k = 0; k < ngroups; ++k) __hip_atomic_store(st + 16 * (9 + k), it, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    unsigned* gen = st + 16 * (9 + g);
    int spins = 0;
    while (__hip_atomic_load(gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != it) {
        __builtin_amdgcn_s_sleep(1);
        if (++spins > (1 << 16)) return false;
    }
    return true;
}

__global__ __launch_bounds__(512) void grid_barrier_bench_kernel(int variant_in, int iters, int payload, unsigned* ctr, unsigned* flags,
                                                                 float* data, unsigned* stale) {
    const int b = blockIdx.x, n = gridDim.x, tid = threadIdx.x;
    const int variant = variant_in & 0xff;
    const bool uneven = (variant_in & 0x100) != 0;
    const bool fences = variant == 0 || variant == 2 || variant == 4 || variant == 5;
    const bool coherent_payload = variant == 3;
    unsigned bad = 0;
    if (variant >= 6) {
        const __amdgpu_buffer_rsrc_t rd = __builtin_amdgcn_make_buffer_rsrc(data, 0, 0x7fffffff, 0x00020000);
        for (int it = 1; it <= iters; ++it) {
            if (uneven && ((unsigned)(b * 2654435761u + it * 40503u) >> 7) % 6u == 0u)
                for (int k = 0; k < 40; ++k) __builtin_amdgcn_s_sleep(127);   // ~2 us
            // two payload buffers, alternating per round: the writer of round it + 1 cannot overtake a reader of round it
            // (a slice is rewritten only two barriers later), so every mismatch below is a genuine visibility failure
            const size_t half_off = (size_t)(it & 1) * (size_t)n * payload;
            float* mine = data + half_off + (size_t)b * payload;
            const float val = (float)(it * 1000 + b);
            for (int i = tid; i < payload; i += 512) __hip_atomic_store(mine + i, val, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // sc1 write-through
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // every storing wave drains before the arrival
            __syncthreads();
            if (tid == 0) {
                bool ok;
                if (variant == 8) {
                    __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    ok = gb_spin_counter(ctr, (unsigned)(it * n));
                } else {
                    ok = gb_xcd_barrier(ctr, (unsigned)it, b, n);
                }
                if (!ok) stale[1] = 1;
                if (variant == 6) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            }
            __syncthreads();
            // consumer: the slices of 64 consecutive producers starting at (b % 4) * 64 (+ wrap), 16-byte loads
            const int p0 = (b & 3) * (n / 4);
            const int total4 = (n / 4) * payload / 4;   // float4 elements to read
            auto ld = [&](unsigned off) {
                return variant == 6 ? __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rd, (int)off, 0, 0))
                                    : __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rd, (int)off, 0, 16));  // aux 16 = sc1
            };
            if (total4 == 16 * 512) {   // the decode shape (512 floats per producer): all 16 loads of a thread in flight at once
                f32x4 v[16];
                unsigned prodv[16];
#pragma unroll
                for (int u = 0; u < 16; ++u) {
                    const int e = (tid + 512 * u) * 4;
                    prodv[u] = (unsigned)((p0 + e / payload) % n);
                    v[u] = ld((unsigned)((half_off + (size_t)prodv[u] * payload + e % payload) * 4));
                }
#pragma unroll
                for (int u = 0; u < 16; ++u) {
                    const float want = (float)(it * 1000 + (int)prodv[u]);
                    bad += (v[u][0] != want) + (v[u][1] != want) + (v[u][2] != want) + (v[u][3] != want);
                }
            } else {
                for (int i = tid; i < total4; i += 512) {
                    const int e = i * 4;
                    const int prod = (p0 + e / payload) % n;
                    const f32x4 v = ld((unsigned)((half_off + (size_t)prod * payload + e % payload) * 4));
                    const float want = (float)(it * 1000 + prod);
                    bad += (v[0] != want) + (v[1] != want) + (v[2] != want) + (v[3] != want);
                }
            }
            __syncthreads();
        }
        if (bad) atomicAdd(stale, bad);
        return;
    }
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

// Which XCD does workgroup b of a launch run on?  out[b] = XCC id (placement probe for the L2 prefetch share-out)
__global__ void xcc_probe_kernel(int* out) {
    if (threadIdx.x == 0) out[blockIdx.x] = xcc_id();
}
}  // namespace rgrg

extern "C" int rgrg_debug_grid_barrier(int variant, int iters, int payload_floats, float* us_per_barrier, unsigned* stale_out) {
    RGRG_CHECK_ARG((variant & 0xff) >= 0 && (variant & 0xff) <= 8 && iters > 0 && payload_floats >= 0 && payload_floats % 4 == 0 && us_per_barrier && stale_out);
    const int grid = 256;
    unsigned *ctr = nullptr, *flags = nullptr, *stale = nullptr;
    float* data = nullptr;
    RGRG_HIP(hipMalloc((void**)&ctr, 64 * 16 * sizeof(unsigned)));
    RGRG_HIP(hipMalloc((void**)&flags, (size_t)grid * 16 * sizeof(unsigned)));
    RGRG_HIP(hipMalloc((void**)&stale, 2 * sizeof(unsigned)));
    RGRG_HIP(hipMalloc((void**)&data, (size_t)2 * grid * (payload_floats + 4) * sizeof(float)));
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

// XCC id of every workgroup of `launches` back-to-back launches of `blocks` workgroups: out[launch * blocks + b]
extern "C" int rgrg_debug_xcc_map(int blocks, int launches, int* out_host) {
    RGRG_CHECK_ARG(blocks > 0 && launches > 0 && out_host);
    int* d = nullptr;
    RGRG_HIP(hipMalloc((void**)&d, (size_t)blocks * launches * sizeof(int)));
    for (int l = 0; l < launches; ++l) hipLaunchKernelGGL(rgrg::xcc_probe_kernel, dim3(blocks), dim3(64), 0, nullptr, d + (size_t)l * blocks);
    RGRG_HIP(hipDeviceSynchronize());
    RGRG_HIP(hipMemcpy(out_host, d, (size_t)blocks * launches * sizeof(int), hipMemcpyDeviceToHost));
    (void)hipFree(d);
    return RGRG_OK;
}
