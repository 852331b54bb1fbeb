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
extern "C" int rgrg_abi_version(void) { return 7; }  // keep in step with rgrg_amd/_hip.py ABI_VERSION
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
