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
extern "C" int rgrg_abi_version(void) { return 1; }
extern "C" int rgrg_device_arch(int dev, char* buf, int buflen) {
    RGRG_CHECK_ARG(buf && buflen > 1);
    hipDeviceProp_t prop;
    RGRG_HIP(hipGetDeviceProperties(&prop, dev));
    strncpy(buf, prop.gcnArchName, buflen - 1);
    buf[buflen - 1] = 0;
    return RGRG_OK;
}
