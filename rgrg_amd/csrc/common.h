// Shared helpers for librgrg_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/rgrg_hip.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

namespace rgrg {

void set_error(const char* fmt, ...);

inline hipStream_t as_stream(void* s) { return reinterpret_cast<hipStream_t>(s); }

#define RGRG_CHECK_ARG(cond)                                                   \
    do {                                                                       \
        if (!(cond)) {                                                         \
            rgrg::set_error("%s:%d: invalid argument: %s", __FILE__, __LINE__, #cond); \
            return RGRG_EINVAL;                                                \
        }                                                                      \
    } while (0)

#define RGRG_HIP(call)                                                         \
    do {                                                                       \
        hipError_t e_ = (call);                                                \
        if (e_ != hipSuccess) {                                                \
            rgrg::set_error("%s:%d: %s -> %s", __FILE__, __LINE__, #call, hipGetErrorString(e_)); \
            return RGRG_EHIP;                                                  \
        }                                                                      \
    } while (0)

#define RGRG_LAUNCH_CHECK() RGRG_HIP(hipGetLastError())

__device__ __forceinline__ float gelu_new(float x) {
    const float k = 0.7978845608028654f;  // sqrt(2/pi)
    return 0.5f * x * (1.0f + tanhf(k * (x + 0.044715f * x * x * x)));
}

__device__ __forceinline__ float apply_act(float v, int act) {
    if (act == RGRG_ACT_RELU) return v > 0.f ? v : 0.f;
    if (act == RGRG_ACT_GELU_NEW) return gelu_new(v);
    return v;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

}  // namespace rgrg
