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
// Wave-wide sum in 7 VALU instructions (DPP butterflies inside the 16-lane rows, row_bcast across rows, readlane),
// ~10x less latency than six ds_bpermute round trips; the result is wave-uniform.  Summation order differs from
// wave_sum: use one or the other consistently where bit-reproducibility against another kernel matters.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_get(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, ROW_MASK, 0xf, false));
}
__device__ __forceinline__ float wave_sum_dpp(float v) {
    v += dpp_get<0xB1, 0xf>(v);   // quad_perm [1,0,3,2]
    v += dpp_get<0x4E, 0xf>(v);   // quad_perm [2,3,0,1]
    v += dpp_get<0x141, 0xf>(v);  // row_half_mirror
    v += dpp_get<0x140, 0xf>(v);  // row_mirror: every lane of a 16-lane row holds the row sum
    v += dpp_get<0x142, 0xa>(v);  // row_bcast15 -> rows 1, 3
    v += dpp_get<0x143, 0xc>(v);  // row_bcast31 -> rows 2, 3: row 3 holds the wave sum
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}
// XCD (accelerator complex die, 0..7) this wave runs on: every XCD has its own 4 MiB L2
__device__ __forceinline__ int xcc_id() {
    int x;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID, 0, 4)" : "=s"(x));
    return x;
}

__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}


// Counter-based dropout (training pass): Philox4x32-10 keyed by the call's seed, counter = (element index, stream id).
// Every element's mask is a pure function of (seed, stream, index), so the backward pass recomputes exactly the
// mask of the forward pass without storing it, in any kernel and in any order.  Returns 0 (dropped) or 1/(1-p).
struct DropoutParams {
    unsigned long long seed;
    unsigned stream;  // layer * 4 + site (0 embedding, 1 attention probabilities, 2 attn c_proj output, 3 mlp c_proj output)
    float p;          // 0: no dropout (every helper short-circuits)
};
__host__ __device__ __forceinline__ unsigned philox_first_word(unsigned long long seed, unsigned stream, unsigned long long idx) {
    unsigned c0 = (unsigned)idx, c1 = (unsigned)(idx >> 32), c2 = stream, c3 = 0u;
    unsigned k0 = (unsigned)seed, k1 = (unsigned)(seed >> 32);
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const unsigned long long p0 = 0xD2511F53ull * c0, p1 = 0xCD9E8D57ull * c2;
        const unsigned n0 = (unsigned)(p1 >> 32) ^ c1 ^ k0, n1 = (unsigned)p1;
        const unsigned n2 = (unsigned)(p0 >> 32) ^ c3 ^ k1, n3 = (unsigned)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    return c0;
}
__host__ __device__ __forceinline__ float dropout_mask(const DropoutParams& d, unsigned long long idx) {
    if (d.p <= 0.f) return 1.f;
    const float u = (float)(philox_first_word(d.seed, d.stream, idx) >> 8) * (1.0f / 16777216.0f);  // [0,1), 24 bits
    return u < d.p ? 0.f : 1.0f / (1.0f - d.p);
}

}  // namespace rgrg
