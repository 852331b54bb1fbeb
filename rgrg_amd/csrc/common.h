// Shared helpers for librgrg_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/rgrg_hip.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

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
// The two 16-bit storage / matrix-core input types of the opt-in reduced-precision path (torch.autocast's dtype): bf16 and
// IEEE fp16 - the reference's scripts run under torch.autocast(float16) (generate_reports_for_images.py:108,
// train_full_model.py:172).  Both round to nearest even; fp16 saturates to inf beyond 65504 as torch's does.
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ unsigned f32_to_bf16_bits(float f) {
    // the hardware conversion (v_cvt_pk_bf16_f32, round to nearest even): one instruction instead of the four of the integer
    // form (add 0x7fff + lsb, shift) - the 16-bit GEMM epilogues convert 128-256 values per lane
    return (unsigned)__builtin_bit_cast(unsigned short, (__bf16)f);
}
__device__ __forceinline__ unsigned f32_to_f16_bits(float f) { return (unsigned)__builtin_bit_cast(unsigned short, (_Float16)f); }
__device__ __forceinline__ float bf16_bits_to_f32(unsigned h) { return __uint_as_float(h << 16); }
__device__ __forceinline__ float f16_bits_to_f32(unsigned h) { return (float)__builtin_bit_cast(_Float16, (unsigned short)h); }
template <bool F16>
__device__ __forceinline__ unsigned to16(float f) {
    if constexpr (F16) return f32_to_f16_bits(f);
    else return f32_to_bf16_bits(f);
}
template <bool F16>
__device__ __forceinline__ float from16(unsigned h) {
    if constexpr (F16) return f16_bits_to_f32(h);
    else return bf16_bits_to_f32(h);
}
__device__ __forceinline__ unsigned to16_rt(float f, int f16) { return f16 ? f32_to_f16_bits(f) : f32_to_bf16_bits(f); }
__device__ __forceinline__ float from16_rt(unsigned h, int f16) { return f16 ? f16_bits_to_f32(h) : bf16_bits_to_f32(h); }

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


// Counter-based dropout (training pass): Philox4x32-7 keyed by the call's seed, counter = (element index / 4, stream id),
// element i takes word i % 4 of its call - a kernel that owns 4 consecutive elements pays ONE generator call for them (round 5:
// one call per element made the row kernels and the attention kernels generator-bound: ~440 cycles of quarter-rate integer
// multiplies per wave call).  Every element's mask is a pure function of (seed, stream, index), so the backward pass
// recomputes exactly the mask of the forward pass without storing it, in any kernel and in any order.  0 (dropped) or 1/(1-p).
struct DropoutParams {
    unsigned long long seed;
    unsigned stream;  // layer * 4 + site (0 embedding, 1 attention probabilities, 2 attn c_proj output, 3 mlp c_proj output)
    float p;          // 0: no dropout (every helper short-circuits)
};
struct Philox4 { unsigned w[4]; };
constexpr int DROPOUT_PHILOX_ROUNDS = 7;   // Philox4x32-7: the fewest rounds Random123 reports as passing BigCrush (the sampler
                                           // keys of det_train.hip, which are pinned against published vectors, keep 10)
__host__ __device__ __forceinline__ Philox4 philox4x32(unsigned long long seed, unsigned stream, unsigned long long ctr) {
    unsigned c0 = (unsigned)ctr, c1 = (unsigned)(ctr >> 32), c2 = stream, c3 = 0u;
    unsigned k0 = (unsigned)seed, k1 = (unsigned)(seed >> 32);
#pragma unroll
    for (int r = 0; r < DROPOUT_PHILOX_ROUNDS; ++r) {
        const unsigned long long p0 = 0xD2511F53ull * c0, p1 = 0xCD9E8D57ull * c2;
        const unsigned n0 = (unsigned)(p1 >> 32) ^ c1 ^ k0, n1 = (unsigned)p1;
        const unsigned n2 = (unsigned)(p0 >> 32) ^ c3 ^ k1, n3 = (unsigned)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    return Philox4{{c0, c1, c2, c3}};
}
__host__ __device__ __forceinline__ float dropout_word_mask(const DropoutParams& d, unsigned w) {
    const float u = (float)(w >> 8) * (1.0f / 16777216.0f);  // [0,1), 24 bits
    return u < d.p ? 0.f : 1.0f / (1.0f - d.p);
}
// masks of the elements 4 * idx4 .. 4 * idx4 + 3
__host__ __device__ __forceinline__ void dropout_mask4(const DropoutParams& d, unsigned long long idx4, float (&m)[4]) {
    if (d.p <= 0.f) { m[0] = m[1] = m[2] = m[3] = 1.f; return; }
    const Philox4 w = philox4x32(d.seed, d.stream, idx4);
#pragma unroll
    for (int e = 0; e < 4; ++e) m[e] = dropout_word_mask(d, w.w[e]);
}
__host__ __device__ __forceinline__ float dropout_mask(const DropoutParams& d, unsigned long long idx) {
    if (d.p <= 0.f) return 1.f;
    const Philox4 w = philox4x32(d.seed, d.stream, idx >> 2);
    const unsigned e = (unsigned)idx & 3u;
    return dropout_word_mask(d, e == 0 ? w.w[0] : e == 1 ? w.w[1] : e == 2 ? w.w[2] : w.w[3]);
}
// Attention-probability masks (site 1) are indexed [sentence][head][query][key] with the key pitch rounded up to a multiple of 4,
// so that 4 consecutive keys of one query share a generator call whatever T is.
__host__ __device__ __forceinline__ int dropout_key_pitch(int n_keys) { return (n_keys + 3) & ~3; }

#ifdef __HIPCC__
// 4 x 4 transpose inside the lanes of a quad: lane i ends up with (lane 0's a_i, lane 1's a_i, lane 2's a_i, lane 3's a_i)
template <int CTRL>
__device__ __forceinline__ float quad_dpp(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, false));
}
__device__ __forceinline__ void quad_transpose4(float& a0, float& a1, float& a2, float& a3, bool odd, bool hi) {
    float x = odd ? a0 : a1, y = quad_dpp<0xB1>(x);   // quad_perm [1,0,3,2]
    if (odd) a0 = y; else a1 = y;
    x = odd ? a2 : a3; y = quad_dpp<0xB1>(x);
    if (odd) a2 = y; else a3 = y;
    x = hi ? a0 : a2; y = quad_dpp<0x4E>(x);          // quad_perm [2,3,0,1]
    if (hi) a0 = y; else a2 = y;
    x = hi ? a1 : a3; y = quad_dpp<0x4E>(x);
    if (hi) a1 = y; else a3 = y;
}
#endif

}  // namespace rgrg
