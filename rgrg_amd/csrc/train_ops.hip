// Backward kernels of the teacher-forced language-model pass (the gradients of
// LanguageModel.forward(return_loss=True), src/language_model/language_model.py:258-399, w.r.t. what the reference
// trains in the decoder: uk / uv of every GPT2PseudoAttention (:50-57,:145-150) and feature_space_transformation_nn
// (:230-236); every other GPT-2 tensor is frozen (:207-213, Conv1DWithTrainedWeights :11-29), so only activation
// gradients flow through it).  The GEMMs of the backward pass reuse gemm_f32.hip on transposed weight copies; this
// file holds the element-wise / row-wise / attention parts.  Launchers are called from decoder.hip.
#include "common.h"

namespace rgrg {

constexpr float LN_EPS_T = 1e-5f;

__device__ __forceinline__ float block_sum256(float v, float* sh) {
    v = wave_sum(v);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) sh[wave] = v;
    __syncthreads();
    return (sh[0] + sh[1]) + (sh[2] + sh[3]);
}

// ff = gelu_new(pre)   (NewGELUActivation; the training pass keeps the pre-activation for the backward)
__global__ __launch_bounds__(256) void gelu_apply_kernel(const float* __restrict__ pre, float* __restrict__ out, size_t n4) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
        const f32x4 v = reinterpret_cast<const f32x4*>(pre)[i];
        f32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = gelu_new(v[e]);
        reinterpret_cast<f32x4*>(out)[i] = o;
    }
}

// d *= gelu_new'(pre):  0.5 (1 + tanh u) + 0.5 x (1 - tanh^2 u) sqrt(2/pi) (1 + 3 * 0.044715 x^2)
__global__ __launch_bounds__(256) void gelu_backward_kernel(float* __restrict__ d, const float* __restrict__ pre, size_t n4) {
    const float k = 0.7978845608028654f;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
        const f32x4 x = reinterpret_cast<const f32x4*>(pre)[i];
        f32x4 g = reinterpret_cast<f32x4*>(d)[i];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float th = tanhf(k * (x[e] + 0.044715f * x[e] * x[e] * x[e]));
            g[e] *= 0.5f * (1.0f + th) + 0.5f * x[e] * (1.0f - th * th) * k * (1.0f + 3.0f * 0.044715f * x[e] * x[e]);
        }
        reinterpret_cast<f32x4*>(d)[i] = g;
    }
}

// d *= (h > 0)   (ReLU of feature_space_transformation_nn; h is the post-ReLU activation)
__global__ __launch_bounds__(256) void relu_backward_kernel(float* __restrict__ d, const float* __restrict__ h, size_t n) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256)
        if (!(h[i] > 0.f)) d[i] = 0.f;
}

// out = (resid ? resid : 0) + src * mask(index)  (element index = flat position; src may alias out)
__global__ __launch_bounds__(256) void dropout_add_kernel(const float* __restrict__ src, const float* __restrict__ resid,
                                                          float* __restrict__ out, size_t n, const DropoutParams drop) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const float v = src[i] * dropout_mask(drop, i);
        out[i] = resid ? resid[i] + v : v;
    }
}

// row_len > 0: the elements are rows of row_len whose index pitch is dropout_key_pitch(row_len) (attention-probability masks)
__global__ __launch_bounds__(256) void dropout_mask_kernel(float* __restrict__ out, size_t n, const DropoutParams drop, int row_len) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256)
        out[i] = dropout_mask(drop, row_len > 0 ? (i / row_len) * dropout_key_pitch(row_len) + i % row_len : i);
}

// LayerNorm backward w.r.t. its input (weight / bias are frozen): one workgroup per row, D == 1024.
//   xhat = (x - mean) rstd ; t = dy * g ; dx = rstd (t - mean(t) - xhat mean(t xhat)) ; out = (acc ? out : 0) + dx
__global__ __launch_bounds__(256) void ln_backward_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                          const float* __restrict__ g, float* __restrict__ out, int D,
                                                          int accumulate) {
    __shared__ float sh[4];
    const int row = blockIdx.x, tid = threadIdx.x;
    const f32x4 v = reinterpret_cast<const f32x4*>(x + (size_t)row * D)[tid];
    const float mean = block_sum256((v[0] + v[1]) + (v[2] + v[3]), sh) / (float)D;
    f32x4 c;
#pragma unroll
    for (int e = 0; e < 4; ++e) c[e] = v[e] - mean;
    const float var = block_sum256((c[0] * c[0] + c[1] * c[1]) + (c[2] * c[2] + c[3] * c[3]), sh) / (float)D;
    const float rstd = 1.0f / sqrtf(var + LN_EPS_T);
    const f32x4 gg = reinterpret_cast<const f32x4*>(g)[tid];
    const f32x4 dd = reinterpret_cast<const f32x4*>(dy + (size_t)row * D)[tid];
    f32x4 t, xh;
#pragma unroll
    for (int e = 0; e < 4; ++e) { t[e] = dd[e] * gg[e]; xh[e] = c[e] * rstd; }
    const float m1 = block_sum256((t[0] + t[1]) + (t[2] + t[3]), sh) / (float)D;
    const float m2 = block_sum256((t[0] * xh[0] + t[1] * xh[1]) + (t[2] * xh[2] + t[3] * xh[3]), sh) / (float)D;
    f32x4 o = {0.f, 0.f, 0.f, 0.f};
    if (accumulate) o = reinterpret_cast<const f32x4*>(out + (size_t)row * D)[tid];
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] += rstd * (t[e] - m1 - xh[e] * m2);
    reinterpret_cast<f32x4*>(out + (size_t)row * D)[tid] = o;
}

// ---------------------------------------------------------------- 16-bit activation flow of the training pass (round 5)
// Under torch.autocast with more than 128 token rows every GEMM of the pass reads 16-bit A operands.  Rounds 1-4 kept all
// activations fp32 and converted in front of every GEMM (272 conversion launches = 10 % of a configs[4] step); now every
// producer writes the 16-bit copy its consumer GEMM reads, and the element-wise steps between two GEMMs are one kernel.
__device__ __forceinline__ void store16x4(unsigned short* dst, const f32x4& v, int f16) {
    const unsigned lo = to16_rt(v[0], f16) | (to16_rt(v[1], f16) << 16), hi = to16_rt(v[2], f16) | (to16_rt(v[3], f16) << 16);
    *reinterpret_cast<uint2*>(dst) = make_uint2(lo, hi);
}

// x = resid + dropout(y)  (resid_dropout / the embedding dropout when resid == nullptr; y may alias x), stored fp32 for the
// backward pass, and xn16 = round16(LayerNorm(x) * g + b) for the GEMM behind the LayerNorm.  One wave per row of 1024,
// reductions by DPP, two-pass variance like nn.LayerNorm (the arithmetic of ln_rows_kernel, decoder.hip).
__device__ __forceinline__ f32x4 load16x4(const unsigned short* src, int f16) {
    const uint2 r = *reinterpret_cast<const uint2*>(src);
    return f32x4{from16_rt(r.x & 0xffffu, f16), from16_rt(r.x >> 16, f16), from16_rt(r.y & 0xffffu, f16), from16_rt(r.y >> 16, f16)};
}
// y comes as fp32 (y) or as 16 bit (y16: the projection GEMMs of the 16-bit flow write their result in the autocast type, as
// the reference's do under torch.autocast) - exactly one of the two is non-null
__global__ __launch_bounds__(256) void resid_dropout_ln16_kernel(const float* y, const unsigned short* __restrict__ y16,
                                                                 const float* __restrict__ resid, float* x,
                                                                 const float* __restrict__ g, const float* __restrict__ b,
                                                                 unsigned short* __restrict__ xn16, const DropoutParams drop, int f16,
                                                                 int rows) {
    const int lane = threadIdx.x & 63, row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const size_t base = (size_t)row * 1024;
    f32x4 v[4], rr[4], gg[4], bb[4];
#pragma unroll
    for (int j = 0; j < 4; ++j)
        v[j] = y16 ? load16x4(y16 + base + 4 * (j * 64 + lane), f16) : reinterpret_cast<const f32x4*>(y + base)[j * 64 + lane];
#pragma unroll
    for (int j = 0; j < 4; ++j) rr[j] = resid ? reinterpret_cast<const f32x4*>(resid + base)[j * 64 + lane] : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        gg[j] = reinterpret_cast<const f32x4*>(g)[j * 64 + lane];
        bb[j] = reinterpret_cast<const f32x4*>(b)[j * 64 + lane];
    }
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        float mk[4];
        dropout_mask4(drop, (base >> 2) + j * 64 + lane, mk);
#pragma unroll
        for (int e = 0; e < 4; ++e) v[j][e] = rr[j][e] + v[j][e] * mk[e];
        reinterpret_cast<f32x4*>(x + base)[j * 64 + lane] = v[j];
        s += (v[j][0] + v[j][1]) + (v[j][2] + v[j][3]);
    }
    const float mean = wave_sum_dpp(s) * (1.0f / 1024.0f);
    float q = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int e = 0; e < 4; ++e) { v[j][e] -= mean; q += v[j][e] * v[j][e]; }
    const float rstd = 1.0f / sqrtf(wave_sum_dpp(q) * (1.0f / 1024.0f) + LN_EPS_T);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        f32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = v[j][e] * rstd * gg[j][e] + bb[j][e];
        store16x4(xn16 + base + 4 * (j * 64 + lane), o, f16);
    }
}

// LayerNorm backward w.r.t. its input (as ln_backward_kernel), one wave per row, plus the 16-bit copy the next dgrad GEMM
// reads: out16 = round16(out * mask) - the dropout mask of the residual branch the gradient enters next (p = 0: plain copy).
__global__ __launch_bounds__(256) void ln_backward16_kernel(const float* __restrict__ dy, const unsigned short* __restrict__ dy16,
                                                            const float* __restrict__ x, const float* __restrict__ g, float* out,
                                                            unsigned short* __restrict__ out16, int accumulate,
                                                            const DropoutParams drop, int f16, int rows) {
    const int lane = threadIdx.x & 63, row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const size_t base = (size_t)row * 1024;
    f32x4 v[4], dd[4], gg[4], oo[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) v[j] = reinterpret_cast<const f32x4*>(x + base)[j * 64 + lane];
#pragma unroll
    for (int j = 0; j < 4; ++j)
        dd[j] = dy16 ? load16x4(dy16 + base + 4 * (j * 64 + lane), f16) : reinterpret_cast<const f32x4*>(dy + base)[j * 64 + lane];
#pragma unroll
    for (int j = 0; j < 4; ++j) gg[j] = reinterpret_cast<const f32x4*>(g)[j * 64 + lane];
#pragma unroll
    for (int j = 0; j < 4; ++j) oo[j] = accumulate ? reinterpret_cast<const f32x4*>(out + base)[j * 64 + lane] : f32x4{0.f, 0.f, 0.f, 0.f};
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j) s += (v[j][0] + v[j][1]) + (v[j][2] + v[j][3]);
    const float mean = wave_sum_dpp(s) * (1.0f / 1024.0f);
    float q = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int e = 0; e < 4; ++e) { v[j][e] -= mean; q += v[j][e] * v[j][e]; }
    const float rstd = 1.0f / sqrtf(wave_sum_dpp(q) * (1.0f / 1024.0f) + LN_EPS_T);
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            dd[j][e] *= gg[j][e];        // t
            v[j][e] *= rstd;             // xhat
            s1 += dd[j][e];
            s2 += dd[j][e] * v[j][e];
        }
    const float m1 = wave_sum_dpp(s1) * (1.0f / 1024.0f), m2 = wave_sum_dpp(s2) * (1.0f / 1024.0f);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        f32x4 m;
        float mk[4];
        dropout_mask4(drop, (base >> 2) + j * 64 + lane, mk);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            oo[j][e] += rstd * (dd[j][e] - m1 - v[j][e] * m2);
            m[e] = oo[j][e] * mk[e];
        }
        reinterpret_cast<f32x4*>(out + base)[j * 64 + lane] = oo[j];
        if (out16) store16x4(out16 + base + 4 * (j * 64 + lane), m, f16);
    }
}

// d logits as ce_backward_kernel, written as 16 bit to a separate [rows, ld] buffer (the A operand of the lm_head dgrad);
// the K-padding columns [V, ld) of that buffer are zeroed once at allocation and never written.
__global__ __launch_bounds__(256) void ce_backward16_kernel(const float* __restrict__ logits, size_t ld, int V, int row0,
                                                            const long long* __restrict__ ids, const int* __restrict__ row_valid,
                                                            const float* __restrict__ row_lse, const int* __restrict__ n_scored,
                                                            float scale, const int* __restrict__ id_error,
                                                            unsigned short* __restrict__ out16, int f16) {
    const int r = row0 + blockIdx.x;
    const float* x = logits + (size_t)blockIdx.x * ld;
    unsigned short* o = out16 + (size_t)blockIdx.x * ld;
    const int V4 = V & ~3;
    if (!row_valid[r]) {
        for (int i = threadIdx.x * 4; i < V4; i += 1024) *reinterpret_cast<uint2*>(o + i) = make_uint2(0u, 0u);
        for (int i = V4 + threadIdx.x; i < V; i += 256) o[i] = 0;
        return;
    }
    const float lse = row_lse[r], f = *id_error ? nanf("") : scale / (float)(*n_scored);
    const int label = (int)ids[r + 1];
    for (int i = threadIdx.x * 4; i < V4; i += 1024) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(x + i);
        f32x4 gq;
#pragma unroll
        for (int e = 0; e < 4; ++e) gq[e] = (expf(v[e] - lse) - (i + e == label ? 1.f : 0.f)) * f;
        store16x4(o + i, gq, f16);
    }
    for (int i = V4 + threadIdx.x; i < V; i += 256) o[i] = (unsigned short)to16_rt((expf(x[i] - lse) - (i == label ? 1.f : 0.f)) * f, f16);
}

// d logits of CrossEntropyLoss(ignore_index=-100, mean) in place on a chunk of logits rows:
//   scored row: (softmax - onehot(label)) * scale / n_scored ; other rows (and the padding columns) : 0
__global__ __launch_bounds__(256) void ce_backward_kernel(float* __restrict__ logits, size_t ld, int V, int row0,
                                                          const long long* __restrict__ ids, const int* __restrict__ row_valid,
                                                          const float* __restrict__ row_lse, const int* __restrict__ n_scored,
                                                          float scale, const int* __restrict__ id_error) {
    const int r = row0 + blockIdx.x;
    float* x = logits + (size_t)blockIdx.x * ld;
    if (!row_valid[r]) {
        for (int i = threadIdx.x; i < V; i += 256) x[i] = 0.f;
        return;
    }
    // an invalid token id in this pass (ids were clamped for the loads): the gradients are poisoned like the loss, so an
    // optimizer step taken before the error is reported cannot apply finite-but-wrong updates
    const float lse = row_lse[r], f = *id_error ? nanf("") : scale / (float)(*n_scored);
    const int label = (int)ids[r + 1];
    for (int i = threadIdx.x; i < V; i += 256) x[i] = (expf(x[i] - lse) - (i == label ? 1.f : 0.f)) * f;
}

// dst[c][r] = src[r][c] for r < R, 0 for R <= r < Rp   (K-padding of a transposed GEMM operand)
__global__ __launch_bounds__(256) void transpose_pad_kernel(const float* __restrict__ src, float* __restrict__ dst, int R, int Cc,
                                                            int Rp) {
    __shared__ float tile[32][33];
    const int bx = blockIdx.x * 32, by = blockIdx.y * 32;  // bx: column block of src, by: row block of src
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
    for (int j = ty; j < 32; j += 8) {
        const int r = by + j, c = bx + tx;
        tile[j][tx] = (r < R && c < Cc) ? src[(size_t)r * Cc + c] : 0.f;
    }
    __syncthreads();
    for (int j = ty; j < 32; j += 8) {
        const int c = bx + j, r = by + tx;
        if (c < Cc && r < Rp) dst[(size_t)c * Rp + r] = tile[tx][j];
    }
}

// out[c] = sum_r src[r][c]  (bias gradients), rows added in index order
__global__ __launch_bounds__(256) void colsum_kernel(const float* __restrict__ src, float* __restrict__ out, int R, int Cc) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= Cc) return;
    float a = 0.f;
    for (int r = 0; r < R; ++r) a += src[(size_t)r * Cc + c];
    out[c] = a;
}

// ---------------------------------------------------------------- attention backward on the fp32 matrix core
// Same register-only structure as attn_prefill_kernel (decoder.hip).  The forward pass kept, per (token, head), the
// row log-sum-exp; delta = rowsum(dO . O) (= sum_keys P dP) comes from attn_delta_kernel.  Two kernels, no atomics:
//   dQ : one wave per (sequence, head, 32-query tile), TRANSPOSED tiles S^T = K Q^T and dP^T = V dO^T (a lane = one
//        query), dS^T registers are the B operand of dQ^T = K^T dS^T;
//   dK, dV : one wave per (sequence, head, 32-key tile) looping over the query tiles that can see it, UNTRANSPOSED
//        tiles S = Q K^T and dP = dO V^T (a lane = one key), P / dS registers are the B operand of
//        dV^T = dO^T P and dK^T = Q^T dS.
typedef float f32x16 __attribute__((ext_vector_type(16)));

__global__ __launch_bounds__(256) void attn_delta_kernel(const float* __restrict__ dO, const float* __restrict__ O,
                                                         float* __restrict__ delta, int D, int H) {
    const int row = blockIdx.x, tid = threadIdx.x;
    const f32x4 a = reinterpret_cast<const f32x4*>(dO + (size_t)row * D)[tid];
    const f32x4 b = reinterpret_cast<const f32x4*>(O + (size_t)row * D)[tid];
    float v = (a[0] * b[0] + a[1] * b[1]) + (a[2] * b[2] + a[3] * b[3]);
    v += __shfl_xor(v, 1, 64);
    v += __shfl_xor(v, 2, 64);
    v += __shfl_xor(v, 4, 64);
    v += __shfl_xor(v, 8, 64);
    if ((tid & 15) == 0) delta[(size_t)row * H + (tid >> 4)] = v;  // 16 threads x 4 floats = one 64-wide head
}

__device__ __forceinline__ int mfma_row(int r, int half) { return (r & 3) + 8 * (r >> 2) + 4 * half; }

// Key tiles are streamed (any T): dS^T of one 32-key tile is computed and consumed by dQ^T += K^T dS^T before the next.
__global__ __launch_bounds__(256) void attn_bwd_dq_kernel(const float* __restrict__ qkv, const float* __restrict__ ukv, int ld_ukv,
                                                          int kcol, const float* __restrict__ am, const float* __restrict__ d_att,
                                                          const float* __restrict__ lse, const float* __restrict__ delta,
                                                          float* __restrict__ d_qkv, int S, int H, int T, const DropoutParams drop,
                                                          unsigned short* __restrict__ d_qkv16, int f16) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int QT = (T + 31) / 32;
    const int item = blockIdx.x * 4 + wave;
    if (item >= S * H * QT) return;
    const int qt = item % QT, sh = item / QT, hd = sh % H, s = sh / H;
    const int NK = T + 1, D = H * 64;
    const int col = lane & 31, half = lane >> 5;
    const int iq = qt * 32 + col, iqc = min(iq, T - 1);
    const int need = min(NK - 1, qt * 32 + 32) / 32 + 1;  // key tiles a query of this tile can see
    auto krow = [&](int c) -> const float* {
        c = min(c, NK - 1);
        return c == 0 ? ukv + (size_t)s * ld_ukv + kcol + hd * 64 : qkv + ((size_t)s * T + c - 1) * 3 * D + D + hd * 64;
    };
    f32x4 qf[8], gf[8];
    {
        const float* qp = qkv + ((size_t)s * T + iqc) * 3 * D + hd * 64 + half * 32;
        const float* gp = d_att + ((size_t)s * T + iqc) * D + hd * 64 + half * 32;
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            qf[u] = *reinterpret_cast<const f32x4*>(qp + 4 * u);
            gf[u] = *reinterpret_cast<const f32x4*>(gp + 4 * u);
        }
    }
    const float lse_q = lse[((size_t)s * T + iqc) * H + hd], delta_q = delta[((size_t)s * T + iqc) * H + hd];
    f32x16 o0, o1;
#pragma unroll
    for (int r = 0; r < 16; ++r) { o0[r] = 0.f; o1[r] = 0.f; }
    for (int kt = 0; kt < need; ++kt) {
        f32x4 kf[8], vf[8];
        const float* kp = krow(kt * 32 + col) + half * 32;
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            kf[u] = *reinterpret_cast<const f32x4*>(kp + 4 * u);
            vf[u] = *reinterpret_cast<const f32x4*>(kp + D + 4 * u);
        }
        float k0[16], k1[16];  // the same K rows again, a lane per dim: A operand of dQ^T = K^T dS^T
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const float* kr = krow(kt * 32 + mfma_row(j, half));
            k0[j] = kr[col];
            k1[j] = kr[32 + col];
        }
        f32x16 aS, aP;
#pragma unroll
        for (int r = 0; r < 16; ++r) { aS[r] = 0.f; aP[r] = 0.f; }
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                aS = __builtin_amdgcn_mfma_f32_32x32x2f32(kf[u][e], qf[u][e], aS, 0, 0, 0);
                aP = __builtin_amdgcn_mfma_f32_32x32x2f32(vf[u][e], gf[u][e], aP, 0, 0, 0);
            }
        f32x16 ds;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int c = kt * 32 + mfma_row(r, half);
            float dsv = 0.f;
            if (c < NK) {
                const bool allowed = (c == 0) || (c - 1 <= iq);
                const float addm = (c == 0 || !am) ? 0.f : (1.0f - am[(size_t)s * T + c - 1]) * -10000.0f;
                const float pr = expf((allowed ? aS[r] / 8.0f : -1e4f) + addm - lse_q);
                // attn_dropout: O = (P * mask) V  =>  dP = (dO . V) * mask ; delta = rowsum(dO . O) already includes it
                const float mk = dropout_mask(drop, (((unsigned long long)s * H + hd) * T + iqc) * dropout_key_pitch(NK) + c);
                dsv = allowed ? pr * (aP[r] * mk - delta_q) / 8.0f : 0.f;
            }
            ds[r] = dsv;
        }
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            o0 = __builtin_amdgcn_mfma_f32_32x32x2f32(k0[j], ds[j], o0, 0, 0, 0);
            o1 = __builtin_amdgcn_mfma_f32_32x32x2f32(k1[j], ds[j], o1, 0, 0, 0);
        }
    }
    if (iq < T) {
        const size_t off = ((size_t)s * T + iq) * 3 * D + hd * 64;
        if (d_qkv16) {   // 16-bit training flow: the only reader is the c_attn dgrad GEMM (4 consecutive dims per store)
#pragma unroll
            for (int q4 = 0; q4 < 4; ++q4) {
                store16x4(d_qkv16 + off + 8 * q4 + 4 * half, f32x4{o0[4 * q4], o0[4 * q4 + 1], o0[4 * q4 + 2], o0[4 * q4 + 3]}, f16);
                store16x4(d_qkv16 + off + 32 + 8 * q4 + 4 * half, f32x4{o1[4 * q4], o1[4 * q4 + 1], o1[4 * q4 + 2], o1[4 * q4 + 3]}, f16);
            }
        } else {
            float* op = d_qkv + off;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                op[mfma_row(r, half)] = o0[r];
                op[32 + mfma_row(r, half)] = o1[r];
            }
        }
    }
}

__global__ __launch_bounds__(256) void attn_bwd_dkv_kernel(const float* __restrict__ qkv, const float* __restrict__ ukv, int ld_ukv,
                                                           int kcol, const float* __restrict__ am, const float* __restrict__ d_att,
                                                           const float* __restrict__ lse, const float* __restrict__ delta,
                                                           float* __restrict__ d_qkv, float* __restrict__ d_ukv, int S, int H, int T,
                                                           const DropoutParams drop, unsigned short* __restrict__ d_qkv16, int f16,
                                                           float ukv_scale) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int NK = T + 1, D = H * 64;
    const int KT = (NK + 31) / 32, QT = (T + 31) / 32;
    const int item = blockIdx.x * 4 + wave;
    if (item >= S * H * KT) return;
    const int kt = item % KT, sh = item / KT, hd = sh % H, s = sh / H;
    const int col = lane & 31, half = lane >> 5;
    const int c = kt * 32 + col, cc = min(c, NK - 1);  // this lane's key (column of S)
    const bool cvalid = c < NK;
    const float* kp = (cc == 0 ? ukv + (size_t)s * ld_ukv + kcol + hd * 64 : qkv + ((size_t)s * T + cc - 1) * 3 * D + D + hd * 64) + half * 32;
    f32x4 kf[8], vf[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
        kf[u] = *reinterpret_cast<const f32x4*>(kp + 4 * u);
        vf[u] = *reinterpret_cast<const f32x4*>(kp + D + 4 * u);
    }
    const float addm = (cc == 0 || !am) ? 0.f : (1.0f - am[(size_t)s * T + cc - 1]) * -10000.0f;
    f32x16 dk0, dk1, dv0, dv1;
#pragma unroll
    for (int r = 0; r < 16; ++r) { dk0[r] = 0.f; dk1[r] = 0.f; dv0[r] = 0.f; dv1[r] = 0.f; }
    const int qt0 = kt == 0 ? 0 : kt - 1;  // first query tile with a query i >= (first key of the tile) - 1
    for (int qt = qt0; qt < QT; ++qt) {
        f32x4 qf[8], gf[8];
        {
            const int i = min(qt * 32 + col, T - 1);
            const float* qp = qkv + ((size_t)s * T + i) * 3 * D + hd * 64 + half * 32;
            const float* gp = d_att + ((size_t)s * T + i) * D + hd * 64 + half * 32;
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                qf[u] = *reinterpret_cast<const f32x4*>(qp + 4 * u);
                gf[u] = *reinterpret_cast<const f32x4*>(gp + 4 * u);
            }
        }
        f32x16 aS, aP;
#pragma unroll
        for (int r = 0; r < 16; ++r) { aS[r] = 0.f; aP[r] = 0.f; }
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                aS = __builtin_amdgcn_mfma_f32_32x32x2f32(qf[u][e], kf[u][e], aS, 0, 0, 0);
                aP = __builtin_amdgcn_mfma_f32_32x32x2f32(gf[u][e], vf[u][e], aP, 0, 0, 0);
            }
        float a_q0[16], a_q1[16], a_g0[16], a_g1[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int i = qt * 32 + mfma_row(r, half), ic = min(i, T - 1);
            float pr = 0.f, dsv = 0.f;
            if (i < T && cvalid) {
                const float lse_i = lse[((size_t)s * T + i) * H + hd], delta_i = delta[((size_t)s * T + i) * H + hd];
                const bool allowed = (c == 0) || (c - 1 <= i);
                pr = expf((allowed ? aS[r] / 8.0f : -1e4f) + addm - lse_i);
                const float mk = dropout_mask(drop, (((unsigned long long)s * H + hd) * T + i) * dropout_key_pitch(NK) + c);
                dsv = allowed ? pr * (aP[r] * mk - delta_i) / 8.0f : 0.f;
                pr *= mk;  // dV^T = dO^T (P * mask)
            }
            aS[r] = pr;   // P (dropped) [query r][key col]
            aP[r] = dsv;  // dS
            const float* qp = qkv + ((size_t)s * T + ic) * 3 * D + hd * 64;
            const float* gp = d_att + ((size_t)s * T + ic) * D + hd * 64;
            a_q0[r] = qp[col]; a_q1[r] = qp[32 + col];
            a_g0[r] = gp[col]; a_g1[r] = gp[32 + col];
        }
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            dv0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a_g0[j], aS[j], dv0, 0, 0, 0);
            dv1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a_g1[j], aS[j], dv1, 0, 0, 0);
            dk0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a_q0[j], aP[j], dk0, 0, 0, 0);
            dk1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a_q1[j], aP[j], dk1, 0, 0, 0);
        }
    }
    if (cvalid) {
        if (c != 0 && d_qkv16) {   // 16-bit training flow (see attn_bwd_dq_kernel)
            unsigned short* kd = d_qkv16 + ((size_t)s * T + c - 1) * 3 * D + D + hd * 64;
            unsigned short* vd = kd + D;
#pragma unroll
            for (int q4 = 0; q4 < 4; ++q4) {
                store16x4(kd + 8 * q4 + 4 * half, f32x4{dk0[4 * q4], dk0[4 * q4 + 1], dk0[4 * q4 + 2], dk0[4 * q4 + 3]}, f16);
                store16x4(kd + 32 + 8 * q4 + 4 * half, f32x4{dk1[4 * q4], dk1[4 * q4 + 1], dk1[4 * q4 + 2], dk1[4 * q4 + 3]}, f16);
                store16x4(vd + 8 * q4 + 4 * half, f32x4{dv0[4 * q4], dv0[4 * q4 + 1], dv0[4 * q4 + 2], dv0[4 * q4 + 3]}, f16);
                store16x4(vd + 32 + 8 * q4 + 4 * half, f32x4{dv1[4 * q4], dv1[4 * q4 + 1], dv1[4 * q4 + 2], dv1[4 * q4 + 3]}, f16);
            }
        } else {
            float* kdst;
            float* vdst;
            float sc = 1.0f;
            if (c == 0) {
                kdst = d_ukv + (size_t)s * ld_ukv + kcol + hd * 64;
                vdst = kdst + D;
                sc = ukv_scale;   // undoes the internal loss scale of the fp16 flow (1 otherwise: exact)
            } else {
                kdst = d_qkv + ((size_t)s * T + c - 1) * 3 * D + D + hd * 64;
                vdst = kdst + D;
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int dim = mfma_row(r, half);
                kdst[dim] = dk0[r] * sc;
                kdst[32 + dim] = dk1[r] * sc;
                vdst[dim] = dv0[r] * sc;
                vdst[32 + dim] = dv1[r] * sc;
            }
        }
    }
}

// d logits of BCEWithLogitsLoss(pos_weight) over the rows with mask != 0 (mean): ((1-y) - lw + lw sigmoid(x)) * scale / n,
// lw = 1 + (w-1) y; other rows 0.  dlogits has row stride ld (K-padding of the next GEMM; only column 0 is written).
__global__ __launch_bounds__(256) void bce_backward_kernel(const float* __restrict__ logits, const unsigned char* __restrict__ mask,
                                                           const unsigned char* __restrict__ target, float pos_weight, int n,
                                                           float scale, float* __restrict__ dlogits, int ld) {
    __shared__ int scnt[256];
    int c = 0;
    for (int i = threadIdx.x; i < n; i += 256) c += mask[i] ? 1 : 0;
    scnt[threadIdx.x] = c;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (threadIdx.x < o) scnt[threadIdx.x] += scnt[threadIdx.x + o];
        __syncthreads();
    }
    const float f = scale / (float)scnt[0];
    for (int i = threadIdx.x; i < n; i += 256) {
        float g = 0.f;
        if (mask[i]) {
            const float x = logits[i], y = target[i] ? 1.f : 0.f;
            const float lw = (pos_weight - 1.f) * y + 1.f;
            const float sg = 1.0f / (1.0f + expf(-x));
            g = ((1.f - y) - lw + lw * sg) * f;
        }
        dlogits[(size_t)i * ld] = g;
    }
}

// AdamW (torch.optim.AdamW semantics, decoupled weight decay): one element per thread
//   p *= 1 - lr wd ; m = b1 m + (1-b1) g ; v = b2 v + (1-b2) g^2 ; p -= lr / (1-b1^t) * m / (sqrt(v) / sqrt(1-b2^t) + eps)
__global__ __launch_bounds__(256) void adamw_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                    float* __restrict__ v, size_t n, float lr, float b1, float b2, float eps,
                                                    float wd, float bc1, float bc2_sqrt, float grad_scale) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const float gi = g[i] * grad_scale;
        float pi = p[i] * (1.0f - lr * wd);
        const float mi = b1 * m[i] + (1.0f - b1) * gi;
        const float vi = b2 * v[i] + (1.0f - b2) * gi * gi;
        m[i] = mi;
        v[i] = vi;
        pi -= (lr / bc1) * (mi / (sqrtf(vi) / bc2_sqrt + eps));
        p[i] = pi;
    }
}

// The same update for MANY tensors in one launch (round 5: a configs[4] step updated its 112 trainable tensors with 112
// launches of ~6 us): blockIdx.y = tensor, blockIdx.x strides over its elements; hyper-parameters and step are shared.
struct AdamItem { float* p; const float* g; float* m; float* v; long long n; };
constexpr int ADAM_BATCH = 64;                       // records per launch: they travel in the kernel arguments (2.5 KiB of the 4 KiB)
struct AdamBatch { AdamItem it[ADAM_BATCH]; };
__global__ __launch_bounds__(256) void adamw_multi_kernel(const AdamBatch batch, float lr, float b1, float b2, float eps,
                                                          float wd, float bc1, float bc2_sqrt, float grad_scale) {
    const AdamItem it = batch.it[blockIdx.y];
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < (size_t)it.n; i += (size_t)gridDim.x * 256) {
        const float gi = it.g[i] * grad_scale;
        float pi = it.p[i] * (1.0f - lr * wd);
        const float mi = b1 * it.m[i] + (1.0f - b1) * gi;
        const float vi = b2 * it.v[i] + (1.0f - b2) * gi * gi;
        it.m[i] = mi;
        it.v[i] = vi;
        pi -= (lr / bc1) * (mi / (sqrtf(vi) / bc2_sqrt + eps));
        it.p[i] = pi;
    }
}

// ------------------------------------------------------------------ launchers (decoder.hip)
static int blocks_for(size_t n) { return (int)((n + 255) / 256 < 8192 ? (n + 255) / 256 : 8192); }

int launch_gelu_apply(const float* pre, float* out, size_t n, hipStream_t st) {
    hipLaunchKernelGGL(gelu_apply_kernel, dim3(blocks_for(n / 4)), dim3(256), 0, st, pre, out, n / 4);
    RGRG_LAUNCH_CHECK();
    return RGRG_OK;
}
int launch_gelu_backward(float* d, const float* pre, size_t n, hipStream_t st) {
    hipLaunchKernelGGL(gelu_backward_kernel, dim3(blocks_for(n / 4)), dim3(256), 0, st, d, pre, n / 4);
    RGRG_LAUNCH_CHECK();
    return RGRG_OK;
}
int launch_dropout_add(const float* src, const float* resid, float* out, size_t n, DropoutParams drop, hipStream_t st) {
    hipLaunchKernelGGL(dropout_add_kernel, dim3(blocks_for(n)), dim3(256), 0, st, src, resid, out, n, drop);
    RGRG_LAUNCH_CHECK();
    return RGRG_OK;
}
int launch_relu_backward(float* d, const float* h, size_t n, hipStream_t st) {
    hipLaunchKernelGGL(relu_backward_kernel, dim3(blocks_for(n)), dim3(256), 0, st, d, h, n);
    RGRG_LAUNCH_CHECK();
    return RGRG_OK;
}
int launch_ln_backward(const float* dy, const float* x, const float* g, float* out, int rows, int D, int accumulate, hipStream_t st) {
    RGRG_CHECK_ARG(D == 1024 && rows > 0);
    hipLaunchKernelGGL(ln_backward_kernel, dim3(rows), dim3(256), 0, st, dy, x, g, out, D, accumulate);
    RGRG_LAUNCH_CHECK();
    return RGRG_OK;
}
int launch_ce_backward(float* logits, size_t ld, int V, int row0, int rows, const long long* ids, const int* row_valid,
                       const float* row_lse, const int* n_scored, float scale, const int* id_error, hipStream_t st) {
    hipLaunchKernelGGL(ce_backward_kernel, dim3(rows), dim3(256), 0, st, logits, ld, V, row0, ids, row_valid, row_lse, n_scored, scale,
                       id_error);
    RGRG_LAUNCH_CHECK();
    return RGRG_OK;
}
int launch_resid_dropout_ln16(const float* y, const unsigned short* y16, const float* resid, float* x, const float* g, const float* b,
                              unsigned short* xn16, DropoutParams drop, int f16, int rows, int D, hipStream_t st) {
    RGRG_CHECK_ARG(D == 1024 && rows > 0 && ((y != nullptr) != (y16 != nullptr)) && x && xn16);
    hipLaunchKernelGGL(resid_dropout_ln16_kernel, dim3((rows + 3) / 4), dim3(256), 0, st, y, y16, resid, x, g, b, xn16, drop, f16, rows);
    RGRG_LAUNCH_CHECK();
    return RGRG_OK;
}
int launch_ln_backward16(const float* dy, const unsigned short* dy16, const float* x, const float* g, float* out, unsigned short* out16,
                         int rows, int D, int accumulate, DropoutParams drop, int f16, hipStream_t st) {
    RGRG_CHECK_ARG(D == 1024 && rows > 0 && ((dy != nullptr) != (dy16 != nullptr)));
    hipLaunchKernelGGL(ln_backward16_kernel, dim3((rows + 3) / 4), dim3(256), 0, st, dy, dy16, x, g, out, out16, accumulate, drop, f16, rows);
    RGRG_LAUNCH_CHECK();
    return RGRG_OK;
}
int launch_ce_backward16(const float* logits, size_t ld, int V, int row0, int rows, const long long* ids, const int* row_valid,
                         const float* row_lse, const int* n_scored, float scale, const int* id_error, unsigned short* out16, int f16,
                         hipStream_t st) {
    hipLaunchKernelGGL(ce_backward16_kernel, dim3(rows), dim3(256), 0, st, logits, ld, V, row0, ids, row_valid, row_lse, n_scored, scale,
                       id_error, out16, f16);
    RGRG_LAUNCH_CHECK();
    return RGRG_OK;
}
int launch_transpose_pad(const float* src, float* dst, int R, int Cc, int Rp, hipStream_t st) {
    RGRG_CHECK_ARG(R > 0 && Cc > 0 && Rp >= R);
    hipLaunchKernelGGL(transpose_pad_kernel, dim3((Cc + 31) / 32, (Rp + 31) / 32), dim3(256), 0, st, src, dst, R, Cc, Rp);
    RGRG_LAUNCH_CHECK();
    return RGRG_OK;
}
int launch_colsum(const float* src, float* out, int R, int Cc, hipStream_t st) {
    hipLaunchKernelGGL(colsum_kernel, dim3((Cc + 255) / 256), dim3(256), 0, st, src, out, R, Cc);
    RGRG_LAUNCH_CHECK();
    return RGRG_OK;
}
int launch_attn_backward(const float* qkv, const float* ukv, int ld_ukv, int kcol, const float* am, const float* d_att,
                         const float* att, const float* lse, float* delta, float* d_qkv, float* d_ukv, int S, int H, int T,
                         DropoutParams drop, hipStream_t st, unsigned short* d_qkv16, int f16, float ukv_scale) {
    RGRG_CHECK_ARG(T >= 1 && att && lse && delta);
    const int NK = T + 1, D = H * 64, M = S * T;
    hipLaunchKernelGGL(attn_delta_kernel, dim3(M), dim3(256), 0, st, d_att, att, delta, D, H);
    RGRG_LAUNCH_CHECK();
    const int qitems = S * H * ((T + 31) / 32), kitems = S * H * ((NK + 31) / 32);
    hipLaunchKernelGGL(attn_bwd_dq_kernel, dim3((qitems + 3) / 4), dim3(256), 0, st, qkv, ukv, ld_ukv, kcol, am, d_att, lse, delta,
                       d_qkv, S, H, T, drop, d_qkv16, f16);
    RGRG_LAUNCH_CHECK();
    hipLaunchKernelGGL(attn_bwd_dkv_kernel, dim3((kitems + 3) / 4), dim3(256), 0, st, qkv, ukv, ld_ukv, kcol, am, d_att, lse, delta,
                       d_qkv, d_ukv, S, H, T, drop, d_qkv16, f16, ukv_scale);
    RGRG_LAUNCH_CHECK();
    return RGRG_OK;
}

}  // namespace rgrg

using namespace rgrg;

extern "C" int rgrg_transpose_pad_f32(const float* src, float* dst, int rows, int cols, int rows_padded, void* stream) {
    RGRG_CHECK_ARG(src && dst);
    return launch_transpose_pad(src, dst, rows, cols, rows_padded, as_stream(stream));
}

extern "C" int rgrg_colsum_f32(const float* src, float* out, int rows, int cols, void* stream) {
    RGRG_CHECK_ARG(src && out && rows > 0 && cols > 0);
    return launch_colsum(src, out, rows, cols, as_stream(stream));
}

extern "C" int rgrg_relu_backward_f32(float* d, const float* h, int64_t n, void* stream) {
    RGRG_CHECK_ARG(d && h && n > 0);
    return launch_relu_backward(d, h, (size_t)n, as_stream(stream));
}

extern "C" int rgrg_bce_with_logits_masked_backward_f32(const float* logits, const uint8_t* mask, const uint8_t* target,
                                                        float pos_weight, int n, float scale, float* dlogits, int ld,
                                                        void* stream) {
    RGRG_CHECK_ARG(logits && mask && target && dlogits && n > 0 && ld >= 1);
    hipLaunchKernelGGL(bce_backward_kernel, dim3(1), dim3(256), 0, as_stream(stream), logits, mask, target, pos_weight, n, scale,
                       dlogits, ld);
    RGRG_LAUNCH_CHECK();
    return RGRG_OK;
}

extern "C" int rgrg_dropout_mask_f32(uint64_t seed, uint32_t stream_id, float p, int64_t n, int row_len, float* out, void* stream) {
    RGRG_CHECK_ARG(out && n > 0 && p >= 0.f && p < 1.f && row_len >= 0);
    hipLaunchKernelGGL(dropout_mask_kernel, dim3(blocks_for((size_t)n)), dim3(256), 0, as_stream(stream), out, (size_t)n,
                       DropoutParams{seed, stream_id, p}, row_len);
    RGRG_LAUNCH_CHECK();
    return RGRG_OK;
}

extern "C" int rgrg_adamw_step_f32(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n, float lr,
                                   float beta1, float beta2, float eps, float weight_decay, int step, float grad_scale,
                                   void* stream) {
    RGRG_CHECK_ARG(param && grad && exp_avg && exp_avg_sq && n > 0 && step >= 1 && lr >= 0.f);
    const float bc1 = 1.0f - powf(beta1, (float)step);
    const float bc2_sqrt = sqrtf(1.0f - powf(beta2, (float)step));
    hipLaunchKernelGGL(adamw_kernel, dim3(blocks_for((size_t)n)), dim3(256), 0, as_stream(stream), param, grad, exp_avg, exp_avg_sq,
                       (size_t)n, lr, beta1, beta2, eps, weight_decay, bc1, bc2_sqrt, grad_scale);
    RGRG_LAUNCH_CHECK();
    return RGRG_OK;
}

// items: HOST array of n_items records {param, grad, exp_avg, exp_avg_sq (device pointers to f32), element count (int64)} - five
// 64-bit words each.  The records ride in the kernel arguments (64 per launch): no device-side table to keep in sync with
// gradients that are re-allocated every step (zero_grad(set_to_none=True)), no host-to-device copy, nothing to wait for.
extern "C" int rgrg_adamw_multi_step_f32(const void* items, int n_items, float lr, float beta1, float beta2, float eps,
                                         float weight_decay, int step, float grad_scale, void* stream) {
    RGRG_CHECK_ARG(items && n_items > 0 && step >= 1 && lr >= 0.f);
    static_assert(sizeof(AdamItem) == 40, "five 64-bit words per record");
    const float bc1 = 1.0f - powf(beta1, (float)step);
    const float bc2_sqrt = sqrtf(1.0f - powf(beta2, (float)step));
    const AdamItem* src = reinterpret_cast<const AdamItem*>(items);
    for (int i0 = 0; i0 < n_items; i0 += ADAM_BATCH) {
        const int n = n_items - i0 < ADAM_BATCH ? n_items - i0 : ADAM_BATCH;
        AdamBatch b{};
        long long max_n = 1;
        for (int i = 0; i < n; ++i) {
            b.it[i] = src[i0 + i];
            RGRG_CHECK_ARG(b.it[i].p && b.it[i].g && b.it[i].m && b.it[i].v && b.it[i].n > 0);
            if (b.it[i].n > max_n) max_n = b.it[i].n;
        }
        const int gx = (int)(((size_t)max_n + 255) / 256 < 1024 ? ((size_t)max_n + 255) / 256 : 1024);
        hipLaunchKernelGGL(adamw_multi_kernel, dim3(gx, n), dim3(256), 0, as_stream(stream), b, lr, beta1, beta2, eps, weight_decay, bc1,
                           bc2_sqrt, grad_scale);
        RGRG_LAUNCH_CHECK();
    }
    return RGRG_OK;
}
