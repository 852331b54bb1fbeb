// Pseudo self-attention of the TRAINING pass on the 16-bit matrix core (round 5; torch.autocast around the reference's
// training step, src/full_model/train_full_model.py:172-237: under autocast torch.matmul runs q k^T and p v in the autocast
// type, softmax in fp32 - src/language_model/language_model.py:84-122, :124-160).
//
// The fp32 kernels (decoder.hip attn_prefill_kernel, train_ops.hip attn_bwd_*) spend 17 of the 66 ms of a configs[4] step:
// one wave per (sentence, head, 32-query tile) at one wave per SIMD (250 registers), the exact-fp32 MFMA at 1/16 of the
// 16-bit rate, the V / K^T operands gathered from global memory one float per lane.  Here: one WORKGROUP per (sentence, head)
// for T + 1 <= 128 keys (the reports of the reference are cut to <= ~100 tokens; longer sequences keep the fp32 kernels),
//   * q / k / v arrive as 16 bit from c_attn's epilogue, d(attention output) as 16 bit from attn_proj's dgrad epilogue, the
//     image key / value of slot 0 from a 16-bit copy of uk(img) / uv(img) made once per pass;
//   * row-major operands (A = 32 rows x 16 dims, 16 bytes per lane) are read straight from global memory / L2; the operands
//     an MFMA needs TRANSPOSED (V^T for P V, K^T for dS K, Q^T and dO^T for dS^T Q / P^T dO) are transposed ONCE per workgroup
//     into LDS (ds_write_b16), with the key / query index permuted inside each group of 16 so that the 8 values a lane's
//     accumulator registers hold for one 16-wide K step are the 8 consecutive LDS elements of the other operand:
//         C layout of v_mfma_f32_32x32x16: register r = row (r & 3) + 8 (r >> 2) + 4 (lane >> 5)  =>  for K step u the registers
//         8u .. 8u + 7 of lane half h are the indices 16u + {0,1,2,3,8,9,10,11} + 4h  =>  position 16u + 8h + e <-> index
//         16u + (e & 3) + 8 (e >> 2) + 4h;
//   * scores are computed transposed (S^T = K Q^T: a lane = one query, its registers = keys) where the contraction that
//     follows runs over keys (P V, dS K), and untransposed (S = Q K^T: a lane = one key) where it runs over queries (dK, dV);
//   * softmax / masks / dropout / the dS arithmetic stay fp32 in registers; P and dS are rounded to 16 bit as MFMA operands,
//     O is normalised in fp32 after the product (P enters as exp(s - max) <= 1).
// Dropout masks are the counter-based ones of common.h (same index as the fp32 kernels), so forward and backward agree.
#include "common.h"

namespace rgrg {

typedef short h16x8 __attribute__((ext_vector_type(8)));
typedef unsigned short u16;

template <bool F16>
__device__ __forceinline__ f32x16 mfma_h(const h16x8& a, const h16x8& b, const f32x16& c) {
    if constexpr (F16) return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
    else return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}

constexpr int A16_MAXK = 128;             // keys (T + 1) per sentence
constexpr int A16_TP = A16_MAXK + 8;      // row pitch (elements) of a transposed LDS image: 272 B, conflict-free b128 reads
__host__ __device__ __forceinline__ int perm16(int i) {   // index inside a group of 16 -> position (see the header)
    const int k = i & 15;
    return (i & ~15) | (((k >> 2) & 1) << 3) | ((k >> 3) << 2) | (k & 3);
}
__device__ __forceinline__ int mrow(int r, int half) { return (r & 3) + 8 * (r >> 2) + 4 * half; }

// row `tok` (0 = the image slot, else token tok - 1) of the K (which = 1) / V (which = 2) operand of head hd: 64 16-bit values
__device__ __forceinline__ const u16* kv_row(const u16* qkv16, const u16* ukv16, int ld_ukv, int kcol, int s, int T, int D, int hd,
                                             int tok, int which) {
    return tok == 0 ? ukv16 + (size_t)s * ld_ukv + kcol + (which - 1) * D + hd * 64
                    : qkv16 + ((size_t)s * T + tok - 1) * 3 * D + which * D + hd * 64;
}

// Element (dim, position) of a transposed image: the 16-byte chunk of the position is XORed with dim / 8 - the eight lanes
// that transpose one source row write dims 8 ch + j, ch = 0..7, at the SAME position: without the swizzle all eight hit one
// bank (the row pitch is a multiple of 16 bytes), 74 % of the LDS cycles of the first version were bank conflicts.
__device__ __forceinline__ int timg(int dim, int pos) { return dim * A16_TP + ((((pos >> 3) ^ (dim >> 3)) << 3) | (pos & 7)); }
// dst[dim][perm16(idx)] = row[dim] for the 64 dims of `rows` rows (zero beyond nrows): 8 dims per thread step
template <typename RowFn>
__device__ __forceinline__ void stage_transposed(u16* dst, int rows_padded, int nrows, RowFn row_ptr) {
    for (int i = threadIdx.x; i < rows_padded * 8; i += blockDim.x) {
        const int idx = i >> 3, ch = i & 7;
        h16x8 v = {0, 0, 0, 0, 0, 0, 0, 0};
        if (idx < nrows) v = *reinterpret_cast<const h16x8*>(row_ptr(idx) + ch * 8);
        const int pos = perm16(idx);
#pragma unroll
        for (int j = 0; j < 8; ++j) dst[timg(ch * 8 + j, pos)] = (u16)v[j];
    }
}

template <bool F16>
__device__ __forceinline__ h16x8 pack8(const float (&x)[8]) {
    h16x8 o;
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = (short)to16<F16>(x[e]);
    return o;
}

// ---------------------------------------------------------------- forward
// grid = S * H workgroups of 256 threads; wave w = query tile w (32 queries).  out16 [S*T, D] 16 bit, lse [S*T, H] fp32.
template <bool F16>
__global__ __launch_bounds__(256) void attn16_fwd_kernel(const u16* __restrict__ qkv16, const u16* __restrict__ ukv16, int ld_ukv, int kcol,
                                                         const float* __restrict__ am, u16* __restrict__ out16, float* __restrict__ lse,
                                                         int S, int H, int T, const DropoutParams drop) {
    __shared__ __attribute__((aligned(16))) u16 VT[64 * A16_TP];
    __shared__ float addm_s[A16_MAXK];   // additive padding mask of every key (0 for the image key)
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int hd = blockIdx.x % H, s = blockIdx.x / H;
    const int NK = T + 1, D = H * 64, NKP = (NK + 31) & ~31, QT = (T + 31) / 32;
    stage_transposed(VT, NKP, NK, [&](int c) { return kv_row(qkv16, ukv16, ld_ukv, kcol, s, T, D, hd, c, 2); });
    for (int c = threadIdx.x; c < A16_MAXK; c += blockDim.x)
        addm_s[c] = (c == 0 || c >= NK || !am) ? 0.f : (1.0f - am[(size_t)s * T + c - 1]) * -10000.0f;
    __syncthreads();
    if (wave >= QT) return;
    const int qt = wave, col = lane & 31, half = lane >> 5;
    const int iq = qt * 32 + col, iqc = min(iq, T - 1);
    const int need = min(NK - 1, qt * 32 + 32) / 32 + 1;   // key tiles with a key this query tile can see
    h16x8 qf[4];
    {
        const u16* qp = qkv16 + ((size_t)s * T + iqc) * 3 * D + hd * 64 + half * 8;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) qf[ks] = *reinterpret_cast<const h16x8*>(qp + ks * 16);
    }
    // (four named tiles and macro-unrolled loops: as an array indexed through nested unrolled loops hipcc keeps the score tiles
    // in scratch memory)
    f32x16 sc_0, sc_1, sc_2, sc_3;
#define A16_FOR_KT(X) X(0, sc_0) X(1, sc_1) X(2, sc_2) X(3, sc_3)
#define A16_SCORES(KT_, SC_)                                                                                              \
    if ((KT_) < need) {                                                                                                   \
        const u16* kp = kv_row(qkv16, ukv16, ld_ukv, kcol, s, T, D, hd, min((KT_) * 32 + col, NK - 1), 1) + half * 8;      \
        h16x8 kf0 = *reinterpret_cast<const h16x8*>(kp), kf1 = *reinterpret_cast<const h16x8*>(kp + 16);                  \
        h16x8 kf2 = *reinterpret_cast<const h16x8*>(kp + 32), kf3 = *reinterpret_cast<const h16x8*>(kp + 48);             \
        f32x16 a;                                                                                                         \
        _Pragma("unroll") for (int r = 0; r < 16; ++r) a[r] = 0.f;                                                        \
        a = mfma_h<F16>(kf0, qf[0], a); a = mfma_h<F16>(kf1, qf[1], a);                                                   \
        a = mfma_h<F16>(kf2, qf[2], a); a = mfma_h<F16>(kf3, qf[3], a);                                                   \
        SC_ = a;                                                                                                          \
    }
    A16_FOR_KT(A16_SCORES)
#undef A16_SCORES
    float m = -INFINITY;
#define A16_MASK(KT_, SC_)                                                                                                \
    if ((KT_) < need) {                                                                                                   \
        _Pragma("unroll") for (int r = 0; r < 16; ++r) {                                                                  \
            const int c = (KT_) * 32 + mrow(r, half);                                                                     \
            float w = -INFINITY;                                                                                          \
            if (c < NK) {                                                                                                 \
                const bool allowed = (c == 0) || (c - 1 <= iq);                                                           \
                w = (allowed ? SC_[r] * 0.125f : -1e4f) + addm_s[c];                                                      \
            }                                                                                                             \
            SC_[r] = w;                                                                                                   \
            m = fmaxf(m, w);                                                                                              \
        }                                                                                                                 \
    }
    A16_FOR_KT(A16_MASK)
#undef A16_MASK
    m = fmaxf(m, __shfl_xor(m, 32, 64));
    float sum = 0.f;
#define A16_EXP(KT_, SC_)                                                                                                 \
    if ((KT_) < need) {                                                                                                   \
        _Pragma("unroll") for (int r = 0; r < 16; ++r) {                                                                  \
            const float pe = __expf(SC_[r] - m);                                                                          \
            SC_[r] = pe;                                                                                                  \
            sum += pe;                                                                                                    \
        }                                                                                                                 \
    }
    A16_FOR_KT(A16_EXP)
#undef A16_EXP
    sum += __shfl_xor(sum, 32, 64);
    if (half == 0 && iq < T) lse[((size_t)s * T + iq) * H + hd] = m + logf(sum);
    // O^T = V^T P^T : A = VT[dim][permuted keys] (LDS), B = the probabilities of this lane's query (registers)
    const unsigned long long mrow4 = (((unsigned long long)s * H + hd) * T + iqc) * (dropout_key_pitch(NK) >> 2);   // mask row / 4
    f32x16 o0, o1;
#pragma unroll
    for (int r = 0; r < 16; ++r) { o0[r] = 0.f; o1[r] = 0.f; }
#define A16_PV_STEP(KT_, SC_, U_)                                                                                         \
    {                                                                                                                     \
        float pv[8];                                                                                                      \
        _Pragma("unroll") for (int q = 0; q < 2; ++q) {   /* registers 4 (2 U + q) .. + 3 = 4 consecutive keys */         \
            float mk[4];                                                                                                  \
            dropout_mask4(drop, mrow4 + (((KT_) * 32 + 8 * (2 * (U_) + q) + 4 * half) >> 2), mk);                          \
            _Pragma("unroll") for (int e = 0; e < 4; ++e) pv[4 * q + e] = SC_[8 * (U_) + 4 * q + e] * mk[e];              \
        }                                                                                                                 \
        const h16x8 pb = pack8<F16>(pv);                                                                                  \
        o0 = mfma_h<F16>(*reinterpret_cast<const h16x8*>(&VT[timg(col, (KT_) * 32 + (U_) * 16 + half * 8)]), pb, o0);      \
        o1 = mfma_h<F16>(*reinterpret_cast<const h16x8*>(&VT[timg(32 + col, (KT_) * 32 + (U_) * 16 + half * 8)]), pb, o1); \
    }
#define A16_PV(KT_, SC_)                                                                                                  \
    if ((KT_) < need) { A16_PV_STEP(KT_, SC_, 0) A16_PV_STEP(KT_, SC_, 1) }
    A16_FOR_KT(A16_PV)
#undef A16_PV
#undef A16_PV_STEP
#undef A16_FOR_KT
    if (iq < T) {
        const float inv = 1.0f / sum;
        u16* op = out16 + ((size_t)s * T + iq) * D + hd * 64 + 4 * half;
#pragma unroll
        for (int q4 = 0; q4 < 4; ++q4) {
            *reinterpret_cast<uint2*>(op + 8 * q4) =
                make_uint2(to16<F16>(o0[4 * q4] * inv) | (to16<F16>(o0[4 * q4 + 1] * inv) << 16),
                           to16<F16>(o0[4 * q4 + 2] * inv) | (to16<F16>(o0[4 * q4 + 3] * inv) << 16));
            *reinterpret_cast<uint2*>(op + 32 + 8 * q4) =
                make_uint2(to16<F16>(o1[4 * q4] * inv) | (to16<F16>(o1[4 * q4 + 1] * inv) << 16),
                           to16<F16>(o1[4 * q4 + 2] * inv) | (to16<F16>(o1[4 * q4 + 3] * inv) << 16));
        }
    }
}

// ---------------------------------------------------------------- backward (dQ, dK, dV in one workgroup per (sentence, head))
// d_att16 / att16 [S*T, D] 16 bit (d(attention output) and the forward's output), lse from the forward.  d_qkv16 [S*T, 3D]
// 16 bit receives dq | dk | dv of the token rows, d_ukv (fp32 [S, ld_ukv], columns kcol .. kcol + 2D of this layer) the
// gradient of the image key / value times ukv_scale (the inverse of the fp16 flow's internal loss scale).
template <bool F16>
__global__ __launch_bounds__(256, 3) void attn16_bwd_kernel(const u16* __restrict__ qkv16, const u16* __restrict__ ukv16, int ld_ukv, int kcol,
                                                         const float* __restrict__ am, const u16* __restrict__ d_att16,
                                                         const u16* __restrict__ att16, const float* __restrict__ lse,
                                                         u16* __restrict__ d_qkv16, float* __restrict__ d_ukv, int S, int H, int T,
                                                         const DropoutParams drop, float ukv_scale) {
    __shared__ __attribute__((aligned(16))) u16 QT_[64 * A16_TP];    // Q^T   [dim][permuted query]
    __shared__ __attribute__((aligned(16))) u16 GT_[64 * A16_TP];    // dO^T  [dim][permuted query]
    __shared__ __attribute__((aligned(16))) u16 KT_[64 * A16_TP];    // K^T   [dim][permuted key]
    __shared__ float delta_s[A16_MAXK];                              // rowsum(dO . O) per query
    __shared__ float lse_s[A16_MAXK];                                // log-sum-exp of every query's scores (forward pass)
    __shared__ float addm_s[A16_MAXK];                               // additive padding mask of every key
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int hd = blockIdx.x % H, s = blockIdx.x / H;
    const int NK = T + 1, D = H * 64, NKP = (NK + 31) & ~31, QTN = (T + 31) / 32, KTN = NKP / 32, TP32 = QTN * 32;
    const int col = lane & 31, half = lane >> 5;
    auto qrow = [&](int i) { return qkv16 + ((size_t)s * T + i) * 3 * D + hd * 64; };
    auto grow = [&](int i) { return d_att16 + ((size_t)s * T + i) * D + hd * 64; };
    stage_transposed(QT_, TP32, T, qrow);
    stage_transposed(GT_, TP32, T, grow);
    stage_transposed(KT_, NKP, NK, [&](int c) { return kv_row(qkv16, ukv16, ld_ukv, kcol, s, T, D, hd, c, 1); });
    for (int c = threadIdx.x; c < A16_MAXK; c += blockDim.x) {
        addm_s[c] = (c == 0 || c >= NK || !am) ? 0.f : (1.0f - am[(size_t)s * T + c - 1]) * -10000.0f;
        lse_s[c] = lse[((size_t)s * T + min(c, T - 1)) * H + hd];
    }
    // delta of query i: lanes (i, half 0 / 1) take 32 dims each
    for (int i0 = wave * 32; i0 < TP32; i0 += 128) {
        const int i = min(i0 + col, T - 1);
        const u16* gp = grow(i) + half * 32;
        const u16* op = att16 + ((size_t)s * T + i) * D + hd * 64 + half * 32;
        float a = 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const h16x8 gv = *reinterpret_cast<const h16x8*>(gp + 8 * j), ov = *reinterpret_cast<const h16x8*>(op + 8 * j);
#pragma unroll
            for (int e = 0; e < 8; ++e) a += from16<F16>((u16)gv[e]) * from16<F16>((u16)ov[e]);
        }
        a += __shfl_xor(a, 32, 64);
        if (half == 0) delta_s[i0 + col] = a;
    }
    __syncthreads();

    // Roles: the QTN dQ tiles and the KTN dK / dV tiles are dealt out over the four waves round robin in the order
    // dQ_0 .. dQ_{QTN-1}, dKV_0 .. dKV_{KTN-1} (role r -> wave r % 4); 64 tokens: dQ_0, dQ_1, dKV_0, dKV_1 on waves 0-3, dKV_2
    // (the last key alone) on wave 0 again.
    // ---- dQ of a query tile: transposed tiles (a lane = one query)
    for (int qt = wave; qt < QTN; qt += 4) {
        const int iq = qt * 32 + col, iqc = min(iq, T - 1);
        const int need = min(NK - 1, qt * 32 + 32) / 32 + 1;
        h16x8 qf[4], gf[4];
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            qf[ks] = *reinterpret_cast<const h16x8*>(qrow(iqc) + half * 8 + ks * 16);
            gf[ks] = *reinterpret_cast<const h16x8*>(grow(iqc) + half * 8 + ks * 16);
        }
        const float lse_q = lse_s[iqc], delta_q = delta_s[qt * 32 + col];
        const unsigned long long mrow4 = (((unsigned long long)s * H + hd) * T + iqc) * (dropout_key_pitch(NK) >> 2);
        f32x16 dq[2];
#pragma unroll
        for (int r = 0; r < 16; ++r) { dq[0][r] = 0.f; dq[1][r] = 0.f; }
        for (int kt = 0; kt < need; ++kt) {
            const u16* kp = kv_row(qkv16, ukv16, ld_ukv, kcol, s, T, D, hd, min(kt * 32 + col, NK - 1), 1) + half * 8;
            const u16* vp = kv_row(qkv16, ukv16, ld_ukv, kcol, s, T, D, hd, min(kt * 32 + col, NK - 1), 2) + half * 8;
            h16x8 kf[4], vf[4];
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                kf[ks] = *reinterpret_cast<const h16x8*>(kp + ks * 16);
                vf[ks] = *reinterpret_cast<const h16x8*>(vp + ks * 16);
            }
            f32x16 aS, aP;
#pragma unroll
            for (int r = 0; r < 16; ++r) { aS[r] = 0.f; aP[r] = 0.f; }
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                aS = mfma_h<F16>(kf[ks], qf[ks], aS);
                aP = mfma_h<F16>(vf[ks], gf[ks], aP);
            }
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                float dsv[8];
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    float mk4[4];
                    dropout_mask4(drop, mrow4 + ((kt * 32 + 8 * (2 * u + q) + 4 * half) >> 2), mk4);
#pragma unroll
                    for (int e4 = 0; e4 < 4; ++e4) {
                        const int e = 4 * q + e4, r = 8 * u + e, c = kt * 32 + mrow(r, half);
                        float x = 0.f;
                        if (c < NK) {
                            const bool allowed = (c == 0) || (c - 1 <= iq);
                            const float pr = __expf((allowed ? aS[r] * 0.125f : -1e4f) + addm_s[c] - lse_q);
                            x = allowed ? pr * (aP[r] * mk4[e4] - delta_q) * 0.125f : 0.f;
                        }
                        dsv[e] = x;
                    }
                }
                const h16x8 db8 = pack8<F16>(dsv);
#pragma unroll
                for (int db = 0; db < 2; ++db) {
                    const h16x8 ka = *reinterpret_cast<const h16x8*>(&KT_[timg(db * 32 + col, kt * 32 + u * 16 + half * 8)]);
                    dq[db] = mfma_h<F16>(ka, db8, dq[db]);
                }
            }
        }
        if (iq < T) {
            u16* op = d_qkv16 + ((size_t)s * T + iq) * 3 * D + hd * 64 + 4 * half;
#pragma unroll
            for (int db = 0; db < 2; ++db)
#pragma unroll
                for (int q4 = 0; q4 < 4; ++q4) {
                    const unsigned lo = to16<F16>(dq[db][4 * q4]) | (to16<F16>(dq[db][4 * q4 + 1]) << 16);
                    const unsigned hi = to16<F16>(dq[db][4 * q4 + 2]) | (to16<F16>(dq[db][4 * q4 + 3]) << 16);
                    *reinterpret_cast<uint2*>(op + db * 32 + 8 * q4) = make_uint2(lo, hi);
                }
        }
    }

    // ---- dK, dV of a key tile: untransposed tiles (a lane = one key)
    for (int kt = ((wave - QTN) % 4 + 4) % 4; kt < KTN; kt += 4) {
        const int c = kt * 32 + col, cc = min(c, NK - 1);
        const bool cvalid = c < NK;
        const u16* kp = kv_row(qkv16, ukv16, ld_ukv, kcol, s, T, D, hd, cc, 1) + half * 8;
        const u16* vp = kv_row(qkv16, ukv16, ld_ukv, kcol, s, T, D, hd, cc, 2) + half * 8;
        const float addm = addm_s[cc];
        const int kpitch = dropout_key_pitch(NK);
        f32x16 dk[2], dv[2];
#pragma unroll
        for (int r = 0; r < 16; ++r) { dk[0][r] = 0.f; dk[1][r] = 0.f; dv[0][r] = 0.f; dv[1][r] = 0.f; }
        const int qt0 = kt == 0 ? 0 : kt - 1;   // first query tile with a query i >= (first key of the tile) - 1
        for (int qt = qt0; qt < QTN; ++qt) {
            const int ia = min(qt * 32 + col, T - 1);
            // (the K / V fragments of this lane's key are re-read per query tile - L2 hits - instead of living in 32 registers
            // across the loop: 184 -> <= 168 registers = three waves per SIMD, what the LDS footprint admits anyway)
            f32x16 aS, aP;
#pragma unroll
            for (int r = 0; r < 16; ++r) { aS[r] = 0.f; aP[r] = 0.f; }
            {
                h16x8 qf[4], kf[4];
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) {
                    qf[ks] = *reinterpret_cast<const h16x8*>(qrow(ia) + half * 8 + ks * 16);
                    kf[ks] = *reinterpret_cast<const h16x8*>(kp + ks * 16);
                }
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) aS = mfma_h<F16>(qf[ks], kf[ks], aS);
            }
            {
                h16x8 gf[4], vf[4];
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) {
                    gf[ks] = *reinterpret_cast<const h16x8*>(grow(ia) + half * 8 + ks * 16);
                    vf[ks] = *reinterpret_cast<const h16x8*>(vp + ks * 16);
                }
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) aP = mfma_h<F16>(gf[ks], vf[ks], aP);
            }
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                float pm[8], dsv[8];
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    // registers 4 (2u + q) .. + 3 = queries i0 .. i0 + 3; the 4 lanes of a quad = keys c & ~3 .. + 3: lane j of the
                    // quad generates the masks of query i0 + j for the quad's 4 keys, a 4 x 4 exchange hands every lane the
                    // mask of ITS key for the 4 queries (one generator call per lane instead of four)
                    const int i0 = qt * 32 + 8 * (2 * u + q) + 4 * half;
                    float mk4[4];
                    dropout_mask4(drop, (((unsigned long long)s * H + hd) * T + min(i0 + (lane & 3), T - 1)) * (kpitch >> 2) + (cc >> 2), mk4);
                    if (drop.p > 0.f) quad_transpose4(mk4[0], mk4[1], mk4[2], mk4[3], lane & 1, lane & 2);
#pragma unroll
                    for (int e4 = 0; e4 < 4; ++e4) {
                        const int e = 4 * q + e4, r = 8 * u + e, i = i0 + e4;
                        float pr = 0.f, x = 0.f;
                        if (i < T && cvalid) {
                            const float lse_i = lse_s[i], delta_i = delta_s[i];
                            const bool allowed = (c == 0) || (c - 1 <= i);
                            pr = __expf((allowed ? aS[r] * 0.125f : -1e4f) + addm - lse_i);
                            x = allowed ? pr * (aP[r] * mk4[e4] - delta_i) * 0.125f : 0.f;
                            pr *= mk4[e4];
                        }
                        pm[e] = pr;
                        dsv[e] = x;
                    }
                }
                const h16x8 pb = pack8<F16>(pm), sb = pack8<F16>(dsv);
#pragma unroll
                for (int db = 0; db < 2; ++db) {
                    const int off = timg(db * 32 + col, qt * 32 + u * 16 + half * 8);
                    dv[db] = mfma_h<F16>(*reinterpret_cast<const h16x8*>(&GT_[off]), pb, dv[db]);
                    dk[db] = mfma_h<F16>(*reinterpret_cast<const h16x8*>(&QT_[off]), sb, dk[db]);
                }
            }
        }
        if (cvalid) {
            if (c == 0) {
                float* kd = d_ukv + (size_t)s * ld_ukv + kcol + hd * 64;
#pragma unroll
                for (int db = 0; db < 2; ++db)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        kd[db * 32 + mrow(r, half)] = dk[db][r] * ukv_scale;
                        kd[D + db * 32 + mrow(r, half)] = dv[db][r] * ukv_scale;
                    }
            } else {
                u16* kd = d_qkv16 + ((size_t)s * T + c - 1) * 3 * D + D + hd * 64 + 4 * half;
#pragma unroll
                for (int db = 0; db < 2; ++db)
#pragma unroll
                    for (int q4 = 0; q4 < 4; ++q4) {
                        *reinterpret_cast<uint2*>(kd + db * 32 + 8 * q4) =
                            make_uint2(to16<F16>(dk[db][4 * q4]) | (to16<F16>(dk[db][4 * q4 + 1]) << 16),
                                       to16<F16>(dk[db][4 * q4 + 2]) | (to16<F16>(dk[db][4 * q4 + 3]) << 16));
                        *reinterpret_cast<uint2*>(kd + D + db * 32 + 8 * q4) =
                            make_uint2(to16<F16>(dv[db][4 * q4]) | (to16<F16>(dv[db][4 * q4 + 1]) << 16),
                                       to16<F16>(dv[db][4 * q4 + 2]) | (to16<F16>(dv[db][4 * q4 + 3]) << 16));
                    }
            }
        }
    }
}

// ---------------------------------------------------------------- launchers (decoder.hip)
bool attn16_supported(int T) { return T >= 1 && T + 1 <= A16_MAXK; }

int launch_attn16_forward(const unsigned short* qkv16, const unsigned short* ukv16, int ld_ukv, int kcol, const float* am,
                          unsigned short* out16, float* lse, int S, int H, int T, DropoutParams drop, int f16, hipStream_t st) {
    RGRG_CHECK_ARG(attn16_supported(T) && qkv16 && ukv16 && out16 && lse);
    // a wave per 32-query tile: <= 64 tokens need two waves (a 256-thread workgroup would keep two idle waves resident)
    const dim3 block(T <= 64 ? 128 : 256);
    if (f16) hipLaunchKernelGGL(attn16_fwd_kernel<true>, dim3(S * H), block, 0, st, qkv16, ukv16, ld_ukv, kcol, am, out16, lse, S, H, T, drop);
    else hipLaunchKernelGGL(attn16_fwd_kernel<false>, dim3(S * H), block, 0, st, qkv16, ukv16, ld_ukv, kcol, am, out16, lse, S, H, T, drop);
    RGRG_LAUNCH_CHECK();
    return RGRG_OK;
}

int launch_attn16_backward(const unsigned short* qkv16, const unsigned short* ukv16, int ld_ukv, int kcol, const float* am,
                           const unsigned short* d_att16, const unsigned short* att16, const float* lse, unsigned short* d_qkv16,
                           float* d_ukv, int S, int H, int T, DropoutParams drop, float ukv_scale, int f16, hipStream_t st) {
    RGRG_CHECK_ARG(attn16_supported(T) && qkv16 && ukv16 && d_att16 && att16 && lse && d_qkv16 && d_ukv);
    if (f16)
        hipLaunchKernelGGL(attn16_bwd_kernel<true>, dim3(S * H), dim3(256), 0, st, qkv16, ukv16, ld_ukv, kcol, am, d_att16, att16, lse,
                           d_qkv16, d_ukv, S, H, T, drop, ukv_scale);
    else
        hipLaunchKernelGGL(attn16_bwd_kernel<false>, dim3(S * H), dim3(256), 0, st, qkv16, ukv16, ld_ukv, kcol, am, d_att16, att16, lse,
                           d_qkv16, d_ukv, S, H, T, drop, ukv_scale);
    RGRG_LAUNCH_CHECK();
    return RGRG_OK;
}

}  // namespace rgrg
