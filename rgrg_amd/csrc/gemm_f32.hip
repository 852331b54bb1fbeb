// fp32 GEMM / implicit-GEMM convolution on the gfx950 f32 matrix core
// (v_mfma_f32_32x32x2_f32: exact fp32 FMA chains at the 157 TF/s vector rate).
//
//   Y[m,n] = act( (sum_k A(m,k) W[n,k]) * scale[n] + shift[n] + R[m,n] )
//
// A is either a dense [M,K] matrix or the im2col view of an NHWC tensor; W is
// [N,K] with K contiguous (nn.Linear layout / [Cout][KH][KW][Cin]).  Both operands
// are K-contiguous, so a 32-deep K tile of either is 8 float4 per row and is staged
// global -> registers -> LDS with rows padded to 36 floats: the ds_read_b128 of the
// MFMA fragments (lane = row, 16 B of K) then touches 16 distinct 16-B bank slots
// per 16-lane group - conflict free (MI355X_MICROARCH.md, LDS table).
//
// Fragment trick: one ds_read_b128 per lane feeds FOUR MFMAs.  v_mfma_f32_32x32x2
// takes A[i=lane&31][k=lane>>5]; we let half h=lane>>5 own k = 8*kk + 4*h + j for
// j = 0..3, i.e. MFMA j contracts the k pair {8kk+j, 8kk+4+j}.  A and B use the same
// permutation of K, so the sum over the tile is unchanged.
#include "common.h"

namespace rgrg {

struct GemmParams {
    const float* A;
    const float* W;
    const float* scale;
    const float* shift;
    const float* R;
    float* Y;
    float* ws;
    int M, N, K, lda, ldy, act, splitk;
    int conv, H, Wd, Cin, KH, KW, stride, pad, OH, OW;
};

constexpr int LDT = 36;  // padded LDS row (floats) of a 32-deep K tile

template <int BM, int BN>
__global__ __launch_bounds__(256) void gemm_f32_kernel(const GemmParams p) {
    constexpr int MI = BM / 64, NI = BN / 64;  // 32x32 MFMA tiles per wave (2x2 waves)
    constexpr int AL = BM / 32, BL = BN / 32;  // float4 loads per thread per K tile
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* As = smem;                 // [2][BM][LDT]
    float* Bs = smem + 2 * BM * LDT;  // [2][BN][LDT]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
    const int kchunk = tid & 7, lrow = tid >> 3;
    const int Kper = p.K / p.splitk;
    const int kbeg = blockIdx.z * Kper;
    const int nk = Kper / 32;

    // per-thread row descriptors (fixed over the K loop)
    const float* a_ptr[AL];
    int a_ih0[AL], a_iw0[AL];
    bool a_ok[AL];
#pragma unroll
    for (int i = 0; i < AL; ++i) {
        const int m = m0 + lrow + 32 * i;
        a_ok[i] = m < p.M;
        a_ih0[i] = a_iw0[i] = 0;
        if (p.conv) {
            const int mm = a_ok[i] ? m : 0;
            const int ow = mm % p.OW;
            const int t = mm / p.OW;
            const int oh = t % p.OH;
            const int b = t / p.OH;
            a_ih0[i] = oh * p.stride - p.pad;
            a_iw0[i] = ow * p.stride - p.pad;
            a_ptr[i] = p.A + (size_t)b * p.H * p.Wd * p.Cin;
        } else {
            a_ptr[i] = p.A + (size_t)(a_ok[i] ? m : 0) * p.lda;
        }
    }
    const float* b_ptr[BL];
    bool b_ok[BL];
#pragma unroll
    for (int i = 0; i < BL; ++i) {
        const int n = n0 + lrow + 32 * i;
        b_ok[i] = n < p.N;
        b_ptr[i] = p.W + (size_t)(b_ok[i] ? n : 0) * p.K;
    }

    f32x4 ra[AL], rb[BL];
    auto load_tile = [&](int kt) {
        const int k0 = kbeg + kt * 32;
        int kh = 0, kw = 0, c0 = k0;
        if (p.conv) {
            const int tap = k0 / p.Cin;
            c0 = k0 - tap * p.Cin;
            kh = tap / p.KW;
            kw = tap - kh * p.KW;
        }
#pragma unroll
        for (int i = 0; i < AL; ++i) {
            bool ok = a_ok[i];
            const float* src;
            if (p.conv) {
                const int ih = a_ih0[i] + kh, iw = a_iw0[i] + kw;
                ok = ok && (unsigned)ih < (unsigned)p.H && (unsigned)iw < (unsigned)p.Wd;
                src = a_ptr[i] + ((size_t)(ok ? ih : 0) * p.Wd + (ok ? iw : 0)) * p.Cin + c0 + kchunk * 4;
            } else {
                src = a_ptr[i] + k0 + kchunk * 4;
            }
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (ok) v = *reinterpret_cast<const f32x4*>(src);
            ra[i] = v;
        }
#pragma unroll
        for (int i = 0; i < BL; ++i) {
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (b_ok[i]) v = *reinterpret_cast<const f32x4*>(b_ptr[i] + k0 + kchunk * 4);
            rb[i] = v;
        }
    };
    auto store_tile = [&](int buf) {
#pragma unroll
        for (int i = 0; i < AL; ++i)
            *reinterpret_cast<f32x4*>(&As[(buf * BM + lrow + 32 * i) * LDT + kchunk * 4]) = ra[i];
#pragma unroll
        for (int i = 0; i < BL; ++i)
            *reinterpret_cast<f32x4*>(&Bs[(buf * BN + lrow + 32 * i) * LDT + kchunk * 4]) = rb[i];
    };

    f32x16 acc[MI][NI];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;

    load_tile(0);
    store_tile(0);
    __syncthreads();

    const int frow = lane & 31, fk = (lane >> 5) * 4;
    for (int kt = 0; kt < nk; ++kt) {
        const int buf = kt & 1;
        if (kt + 1 < nk) load_tile(kt + 1);
        const float* Ab = &As[(buf * BM + wm * (BM / 2) + frow) * LDT + fk];
        const float* Bb = &Bs[(buf * BN + wn * (BN / 2) + frow) * LDT + fk];
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            f32x4 a[MI], b[NI];
#pragma unroll
            for (int mi = 0; mi < MI; ++mi) a[mi] = *reinterpret_cast<const f32x4*>(Ab + mi * 32 * LDT + kk * 8);
#pragma unroll
            for (int ni = 0; ni < NI; ++ni) b[ni] = *reinterpret_cast<const f32x4*>(Bb + ni * 32 * LDT + kk * 8);
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                    for (int ni = 0; ni < NI; ++ni)
                        acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[mi][j], b[ni][j], acc[mi][ni], 0, 0, 0);
        }
        if (kt + 1 < nk) store_tile(buf ^ 1);
        __syncthreads();
    }

    // epilogue.  C/D map of the 32x32 MFMA: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
    const int ccol = lane & 31, crow4 = 4 * (lane >> 5);
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) {
            const int col = n0 + wn * (BN / 2) + ni * 32 + ccol;
            if (col >= p.N) continue;
            float sc = 1.f, sh = 0.f;
            if (p.splitk == 1) {
                if (p.scale) sc = p.scale[col];
                if (p.shift) sh = p.shift[col];
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = m0 + wm * (BM / 2) + mi * 32 + (r & 3) + 8 * (r >> 2) + crow4;
                if (row >= p.M) continue;
                float v = acc[mi][ni][r];
                if (p.splitk > 1) {
                    p.ws[((size_t)blockIdx.z * p.M + row) * p.N + col] = v;
                } else {
                    v = v * sc + sh;
                    if (p.R) v += p.R[(size_t)row * p.ldy + col];
                    p.Y[(size_t)row * p.ldy + col] = apply_act(v, p.act);
                }
            }
        }
}

// Y = epilogue(sum_z ws[z]) for split-K launches (fixed summation order -> deterministic).
__global__ __launch_bounds__(256) void splitk_epilogue_kernel(const GemmParams p) {
    const size_t total = (size_t)p.M * p.N;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int row = (int)(i / p.N), col = (int)(i % p.N);
        float v = 0.f;
        for (int z = 0; z < p.splitk; ++z) v += p.ws[(size_t)z * total + i];
        const float sc = p.scale ? p.scale[col] : 1.f, sh = p.shift ? p.shift[col] : 0.f;
        v = v * sc + sh;
        if (p.R) v += p.R[(size_t)row * p.ldy + col];
        p.Y[(size_t)row * p.ldy + col] = apply_act(v, p.act);
    }
}

template <int BM, int BN>
static int set_attr() {
    constexpr size_t lds = (size_t)(2 * BM + 2 * BN) * LDT * sizeof(float);
    static bool attr_set = false;
    if (!attr_set) {
        RGRG_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_f32_kernel<BM, BN>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr_set = true;
    }
    return RGRG_OK;
}

// must run once outside any stream capture (hipFuncSetAttribute is not capturable)
int init_gemm_attrs() {
    int rc = set_attr<128, 128>();
    if (rc) return rc;
    return set_attr<64, 64>();
}

template <int BM, int BN>
static int launch_cfg(const GemmParams& p, hipStream_t st) {
    constexpr size_t lds = (size_t)(2 * BM + 2 * BN) * LDT * sizeof(float);
    int rc0 = set_attr<BM, BN>();
    if (rc0) return rc0;
    dim3 grid((p.N + BN - 1) / BN, (p.M + BM - 1) / BM, p.splitk);
    hipLaunchKernelGGL((gemm_f32_kernel<BM, BN>), grid, dim3(256), lds, st, p);
    RGRG_LAUNCH_CHECK();
    return RGRG_OK;
}

int launch_gemm(GemmParams p, hipStream_t st) {
    RGRG_CHECK_ARG(p.M > 0 && p.N > 0 && p.K > 0 && p.K % 32 == 0);
    RGRG_CHECK_ARG(p.act >= 0 && p.act <= 2);
    if (p.splitk <= 0) p.splitk = 1;
    RGRG_CHECK_ARG(p.K % (32 * p.splitk) == 0);
    RGRG_CHECK_ARG(p.splitk == 1 || p.ws != nullptr);
    const long tiles_big = (long)((p.M + 127) / 128) * ((p.N + 127) / 128) * p.splitk;
    int rc;
    if (tiles_big >= 192 && p.M >= 128 && p.N >= 128)
        rc = launch_cfg<128, 128>(p, st);
    else
        rc = launch_cfg<64, 64>(p, st);
    if (rc != RGRG_OK) return rc;
    if (p.splitk > 1) {
        const size_t total = (size_t)p.M * p.N;
        const int blocks = (int)((total + 255) / 256 < 2048 ? (total + 255) / 256 : 2048);
        hipLaunchKernelGGL(splitk_epilogue_kernel, dim3(blocks), dim3(256), 0, st, p);
        RGRG_LAUNCH_CHECK();
    }
    return RGRG_OK;
}

// Dense GEMM for the decoder's many-sequence path.  `ws` (>= ws_floats floats) lets the K loop be
// split over workgroups when the output has too few tiles to fill the 256 CUs.
int launch_gemm_dense(const float* A, const float* W, const float* shift, const float* R, float* Y, int M, int N, int K,
                      int ldy, int act, float* ws, size_t ws_floats, hipStream_t st) {
    GemmParams p{};
    p.A = A; p.W = W; p.shift = shift; p.R = R; p.Y = Y; p.ws = ws;
    p.M = M; p.N = N; p.K = K; p.lda = K; p.ldy = ldy; p.act = act; p.splitk = 1;
    // 64x64 tiles + enough K splits to fill the CUs (measured better than 128x128 tiles for the decoder's shapes:
    // 13.6 vs 13.2 images/s at batch 8, 18.7 vs 17.5 at batch 32)
    const long tiles = (long)((M + 63) / 64) * ((N + 63) / 64);
    if (ws && tiles < 256) {
        int sk = 1;
        while (sk * 2 <= 16 && tiles * sk < 256 && K % (32 * sk * 2) == 0 && K / (32 * sk * 2) >= 4 &&
               (size_t)(sk * 2) * M * N <= ws_floats)
            sk *= 2;
        p.splitk = sk;
    }
    return launch_gemm(p, st);
}

}  // namespace rgrg

using namespace rgrg;

extern "C" int rgrg_linear_f32(const float* A, const float* W, const float* scale, const float* shift, const float* R,
                               float* Y, int M, int N, int K, int ldy, int act, int splitk, float* ws, void* stream) {
    RGRG_CHECK_ARG(A && W && Y && ldy >= N);
    GemmParams p{};
    p.A = A; p.W = W; p.scale = scale; p.shift = shift; p.R = R; p.Y = Y; p.ws = ws;
    p.M = M; p.N = N; p.K = K; p.lda = K; p.ldy = ldy; p.act = act; p.splitk = splitk;
    p.conv = 0;
    return launch_gemm(p, as_stream(stream));
}

extern "C" int rgrg_conv2d_nhwc_f32(const float* X, const float* W, const float* scale, const float* shift,
                                    const float* R, float* Y, int B, int H, int Wd, int Cin, int Cout, int KH, int KW,
                                    int stride, int pad, int act, int splitk, float* ws, void* stream) {
    RGRG_CHECK_ARG(X && W && Y && B > 0 && H > 0 && Wd > 0 && Cin % 32 == 0 && stride > 0);
    GemmParams p{};
    p.A = X; p.W = W; p.scale = scale; p.shift = shift; p.R = R; p.Y = Y; p.ws = ws;
    p.OH = (H + 2 * pad - KH) / stride + 1;
    p.OW = (Wd + 2 * pad - KW) / stride + 1;
    p.M = B * p.OH * p.OW; p.N = Cout; p.K = KH * KW * Cin; p.lda = 0; p.ldy = Cout; p.act = act; p.splitk = splitk;
    p.conv = 1; p.H = H; p.Wd = Wd; p.Cin = Cin; p.KH = KH; p.KW = KW; p.stride = stride; p.pad = pad;
    return launch_gemm(p, as_stream(stream));
}
