// bf16-weight GEMM on the gfx950 bf16 matrix core (v_mfma_f32_32x32x16_bf16, fp32 accumulate) for the
// decoder when MANY sequences decode together (BASELINE configs[2]: batch 32 -> 928 sequences): the
// fp32 path is then bound by the exact-fp32 MFMA (1/16 of the bf16 rate).
//
//   Y[m,n] = act( sum_k bf16(A[m,k]) * Wb[n,k] + shift[n] + R[m,n] )        (fp32 in, fp32 out)
//
// A stays fp32 in HBM (LayerNorm / residual stream are fp32) and is rounded to bf16 (RNE) while it is
// staged to LDS; Wb is the bf16 copy of the [N,K] weight made once at load time.  Both LDS tiles are
// [rows][64 bf16] with rows padded to 144 B, so the ds_read_b128 of an MFMA fragment (lane = row, 8
// consecutive k) is conflict free.  A and B fragments use the same (lane half, element) -> k mapping, so
// the K order inside a tile is irrelevant.  Not bit-exact with the fp32 reference by construction: this
// path is opt-in (torch.autocast) and its tests are tolerance-based.
#include "common.h"

namespace rgrg {

typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned short u16;

struct GemmBf16Params {
    const float* A;
    const u16* Wb;
    const float* shift;
    const float* R;
    float* Y;
    int M, N, K, ldy, act;
};

constexpr int BK16 = 64;                 // k per tile
constexpr int LDB = (BK16 * 2 + 16) / 2;  // padded LDS row in bf16 elements (144 B)

__device__ __forceinline__ u16 f32_to_bf16_rne(float f) {
    unsigned u = __float_as_uint(f);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (u16)(u >> 16);
}

template <int BM, int BN>
__global__ __launch_bounds__(256) void gemm_bf16w_kernel(const GemmBf16Params p) {
    constexpr int MI = BM / 64, NI = BN / 64;  // 32x32 MFMA tiles per wave (2x2 waves)
    constexpr int AL = BM / 32, BL = BN / 32;  // 8-element chunks per thread per tile
    extern __shared__ __attribute__((aligned(16))) u16 smem16[];
    u16* As = smem16;                  // [2][BM][LDB]
    u16* Bs = smem16 + 2 * BM * LDB;   // [2][BN][LDB]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
    const int kchunk = tid & 7, lrow = tid >> 3;  // 8 chunks of 8 k per row, 32 rows per pass
    const int nk = p.K / BK16;

    f32x4 ra[AL][2];
    bf16x8 rb[BL];
    auto load_tile = [&](int kt) {
        const int k0 = kt * BK16 + kchunk * 8;
#pragma unroll
        for (int i = 0; i < AL; ++i) {
            const int m = m0 + lrow + 32 * i;
            f32x4 z = {0.f, 0.f, 0.f, 0.f};
            ra[i][0] = z; ra[i][1] = z;
            if (m < p.M) {
                const float* src = p.A + (size_t)m * p.K + k0;
                ra[i][0] = *reinterpret_cast<const f32x4*>(src);
                ra[i][1] = *reinterpret_cast<const f32x4*>(src + 4);
            }
        }
#pragma unroll
        for (int i = 0; i < BL; ++i) {
            const int n = n0 + lrow + 32 * i;
            bf16x8 z = {0, 0, 0, 0, 0, 0, 0, 0};
            rb[i] = z;
            if (n < p.N) rb[i] = *reinterpret_cast<const bf16x8*>(p.Wb + (size_t)n * p.K + k0);
        }
    };
    auto store_tile = [&](int buf) {
#pragma unroll
        for (int i = 0; i < AL; ++i) {
            bf16x8 v;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                v[e] = (short)f32_to_bf16_rne(ra[i][0][e]);
                v[4 + e] = (short)f32_to_bf16_rne(ra[i][1][e]);
            }
            *reinterpret_cast<bf16x8*>(&As[(buf * BM + lrow + 32 * i) * LDB + kchunk * 8]) = v;
        }
#pragma unroll
        for (int i = 0; i < BL; ++i)
            *reinterpret_cast<bf16x8*>(&Bs[(buf * BN + lrow + 32 * i) * LDB + kchunk * 8]) = rb[i];
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
    const int frow = lane & 31, fk = (lane >> 5) * 8;
    for (int kt = 0; kt < nk; ++kt) {
        const int buf = kt & 1;
        if (kt + 1 < nk) load_tile(kt + 1);
        const u16* Ab = &As[(buf * BM + wm * (BM / 2) + frow) * LDB + fk];
        const u16* Bb = &Bs[(buf * BN + wn * (BN / 2) + frow) * LDB + fk];
#pragma unroll
        for (int ks = 0; ks < BK16 / 16; ++ks) {
            bf16x8 a[MI], b[NI];
#pragma unroll
            for (int mi = 0; mi < MI; ++mi) a[mi] = *reinterpret_cast<const bf16x8*>(Ab + mi * 32 * LDB + ks * 16);
#pragma unroll
            for (int ni = 0; ni < NI; ++ni) b[ni] = *reinterpret_cast<const bf16x8*>(Bb + ni * 32 * LDB + ks * 16);
#pragma unroll
            for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                for (int ni = 0; ni < NI; ++ni)
                    acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[mi], b[ni], acc[mi][ni], 0, 0, 0);
        }
        if (kt + 1 < nk) store_tile(buf ^ 1);
        __syncthreads();
    }
    // epilogue (C/D map of the 32x32 MFMA: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5))
    const int ccol = lane & 31, crow4 = 4 * (lane >> 5);
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) {
            const int col = n0 + wn * (BN / 2) + ni * 32 + ccol;
            if (col >= p.N) continue;
            const float sh = p.shift ? p.shift[col] : 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = m0 + wm * (BM / 2) + mi * 32 + (r & 3) + 8 * (r >> 2) + crow4;
                if (row >= p.M) continue;
                float v = acc[mi][ni][r] + sh;
                if (p.R) v += p.R[(size_t)row * p.ldy + col];
                p.Y[(size_t)row * p.ldy + col] = apply_act(v, p.act);
            }
        }
}

__global__ __launch_bounds__(256) void f32_to_bf16_kernel(const float* __restrict__ src, u16* __restrict__ dst, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        dst[i] = f32_to_bf16_rne(src[i]);
}

template <int BM, int BN>
static int launch_bf16_cfg(const GemmBf16Params& p, hipStream_t st) {
    constexpr size_t lds = (size_t)(2 * BM + 2 * BN) * LDB * sizeof(u16);
    dim3 grid((p.N + BN - 1) / BN, (p.M + BM - 1) / BM);
    hipLaunchKernelGGL((gemm_bf16w_kernel<BM, BN>), grid, dim3(256), lds, st, p);
    RGRG_LAUNCH_CHECK();
    return RGRG_OK;
}

int init_gemm_bf16_attrs() {
    RGRG_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_bf16w_kernel<128, 128>),
                                 hipFuncAttributeMaxDynamicSharedMemorySize, (int)((size_t)512 * LDB * sizeof(u16))));
    RGRG_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_bf16w_kernel<64, 64>),
                                 hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
    return RGRG_OK;
}

int launch_gemm_bf16w(const float* A, const void* Wb, const float* shift, const float* R, float* Y, int M, int N, int K,
                      int ldy, int act, hipStream_t st) {
    RGRG_CHECK_ARG(A && Wb && Y && M > 0 && N > 0 && K > 0 && K % BK16 == 0 && ldy >= N);
    GemmBf16Params p{A, reinterpret_cast<const u16*>(Wb), shift, R, Y, M, N, K, ldy, act};
    const long tiles_big = (long)((M + 127) / 128) * ((N + 127) / 128);
    if (tiles_big >= 192) return launch_bf16_cfg<128, 128>(p, st);
    return launch_bf16_cfg<64, 64>(p, st);
}

int convert_f32_to_bf16(const float* src, void* dst, size_t n, hipStream_t st) {
    const int blocks = (int)((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096);
    hipLaunchKernelGGL(f32_to_bf16_kernel, dim3(blocks), dim3(256), 0, st, src, reinterpret_cast<u16*>(dst), n);
    RGRG_LAUNCH_CHECK();
    return RGRG_OK;
}

}  // namespace rgrg

using namespace rgrg;

extern "C" int rgrg_f32_to_bf16(const float* src, uint16_t* dst, int64_t n, void* stream) {
    RGRG_CHECK_ARG(src && dst && n > 0);
    return convert_f32_to_bf16(src, dst, (size_t)n, as_stream(stream));
}

extern "C" int rgrg_linear_bf16w_f32(const float* A, const uint16_t* Wb, const float* shift, const float* R, float* Y,
                                     int M, int N, int K, int ldy, int act, void* stream) {
    int rc = init_gemm_bf16_attrs();
    if (rc) return rc;
    return launch_gemm_bf16w(A, Wb, shift, R, Y, M, N, K, ldy, act, as_stream(stream));
}
