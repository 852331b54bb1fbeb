// bf16-weight GEMM on the gfx950 bf16 matrix core (v_mfma_f32_32x32x16_bf16, fp32 accumulate) for the
// decoder when MANY sequences decode together (BASELINE configs[2]: batch 32 -> 928 sequences): the
// fp32 path is then bound by the exact-fp32 MFMA (1/16 of the bf16 rate).
//
//   Y[m,n] = act( sum_k bf16(A[m,k]) * Wb[n,k] + shift[n] + R[m,n] )        (fp32 in, fp32 out)
//
// A stays fp32 in HBM (LayerNorm / residual stream are fp32; adjacent lanes read adjacent 16-byte pieces, so a
// wave's load touches whole cache lines) and is rounded to bf16 (RNE) while it is staged from registers to LDS; Wb is the bf16 copy of the [N,K] weight made once at load time.  Both LDS tiles are
// [rows][64 bf16] with rows padded to 144 B, so the ds_read_b128 of an MFMA fragment (lane = row, 8
// consecutive k) is conflict free.  A and B fragments use the same (lane half, element) -> k mapping, so
// the K order inside a tile is irrelevant.  Not bit-exact with the fp32 reference by construction: this
// path is opt-in (torch.autocast) and its tests are tolerance-based.
#include "common.h"

namespace rgrg {

typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef short bf16x4 __attribute__((ext_vector_type(4)));
typedef unsigned short u16;

struct GemmBf16Params {
    const float* A;    // fp32 activations (rounded to bf16 while staged) ...
    const u16* A16;    // ... or activations already rounded to bf16 by their producer (A16 kernels)
    const u16* Wb;
    const float* shift;
    const float* R;
    float* Y;
    u16* Y16;          // non-null: the result is stored as bf16 [M, ldy] instead of fp32 (feeds the next GEMM only)
    int M, N, K, ldy, act;
};

constexpr int BK16 = 64;                 // k per tile
constexpr int LDB = (BK16 * 2 + 16) / 2;  // padded LDS row in bf16 elements (144 B)

__device__ __forceinline__ u16 f32_to_bf16_rne(float f) {
    unsigned u = __float_as_uint(f);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (u16)(u >> 16);
}

// Workgroup = BM x BN output tile, THREADS/64 waves in a 2 x (WAVES/2) grid, each wave a (BM/2) x (BN/(WAVES/2))
// sub-tile of 32x32 MFMA blocks.  M ~ 1000 rows gives only ~1 workgroup per CU, so memory latency cannot be
// hidden by occupancy: every thread keeps NS K-tiles of global loads in flight in registers (the loads of
// tile kt+NS are issued before tile kt is computed) and the LDS tiles are double buffered, one barrier per
// K tile.  Workgroup ids are remapped so that the 1/8 of the grid an XCD receives (ids i mod 8) covers a
// compact band of column tiles: its L2 then holds that band of W plus A.
template <int BM, int BN, int THREADS, bool A16>
__global__ __launch_bounds__(THREADS) void gemm_bf16w_kernel(const GemmBf16Params p, const int mtiles, const int ntiles) {
    constexpr int WAVES = THREADS / 64, WN = WAVES / 2;
    constexpr int TM = BM / 2, TN = BN / WN;    // wave sub-tile
    constexpr int MI = TM / 32, NI = TN / 32;   // 32x32 MFMA blocks per wave
    constexpr int AL = BM * 8 / THREADS, BL = BN * 8 / THREADS;  // 8-element chunks per thread per K tile
    constexpr int RSTEP = THREADS / 8;          // rows covered by one pass of the workgroup
    constexpr int NS = 4;                       // K tiles in flight per thread
    extern __shared__ __attribute__((aligned(16))) u16 smem16[];
    u16* As = smem16;                  // [2][BM][LDB]
    u16* Bs = smem16 + 2 * BM * LDB;   // [2][BN][LDB]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int total = mtiles * ntiles;
    int t = blockIdx.x;
    if ((total & 7) == 0) t = (t & 7) * (total >> 3) + (t >> 3);
    const int tn = t / mtiles, tm = t - tn * mtiles;
    const int m0 = tm * BM, n0 = tn * BN;
    const int kchunk = tid & 7, lrow = tid >> 3;
    const int nk = p.K / BK16;

    const float* aptr[AL];
    const u16* aptr16[AL];
    const u16* bptr[BL];
#pragma unroll
    for (int i = 0; i < AL; ++i) {
        // rows/columns past the edge read the last valid one: their products are never stored, and the loop
        // stays branch free (the compiler can then count outstanding loads exactly)
        const int m = min(m0 + lrow + RSTEP * i, p.M - 1);
        aptr[i] = A16 ? nullptr : p.A + (size_t)m * p.K + kchunk * 4;  // fp32: k = 4c..4c+3 and 32+4c..32+4c+3 of each K tile
        aptr16[i] = A16 ? p.A16 + (size_t)m * p.K + kchunk * 8 : nullptr;  // bf16: k = 8c..8c+7
    }
#pragma unroll
    for (int i = 0; i < BL; ++i) {
        const int n = min(n0 + lrow + RSTEP * i, p.N - 1);
        bptr[i] = p.Wb + (size_t)n * p.K + kchunk * 8;
    }

    f32x4 ra[NS][AL][2];
    bf16x8 ra16[NS][AL];
    bf16x8 rb[NS][BL];
#define RGRG_LOAD_TILE(S, KT)                                                                    \
    {                                                                                            \
        const int koff = (KT) * BK16;                                                            \
        _Pragma("unroll") for (int i = 0; i < AL; ++i) {                                         \
            if constexpr (A16) {                                                                 \
                ra16[S][i] = *reinterpret_cast<const bf16x8*>(aptr16[i] + koff);                 \
            } else {                                                                             \
                ra[S][i][0] = *reinterpret_cast<const f32x4*>(aptr[i] + koff);                   \
                ra[S][i][1] = *reinterpret_cast<const f32x4*>(aptr[i] + koff + 32);              \
            }                                                                                    \
        }                                                                                        \
        _Pragma("unroll") for (int i = 0; i < BL; ++i)                                           \
            rb[S][i] = *reinterpret_cast<const bf16x8*>(bptr[i] + koff);                         \
    }
    // staging item I of register stage S -> LDS buffer BUF: items 0..AL-1 are A chunks (fp32 -> bf16, RNE),
    // items AL..AL+BL-1 are W chunks
#define RGRG_STORE_ITEM(S, BUF, I)                                                               \
    {                                                                                            \
        if ((I) < AL) {                                                                          \
            constexpr int i = (I) < AL ? (I) : 0;                                                \
            if constexpr (A16) {                                                                 \
                *reinterpret_cast<bf16x8*>(&As[((BUF) * BM + lrow + RSTEP * i) * LDB + kchunk * 8]) = ra16[S][i]; \
            } else {                                                                             \
                bf16x4 lo, hi;                                                                   \
                _Pragma("unroll") for (int e = 0; e < 4; ++e) {                                  \
                    lo[e] = (short)f32_to_bf16_rne(ra[S][i][0][e]);                              \
                    hi[e] = (short)f32_to_bf16_rne(ra[S][i][1][e]);                              \
                }                                                                                \
                u16* dst = &As[((BUF) * BM + lrow + RSTEP * i) * LDB + kchunk * 4];              \
                *reinterpret_cast<bf16x4*>(dst) = lo;                                            \
                *reinterpret_cast<bf16x4*>(dst + 32) = hi;                                       \
            }                                                                                    \
        } else {                                                                                 \
            constexpr int i = (I) >= AL ? (I) - AL : 0;                                          \
            *reinterpret_cast<bf16x8*>(&Bs[((BUF) * BN + lrow + RSTEP * i) * LDB + kchunk * 8]) = rb[S][i]; \
        }                                                                                        \
    }
#define RGRG_STORE_TILE(S, BUF) \
    { RGRG_STORE_ITEM(S, BUF, 0) RGRG_STORE_ITEM(S, BUF, 1) RGRG_STORE_ITEM(S, BUF, 2) RGRG_STORE_ITEM(S, BUF, 3) }
    static_assert(AL + BL == BK16 / 16, "one staging item per 16-wide K step");

    f32x16 acc[MI][NI];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;

    // prologue: tiles 0..NS-1 in flight, tile 0 staged
#pragma unroll
    for (int s = 0; s < NS; ++s) RGRG_LOAD_TILE(s, s)  // nk is a multiple of NS (checked by the launcher)
    __builtin_amdgcn_sched_barrier(0);
    RGRG_STORE_TILE(0, 0)
    __syncthreads();

    const int frow = lane & 31, fk = (lane >> 5) * 8;
    for (int kt0 = 0; kt0 < nk; kt0 += NS) {
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            const int kt = kt0 + s;
            {
                const int buf = kt & 1;
                // registers of stage s were staged to LDS one iteration ago: refill them with tile kt+NS (past
                // the end: the last tile again, never consumed)
                RGRG_LOAD_TILE(s, min(kt + NS, nk - 1))
                __builtin_amdgcn_sched_barrier(0);
                const u16* Ab = &As[(buf * BM + wm * TM + frow) * LDB + fk];
                const u16* Bb = &Bs[(buf * BN + wn * TN + frow) * LDB + fk];
                // the 4 K steps of the tile; the fragments of step ks+1 are read before the MFMAs of step ks, and
                // one quarter of tile kt+1 (the oldest loads in flight) is converted and written to the other LDS
                // buffer in the shadow of each step's MFMAs (its last readers passed the barrier of iteration kt-1)
                bf16x8 a[2][MI], b[2][NI];
#pragma unroll
                for (int mi = 0; mi < MI; ++mi) a[0][mi] = *reinterpret_cast<const bf16x8*>(Ab + mi * 32 * LDB);
#pragma unroll
                for (int ni = 0; ni < NI; ++ni) b[0][ni] = *reinterpret_cast<const bf16x8*>(Bb + ni * 32 * LDB);
#define RGRG_KSTEP(KS)                                                                                       \
    {                                                                                                        \
        if ((KS) + 1 < BK16 / 16) {                                                                          \
            _Pragma("unroll") for (int mi = 0; mi < MI; ++mi) a[((KS) + 1) & 1][mi] =                        \
                *reinterpret_cast<const bf16x8*>(Ab + mi * 32 * LDB + ((KS) + 1) * 16);                      \
            _Pragma("unroll") for (int ni = 0; ni < NI; ++ni) b[((KS) + 1) & 1][ni] =                        \
                *reinterpret_cast<const bf16x8*>(Bb + ni * 32 * LDB + ((KS) + 1) * 16);                      \
        }                                                                                                    \
        _Pragma("unroll") for (int mi = 0; mi < MI; ++mi) _Pragma("unroll") for (int ni = 0; ni < NI; ++ni)  \
            acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[(KS) & 1][mi], b[(KS) & 1][ni], acc[mi][ni], 0, 0, 0); \
        RGRG_STORE_ITEM((s + 1) % NS, buf ^ 1, KS)                                                           \
    }
                RGRG_KSTEP(0) RGRG_KSTEP(1) RGRG_KSTEP(2) RGRG_KSTEP(3)
#undef RGRG_KSTEP
                __syncthreads();
            }
        }
    }
#undef RGRG_LOAD_TILE
#undef RGRG_STORE_TILE
#undef RGRG_STORE_ITEM
    // epilogue (C/D map of the 32x32 MFMA: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)).  Residual
    // reads use clamped addresses and are issued together; only the stores are predicated.
    const int ccol = lane & 31, crow4 = 4 * (lane >> 5);
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) {
            const int col = n0 + wn * TN + ni * 32 + ccol;
            const int colc = min(col, p.N - 1);
            const int rbase = m0 + wm * TM + mi * 32 + crow4;
            const float sh = p.shift ? p.shift[colc] : 0.f;
            float rv[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) rv[r] = 0.f;
            if (p.R) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = min(rbase + (r & 3) + 8 * (r >> 2), p.M - 1);
                    rv[r] = p.R[(size_t)row * p.ldy + colc];
                }
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = rbase + (r & 3) + 8 * (r >> 2);
                const float v = apply_act(acc[mi][ni][r] + sh + rv[r], p.act);
                if (row < p.M && col < p.N) {
                    if (p.Y16) p.Y16[(size_t)row * p.ldy + col] = f32_to_bf16_rne(v);
                    else p.Y[(size_t)row * p.ldy + col] = v;
                }
            }
        }
}

__global__ __launch_bounds__(256) void f32_to_bf16_kernel(const float* __restrict__ src, u16* __restrict__ dst, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        dst[i] = f32_to_bf16_rne(src[i]);
}

template <int BM, int BN, int THREADS>
static int launch_bf16_cfg(const GemmBf16Params& p, hipStream_t st) {
    constexpr size_t lds = (size_t)(2 * BM + 2 * BN) * LDB * sizeof(u16);
    const int mtiles = (p.M + BM - 1) / BM, ntiles = (p.N + BN - 1) / BN;
    if (p.A16)
        hipLaunchKernelGGL((gemm_bf16w_kernel<BM, BN, THREADS, true>), dim3(mtiles * ntiles), dim3(THREADS), lds, st, p, mtiles, ntiles);
    else
        hipLaunchKernelGGL((gemm_bf16w_kernel<BM, BN, THREADS, false>), dim3(mtiles * ntiles), dim3(THREADS), lds, st, p, mtiles, ntiles);
    RGRG_LAUNCH_CHECK();
    return RGRG_OK;
}

template <int BM, int BN, int THREADS>
static int bf16_attr() {
    RGRG_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_bf16w_kernel<BM, BN, THREADS, false>),
                                 hipFuncAttributeMaxDynamicSharedMemorySize, 81920));
    RGRG_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_bf16w_kernel<BM, BN, THREADS, true>),
                                 hipFuncAttributeMaxDynamicSharedMemorySize, 81920));
    return RGRG_OK;
}

int init_gemm_bf16_attrs() {
    int rc;
    if ((rc = bf16_attr<128, 128, 512>())) return rc;
    return bf16_attr<64, 64, 256>();
}

// RGRG_BF16_TILE=1 (128x128, 8 waves) / 5 (64x64, 4 waves) forces one tile configuration (tools/gemm_bf16_bench.py); unset/0 = the heuristic
static int forced_bf16_cfg() {
    static const int v = [] {
        const char* e = getenv("RGRG_BF16_TILE");
        return e ? atoi(e) : 0;
    }();
    return v;
}

// A16 / Y16 (either may be null): bf16 activations in / out, see GemmBf16Params
int launch_gemm_bf16w_ex(const float* A, const void* A16, const void* Wb, const float* shift, const float* R, float* Y, void* Y16,
                         int M, int N, int K, int ldy, int act, hipStream_t st) {
    RGRG_CHECK_ARG((A || A16) && Wb && (Y || Y16) && M > 0 && N > 0 && K > 0 && K % (4 * BK16) == 0 && ldy >= N);  // 4 = depth NS
    GemmBf16Params p{A, reinterpret_cast<const u16*>(A16), reinterpret_cast<const u16*>(Wb), shift, R, Y,
                     reinterpret_cast<u16*>(Y16), M, N, K, ldy, act};
    int cfg = forced_bf16_cfg();
    if (cfg == 0) {
        const long tiles_big = (long)((M + 127) / 128) * ((N + 127) / 128);
        cfg = tiles_big >= 192 ? 1 : 5;
    }
    switch (cfg) {
        case 1: return launch_bf16_cfg<128, 128, 512>(p, st);
        default: return launch_bf16_cfg<64, 64, 256>(p, st);
    }
}

int launch_gemm_bf16w(const float* A, const void* Wb, const float* shift, const float* R, float* Y, int M, int N, int K,
                      int ldy, int act, hipStream_t st) {
    return launch_gemm_bf16w_ex(A, nullptr, Wb, shift, R, Y, nullptr, M, N, K, ldy, act, st);
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
