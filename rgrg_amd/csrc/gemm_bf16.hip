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

// optional LayerNorm-fold operands of launch_gemm_bf16w_ex (see GemmBf16Params)
struct GemmLnFold {
    void* Yb16 = nullptr;
    float* stats_out = nullptr;
    const float* ln_stats = nullptr;
    const float* ln_colsum = nullptr;
    void* Ypre16 = nullptr;       // training pass: GemmBf16Params::Ypre16 / G16
    const void* G16 = nullptr;
    int ksplit = 0;               // split-K (GemmBf16Params::ksplit / sk_ws / sk_cnt)
    float* sk_ws = nullptr;
    unsigned* sk_cnt = nullptr;
    float* cand_val = nullptr;    // arg-max candidates instead of Y (GemmBf16Params::cand_val / cand_idx)
    int* cand_idx = nullptr;
    int kp = 0;                   // 1: the K-parity ping-pong kernel (gemm_kp.inc), tile picked from (N, K) only
};

struct GemmBf16Params {
    const float* A;    // fp32 activations (rounded to bf16 while staged) ...
    const u16* A16;    // ... or activations already rounded to bf16 by their producer (A16 kernels)
    const u16* Wb;
    const float* shift;
    const float* R;
    float* Y;
    u16* Y16;          // non-null: the result is stored as bf16 [M, ldy] instead of fp32 (feeds the next GEMM only)
    int M, N, K, ldy, act;
    int lda = 0, ldw = 0;  // row pitch of A16 / Wb in elements (0 = K); LDS-DMA kernel only
    // LDS-DMA kernel, implicit-GEMM convolution (CONV instantiation): A16 = X [B,H,W,Cin] bf16 NHWC, PRECEDED IN MEMORY BY
    // 128 ZERO ELEMENTS (the line padding taps read); row m = output pixel (b, oy, ox), K = KH * KW * Cin (tap major)
    int cH = 0, cW = 0, cCin = 0, cKH = 0, cKW = 0, cStride = 1, cPad = 0, cOH = 0, cOW = 0;
    const u16* R16 = nullptr;  // residual as bf16 [M, ldy] (CONV: the bottleneck's identity branch)
    int f16 = 0;               // the 16-bit type of A16 / Wb / Y16 / R16: 0 = bf16, 1 = IEEE fp16 (common.h)
    // LayerNorm folded around the GEMMs of the decode step (LDS-DMA kernel, decoder.hip enqueue_step):
    //  * producer side (the GEMMs that write the residual stream x, N = the normalised width = 1024): besides Y (fp32) the
    //    result is stored as 16 bit in Yb16 [M, ldy] and every 64-column block leaves its per-row (sum, sum of squares) in
    //    stats_out [M][16][2];
    //  * consumer side (the GEMM behind the LayerNorm, K = 1024): A16 is the RAW 16-bit x, Wb holds gain-scaled weights, and
    //    the epilogue applies  rstd_m * (acc - mean_m * colsum_n) + shift_n  with mean / rstd of row m reduced (fixed order)
    //    from ln_stats [M][16][2]; colsum_n = the sum over k of the ROUNDED scaled weights, shift_n = bias + beta . W.
    u16* Yb16 = nullptr;
    float* stats_out = nullptr;
    const float* ln_stats = nullptr;
    const float* ln_colsum = nullptr;
    // training pass (round 5, LDS-DMA kernel, plain variant): Ypre16 [M, ldy] receives the 16-bit PRE-activation value next to
    // the activated Y16 (c_fc keeps what gelu_new' needs); G16 [M, ldy] holds 16-bit pre-activations whose gelu_new' multiplies
    // the result (the dgrad of mlp_proj lands directly as d(c_fc output))
    u16* Ypre16 = nullptr;
    const u16* G16 = nullptr;
    // tile order (LDS-DMA kernel): row tiles are walked in groups of gm (all column tiles of a group before the next group,
    // rows fastest inside a column), so the ~64 workgroups an XCD runs at a time form a gm x (64 / gm) block of the tile grid
    // and its L2 serves gm + 64 / gm operand panels instead of 65.  gm >= mtiles (or 0) = one group = column-major order.
    int gm = 0;
    // split-K (round 5, LDS-DMA kernel, not CONV): ksplit > 1 workgroups share an output tile, each multiplies K / ksplit;
    // every slice writes its fp32 accumulators to sk_ws [tile][slice][BM x BN], draws a ticket from sk_cnt [tile], and the LAST
    // arriver adds the slabs in slice order (its own from registers), runs the epilogue and resets the ticket - nobody waits.
    // For the N = 1024 projections of the many-sequence decode step (240 tiles of 64 x 64 on 256 CUs, a serial K loop of 16-64
    // tiles at the ~50 GB/s a CU delivers to ONE workgroup): two to four workgroups per CU overlap their K loops.
    int ksplit = 0;
    float* sk_ws = nullptr;
    unsigned* sk_cnt = nullptr;
    // ping-pong kernel, greedy decoding (lm_head of the many-sequence step): instead of storing the 256 x 256 block of logits a
    // workgroup leaves ONE (maximum, column) pair per row - first maximum wins, like torch.argmax - in cand_val / cand_idx
    // [M][ntiles]; argmax_update_kernel picks the row's token from the ceil(N / 256) candidates.  The 186 MB of fp32 logits per
    // step (923 x 50 257) are neither written nor re-read.  Y may be null.
    float* cand_val = nullptr;
    int* cand_idx = nullptr;
    int dbg = 0;   // ping-pong kernel, measurements only (RGRG_PP_DBG): 1 no fragment reads, 2 no refill DMAs, 4 no MFMAs, 8 no stores
};

constexpr float LN_EPS16 = 1e-5f;   // nn.LayerNorm(eps=1e-5) of GPT-2

constexpr int BK16 = 64;                 // k per tile
constexpr int LDB = (BK16 * 2 + 16) / 2;  // padded LDS row in bf16 elements (144 B)

// one 32x32x16 matrix-core step on 16-bit fragments of either type (fp32 accumulate)
template <bool F16>
__device__ __forceinline__ f32x16 mfma16(const bf16x8& a, const bf16x8& b, const f32x16& c) {
    if constexpr (F16) return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
    else return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}

// Workgroup = BM x BN output tile, THREADS/64 waves in a 2 x (WAVES/2) grid, each wave a (BM/2) x (BN/(WAVES/2))
// sub-tile of 32x32 MFMA blocks.  M ~ 1000 rows gives only ~1 workgroup per CU, so memory latency cannot be
// hidden by occupancy: every thread keeps NS K-tiles of global loads in flight in registers (the loads of
// tile kt+NS are issued before tile kt is computed) and the LDS tiles are double buffered, one barrier per
// K tile.  Workgroup ids are remapped so that the 1/8 of the grid an XCD receives (ids i mod 8) covers a
// compact band of column tiles: its L2 then holds that band of W plus A.
template <int BM, int BN, int THREADS, bool A16, bool F16>
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
                    lo[e] = (short)to16<F16>(ra[S][i][0][e]);                                    \
                    hi[e] = (short)to16<F16>(ra[S][i][1][e]);                                    \
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
            acc[mi][ni] = mfma16<F16>(a[(KS) & 1][mi], b[(KS) & 1][ni], acc[mi][ni]);                        \
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
                    if (p.Y16) p.Y16[(size_t)row * p.ldy + col] = (u16)to16<F16>(v);
                    else p.Y[(size_t)row * p.ldy + col] = v;
                }
            }
        }
}

// ------------------------------------------------------------------ LDS-DMA kernel (round 3): both operands bf16
// Y = act(A16 Wb^T + shift + R) with A16 [M,K] and Wb [N,K] both bf16 in HBM (the decode / box-head paths write their
// activations as bf16 once, so the GEMM never converts).  Built for the M ~ 1000 regime of BASELINE configs[2], where a
// GEMM is 2-8 GFLOP (16 K tiles of 64) and a workgroup lives ~15 us:
//   * operands go HBM/L2 -> LDS by `buffer_load_dwordx4 ... lds` (LDS-DMA): no staging registers, no ds_write pass, no
//     conversion; a wave instruction moves 8 tile rows x 128 B (one K tile of 64 bf16) = 1 KiB;
//   * NST LDS stages of (BM + BN) x 128 B; the loads of K tile kt + NST - 1 are issued while tile kt is multiplied, kept
//     in flight ACROSS the barriers with a counted `s_waitcnt vmcnt(N)` + raw `s_barrier` (a `__syncthreads()` would
//     drain them: its fence waits vmcnt(0)); ONE barrier per K tile: "my loads of tile kt have landed" (vmcnt) +
//     "everybody's have, and everybody is done reading tile kt - 1" (barrier), whose stage is then refilled;
//   * 4 waves as 2 x 2, each a (BM/2) x (BN/2) sub-tile of 32x32x16 MFMA blocks; fragment reads of K step ks + 1 are
//     issued before the MFMAs of step ks (order pinned with sched_group_barrier: hipcc re-serialises it otherwise);
//   * LDS rows are 128 B, so a fragment read (lane = row, 16 B at one K offset) would hit 2 of 16 slots of the 256-B
//     bank row 8 ways; the 16-byte chunk index is XORed with (row >> 1) & 7, which spreads the 16 rows of every
//     ds_read_b128 lane group over all 16 slots.  LDS-DMA writes lane-linear, so the swizzle is applied to the per-lane
//     SOURCE address (chunk' ^ f(row) of the same 128-B row segment: still whole cache lines) and again on the read;
//   * buffer descriptors are rebased to the tile (32-bit offsets stay small: fc6's A is 6.9 GB), rows past M / N read
//     the last valid row (never stored);
//   * tile id -> XCD-aware band (every XCD's L2 holds A plus one band of W), bijective for any tile count.
// What bounds it (profiles/HISTORY.md 6c, profiles/r03_gemm_bf16_bench_*.log, r03_pmc_gemm_bf16_v1.md): a workgroup spends about as
// long in its prologue (first tiles: HBM / cross-XCD latency) and epilogue as in its 16 K tiles, and a CU delivers
// ~25-45 GB/s of operands to ONE workgroup whatever its wave organisation (4 waves, 8 waves as two K halves, 4 + 4 loader
// waves, an L2-prefetch wave, K rotation, padded pitches: all measured within +-5 %, the prefetch wave -20 %).  What helps
// is MORE WORKGROUPS PER CU (their prologues / epilogues overlap): 64 x 64 tiles with 2-3 stages where the GEMM has few
// tiles, 128 x 128 with 2 stages (two workgroups per CU) for the vocabulary-sized lm_head.
typedef __attribute__((address_space(3))) void* lds_ptr_t;

__device__ __forceinline__ __amdgpu_buffer_rsrc_t bf16_rsrc(const void* p) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, 0x7fffffff, 0x00020000);
}

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
    static_assert(N >= 0 && N < 64, "vmcnt is a 6-bit field");
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// gelu_new for a bf16 consumer: x * sigmoid(2 sqrt(2/pi) (x + 0.044715 x^3)) on the hardware exp2 / rcp (~1e-7 relative
// to the tanhf form, far below bf16 resolution).  tanhf costs ~40 VALU instructions; a lane of the 128 x 128 tile has 64
// outputs and the wave is alone on its SIMD: the tanhf epilogue was 8 of the 25 us of c_fc.
__device__ __forceinline__ float gelu_new_fast(float x) {
    const float u = x * (1.0f + 0.044715f * x * x);
    return x * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-2.302208198f * u));  // -2 * sqrt(2/pi) * log2(e)
}
// gelu_new'(x) with s = sigmoid(2u), u = sqrt(2/pi) (x + 0.044715 x^3):  tanh u = 2s - 1, so
//   0.5 (1 + tanh u) + 0.5 x (1 - tanh^2 u) u'  =  s + 2 x s (1 - s) sqrt(2/pi) (1 + 3 * 0.044715 x^2)
__device__ __forceinline__ float gelu_new_grad_fast(float x) {
    const float x2 = x * x;
    const float u = x * (1.0f + 0.044715f * x2);
    const float sg = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-2.302208198f * u));
    return sg + 2.0f * x * sg * (1.0f - sg) * 0.7978845608028654f * (1.0f + 0.134145f * x2);
}
__device__ __forceinline__ float apply_act_fast(float v, int act) {
    if (act == RGRG_ACT_RELU) return v > 0.f ? v : 0.f;
    if (act == RGRG_ACT_GELU_NEW) return gelu_new_fast(v);
    return v;
}

// the LDS-DMA requests of one wave for one K tile: LA + LB wave instructions of 1 KiB (8 tile rows x 128 B); instruction
// j of wave w covers the rows (j * 4 + w) * 8 .. + 8 of its operand.
// (TAG: one specialization per calling kernel - hipcc's host-side pass rejects a second kernel template that reuses an
// already instantiated specialization of a function holding this builtin, "no matching function", ROCm 7.2; descriptors
// are built here from the tile's base pointers - loop-invariant scalar work that hipcc hoists)
template <int BM, int LA, int LB, int TAG>
__device__ __forceinline__ void glds_issue(const u16* abase, const u16* wbase, unsigned char* sb,
                                           const int (&va)[LA], const int (&vb)[LB], int koff_a, int koff_w) {
    const __amdgpu_buffer_rsrc_t ra = bf16_rsrc(abase), rb = bf16_rsrc(wbase);
#pragma unroll
    for (int j = 0; j < LA; ++j)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, (lds_ptr_t)(sb + j * 4096), 16, va[j], koff_a, 0, 0);
#pragma unroll
    for (int j = 0; j < LB; ++j)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rb, (lds_ptr_t)(sb + BM * 128 + j * 4096), 16, vb[j], koff_w, 0, 0);
}

// the MFMAs of one wave on one staged K tile (4 K steps of 16): sa / sw = the wave's first A / W row in the stage
template <int MI, int NI, bool F16>
__device__ __forceinline__ void glds_compute(const unsigned char* sa, const unsigned char* sw, const int (&foff)[4],
                                             f32x16 (&acc)[MI][NI]) {
    // two fragment register sets: the ds_reads of K step ks + 1 are issued before the MFMAs of step ks
    bf16x8 a[2][MI], b[2][NI];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) a[0][mi] = *reinterpret_cast<const bf16x8*>(sa + mi * 4096 + foff[0]);
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) b[0][ni] = *reinterpret_cast<const bf16x8*>(sw + ni * 4096 + foff[0]);
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
        if (ks + 1 < 4) {
#pragma unroll
            for (int mi = 0; mi < MI; ++mi) a[(ks + 1) & 1][mi] = *reinterpret_cast<const bf16x8*>(sa + mi * 4096 + foff[ks + 1]);
#pragma unroll
            for (int ni = 0; ni < NI; ++ni) b[(ks + 1) & 1][ni] = *reinterpret_cast<const bf16x8*>(sw + ni * 4096 + foff[ks + 1]);
        }
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
            for (int ni = 0; ni < NI; ++ni)
                acc[mi][ni] = mfma16<F16>(a[ks & 1][mi], b[ks & 1][ni], acc[mi][ni]);
    }
    // pin that order (hipcc otherwise re-serialises read -> wait -> MFMA per K step on ONE register set):
    // reads(0) reads(1) | MFMA(0) reads(2) | MFMA(1) reads(3) | MFMA(2) | MFMA(3)       (0x100 = DS read, 0x008 = MFMA)
    __builtin_amdgcn_sched_group_barrier(0x100, 2 * (MI + NI), 0);
    __builtin_amdgcn_sched_group_barrier(0x008, MI * NI, 0);
    __builtin_amdgcn_sched_group_barrier(0x100, MI + NI, 0);
    __builtin_amdgcn_sched_group_barrier(0x008, MI * NI, 0);
    __builtin_amdgcn_sched_group_barrier(0x100, MI + NI, 0);
    __builtin_amdgcn_sched_group_barrier(0x008, MI * NI, 0);
    __builtin_amdgcn_sched_group_barrier(0x008, MI * NI, 0);
}

// CONV: the A operand is the im2col view of an NHWC bf16 image - row m of a K tile = 64 channels [c0, c0 + 64) of input pixel
// (oy * stride + kh - pad, ox * stride + kw - pad): still one contiguous 128-byte line per row, so the LDS-DMA path is
// unchanged; only the per-lane source offset is recomputed per K tile (tap = kt * 64 / Cin, a handful of integer
// instructions), and taps outside the image read the zero line that precedes the tensor.
// LNF: 0 plain; 1 producer of a folded LayerNorm (Yb16 + stats_out); 2 consumer (ln_stats + ln_colsum) - GemmBf16Params.
// Compile-time so that the plain kernel carries none of it (as runtime branches the conditional loads made hipcc drain
// the whole operand prologue - vmcnt(0) - at the join in front of the K loop of EVERY variant).
template <int BM, int BN, int NST, bool CONV, bool F16, int LNF>
__global__ __launch_bounds__(256) void gemm_bf16_glds_kernel(const GemmBf16Params p, const int mtiles, const int ntiles) {
    constexpr int BK = 64;
    constexpr int MI = BM / 64, NI = BN / 64;   // 32x32 MFMA blocks per wave
    constexpr int LA = BM / 32, LB = BN / 32;   // LDS-DMA instructions (1 KiB = 8 rows) per wave per stage
    constexpr int LPW = LA + LB;
    constexpr int STAGE = (BM + BN) * 128;      // bytes per stage
    static_assert((NST - 2) * LPW < 64, "counted vmcnt out of range");
    static_assert(NST >= 2 && NST <= 4, "tail is written for up to 3 trailing tiles");
    extern __shared__ __attribute__((aligned(16))) unsigned char glds_smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int ks = (!CONV && p.ksplit > 1) ? p.ksplit : 1;   // K slices per output tile (consecutive workgroup ids: one XCD)
    int t = blockIdx.x;
    {
        const int total = mtiles * ntiles * ks, q = total >> 3, r = total & 7, x = t & 7, i = t >> 3;
        t = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + i;
    }
    const int slice = t % ks, tile = t / ks;
    t = tile;
    int tn, tm;
    if (p.gm <= 0 || p.gm >= mtiles) {
        tn = t / mtiles; tm = t - tn * mtiles;
    } else {   // grouped order (GemmBf16Params::gm): all groups but the last hold gm row tiles
        const int gsz = p.gm * ntiles, g = t / gsz, r = t - g * gsz;
        const int rows = min(p.gm, mtiles - g * p.gm);
        tn = r / rows; tm = g * p.gm + (r - tn * rows);
    }
    const int m0 = tm * BM, n0 = tn * BN;
    const int nk = p.K / BK / ks;          // K tiles of this slice (the launcher checks divisibility and nk >= NST - 1)
    const int lda = p.lda ? p.lda : p.K, ldw = p.ldw ? p.ldw : p.K;
    // buffer descriptors are rebased to the tile (and to the K slice); CONV: to the zero line in front of the image (offsets are absolute)
    const u16* abase = CONV ? p.A16 - 128 : p.A16 + (size_t)m0 * lda + (size_t)slice * nk * BK;
    const u16* wbase = p.Wb + (size_t)n0 * ldw + (size_t)slice * nk * BK;
    // per-lane source offsets (bytes) of this wave's LDS-DMA instructions: lane -> (row = lane / 8, LDS chunk = lane % 8)
    const int lrow = lane >> 3, lch = lane & 7;
    int va[LA], vb[LB];
    int cy0[LA], cx0[LA], cpix[LA], cswz[LA];   // CONV: top-left input coordinate, first pixel of the image, swizzled chunk
#pragma unroll
    for (int j = 0; j < LA; ++j) {
        const int row = (j * 4 + wave) * 8 + lrow;
        if constexpr (CONV) {
            const int m = min(m0 + row, p.M - 1);
            const int b = m / (p.cOH * p.cOW), rem = m - b * (p.cOH * p.cOW);
            const int oy = rem / p.cOW, ox = rem - oy * p.cOW;
            cy0[j] = oy * p.cStride - p.cPad;
            cx0[j] = ox * p.cStride - p.cPad;
            cpix[j] = b * p.cH * p.cW;
            cswz[j] = (lch ^ ((row >> 1) & 7)) << 4;
            va[j] = 0;
        } else {
            va[j] = min(row, p.M - 1 - m0) * lda * 2 + ((lch ^ ((row >> 1) & 7)) << 4);
        }
    }
#pragma unroll
    for (int j = 0; j < LB; ++j) {
        const int row = (j * 4 + wave) * 8 + lrow;
        vb[j] = min(row, p.N - 1 - n0) * ldw * 2 + ((lch ^ ((row >> 1) & 7)) << 4);
    }
#define RGRG_GLDS_ISSUE(STAGE_, KT_)                                                                                   \
    {                                                                                                                  \
        if constexpr (CONV) {                                                                                          \
            const int kbeg_ = (KT_) * BK, tap_ = kbeg_ / p.cCin, c0_ = kbeg_ - tap_ * p.cCin;                          \
            const int kh_ = tap_ / p.cKW, kw_ = tap_ - kh_ * p.cKW;                                                    \
            _Pragma("unroll") for (int j = 0; j < LA; ++j) {                                                           \
                const int iy_ = cy0[j] + kh_, ix_ = cx0[j] + kw_;                                                      \
                const bool in_ = (unsigned)iy_ < (unsigned)p.cH && (unsigned)ix_ < (unsigned)p.cW;                     \
                va[j] = (in_ ? 256 + (cpix[j] + iy_ * p.cW + ix_) * p.cCin * 2 + c0_ * 2 : 0) + cswz[j];               \
            }                                                                                                          \
            glds_issue<BM, LA, LB, LNF * 64 + NST * 4 + 2 + (F16 ? 1 : 0)>(abase, wbase, glds_smem + (STAGE_) * STAGE + wave * 1024, va, vb, 0, (KT_) * (BK * 2)); \
        } else {                                                                                                       \
            glds_issue<BM, LA, LB, LNF * 64 + NST * 4 + (F16 ? 1 : 0)>(abase, wbase, glds_smem + (STAGE_) * STAGE + wave * 1024, va, vb, (KT_) * (BK * 2),       \
                                            (KT_) * (BK * 2));                                                         \
        }                                                                                                              \
    }
    // fragment read offsets: lane -> (row = lane & 31, k half = lane >> 5) of a 32-row block; 4 K steps of 16
    const int frow = lane & 31, fh = lane >> 5, fsw = (frow >> 1) & 7;
    int foff[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) foff[ks] = frow * 128 + (((2 * ks + fh) ^ fsw) << 4);

    f32x16 acc[MI][NI];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;

    // Epilogue operands requested NOW, ahead of the operand stream: the per-column shift of every block and - for a 64 x 64
    // tile, where it is 16 registers - the fp32 residual of this lane's outputs.  Loaded in the epilogue they add a cold
    // L2 / HBM round trip to the tail of every launch.  (Vector loads return in order and these are the oldest ones, so the
    // counted vmcnt waits of the pipeline below are unaffected.  In-place residual (R == Y) is fine: an element is read and
    // written by the same thread only.)
    const int ccol = lane & 31, crow4 = 4 * (lane >> 5);
    float sh_pre[MI][NI], cs_pre[MI][NI];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) {
            const int colc = min(n0 + wn * (BN / 2) + ni * 32 + ccol, p.N - 1);
            sh_pre[mi][ni] = p.shift ? p.shift[colc] : 0.f;
            cs_pre[mi][ni] = LNF == 2 ? p.ln_colsum[colc] : 0.f;
        }
    constexpr bool PRE_R = MI * NI == 1 && LNF != 2;   // (a consumer of a folded LayerNorm has no residual: the launcher checks)
    float rv_pre[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) rv_pre[r] = 0.f;
    if constexpr (PRE_R) {
        if (p.R) {
            const int colc = min(n0 + wn * (BN / 2) + ccol, p.N - 1);
            const int rbase = m0 + wm * (BM / 2) + crow4;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = min(rbase + (r & 3) + 8 * (r >> 2), p.M - 1);
                rv_pre[r] = p.R[(size_t)row * p.ldy + colc];
            }
        }
    }

    // consumer of a folded LayerNorm: the (sum, sum of squares) slots of this tile's rows, requested ahead of the operand
    // stream like the other epilogue operands (the oldest loads: the counted waits of the pipeline are unaffected) and not
    // looked at before the K loop is through - reduced in front of the loop they held up its first barrier by a cold miss.
    // 256 / BM threads share a row, each adds its slots in index order.
    constexpr int TPR = 256 / BM, PER = 16 / TPR;   // 16 slots per row (64-column blocks of the 1024-wide producer)
    f32x4 lsq[PER / 2];
    if constexpr (LNF == 2) {
        const int srow = min(m0 + tid / TPR, p.M - 1);
        const f32x4* sp = reinterpret_cast<const f32x4*>(p.ln_stats + ((size_t)srow * 16 + (tid % TPR) * PER) * 2);
#pragma unroll
        for (int j = 0; j < PER / 2; ++j) lsq[j] = sp[j];
    }

#define RGRG_GLDS_COMPUTE(STAGE_)                                                                         \
    glds_compute<MI, NI, F16>(glds_smem + (STAGE_) * STAGE + wm * (BM / 2) * 128,                         \
                         glds_smem + (STAGE_) * STAGE + BM * 128 + wn * (BN / 2) * 128, foff, acc)

    // prologue: K tiles 0 .. NST-2 in flight (the launcher guarantees nk >= NST - 1)
#pragma unroll
    for (int s = 0; s < NST - 1; ++s) RGRG_GLDS_ISSUE(s, s);
    int kt = 0, cur = 0, nxt = NST - 1;  // stage of tile kt / stage the next issue fills
    for (; kt + NST - 1 < nk; ++kt) {
        wait_vmcnt<(NST - 2) * LPW>();   // this wave's loads of tile kt have landed (NST - 2 younger tiles may still fly)
        __builtin_amdgcn_s_barrier();    // ... everybody's have; and everybody has finished reading tile kt - 1
        __builtin_amdgcn_sched_barrier(0);
        RGRG_GLDS_ISSUE(nxt, kt + NST - 1);  // refill the stage of tile kt - 1
        __builtin_amdgcn_sched_barrier(0);
        RGRG_GLDS_COMPUTE(cur);
        cur = cur + 1 == NST ? 0 : cur + 1;
        nxt = nxt + 1 == NST ? 0 : nxt + 1;
    }
    // tail: the last NST - 1 tiles, nothing left to issue
#define RGRG_GLDS_TAIL(R)                                   \
    if constexpr (NST - 1 >= (R)) {                         \
        wait_vmcnt<((R) - 1) * LPW>();                      \
        __builtin_amdgcn_s_barrier();                       \
        __builtin_amdgcn_sched_barrier(0);                  \
        RGRG_GLDS_COMPUTE(cur);                             \
        cur = cur + 1 == NST ? 0 : cur + 1;                 \
    }
    RGRG_GLDS_TAIL(3) RGRG_GLDS_TAIL(2) RGRG_GLDS_TAIL(1)
#undef RGRG_GLDS_TAIL
#undef RGRG_GLDS_ISSUE
#undef RGRG_GLDS_COMPUTE

    if constexpr (!CONV) {
        if (ks > 1) {
            // split-K: publish this slice's accumulators with WRITE-THROUGH 16-byte stores (sc1: no release fence - a
            // `buffer_wbl2` writes back the whole XCD's L2, and 480 workgroups doing that cost more than the K loops they save:
            // measured +15 us per GEMM), every wave drains, barrier, ONE relaxed agent-scope ticket; all but the last arriver
            // leave; the last arriver reads the other slabs with sc1 loads (they bypass its L1; the producers stored sc1, so no
            // acquire fence either - MI355X_MICROARCH.md "valid forms")
            const __amdgpu_buffer_rsrc_t rs = bf16_rsrc(p.sk_ws + (size_t)tile * ks * (BM * BN));
            const int soff = slice * (BM * BN * 4);
#pragma unroll
            for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                for (int ni = 0; ni < NI; ++ni)
#pragma unroll
                    for (int q = 0; q < 4; ++q)
                        __builtin_amdgcn_raw_buffer_store_b128(
                            __builtin_bit_cast(u32x4, f32x4{acc[mi][ni][4 * q], acc[mi][ni][4 * q + 1], acc[mi][ni][4 * q + 2], acc[mi][ni][4 * q + 3]}),
                            rs, (((mi * NI + ni) * 4 + q) * 256 + tid) * 16, soff, 16);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            unsigned* flag = reinterpret_cast<unsigned*>(glds_smem + NST * STAGE + BM * 8);   // behind row_stat, inside the one LDS array
            if (tid == 0) *flag = __hip_atomic_fetch_add(p.sk_cnt + tile, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __syncthreads();
            if (*flag != (unsigned)(ks - 1)) return;
            if (tid == 0) __hip_atomic_store(p.sk_cnt + tile, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // for the next launch (graph replay)
            // the sum in slice order, this workgroup's own slice from its registers: independent of who arrives last
#pragma unroll
            for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                for (int ni = 0; ni < NI; ++ni)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        f32x4 tot = {0.f, 0.f, 0.f, 0.f};
                        for (int sl = 0; sl < ks; ++sl) {
                            f32x4 v = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, (((mi * NI + ni) * 4 + q) * 256 + tid) * 16,
                                                                                                         sl * (BM * BN * 4), 16));
                            if (sl == slice) v = f32x4{acc[mi][ni][4 * q], acc[mi][ni][4 * q + 1], acc[mi][ni][4 * q + 2], acc[mi][ni][4 * q + 3]};
                            tot += v;
                        }
#pragma unroll
                        for (int e = 0; e < 4; ++e) acc[mi][ni][4 * q + e] = tot[e];
                    }
        }
    }

    // folded LayerNorm: (mean, rstd) of the tile's rows -> LDS behind the stages (the last stage may still be read by slower
    // waves), from where the epilogue fetches the 16 rows of its C layout
    float2* row_stat = reinterpret_cast<float2*>(glds_smem + NST * STAGE);   // [BM]
    if constexpr (LNF == 2) {
        // ONE summation order for every tile shape (a row's result must not depend on the tile the launcher picks for M): the 16
        // slots are added in groups of four (slot order, from zero), the groups pairwise: (G0 + G1) + (G2 + G3).  64-row tiles:
        // a thread owns one group, the butterfly below adds them; 128-row tiles: a thread owns two groups and adds them first.
        static_assert(PER == 4 || PER == 8, "slot reduction is written for 64- and 128-row tiles");
        float ls1 = 0.f, ls2 = 0.f;
#pragma unroll
        for (int j0 = 0; j0 < PER / 2; j0 += 2) {
            float g1 = 0.f, g2 = 0.f;
#pragma unroll
            for (int j = j0; j < j0 + 2; ++j) { g1 += lsq[j][0]; g2 += lsq[j][1]; g1 += lsq[j][2]; g2 += lsq[j][3]; }
            if (j0 == 0) { ls1 = g1; ls2 = g2; } else { ls1 += g1; ls2 += g2; }
        }
#pragma unroll
        for (int o = 1; o < TPR; o <<= 1) { ls1 += __shfl_xor(ls1, o, 64); ls2 += __shfl_xor(ls2, o, 64); }
        const float mean = ls1 * (1.0f / 1024.0f);
        const float var = fmaxf(ls2 * (1.0f / 1024.0f) - mean * mean, 0.f);
        if (tid % TPR == 0) row_stat[tid / TPR] = make_float2(mean, 1.0f / sqrtf(var + LN_EPS16));
        __syncthreads();
    }

    float blk1[MI], blk2[MI];   // producer of a folded LayerNorm: this wave's row sums over its column blocks
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) { blk1[mi] = 0.f; blk2[mi] = 0.f; }
    // epilogue (C/D map of the 32x32 MFMA: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)).  Residual reads use
    // clamped rows and are issued together; only the stores are predicated.  Offsets are 32-bit inside the tile's rows.
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) {
            const int col = n0 + wn * (BN / 2) + ni * 32 + ccol;
            const int colc = min(col, p.N - 1);
            const int rbase = m0 + wm * (BM / 2) + mi * 32 + crow4;
            const float sh = sh_pre[mi][ni];
            float rv[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) rv[r] = rv_pre[r];   // 64 x 64 tiles: the residual prefetched above (zeros otherwise)
            if (!PRE_R && LNF != 2 && p.R) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = min(rbase + (r & 3) + 8 * (r >> 2), p.M - 1);
                    rv[r] = p.R[(size_t)row * p.ldy + colc];
                }
            }
            if (LNF != 2 && p.R16) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = min(rbase + (r & 3) + 8 * (r >> 2), p.M - 1);
                    rv[r] = from16<F16>(p.R16[(size_t)row * p.ldy + colc]);
                }
            }
            float vv[16];
            if constexpr (LNF == 2) {
                const float cs = cs_pre[mi][ni];
                const int lrow0 = wm * (BM / 2) + mi * 32 + crow4;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float2 st = row_stat[lrow0 + (r & 3) + 8 * (r >> 2)];
                    vv[r] = apply_act_fast(st.y * (acc[mi][ni][r] - st.x * cs) + sh + rv[r], p.act);
                }
            } else if constexpr (LNF == 0 && !CONV) {
                if (p.G16) {   // dgrad through gelu_new: the result times gelu_new'(saved pre-activation)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int row = min(rbase + (r & 3) + 8 * (r >> 2), p.M - 1);
                        rv[r] = from16<F16>(p.G16[(size_t)row * p.ldy + colc]);
                    }
#pragma unroll
                    for (int r = 0; r < 16; ++r) vv[r] = (acc[mi][ni][r] + sh) * gelu_new_grad_fast(rv[r]);
                } else {
                    if (p.Ypre16 && col < p.N) {
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            const int dr = (r & 3) + 8 * (r >> 2);
                            if (rbase + dr < p.M)
                                p.Ypre16[(size_t)rbase * p.ldy + col + (size_t)(dr * p.ldy)] = (u16)to16<F16>(acc[mi][ni][r] + sh + rv[r]);
                        }
                    }
#pragma unroll
                    for (int r = 0; r < 16; ++r) vv[r] = apply_act_fast(acc[mi][ni][r] + sh + rv[r], p.act);
                }
            } else {
#pragma unroll
                for (int r = 0; r < 16; ++r) vv[r] = apply_act_fast(acc[mi][ni][r] + sh + rv[r], p.act);
            }
            if (col < p.N) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int dr = (r & 3) + 8 * (r >> 2);
                    const float v = vv[r];
                    if (rbase + dr < p.M) {
                        const size_t o = (size_t)rbase * p.ldy + col + (size_t)(dr * p.ldy);
                        if (p.Y16) p.Y16[o] = (u16)to16<F16>(v);
                        else if (p.N > 8192) __builtin_nontemporal_store(v, &p.Y[o]);  // logits: streamed, keep A / W in L2
                        else p.Y[o] = v;
                        if constexpr (LNF == 1) p.Yb16[o] = (u16)to16<F16>(v);
                    }
                }
            }
            if constexpr (LNF == 1) {
                // per-row (sum, sum of squares) of this 32-column block: a halving butterfly over the 32 lanes that share the
                // block's rows (16 -> 8 -> 4 -> 2 -> 1 rows per lane, then the pair), fixed order.  (Written with constant
                // steps: as nested loops hipcc kept the arrays indexed dynamically - ~900 v_cndmask, 1.5 us per launch.)
                float a1[16], a2[16];
#pragma unroll
                for (int r = 0; r < 16; ++r) { a1[r] = vv[r]; a2[r] = vv[r] * vv[r]; }
#define RGRG_BFLY(H, BIT)                                                                         \
                {                                                                                         \
                    const bool up = (lane & (BIT)) != 0;                                                  \
                    _Pragma("unroll") for (int i = 0; i < (H); ++i) {                                     \
                        const float s1 = up ? a1[i] : a1[i + (H)], s2 = up ? a2[i] : a2[i + (H)];         \
                        const float k1 = up ? a1[i + (H)] : a1[i], k2 = up ? a2[i + (H)] : a2[i];         \
                        a1[i] = k1 + __shfl_xor(s1, (BIT), 64);                                           \
                        a2[i] = k2 + __shfl_xor(s2, (BIT), 64);                                           \
                    }                                                                                     \
                }
                RGRG_BFLY(8, 16) RGRG_BFLY(4, 8) RGRG_BFLY(2, 4) RGRG_BFLY(1, 2)
#undef RGRG_BFLY
                a1[0] += __shfl_xor(a1[0], 1, 64);
                a2[0] += __shfl_xor(a2[0], 1, 64);
                // lane L (and L ^ 1) now holds row (L >> 1) & 15 of its half of the block; blocks of one row block add up in ni order
                blk1[mi] += a1[0];
                blk2[mi] += a2[0];
            }
        }
    if constexpr (LNF == 1) {
        // one slot per 64 columns: a 128-wide tile's wave already covers 64 (its two blocks were added above); in a 64-wide tile
        // the two waves of a row block each hold 32 columns and meet in LDS (behind the stages), wn = 0 adds and stores
        float2* xch = reinterpret_cast<float2*>(glds_smem + NST * STAGE);   // [4 waves][MI][32 rows]
        const int r = (lane >> 1) & 15, lrow = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        if constexpr (NI == 1) {
            if (wn == 1 && !(lane & 1)) {
#pragma unroll
                for (int mi = 0; mi < MI; ++mi) xch[(wave * MI + mi) * 32 + lrow] = make_float2(blk1[mi], blk2[mi]);
            }
            __syncthreads();
        }
        if (!(lane & 1) && (NI == 2 || wn == 0)) {
#pragma unroll
            for (int mi = 0; mi < MI; ++mi) {
                float s1 = blk1[mi], s2 = blk2[mi];
                if constexpr (NI == 1) {
                    const float2 o = xch[((wave + 1) * MI + mi) * 32 + lrow];
                    s1 += o.x; s2 += o.y;
                }
                const int row = m0 + wm * (BM / 2) + mi * 32 + lrow;
                const int slot = NI == 2 ? (n0 + wn * 64) / 64 : n0 / 64;
                if (row < p.M) *reinterpret_cast<float2*>(p.stats_out + ((size_t)row * 16 + slot) * 2) = make_float2(s1, s2);
            }
        }
    }
}

// ------------------------------------------------------------------ 256 x 256 ping-pong kernel (round 5): both operands 16-bit
// The kernel for the LARGE GEMMs (training step at M = 14 848 rows, fc6, lm_head over thousands of rows).  What bounds the
// 128 x 128 kernel above on such shapes is not the matrix core but operand delivery: a 128 x 128 x 64 K tile moves 32 KiB
// L2 -> LDS for 16 MFMAs per wave, and the texture-address path of a CU takes 64 B per clock - with two workgroups per CU that
// is 100 % of it for 100 % of the matrix core (measured ceiling ~0.35-0.4 of peak).  A 256 x 256 tile halves the bytes per flop.
//   * 8 waves = two GROUPS of four (group g = rows g * 128 .. + 128 of the tile; wave wq of a group = columns wq * 64 .. + 64), a
//     wave owns 128 x 64 = 4 x 2 blocks of v_mfma_f32_32x32x16 (128 accumulator registers); waves w and w + 4 share a SIMD;
//   * PING-PONG: time is cut into slots by workgroup barriers; in every slot ONE group multiplies (16 MFMAs per wave on
//     fragments already in registers) while the OTHER reads its next fragments from LDS and issues the LDS-DMA requests -
//     every SIMD always has one wave in the matrix core and one on the LDS / DMA side, and the multiplying waves issue nothing
//     but MFMAs.  A wave's K tile is split by ROWS: first its rows 0..63 against its 64 columns (A and W fragments of the whole
//     K tile: 16 ds_read_b128), then its rows 64..127 (8 reads; the W fragments stay in registers):
//         slot      group 0                         group 1
//         4kt + 0   read A-lo[0:64], W  of kt       multiply rows 64.. of kt - 1
//         4kt + 1   multiply rows 0..63             read A-hi[0:64], W of kt
//         4kt + 2   read A-lo[64:128]               multiply rows 0..63
//         4kt + 3   multiply rows 64..127           read A-hi[64:128]
//   * two LDS stages of (256 + 256) rows x 128 B = 2 x 64 KiB (one workgroup per CU).  Every row group of a stage is read in
//     exactly ONE slot of a K tile (W in two: slots 0 and 1), so it is REFILLED - with the rows of the K tile two ahead - by the
//     reading group of a following slot, and every request has at least FOUR slots (~1 us) to land before its rows are read
//     (measured: a request lands ~700 cycles after issue under this load, `profiles/r05_pp_ablation_*.log`):
//         issued in slot    by        A rows (64 = 8 instructions of 1 KiB)     W rows
//         4kt + 0           group 0   A-hi[64:128] of kt + 1                    -
//         4kt + 1           group 1   A-lo[0:64]   of kt + 2                    -
//         4kt + 2           group 0   A-hi[0:64]   of kt + 2                    W[0:128]   of kt + 2 (16 instructions)
//         4kt + 3           group 1   A-lo[64:128] of kt + 2                    W[128:256] of kt + 2
//     A reading wave issues its 2 or 6 requests behind its ds_reads and then waits `vmcnt(8)`: all but its requests of this and
//     of its previous reading slot have landed (loads return in order); `lgkmcnt(0)` before the barrier frees the rows it read;
//   * tile order as in the kernel above (XCD bands, grouped rows); shift, residual, activation, 16-bit outputs, gelu' multiplier.
//     No LayerNorm fold, no convolution (the callers of those shapes have few row tiles).
template <int TAG>
__device__ __forceinline__ void pp_dma4(const u16* base, unsigned char* dst, const int (&v)[4], int soff) {
    const __amdgpu_buffer_rsrc_t r = bf16_rsrc(base);
#pragma unroll
    for (int j = 0; j < 4; ++j) __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_ptr_t)(dst + j * 8192), 16, v[j], soff, 0, 0);
}
template <int TAG>
__device__ __forceinline__ void pp_dma2(const u16* base, unsigned char* dst, int v0, int v1, int soff) {
    const __amdgpu_buffer_rsrc_t r = bf16_rsrc(base);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_ptr_t)dst, 16, v0, soff, 0, 0);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_ptr_t)(dst + 1024), 16, v1, soff, 0, 0);
}
template <bool F16>
__global__ __launch_bounds__(512) void gemm_bf16_pp_kernel(const GemmBf16Params p, const int mtiles, const int ntiles) {
    constexpr int BM = 256, BN = 256, BK = 64, MI = 4, NI = 2;
    constexpr int STAGE = (BM + BN) * 128, WOFF = BM * 128;
    constexpr int TAG = F16 ? 901 : 900;
    extern __shared__ __attribute__((aligned(16))) unsigned char pp_smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = wave >> 2, wq = wave & 3;
    int t = blockIdx.x;
    {
        const int total = mtiles * ntiles, q = total >> 3, r = total & 7, x = t & 7, i = t >> 3;
        t = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + i;
    }
    int tn, tm;
    if (p.gm <= 0 || p.gm >= mtiles) {
        tn = t / mtiles; tm = t - tn * mtiles;
    } else {
        const int gsz = p.gm * ntiles, gi = t / gsz, r = t - gi * gsz;
        const int rows = min(p.gm, mtiles - gi * p.gm);
        tn = r / rows; tm = gi * p.gm + (r - tn * rows);
    }
    const int m0 = tm * BM, n0 = tn * BN;
    const int nk = p.K / BK;
    const int lda = p.lda ? p.lda : p.K, ldw = p.ldw ? p.ldw : p.K;
    const u16* abase = p.A16 + (size_t)m0 * lda;
    const u16* wbase = p.Wb + (size_t)n0 * ldw;
    const int lrow = lane >> 3, lch = lane & 7;
    // per-lane source offset (bytes) of the 8 tile rows [row0, row0 + 8) of an operand: row clamp + XOR-swizzled 16-byte chunk
    auto src_off = [&](int row0, int limit, int ld) { const int row = row0 + lrow; return min(row, limit) * ld * 2 + ((lch ^ ((row >> 1) & 7)) << 4); };
    // steady state (table above): in its first reading slot of a K tile a wave refills 16 rows of one A unit, in its second
    // 16 rows of another A unit and 16 rows of each of two 64-row W units
    const int ra0 = (g == 0 ? 192 : 0) + wq * 16, ra1 = (g == 0 ? 128 : 64) + wq * 16;
    const int rw0 = (g == 0 ? 0 : 128) + wq * 16, rw1 = rw0 + 64;
    const int va00 = src_off(ra0, p.M - 1 - m0, lda), va01 = src_off(ra0 + 8, p.M - 1 - m0, lda);
    const int va10 = src_off(ra1, p.M - 1 - m0, lda), va11 = src_off(ra1 + 8, p.M - 1 - m0, lda);
    const int vw00 = src_off(rw0, p.N - 1 - n0, ldw), vw01 = src_off(rw0 + 8, p.N - 1 - n0, ldw);
    const int vw10 = src_off(rw1, p.N - 1 - n0, ldw), vw11 = src_off(rw1 + 8, p.N - 1 - n0, ldw);
    unsigned char* const da0 = pp_smem + ra0 * 128;          // + stage * STAGE
    unsigned char* const da1 = pp_smem + ra1 * 128;
    unsigned char* const dw0 = pp_smem + WOFF + rw0 * 128;
    unsigned char* const dw1 = pp_smem + WOFF + rw1 * 128;
    // fragment reads: lane -> (row = lane & 31, k half = lane >> 5) of a 32-row block, 4 K steps of 16 per tile
    const int frow = lane & 31, fh = lane >> 5, fsw = (frow >> 1) & 7;
    int foff[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) foff[ks] = frow * 128 + (((2 * ks + fh) ^ fsw) << 4);
    const unsigned char* const fa_base = pp_smem + g * (128 * 128);
    const unsigned char* const fb_base = pp_smem + WOFF + wq * (64 * 128);

    f32x16 acc[MI][NI];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;
    bf16x8 fa[4][2], fb[4][NI];   // [K step][32-row block of the current row half] / [K step][32-column block]
    // the shift of this lane's two output columns, requested ahead of the operand stream (loaded in the epilogue, between the
    // stores of two blocks, a load's wait also waits for every store issued before it: one counter, in order)
    float esh[NI];
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) esh[ni] = p.shift ? p.shift[min(n0 + wq * 64 + ni * 32 + (lane & 31), p.N - 1)] : 0.f;
    const int dbg = __builtin_amdgcn_readfirstlane(p.dbg);
    if (dbg) {   // measurement modes: defined operands whatever is skipped
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
#pragma unroll
            for (int i = 0; i < 2; ++i) { fa[ks][i] = bf16x8{1, 2, 3, 4, 5, 6, 7, 8}; fb[ks][i] = bf16x8{8, 7, 6, 5, 4, 3, 2, 1}; }
    }

#define PP_READ_A(ST_, H_)                                                                                       \
    if (!(dbg & 1)) {                                                                                            \
        const unsigned char* sa_ = fa_base + (ST_) * STAGE + (H_) * (64 * 128);                                  \
        _Pragma("unroll") for (int ks = 0; ks < 4; ++ks)                                                         \
            _Pragma("unroll") for (int m2 = 0; m2 < 2; ++m2)                                                     \
                fa[ks][m2] = *reinterpret_cast<const bf16x8*>(sa_ + m2 * 4096 + foff[ks]);                       \
    }
#define PP_READ_W(ST_)                                                                                           \
    if (!(dbg & 1)) {                                                                                            \
        const unsigned char* sw_ = fb_base + (ST_) * STAGE;                                                      \
        _Pragma("unroll") for (int ks = 0; ks < 4; ++ks)                                                         \
            _Pragma("unroll") for (int ni = 0; ni < NI; ++ni)                                                    \
                fb[ks][ni] = *reinterpret_cast<const bf16x8*>(sw_ + ni * 4096 + foff[ks]);                       \
    }
    // 16 MFMAs of row half H_
#define PP_MUL(H_)                                                                                               \
    if (!(dbg & 4)) {                                                                                            \
        __builtin_amdgcn_s_setprio(1);                                                                           \
        _Pragma("unroll") for (int ks = 0; ks < 4; ++ks)                                                         \
            _Pragma("unroll") for (int m2 = 0; m2 < 2; ++m2)                                                     \
                _Pragma("unroll") for (int ni = 0; ni < NI; ++ni)                                                \
                    acc[(H_) * 2 + m2][ni] = mfma16<F16>(fa[ks][m2], fb[ks][ni], acc[(H_) * 2 + m2][ni]);        \
        __builtin_amdgcn_s_setprio(0);                                                                           \
    }
#define PP_BAR()                                  \
    {                                             \
        __builtin_amdgcn_sched_barrier(0);        \
        __builtin_amdgcn_s_barrier();             \
        __builtin_amdgcn_sched_barrier(0);        \
    }
    // end of a reading slot: this slot's refills (FIRST_: the group's first reading slot of the K tile - one A unit into K tile
    // TA_; else one A unit and two W units into K tile TA_), the rows just read are free (lgkmcnt), all but the WAIT_ youngest
    // requests of this wave have landed, barrier
#define PP_END_READ(FIRST_, DO_, TA_, WAIT_)                                                                                  \
    {                                                                                                                          \
        if ((DO_) && !(dbg & 2)) {                                                                                             \
            pp_dma2<TAG>(abase, ((FIRST_) ? da0 : da1) + ((TA_) & 1) * STAGE, (FIRST_) ? va00 : va10, (FIRST_) ? va01 : va11, (TA_) * (BK * 2)); \
            if (!(FIRST_)) {                                                                                                   \
                pp_dma2<TAG>(wbase, dw0 + ((TA_) & 1) * STAGE, vw00, vw01, (TA_) * (BK * 2));                                  \
                pp_dma2<TAG>(wbase, dw1 + ((TA_) & 1) * STAGE, vw10, vw11, (TA_) * (BK * 2));                                  \
            }                                                                                                                  \
        }                                                                                                                      \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                                     \
        if (dbg & 2) wait_vmcnt<0>(); else wait_vmcnt<WAIT_>();                                                                \
        PP_BAR()                                                                                                               \
    }

    // prologue: K tiles 0 and 1 requested by all eight waves (both stages are free), tile 0 waited for
    {
        int vpa[4], vpw[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            vpa[j] = src_off((j * 8 + wave) * 8, p.M - 1 - m0, lda);
            vpw[j] = src_off((j * 8 + wave) * 8, p.N - 1 - n0, ldw);
        }
        unsigned char* const dst = pp_smem + wave * 1024;
        pp_dma4<TAG>(wbase, dst + WOFF, vpw, 0);
        pp_dma4<TAG>(abase, dst, vpa, 0);
        pp_dma4<TAG>(wbase, dst + STAGE + WOFF, vpw, BK * 2);
        pp_dma4<TAG>(abase, dst + STAGE, vpa, BK * 2);
    }
    wait_vmcnt<8>();
    PP_BAR()

    // One K tile of a group's program (both groups pass the same four barriers).  D0_ / D2_ (D1_ / D3_): do the refills of the
    // group's two reading slots exist for this kt (K tiles 0 and 1 come from the prologue; nothing lies beyond tile nk - 1);
    // W0_ .. W3_: how many of the wave's youngest requests may still be in flight at the end of that slot.
#define PP_TILE_G0(D0_, W0_, D2_, W2_)                                              \
    {                                                                               \
        const int cur = kt & 1;                                                     \
        PP_READ_A(cur, 0) PP_READ_W(cur)                                            \
        PP_END_READ(true, D0_, kt + 1, W0_)                                         \
        PP_MUL(0)                                                                   \
        PP_BAR()                                                                    \
        PP_READ_A(cur, 1)                                                           \
        PP_END_READ(false, D2_, kt + 2, W2_)                                        \
        PP_MUL(1)                                                                   \
        PP_BAR()                                                                    \
    }
#define PP_TILE_G1(FIRSTKT_, D1_, W1_, D3_, W3_)                                    \
    {                                                                               \
        const int cur = kt & 1;                                                     \
        if (!(FIRSTKT_)) PP_MUL(1)                                                  \
        PP_BAR()                                                                    \
        PP_READ_A(cur, 0) PP_READ_W(cur)                                            \
        PP_END_READ(true, D1_, kt + 2, W1_)                                         \
        PP_MUL(0)                                                                   \
        PP_BAR()                                                                    \
        PP_READ_A(cur, 1)                                                           \
        PP_END_READ(false, D3_, kt + 2, W3_)                                        \
    }
    if (g == 0) {
        int kt = 0;
        PP_TILE_G0(0, 0, 1, 6)                                   // the prologue's requests have all landed after slot 0
        for (kt = 1; kt < nk - 2; ++kt) PP_TILE_G0(1, 8, 1, 8)
        PP_TILE_G0(1, 8, 0, 2)                                   // kt = nk - 2: A-hi[64:128] of the last tile, nothing else
        ++kt;
        PP_TILE_G0(0, 0, 0, 0)                                   // kt = nk - 1
    } else {
        int kt = 0;
        PP_TILE_G1(true, 1, 2, 1, 8)
        for (kt = 1; kt < nk - 2; ++kt) PP_TILE_G1(false, 1, 8, 1, 8)
        PP_TILE_G1(false, 0, 6, 0, 0)                            // kt = nk - 2
        ++kt;
        PP_TILE_G1(false, 0, 0, 0, 0)
        PP_MUL(1)
    }
#undef PP_TILE_G0
#undef PP_TILE_G1
#undef PP_END_READ
#undef PP_MUL
#undef PP_READ_A
#undef PP_READ_W
#undef PP_BAR

    if (dbg & 8) {   // no stores: keep the accumulators alive
        float z = 0.f;
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
            for (int ni = 0; ni < NI; ++ni)
#pragma unroll
                for (int r = 0; r < 16; ++r) z += acc[mi][ni][r];
        if (z == 12345.678f) p.Y[0] = z;
        return;
    }
    if (p.cand_val) {
        // Row arg-max of the block.  A lane holds, per 32-row block mi, 16 rows x 2 columns (ni): combine the two columns, then a
        // butterfly over the 32 lanes of its half that HALVES the rows a lane carries at every step (lane ^ 16: rows 0-7 stay in
        // the lower lane, 8-15 in the upper one, ...): 8 + 4 + 2 + 1 + 1 exchanges per block instead of 16 x 5.  Ties keep the
        // lower column (explicit index compare), columns >= N are -inf.  The four column quarters (waves) of a row meet in LDS.
        __syncthreads();   // every wave is done with the operand stages: the LDS is reused below
        float* cv = reinterpret_cast<float*>(pp_smem);            // [2 groups][4 quarters][128 rows]
        int* ci = reinterpret_cast<int*>(pp_smem + 4096);
        const int ccol = lane & 31;
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) {
            float bv[16];
            int bi[16];
            const int c0 = n0 + wq * 64 + ccol, c1 = c0 + 32;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float v0 = c0 < p.N ? acc[mi][0][r] + esh[0] : -INFINITY;
                const float v1 = c1 < p.N ? acc[mi][1][r] + esh[1] : -INFINITY;
                const bool hi = v1 > v0;
                bv[r] = hi ? v1 : v0;
                bi[r] = hi ? c1 : c0;
            }
#define PP_CAND_STEP(CNT_, XOR_)                                                                   \
    {                                                                                              \
        const bool up = (lane & (XOR_)) != 0;                                                      \
        _Pragma("unroll") for (int j = 0; j < (CNT_); ++j) {                                       \
            const float keep_v = up ? bv[j + (CNT_)] : bv[j], send_v = up ? bv[j] : bv[j + (CNT_)]; \
            const int keep_i = up ? bi[j + (CNT_)] : bi[j], send_i = up ? bi[j] : bi[j + (CNT_)];   \
            const float ov = __shfl_xor(send_v, (XOR_), 64);                                       \
            const int oi = __shfl_xor(send_i, (XOR_), 64);                                         \
            const bool take = ov > keep_v || (ov == keep_v && oi < keep_i);                        \
            bv[j] = take ? ov : keep_v;                                                            \
            bi[j] = take ? oi : keep_i;                                                            \
        }                                                                                          \
    }
            PP_CAND_STEP(8, 16) PP_CAND_STEP(4, 8) PP_CAND_STEP(2, 4) PP_CAND_STEP(1, 2)
#undef PP_CAND_STEP
            {
                const float ov = __shfl_xor(bv[0], 1, 64);
                const int oi = __shfl_xor(bi[0], 1, 64);
                if (ov > bv[0] || (ov == bv[0] && oi < bi[0])) { bv[0] = ov; bi[0] = oi; }
            }
            if ((lane & 1) == 0) {   // the register this lane ended up with: r = bit4 * 8 + bit3 * 4 + bit2 * 2 + bit1 of its lane id
                const int r = ((lane >> 4) & 1) * 8 + ((lane >> 3) & 1) * 4 + ((lane >> 2) & 1) * 2 + ((lane >> 1) & 1);
                const int lrow_t = mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);   // row inside the group's 128
                cv[(g * 4 + wq) * 128 + lrow_t] = bv[0];
                ci[(g * 4 + wq) * 128 + lrow_t] = bi[0];
            }
        }
        __syncthreads();
        if (tid < 256) {
            const int gg = tid >> 7, lr = tid & 127, row = m0 + tid;
            float best = cv[(gg * 4) * 128 + lr];
            int idx = ci[(gg * 4) * 128 + lr];
#pragma unroll
            for (int q = 1; q < 4; ++q) {   // ascending columns: a strict compare keeps the first maximum
                const float v = cv[(gg * 4 + q) * 128 + lr];
                if (v > best) { best = v; idx = ci[(gg * 4 + q) * 128 + lr]; }
            }
            if (row < p.M) {
                p.cand_val[(size_t)row * ntiles + tn] = best;
                p.cand_idx[(size_t)row * ntiles + tn] = idx;
            }
        }
        return;
    }
    // epilogue straight from the C layout of the 32x32 MFMA (register r of block (mi, ni) = column lane & 31, row
    // (r & 3) + 8 (r >> 2) + 4 (lane >> 5)): a store instruction writes two rows x 128 contiguous bytes (fp32).  Measured
    // against two alternatives on the store-heavy shapes (c_attn / lm_head at 14 848 rows: 182 MB / 3 GB of fp32 output,
    // `profiles/r05_gemm_big_*.log`): swapped MFMA operands (a lane = 4 consecutive columns, 16-byte stores of 32-byte row
    // pieces) 142 / 3577 us, quad-transposed accumulators (16-byte stores, 8 rows x 128 B per instruction) 129 / 2377 us, this
    // form 118 / 2232 us - the tile's 256 KiB leave through the HBM write path at ~3 TB/s whatever the instruction count
    // (without the stores: 75 us), so the form with no extra VALU work in front of the first store wins.
    const int ccol = lane & 31, crow4 = 4 * (lane >> 5);
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) {
            const int col = n0 + wq * 64 + ni * 32 + ccol;
            const int colc = min(col, p.N - 1);
            const int rbase = m0 + g * 128 + mi * 32 + crow4;
            const float sh = esh[ni];
            float rv[16], vv[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) rv[r] = 0.f;
            if (p.R) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = min(rbase + (r & 3) + 8 * (r >> 2), p.M - 1);
                    rv[r] = p.R[(size_t)row * p.ldy + colc];
                }
            }
            if (p.G16) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = min(rbase + (r & 3) + 8 * (r >> 2), p.M - 1);
                    rv[r] = from16<F16>(p.G16[(size_t)row * p.ldy + colc]);
                }
#pragma unroll
                for (int r = 0; r < 16; ++r) vv[r] = (acc[mi][ni][r] + sh) * gelu_new_grad_fast(rv[r]);
            } else {
                if (p.Ypre16 && col < p.N) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int dr = (r & 3) + 8 * (r >> 2);
                        if (rbase + dr < p.M)
                            p.Ypre16[(size_t)rbase * p.ldy + col + (size_t)(dr * p.ldy)] = (u16)to16<F16>(acc[mi][ni][r] + sh + rv[r]);
                    }
                }
#pragma unroll
                for (int r = 0; r < 16; ++r) vv[r] = apply_act_fast(acc[mi][ni][r] + sh + rv[r], p.act);
            }
            if (col < p.N) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int dr = (r & 3) + 8 * (r >> 2);
                    if (rbase + dr < p.M) {
                        const size_t o = (size_t)rbase * p.ldy + col + (size_t)(dr * p.ldy);
                        if (p.Y16) p.Y16[o] = (u16)to16<F16>(vv[r]);
                        else if (p.N > 8192) __builtin_nontemporal_store(vv[r], &p.Y[o]);
                        else p.Y[o] = vv[r];
                    }
                }
            }
        }
}

__global__ __launch_bounds__(256) void f32_to_bf16_kernel(const float* __restrict__ src, u16* __restrict__ dst, size_t n, int f16) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        dst[i] = (u16)to16_rt(src[i], f16);
}

template <int BM, int BN, int THREADS>
static int launch_bf16_cfg(const GemmBf16Params& p, hipStream_t st) {
    constexpr size_t lds = (size_t)(2 * BM + 2 * BN) * LDB * sizeof(u16);
    const int mtiles = (p.M + BM - 1) / BM, ntiles = (p.N + BN - 1) / BN;
#define RGRG_W_LAUNCH(A16_, F16_) hipLaunchKernelGGL((gemm_bf16w_kernel<BM, BN, THREADS, A16_, F16_>), dim3(mtiles * ntiles), dim3(THREADS), lds, st, p, mtiles, ntiles)
    if (p.A16) { if (p.f16) RGRG_W_LAUNCH(true, true); else RGRG_W_LAUNCH(true, false); }
    else { if (p.f16) RGRG_W_LAUNCH(false, true); else RGRG_W_LAUNCH(false, false); }
#undef RGRG_W_LAUNCH
    RGRG_LAUNCH_CHECK();
    return RGRG_OK;
}

template <int BM, int BN, int THREADS>
static int bf16_attr() {
    RGRG_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_bf16w_kernel<BM, BN, THREADS, false, false>),
                                 hipFuncAttributeMaxDynamicSharedMemorySize, 81920));
    RGRG_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_bf16w_kernel<BM, BN, THREADS, true, false>),
                                 hipFuncAttributeMaxDynamicSharedMemorySize, 81920));
    RGRG_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_bf16w_kernel<BM, BN, THREADS, false, true>),
                                 hipFuncAttributeMaxDynamicSharedMemorySize, 81920));
    RGRG_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_bf16w_kernel<BM, BN, THREADS, true, true>),
                                 hipFuncAttributeMaxDynamicSharedMemorySize, 81920));
    return RGRG_OK;
}

template <int BM, int BN, int NST>
static int glds_attr() {
    constexpr int lds = NST * (BM + BN) * 128 + BM * 16;
#define RGRG_G_ATTR(...) RGRG_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_bf16_glds_kernel<BM, BN, NST, __VA_ARGS__>), hipFuncAttributeMaxDynamicSharedMemorySize, lds))
    RGRG_G_ATTR(false, false, 0); RGRG_G_ATTR(true, false, 0); RGRG_G_ATTR(false, true, 0); RGRG_G_ATTR(true, true, 0);
    RGRG_G_ATTR(false, false, 1); RGRG_G_ATTR(false, true, 1); RGRG_G_ATTR(false, false, 2); RGRG_G_ATTR(false, true, 2);
#undef RGRG_G_ATTR
    return RGRG_OK;
}
template <int BM, int BN>
static int glds_attrs() {
    int rc;
    if ((rc = glds_attr<BM, BN, 2>()) || (rc = glds_attr<BM, BN, 3>())) return rc;
    return glds_attr<BM, BN, 4>();
}

static int gemm_gm_override();
#include "gemm_kp.inc"

constexpr int PP_LDS = 2 * (256 + 256) * 128;   // two stages of the 256 x 256 ping-pong kernel

int init_gemm_bf16_attrs() {
    static bool done = false;  // hipFuncSetAttribute is not capturable and not free: once per process
    if (done) return RGRG_OK;
    int rc;
    if ((rc = bf16_attr<128, 128, 512>()) || (rc = bf16_attr<64, 64, 256>())) return rc;
    if ((rc = glds_attrs<128, 128>()) || (rc = glds_attrs<64, 64>()) || (rc = glds_attrs<128, 64>()) || (rc = glds_attrs<64, 128>())) return rc;
    RGRG_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_bf16_pp_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, PP_LDS));
    RGRG_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_bf16_pp_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, PP_LDS));
    if ((rc = kp_attr<128, 128, 2>()) || (rc = kp_attr<64, 64, 4>()) || (rc = kp_attr<64, 64, 3>()) || (rc = kp_attr<128, 64, 3>()) || (rc = kp_attr<64, 128, 3>()) || (rc = kp_attr<64, 64, 2>()) || (rc = pr_attrs())) return rc;
    done = true;
    return RGRG_OK;
}

// RGRG_GEMM_GM: row tiles per group of the tile order (GemmBf16Params::gm); unset = 8 when there are more than 16 row tiles,
// -1 = column-major order always (the round-3/4 order; A/B runs of tools/gemm_bf16_bench.py)
static int gemm_gm_override() {
    static const int v = [] { const char* e = getenv("RGRG_GEMM_GM"); return e ? atoi(e) : 0; }();
    return v;
}

template <int BM, int BN, int NST>
static int launch_glds_cfg(const GemmBf16Params& p0, hipStream_t st) {
    GemmBf16Params p = p0;
    const int mtiles = (p.M + BM - 1) / BM, ntiles = (p.N + BN - 1) / BN;
    if (p.gm == 0) {
        const int o = gemm_gm_override();
        p.gm = o < 0 ? 0 : o > 0 ? o : (mtiles > 16 ? 8 : 0);
    }
    const int ks = (!p.cCin && p.ksplit > 1 && p.sk_ws && p.sk_cnt && (p.K / 64) % p.ksplit == 0 && p.K / 64 / p.ksplit >= NST - 1) ? p.ksplit : 1;
    p.ksplit = ks;
#define RGRG_G_LAUNCH(CONV_, F16_, LNF_) hipLaunchKernelGGL((gemm_bf16_glds_kernel<BM, BN, NST, CONV_, F16_, LNF_>), dim3(mtiles * ntiles * ks), dim3(256), NST * (BM + BN) * 128 + BM * 16, st, p, mtiles, ntiles)
    if (p.cCin) { if (p.f16) RGRG_G_LAUNCH(true, true, 0); else RGRG_G_LAUNCH(true, false, 0); }
    else if (p.Yb16) { if (p.f16) RGRG_G_LAUNCH(false, true, 1); else RGRG_G_LAUNCH(false, false, 1); }
    else if (p.ln_colsum) { if (p.f16) RGRG_G_LAUNCH(false, true, 2); else RGRG_G_LAUNCH(false, false, 2); }
    else { if (p.f16) RGRG_G_LAUNCH(false, true, 0); else RGRG_G_LAUNCH(false, false, 0); }
#undef RGRG_G_LAUNCH
    RGRG_LAUNCH_CHECK();
    return RGRG_OK;
}
template <int BM, int BN>
static int launch_glds_nst(const GemmBf16Params& p, int nst, hipStream_t st) {
    if (nst == 2) return launch_glds_cfg<BM, BN, 2>(p, st);
    if (nst == 3) return launch_glds_cfg<BM, BN, 3>(p, st);
    return launch_glds_cfg<BM, BN, 4>(p, st);
}

// 256 x 256 ping-pong kernel: plain GEMMs only (no LayerNorm fold, no convolution, no 16-bit residual), K tiles >= 4
static bool pp_eligible(const GemmBf16Params& p) {
    return !p.cCin && !p.Yb16 && !p.ln_colsum && !p.R16 && p.K >= 256 && p.K % 64 == 0;
}
static int launch_pp(const GemmBf16Params& p0, hipStream_t st) {
    GemmBf16Params p = p0;
    const int mtiles = (p.M + 255) / 256, ntiles = (p.N + 255) / 256;
    if (p.gm == 0) {
        const int o = gemm_gm_override();
        p.gm = o < 0 ? 0 : o > 0 ? o : (mtiles > 8 ? 8 : 0);
    }
    { const char* e = getenv("RGRG_PP_DBG"); p.dbg = e ? atoi(e) : 0; }
    if (p.f16) hipLaunchKernelGGL(gemm_bf16_pp_kernel<true>, dim3(mtiles * ntiles), dim3(512), PP_LDS, st, p, mtiles, ntiles);
    else hipLaunchKernelGGL(gemm_bf16_pp_kernel<false>, dim3(mtiles * ntiles), dim3(512), PP_LDS, st, p, mtiles, ntiles);
    RGRG_LAUNCH_CHECK();
    return RGRG_OK;
}
// RGRG_GEMM_PP=0: never pick the ping-pong kernel (A/B runs)
static bool pp_enabled() {
    static const bool v = [] { const char* e = getenv("RGRG_GEMM_PP"); return !e || atoi(e) != 0; }();
    return v;
}

static bool pp_lm_head() {   // RGRG_GEMM_PP_LMHEAD=0: keep the decode lm_head on the 128 x 128 kernel (A/B)
    static const bool v = [] { const char* e = getenv("RGRG_GEMM_PP_LMHEAD"); return !e || atoi(e) != 0; }();
    return v;
}

// tile = shape + 16 * stages; shape: 0 = heuristic, 1 = 128x128, 2 = 64x64, 3 = 128x64, 4 = 64x128, 5 = 256x256 ping-pong;
// stages: 0 (= 4), 2, 3, 4
// (tools/gemm_bf16_bench.py measures them).  Heuristic from the COLD-weights table at M = 923
// (profiles/r03_gemm_bf16_bench_v7_tiles_x_stages_cold.log: a decode step streams 0.7 GB of weights between two uses of a
// matrix, so a GEMM never finds its W in a cache; a warm-cache bench flatters the short pipelines by 20-40 %).  What
// matters is how many workgroups a CU can overlap and that the grid is close to a whole number of such rounds:
//   lm_head  (3144 tiles of 128^2)  128 x 128, 2 stages: two workgroups per CU           173 us (4 stages: 230)
//   c_fc     (N 4096, K 1024)       128 x 64,  3 stages: 512 workgroups, two per CU        18.1 us (64^2: 19.5, 128^2: 20.6)
//   c_attn   (N 3072, K 1024)       64 x 64,   3 stages                                    15.7 us (128^2: 18.6)
//   attn_proj (N 1024, K 1024)      64 x 64,   4 stages                                     8.1 us (2 stages: 14.4)
//   mlp_proj (N 1024, K 4096)       64 x 64,   4 stages (long K: depth pays)               20.7 us (2 stages: 47)
// shapes 6 .. 10: the K-parity ping-pong kernel (gemm_kp.inc): 6 = 128x128x2, 7 = 64x64x4, 8 = 128x64x3, 9 = 64x128x3,
// 10 = 64x64x3, 11 = its (N, K) heuristic; 12 = the 128 x 128 row-split ping-pong kernel (gemm_kp.inc)
static int launch_glds(const GemmBf16Params& p, int tile, hipStream_t st) {
    int shape = tile & 15, nst = tile >> 4;
    switch (shape) {
        case 6: return launch_kp_cfg<128, 128, 2>(p, st);
        case 7: return launch_kp_cfg<64, 64, 4>(p, st);
        case 8: return launch_kp_cfg<128, 64, 3>(p, st);
        case 9: return launch_kp_cfg<64, 128, 3>(p, st);
        case 10: return launch_kp_cfg<64, 64, 3>(p, st);
        case 11: return launch_kp_auto(p, st);
        case 12: return launch_pr(p, st);
        case 13: return launch_kp_cfg<64, 64, 2>(p, st);
    }
    if (shape == 5) {
        if (!pp_eligible(p)) { set_error("bf16 GEMM: the 256 x 256 kernel does not take this variant"); return RGRG_EINVAL; }
        return launch_pp(p, st);
    }
    if (shape == 0 && pp_enabled() && pp_eligible(p)) {
        // at least ~3/4 of a round of one 256 x 256 workgroup per CU, and enough rows that the 128 x 128 kernel is the
        // alternative (the M = 923 decode shapes keep their tuned tiles)
        const long tiles_pp = (long)((p.M + 255) / 256) * ((p.N + 255) / 256);
        if (p.M >= 2048 && tiles_pp >= 192) return launch_pp(p, st);
        // the vocabulary-sized lm_head of a many-sequence decode step (923 rows x 50 257 columns: 4 x 197 tiles): 160 us against
        // 173 us on 128 x 128 tiles (profiles/r05_gemm_big_fc6_v1.log)
        if (p.M >= 512 && p.N >= 32768 && tiles_pp >= 512 && pp_lm_head()) return launch_pp(p, st);
    }
    if (shape == 0 && p.ln_colsum && p.M <= 320) {   // measurement: RGRG_CONS_TILE_SMALLM = tile code of c_attn / c_fc on a row range (M <= 320)
        static const int o = [] { const char* e = getenv("RGRG_CONS_TILE_SMALLM"); return e ? atoi(e) : 0; }();
        if (o) { shape = o & 15; nst = o >> 4; }
    }
    if (shape == 0) {
        const long tiles_big = (long)((p.M + 127) / 128) * ((p.N + 127) / 128);
        if (p.N <= 64) { shape = 3; nst = 2; }            // a 64-channel conv: no point in a 128-wide column tile
        else if (tiles_big >= 512) { shape = 1; nst = 2; }   // >= one round of two workgroups per CU (round 5: was 1024 - the
                                                              // M = 14 848 training GEMMs with N = 1024 ran on 64 x 64 tiles)
        else if (p.K >= 8192 && tiles_big >= 192) { shape = 1; nst = 2; }   // fc6 at 8 images (6 650 x 1024 x 131 072): 2.1 ms, 64 x 64: 4.2
        else if (p.K > 1024) { shape = 2; nst = 4; }
        else if (tiles_big >= 256) { shape = 3; nst = 3; }
        else if (tiles_big >= 128) { shape = 2; nst = 3; }
        else { shape = 2; nst = 4; }
    }
    if (nst == 0) nst = 4;
    if (p.K / 64 < nst - 1) nst = 2;
    switch (shape) {
        case 1: return launch_glds_nst<128, 128>(p, nst, st);
        case 2: return launch_glds_nst<64, 64>(p, nst, st);
        case 3: return launch_glds_nst<128, 64>(p, nst, st);
        case 4: return launch_glds_nst<64, 128>(p, nst, st);
    }
    set_error("bf16 GEMM: unknown tile configuration %d", tile);
    return RGRG_EINVAL;
}

// The greedy lm_head of the many-sequence decode step: is this the shape launch_glds sends to the 256 x 256 kernel (whose
// epilogue can leave arg-max candidates instead of logits)?  RGRG_LMHEAD_CAND=0: keep the logits + candidates pass (A/B runs).
bool gemm_bf16_cand_epilogue_ok(int M, int N, int K) {
    static const bool on = [] { const char* e = getenv("RGRG_LMHEAD_CAND"); return !e || atoi(e) != 0; }();
    const long tiles_pp = (long)((M + 255) / 256) * ((N + 255) / 256);
    return on && pp_enabled() && pp_lm_head() && M >= 512 && N >= 32768 && tiles_pp >= 512 && K >= 256 && K % 64 == 0;
}

// A16 / Y16 (either may be null): bf16 activations in / out, see GemmBf16Params
int launch_gemm_bf16w_ex(const float* A, const void* A16, const void* Wb, const float* shift, const float* R, float* Y, void* Y16,
                         int M, int N, int K, int ldy, int act, hipStream_t st, int f16, const GemmLnFold* ln) {
    RGRG_CHECK_ARG((A || A16) && Wb && (Y || Y16 || (ln && ln->cand_val)) && M > 0 && N > 0 && K > 0 && K % (4 * BK16) == 0 && ldy >= N);  // 4 = depth NS
    GemmBf16Params p{A, reinterpret_cast<const u16*>(A16), reinterpret_cast<const u16*>(Wb), shift, R, Y,
                     reinterpret_cast<u16*>(Y16), M, N, K, ldy, act};
    p.f16 = f16 ? 1 : 0;
    if (ln && (ln->Ypre16 || ln->G16)) {   // training-pass epilogues (GemmBf16Params): LDS-DMA kernel, plain variant only
        RGRG_CHECK_ARG(A16 && (size_t)128 * K * 2 < ((size_t)1 << 31) && !ln->Yb16 && !ln->ln_colsum);
        RGRG_CHECK_ARG(!(ln->Ypre16 && ln->G16) && (!ln->G16 || (!R && act == RGRG_ACT_NONE)));
        p.Ypre16 = reinterpret_cast<u16*>(ln->Ypre16); p.G16 = reinterpret_cast<const u16*>(ln->G16);
    } else if (ln && (ln->Yb16 || ln->ln_colsum)) {   // LayerNorm folded around the GEMM (GemmBf16Params): LDS-DMA kernel only
        RGRG_CHECK_ARG(A16 && (size_t)128 * K * 2 < ((size_t)1 << 31));
        RGRG_CHECK_ARG((ln->Yb16 != nullptr) != (ln->ln_colsum != nullptr));
        RGRG_CHECK_ARG(!ln->Yb16 || (Y && ln->stats_out && N == 1024));
        RGRG_CHECK_ARG(!ln->ln_colsum || (ln->ln_stats && K == 1024 && !R));
        p.Yb16 = reinterpret_cast<u16*>(ln->Yb16); p.stats_out = ln->stats_out;
        p.ln_stats = ln->ln_stats; p.ln_colsum = ln->ln_colsum;
    }
    if (ln && ln->ksplit > 1) { p.ksplit = ln->ksplit; p.sk_ws = ln->sk_ws; p.sk_cnt = ln->sk_cnt; }   // split-K (LDS-DMA kernel)
    if (ln && ln->cand_val) {   // arg-max candidates instead of logits: 256 x 256 ping-pong kernel only
        RGRG_CHECK_ARG(A16 && ln->cand_idx && !R && act == RGRG_ACT_NONE && !Y16 && !ln->Yb16 && !ln->ln_colsum && !ln->Ypre16 && !ln->G16);
        p.cand_val = ln->cand_val; p.cand_idx = ln->cand_idx;
        return launch_glds(p, 5, st);
    }
    if (ln && ln->kp && A16 && (size_t)128 * K * 2 < ((size_t)1 << 31) && kp_eligible(p, 4)) return launch_glds(p, 11, st);
    if (A16 && (size_t)128 * K * 2 < ((size_t)1 << 31)) return launch_glds(p, 0, st);  // both operands bf16: LDS-DMA kernel
    // fp32 activations (rounded to bf16 while they are staged through registers)
    const long tiles_big = (long)((M + 127) / 128) * ((N + 127) / 128);
    return tiles_big >= 192 ? launch_bf16_cfg<128, 128, 512>(p, st) : launch_bf16_cfg<64, 64, 256>(p, st);
}

int launch_gemm_bf16w(const float* A, const void* Wb, const float* shift, const float* R, float* Y, int M, int N, int K,
                      int ldy, int act, hipStream_t st, int f16) {
    return launch_gemm_bf16w_ex(A, nullptr, Wb, shift, R, Y, nullptr, M, N, K, ldy, act, st, f16, nullptr);
}

__global__ __launch_bounds__(256) void bf16_to_f32_kernel(const u16* __restrict__ src, float* __restrict__ dst, size_t n, int f16) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        dst[i] = from16_rt(src[i], f16);
}

int convert_f32_to_bf16(const float* src, void* dst, size_t n, hipStream_t st, int f16) {
    const int blocks = (int)((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096);
    hipLaunchKernelGGL(f32_to_bf16_kernel, dim3(blocks), dim3(256), 0, st, src, reinterpret_cast<u16*>(dst), n, f16 ? 1 : 0);
    RGRG_LAUNCH_CHECK();
    return RGRG_OK;
}

}  // namespace rgrg

using namespace rgrg;

extern "C" int rgrg_f32_to_bf16(const float* src, uint16_t* dst, int64_t n, int fp16, void* stream) {
    RGRG_CHECK_ARG(src && dst && n > 0);
    return convert_f32_to_bf16(src, dst, (size_t)n, as_stream(stream), fp16);
}

extern "C" int rgrg_bf16_to_f32(const uint16_t* src, float* dst, int64_t n, int fp16, void* stream) {
    RGRG_CHECK_ARG(src && dst && n > 0);
    const int blocks = (int)(((size_t)n + 255) / 256 < 4096 ? ((size_t)n + 255) / 256 : 4096);
    hipLaunchKernelGGL(bf16_to_f32_kernel, dim3(blocks), dim3(256), 0, as_stream(stream), src, dst, (size_t)n, fp16 ? 1 : 0);
    RGRG_LAUNCH_CHECK();
    return RGRG_OK;
}

extern "C" int rgrg_linear_bf16w_f32(const float* A, const uint16_t* Wb, const float* shift, const float* R, float* Y,
                                     int M, int N, int K, int ldy, int act, int fp16, void* stream) {
    int rc = init_gemm_bf16_attrs();
    if (rc) return rc;
    return launch_gemm_bf16w(A, Wb, shift, R, Y, M, N, K, ldy, act, as_stream(stream), fp16);
}

extern "C" int rgrg_linear_bf16_f32(const uint16_t* A16, const uint16_t* Wb, const float* shift, const float* R, float* Y,
                                    uint16_t* Y16, int M, int N, int K, int ldy, int act, int fp16, void* stream) {
    int rc = init_gemm_bf16_attrs();
    if (rc) return rc;
    RGRG_CHECK_ARG(A16 && ((Y != nullptr) != (Y16 != nullptr)));
    return launch_gemm_bf16w_ex(nullptr, A16, Wb, shift, R, Y, Y16, M, N, K, ldy, act, as_stream(stream), fp16, nullptr);
}

extern "C" int rgrg_debug_linear_bf16_tile(const uint16_t* A16, const uint16_t* Wb, const float* shift, const float* R, float* Y,
                                           int M, int N, int K, int ldy, int act, int tile, int lda, int ldw, int fp16, void* stream) {
    int rc = init_gemm_bf16_attrs();
    if (rc) return rc;
    RGRG_CHECK_ARG(A16 && Wb && Y && M > 0 && N > 0 && K > 0 && K % 256 == 0 && ldy >= N && (size_t)128 * K * 2 < ((size_t)1 << 31));
    RGRG_CHECK_ARG((lda == 0 || lda >= K) && (ldw == 0 || ldw >= K));
    GemmBf16Params p{nullptr, A16, Wb, shift, R, Y, nullptr, M, N, K, ldy, act, lda, ldw};
    p.f16 = fp16 ? 1 : 0;
    return launch_glds(p, tile, as_stream(stream));
}

// Test / measurement hook for the training-pass epilogues (GemmBf16Params::Ypre16 / G16) and the 16-bit output on a forced
// tile (tile as in rgrg_debug_linear_bf16_tile; 5 = the 256 x 256 ping-pong kernel).
extern "C" int rgrg_debug_linear_bf16_train(const uint16_t* A16, const uint16_t* Wb, const float* shift, const float* R, float* Y,
                                            uint16_t* Y16, uint16_t* Ypre16, const uint16_t* G16, int M, int N, int K, int ldy,
                                            int act, int tile, int fp16, void* stream) {
    int rc = init_gemm_bf16_attrs();
    if (rc) return rc;
    RGRG_CHECK_ARG(A16 && Wb && ((Y != nullptr) != (Y16 != nullptr)) && M > 0 && N > 0 && K > 0 && K % 256 == 0 && ldy >= N);
    RGRG_CHECK_ARG((size_t)128 * K * 2 < ((size_t)1 << 31) && !(Ypre16 && G16) && (!G16 || (!R && act == RGRG_ACT_NONE)));
    GemmBf16Params p{nullptr, A16, Wb, shift, R, Y, Y16, M, N, K, ldy, act};
    p.f16 = fp16 ? 1 : 0;
    p.Ypre16 = Ypre16; p.G16 = G16;
    return launch_glds(p, tile, as_stream(stream));
}

// Test hook for the arg-max epilogue of the 256 x 256 kernel (the greedy lm_head of the many-sequence decode step):
// cand_val / cand_idx [M][ceil(N / 256)] = per row and 256-column tile the maximum of A16 Wb^T + shift and its column.
extern "C" int rgrg_debug_linear_bf16_argmax(const uint16_t* A16, const uint16_t* Wb, const float* shift, int M, int N, int K,
                                             float* cand_val, int* cand_idx, int fp16, void* stream) {
    int rc = init_gemm_bf16_attrs();
    if (rc) return rc;
    RGRG_CHECK_ARG(A16 && Wb && cand_val && cand_idx && M > 0 && N > 0 && K >= 256 && K % 256 == 0);
    GemmLnFold f{};
    f.cand_val = cand_val; f.cand_idx = cand_idx;
    return launch_gemm_bf16w_ex(nullptr, A16, Wb, shift, nullptr, nullptr, nullptr, M, N, K, N, RGRG_ACT_NONE, as_stream(stream), fp16, &f);
}

// Test hook for the LayerNorm-folded variants of the LDS-DMA kernel (the decoder uses them internally, decoder.hip
// enqueue_step): producer when Yb16 / stats_out are given (N == 1024), consumer when ln_stats / ln_colsum are (K == 1024).
extern "C" int rgrg_debug_linear_bf16_ln(const uint16_t* A16, const uint16_t* Wb, const float* shift, const float* R, float* Y,
                                         uint16_t* Yb16, float* stats_out, const float* ln_stats, const float* ln_colsum, int M,
                                         int N, int K, int ldy, int act, int fp16, void* stream) {
    int rc = init_gemm_bf16_attrs();
    if (rc) return rc;
    RGRG_CHECK_ARG(A16 && Wb && Y);
    GemmLnFold f{};
    f.Yb16 = Yb16; f.stats_out = stats_out; f.ln_stats = ln_stats; f.ln_colsum = ln_colsum;
    return launch_gemm_bf16w_ex(nullptr, A16, Wb, shift, R, Y, nullptr, M, N, K, ldy, act, as_stream(stream), fp16, &f);
}

// The same on the K-parity ping-pong kernel (gemm_kp.inc), which the decoder selects for the many-sequence step: kp != 0.
extern "C" int rgrg_debug_linear_bf16_ln_kp(const uint16_t* A16, const uint16_t* Wb, const float* shift, const float* R, float* Y,
                                            uint16_t* Y16, uint16_t* Yb16, float* stats_out, const float* ln_stats, const float* ln_colsum,
                                            int M, int N, int K, int ldy, int act, int fp16, int kp, void* stream) {
    int rc = init_gemm_bf16_attrs();
    if (rc) return rc;
    RGRG_CHECK_ARG(A16 && Wb && ((Y != nullptr) != (Y16 != nullptr)));
    GemmLnFold f{};
    f.Yb16 = Yb16; f.stats_out = stats_out; f.ln_stats = ln_stats; f.ln_colsum = ln_colsum; f.kp = kp;
    return launch_gemm_bf16w_ex(nullptr, A16, Wb, shift, R, Y, Y16, M, N, K, ldy, act, as_stream(stream), fp16, &f);
}

// nn.Conv2d (+ folded eval BatchNorm + residual + ReLU) as an implicit GEMM on the bf16 matrix core, for the detector
// under torch.autocast (the reference runs trunk / RPN in half precision there, generate_reports_for_images.py:108).
//   X16 [B,H,W,Cin] bf16 NHWC with 128 zero elements in front of it (X16[-128 .. -1] == 0), Cin % 64 == 0
//   Wb  [Cout][KH][KW][Cin] bf16 (BatchNorm scale folded in), shift [Cout] f32 (bias / folded BatchNorm shift) or NULL
//   R16 [B,OH,OW,Cout] bf16 or NULL; exactly one of Y (f32) / Y16 (bf16) [B,OH,OW,Cout]
extern "C" int rgrg_conv2d_nhwc_bf16(const uint16_t* X16, const uint16_t* Wb, const float* shift, const uint16_t* R16, float* Y,
                                     uint16_t* Y16, int B, int H, int Wd, int Cin, int Cout, int KH, int KW, int stride, int pad,
                                     int act, int fp16, void* stream) {
    int rc = init_gemm_bf16_attrs();
    if (rc) return rc;
    RGRG_CHECK_ARG(X16 && Wb && ((Y != nullptr) != (Y16 != nullptr)) && B > 0 && H > 0 && Wd > 0 && Cin > 0 && Cin % 64 == 0);
    RGRG_CHECK_ARG(Cout > 0 && KH > 0 && KW > 0 && stride > 0 && pad >= 0);
    RGRG_CHECK_ARG((size_t)B * H * Wd * Cin * 2 + 256 < ((size_t)1 << 31));   // 32-bit byte offsets into the image
    GemmBf16Params p{};
    p.A16 = X16; p.Wb = Wb; p.shift = shift; p.R16 = R16; p.Y = Y; p.Y16 = Y16; p.f16 = fp16 ? 1 : 0;
    p.cH = H; p.cW = Wd; p.cCin = Cin; p.cKH = KH; p.cKW = KW; p.cStride = stride; p.cPad = pad;
    p.cOH = (H + 2 * pad - KH) / stride + 1;
    p.cOW = (Wd + 2 * pad - KW) / stride + 1;
    RGRG_CHECK_ARG(p.cOH > 0 && p.cOW > 0);
    p.M = B * p.cOH * p.cOW; p.N = Cout; p.K = KH * KW * Cin; p.ldy = Cout; p.act = act;
    RGRG_CHECK_ARG((size_t)128 * p.K * 2 < ((size_t)1 << 31));
    return launch_glds(p, 0, as_stream(stream));
}
