// Detector-side kernels that are not GEMMs: ResNet stem, max-pool, RPN proposal
// selection (top-k + decode + NMS), RoIAlign fused with the 8x8 average pool, the
// reference's top-1-box-per-class post-processing and the region-selection mask.
//
// All box / sampling arithmetic is compiled with FP contraction OFF and follows the
// operation order of the CPU algorithms it replaces, so that given identical inputs
// the results are bit-identical except where a transcendental (exp) is involved.
#include "common.h"
#include <cfloat>
#include <cmath>

#pragma clang fp contract(off)

namespace rgrg {

// ----------------------------------------------------------------------------------
// ResNet stem: 7x7 stride-2 pad-3 conv on ONE input channel -> 64, + BN + ReLU.
// One 16x16 output tile per workgroup; the 37x37 input patch and the 49x64 filter
// bank sit in LDS; a lane owns one output pixel and sweeps 16 channels at a time.
// ----------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void stem_conv7x7_kernel(const float* __restrict__ X, const float* __restrict__ Wt,
                                                           const float* __restrict__ scale,
                                                           const float* __restrict__ shift, float* __restrict__ Y,
                                                           int H, int Wd, int OH, int OW) {
    __shared__ float tile[37][38];
    __shared__ __attribute__((aligned(16))) float w[49 * 64];
    const int tid = threadIdx.x, b = blockIdx.z;
    const int oy0 = blockIdx.y * 16, ox0 = blockIdx.x * 16;
    for (int i = tid; i < 49 * 64; i += 256) w[i] = Wt[i];
    for (int i = tid; i < 37 * 37; i += 256) {
        const int r = i / 37, c = i - r * 37;
        const int ih = oy0 * 2 - 3 + r, iw = ox0 * 2 - 3 + c;
        float v = 0.f;
        if ((unsigned)ih < (unsigned)H && (unsigned)iw < (unsigned)Wd) v = X[((size_t)b * H + ih) * Wd + iw];
        tile[r][c] = v;
    }
    __syncthreads();
    const int ty = tid >> 4, tx = tid & 15;
    const int oy = oy0 + ty, ox = ox0 + tx;
    float* out = Y + (((size_t)b * OH + oy) * OW + ox) * 64;
    for (int cg = 0; cg < 4; ++cg) {
        float acc[16];
#pragma unroll
        for (int c = 0; c < 16; ++c) acc[c] = 0.f;
        for (int kh = 0; kh < 7; ++kh)
#pragma unroll
            for (int kw = 0; kw < 7; ++kw) {
                const float x = tile[ty * 2 + kh][tx * 2 + kw];
                const f32x4* wp = reinterpret_cast<const f32x4*>(&w[(kh * 7 + kw) * 64 + cg * 16]);
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const f32x4 wv = wp[q];
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc[q * 4 + e] = __builtin_fmaf(x, wv[e], acc[q * 4 + e]);
                }
            }
        if (oy < OH && ox < OW) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                f32x4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int c = cg * 16 + q * 4 + e;
                    const float v = acc[q * 4 + e] * scale[c] + shift[c];
                    o[e] = v > 0.f ? v : 0.f;
                }
                *reinterpret_cast<f32x4*>(out + cg * 16 + q * 4) = o;
            }
        }
    }
}

// nn.MaxPool2d(3, stride 2, padding 1), NHWC; one thread per (pixel, 4 channels).
__global__ __launch_bounds__(256) void maxpool3x3s2_kernel(const float* __restrict__ X, float* __restrict__ Y, int B,
                                                           int H, int Wd, int C, int OH, int OW) {
    const int C4 = C >> 2;
    const size_t total = (size_t)B * OH * OW * C4;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int c4 = (int)(i % C4);
        size_t t = i / C4;
        const int ox = (int)(t % OW);
        t /= OW;
        const int oy = (int)(t % OH);
        const int b = (int)(t / OH);
        f32x4 m = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
#pragma unroll
        for (int kh = 0; kh < 3; ++kh) {
            const int ih = oy * 2 - 1 + kh;
            if ((unsigned)ih >= (unsigned)H) continue;
#pragma unroll
            for (int kw = 0; kw < 3; ++kw) {
                const int iw = ox * 2 - 1 + kw;
                if ((unsigned)iw >= (unsigned)Wd) continue;
                const f32x4 v = *reinterpret_cast<const f32x4*>(X + (((size_t)b * H + ih) * Wd + iw) * C + c4 * 4);
#pragma unroll
                for (int e = 0; e < 4; ++e) m[e] = fmaxf(m[e], v[e]);
            }
        }
        *reinterpret_cast<f32x4*>(Y + (((size_t)b * OH + oy) * OW + ox) * C + c4 * 4) = m;
    }
}

// ----------------------------------------------------------------------------------
// RPN proposals.  One 1024-thread workgroup per image:
//   1. radix-select the pre_nms-th largest 48-bit composite key
//      (order-preserving logit bits << 16 | (0xFFFF - index)): unique keys, so ties
//      between equal logits resolve to the LOWER anchor index, like a stable sort;
//   2. bitonic-sort the <=1024 survivors descending in LDS;
//   3. decode (BoxCoder weights 1,1,1,1; dw/dh clamped at log(1000/16)), clip, drop
//      boxes smaller than min_size, keep order;
//   4. 64-bit suppression bitmask (IoU > thr) in LDS, greedy scan by one wave.
// ----------------------------------------------------------------------------------
constexpr int PROP_THREADS = 1024;
constexpr int PROP_MAX = 1024;   // sort width
constexpr int PROP_ROWS = 1000;  // max boxes entering NMS (LDS budget: 160 KiB total)

struct PropSmem {
    unsigned long long mask[PROP_ROWS][16];  // 125 KiB
    unsigned long long keys[PROP_MAX];      // 8 KiB
    float box[PROP_MAX][4];                 // 16 KiB
    float area[PROP_MAX];                   // 4 KiB
    int keep[PROP_MAX];                     // 4 KiB
    unsigned hist[256];
    int wave_tot[16];
    unsigned long long prefix;
    int kleft, cnt, nvalid, nkeep;
};

static_assert(sizeof(PropSmem) <= 160 * 1024, "PropSmem must fit the 160 KiB LDS of a CU");

__device__ __forceinline__ unsigned order_key(float f) {
    const unsigned u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

__global__ __launch_bounds__(PROP_THREADS) void rpn_proposals_kernel(
    const float* __restrict__ head_out, const float* __restrict__ anchors, float* __restrict__ proposals,
    int* __restrict__ counts, int HW, int A, int pre_nms, int post_nms, float nms_thresh, float min_size, float img_w,
    float img_h) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    PropSmem& s = *reinterpret_cast<PropSmem*>(smem_raw);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int b = blockIdx.x;
    const int n_total = HW * A, ld = 5 * A;
    const float* ho = head_out + (size_t)b * HW * ld;
    const int k_sel = pre_nms < n_total ? pre_nms : n_total;

    auto composite = [&](int idx) -> unsigned long long {
        const int loc = idx / A, a = idx - loc * A;
        const unsigned k = order_key(ho[(size_t)loc * ld + a]);
        return ((unsigned long long)k << 16) | (unsigned long long)(0xFFFF - idx);
    };

    // 1. radix select, 6 passes of 8 bits, most significant first
    if (tid == 0) { s.prefix = 0ull; s.kleft = k_sel; s.cnt = 0; }
    for (int pass = 5; pass >= 0; --pass) {
        if (tid < 256) s.hist[tid] = 0u;
        __syncthreads();
        const unsigned long long pre = s.prefix;
        const int sh = 8 * pass;
        for (int idx = tid; idx < n_total; idx += PROP_THREADS) {
            const unsigned long long c = composite(idx);
            if (pass == 5 || (c >> (sh + 8)) == pre) atomicAdd(&s.hist[(unsigned)(c >> sh) & 255u], 1u);
        }
        __syncthreads();
        if (tid == 0) {
            int cum = 0, k = s.kleft, bsel = 0;
            for (int bin = 255; bin >= 0; --bin) {
                const int h = (int)s.hist[bin];
                if (cum + h >= k) { bsel = bin; k -= cum; break; }
                cum += h;
            }
            s.kleft = k;
            s.prefix = (pre << 8) | (unsigned long long)bsel;
        }
        __syncthreads();
    }
    const unsigned long long thr_key = s.prefix;
    s.keys[tid] = 0ull;
    __syncthreads();
    for (int idx = tid; idx < n_total; idx += PROP_THREADS) {
        const unsigned long long c = composite(idx);
        if (c >= thr_key) {
            const int pos = atomicAdd(&s.cnt, 1);
            if (pos < PROP_MAX) s.keys[pos] = c;
        }
    }
    __syncthreads();
    const int cnt = s.cnt < PROP_MAX ? s.cnt : PROP_MAX;

    // 2. bitonic sort, descending
    for (int k = 2; k <= PROP_MAX; k <<= 1)
        for (int j = k >> 1; j > 0; j >>= 1) {
            const int ixj = tid ^ j;
            if (ixj > tid) {
                const unsigned long long x = s.keys[tid], y = s.keys[ixj];
                const bool desc = (tid & k) == 0;
                if ((x < y) == desc) { s.keys[tid] = y; s.keys[ixj] = x; }
            }
            __syncthreads();
        }

    // 3. decode + clip + small-box filter, order-preserving compaction
    float bx[4] = {0.f, 0.f, 0.f, 0.f};
    bool valid = false;
    if (tid < cnt) {
        const int idx = 0xFFFF - (int)(s.keys[tid] & 0xFFFFull);
        const int loc = idx / A, a = idx - loc * A;
        const float* d = ho + (size_t)loc * ld + A + a * 4;
        const float* an = anchors + (size_t)idx * 4;
        const float aw = an[2] - an[0], ah = an[3] - an[1];
        const float cx = an[0] + 0.5f * aw, cy = an[1] + 0.5f * ah;
        const float clipv = 4.135166556742356f;  // log(1000/16)
        const float dw = fminf(d[2], clipv), dh = fminf(d[3], clipv);
        const float pcx = d[0] * aw + cx, pcy = d[1] * ah + cy;
        const float pw = expf(dw) * aw, ph = expf(dh) * ah;
        const float hw_ = 0.5f * pw, hh_ = 0.5f * ph;
        bx[0] = fminf(fmaxf(pcx - hw_, 0.f), img_w);
        bx[1] = fminf(fmaxf(pcy - hh_, 0.f), img_h);
        bx[2] = fminf(fmaxf(pcx + hw_, 0.f), img_w);
        bx[3] = fminf(fmaxf(pcy + hh_, 0.f), img_h);
        valid = (bx[2] - bx[0] >= min_size) && (bx[3] - bx[1] >= min_size);
    }
    {
        const unsigned long long bal = __ballot(valid);
        const int wpre = __popcll(bal & ((1ull << lane) - 1ull));
        if (lane == 0) s.wave_tot[wave] = __popcll(bal);
        __syncthreads();
        int off = 0, tot = 0;
        for (int w2 = 0; w2 < 16; ++w2) {
            if (w2 < wave) off += s.wave_tot[w2];
            tot += s.wave_tot[w2];
        }
        if (valid) {
            const int pos = off + wpre;
            s.box[pos][0] = bx[0]; s.box[pos][1] = bx[1]; s.box[pos][2] = bx[2]; s.box[pos][3] = bx[3];
            s.area[pos] = (bx[2] - bx[0]) * (bx[3] - bx[1]);
        }
        if (tid == 0) s.nvalid = tot;
        __syncthreads();
    }
    const int nv = s.nvalid;

    // 4a. suppression bitmask: bit j of mask[i][w] set iff j > i and IoU(i,j) > thr
    for (int item = tid; item < nv * 16; item += PROP_THREADS) {
        const int i = item >> 4, w2 = item & 15;
        unsigned long long bits = 0ull;
        const int j0 = w2 * 64;
        if (j0 + 63 > i) {
            const float x1 = s.box[i][0], y1 = s.box[i][1], x2 = s.box[i][2], y2 = s.box[i][3], ai = s.area[i];
            for (int jj = 0; jj < 64; ++jj) {
                const int j = j0 + jj;
                if (j <= i || j >= nv) continue;
                const float xx1 = fmaxf(x1, s.box[j][0]), yy1 = fmaxf(y1, s.box[j][1]);
                const float xx2 = fminf(x2, s.box[j][2]), yy2 = fminf(y2, s.box[j][3]);
                const float iw = fmaxf(0.f, xx2 - xx1), ih = fmaxf(0.f, yy2 - yy1);
                const float inter = iw * ih;
                const float ovr = inter / (ai + s.area[j] - inter);
                if (ovr > nms_thresh) bits |= 1ull << jj;
            }
        }
        s.mask[i][w2] = bits;
    }
    __syncthreads();

    // 4b. greedy scan by wave 0 (lane w < 16 owns suppression word w)
    if (wave == 0) {
        unsigned long long removed = 0ull;
        int nkeep = 0;
        for (int i = 0; i < nv && nkeep < post_nms; ++i) {
            const unsigned lo = __shfl((unsigned)(removed & 0xFFFFFFFFull), i >> 6, 64);
            const unsigned hi = __shfl((unsigned)(removed >> 32), i >> 6, 64);
            const unsigned long long word = ((unsigned long long)hi << 32) | lo;
            if (!((word >> (i & 63)) & 1ull)) {
                if (lane == 0) s.keep[nkeep] = i;
                ++nkeep;
                if (lane < 16) removed |= s.mask[i][lane];
            }
        }
        if (lane == 0) s.nkeep = nkeep;
    }
    __syncthreads();
    const int nkeep = s.nkeep;
    float* pout = proposals + (size_t)b * post_nms * 4;
    for (int r = tid; r < post_nms; r += PROP_THREADS) {
        f32x4 o = {0.f, 0.f, 0.f, 0.f};
        if (r < nkeep) {
            const int i = s.keep[r];
            o[0] = s.box[i][0]; o[1] = s.box[i][1]; o[2] = s.box[i][2]; o[3] = s.box[i][3];
        }
        *reinterpret_cast<f32x4*>(pout + (size_t)r * 4) = o;
    }
    if (tid == 0) counts[b] = nkeep;
}

__global__ void prefix_counts_kernel(const int* __restrict__ counts, int* __restrict__ offsets, int B) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        int acc = 0;
        offsets[0] = 0;
        for (int b = 0; b < B; ++b) { acc += counts[b]; offsets[b + 1] = acc; }
    }
}

// ----------------------------------------------------------------------------------
// RoIAlign(8x8, sampling_ratio 2, aligned=False) + AvgPool2d(8), NHWC.
// Workgroup = (64-channel slab, 32 RoIs, image), 512 threads, TWO per CU (the slab of the WHOLE feature map - FH*FW
// positions x 64 channels fp32 = 64 KiB at 16x16 - plus 12 KiB of sampling tables is 76 KiB of LDS): while one stages its
// slab the other computes.  A 16-lane group owns ONE RoI (all 64 bins of its 4 channels per lane): 32 RoIs in flight per
// workgroup, no barrier inside, the 8x8 average stays in registers; 16 weighted ds_read_b128 per output float4, output
// rows written as 256-B runs in [bin][channel] order.  (Rounds 1-3: a 128-channel slab with 1024 threads, one workgroup
// per CU, RoIs dealt out in 32 even chunks per image - 26 of 32 lane groups busy at 831 RoIs and nobody computing while a
// slab was staged; measured 0.39 / 0.21 of HBM for fp32 / 16-bit maps.  The arithmetic - VALU - bounds this kernel, not
// the stores: see DESIGN.md.)
// ----------------------------------------------------------------------------------
// Sampling tables of one RoI, per y / x sample (bin*2 + sample):
//   y: 8 bytes  {lo (BYTE offset of the row in the slab: y * FW * 256; bit 31 = dead), l}
//   x: 16 bytes {h, l, lo (byte offset of the column: x * 256) as int bits, -}   ((h, l) is an aligned register pair: the four
//      bilinear weights of a sample are two packed products)
// h = 1 - l as torchvision computes it; a sample outside the map ("dead") gets l = h = 0, so all four of its bilinear
// weights are exactly 0 and it adds exactly 0 - same result as skipping it.  The HIGH neighbour is not stored: it is always
// lo + one column / one row.  At the right / bottom border torchvision uses hi = lo with l = 0, i.e. the high taps carry
// weight 0; here they read the next position (the next row's first column, or the finite table words behind the slab)
// and 0 * finite is the same 0: four address adds per bin instead of sixteen, the rest are instruction offsets.
struct RoiTables {
    uint2 y[16];
    f32x4 x[16];
};
constexpr int ROI_THREADS = 512;   // 8 waves; two workgroups per CU
constexpr int ROI_SUB = 32;        // RoIs per workgroup = 16-lane groups

// OUT16 (1 bf16, 2 fp16; 0 = fp32 maps): the [roi][bin][channel] maps are stored as 16-bit values (round to nearest even) - under torch.autocast the box head
// runs in reduced precision, and fc6 (81 % of the detector's FLOPs) then reads its A operand through the LDS-DMA GEMM at
// half the bytes; the 8x8 average (top_region_features) is still formed from the unrounded values.
// RSB: bytes per slab row when known at compile time (FW = 16: 4096; the tap offsets are then instruction immediates), 0 = FW * 256 at run time.
template <int OUT16, int RSB>
__global__ __launch_bounds__(ROI_THREADS) __attribute__((amdgpu_waves_per_eu(4, 4))) void roi_align_avg_kernel(const float* __restrict__ feat,
                                                            const float* __restrict__ proposals,
                                                            const int* __restrict__ offsets, float* __restrict__ out,
                                                            float* __restrict__ pooled, int FH, int FW, int C,
                                                            int max_props, float spatial_scale) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const int npos = FH * FW;
    float* slab = reinterpret_cast<float*>(smem_raw);                         // [npos][64]
    RoiTables* tabs = reinterpret_cast<RoiTables*>(slab + (size_t)npos * 64);  // [ROI_SUB]; also what taps past the slab's end read
    const int tid = threadIdx.x;
    const int slab_i = blockIdx.x, chunk = blockIdx.y, b = blockIdx.z;
    const int off = offsets[b], nb = offsets[b + 1] - off;
    const int s0 = chunk * ROI_SUB;
    if (s0 >= nb) return;
    const int ns = (nb - s0 < ROI_SUB) ? nb - s0 : ROI_SUB;

    const float* fsrc = feat + (size_t)b * npos * C + slab_i * 64;
    for (int i = tid; i < npos * 16; i += ROI_THREADS) {
        const int pos = i >> 4, c4 = i & 15;
        *reinterpret_cast<f32x4*>(slab + pos * 64 + c4 * 4) =
            *reinterpret_cast<const f32x4*>(fsrc + (size_t)pos * C + c4 * 4);
    }
    // all 32 tables are written (zeros behind the last RoI): the words behind the slab must be finite
    for (int i = tid; i < ROI_SUB * 32; i += ROI_THREADS) {
        const int rr = i >> 5, t32 = i & 31;
        const int ax = t32 >> 4;  // 0: y, 1: x
        const int sidx = t32 & 15, bin = sidx >> 1, g = sidx & 1;
        int lo = 0, dead = 1;
        float l = 0.f;
        if (rr < ns) {
            const float* pb = proposals + ((size_t)b * max_props + s0 + rr) * 4;
            const float start = (ax ? pb[0] : pb[1]) * spatial_scale;
            const float end = (ax ? pb[2] : pb[3]) * spatial_scale;
            const float roi = fmaxf(end - start, 1.0f);
            const float bsz = roi / 8.0f;
            const int size = ax ? FW : FH;
            float v = start + (float)bin * bsz + ((float)g + 0.5f) * bsz / 2.0f;
            dead = (v < -1.0f || v > (float)size) ? 1 : 0;
            v = fmaxf(v, 0.0f);
            lo = (int)v;
            if (lo >= size - 1) { lo = size - 1; v = (float)lo; }
            l = v - (float)lo;
        }
        if (ax) {
            f32x4 rec;
            rec[0] = dead ? 0.0f : 1.0f - l; rec[1] = dead ? 0.0f : l; rec[2] = __int_as_float(lo * 256); rec[3] = 0.0f;
            tabs[rr].x[sidx] = rec;
        } else {
            tabs[rr].y[sidx] = make_uint2((unsigned)(lo * FW * 256) | (dead ? 0x80000000u : 0u), __float_as_uint(l));
        }
    }
    __syncthreads();
    const int c4 = tid & 15, rr = tid >> 4;
    if (rr >= ns) return;
    const int rsb = RSB ? RSB : FW * 256;
    const unsigned char* slab_lane = reinterpret_cast<const unsigned char*>(slab) + c4 * 16;   // this lane's 4 channels of position 0
    const RoiTables& T = tabs[rr];
    const int r = s0 + rr;
    const size_t oidx = (size_t)(off + r) * 64 * C + slab_i * 64 + c4 * 4;
    float* obase = out + oidx;
    unsigned short* obase16 = reinterpret_cast<unsigned short*>(out) + oidx;
    f32x4 total = {0.f, 0.f, 0.f, 0.f};
    for (int ph = 0; ph < 8; ++ph) {
        f32x4 psum = {0.f, 0.f, 0.f, 0.f};
        int ylo[2];
        float ly[2], hy[2];
#pragma unroll
        for (int iy = 0; iy < 2; ++iy) {
            const uint2 ry = T.y[ph * 2 + iy];
            const bool dead = (ry.x >> 31) != 0;
            const float l = __uint_as_float(ry.y);
            ylo[iy] = (int)(ry.x & 0x7fffffffu);
            ly[iy] = dead ? 0.0f : l;
            hy[iy] = dead ? 0.0f : 1.0f - l;
        }
        // The column records of bin pw + 1 are fetched while bin pw is computed, and the 16 taps of a bin are all requested
        // before the first product (pinned with a scheduling barrier).
        f32x4 tx0 = T.x[0], tx1 = T.x[1];
#pragma unroll 1
        for (int pw = 0; pw < 8; ++pw) {
            const int pn = (pw + 1) & 7;
            const f32x4 nx0 = T.x[pn * 2], nx1 = T.x[pn * 2 + 1];
            f32x4 v[16];
#pragma unroll
            for (int iy = 0; iy < 2; ++iy) {
#pragma unroll
                for (int ix = 0; ix < 2; ++ix) {
                    const unsigned char* a = slab_lane + ylo[iy] + __float_as_int(ix ? tx1[2] : tx0[2]);
                    const int k = (iy * 2 + ix) * 4;
                    v[k + 0] = *reinterpret_cast<const f32x4*>(a);
                    v[k + 1] = *reinterpret_cast<const f32x4*>(a + 256);
                    v[k + 2] = *reinterpret_cast<const f32x4*>(a + rsb);
                    v[k + 3] = *reinterpret_cast<const f32x4*>(a + rsb + 256);
                }
            }
            __builtin_amdgcn_sched_barrier(0);   // (without it: 0.470 instead of 0.488 of HBM, profiles/r04_roialign_variants.log)
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int iy = 0; iy < 2; ++iy) {
#pragma unroll
                for (int ix = 0; ix < 2; ++ix) {
                    const f32x4 tx = ix ? tx1 : tx0;
                    typedef float f32x2_t __attribute__((ext_vector_type(2)));
                    const f32x2_t hl = {tx[0], tx[1]};                                   // (hx, lx)
                    const f32x2_t w12 = f32x2_t{hy[iy], hy[iy]} * hl, w34 = f32x2_t{ly[iy], ly[iy]} * hl;   // v_pk_mul_f32
                    const float w1 = w12[0], w2 = w12[1], w3 = w34[0], w4 = w34[1];
                    const int k = (iy * 2 + ix) * 4;
                    if constexpr (OUT16) {
                        // 16-bit maps (torch.autocast): the result is rounded to 11 / 8 bits anyway and the reference's own
                        // GPU kernel contracts - fused multiply-adds, half the VALU work (this kernel is VALU / power bound)
                        acc = __builtin_elementwise_fma(f32x4{w1, w1, w1, w1}, v[k], acc);
                        acc = __builtin_elementwise_fma(f32x4{w2, w2, w2, w2}, v[k + 1], acc);
                        acc = __builtin_elementwise_fma(f32x4{w3, w3, w3, w3}, v[k + 2], acc);
                        acc = __builtin_elementwise_fma(f32x4{w4, w4, w4, w4}, v[k + 3], acc);
                    } else {
#pragma unroll
                        for (int e = 0; e < 4; ++e) {   // torchvision's CPU order, unfused: bit-exact with the oracle
                            const float val = w1 * v[k][e] + w2 * v[k + 1][e] + w3 * v[k + 2][e] + w4 * v[k + 3][e];
                            acc[e] = acc[e] + val;
                        }
                    }
                }
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) { acc[e] = acc[e] / 4.0f; psum[e] += acc[e]; }
            if constexpr (OUT16) {
                uint2 pk;
                pk.x = to16<OUT16 == 2>(acc[0]) | (to16<OUT16 == 2>(acc[1]) << 16);
                pk.y = to16<OUT16 == 2>(acc[2]) | (to16<OUT16 == 2>(acc[3]) << 16);
                *reinterpret_cast<uint2*>(obase16 + (size_t)(ph * 8 + pw) * C) = pk;
            } else {
                *reinterpret_cast<f32x4*>(obase + (size_t)(ph * 8 + pw) * C) = acc;
            }
            tx0 = nx0; tx1 = nx1;
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) total[e] += psum[e];  // row sums added in row order (same association as before)
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) total[e] = total[e] / 64.0f;
    *reinterpret_cast<f32x4*>(pooled + (size_t)(off + r) * C + slab_i * 64 + c4 * 4) = total;
}

// ----------------------------------------------------------------------------------
// Top-1 box per class (custom_roi_heads.py:63-208).  One workgroup per image.
// ----------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void top1_per_class_kernel(const float* __restrict__ pred, int ldp,
                                                             const float* __restrict__ proposals,
                                                             const int* __restrict__ offsets,
                                                             const float* __restrict__ pooled,
                                                             unsigned char* __restrict__ class_detected,
                                                             float* __restrict__ top_scores,
                                                             float* __restrict__ top_boxes,
                                                             float* __restrict__ top_feats, int C, int max_props,
                                                             float img_w, float img_h) {
    __shared__ int cstar[1024];
    __shared__ float cscore[1024];
    __shared__ int best_idx[29];
    const int tid = threadIdx.x, b = blockIdx.x;
    const int off = offsets[b], nb = offsets[b + 1] - off;
    for (int r = tid; r < nb; r += 256) {
        const float* x = pred + (size_t)(off + r) * ldp;
        float m = x[0];
        for (int i = 1; i < 30; ++i) m = fmaxf(m, x[i]);
        float e[30], sum = 0.f;
        for (int i = 0; i < 30; ++i) { e[i] = expf(x[i] - m); sum += e[i]; }
        int bi = 0;
        float bp = e[1] / sum;
        for (int i = 1; i < 29; ++i) {
            const float p = e[i + 1] / sum;
            if (p > bp) { bp = p; bi = i; }
        }
        cstar[r] = bi;
        cscore[r] = bp;
    }
    __syncthreads();
    if (tid < 29) {
        float best = 0.f;
        int idx = 0, cnt = 0;
        for (int r = 0; r < nb; ++r)
            if (cstar[r] == tid) {
                ++cnt;
                if (cscore[r] > best) { best = cscore[r]; idx = r; }
            }
        best_idx[tid] = idx;
        class_detected[b * 29 + tid] = cnt > 0 ? 1 : 0;
        top_scores[b * 29 + tid] = best;
        float o[4] = {0.f, 0.f, 0.f, 0.f};
        if (nb > 0) {
            const float* d = pred + (size_t)(off + idx) * ldp + 30 + (tid + 1) * 4;
            const float* pb = proposals + ((size_t)b * max_props + idx) * 4;
            const float w = pb[2] - pb[0], h = pb[3] - pb[1];
            const float cx = pb[0] + 0.5f * w, cy = pb[1] + 0.5f * h;
            const float clipv = 4.135166556742356f;
            const float dx = d[0] / 10.0f, dy = d[1] / 10.0f;
            const float dw = fminf(d[2] / 5.0f, clipv), dh = fminf(d[3] / 5.0f, clipv);
            const float pcx = dx * w + cx, pcy = dy * h + cy;
            const float pw = expf(dw) * w, phh = expf(dh) * h;
            const float hw_ = 0.5f * pw, hh_ = 0.5f * phh;
            o[0] = fminf(fmaxf(pcx - hw_, 0.f), img_w);
            o[1] = fminf(fmaxf(pcy - hh_, 0.f), img_h);
            o[2] = fminf(fmaxf(pcx + hw_, 0.f), img_w);
            o[3] = fminf(fmaxf(pcy + hh_, 0.f), img_h);
        }
        for (int k = 0; k < 4; ++k) top_boxes[((size_t)b * 29 + tid) * 4 + k] = o[k];
    }
    __syncthreads();
    const int C4 = C >> 2;
    for (int i = tid; i < 29 * C4; i += 256) {
        const int c = i / C4, q = i - c * C4;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (nb > 0) v = *reinterpret_cast<const f32x4*>(pooled + (size_t)(off + best_idx[c]) * C + q * 4);
        *reinterpret_cast<f32x4*>(top_feats + ((size_t)b * 29 + c) * C + q * 4) = v;
    }
}

// selected = (logit > thr) & detected; ordered compaction of the selected flat indices.
__global__ __launch_bounds__(1024) void select_regions_kernel(const float* __restrict__ logits,
                                                              const unsigned char* __restrict__ detected, float thr,
                                                              unsigned char* __restrict__ selected,
                                                              int* __restrict__ sel_rows, int* __restrict__ n_selected,
                                                              int n) {
    __shared__ int wave_tot[16];
    __shared__ int base;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid == 0) base = 0;
    __syncthreads();
    for (int start = 0; start < n; start += 1024) {
        const int i = start + tid;
        const bool sel = (i < n) && (logits[i] > thr) && (detected[i] != 0);
        if (i < n) selected[i] = sel ? 1 : 0;
        const unsigned long long bal = __ballot(sel);
        if (lane == 0) wave_tot[wave] = __popcll(bal);
        __syncthreads();
        int off = base, tot = 0;
        for (int w2 = 0; w2 < 16; ++w2) {
            if (w2 < wave) off += wave_tot[w2];
            tot += wave_tot[w2];
        }
        if (sel) sel_rows[off + __popcll(bal & ((1ull << lane) - 1ull))] = i;
        __syncthreads();
        if (tid == 0) base += tot;
        __syncthreads();
    }
    if (tid == 0) *n_selected = base;
}

// nn.BCEWithLogitsLoss(pos_weight=w) over the rows with mask != 0, mean reduction
// (binary_classifier_region_selection.py:22,40-44, binary_classifier_region_abnormal.py:29,43-47):
//   l_i = (1 - y_i) x_i - (1 + (w - 1) y_i) log_sigmoid(x_i),  log_sigmoid(x) = min(x, 0) - log1p(exp(-|x|))
// One workgroup, fixed summation order (double); no row -> nan, like torch's mean of an empty tensor.
__global__ __launch_bounds__(256) void bce_logits_masked_kernel(const float* __restrict__ logits,
                                                                const unsigned char* __restrict__ mask,
                                                                const unsigned char* __restrict__ target, float pos_weight,
                                                                int n, float* __restrict__ loss) {
    __shared__ double ssum[256];
    __shared__ int scnt[256];
    double a = 0.0;
    int c = 0;
    for (int i = threadIdx.x; i < n; i += 256) {
        if (!mask[i]) continue;
        const float x = logits[i], y = target[i] ? 1.f : 0.f;
        const float ls = fminf(x, 0.f) - log1pf(expf(-fabsf(x)));
        const float lw = (pos_weight - 1.f) * y + 1.f;
        a += (double)((1.f - y) * x - lw * ls);
        ++c;
    }
    ssum[threadIdx.x] = a;
    scnt[threadIdx.x] = c;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (threadIdx.x < o) { ssum[threadIdx.x] += ssum[threadIdx.x + o]; scnt[threadIdx.x] += scnt[threadIdx.x + o]; }
        __syncthreads();
    }
    if (threadIdx.x == 0) *loss = (float)(ssum[0] / (double)scnt[0]);
}

// get_image_tensor (generate_reports_for_images.py:129-147): LongestMaxSize(512, INTER_AREA) -> centred zero
// PadIfNeeded(512, 512) -> Normalize(mean, std) of an 8-bit gray image, one thread per pixel of the [512,512] output.
// The INTER_AREA arithmetic follows OpenCV's resize.cpp for 8-bit down-scaling (third-party, restated: see
// oracle/preprocess.py): mode 1 = 2x2 integer ((a+b+c+d+2)>>2), mode 2 = integer scales (int sum * float(1/area)),
// mode 3 = fractional coverage tables (double geometry, float weights, float sums: x in table order, then rows in
// order), saturate_cast = round half to even; mode 4 = an axis is ENLARGED (images smaller than 512 px): OpenCV
// emulates INTER_AREA with its 8-bit fixed-point bilinear path and the area coordinate rule (s = floor(d * scale),
// f = (d + 1) - (s + 1) * inv_scale, f <= 0 ? 0 : f - floor(f); short weights x 2048; columns with s + 1 >= width read
// S[width - 1] * 2048; second row clamped; ((b0 * (S0 >> 4)) >> 16) + ((b1 * (S1 >> 4)) >> 16) + 2 >> 2).
// Compiled with fp contraction off (file pragma).
struct LinTap { int s; int a0, a1; bool edge; };
__device__ __forceinline__ LinTap area_linear_tap(int d, int ssize, int dsize) {
    const double inv_scale = (double)dsize / (double)ssize, scale = 1.0 / inv_scale;
    LinTap t;
    t.s = (int)floor((double)d * scale);
    float f = (float)(((double)d + 1.0) - ((double)t.s + 1.0) * inv_scale);
    f = f <= 0.f ? 0.f : f - floorf(f);
    t.edge = t.s + 1 >= ssize;
    t.a0 = (int)fminf(fmaxf(rintf((1.0f - f) * 2048.0f), -32768.f), 32767.f);
    t.a1 = (int)fminf(fmaxf(rintf(f * 2048.0f), -32768.f), 32767.f);
    return t;
}
__device__ __forceinline__ int area_linear_hrow(const unsigned char* __restrict__ r, const LinTap& x, int w) {
    if (x.edge) return (int)r[w - 1] * 2048;   // s >= w - 1 there: fx = 0, sx = w - 1
    return (int)r[x.s] * x.a0 + (int)r[x.s + 1] * x.a1;
}
struct AreaTaps {  // the <= 3 kinds of entries of one destination index: [partial first][full ...][partial last]
    int first, n_full, last;  // source index of the partial-first entry (-1: none), count of full entries, last (-1: none)
    int full0;
    float a_first, a_full, a_last;
};
__device__ __forceinline__ AreaTaps area_taps(int d, int ssize, double scale) {
    AreaTaps t;
    const double f1 = d * scale, f2 = f1 + scale;
    const double cell = fmin(scale, (double)ssize - f1);
    int s1 = (int)ceil(f1), s2 = (int)floor(f2);
    s2 = min(s2, ssize - 1);
    s1 = min(s1, s2);
    t.first = -1; t.last = -1;
    t.a_first = 0.f; t.a_last = 0.f;
    if (s1 - f1 > 1e-3) { t.first = s1 - 1; t.a_first = (float)((s1 - f1) / cell); }
    t.full0 = s1; t.n_full = s2 - s1; t.a_full = (float)(1.0 / cell);
    if (f2 - s2 > 1e-3) { t.last = s2; t.a_last = (float)(fmin(fmin(f2 - s2, 1.0), cell) / cell); }
    return t;
}
__device__ __forceinline__ float area_row(const unsigned char* __restrict__ r, const AreaTaps& x) {
    float buf = 0.f;
    if (x.first >= 0) buf += (float)r[x.first] * x.a_first;
    for (int i = 0; i < x.n_full; ++i) buf += (float)r[x.full0 + i] * x.a_full;
    if (x.last >= 0) buf += (float)r[x.last] * x.a_last;
    return buf;
}
__global__ __launch_bounds__(256) void preprocess_u8_kernel(const unsigned char* __restrict__ src, int h, int w, int stride, int nh,
                                                            int nw, int top, int left, double sx, double sy, int mode, int isx,
                                                            int isy, float mean255, float denom, float* __restrict__ out, int size) {
    const int ox = blockIdx.x * 16 + (threadIdx.x & 15), oy = blockIdx.y * 16 + (threadIdx.x >> 4);
    if (ox >= size || oy >= size) return;
    const int x = ox - left, y = oy - top;
    float v = 0.f;
    if (x >= 0 && x < nw && y >= 0 && y < nh) {
        if (mode == 0) {
            v = (float)src[(size_t)y * stride + x];
        } else if (mode == 1) {
            const unsigned char* r0 = src + (size_t)(2 * y) * stride + 2 * x;
            v = (float)((r0[0] + r0[1] + r0[stride] + r0[stride + 1] + 2) >> 2);
        } else if (mode == 2) {
            int sum = 0;
            for (int j = 0; j < isy; ++j)
                for (int i = 0; i < isx; ++i) sum += src[(size_t)(y * isy + j) * stride + x * isx + i];
            v = fminf(fmaxf(rintf((float)sum * (1.0f / (float)(isx * isy))), 0.f), 255.f);
        } else if (mode == 4) {
            const LinTap tx = area_linear_tap(x, w, nw), ty = area_linear_tap(y, h, nh);
            const int r0 = min(max(ty.s, 0), h - 1), r1 = min(max(ty.s + 1, 0), h - 1);
            const int s0 = area_linear_hrow(src + (size_t)r0 * stride, tx, w) >> 4;
            const int s1 = area_linear_hrow(src + (size_t)r1 * stride, tx, w) >> 4;
            v = (float)(((((ty.a0 * s0) >> 16) + ((ty.a1 * s1) >> 16) + 2) >> 2) & 0xFF);
        } else {
            const AreaTaps tx = area_taps(x, w, sx), ty = area_taps(y, h, sy);
            float sum = 0.f;
            bool firstrow = true;
            auto add_row = [&](int srow, float beta) {
                const float b = area_row(src + (size_t)srow * stride, tx);
                sum = firstrow ? beta * b : sum + beta * b;
                firstrow = false;
            };
            if (ty.first >= 0) add_row(ty.first, ty.a_first);
            for (int j = 0; j < ty.n_full; ++j) add_row(ty.full0 + j, ty.a_full);
            if (ty.last >= 0) add_row(ty.last, ty.a_last);
            v = fminf(fmaxf(rintf(sum), 0.f), 255.f);
        }
    }
    out[(size_t)oy * size + ox] = (v - mean255) * denom;
}

__global__ __launch_bounds__(256) void gather_rows_kernel(const float* __restrict__ src, const int* __restrict__ rows,
                                                          float* __restrict__ dst, int D4) {
    const int r = blockIdx.x;
    const int s = rows[r];
    for (int q = threadIdx.x; q < D4; q += 256)
        reinterpret_cast<f32x4*>(dst + (size_t)r * D4 * 4)[q] = reinterpret_cast<const f32x4*>(src + (size_t)s * D4 * 4)[q];
}

}  // namespace rgrg

using namespace rgrg;

extern "C" int rgrg_stem_conv7x7_f32(const float* X, const float* Wt, const float* scale, const float* shift, float* Y,
                                     int B, int H, int Wd, void* stream) {
    RGRG_CHECK_ARG(X && Wt && scale && shift && Y && B > 0 && H % 32 == 0 && Wd % 32 == 0);
    const int OH = H / 2, OW = Wd / 2;
    hipLaunchKernelGGL(stem_conv7x7_kernel, dim3(OW / 16, OH / 16, B), dim3(256), 0, as_stream(stream), X, Wt, scale,
                       shift, Y, H, Wd, OH, OW);
    RGRG_LAUNCH_CHECK();
    return RGRG_OK;
}

extern "C" int rgrg_maxpool3x3s2_nhwc_f32(const float* X, float* Y, int B, int H, int Wd, int C, void* stream) {
    RGRG_CHECK_ARG(X && Y && B > 0 && C % 4 == 0);
    const int OH = (H + 2 - 3) / 2 + 1, OW = (Wd + 2 - 3) / 2 + 1;
    const size_t total = (size_t)B * OH * OW * (C / 4);
    const int blocks = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
    hipLaunchKernelGGL(maxpool3x3s2_kernel, dim3(blocks), dim3(256), 0, as_stream(stream), X, Y, B, H, Wd, C, OH, OW);
    RGRG_LAUNCH_CHECK();
    return RGRG_OK;
}

extern "C" int rgrg_rpn_proposals_f32(const float* head_out, const float* anchors, float* proposals, int32_t* counts,
                                      int32_t* offsets, int B, int HW, int A, int pre_nms, int post_nms,
                                      float nms_thresh, float min_size, float img_w, float img_h, void* stream) {
    RGRG_CHECK_ARG(head_out && anchors && proposals && counts && offsets && B > 0);
    RGRG_CHECK_ARG(pre_nms > 0 && pre_nms <= PROP_ROWS && post_nms > 0 && post_nms <= pre_nms && HW * A <= 65536);
    static bool attr_set = false;
    if (!attr_set) {
        RGRG_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&rpn_proposals_kernel),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(PropSmem)));
        attr_set = true;
    }
    hipLaunchKernelGGL(rpn_proposals_kernel, dim3(B), dim3(PROP_THREADS), sizeof(PropSmem), as_stream(stream), head_out,
                       anchors, proposals, counts, HW, A, pre_nms, post_nms, nms_thresh, min_size, img_w, img_h);
    RGRG_LAUNCH_CHECK();
    hipLaunchKernelGGL(prefix_counts_kernel, dim3(1), dim3(64), 0, as_stream(stream), counts, offsets, B);
    RGRG_LAUNCH_CHECK();
    return RGRG_OK;
}

static int roi_align_launch(const float* feat, const float* proposals, const int32_t* offsets, void* out, int out16,
                            float* pooled, int B, int FH, int FW, int C, int max_props, int R_total,
                            float spatial_scale, void* stream) {
    RGRG_CHECK_ARG(feat && proposals && offsets && out && pooled && B > 0 && C % 64 == 0 && FH > 0 && FW > 0);
    const size_t lds = (size_t)FH * FW * 64 * 4 + ROI_SUB * sizeof(RoiTables);
    // taps of the last row / column read up to one row + one column past the slab: that must stay inside the tables
    RGRG_CHECK_ARG(lds <= 160 * 1024 && (size_t)(FW + 2) * 256 <= ROI_SUB * sizeof(RoiTables));
    if (R_total <= 0) return RGRG_OK;
    static bool attr_set = false;
    if (!attr_set) {
#define ROI_ATTR(O_, R_) RGRG_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&roi_align_avg_kernel<O_, R_>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024))
        ROI_ATTR(0, 4096); ROI_ATTR(1, 4096); ROI_ATTR(2, 4096); ROI_ATTR(0, 0); ROI_ATTR(1, 0); ROI_ATTR(2, 0);
#undef ROI_ATTR
        attr_set = true;
    }
    const int slabs = C / 64;
    // chunks of 32 RoIs (one per 16-lane group); chunks beyond an image's RoI count exit immediately
    int nchunk = (max_props + ROI_SUB - 1) / ROI_SUB;
    if (nchunk < 1) nchunk = 1;
#define ROI_LAUNCH(O_, R_) hipLaunchKernelGGL((roi_align_avg_kernel<O_, R_>), dim3(slabs, nchunk, B), dim3(ROI_THREADS), lds, as_stream(stream), feat, proposals, \
                                              offsets, reinterpret_cast<float*>(out), pooled, FH, FW, C, max_props, spatial_scale)
    if (FW == 16) {
        if (out16 == 2) ROI_LAUNCH(2, 4096);
        else if (out16 == 1) ROI_LAUNCH(1, 4096);
        else ROI_LAUNCH(0, 4096);
    } else {
        if (out16 == 2) ROI_LAUNCH(2, 0);
        else if (out16 == 1) ROI_LAUNCH(1, 0);
        else ROI_LAUNCH(0, 0);
    }
#undef ROI_LAUNCH
    RGRG_LAUNCH_CHECK();
    return RGRG_OK;
}

extern "C" int rgrg_roi_align_avgpool_f32(const float* feat, const float* proposals, const int32_t* offsets, float* out,
                                          float* pooled, int B, int FH, int FW, int C, int max_props, int R_total,
                                          float spatial_scale, void* stream) {
    return roi_align_launch(feat, proposals, offsets, out, 0, pooled, B, FH, FW, C, max_props, R_total, spatial_scale, stream);
}

extern "C" int rgrg_roi_align_avgpool_bf16maps(const float* feat, const float* proposals, const int32_t* offsets, uint16_t* out16,
                                               float* pooled, int B, int FH, int FW, int C, int max_props, int R_total,
                                               float spatial_scale, int fp16, void* stream) {
    return roi_align_launch(feat, proposals, offsets, out16, fp16 ? 2 : 1, pooled, B, FH, FW, C, max_props, R_total, spatial_scale, stream);
}

extern "C" int rgrg_top1_per_class_f32(const float* pred, int ldp, const float* proposals, const int32_t* offsets,
                                       const float* pooled, uint8_t* class_detected, float* top_scores,
                                       float* top_boxes, float* top_feats, int B, int C, int max_props, float img_w,
                                       float img_h, void* stream) {
    RGRG_CHECK_ARG(pred && proposals && offsets && pooled && class_detected && top_scores && top_boxes && top_feats);
    RGRG_CHECK_ARG(B > 0 && ldp >= 150 && C % 4 == 0 && max_props <= 1024);
    hipLaunchKernelGGL(top1_per_class_kernel, dim3(B), dim3(256), 0, as_stream(stream), pred, ldp, proposals, offsets,
                       pooled, class_detected, top_scores, top_boxes, top_feats, C, max_props, img_w, img_h);
    RGRG_LAUNCH_CHECK();
    return RGRG_OK;
}

extern "C" int rgrg_select_regions_f32(const float* logits, const uint8_t* class_detected, float thr, uint8_t* selected,
                                       int32_t* sel_rows, int32_t* n_selected, int n, void* stream) {
    RGRG_CHECK_ARG(logits && class_detected && selected && sel_rows && n_selected && n > 0);
    hipLaunchKernelGGL(select_regions_kernel, dim3(1), dim3(1024), 0, as_stream(stream), logits, class_detected, thr,
                       selected, sel_rows, n_selected, n);
    RGRG_LAUNCH_CHECK();
    return RGRG_OK;
}

extern "C" int rgrg_bce_with_logits_masked_f32(const float* logits, const uint8_t* mask, const uint8_t* target,
                                               float pos_weight, int n, float* loss, void* stream) {
    RGRG_CHECK_ARG(logits && mask && target && loss && n > 0);
    hipLaunchKernelGGL(bce_logits_masked_kernel, dim3(1), dim3(256), 0, as_stream(stream), logits, mask, target, pos_weight, n, loss);
    RGRG_LAUNCH_CHECK();
    return RGRG_OK;
}

extern "C" int rgrg_preprocess_u8_f32(const uint8_t* src, int h, int w, int src_stride, int new_h, int new_w, float mean,
                                      float std, float* dst, void* stream) {
    constexpr int SIZE = 512;
    RGRG_CHECK_ARG(src && dst && h > 0 && w > 0 && src_stride >= w && new_h > 0 && new_w > 0 && new_h <= SIZE && new_w <= SIZE);
    const double sx = (double)w / new_w, sy = (double)h / new_h;
    const int isx = (int)lround(sx), isy = (int)lround(sy);
    int mode = 3;
    if (new_h == h && new_w == w) mode = 0;
    else if (!(sx >= 1.0 && sy >= 1.0)) mode = 4;  // an axis is enlarged: OpenCV's bilinear emulation of INTER_AREA
    else if (fabs(sx - isx) < DBL_EPSILON && fabs(sy - isy) < DBL_EPSILON) mode = (isx == 2 && isy == 2) ? 1 : 2;
    const int top = (int)((SIZE - new_h) / 2.0), left = (int)((SIZE - new_w) / 2.0);
    const float mean255 = mean * 255.0f, denom = 1.0f / (std * 255.0f);
    hipLaunchKernelGGL(preprocess_u8_kernel, dim3(SIZE / 16, SIZE / 16), dim3(256), 0, as_stream(stream), src, h, w, src_stride, new_h,
                       new_w, top, left, sx, sy, mode, isx, isy, mean255, denom, dst, SIZE);
    RGRG_LAUNCH_CHECK();
    return RGRG_OK;
}

extern "C" int rgrg_gather_rows_f32(const float* src, const int32_t* rows, float* dst, int n_rows, int D, void* stream) {
    RGRG_CHECK_ARG(src && rows && dst && D % 4 == 0);
    if (n_rows <= 0) return RGRG_OK;
    hipLaunchKernelGGL(gather_rows_kernel, dim3(n_rows), dim3(256), 0, as_stream(stream), src, rows, dst, D / 4);
    RGRG_LAUNCH_CHECK();
    return RGRG_OK;
}
