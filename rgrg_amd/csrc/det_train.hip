// Detector targets and losses for gfx950: what torchvision 0.13.1's RegionProposalNetwork / RoIHeads compute when
// ``targets`` are given (custom_rpn.py:74-83 -> assign_targets_to_anchors / box_coder.encode / compute_loss;
// custom_roi_heads.py:225-242 -> select_training_samples / fastrcnn_loss), i.e. the detector half of
// ReportGenerationModel.forward(images, image_targets, ...) (report_generation_model.py:55,91) that the reference's
// validation loop calls in eval mode (evaluate_model.py:413).
//
// Everything between the RPN head / the proposals and the four loss scalars runs here: the O(gt x boxes) matching, the
// BalancedPositiveNegativeSampler (an exact radix SELECT of the k smallest keys per class - one workgroup per image,
// keys from Philox4x32-10 or injected for tests - instead of torchvision's randperm over index lists),
// add_gt_proposals, the gather / encode of the sampled rows and the loss reductions.  The caller only allocates.
// Everything is fp32 with IEEE division and no contraction, in torchvision's operation order, so match decisions
// (IoU >= threshold, IoU == row maximum) agree with a CPU evaluation.
#include <math.h>

#include "common.h"

#pragma clang fp contract(off)

namespace rgrg {

constexpr int MATCH_BELOW = -1, MATCH_BETWEEN = -2;

__device__ __forceinline__ float box_iou1(const f32x4 g, const f32x4 b) {  // ops.boxes.box_iou, one pair
    const float area1 = (g[2] - g[0]) * (g[3] - g[1]);
    const float area2 = (b[2] - b[0]) * (b[3] - b[1]);
    const float w = fmaxf(fminf(g[2], b[2]) - fmaxf(g[0], b[0]), 0.f);
    const float h = fmaxf(fminf(g[3], b[3]) - fmaxf(g[1], b[1]), 0.f);
    const float inter = w * h;
    return inter / (area1 + area2 - inter);
}

// Matcher.__call__, first pass: per box the best gt (first maximum) and the thresholded code; per gt the best IoU over
// all boxes of the image (max on the bit pattern: IoUs are >= 0, so float order == int order) - reduced across the wave
// first, then through LDS, so that a workgroup issues G global atomics instead of 256 x G.
__global__ __launch_bounds__(256) void box_match_kernel(const float* __restrict__ gt, const int* __restrict__ gt_count, int G,
                                                        const float* __restrict__ boxes, size_t box_image_stride,
                                                        const int* __restrict__ box_count, int N, float high, float low,
                                                        int* __restrict__ matched, int* __restrict__ best_per_gt) {
    extern __shared__ int sh_best[];   // [G]
    const int b = blockIdx.y, i = blockIdx.x * 256 + threadIdx.x;
    const int ng = gt_count[b], nb = box_count ? box_count[b] : N;
    if (ng == 0 || blockIdx.x * 256 >= nb) {   // uniform per workgroup
        if (i < N) matched[(size_t)b * N + i] = MATCH_BELOW;
        return;
    }
    for (int g = threadIdx.x; g < ng; g += 256) sh_best[g] = -1;
    __syncthreads();
    const bool live = i < nb;
    f32x4 bx = {0.f, 0.f, 0.f, 0.f};
    if (live) bx = *reinterpret_cast<const f32x4*>(boxes + (size_t)b * box_image_stride + (size_t)i * 4);
    float best = -1.f;
    int arg = 0;
    for (int g = 0; g < ng; ++g) {
        const float q = box_iou1(*reinterpret_cast<const f32x4*>(gt + ((size_t)b * G + g) * 4), bx);
        if (live && q > best) { best = q; arg = g; }
        int w = live ? __float_as_int(q) : -1;
        for (int o = 32; o > 0; o >>= 1) w = max(w, __shfl_xor(w, o, 64));
        if ((threadIdx.x & 63) == 0) atomicMax(sh_best + g, w);
    }
    if (i < N) matched[(size_t)b * N + i] = !live || best < low ? MATCH_BELOW : (best < high ? MATCH_BETWEEN : arg);
    __syncthreads();
    for (int g = threadIdx.x; g < ng; g += 256) atomicMax(best_per_gt + (size_t)b * G + g, sh_best[g]);
}

// set_low_quality_matches_: every box whose IoU with some gt EQUALS that gt's best IoU keeps its arg-max match
__global__ __launch_bounds__(256) void box_match_low_quality_kernel(const float* __restrict__ gt, const int* __restrict__ gt_count,
                                                                    int G, const float* __restrict__ boxes, size_t box_image_stride,
                                                                    const int* __restrict__ box_count, int N,
                                                                    int* __restrict__ matched, const int* __restrict__ best_per_gt) {
    const int b = blockIdx.y, i = blockIdx.x * 256 + threadIdx.x;
    const int ng = gt_count[b], nb = box_count ? box_count[b] : N;
    if (i >= nb || ng == 0) return;
    const f32x4 bx = *reinterpret_cast<const f32x4*>(boxes + (size_t)b * box_image_stride + (size_t)i * 4);
    float best = -1.f;
    int arg = 0;
    bool restore = false;
    for (int g = 0; g < ng; ++g) {
        const float q = box_iou1(*reinterpret_cast<const f32x4*>(gt + ((size_t)b * G + g) * 4), bx);
        if (q > best) { best = q; arg = g; }
        restore = restore || (__float_as_int(q) == best_per_gt[(size_t)b * G + g]);
    }
    if (restore) matched[(size_t)b * N + i] = arg;
}

// det_utils.encode_boxes, one pair
__device__ __forceinline__ f32x4 box_encode1(const f32x4 r, const f32x4 p, float wx, float wy, float ww, float wh) {
    const float ex_w = p[2] - p[0], ex_h = p[3] - p[1];
    const float ex_cx = p[0] + 0.5f * ex_w, ex_cy = p[1] + 0.5f * ex_h;
    const float gt_w = r[2] - r[0], gt_h = r[3] - r[1];
    const float gt_cx = r[0] + 0.5f * gt_w, gt_cy = r[1] + 0.5f * gt_h;
    f32x4 o;
    o[0] = wx * (gt_cx - ex_cx) / ex_w;
    o[1] = wy * (gt_cy - ex_cy) / ex_h;
    o[2] = ww * logf(gt_w / ex_w);
    o[3] = wh * logf(gt_h / ex_h);
    return o;
}

// ---------------------------------------------------------------------------------------------------------------------
// BalancedPositiveNegativeSampler (det_utils.py; custom_rpn.py:79 with 256 / 0.5, custom_roi_heads.py:225 with 512 / 0.25).
// torchvision: positive[randperm(|positive|)[:num_pos]] - a uniformly random num_pos-subset.  Here every candidate gets an
// i.i.d. 32-bit key and the num_pos SMALLEST keys of the class are taken (ties: lower index first), which is the same
// distribution and needs no permutation, no sort and no index list on the host.

// Philox4x32-10 (Salmon et al., SC'11), first output word; counter = (element, image, stage, 0), key = the call's seed.
__device__ __forceinline__ unsigned philox_key(unsigned c0, unsigned c1, unsigned c2, unsigned k0, unsigned k1) {
    unsigned c3 = 0;
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const unsigned hi0 = __umulhi(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
        const unsigned hi1 = __umulhi(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
        c0 = hi1 ^ c1 ^ k0; c1 = lo1; c2 = hi0 ^ c3 ^ k1; c3 = lo0;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    return c0;
}

// label of candidate i from its match code: RPN (gt_labels == NULL) 1 / 0 / -1 (custom_rpn -> assign_targets_to_anchors);
// RoI heads the matched box's class, 0 below the threshold, -1 between (assign_targets_to_proposals).  Slots past
// box_count are no candidates.
__device__ __forceinline__ int sample_label(int m, const long long* __restrict__ gt_labels_b, int G) {
    if (m == MATCH_BETWEEN) return -1;
    if (m < 0) return 0;
    if (!gt_labels_b) return 1;
    const long long l = gt_labels_b[m < G ? m : G - 1];
    return l >= 1 ? (l > 0x7fffffffll ? 0x7fffffff : (int)l) : (l == 0 ? 0 : -1);
}

// Wide pre-pass: class byte (0 none, 1 positive, 2 negative) and order-preserving 32-bit key per candidate.
__global__ __launch_bounds__(256) void sample_prepare_kernel(const int* __restrict__ matched, const long long* __restrict__ gt_labels,
                                                             int G, const int* __restrict__ box_count,
                                                             const float* __restrict__ keys_f, unsigned seed_lo, unsigned seed_hi,
                                                             unsigned stage, int n, unsigned* __restrict__ keys,
                                                             unsigned char* __restrict__ cls) {
    const int b = blockIdx.y, i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const size_t at = (size_t)b * n + i;
    int lab = -1;
    if (!box_count || i < box_count[b]) lab = sample_label(matched[at], gt_labels ? gt_labels + (size_t)b * G : nullptr, G);
    cls[at] = lab >= 1 ? 1 : (lab == 0 ? 2 : 0);
    unsigned k;
    if (keys_f) {
        const float f = keys_f[at];
        const unsigned u = f == 0.f ? 0u : __float_as_uint(f);       // -0 == +0
        k = (u & 0x80000000u) ? ~u : (u | 0x80000000u);              // float order -> unsigned order
    } else {
        k = philox_key((unsigned)i, (unsigned)b, stage, seed_lo, seed_hi);
    }
    keys[at] = k;
}

__device__ __forceinline__ int wave_incl_scan(int v) {
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int t = __shfl_up(v, o, 64);
        if (lane >= o) v += t;
    }
    return v;
}

// One workgroup (16 waves) per image.  Three histogram passes (11 + 11 + 10 bits, both classes at once) find per class the
// key T with #(key < T) < k <= #(key <= T) and how many of the keys == T to take; the last pass takes them in index order
// (each wave owns a contiguous segment and walks it in 64-wide tiles, ranks from ballots) and writes the byte mask
// (0 / 1 sampled positive / 2 sampled negative, over the class bytes) plus the compact list of sampled indices in
// ascending order - what torch.where(pos_mask | neg_mask) returns in torchvision.
__global__ __launch_bounds__(1024) void balanced_sample_kernel(const unsigned* __restrict__ keys, unsigned char* __restrict__ mask,
                                                               int n, int batch, int max_pos, int* __restrict__ list,
                                                               int* __restrict__ count) {
    __shared__ int hist[2][2048];
    __shared__ unsigned prefix[2];
    __shared__ int krem[2], total[2];
    __shared__ int w_less[2][16], w_tie[2][16], w_tie_base[2][16], w_base[16];
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    keys += (size_t)b * n;
    mask += (size_t)b * n;
    list += (size_t)b * batch;
    if (tid < 2) { prefix[tid] = 0; krem[tid] = 0; }
    for (int pass = 0; pass < 3; ++pass) {
        const int shift = pass == 0 ? 21 : (pass == 1 ? 10 : 0), bits = pass == 2 ? 10 : 11;
        for (int j = tid; j < 4096; j += 1024) (&hist[0][0])[j] = 0;
        __syncthreads();
        const unsigned p0 = prefix[0], p1 = prefix[1];
        const bool a0 = pass == 0 || krem[0] > 0, a1 = pass == 0 || krem[1] > 0;
        auto vote = [&](int c, unsigned k) {
            if (!c) return;
            const bool in = pass == 0 || (c == 1 ? (a0 && (k >> (shift + bits)) == p0) : (a1 && (k >> (shift + bits)) == p1));
            if (in) atomicAdd(&hist[c - 1][(k >> shift) & ((1u << bits) - 1)], 1);
        };
        if ((n & 3) == 0) {   // rows stay 16-byte / 4-byte aligned: four candidates per load (order is irrelevant here)
            for (int i = tid * 4; i < n; i += 4096) {
                const unsigned c4 = *reinterpret_cast<const unsigned*>(mask + i);
                if (!c4) continue;
                const u32x4 k4 = *reinterpret_cast<const u32x4*>(keys + i);
                vote(c4 & 0xff, k4[0]); vote((c4 >> 8) & 0xff, k4[1]); vote((c4 >> 16) & 0xff, k4[2]); vote(c4 >> 24, k4[3]);
            }
        } else {
            for (int i = tid; i < n; i += 1024) vote(mask[i], keys[i]);
        }
        __syncthreads();
        const int per_lane = (1 << bits) / 64;
        int s = 0, incl = 0;
        if (tid < 128) {
            for (int j = 0; j < per_lane; ++j) s += hist[wave][lane * per_lane + j];
            incl = wave_incl_scan(s);
            if (pass == 0 && lane == 63) total[wave] = incl;
        }
        if (pass == 0) {
            __syncthreads();
            if (tid == 0) {   // det_utils.BalancedPositiveNegativeSampler.__call__: num_pos, then num_neg
                const int kp = min(total[0], max_pos);
                krem[0] = kp;
                krem[1] = min(total[1], batch - kp);
            }
            __syncthreads();
        }
        if (tid < 128) {
            const int k = krem[wave];
            if (k > 0 && incl - s < k && k <= incl) {   // exactly one lane: the crossing is inside its bins
                int cum = incl - s;
                for (int j = 0; j < per_lane; ++j) {
                    const int h = hist[wave][lane * per_lane + j];
                    if (cum + h >= k) {
                        prefix[wave] = (prefix[wave] << bits) | (unsigned)(lane * per_lane + j);
                        krem[wave] = k - cum;
                        break;
                    }
                    cum += h;
                }
            }
        }
        __syncthreads();
    }
    const unsigned T0 = prefix[0], T1 = prefix[1];
    const int r0 = krem[0], r1 = krem[1];   // > 0: ties to take of class 0 / 1; 0: the class takes nothing
    const int seg = ((n + 15) / 16 + 63) & ~63;
    const int lo = wave * seg, hi = min(n, lo + seg);
    int less0 = 0, less1 = 0, tie0 = 0, tie1 = 0;
    auto tally = [&](int c, unsigned k) {
        less0 += (c == 1 && r0 > 0 && k < T0);
        tie0 += (c == 1 && r0 > 0 && k == T0);
        less1 += (c == 2 && r1 > 0 && k < T1);
        tie1 += (c == 2 && r1 > 0 && k == T1);
    };
    const bool vec = (n & 3) == 0;   // then lo, hi are multiples of 4 too
    if (vec) {
        for (int i = lo + lane * 4; i < hi; i += 256) {
            const unsigned c4 = *reinterpret_cast<const unsigned*>(mask + i);
            if (!c4) continue;
            const u32x4 k4 = *reinterpret_cast<const u32x4*>(keys + i);
            tally(c4 & 0xff, k4[0]); tally((c4 >> 8) & 0xff, k4[1]); tally((c4 >> 16) & 0xff, k4[2]); tally(c4 >> 24, k4[3]);
        }
    } else {
        for (int i = lo + lane; i < hi; i += 64) tally(mask[i], keys[i]);
    }
    for (int o = 32; o > 0; o >>= 1) {
        less0 += __shfl_xor(less0, o, 64); less1 += __shfl_xor(less1, o, 64);
        tie0 += __shfl_xor(tie0, o, 64); tie1 += __shfl_xor(tie1, o, 64);
    }
    if (lane == 0) { w_less[0][wave] = less0; w_less[1][wave] = less1; w_tie[0][wave] = tie0; w_tie[1][wave] = tie1; }
    __syncthreads();
    if (tid == 0) {
        int tb0 = 0, tb1 = 0, lb = 0;
        for (int w = 0; w < 16; ++w) {
            w_tie_base[0][w] = tb0; w_tie_base[1][w] = tb1; w_base[w] = lb;
            lb += w_less[0][w] + min(max(r0 - tb0, 0), w_tie[0][w]) + w_less[1][w] + min(max(r1 - tb1, 0), w_tie[1][w]);
            tb0 += w_tie[0][w]; tb1 += w_tie[1][w];
        }
        count[b] = lb;
        total[0] = lb;
    }
    __syncthreads();
    int run_t0 = w_tie_base[0][wave], run_t1 = w_tie_base[1][wave], run_s = w_base[wave];
    const unsigned long long lt = (1ull << lane) - 1ull;
    if (vec) {   // a lane owns 4 consecutive candidates of a 256-wide tile: rank = ties in lower lanes + own earlier ones
        for (int i0 = lo; i0 < hi; i0 += 256) {
            const int i = i0 + lane * 4;
            const bool ok = i < hi;
            const unsigned c4 = ok ? *reinterpret_cast<const unsigned*>(mask + i) : 0u;
            u32x4 k4 = {0u, 0u, 0u, 0u};
            if (c4) k4 = *reinterpret_cast<const u32x4*>(keys + i);
            bool t0[4], t1[4];
            int before0 = run_t0, before1 = run_t1, all0 = 0, all1 = 0;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int c = (c4 >> (8 * j)) & 0xff;
                t0[j] = c == 1 && r0 > 0 && k4[j] == T0;
                t1[j] = c == 2 && r1 > 0 && k4[j] == T1;
                const unsigned long long b0 = __ballot(t0[j]), b1 = __ballot(t1[j]);
                before0 += __popcll(b0 & lt); before1 += __popcll(b1 & lt);
                all0 += __popcll(b0); all1 += __popcll(b1);
            }
            bool sel[4];
            int pos = run_s, alls = 0;
            unsigned out = 0;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int c = (c4 >> (8 * j)) & 0xff;
                sel[j] = (c == 1 && r0 > 0 && (k4[j] < T0 || (t0[j] && before0 < r0))) ||
                         (c == 2 && r1 > 0 && (k4[j] < T1 || (t1[j] && before1 < r1)));
                before0 += t0[j]; before1 += t1[j];
                const unsigned long long bs = __ballot(sel[j]);
                pos += __popcll(bs & lt); alls += __popcll(bs);
                if (sel[j]) out |= (unsigned)c << (8 * j);
            }
            if (ok) *reinterpret_cast<unsigned*>(mask + i) = out;
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (sel[j]) list[pos++] = i + j;
            run_t0 += all0; run_t1 += all1; run_s += alls;
        }
    } else {
        for (int i0 = lo; i0 < hi; i0 += 64) {
            const int i = i0 + lane;
            const bool ok = i < hi;
            const int c = ok ? mask[i] : 0;
            const unsigned k = ok ? keys[i] : 0u;
            const bool t0 = c == 1 && r0 > 0 && k == T0, t1 = c == 2 && r1 > 0 && k == T1;
            const unsigned long long bt0 = __ballot(t0), bt1 = __ballot(t1);
            const bool sel = (c == 1 && r0 > 0 && (k < T0 || (t0 && run_t0 + __popcll(bt0 & lt) < r0))) ||
                             (c == 2 && r1 > 0 && (k < T1 || (t1 && run_t1 + __popcll(bt1 & lt) < r1)));
            const unsigned long long bs = __ballot(sel);
            if (ok) mask[i] = sel ? (unsigned char)c : (unsigned char)0;
            if (sel) list[run_s + __popcll(bs & lt)] = i;
            run_t0 += __popcll(bt0); run_t1 += __popcll(bt1); run_s += __popcll(bs);
        }
    }
    for (int j = total[0] + tid; j < batch; j += 1024) list[j] = 0;
}

// add_gt_proposals (roi_heads.py:  proposals = cat(proposals, gt_boxes)) with static shapes: slots [0, counts) proposals,
// [counts, counts + gt_count) ground truth, the rest zero; box_count = counts + gt_count.
__global__ __launch_bounds__(256) void roi_add_gt_kernel(const float* __restrict__ props, const int* __restrict__ counts, int P,
                                                         const float* __restrict__ gt, const int* __restrict__ gt_count, int G,
                                                         float* __restrict__ boxes, int* __restrict__ box_count) {
    const int b = blockIdx.y, j = blockIdx.x * 256 + threadIdx.x, N = P + G;
    const int cnt = min(counts[b], P), g = min(gt_count[b], G);
    if (j == 0) box_count[b] = cnt + g;
    if (j >= N) return;
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (j < cnt) v = reinterpret_cast<const f32x4*>(props)[(size_t)b * P + j];
    else if (j < cnt + g) v = reinterpret_cast<const f32x4*>(gt)[(size_t)b * G + (j - cnt)];
    reinterpret_cast<f32x4*>(boxes)[(size_t)b * N + j] = v;
}

// The sampled rows of select_training_samples: proposals [B][K][4] (zero padded), offsets [B + 1], and in RoI order
// (row offsets[b] + k) the class labels and the regression targets encode(matched gt, proposal).  grid B, block K.
__global__ __launch_bounds__(1024) void roi_gather_samples_kernel(const float* __restrict__ boxes, const int* __restrict__ matched,
                                                                  const float* __restrict__ gt, const long long* __restrict__ gt_labels,
                                                                  const int* __restrict__ gt_count, int G,
                                                                  const int* __restrict__ list, const int* __restrict__ count, int B,
                                                                  int N, int K, float wx, float wy, float ww, float wh,
                                                                  float* __restrict__ props_s, int* __restrict__ offsets,
                                                                  long long* __restrict__ labels_flat, float* __restrict__ reg) {
    const int b = blockIdx.x;
    int base = 0;
    for (int i = 0; i < b; ++i) base += count[i];
    const int ks = count[b];
    if (threadIdx.x == 0) {
        offsets[b] = base;
        if (b == B - 1) offsets[B] = base + ks;
    }
    for (int k = threadIdx.x; k < K; k += blockDim.x) {
        f32x4 p = {0.f, 0.f, 0.f, 0.f};
        if (k < ks) {
            const int slot = list[(size_t)b * K + k];
            p = reinterpret_cast<const f32x4*>(boxes)[(size_t)b * N + slot];
            const int m = matched[(size_t)b * N + slot];
            const int lab = sample_label(m, gt_labels + (size_t)b * G, G);
            f32x4 r = {0.f, 0.f, 0.f, 0.f};   // images without ground truth: a zero box (roi_heads.py: gt_boxes_in_image = zeros((1, 4)))
            if (gt_count[b] > 0) r = reinterpret_cast<const f32x4*>(gt)[(size_t)b * G + min(max(m, 0), G - 1)];
            labels_flat[base + k] = lab;
            reinterpret_cast<f32x4*>(reg)[base + k] = box_encode1(r, p, wx, wy, ww, wh);
        }
        reinterpret_cast<f32x4*>(props_s)[(size_t)b * K + k] = p;
    }
}

__device__ __forceinline__ double block_sum_d(double v, double* sh) {  // 256 threads, fixed order
    sh[threadIdx.x] = v;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (threadIdx.x < o) sh[threadIdx.x] += sh[threadIdx.x + o];
        __syncthreads();
    }
    const double r = sh[0];
    __syncthreads();
    return r;
}

__device__ __forceinline__ float smooth_l1(float d, float beta) {
    const float a = fabsf(d);
    return a < beta ? 0.5f * d * d / beta : a - 0.5f * beta;
}

// RegionProposalNetwork.compute_loss on the fused RPN head output rpn_out [B * cells][ld]: column a < A_cell is the
// objectness of anchor a of the cell, columns A_cell + 4 a .. + 3 its deltas.  out[0] = BCE-with-logits mean over
// the sampled anchors, out[1] = smooth-L1 (beta 1/9) sum over the sampled positives / number of sampled anchors.
// Driven by the sampler's compact list (list [B][batch] ascending anchor indices, count [B]; mask gives the
// class): the regression target of a sampled positive - encode(matched gt, anchor), weights 1 - is computed here for the
// <= batch / 2 anchors per image that need it instead of for all B x A.
__global__ __launch_bounds__(256) void rpn_loss_sampled_kernel(const float* __restrict__ rpn_out, int ld, int A_cell,
                                                               const int* __restrict__ matched, const float* __restrict__ gt, int G,
                                                               const float* __restrict__ anchors, const unsigned char* __restrict__ mask,
                                                               const int* __restrict__ list, const int* __restrict__ count, int B,
                                                               int A, int batch, float* __restrict__ out) {
    __shared__ double sh[256];
    double bce = 0.0, box = 0.0, cnt = 0.0;
    for (int e = threadIdx.x; e < B * batch; e += 256) {
        const int b = e / batch, k = e % batch;
        if (k >= count[b]) continue;
        const int i = list[e];
        const size_t at = (size_t)b * A + i;
        const unsigned char sm = mask[at];
        const size_t cell = at / A_cell;
        const int a = (int)(at % A_cell);
        const float x = rpn_out[cell * ld + a];
        const float z = sm == 1 ? 1.f : 0.f;
        bce += (double)(fmaxf(x, 0.f) - x * z + log1pf(expf(-fabsf(x))));
        cnt += 1.0;
        if (sm == 1) {
            const int m = matched[at];
            const f32x4 t = box_encode1(reinterpret_cast<const f32x4*>(gt)[(size_t)b * G + min(max(m, 0), G - 1)],
                                        reinterpret_cast<const f32x4*>(anchors)[i], 1.f, 1.f, 1.f, 1.f);
            for (int c = 0; c < 4; ++c) box += (double)smooth_l1(rpn_out[cell * ld + A_cell + a * 4 + c] - t[c], 1.0f / 9.0f);
        }
    }
    bce = block_sum_d(bce, sh);
    box = block_sum_d(box, sh);
    cnt = block_sum_d(cnt, sh);
    if (threadIdx.x == 0) {
        out[0] = cnt > 0.0 ? (float)(bce / cnt) : nanf("");
        out[1] = cnt > 0.0 ? (float)(box / cnt) : nanf("");
    }
}

// roi_heads.fastrcnn_loss on pred [N][ld] = 30 class logits | 30 x 4 deltas: out[0] = cross entropy (mean),
// out[1] = smooth-L1 (beta 1/9) over the positive rows' own class deltas, sum / N.
__global__ __launch_bounds__(256) void fastrcnn_loss_kernel(const float* __restrict__ pred, int ld, int C,
                                                            const long long* __restrict__ labels, const float* __restrict__ reg,
                                                            int N, float* __restrict__ out) {
    __shared__ double sh[256];
    double ce = 0.0, box = 0.0;
    for (int i = threadIdx.x; i < N; i += 256) {
        const float* x = pred + (size_t)i * ld;
        float m = x[0];
        for (int c = 1; c < C; ++c) m = fmaxf(m, x[c]);
        float s = 0.f;
        for (int c = 0; c < C; ++c) s += expf(x[c] - m);
        const int lab = (int)labels[i];
        ce += (double)((m + logf(s)) - x[lab]);
        if (lab > 0) {
            for (int c = 0; c < 4; ++c) box += (double)smooth_l1(x[C + lab * 4 + c] - reg[(size_t)i * 4 + c], 1.0f / 9.0f);
        }
    }
    ce = block_sum_d(ce, sh);
    box = block_sum_d(box, sh);
    if (threadIdx.x == 0) {
        out[0] = N ? (float)(ce / N) : nanf("");
        out[1] = N ? (float)(box / N) : nanf("");
    }
}

}  // namespace rgrg

using namespace rgrg;

extern "C" int rgrg_box_match_f32(const float* gt, const int* gt_count, int G, const float* boxes, int64_t box_image_stride,
                                  const int* box_count, int B, int N, float high, float low, int allow_low_quality,
                                  int* matched, int* ws_best_per_gt, void* stream) {
    RGRG_CHECK_ARG(gt && gt_count && boxes && matched && ws_best_per_gt && B > 0 && N > 0 && G > 0 && box_image_stride >= 0);
    hipStream_t st = as_stream(stream);
    RGRG_HIP(hipMemsetAsync(ws_best_per_gt, 0xff, (size_t)B * G * sizeof(int), st));  // -1 < any IoU bit pattern
    const dim3 grid((N + 255) / 256, B);
    hipLaunchKernelGGL(box_match_kernel, grid, dim3(256), (size_t)G * sizeof(int), st, gt, gt_count, G, boxes, (size_t)box_image_stride, box_count, N,
                       high, low, matched, ws_best_per_gt);
    RGRG_LAUNCH_CHECK();
    if (allow_low_quality) {
        hipLaunchKernelGGL(box_match_low_quality_kernel, grid, dim3(256), 0, st, gt, gt_count, G, boxes, (size_t)box_image_stride,
                           box_count, N, matched, ws_best_per_gt);
        RGRG_LAUNCH_CHECK();
    }
    return RGRG_OK;
}

extern "C" int rgrg_fastrcnn_loss_f32(const float* pred, int ld, int num_classes, const int64_t* labels, const float* reg_targets,
                                      int N, float* out2, void* stream) {
    RGRG_CHECK_ARG(pred && out2 && N >= 0 && num_classes > 0 && ld >= num_classes * 5 && (N == 0 || (labels && reg_targets)));
    hipLaunchKernelGGL(fastrcnn_loss_kernel, dim3(1), dim3(256), 0, as_stream(stream), pred, ld, num_classes,
                       reinterpret_cast<const long long*>(labels), reg_targets, N, out2);
    RGRG_LAUNCH_CHECK();
    return RGRG_OK;
}

extern "C" int rgrg_balanced_sample(const int* matched, const int64_t* gt_labels, int G, const int* box_count, const float* keys,
                                    uint64_t seed, int stage, int B, int n, int batch, int max_pos, uint32_t* ws_keys,
                                    uint8_t* mask, int* list, int* count, void* stream) {
    RGRG_CHECK_ARG(matched && ws_keys && mask && list && count && B > 0 && n > 0 && batch > 0 && max_pos >= 0 && max_pos <= batch &&
                   (!gt_labels || G > 0));
    hipStream_t st = as_stream(stream);
    hipLaunchKernelGGL(sample_prepare_kernel, dim3((n + 255) / 256, B), dim3(256), 0, st, matched,
                       reinterpret_cast<const long long*>(gt_labels), G, box_count, keys, (unsigned)seed, (unsigned)(seed >> 32),
                       (unsigned)stage, n, ws_keys, mask);
    RGRG_LAUNCH_CHECK();
    hipLaunchKernelGGL(balanced_sample_kernel, dim3(B), dim3(1024), 0, st, ws_keys, mask, n, batch, max_pos, list, count);
    RGRG_LAUNCH_CHECK();
    return RGRG_OK;
}

extern "C" int rgrg_rpn_loss_sampled_f32(const float* rpn_out, int ld, int anchors_per_cell, const int* matched, const float* gt,
                                         int G, const float* anchors, const uint8_t* mask, const int* list, const int* count,
                                         int B, int A, int batch, float* out2, void* stream) {
    RGRG_CHECK_ARG(rpn_out && matched && gt && anchors && mask && list && count && out2 && B > 0 && A > 0 && G > 0 && batch > 0 &&
                   anchors_per_cell > 0 && ld >= anchors_per_cell * 5);
    hipLaunchKernelGGL(rpn_loss_sampled_kernel, dim3(1), dim3(256), 0, as_stream(stream), rpn_out, ld, anchors_per_cell, matched, gt,
                       G, anchors, mask, list, count, B, A, batch, out2);
    RGRG_LAUNCH_CHECK();
    return RGRG_OK;
}

extern "C" int rgrg_roi_add_gt_f32(const float* props, const int* counts, int P, const float* gt, const int* gt_count, int G, int B,
                                   float* boxes, int* box_count, void* stream) {
    RGRG_CHECK_ARG(props && counts && gt && gt_count && boxes && box_count && P > 0 && G > 0 && B > 0);
    hipLaunchKernelGGL(roi_add_gt_kernel, dim3((P + G + 255) / 256, B), dim3(256), 0, as_stream(stream), props, counts, P, gt,
                       gt_count, G, boxes, box_count);
    RGRG_LAUNCH_CHECK();
    return RGRG_OK;
}

extern "C" int rgrg_roi_gather_samples_f32(const float* boxes, const int* matched, const float* gt, const int64_t* gt_labels,
                                           const int* gt_count, int G, const int* list, const int* count, int B, int N, int K,
                                           float wx, float wy, float ww, float wh, float* props_s, int* offsets,
                                           int64_t* labels_flat, float* reg_targets, void* stream) {
    RGRG_CHECK_ARG(boxes && matched && gt && gt_labels && gt_count && list && count && props_s && offsets && labels_flat &&
                   reg_targets && B > 0 && N > 0 && K > 0 && G > 0);
    hipLaunchKernelGGL(roi_gather_samples_kernel, dim3(B), dim3(K < 1024 ? ((K + 63) & ~63) : 1024), 0, as_stream(stream), boxes,
                       matched, gt, reinterpret_cast<const long long*>(gt_labels), gt_count, G, list, count, B, N, K, wx, wy, ww,
                       wh, props_s, offsets, reinterpret_cast<long long*>(labels_flat), reg_targets);
    RGRG_LAUNCH_CHECK();
    return RGRG_OK;
}
