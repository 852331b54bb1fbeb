// Detector targets and losses for gfx950: what torchvision 0.13.1's RegionProposalNetwork / RoIHeads compute when
// ``targets`` are given (custom_rpn.py:74-83 -> assign_targets_to_anchors / box_coder.encode / compute_loss;
// custom_roi_heads.py:225-242 -> select_training_samples / fastrcnn_loss), i.e. the detector half of
// ReportGenerationModel.forward(images, image_targets, ...) (report_generation_model.py:55,91) that the reference's
// validation loop calls in eval mode (evaluate_model.py:413).
//
// Split of the work: the O(gt x boxes) matching, the box encoding and the four loss reductions run here; the random
// sampling (torch.randperm in torchvision - injectable for tests) and the index gathers around them are integer
// plumbing done by the caller on small index tensors.  Everything is fp32 with IEEE division and no contraction, in
// torchvision's operation order, so match decisions (IoU >= threshold, IoU == row maximum) agree with a CPU evaluation.
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
// all boxes of the image (atomicMax on the bit pattern: IoUs are >= 0, so float order == int order).
__global__ __launch_bounds__(256) void box_match_kernel(const float* __restrict__ gt, const int* __restrict__ gt_count, int G,
                                                        const float* __restrict__ boxes, size_t box_image_stride,
                                                        const int* __restrict__ box_count, int N, float high, float low,
                                                        int* __restrict__ matched, int* __restrict__ best_per_gt) {
    const int b = blockIdx.y, i = blockIdx.x * 256 + threadIdx.x;
    const int ng = gt_count[b], nb = box_count ? box_count[b] : N;
    if (i >= N) return;
    if (i >= nb || ng == 0) { matched[(size_t)b * N + i] = MATCH_BELOW; return; }
    const f32x4 bx = *reinterpret_cast<const f32x4*>(boxes + (size_t)b * box_image_stride + (size_t)i * 4);
    float best = -1.f;
    int arg = 0;
    for (int g = 0; g < ng; ++g) {
        const float q = box_iou1(*reinterpret_cast<const f32x4*>(gt + ((size_t)b * G + g) * 4), bx);
        if (q > best) { best = q; arg = g; }
        atomicMax(best_per_gt + (size_t)b * G + g, __float_as_int(q));
    }
    matched[(size_t)b * N + i] = best < low ? MATCH_BELOW : (best < high ? MATCH_BETWEEN : arg);
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

// det_utils.encode_boxes
__global__ __launch_bounds__(256) void box_encode_kernel(const float* __restrict__ ref, const float* __restrict__ prop, int n,
                                                         float wx, float wy, float ww, float wh, float* __restrict__ out) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const f32x4 r = reinterpret_cast<const f32x4*>(ref)[i], p = reinterpret_cast<const f32x4*>(prop)[i];
    const float ex_w = p[2] - p[0], ex_h = p[3] - p[1];
    const float ex_cx = p[0] + 0.5f * ex_w, ex_cy = p[1] + 0.5f * ex_h;
    const float gt_w = r[2] - r[0], gt_h = r[3] - r[1];
    const float gt_cx = r[0] + 0.5f * gt_w, gt_cy = r[1] + 0.5f * gt_h;
    f32x4 o;
    o[0] = wx * (gt_cx - ex_cx) / ex_w;
    o[1] = wy * (gt_cy - ex_cy) / ex_h;
    o[2] = ww * logf(gt_w / ex_w);
    o[3] = wh * logf(gt_h / ex_h);
    reinterpret_cast<f32x4*>(out)[i] = o;
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
__global__ __launch_bounds__(256) void rpn_loss_kernel(const float* __restrict__ rpn_out, int ld, int A_cell,
                                                       const float* __restrict__ labels, const float* __restrict__ reg,
                                                       const unsigned char* __restrict__ sampled, long long total,
                                                       float* __restrict__ out) {
    __shared__ double sh[256];
    double bce = 0.0, box = 0.0, cnt = 0.0;
    for (long long i = threadIdx.x; i < total; i += 256) {
        const unsigned char sm = sampled[i];   // 0 not sampled, 1 sampled positive, 2 sampled negative
        if (!sm) continue;
        const size_t cell = (size_t)(i / A_cell);
        const int a = (int)(i % A_cell);
        const float x = rpn_out[cell * ld + a];
        const float z = labels[i];
        bce += (double)(fmaxf(x, 0.f) - x * z + log1pf(expf(-fabsf(x))));
        cnt += 1.0;
        if (sm == 1) {
            for (int c = 0; c < 4; ++c)
                box += (double)smooth_l1(rpn_out[cell * ld + A_cell + a * 4 + c] - reg[(size_t)i * 4 + c], 1.0f / 9.0f);
        }
    }
    bce = block_sum_d(bce, sh);
    box = block_sum_d(box, sh);
    cnt = block_sum_d(cnt, sh);
    if (threadIdx.x == 0) {
        out[0] = cnt > 0.0 ? (float)(bce / cnt) : nanf("");  // mean over an empty set is nan, like torch
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
    hipLaunchKernelGGL(box_match_kernel, grid, dim3(256), 0, st, gt, gt_count, G, boxes, (size_t)box_image_stride, box_count, N,
                       high, low, matched, ws_best_per_gt);
    RGRG_LAUNCH_CHECK();
    if (allow_low_quality) {
        hipLaunchKernelGGL(box_match_low_quality_kernel, grid, dim3(256), 0, st, gt, gt_count, G, boxes, (size_t)box_image_stride,
                           box_count, N, matched, ws_best_per_gt);
        RGRG_LAUNCH_CHECK();
    }
    return RGRG_OK;
}

extern "C" int rgrg_box_encode_f32(const float* ref_boxes, const float* proposals, int n, float wx, float wy, float ww, float wh,
                                   float* out, void* stream) {
    RGRG_CHECK_ARG(ref_boxes && proposals && out && n >= 0);
    if (n == 0) return RGRG_OK;
    hipLaunchKernelGGL(box_encode_kernel, dim3((n + 255) / 256), dim3(256), 0, as_stream(stream), ref_boxes, proposals, n, wx, wy,
                       ww, wh, out);
    RGRG_LAUNCH_CHECK();
    return RGRG_OK;
}

extern "C" int rgrg_rpn_loss_f32(const float* rpn_out, int ld, int anchors_per_cell, const float* labels, const float* reg_targets,
                                 const uint8_t* sampled, int64_t total, float* out2, void* stream) {
    RGRG_CHECK_ARG(rpn_out && labels && reg_targets && sampled && out2 && total >= 0 && ld >= anchors_per_cell * 5);
    hipLaunchKernelGGL(rpn_loss_kernel, dim3(1), dim3(256), 0, as_stream(stream), rpn_out, ld, anchors_per_cell, labels, reg_targets,
                       sampled, (long long)total, out2);
    RGRG_LAUNCH_CHECK();
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
