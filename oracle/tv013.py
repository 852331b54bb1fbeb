"""Functional restatement of the torchvision==0.13.1 arithmetic the reference
delegates to.  TEST INFRASTRUCTURE ONLY (see ``oracle/__init__.py``).

torchvision 0.13.1 is a third-party dependency of the reference
(``/root/reference/environment.yml:46``, ``requirements.txt:23``) that is NOT
vendored under ``/root/reference`` and is not installable in the build image.
Each function restates the published 0.13.1 algorithm and cites the reference
CALL SITE that fixes its configuration.  PARITY UNPINNED against a real
torchvision install; pinned only by the hand-computable known-answer tests in
``tests/test_oracle_kats.py``.

Everything is plain torch-CPU fp32 on tensors; weights come from a state dict
that uses the reference's own key names.
"""
from __future__ import annotations

import math
from typing import Dict, List, Tuple

import torch
import torch.nn.functional as F

Tensor = torch.Tensor
SD = Dict[str, Tensor]

# ---------------------------------------------------------------------------
# ResNet-50 trunk  (call site: src/object_detector/object_detector.py:51-62,219)
# ---------------------------------------------------------------------------
RESNET50_LAYERS = ((64, 3, 1), (128, 4, 2), (256, 6, 2), (512, 3, 2))  # planes, blocks, stride
BN_EPS = 1e-5


def _bn_eval(sd: SD, p: str, x: Tensor) -> Tensor:
    """nn.BatchNorm2d in eval mode: running statistics, eps 1e-5."""
    return F.batch_norm(x, sd[p + "running_mean"], sd[p + "running_var"], sd[p + "weight"], sd[p + "bias"],
                        training=False, eps=BN_EPS)


def _bottleneck(sd: SD, p: str, x: Tensor, stride: int) -> Tensor:
    """torchvision Bottleneck (v1.5: the stride sits on the 3x3 conv)."""
    out = F.relu(_bn_eval(sd, p + "bn1.", F.conv2d(x, sd[p + "conv1.weight"])))
    out = F.relu(_bn_eval(sd, p + "bn2.", F.conv2d(out, sd[p + "conv2.weight"], stride=stride, padding=1)))
    out = _bn_eval(sd, p + "bn3.", F.conv2d(out, sd[p + "conv3.weight"]))
    if (p + "downsample.0.weight") in sd:
        idt = _bn_eval(sd, p + "downsample.1.", F.conv2d(x, sd[p + "downsample.0.weight"], stride=stride))
    else:
        idt = x
    return F.relu(out + idt)


def resnet50_trunk(sd: SD, p: str, images: Tensor) -> Tensor:
    """``nn.Sequential(*list(resnet50.children())[:-2])`` with a 1-channel 7x7
    stem (object_detector.py:51-58).  ``p`` is the key prefix of the Sequential,
    e.g. ``"object_detector.backbone."``; children are index-named
    0=conv1 1=bn1 2=relu 3=maxpool 4..7=layer1..4.
    [B,1,512,512] -> [B,2048,16,16]."""
    x = F.conv2d(images, sd[p + "0.weight"], stride=2, padding=3)
    x = F.relu(_bn_eval(sd, p + "1.", x))
    x = F.max_pool2d(x, kernel_size=3, stride=2, padding=1)
    for li, (_planes, blocks, stride) in enumerate(RESNET50_LAYERS):
        for b in range(blocks):
            x = _bottleneck(sd, f"{p}{4 + li}.{b}.", x, stride if b == 0 else 1)
    return x


# ---------------------------------------------------------------------------
# AnchorGenerator  (call site: object_detector.py:78-81)
# ---------------------------------------------------------------------------
ANCHOR_SIZES = (20, 40, 60, 80, 100, 120, 140, 160, 180, 300)
ANCHOR_RATIOS = (0.2, 0.25, 0.4, 0.5, 0.6, 0.7, 0.8, 0.9, 1.0, 1.3, 1.5, 2.1, 2.6, 3.0, 5.0, 8.0)


def base_anchors(sizes=ANCHOR_SIZES, ratios=ANCHOR_RATIOS) -> Tensor:
    """generate_anchors(): zero-centred [len(ratios)*len(sizes), 4] anchors,
    index = ratio_idx*len(sizes) + size_idx, rounded half-to-even."""
    scales = torch.tensor(sizes, dtype=torch.float32)
    ar = torch.tensor(ratios, dtype=torch.float32)
    h_r = torch.sqrt(ar)
    w_r = 1.0 / h_r
    ws = (w_r[:, None] * scales[None, :]).reshape(-1)
    hs = (h_r[:, None] * scales[None, :]).reshape(-1)
    return (torch.stack([-ws, -hs, ws, hs], dim=1) / 2).round()


def grid_anchors(image_size: Tuple[int, int], grid: Tuple[int, int]) -> Tensor:
    """All anchors of one image: flat index = (y*W_g + x)*A + a; stride =
    image // grid (integer division), no half-stride offset."""
    gh, gw = grid
    sh, sw = image_size[0] // gh, image_size[1] // gw
    base = base_anchors()
    sx = torch.arange(0, gw, dtype=torch.int32) * sw
    sy = torch.arange(0, gh, dtype=torch.int32) * sh
    yy, xx = torch.meshgrid(sy, sx, indexing="ij")
    xx, yy = xx.reshape(-1), yy.reshape(-1)
    shifts = torch.stack((xx, yy, xx, yy), dim=1).to(torch.float32)
    return (shifts.view(-1, 1, 4) + base.view(1, -1, 4)).reshape(-1, 4)


# ---------------------------------------------------------------------------
# RPN head + proposal filtering  (call sites: custom_rpn.py:61-71,
# object_detector.py:83-97)
# ---------------------------------------------------------------------------
def rpn_head(sd: SD, p: str, feat: Tensor) -> Tuple[Tensor, Tensor]:
    """0.13 RPNHead: conv = Sequential(Conv2dNormActivation(C,C,3,norm=None))
    -> keys ``conv.0.0.*``; cls_logits 1x1 -> A; bbox_pred 1x1 -> 4A."""
    t = F.relu(F.conv2d(feat, sd[p + "conv.0.0.weight"], sd[p + "conv.0.0.bias"], padding=1))
    obj = F.conv2d(t, sd[p + "cls_logits.weight"], sd[p + "cls_logits.bias"])
    reg = F.conv2d(t, sd[p + "bbox_pred.weight"], sd[p + "bbox_pred.bias"])
    return obj, reg


def permute_and_flatten(layer: Tensor, C: int) -> Tensor:
    """concat_box_prediction_layers for one level:
    [N, A*C, H, W] -> [N, H*W*A, C] (same (y,x,a) order as the anchors)."""
    N, AC, H, W = layer.shape
    A = AC // C
    return layer.view(N, A, C, H, W).permute(0, 3, 4, 1, 2).reshape(N, -1, C)


BBOX_XFORM_CLIP = math.log(1000.0 / 16)


def box_decode(deltas: Tensor, boxes: Tensor, weights: Tuple[float, float, float, float]) -> Tensor:
    """BoxCoder.decode_single: deltas [N, 4*k] (k box sets per row), boxes [N,4]
    -> [N, 4*k]."""
    boxes = boxes.to(deltas.dtype)
    widths = boxes[:, 2] - boxes[:, 0]
    heights = boxes[:, 3] - boxes[:, 1]
    ctr_x = boxes[:, 0] + 0.5 * widths
    ctr_y = boxes[:, 1] + 0.5 * heights
    wx, wy, ww, wh = weights
    dx = deltas[:, 0::4] / wx
    dy = deltas[:, 1::4] / wy
    dw = torch.clamp(deltas[:, 2::4] / ww, max=BBOX_XFORM_CLIP)
    dh = torch.clamp(deltas[:, 3::4] / wh, max=BBOX_XFORM_CLIP)
    pcx = dx * widths[:, None] + ctr_x[:, None]
    pcy = dy * heights[:, None] + ctr_y[:, None]
    pw = torch.exp(dw) * widths[:, None]
    ph = torch.exp(dh) * heights[:, None]
    c2c_h = torch.tensor(0.5, dtype=pcy.dtype) * ph
    c2c_w = torch.tensor(0.5, dtype=pcx.dtype) * pw
    x1, y1, x2, y2 = pcx - c2c_w, pcy - c2c_h, pcx + c2c_w, pcy + c2c_h
    return torch.stack((x1, y1, x2, y2), dim=2).flatten(1)


def clip_boxes_to_image(boxes: Tensor, size: Tuple[int, int]) -> Tensor:
    h, w = size
    bx = boxes[..., 0::2].clamp(min=0, max=w)
    by = boxes[..., 1::2].clamp(min=0, max=h)
    return torch.stack((bx, by), dim=boxes.dim()).reshape(boxes.shape)


def nms(boxes: Tensor, scores: Tensor, thr: float) -> Tensor:
    """torchvision CPU nms kernel: stable descending sort by score, greedy
    suppression of ``inter/(a_i+a_j-inter) > thr``; returns kept indices in
    score order."""
    n = boxes.shape[0]
    if n == 0:
        return torch.empty(0, dtype=torch.int64)
    order = torch.sort(scores, descending=True, stable=True).indices
    b = boxes[order]
    x1, y1, x2, y2 = b.unbind(1)
    areas = (x2 - x1) * (y2 - y1)
    suppressed = torch.zeros(n, dtype=torch.bool)
    keep: List[int] = []
    zero = torch.zeros((), dtype=boxes.dtype)
    for i in range(n):
        if suppressed[i]:
            continue
        keep.append(i)
        if i + 1 == n:
            break
        xx1 = torch.maximum(x1[i], x1[i + 1:])
        yy1 = torch.maximum(y1[i], y1[i + 1:])
        xx2 = torch.minimum(x2[i], x2[i + 1:])
        yy2 = torch.minimum(y2[i], y2[i + 1:])
        w = torch.maximum(zero, xx2 - xx1)
        h = torch.maximum(zero, yy2 - yy1)
        inter = w * h
        ovr = inter / (areas[i] + areas[i + 1:] - inter)
        suppressed[i + 1:] |= ovr > thr
    return order[torch.tensor(keep, dtype=torch.int64)]


def topk_stable(scores: Tensor, k: int) -> Tensor:
    """Top-k indices, descending, ties broken by ascending index.  torch.topk's
    tie order is unspecified; both the oracle and the HIP kernel use this
    (stable) definition."""
    return torch.sort(scores, descending=True, stable=True).indices[:k]


def filter_proposals(proposals: Tensor, objectness: Tensor, image_size: Tuple[int, int],
                     pre_nms_top_n: int = 1000, post_nms_top_n: int = 1000, nms_thresh: float = 0.7,
                     score_thresh: float = 0.0, min_size: float = 1e-3) -> Tuple[List[Tensor], List[Tensor]]:
    """RegionProposalNetwork.filter_proposals, eval, single feature level.
    proposals [B,A,4], objectness [B,A] (logits)."""
    out_boxes, out_scores = [], []
    for b in range(proposals.shape[0]):
        idx = topk_stable(objectness[b], min(pre_nms_top_n, objectness.shape[1]))
        scores = torch.sigmoid(objectness[b][idx])
        boxes = clip_boxes_to_image(proposals[b][idx], image_size)
        ws, hs = boxes[:, 2] - boxes[:, 0], boxes[:, 3] - boxes[:, 1]
        keep = torch.where((ws >= min_size) & (hs >= min_size))[0]
        boxes, scores = boxes[keep], scores[keep]
        keep = torch.where(scores >= score_thresh)[0]
        boxes, scores = boxes[keep], scores[keep]
        keep = nms(boxes, scores, nms_thresh)[:post_nms_top_n]
        out_boxes.append(boxes[keep])
        out_scores.append(scores[keep])
    return out_boxes, out_scores


# ---------------------------------------------------------------------------
# roi_align (aligned=False)  (call site: object_detector.py:105-106,
# custom_roi_heads.py:232)
# ---------------------------------------------------------------------------
def roi_align(feat: Tensor, rois: Tensor, spatial_scale: float, out_size: int = 8, sampling_ratio: int = 2,
              chunk: int = 64) -> Tensor:
    """torchvision.ops.roi_align, aligned=False.  feat [B,C,H,W], rois [K,5]
    (batch_idx,x1,y1,x2,y2) -> [K,C,out,out]."""
    K = rois.shape[0]
    B, C, H, W = feat.shape
    P, G = out_size, sampling_ratio
    out = feat.new_zeros((K, C, P, P))
    if K == 0:
        return out
    bidx = rois[:, 0].to(torch.int64)
    x1, y1 = rois[:, 1] * spatial_scale, rois[:, 2] * spatial_scale
    x2, y2 = rois[:, 3] * spatial_scale, rois[:, 4] * spatial_scale
    roi_w = torch.clamp(x2 - x1, min=1.0)
    roi_h = torch.clamp(y2 - y1, min=1.0)
    bin_w, bin_h = roi_w / P, roi_h / P
    pidx = torch.arange(P, dtype=feat.dtype)
    gidx = torch.arange(G, dtype=feat.dtype)
    # sample coordinates [K, P, G]
    ys = y1[:, None, None] + pidx[None, :, None] * bin_h[:, None, None] + (gidx[None, None, :] + 0.5) * bin_h[:, None, None] / G
    xs = x1[:, None, None] + pidx[None, :, None] * bin_w[:, None, None] + (gidx[None, None, :] + 0.5) * bin_w[:, None, None] / G

    def prep(v: Tensor, size: int):
        oob = (v < -1.0) | (v > size)
        v = torch.clamp(v, min=0.0)
        lo = v.to(torch.int64)  # truncation, v >= 0
        top = lo >= size - 1
        lo = torch.where(top, torch.full_like(lo, size - 1), lo)
        hi = torch.where(top, torch.full_like(lo, size - 1), lo + 1)
        v = torch.where(top, lo.to(v.dtype), v)
        l = v - lo.to(v.dtype)
        return lo, hi, l, 1.0 - l, oob

    ylo, yhi, ly, hy, yo = prep(ys.reshape(K, P * G), H)
    xlo, xhi, lx, hx, xo = prep(xs.reshape(K, P * G), W)
    flat = feat.reshape(B, C, H * W)
    for s in range(0, K, chunk):
        e = min(K, s + chunk)
        f = flat[bidx[s:e]]  # [k,C,HW]
        k = e - s

        def gather(yi, xi):
            ind = (yi[:, :, None] * W + xi[:, None, :]).reshape(k, 1, -1).expand(k, C, -1)
            return torch.gather(f, 2, ind).reshape(k, C, P * G, P * G)

        w1 = (hy[s:e, :, None] * hx[s:e, None, :])[:, None]
        w2 = (hy[s:e, :, None] * lx[s:e, None, :])[:, None]
        w3 = (ly[s:e, :, None] * hx[s:e, None, :])[:, None]
        w4 = (ly[s:e, :, None] * lx[s:e, None, :])[:, None]
        val = (w1 * gather(ylo[s:e], xlo[s:e]) + w2 * gather(ylo[s:e], xhi[s:e])
               + w3 * gather(yhi[s:e], xlo[s:e]) + w4 * gather(yhi[s:e], xhi[s:e]))
        dead = (yo[s:e, :, None] | xo[s:e, None, :])[:, None]
        val = torch.where(dead, torch.zeros((), dtype=val.dtype), val)
        val = val.reshape(k, C, P, G, P, G)
        acc = val[:, :, :, 0, :, 0]
        for iy in range(G):
            for ix in range(G):
                if iy or ix:
                    acc = acc + val[:, :, :, iy, :, ix]
        out[s:e] = acc / float(G * G)
    return out


def infer_scale(feat_size: int, image_size: int) -> float:
    """MultiScaleRoIAlign.infer_scale: 2**round(log2(feat/image))."""
    return 2.0 ** float(torch.tensor(float(feat_size) / float(image_size)).log2().round())


# ---------------------------------------------------------------------------
# box head / predictor  (call sites: object_detector.py:111-112,
# custom_roi_heads.py:235-236)
# ---------------------------------------------------------------------------
def two_mlp_head(sd: SD, p: str, x: Tensor) -> Tensor:
    x = x.flatten(start_dim=1)
    x = F.relu(F.linear(x, sd[p + "fc6.weight"], sd[p + "fc6.bias"]))
    return F.relu(F.linear(x, sd[p + "fc7.weight"], sd[p + "fc7.bias"]))


def fastrcnn_predictor(sd: SD, p: str, x: Tensor) -> Tuple[Tensor, Tensor]:
    x = x.flatten(start_dim=1)
    return (F.linear(x, sd[p + "cls_score.weight"], sd[p + "cls_score.bias"]),
            F.linear(x, sd[p + "bbox_pred.weight"], sd[p + "bbox_pred.bias"]))


# ---------------------------------------------------------------------------------------------------------------------
# Targets and losses (torchvision 0.13.1 ``models/detection/_utils.py``, ``rpn.py``, ``roi_heads.py``, ``ops/boxes.py``):
# what RegionProposalNetwork / RoIHeads do when ``targets`` are given - in EVAL mode too, which is how the reference's
# validation loop calls the model (evaluate_model.py:413; custom_rpn.py:74-83; custom_roi_heads.py:225-242).
# torchvision samples anchors / proposals with torch.randperm; here the permutations come from ``perm_fn(n, tag)`` so
# that a test can give the oracle and the HIP path the SAME draws (the default draws with torch.randperm like
# torchvision).  Third-party semantics, restated from the published source: parity unpinned.
BELOW_LOW_THRESHOLD, BETWEEN_THRESHOLDS = -1, -2


def box_iou(boxes1: Tensor, boxes2: Tensor) -> Tensor:
    """ops.boxes.box_iou: [N,4] x [M,4] -> [N,M]."""
    area1 = (boxes1[:, 2] - boxes1[:, 0]) * (boxes1[:, 3] - boxes1[:, 1])
    area2 = (boxes2[:, 2] - boxes2[:, 0]) * (boxes2[:, 3] - boxes2[:, 1])
    lt = torch.max(boxes1[:, None, :2], boxes2[:, :2])
    rb = torch.min(boxes1[:, None, 2:], boxes2[:, 2:])
    wh = (rb - lt).clamp(min=0)
    inter = wh[:, :, 0] * wh[:, :, 1]
    return inter / (area1[:, None] + area2 - inter)


def matcher(mqm: Tensor, high: float, low: float, allow_low_quality_matches: bool) -> Tensor:
    """det_utils.Matcher.__call__ on a [num_gt, num_boxes] quality matrix -> per box: gt index, -1 (below low) or -2."""
    matched_vals, matches = mqm.max(dim=0)
    all_matches = matches.clone()
    below = matched_vals < low
    between = (matched_vals >= low) & (matched_vals < high)
    matches[below] = BELOW_LOW_THRESHOLD
    matches[between] = BETWEEN_THRESHOLDS
    if allow_low_quality_matches:  # set_low_quality_matches_: every box that ties a gt's best IoU keeps its arg-max match
        highest, _ = mqm.max(dim=1)
        pred_inds = torch.where(mqm == highest[:, None])[1]
        matches[pred_inds] = all_matches[pred_inds]
    return matches


def box_encode(reference_boxes: Tensor, proposals: Tensor, weights: Tuple[float, float, float, float]) -> Tensor:
    """det_utils.encode_boxes."""
    wx, wy, ww, wh = weights
    px1, py1, px2, py2 = proposals.unbind(1)
    rx1, ry1, rx2, ry2 = reference_boxes.unbind(1)
    ex_w, ex_h = px2 - px1, py2 - py1
    ex_cx, ex_cy = px1 + 0.5 * ex_w, py1 + 0.5 * ex_h
    gt_w, gt_h = rx2 - rx1, ry2 - ry1
    gt_cx, gt_cy = rx1 + 0.5 * gt_w, ry1 + 0.5 * gt_h
    return torch.stack((wx * (gt_cx - ex_cx) / ex_w, wy * (gt_cy - ex_cy) / ex_h,
                        ww * torch.log(gt_w / ex_w), wh * torch.log(gt_h / ex_h)), dim=1)


def default_perm(n: int, tag) -> Tensor:
    return torch.randperm(n)


def balanced_sample(labels: Tensor, batch_size_per_image: int, positive_fraction: float, perm_fn, tag) -> Tuple[Tensor, Tensor]:
    """det_utils.BalancedPositiveNegativeSampler for one image -> (positive indices, negative indices) as drawn."""
    positive = torch.where(labels >= 1)[0]
    negative = torch.where(labels == 0)[0]
    num_pos = min(positive.numel(), int(batch_size_per_image * positive_fraction))
    num_neg = min(negative.numel(), batch_size_per_image - num_pos)
    perm1 = perm_fn(positive.numel(), (tag, "pos"))[:num_pos]
    perm2 = perm_fn(negative.numel(), (tag, "neg"))[:num_neg]
    return positive[perm1], negative[perm2]


def assign_targets_to_anchors(anchors: List[Tensor], targets) -> Tuple[List[Tensor], List[Tensor]]:
    """RegionProposalNetwork.assign_targets_to_anchors, fg 0.7 / bg 0.3 with low-quality matches (object_detector.py:87-88)."""
    labels, matched_gt_boxes = [], []
    for anc, t in zip(anchors, targets):
        gt = t["boxes"].to(torch.float32)
        if gt.numel() == 0:
            matched_gt_boxes.append(torch.zeros_like(anc))
            labels.append(torch.zeros((anc.shape[0],), dtype=torch.float32))
            continue
        m = matcher(box_iou(gt, anc), 0.7, 0.3, True)
        matched_gt_boxes.append(gt[m.clamp(min=0)])
        lab = (m >= 0).to(torch.float32)
        lab[m == BELOW_LOW_THRESHOLD] = 0.0
        lab[m == BETWEEN_THRESHOLDS] = -1.0
        labels.append(lab)
    return labels, matched_gt_boxes


def rpn_compute_loss(objectness: Tensor, pred_bbox_deltas: Tensor, labels: List[Tensor], regression_targets: List[Tensor],
                     perm_fn=default_perm):
    """RegionProposalNetwork.compute_loss: 256 anchors per image, half positive (object_detector.py:89-90)."""
    pos_all, neg_all = [], []
    for i, lab in enumerate(labels):
        p, n = balanced_sample(lab, 256, 0.5, perm_fn, ("rpn", i))
        pos_mask = torch.zeros_like(lab, dtype=torch.bool)
        neg_mask = torch.zeros_like(lab, dtype=torch.bool)
        pos_mask[p] = True
        neg_mask[n] = True
        pos_all.append(pos_mask)
        neg_all.append(neg_mask)
    sampled_pos = torch.where(torch.cat(pos_all))[0]
    sampled_neg = torch.where(torch.cat(neg_all))[0]
    sampled = torch.cat([sampled_pos, sampled_neg])
    obj = objectness.flatten()
    lab = torch.cat(labels)
    reg = torch.cat(regression_targets)
    box_loss = F.smooth_l1_loss(pred_bbox_deltas[sampled_pos], reg[sampled_pos], beta=1 / 9, reduction="sum") / sampled.numel()
    obj_loss = F.binary_cross_entropy_with_logits(obj[sampled], lab[sampled])
    return obj_loss, box_loss


def rpn_targets_and_loss(objectness: Tensor, pred_bbox_deltas: Tensor, anchors: Tensor, targets, perm_fn=default_perm):
    """assign_targets_to_anchors + box_coder.encode (weights 1,1,1,1) + compute_loss, as custom_rpn.py:74-83 chains them.
    objectness [B*A,1], pred_bbox_deltas [B*A,4] (image-major), anchors [A,4] (the same grid for every image)."""
    labels, matched = assign_targets_to_anchors([anchors] * len(targets), targets)
    reg_targets = [box_encode(m, anchors, (1.0, 1.0, 1.0, 1.0)) for m in matched]
    return rpn_compute_loss(objectness, pred_bbox_deltas, labels, reg_targets, perm_fn)


def select_training_samples(proposals: List[Tensor], targets, perm_fn=default_perm):
    """RoIHeads.select_training_samples (roi_heads.py), configured as object_detector.py:118-123: add the gt boxes to
    the proposals, match at IoU 0.5 (no low-quality matches), sample 512 per image with a quarter positive, keep the
    sampled proposals IN INDEX ORDER, encode the matched gt boxes with weights (10, 10, 5, 5)."""
    out_props, out_labels, out_reg = [], [], []
    for i, (props, t) in enumerate(zip(proposals, targets)):
        gt = t["boxes"].to(props.dtype)
        props = torch.cat((props, gt))
        if gt.numel() == 0:
            clamped = torch.zeros((props.shape[0],), dtype=torch.int64)
            lab = torch.zeros((props.shape[0],), dtype=torch.int64)
        else:
            m = matcher(box_iou(gt, props), 0.5, 0.5, False)
            clamped = m.clamp(min=0)
            lab = t["labels"][clamped].to(torch.int64)
            lab[m == BELOW_LOW_THRESHOLD] = 0
            lab[m == BETWEEN_THRESHOLDS] = -1
        p, n = balanced_sample(lab, 512, 0.25, perm_fn, ("roi", i))
        mask = torch.zeros((props.shape[0],), dtype=torch.bool)
        mask[p] = True
        mask[n] = True
        inds = torch.where(mask)[0]
        gt_in = gt if gt.numel() else torch.zeros((1, 4), dtype=props.dtype)
        out_props.append(props[inds])
        out_labels.append(lab[inds])
        out_reg.append(box_encode(gt_in[clamped[inds]], props[inds], (10.0, 10.0, 5.0, 5.0)))
    return out_props, out_labels, out_reg


def fastrcnn_loss(class_logits: Tensor, box_regression: Tensor, labels: List[Tensor], regression_targets: List[Tensor]):
    """roi_heads.fastrcnn_loss."""
    lab = torch.cat(labels)
    reg = torch.cat(regression_targets)
    cls_loss = F.cross_entropy(class_logits, lab)
    pos = torch.where(lab > 0)[0]
    N = class_logits.shape[0]
    br = box_regression.reshape(N, box_regression.size(-1) // 4, 4)
    box_loss = F.smooth_l1_loss(br[pos, lab[pos]], reg[pos], beta=1 / 9, reduction="sum") / lab.numel()
    return cls_loss, box_loss
