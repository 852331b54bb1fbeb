"""CPU restatement of the reference's object detector, inference branch.
TEST INFRASTRUCTURE ONLY (see ``oracle/__init__.py``).

Follows ``src/object_detector/object_detector.py:184-261`` (forward),
``src/object_detector/custom_rpn.py:53-85`` and
``src/object_detector/custom_roi_heads.py:63-269`` of ttanida/rgrg; the
torchvision arithmetic underneath is ``oracle/tv013.py``.
"""
from __future__ import annotations

from typing import Dict, List, Tuple

import torch
import torch.nn.functional as F

from . import tv013

Tensor = torch.Tensor
SD = Dict[str, Tensor]

NUM_CLASSES = 30  # 29 regions + background, object_detector.py:48-49


def rpn_forward(sd: SD, p: str, feat: Tensor, image_size: Tuple[int, int]) -> List[Tensor]:
    """CustomRegionProposalNetwork.forward with targets=None
    (custom_rpn.py:53-72): head -> anchors -> concat/permute -> decode with
    weights (1,1,1,1) -> filter_proposals(eval: 1000/1000, nms 0.7, score 0.0;
    object_detector.py:93-96)."""
    B = feat.shape[0]
    obj, reg = tv013.rpn_head(sd, p + "head.", feat)
    anchors = tv013.grid_anchors(image_size, tuple(feat.shape[-2:]))
    objectness = tv013.permute_and_flatten(obj, 1).reshape(B, -1)
    deltas = tv013.permute_and_flatten(reg, 4).reshape(-1, 4)
    proposals = tv013.box_decode(deltas, anchors.repeat(B, 1), (1.0, 1.0, 1.0, 1.0)).view(B, -1, 4)
    boxes, _scores = tv013.filter_proposals(proposals, objectness, image_size)
    return boxes


def top_region_postprocess(box_features: Tensor, box_regression: Tensor, class_logits: Tensor,
                           proposals: List[Tensor], image_sizes: List[Tuple[int, int]]):
    """CustomRoIHeads.get_top_region_features_detections_class_detected
    (custom_roi_heads.py:63-208), eval + return_feature_vectors=True.

    softmax over 30 incl. background -> drop background column -> per image:
    argmax class per box -> scores masked to the arg-max class -> per-class
    max over boxes (value, index; undetected class -> score 0 / index 0) ->
    class_detected = (#boxes predicting the class) > 0 -> gather features and
    the class's decoded+clipped box."""
    scores = F.softmax(class_logits, -1)[:, 1:]
    per_img = [p.shape[0] for p in proposals]
    boxes = tv013.box_decode(box_regression, torch.cat(proposals, 0), (10.0, 10.0, 5.0, 5.0))
    boxes = boxes.reshape(sum(per_img), -1, 4)  # [N,30,4] (custom_roi_heads.py:125; BoxCoder default weights)
    cls_det, feats, top_boxes, top_scores = [], [], [], []
    for sc, bx, ft, shape in zip(scores.split(per_img), boxes.split(per_img), box_features.split(per_img), image_sizes):
        pred = torch.argmax(sc, dim=1)
        mask = F.one_hot(pred, num_classes=NUM_CLASSES - 1)
        ts, ti = torch.max(sc * mask, dim=0)
        cls_det.append(mask.sum(0) > 0)
        feats.append(ft[ti])
        bx = tv013.clip_boxes_to_image(bx, shape)[:, 1:]
        top_boxes.append(bx[ti, torch.arange(NUM_CLASSES - 1)])
        top_scores.append(ts)
    return (torch.stack(cls_det), torch.stack(feats), torch.stack(top_boxes), torch.stack(top_scores))


def roi_heads_forward(sd: SD, p: str, feat: Tensor, proposals: List[Tensor], image_sizes: List[Tuple[int, int]],
                      return_intermediates: bool = False):
    """CustomRoIHeads.forward, targets=None (custom_roi_heads.py:210-269)."""
    rois = torch.cat([torch.cat([torch.full((b.shape[0], 1), float(i), dtype=b.dtype), b], 1)
                      for i, b in enumerate(proposals)], 0)
    scale = tv013.infer_scale(feat.shape[-1], image_sizes[0][-1])
    pooled = tv013.roi_align(feat, rois, scale, out_size=8, sampling_ratio=2)  # [N,2048,8,8]
    vec = tv013.two_mlp_head(sd, p + "box_head.", pooled)
    class_logits, box_regression = tv013.fastrcnn_predictor(sd, p + "box_predictor.", vec)
    box_features = torch.squeeze(F.avg_pool2d(pooled, 8))  # quirk: drops batch dim if N == 1 (:253-256)
    if box_features.dim() == 1:
        box_features = box_features[None]
    cls_det, feats, top_boxes, top_scores = top_region_postprocess(box_features, box_regression, class_logits,
                                                                   proposals, image_sizes)
    top_feats = F.linear(feats, sd[p + "dim_reduction.weight"], sd[p + "dim_reduction.bias"])  # :264
    out = {"class_detected": cls_det, "top_region_features": top_feats,
           "detections": {"top_region_boxes": top_boxes, "top_scores": top_scores}}
    if return_intermediates:
        out["_class_logits"], out["_box_regression"], out["_pooled_avg"] = class_logits, box_regression, box_features
    return out


def object_detector_forward(sd: SD, images: Tensor, p: str = "object_detector.", return_intermediates: bool = False,
                            targets=None, perm_fn=tv013.default_perm):
    """ObjectDetector.forward(images, targets) in eval mode with return_feature_vectors=True
    (object_detector.py:184-261) -> (losses, detections, top_region_features [B,29,1024], class_detected [B,29]).
    targets=None: inference, losses = {}.  With targets (the reference's validation loop, evaluate_model.py:413) the
    four detector losses are computed AND - exactly like the reference - the RoI heads run on the SAMPLED training
    proposals (custom_roi_heads.py:225-226), so detections / region features depend on the sampler's draws
    (``perm_fn``, see tv013.balanced_sample)."""
    feat = tv013.resnet50_trunk(sd, p + "backbone.", images)
    image_sizes = [tuple(images.shape[-2:])] * images.shape[0]  # image_list.py:15-20
    losses = {}
    if targets is None:
        proposals = rpn_forward(sd, p + "rpn.", feat, image_sizes[0])
        labels = reg_targets = None
    else:
        B = feat.shape[0]
        obj, reg = tv013.rpn_head(sd, p + "rpn.head.", feat)
        anchors = tv013.grid_anchors(image_sizes[0], tuple(feat.shape[-2:]))
        objectness = tv013.permute_and_flatten(obj, 1).reshape(-1, 1)
        deltas = tv013.permute_and_flatten(reg, 4).reshape(-1, 4)
        boxes = tv013.box_decode(deltas, anchors.repeat(B, 1), (1.0, 1.0, 1.0, 1.0)).view(B, -1, 4)
        proposals, _ = tv013.filter_proposals(boxes, objectness.reshape(B, -1), image_sizes[0])
        losses["loss_objectness"], losses["loss_rpn_box_reg"] = tv013.rpn_targets_and_loss(objectness, deltas, anchors, targets, perm_fn)
        proposals, labels, reg_targets = tv013.select_training_samples(proposals, targets, perm_fn)
    out = roi_heads_forward(sd, p + "roi_heads.", feat, proposals, image_sizes, return_intermediates or targets is not None)
    if targets is not None:
        cls_loss, box_loss = tv013.fastrcnn_loss(out["_class_logits"], out["_box_regression"], labels, reg_targets)
        # the reference's dict order: roi-head losses first, then the RPN's (object_detector.py:240-242)
        losses = {"loss_classifier": cls_loss, "loss_box_reg": box_loss, **losses}
    if return_intermediates:
        out["_features"], out["_proposals"], out["_losses"] = feat, proposals, losses
        return out
    return losses, out["detections"], out["top_region_features"], out["class_detected"]


def bbox_features(sd: SD, images: Tensor, bbox_coordinates: List[Tensor], p: str = "object_detector.") -> Tensor:
    """get_bbox_features (evaluate_bbox_variations/evaluate_bbox_variations.py:92-109): backbone ->
    box_roi_pool on the GIVEN boxes -> AvgPool2d(8) -> squeeze -> dim_reduction."""
    feat = tv013.resnet50_trunk(sd, p + "backbone.", images)
    rois = torch.cat([torch.cat([torch.full((b.shape[0], 1), float(i), dtype=b.dtype), b], 1)
                      for i, b in enumerate(bbox_coordinates)], 0)
    scale = tv013.infer_scale(feat.shape[-1], images.shape[-1])
    pooled = torch.squeeze(F.avg_pool2d(tv013.roi_align(feat, rois, scale, 8, 2), 8))
    r = p + "roi_heads.dim_reduction."
    return F.linear(pooled, sd[r + "weight"], sd[r + "bias"])
