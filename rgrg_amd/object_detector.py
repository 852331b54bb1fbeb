"""Host-side mirror of the reference's ``src/object_detector`` package.

Same class names, attribute names and state-dict keys as ttanida/rgrg
(``object_detector.py:18-131``: ``backbone`` = Sequential-indexed ResNet-50 children,
``rpn.head.{conv.0.0,cls_logits,bbox_pred}``, ``roi_heads.{box_head.fc6,fc7,
box_predictor.cls_score,bbox_pred,dim_reduction}``) so a reference checkpoint loads
unchanged.  The modules only HOLD parameters; all arithmetic runs in the HIP library
through ``rgrg_amd.engine.HipEngine`` (no torch/torchvision compute, no CPU fallback).
"""
from __future__ import annotations

from collections import OrderedDict
from typing import Dict, List, Optional

import torch
import torch.nn as nn
from torch import Tensor

from .constants import RESNET50_LAYERS
from . import _hip
from ._owner import EngineOwner


class _Holder(nn.Module):
    """Parameter container: calling it is a bug (compute lives in librgrg_hip.so)."""

    def forward(self, *a, **k):  # pragma: no cover
        raise RuntimeError(f"{type(self).__name__} only holds parameters; the computation runs in the HIP engine")


class Bottleneck(_Holder):
    def __init__(self, inplanes: int, planes: int, downsample: bool):
        super().__init__()
        self.conv1 = nn.Conv2d(inplanes, planes, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.conv2 = nn.Conv2d(planes, planes, 3, bias=False)
        self.bn2 = nn.BatchNorm2d(planes)
        self.conv3 = nn.Conv2d(planes, planes * 4, 1, bias=False)
        self.bn3 = nn.BatchNorm2d(planes * 4)
        if downsample:
            self.downsample = nn.Sequential(nn.Conv2d(inplanes, planes * 4, 1, bias=False), nn.BatchNorm2d(planes * 4))


class _Backbone(nn.Sequential):
    """``self.backbone`` of the reference is callable (evaluate_bbox_variations.py:93): NCHW in, NCHW
    [B,2048,H/32,W/32] out.  The trunk itself runs NHWC in the HIP engine; the permute is a view."""

    def forward(self, images: Tensor) -> Tensor:
        eng = self.__dict__["_owner"].engine()
        return eng.backbone(images.to(torch.float32)).permute(0, 3, 1, 2)


class MultiScaleRoIAlign(nn.Module):
    """Single-level ``box_roi_pool`` (object_detector.py:105-106): features {"0": NCHW}, boxes list of
    [n_i,4] -> [N,2048,8,8]."""

    def __init__(self):
        super().__init__()
        self.output_size = (8, 8)
        self.sampling_ratio = 2

    def forward(self, features, boxes, image_shapes):
        heads = self.__dict__["_heads"]
        eng = heads.__dict__["_owner"].engine()
        feat = features["0"] if isinstance(features, dict) else features
        maps, pooled = eng.roi_align_boxes(feat.permute(0, 2, 3, 1).contiguous(), list(boxes))
        out = maps.view(maps.shape[0], 8, 8, maps.shape[2]).permute(0, 3, 1, 2)
        heads.__dict__["_last_pool"] = (out, pooled)  # avg_pool of exactly this tensor is already computed (fused kernel)
        return out


class _AvgPool8(nn.Module):
    """``self.avg_pool = nn.AvgPool2d(8)`` (custom_roi_heads.py:60): [N,C,8,8] -> [N,C,1,1]."""

    def forward(self, x: Tensor) -> Tensor:
        heads = self.__dict__["_heads"]
        last = heads.__dict__.get("_last_pool")
        if last is not None and last[0] is x:
            return last[1].view(x.shape[0], x.shape[1], 1, 1)
        eng = heads.__dict__["_owner"].engine()
        N, Cc = x.shape[:2]
        w = torch.full((1, 64), 1.0 / 64.0, dtype=torch.float32, device=x.device)
        return eng.linear(x.reshape(N * Cc, 64).contiguous().to(torch.float32), w, None).view(N, Cc, 1, 1)


class _EngineLinear(nn.Linear):
    """nn.Linear whose forward runs on the f32-MFMA GEMM of the HIP library."""

    def forward(self, x: Tensor) -> Tensor:
        eng = self.__dict__["_owner"].engine()
        lead = x.shape[:-1]
        y = eng.linear(x.reshape(-1, x.shape[-1]).contiguous().to(torch.float32), self.weight.detach().contiguous(),
                       self.bias.detach().contiguous())
        return y.view(*lead, -1)


def _resnet50_trunk() -> nn.Sequential:
    """children()[:-2] of a ResNet-50 with a 1-channel stem (object_detector.py:51-58)."""
    mods: List[nn.Module] = [nn.Conv2d(1, 64, 7, bias=False), nn.BatchNorm2d(64), nn.ReLU(), nn.MaxPool2d(3, 2, 1)]
    inpl = 64
    for planes, blocks, _stride in RESNET50_LAYERS:
        layer = []
        for b in range(blocks):
            layer.append(Bottleneck(inpl, planes, downsample=(b == 0)))
            inpl = planes * 4
        mods.append(nn.Sequential(*layer))
    seq = _Backbone(*mods)
    seq.out_channels = 2048
    return seq


class RPNHead(_Holder):
    def __init__(self, in_channels: int, num_anchors: int):
        super().__init__()
        self.conv = nn.Sequential(nn.Sequential(nn.Conv2d(in_channels, in_channels, 3), nn.ReLU()))  # keys conv.0.0.*
        self.cls_logits = nn.Conv2d(in_channels, num_anchors, 1)
        self.bbox_pred = nn.Conv2d(in_channels, num_anchors * 4, 1)


class CustomRegionProposalNetwork(_Holder):
    def __init__(self, in_channels: int = 2048, num_anchors: int = 160):
        super().__init__()
        self.head = RPNHead(in_channels, num_anchors)


class TwoMLPHead(_Holder):
    def __init__(self, in_channels: int, representation_size: int):
        super().__init__()
        self.fc6 = nn.Linear(in_channels, representation_size)
        self.fc7 = nn.Linear(representation_size, representation_size)


class FastRCNNPredictor(_Holder):
    def __init__(self, in_channels: int, num_classes: int):
        super().__init__()
        self.cls_score = nn.Linear(in_channels, num_classes)
        self.bbox_pred = nn.Linear(in_channels, num_classes * 4)


class CustomRoIHeads(_Holder):
    def __init__(self, return_feature_vectors: bool):
        super().__init__()
        self.return_feature_vectors = return_feature_vectors
        self.box_roi_pool = MultiScaleRoIAlign()
        self.box_head = TwoMLPHead(2048 * 8 * 8, 1024)
        self.box_predictor = FastRCNNPredictor(1024, 30)
        self.avg_pool = _AvgPool8()
        self.dim_reduction = _EngineLinear(2048, 1024)


class ImageList:
    """src/object_detector/image_list.py:5-24."""

    def __init__(self, images_tensor: Tensor) -> None:
        self.tensors = images_tensor
        self.image_sizes = [tuple(images_tensor.shape[-2:]) for _ in range(images_tensor.shape[0])]


class ObjectDetector(EngineOwner):
    """Faster R-CNN with a ResNet-50 C5 trunk (object_detector.py:18), eval mode: inference (``targets=None``) and
    the validation forward with targets (detector losses)."""

    _engine_prefix = "object_detector."

    def __init__(self, return_feature_vectors: bool = False):
        super().__init__()
        self.return_feature_vectors = return_feature_vectors
        self.num_classes = 30
        self.backbone = _resnet50_trunk()
        self.rpn = CustomRegionProposalNetwork(2048, 160)
        self.roi_heads = CustomRoIHeads(return_feature_vectors)
        # callable pieces the reference's selection-based generation reaches into (plain attributes: no module cycle)
        self.backbone.__dict__["_owner"] = self
        self.roi_heads.__dict__["_owner"] = self
        self.roi_heads.box_roi_pool.__dict__["_heads"] = self.roi_heads
        self.roi_heads.avg_pool.__dict__["_heads"] = self.roi_heads
        self.roi_heads.dim_reduction.__dict__["_owner"] = self

    def _transform_inputs_for_rpn_and_roi(self, images, features):
        return ImageList(images), OrderedDict([("0", features)])

    def _check_targets(self, targets) -> None:
        """object_detector.py:133-162 (torch._assert -> AssertionError, same messages) plus what torchvision's
        RoIHeads.check_targets and the losses enforce later in the reference's call: float boxes / int64 labels (TypeError)
        and class labels below the 30 logits (torch's cross entropy: "Target N is out of bounds").  A NEGATIVE label raises
        nothing, in the reference as here: proposals matched to that box count as neither positive nor negative for the
        balanced sampler (torchvision samples labels >= 1 and == 0 only) and drop out of the losses."""
        for target_idx, t in enumerate(targets):   # shape / dtype checks: host metadata only
            boxes = t["boxes"]
            if not isinstance(boxes, torch.Tensor):
                raise AssertionError(f"Expected target boxes to be of type Tensor, got {type(boxes)}.")
            if boxes.dim() != 2 or boxes.shape[-1] != 4:
                raise AssertionError(f"Expected target boxes to be a tensor of shape [N, 4], got {boxes.shape}.")
            if boxes.dtype not in (torch.float, torch.double, torch.half):
                raise TypeError("target boxes must of float type")
            labels = t["labels"]
            if not isinstance(labels, torch.Tensor) or labels.dtype != torch.int64:
                raise TypeError("target labels must of int64 type")
            if labels.numel() != boxes.shape[0]:
                raise AssertionError(f"{boxes.shape[0]} boxes but {labels.numel()} labels for target at index {target_idx}")
        # value checks (degenerate boxes, label range): ONE device reduction and one 3-word read-back for the whole batch
        # (the reference synchronises once per image here, object_detector.py:146); details are fetched on the error path only
        sizes = [int(t["boxes"].shape[0]) for t in targets]
        if not sum(sizes):
            return
        boxes = torch.cat([t["boxes"].reshape(-1, 4) for t in targets]).to(torch.float32)
        labels = torch.cat([t["labels"].reshape(-1).to(boxes.device) for t in targets])   # boxes / labels may sit on different devices
        degenerate = (boxes[:, 2:] <= boxes[:, :2]).any(dim=1)
        flags = torch.stack([degenerate.any().to(torch.int64), labels.max()]).tolist()
        if flags[0]:
            bb = int(torch.where(degenerate)[0][0])
            target_idx = next(i for i in range(len(sizes)) if bb < sum(sizes[:i + 1]))
            raise AssertionError("All bounding boxes should have positive height and width."
                                 f" Found invalid box {boxes[bb].tolist()} for target at index {target_idx}.")
        if flags[1] >= 30:
            raise IndexError(f"Target {flags[1]} is out of bounds.")

    def forward(self, images: Tensor, targets: Optional[List[Dict[str, Tensor]]] = None):
        """Eval-mode ``ObjectDetector.forward`` (object_detector.py:184-261).  ``targets=None``: inference, losses = {}.
        With targets (the reference's validation loop) the four detector losses are returned and - exactly as in the
        reference - detections / region features come from the SAMPLED training proposals.  ``self.sampler_keys`` (a
        callable ``(stage, B, n) -> fp32 keys [B, n]``; the candidates with the smallest keys are sampled) replaces the
        device's torch.rand draws in the two samplers when set (tests)."""
        if self.training:
            raise NotImplementedError("rgrg_amd runs the detector in eval mode (BatchNorm running statistics, test-time "
                                      "proposal counts); training the detector is not implemented")
        low = _hip.autocast_mode() if images.is_cuda else 0
        losses: Dict[str, Tensor] = {}
        if targets is not None:
            self._check_targets(targets)
            losses, detections, top_region_features, class_detected = self.engine().detect(
                images, bf16=low, targets=targets, keys_fn=getattr(self, "sampler_keys", None))
        else:
            detections, top_region_features, class_detected = self.engine().detect(images, bf16=low)
        if not self.return_feature_vectors:
            return losses, detections, class_detected
        return losses, detections, top_region_features, class_detected
