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
    seq = nn.Sequential(*mods)
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
        self.box_head = TwoMLPHead(2048 * 8 * 8, 1024)
        self.box_predictor = FastRCNNPredictor(1024, 30)
        self.dim_reduction = nn.Linear(2048, 1024)


class ImageList:
    """src/object_detector/image_list.py:5-24."""

    def __init__(self, images_tensor: Tensor) -> None:
        self.tensors = images_tensor
        self.image_sizes = [tuple(images_tensor.shape[-2:]) for _ in range(images_tensor.shape[0])]


class ObjectDetector(EngineOwner):
    """Faster R-CNN with a ResNet-50 C5 trunk (object_detector.py:18).  Only the
    inference branch (``targets=None``, eval) is implemented on the HIP path."""

    _engine_prefix = "object_detector."

    def __init__(self, return_feature_vectors: bool = False):
        super().__init__()
        self.return_feature_vectors = return_feature_vectors
        self.num_classes = 30
        self.backbone = _resnet50_trunk()
        self.rpn = CustomRegionProposalNetwork(2048, 160)
        self.roi_heads = CustomRoIHeads(return_feature_vectors)

    def _transform_inputs_for_rpn_and_roi(self, images, features):
        return ImageList(images), OrderedDict([("0", features)])

    def forward(self, images: Tensor, targets: Optional[List[Dict[str, Tensor]]] = None):
        if targets is not None or self.training:
            raise NotImplementedError("rgrg_amd implements the inference branch of ObjectDetector.forward "
                                      "(targets=None, eval mode); training is a later row of SURVEY.md 8(f)")
        detections, top_region_features, class_detected = self.engine().detect(images)
        losses: Dict[str, Tensor] = {}
        if not self.return_feature_vectors:
            return losses, detections, class_detected
        return losses, detections, top_region_features, class_detected
