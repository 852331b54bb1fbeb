"""Stand-in ``torchvision`` package used ONLY by ``make_golden.py`` in the build
container, so that the reference's own ``object_detector.py`` / ``custom_rpn.py``
/ ``custom_roi_heads.py`` / ``report_generation_model.py`` can be imported and
run unmodified on top of ``oracle/tv013.py`` (torchvision itself is not
installable here).  It only provides the class skeletons (attribute and
parameter names of torchvision 0.13.1) and forwards all arithmetic to the
oracle's functional restatement.  It never travels to the GPU box as part of a
test: the fixtures it helps produce do.
"""
from __future__ import annotations

import sys
import types
from collections import OrderedDict

import torch
import torch.nn as nn

from oracle import tv013


def _sd(module: nn.Module):
    return {k: v for k, v in module.state_dict().items()}


# ---- torchvision.models -----------------------------------------------------
class Bottleneck(nn.Module):
    def __init__(self, inplanes, planes, stride, downsample):
        super().__init__()
        self.conv1 = nn.Conv2d(inplanes, planes, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.conv2 = nn.Conv2d(planes, planes, 3, stride=stride, padding=1, bias=False)
        self.bn2 = nn.BatchNorm2d(planes)
        self.conv3 = nn.Conv2d(planes, planes * 4, 1, bias=False)
        self.bn3 = nn.BatchNorm2d(planes * 4)
        self.relu = nn.ReLU(inplace=True)
        if downsample:
            self.downsample = nn.Sequential(nn.Conv2d(inplanes, planes * 4, 1, stride=stride, bias=False),
                                            nn.BatchNorm2d(planes * 4))
        else:
            self.downsample = None
        self.stride = stride

    def forward(self, x):
        assert not self.training, "shim supports eval only"
        return tv013._bottleneck(_sd(self), "", x, self.stride)


class ResNet50(nn.Module):
    def __init__(self):
        super().__init__()
        self.conv1 = nn.Conv2d(3, 64, 7, stride=2, padding=3, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = nn.MaxPool2d(3, stride=2, padding=1)
        inpl = 64
        for i, (planes, blocks, stride) in enumerate(tv013.RESNET50_LAYERS):
            layers = []
            for b in range(blocks):
                layers.append(Bottleneck(inpl, planes, stride if b == 0 else 1, downsample=(b == 0)))
                inpl = planes * 4
            setattr(self, f"layer{i + 1}", nn.Sequential(*layers))
        self.avgpool = nn.AdaptiveAvgPool2d((1, 1))
        self.fc = nn.Linear(2048, 1000)


class ResNet50_Weights:
    DEFAULT = None


def resnet50(weights=None):
    return ResNet50()


# ---- torchvision.models.detection.rpn ---------------------------------------
class AnchorGenerator(nn.Module):
    def __init__(self, sizes, aspect_ratios):
        super().__init__()
        assert tuple(sizes[0]) == tv013.ANCHOR_SIZES and tuple(aspect_ratios[0]) == tv013.ANCHOR_RATIOS
        self.sizes, self.aspect_ratios = sizes, aspect_ratios

    def num_anchors_per_location(self):
        return [len(s) * len(a) for s, a in zip(self.sizes, self.aspect_ratios)]

    def forward(self, image_list, feature_maps):
        grid = tuple(feature_maps[0].shape[-2:])
        size = tuple(image_list.tensors.shape[-2:])
        a = tv013.grid_anchors(size, grid)
        return [a for _ in image_list.image_sizes]


class RPNHead(nn.Module):
    def __init__(self, in_channels, num_anchors):
        super().__init__()
        self.conv = nn.Sequential(nn.Sequential(nn.Conv2d(in_channels, in_channels, 3, padding=1), nn.ReLU(inplace=True)))
        self.cls_logits = nn.Conv2d(in_channels, num_anchors, 1)
        self.bbox_pred = nn.Conv2d(in_channels, num_anchors * 4, 1)
        for layer in self.modules():
            if isinstance(layer, nn.Conv2d):
                nn.init.normal_(layer.weight, std=0.01)
                nn.init.constant_(layer.bias, 0)

    def forward(self, x):
        sd = _sd(self)
        outs = [tv013.rpn_head(sd, "", f) for f in x]
        return [o[0] for o in outs], [o[1] for o in outs]


def concat_box_prediction_layers(box_cls, box_regression):
    cls = [tv013.permute_and_flatten(c, 1) for c in box_cls]
    reg = [tv013.permute_and_flatten(r, 4) for r in box_regression]
    return torch.cat(cls, dim=1).flatten(0, -2), torch.cat(reg, dim=1).reshape(-1, 4)


PERM_FN = tv013.default_perm  # the samplers' draws (torch.randperm in torchvision); fixture scripts inject a seeded one


class _BoxCoder:
    def __init__(self, weights):
        self.weights = weights

    def encode(self, reference_boxes, proposals):
        return tuple(tv013.box_encode(r, p, self.weights) for r, p in zip(reference_boxes, proposals))

    def decode(self, rel_codes, boxes):
        concat = torch.cat(list(boxes), dim=0)
        n = concat.shape[0]
        pred = tv013.box_decode(rel_codes.reshape(n, -1), concat, self.weights)
        return pred.reshape(n, -1, 4)


class RegionProposalNetwork(nn.Module):
    def __init__(self, anchor_generator, head, fg_iou_thresh, bg_iou_thresh, batch_size_per_image, positive_fraction,
                 pre_nms_top_n, post_nms_top_n, nms_thresh, score_thresh=0.0):
        super().__init__()
        self.anchor_generator, self.head = anchor_generator, head
        self.box_coder = _BoxCoder((1.0, 1.0, 1.0, 1.0))
        self._pre, self._post = pre_nms_top_n, post_nms_top_n
        self.nms_thresh, self.score_thresh, self.min_size = nms_thresh, score_thresh, 1e-3

    def assign_targets_to_anchors(self, anchors, targets):
        return tv013.assign_targets_to_anchors(anchors, targets)

    def compute_loss(self, objectness, pred_bbox_deltas, labels, regression_targets):
        return tv013.rpn_compute_loss(objectness, pred_bbox_deltas, labels, list(regression_targets), PERM_FN)

    def filter_proposals(self, proposals, objectness, image_shapes, num_anchors_per_level):
        assert not self.training
        B = proposals.shape[0]
        return tv013.filter_proposals(proposals, objectness.detach().reshape(B, -1), tuple(image_shapes[0]),
                                      self._pre["testing"], self._post["testing"], self.nms_thresh, self.score_thresh,
                                      self.min_size)


# ---- torchvision.models.detection.faster_rcnn / roi_heads ----------------------
class TwoMLPHead(nn.Module):
    def __init__(self, in_channels, representation_size):
        super().__init__()
        self.fc6 = nn.Linear(in_channels, representation_size)
        self.fc7 = nn.Linear(representation_size, representation_size)

    def forward(self, x):
        return tv013.two_mlp_head(_sd(self), "", x)


class FastRCNNPredictor(nn.Module):
    def __init__(self, in_channels, num_classes):
        super().__init__()
        self.cls_score = nn.Linear(in_channels, num_classes)
        self.bbox_pred = nn.Linear(in_channels, num_classes * 4)

    def forward(self, x):
        return tv013.fastrcnn_predictor(_sd(self), "", x)


class MultiScaleRoIAlign(nn.Module):
    def __init__(self, featmap_names, output_size, sampling_ratio):
        super().__init__()
        self.featmap_names = featmap_names
        self.output_size = (output_size, output_size) if isinstance(output_size, int) else tuple(output_size)
        self.sampling_ratio = sampling_ratio

    def forward(self, x, boxes, image_shapes):
        feat = x[self.featmap_names[0]]
        rois = torch.cat([torch.cat([torch.full((b.shape[0], 1), float(i), dtype=b.dtype), b], 1)
                          for i, b in enumerate(boxes)], 0)
        scale = tv013.infer_scale(feat.shape[-1], image_shapes[0][-1])
        return tv013.roi_align(feat, rois, scale, self.output_size[0], self.sampling_ratio)


class RoIHeads(nn.Module):
    def __init__(self, box_roi_pool, box_head, box_predictor, fg_iou_thresh, bg_iou_thresh, batch_size_per_image,
                 positive_fraction, bbox_reg_weights, score_thresh, nms_thresh, detections_per_img,
                 mask_roi_pool=None, mask_head=None, mask_predictor=None, keypoint_roi_pool=None,
                 keypoint_head=None, keypoint_predictor=None):
        super().__init__()
        if bbox_reg_weights is None:
            bbox_reg_weights = (10.0, 10.0, 5.0, 5.0)
        self.box_coder = _BoxCoder(bbox_reg_weights)
        self.box_roi_pool, self.box_head, self.box_predictor = box_roi_pool, box_head, box_predictor

    def select_training_samples(self, proposals, targets):
        props, labels, reg = tv013.select_training_samples(proposals, targets, PERM_FN)
        return props, None, labels, reg


def fastrcnn_loss(class_logits, box_regression, labels, regression_targets):
    return tv013.fastrcnn_loss(class_logits, box_regression, labels, regression_targets)


def install():
    """Register the stand-in package tree in sys.modules."""
    def mod(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m

    boxes = mod("torchvision.ops.boxes", clip_boxes_to_image=tv013.clip_boxes_to_image)
    ops = mod("torchvision.ops", MultiScaleRoIAlign=MultiScaleRoIAlign, boxes=boxes)
    rpn = mod("torchvision.models.detection.rpn", AnchorGenerator=AnchorGenerator, RPNHead=RPNHead,
              RegionProposalNetwork=RegionProposalNetwork, concat_box_prediction_layers=concat_box_prediction_layers)
    frcnn = mod("torchvision.models.detection.faster_rcnn", TwoMLPHead=TwoMLPHead, FastRCNNPredictor=FastRCNNPredictor)
    rh = mod("torchvision.models.detection.roi_heads", RoIHeads=RoIHeads, fastrcnn_loss=fastrcnn_loss)
    det = mod("torchvision.models.detection", rpn=rpn, faster_rcnn=frcnn, roi_heads=rh)
    models = mod("torchvision.models", resnet50=resnet50, ResNet50_Weights=ResNet50_Weights, detection=det)
    mod("torchvision", ops=ops, models=models, __version__="0.13.1-shim")
