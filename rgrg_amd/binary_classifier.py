"""Host-side mirror of ``src/binary_classifier`` (ttanida/rgrg).

``BinaryClassifierRegionSelection`` runs inside ``generate()``
(binary_classifier_region_selection.py:24-68) and, with its loss, inside the eval-mode
``forward()``; ``BinaryClassifierRegionAbnormal`` runs in ``forward()`` only
(report_generation_model.py:67-69,104-106).  Eval mode only: training needs backward
(SURVEY.md 8(f)).
"""
from __future__ import annotations

import torch
import torch.nn as nn

from ._owner import EngineOwner


class _BCEWithLogitsLossHolder(nn.Module):
    """Keeps the ``loss_fn.pos_weight`` buffer key of the reference state dict."""

    def __init__(self, pos_weight: float):
        super().__init__()
        self.register_buffer("pos_weight", torch.tensor([pos_weight]))


def _classifier() -> nn.Sequential:
    return nn.Sequential(nn.Linear(1024, 512), nn.ReLU(), nn.Linear(512, 128), nn.ReLU(), nn.Linear(128, 1))


class BinaryClassifierRegionSelection(EngineOwner):
    _engine_prefix = "binary_classifier_region_selection."

    def __init__(self):
        super().__init__()
        self.classifier = _classifier()
        self.loss_fn = _BCEWithLogitsLossHolder(2.2)

    def forward(self, top_region_features, class_detected, return_loss, region_has_sentence=None):
        """Eval mode of the reference (binary_classifier_region_selection.py:32-68):
        return_loss=False -> (selected_regions bool [B,29], selected_region_features [S,1024]);
        return_loss=True  -> (loss, selected_regions, selected_region_features), loss = BCEWithLogits(pos_weight
        2.2) over the detected regions against ``region_has_sentence``."""
        if self.training:
            raise NotImplementedError("rgrg_amd implements eval mode; the training step is SURVEY.md 8(f)")
        taps = {} if return_loss else None
        selected_regions, feats = self.engine().select(top_region_features, class_detected, taps)
        if not return_loss:
            return selected_regions, feats
        loss = self.engine().bce_masked(taps["selection_logits"].reshape(-1), class_detected, region_has_sentence,
                                        float(self.loss_fn.pos_weight.item()))
        return loss, selected_regions, feats


class BinaryClassifierRegionAbnormal(EngineOwner):
    _engine_prefix = "binary_classifier_region_abnormal."

    def __init__(self):
        super().__init__()
        self.classifier = _classifier()
        self.loss_fn = _BCEWithLogitsLossHolder(6.0)

    def forward(self, top_region_features, class_detected, region_is_abnormal):
        """Eval mode (binary_classifier_region_abnormal.py:32-60): -> (loss, predicted_abnormal_regions bool [B,29])."""
        if self.training:
            raise NotImplementedError("rgrg_amd implements eval mode; the training step is SURVEY.md 8(f)")
        return self.engine().abnormal(top_region_features, class_detected, region_is_abnormal,
                                      float(self.loss_fn.pos_weight.item()))
