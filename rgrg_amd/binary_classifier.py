"""Host-side mirror of ``src/binary_classifier`` (ttanida/rgrg).

``BinaryClassifierRegionSelection`` runs inside ``generate()``
(binary_classifier_region_selection.py:24-68); ``BinaryClassifierRegionAbnormal`` only
holds its parameters (it runs in forward()/training only: report_generation_model.py
:67-69,104-106 - SURVEY.md 8(f) "next").
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
        """Inference mode of the reference (eval, return_loss=False):
        -> (selected_regions bool [B,29], selected_region_features [S,1024])."""
        if return_loss or self.training:
            raise NotImplementedError("rgrg_amd implements the inference branch (eval mode, return_loss=False); "
                                      "the loss branches belong to the training step (SURVEY.md 8(f))")
        return self.engine().select(top_region_features, class_detected)


class BinaryClassifierRegionAbnormal(nn.Module):
    def __init__(self):
        super().__init__()
        self.classifier = _classifier()
        self.loss_fn = _BCEWithLogitsLossHolder(6.0)

    def forward(self, top_region_features, class_detected, region_is_abnormal):
        raise NotImplementedError("BinaryClassifierRegionAbnormal is not on the generate() path "
                                  "(report_generation_model.py:67-69); training/eval step is SURVEY.md 8(f)")
