"""Host-side mirror of ``src/binary_classifier`` (ttanida/rgrg).

``BinaryClassifierRegionSelection`` runs inside ``generate()``
(binary_classifier_region_selection.py:24-68) and, with its loss, inside the eval-mode
``forward()``; ``BinaryClassifierRegionAbnormal`` runs in ``forward()`` only
(report_generation_model.py:67-69,104-106).  In ``train()`` mode both return only their loss, with a ``grad_fn``
whose backward runs on the HIP kernels.
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


class _ClassifierLoss(torch.autograd.Function):
    """Autograd bridge of a region classifier's training loss: forward runs the HIP forward + backward
    (engine.classifier_loss_grad), backward hands the six parameter gradients to autograd."""

    @staticmethod
    def forward(ctx, owner, which, feats, mask, target, pos_weight, *params):
        eng = owner.engine()
        mlp = eng.sel if which == "selection" else eng.abn
        if mlp is None:
            raise RuntimeError(f"the loaded state dict has no binary_classifier_region_{which}.* weights")
        x = feats.detach().reshape(-1, feats.shape[-1]).contiguous().to(torch.float32)
        loss, _, grads = eng.classifier_loss_grad(mlp, x, mask, target, pos_weight)
        ctx.grads = grads
        return loss

    @staticmethod
    def backward(ctx, grad_out):
        return (None,) * 6 + tuple(g * grad_out for g in ctx.grads)


def _classifier_params(seq: nn.Sequential):
    return [seq[0].weight, seq[0].bias, seq[2].weight, seq[2].bias, seq[4].weight, seq[4].bias]


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
            # training: only the loss is returned (:46-47); it carries a grad_fn for the six classifier tensors
            return _ClassifierLoss.apply(self, "selection", top_region_features, class_detected, region_has_sentence,
                                         float(self.loss_fn.pos_weight.item()), *_classifier_params(self.classifier))
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
            return _ClassifierLoss.apply(self, "abnormal", top_region_features, class_detected, region_is_abnormal,
                                         float(self.loss_fn.pos_weight.item()), *_classifier_params(self.classifier))
        return self.engine().abnormal(top_region_features, class_detected, region_is_abnormal,
                                      float(self.loss_fn.pos_weight.item()))
