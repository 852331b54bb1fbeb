"""CPU restatement of ``ReportGenerationModel.generate`` and the region
selection head.  TEST INFRASTRUCTURE ONLY (see ``oracle/__init__.py``).

Follows ``src/full_model/report_generation_model.py:212-276`` and
``src/binary_classifier/binary_classifier_region_selection.py:24-68``.
"""
from __future__ import annotations

from typing import Dict, Optional

import torch
import torch.nn.functional as F

from .detector import object_detector_forward
from .language_model import greedy_generate

Tensor = torch.Tensor
SD = Dict[str, Tensor]


def region_selection(sd: SD, top_region_features: Tensor, class_detected: Tensor,
                     p: str = "binary_classifier_region_selection."):
    """BinaryClassifierRegionSelection.forward(return_loss=False), eval
    (binary_classifier_region_selection.py:32-68): 1024->512->128->1 ReLU MLP,
    ``logits > -1`` (strict), ``&= class_detected``, row-major boolean gather."""
    c = p + "classifier."
    h = F.relu(F.linear(top_region_features, sd[c + "0.weight"], sd[c + "0.bias"]))
    h = F.relu(F.linear(h, sd[c + "2.weight"], sd[c + "2.bias"]))
    logits = F.linear(h, sd[c + "4.weight"], sd[c + "4.bias"]).squeeze(-1)
    selected = logits > -1
    selected = selected & class_detected
    return selected, top_region_features[selected], logits


@torch.no_grad()
def generate(sd: SD, images: Tensor, max_length: Optional[int] = None, return_intermediates: bool = False,
             num_beams: int = 1, early_stopping: bool = False):
    """ReportGenerationModel.generate(images, max_length, num_beams=1): returns
    ``(output_ids, selected_regions, detections, class_detected)`` or the int
    ``-1`` when no region is both detected and selected (:260-261)."""
    _, detections, top_region_features, class_detected = object_detector_forward(sd, images)
    selected_regions, selected_feats, sel_logits = region_selection(sd, top_region_features, class_detected)
    if selected_feats.shape[0] == 0:
        return -1
    if num_beams > 1:
        from .language_model import beam_generate
        ids = beam_generate(sd, selected_feats, max_length, num_beams, early_stopping)
    else:
        ids = greedy_generate(sd, selected_feats, max_length)
    if return_intermediates:
        return ids, selected_regions, detections, class_detected, {"top_region_features": top_region_features,
                                                                   "selection_logits": sel_logits}
    return ids, selected_regions, detections, class_detected
