"""CPU restatement of ``ReportGenerationModel.generate`` and the region
selection head.  TEST INFRASTRUCTURE ONLY (see ``oracle/__init__.py``).

Follows ``src/full_model/report_generation_model.py:212-276`` (generate) and ``:35-168`` (forward, eval
branch), ``src/binary_classifier/binary_classifier_region_selection.py:24-68`` and
``src/binary_classifier/binary_classifier_region_abnormal.py:32-60``.
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


def _classifier_logits(sd: SD, c: str, x: Tensor) -> Tensor:
    h = F.relu(F.linear(x, sd[c + "0.weight"], sd[c + "0.bias"]))
    h = F.relu(F.linear(h, sd[c + "2.weight"], sd[c + "2.bias"]))
    return F.linear(h, sd[c + "4.weight"], sd[c + "4.bias"]).squeeze(-1)


@torch.no_grad()
def forward_eval(sd: SD, images: Tensor, input_ids: Tensor, attention_mask: Tensor, region_has_sentence: Tensor,
                 region_is_abnormal: Tensor, image_targets=None, perm_fn=None):
    """ReportGenerationModel.forward in eval mode (report_generation_model.py:87-168): detector (losses {} when
    ``image_targets`` is None; with targets its four losses, computed on the sampled proposals - ``perm_fn`` supplies the
    sampler's draws, see oracle.tv013), selection classifier loss (BCEWithLogits pos_weight 2.2 on detected regions) and
    ``selected_regions``, abnormal classifier loss (pos_weight 6.0) and ``logits > -1`` predictions, decoder inputs of
    the selected regions (:196-210), teacher-forced LM loss.  Returns the reference's 8-tuple, or -1 (:136)."""
    from .language_model import lm_teacher_forced
    from . import tv013
    det_losses, detections, top_region_features, class_detected = object_detector_forward(
        sd, images, targets=image_targets, perm_fn=perm_fn or tv013.default_perm)
    selected_regions, selected_feats, sel_logits = region_selection(sd, top_region_features, class_detected)
    loss_sel = F.binary_cross_entropy_with_logits(sel_logits[class_detected], region_has_sentence[class_detected].float(),
                                                  pos_weight=torch.tensor([2.2]))
    abn_logits = _classifier_logits(sd, "binary_classifier_region_abnormal.classifier.", top_region_features)
    loss_abn = F.binary_cross_entropy_with_logits(abn_logits[class_detected], region_is_abnormal[class_detected].float(),
                                                  pos_weight=torch.tensor([6.0]))
    predicted_abnormal = abn_logits > -1
    flat = selected_regions.reshape(-1)
    ids, mask = input_ids[flat], attention_mask[flat]
    if ids.shape[0] == 0:
        return -1
    lm_loss = lm_teacher_forced(sd, ids, mask, selected_feats, return_loss=True)
    return det_losses, loss_sel, loss_abn, lm_loss, detections, class_detected, selected_regions, predicted_abnormal


def train_losses_and_grads(sd: SD, images: Tensor, input_ids: Tensor, attention_mask: Tensor, region_has_sentence: Tensor,
                           region_is_abnormal: Tensor):
    """Training branch of ReportGenerationModel.forward (report_generation_model.py:52-84,136-157) with the object
    detector frozen (its inference branch provides the features; BASELINE configs[4]) and dropout off: the two
    classifier losses over the detected regions, the LM loss over the regions that are detected AND have a sentence
    (:170-194), and the gradients torch autograd gives for the classifiers and the decoder's trainable tensors.
    Returns ((loss_sel, loss_abn, loss_lm), {key: grad})."""
    from .language_model import lm_loss_and_grads
    with torch.no_grad():
        _, _detections, top_region_features, class_detected = object_detector_forward(sd, images)
    grads, losses = {}, []
    for name, target, pw in (("binary_classifier_region_selection", region_has_sentence, 2.2),
                             ("binary_classifier_region_abnormal", region_is_abnormal, 6.0)):
        c = name + ".classifier."
        sd2 = dict(sd)
        keys = [c + f"{i}.{wb}" for i in (0, 2, 4) for wb in ("weight", "bias")]
        for k in keys:
            sd2[k] = sd[k].detach().clone().requires_grad_(True)
        with torch.enable_grad():
            logits = _classifier_logits(sd2, c, top_region_features)
            loss = F.binary_cross_entropy_with_logits(logits[class_detected], target[class_detected].float(),
                                                      pos_weight=torch.tensor([pw]))
            loss.backward()
        losses.append(loss.detach())
        grads.update({k: sd2[k].grad for k in keys})
    valid = torch.logical_and(class_detected, region_has_sentence)
    flat = valid.reshape(-1)
    if int(flat.sum()) == 0:
        return -1
    lm_loss, lm_grads = lm_loss_and_grads(sd, input_ids[flat], attention_mask[flat], top_region_features[valid])
    grads.update(lm_grads)
    return (losses[0], losses[1], lm_loss), grads


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
