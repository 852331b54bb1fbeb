"""Host-side mirror of ``src/full_model/report_generation_model.py`` (ttanida/rgrg).

Drop-in for the inference hot path: same constructor, sub-module attribute names,
``load_state_dict(checkpoint["model"])`` keys and ``generate()`` signature / return
tuple / ``-1`` sentinel (report_generation_model.py:12-33, :212-276).
"""
from __future__ import annotations

from typing import Dict, Optional

import torch
from torch import Tensor

from . import _hip
from ._owner import EngineOwner
from .binary_classifier import BinaryClassifierRegionAbnormal, BinaryClassifierRegionSelection
from .language_model import LanguageModel
from .object_detector import ObjectDetector

_LM = "language_model."
_G = _LM + "gpt_with_lm_head.transformer."
_BLOCK_SUB = {"ln_1": "0", "attn": "1", "ln_2": "2", "mlp": "3"}


def expand_alias_keys(sd: Dict[str, Tensor]) -> Dict[str, Tensor]:
    """The reference state dict holds every GPT-2 tensor under three aliased key families
    (SURVEY.md 8(b)).  Accept a dict that carries only one family (canonical
    ``gpt_with_lm_head.transformer.*``, ``gpt.*`` or ``gpt2_blocks.*`` + ``wte/wpe/...``)
    by filling in the others; also accept the pre-0.13 ``rpn.head.conv.{weight,bias}``
    names (generate_reports_for_images.py:156-159)."""
    out = dict(sd)
    for old, new in (("object_detector.rpn.head.conv.weight", "object_detector.rpn.head.conv.0.0.weight"),
                     ("object_detector.rpn.head.conv.bias", "object_detector.rpn.head.conv.0.0.bias")):
        if old in out and new not in out:
            out[new] = out.pop(old)
    # gpt.* / gpt2_blocks.* / top-level -> canonical
    for k in list(out):
        if k.startswith(_LM + "gpt."):
            out.setdefault(_G + k[len(_LM + "gpt."):], out[k])
        elif k.startswith(_LM + "gpt2_blocks."):
            n, sub, *rest = k[len(_LM + "gpt2_blocks."):].split(".")
            name = {v: kk for kk, v in _BLOCK_SUB.items()}[sub]
            out.setdefault(f"{_G}h.{n}.{name}." + ".".join(rest), out[k])
    for top, canon in ((_LM + "wte.weight", _G + "wte.weight"), (_LM + "wpe.weight", _G + "wpe.weight"),
                       (_LM + "final_layernorm.weight", _G + "ln_f.weight"), (_LM + "final_layernorm.bias", _G + "ln_f.bias"),
                       (_LM + "lm_head.weight", _LM + "gpt_with_lm_head.lm_head.weight")):
        if top in out:
            out.setdefault(canon, out[top])
    if _G + "wte.weight" in out:
        out.setdefault(_LM + "gpt_with_lm_head.lm_head.weight", out[_G + "wte.weight"])
    # canonical -> aliases
    for k in [k for k in out if k.startswith(_G)]:
        rest = k[len(_G):]
        out.setdefault(_LM + "gpt." + rest, out[k])
        parts = rest.split(".")
        if parts[0] == "h":
            out.setdefault(f"{_LM}gpt2_blocks.{parts[1]}.{_BLOCK_SUB[parts[2]]}." + ".".join(parts[3:]), out[k])
        elif parts[0] in ("wte", "wpe"):
            out.setdefault(f"{_LM}{parts[0]}.weight", out[k])
        elif parts[0] == "ln_f":
            out.setdefault(_LM + "final_layernorm." + parts[1], out[k])
    if _LM + "gpt_with_lm_head.lm_head.weight" in out:
        out.setdefault(_LM + "lm_head.weight", out[_LM + "gpt_with_lm_head.lm_head.weight"])
    return out


class ReportGenerationModel(EngineOwner):
    """Object detector encoder -> region-selection classifier -> language-model decoder."""

    def __init__(self, pretrain_without_lm_model: bool = False):
        super().__init__()
        self.pretrain_without_lm_model = pretrain_without_lm_model
        self.object_detector = ObjectDetector(return_feature_vectors=True)
        self.binary_classifier_region_selection = BinaryClassifierRegionSelection()
        self.binary_classifier_region_abnormal = BinaryClassifierRegionAbnormal()
        self.language_model = LanguageModel()
        for child in (self.object_detector, self.binary_classifier_region_selection, self.binary_classifier_region_abnormal,
                      self.language_model):
            self._adopt(child)

    def load_state_dict(self, state_dict, strict: bool = True, **kw):
        """Loads a reference ``checkpoint["model"]`` unchanged.  Missing alias families and the
        persistent buffers the kernels do not need (``causal_mask``, ``mask_out_value``,
        ``num_batches_tracked``, ``pos_weight``) are filled from the module; anything else that
        is missing or unexpected is reported (raised when ``strict``), never silently dropped."""
        sd = expand_alias_keys(state_dict)
        own = self.state_dict()
        optional = ("causal_mask", "mask_out_value", "num_batches_tracked", "loss_fn.pos_weight")
        for k, v in own.items():
            if k not in sd and k.endswith(optional):
                sd[k] = v
        return super().load_state_dict(sd, strict=strict, **kw)

    def forward(self, images: torch.FloatTensor, image_targets, input_ids: torch.LongTensor,
                attention_mask: torch.FloatTensor, region_has_sentence: torch.BoolTensor,
                region_is_abnormal: torch.BoolTensor, return_loss: bool = True, past_key_values=None,
                position_ids: Optional[torch.LongTensor] = None, use_cache: Optional[bool] = False):
        """Eval-mode branch of report_generation_model.py:35-168:
        detector -> both region classifiers with their losses -> decoder inputs of the SELECTED regions
        (get_valid_decoder_input_for_evaluation, :196-210) -> teacher-forced language-model loss.  Returns
        ``(obj_detector_loss_dict, classifier_loss_region_selection, classifier_loss_region_abnormal,
        language_model_loss, detections, class_detected, selected_regions, predicted_abnormal_regions)``, the
        7-tuple without the LM loss when ``pretrain_without_lm_model``, or ``-1`` when no region is selected (:136).

        ``image_targets`` (list of {"boxes", "labels"} per image, as evaluate_model.py:413 passes them): the detector
        returns its four validation losses and - as in the reference - runs its RoI heads on the randomly SAMPLED
        training proposals (custom_roi_heads.py:225-226), so the detections / region features of this call depend on the
        sampler's draws.  ``image_targets=None``: ``obj_detector_loss_dict`` is ``{}`` (object_detector.py:195-197).
        In ``train()`` mode see ``_forward_train`` (frozen detector)."""
        if self.training:
            return self._forward_train(images, input_ids, attention_mask, region_has_sentence, region_is_abnormal, return_loss,
                                       past_key_values, position_ids, use_cache)
        obj_detector_loss_dict, detections, top_region_features, class_detected = self.object_detector(images, image_targets)
        del images
        classifier_loss_region_selection, selected_regions, selected_region_features = self.binary_classifier_region_selection(
            top_region_features, class_detected, return_loss=True, region_has_sentence=region_has_sentence)
        classifier_loss_region_abnormal, predicted_abnormal_regions = self.binary_classifier_region_abnormal(
            top_region_features, class_detected, region_is_abnormal)
        if self.pretrain_without_lm_model:
            return (obj_detector_loss_dict, classifier_loss_region_selection, classifier_loss_region_abnormal, detections,
                    class_detected, selected_regions, predicted_abnormal_regions)
        valid_input_ids, valid_attention_mask = self.get_valid_decoder_input_for_evaluation(selected_regions, input_ids,
                                                                                            attention_mask)
        if valid_input_ids.shape[0] == 0:
            return -1
        language_model_loss = self.language_model(valid_input_ids, valid_attention_mask, selected_region_features, return_loss,
                                                  past_key_values, position_ids, use_cache)
        return (obj_detector_loss_dict, classifier_loss_region_selection, classifier_loss_region_abnormal, language_model_loss,
                detections, class_detected, selected_regions, predicted_abnormal_regions)

    def _forward_train(self, images, input_ids, attention_mask, region_has_sentence, region_is_abnormal, return_loss,
                       past_key_values, position_ids, use_cache):
        """Training branch (report_generation_model.py:52-84,136-157) with the object detector FROZEN (BASELINE
        configs[4]; the reference also fine-tunes it): the detector runs its inference branch, ``image_targets`` are
        not used and ``obj_detector_loss_dict`` is ``{}``.  Returns ``(obj_detector_loss_dict,
        classifier_loss_region_selection, classifier_loss_region_abnormal[, language_model_loss])`` - losses with a
        ``grad_fn`` for the classifiers and for uk/uv/feature_space_transformation_nn - or ``-1`` (:136)."""
        low = _hip.autocast_mode()
        with torch.no_grad():
            _detections, top_region_features, class_detected = self.engine().detect(images, bf16=low)
        del images
        classifier_loss_region_selection = self.binary_classifier_region_selection(
            top_region_features, class_detected, return_loss=True, region_has_sentence=region_has_sentence)
        classifier_loss_region_abnormal = self.binary_classifier_region_abnormal(top_region_features, class_detected,
                                                                                 region_is_abnormal)
        if self.pretrain_without_lm_model:
            return {}, classifier_loss_region_selection, classifier_loss_region_abnormal
        valid_input_ids, valid_attention_mask, valid_region_features = self.get_valid_decoder_input_for_training(
            class_detected, region_has_sentence, input_ids, attention_mask, top_region_features)
        if valid_input_ids.shape[0] == 0:
            return -1
        language_model_loss = self.language_model(valid_input_ids, valid_attention_mask, valid_region_features, return_loss,
                                                  past_key_values, position_ids, use_cache)
        return {}, classifier_loss_region_selection, classifier_loss_region_abnormal, language_model_loss

    def get_valid_decoder_input_for_training(self, class_detected, region_has_sentence, input_ids, attention_mask, region_features):
        """report_generation_model.py:170-194: regions that were detected AND have a ground-truth sentence."""
        valid = torch.logical_and(class_detected, region_has_sentence)
        flat = valid.reshape(-1)
        return input_ids[flat], attention_mask[flat], region_features[valid]

    def trainable_parameters(self):
        """What a frozen-detector training run optimises: both region classifiers and the decoder's uk/uv/fst-nn
        (53.66 M values)."""
        return (list(self.binary_classifier_region_selection.classifier.parameters()) +
                list(self.binary_classifier_region_abnormal.classifier.parameters()) +
                self.language_model.trainable_parameters())

    def get_valid_decoder_input_for_evaluation(self, selected_regions, input_ids, attention_mask):
        """report_generation_model.py:196-210: rows of the (batch*29) sentences whose region was selected."""
        selected_regions = selected_regions.reshape(-1)
        return input_ids[selected_regions], attention_mask[selected_regions]

    @torch.no_grad()
    def generate(self, images: torch.FloatTensor, max_length: int = None, num_beams: int = 1, num_beam_groups: int = 1,
                 do_sample: bool = False, num_return_sequences: int = 1, early_stopping: bool = False):
        """images [B,1,512,512] -> (output_ids int64 [S,L'], selected_regions bool [B,29],
        detections {top_region_boxes [B,29,4], top_scores [B,29]}, class_detected bool [B,29]) or ``-1``
        when no region is both detected and selected (report_generation_model.py:260-261)."""
        _, detections, top_region_features, class_detected = self.object_detector(images)
        del images
        selected_regions, selected_region_features = self.binary_classifier_region_selection(
            top_region_features, class_detected, return_loss=False)
        del top_region_features
        if selected_region_features.shape[0] == 0:
            return -1
        output_ids = self.language_model.generate(selected_region_features, max_length, num_beams, num_beam_groups,
                                                  do_sample, num_return_sequences, early_stopping)
        del selected_region_features
        return output_ids, selected_regions, detections, class_detected
