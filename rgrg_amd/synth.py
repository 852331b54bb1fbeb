"""Seeded synthetic weights and images for the RGRG inference path.

There is no network for the real checkpoint (ResNet-50 ImageNet weights,
``healx/gpt-2-pubmed-medium``, the authors' RGRG checkpoint) and MIMIC-CXR is
credentialed, so tests and ``bench.py`` run on random-init weights of the exact
architecture and on synthetic 512x512 "CXR-like" images.  The generated state
dict uses the reference's own key names (``ReportGenerationModel.state_dict()``
of ttanida/rgrg), so the same dict loads into the real reference
(``tests/golden/make_golden.py``), the CPU oracle and this package.

Initialisation follows the constructors' defaults in spirit (torchvision RPNHead
normal(0.01), nn.Linear uniform(+-1/sqrt(fan_in)), GPT-2 normal(0.02)) with two
documented deviations that keep random activations well-scaled and the workload
at its full size:
  * ResNet-50 BatchNorm running statistics are set to the analytically tracked
    second moment of their input (a trained net's BN does the same job);
  * profile "bench": ``cls_score`` rows are scaled so every one of the 29 regions
    wins at least one proposal, and the selection head's last bias is +4 so every
    detected region is selected  ->  S = 29*B sequences.
  * profile "ragged": every third region made rare, selection logits spread around
    the threshold (some regions undetected / unselected) and an EOS logit bias (see
    ``set_eos_bias``) so sequences finish at different steps.
"""
from __future__ import annotations

import math
from typing import Dict

import torch

Tensor = torch.Tensor

RESNET50_LAYERS = ((64, 3, 1), (128, 4, 2), (256, 6, 2), (512, 3, 2))
N_LAYER, D_MODEL, VOCAB = 24, 1024, 50257
IMAGE_MEAN, IMAGE_STD = 0.471, 0.302  # generate_reports_for_images.py:29-30


class _Rng:
    def __init__(self, seed: int):
        self.g = torch.Generator().manual_seed(seed)

    def normal(self, shape, std=1.0):
        return torch.randn(shape, generator=self.g, dtype=torch.float32) * std

    def uniform(self, shape, lo, hi):
        return torch.rand(shape, generator=self.g, dtype=torch.float32) * (hi - lo) + lo


def _linear(sd, rng, name, out_f, in_f, w_scale=1.0):
    b = 1.0 / math.sqrt(in_f)  # nn.Linear default (kaiming_uniform a=sqrt(5))
    sd[name + ".weight"] = rng.uniform((out_f, in_f), -b, b) * w_scale
    sd[name + ".bias"] = rng.uniform((out_f,), -b, b)


def _bn(sd, rng, name, ch, in_m2):
    sd[name + ".weight"] = rng.uniform((ch,), 0.8, 1.2)
    sd[name + ".bias"] = rng.normal((ch,), 0.1)
    sd[name + ".running_mean"] = rng.normal((ch,), 0.1 * math.sqrt(in_m2))
    sd[name + ".running_var"] = in_m2 * rng.uniform((ch,), 0.8, 1.25)
    sd[name + ".num_batches_tracked"] = torch.tensor(0, dtype=torch.int64)


def _conv(sd, rng, name, cout, cin, k):
    sd[name + ".weight"] = rng.normal((cout, cin, k, k), 1.0 / math.sqrt(cin * k * k))


def make_state_dict(seed: int = 0, profile: str = "bench") -> Dict[str, Tensor]:
    """Canonical (un-aliased) state dict with the reference's key names."""
    assert profile in ("bench", "ragged")
    rng = _Rng(seed)
    sd: Dict[str, Tensor] = {}
    # ---- object detector: ResNet-50 trunk (object_detector.py:51-58) ------------
    bb = "object_detector.backbone."
    _conv(sd, rng, bb + "0", 64, 1, 7)
    _bn(sd, rng, bb + "1", 64, 0.92)
    m2 = 1.6  # after ReLU + 3x3 max-pool
    inpl = 64
    for li, (planes, blocks, _stride) in enumerate(RESNET50_LAYERS):
        for b in range(blocks):
            p = f"{bb}{4 + li}.{b}."
            _conv(sd, rng, p + "conv1", planes, inpl, 1)
            _bn(sd, rng, p + "bn1", planes, m2)
            _conv(sd, rng, p + "conv2", planes, planes, 3)
            _bn(sd, rng, p + "bn2", planes, 0.55)
            _conv(sd, rng, p + "conv3", planes * 4, planes, 1)
            _bn(sd, rng, p + "bn3", planes * 4, 0.55)
            if b == 0:
                _conv(sd, rng, p + "downsample.0", planes * 4, inpl, 1)
                _bn(sd, rng, p + "downsample.1", planes * 4, m2)
                m2 = 1.0
            else:
                m2 = m2 + 0.5
            inpl = planes * 4
    # ---- RPN head (torchvision RPNHead init: normal(0.01), bias 0) --------------
    rp = "object_detector.rpn.head."
    sd[rp + "conv.0.0.weight"] = rng.normal((2048, 2048, 3, 3), 0.01)
    sd[rp + "conv.0.0.bias"] = torch.zeros(2048)
    sd[rp + "cls_logits.weight"] = rng.normal((160, 2048, 1, 1), 0.01)
    sd[rp + "cls_logits.bias"] = torch.zeros(160)
    sd[rp + "bbox_pred.weight"] = rng.normal((640, 2048, 1, 1), 0.01)
    sd[rp + "bbox_pred.bias"] = torch.zeros(640)
    # ---- RoI heads ---------------------------------------------------------------
    rh = "object_detector.roi_heads."
    _linear(sd, rng, rh + "box_head.fc6", 1024, 2048 * 64)
    _linear(sd, rng, rh + "box_head.fc7", 1024, 1024)
    _linear(sd, rng, rh + "box_predictor.cls_score", 30, 1024)
    _linear(sd, rng, rh + "box_predictor.bbox_pred", 120, 1024)
    _linear(sd, rng, rh + "dim_reduction", 1024, 2048)
    # ---- binary classifiers --------------------------------------------------------
    for nm, pw in (("binary_classifier_region_selection", 2.2), ("binary_classifier_region_abnormal", 6.0)):
        _linear(sd, rng, nm + ".classifier.0", 512, 1024)
        _linear(sd, rng, nm + ".classifier.2", 128, 512)
        _linear(sd, rng, nm + ".classifier.4", 1, 128, w_scale=(1.0 if profile == "bench" else 30.0))
        sd[nm + ".loss_fn.pos_weight"] = torch.tensor([pw])
    if profile == "bench":
        sd["binary_classifier_region_selection.classifier.4.bias"] = torch.tensor([4.0])
    else:
        sd["binary_classifier_region_selection.classifier.4.bias"] = torch.tensor([-1.0])
    # ---- language model (GPT-2 medium, normal(0.02)) -----------------------------
    g = "language_model.gpt_with_lm_head.transformer."
    wte = rng.normal((VOCAB, D_MODEL), 0.02)
    sd[g + "wte.weight"] = wte
    sd[g + "wpe.weight"] = rng.normal((1024, D_MODEL), 0.02)
    for l in range(N_LAYER):
        b = f"{g}h.{l}."
        sd[b + "ln_1.weight"] = rng.uniform((D_MODEL,), 0.9, 1.1)
        sd[b + "ln_1.bias"] = rng.normal((D_MODEL,), 0.02)
        sd[b + "attn.c_attn.weight"] = rng.normal((D_MODEL, 3 * D_MODEL), 0.02)
        sd[b + "attn.c_attn.bias"] = rng.normal((3 * D_MODEL,), 0.02)
        sd[b + "attn.c_proj.weight"] = rng.normal((D_MODEL, D_MODEL), 0.02)
        sd[b + "attn.c_proj.bias"] = rng.normal((D_MODEL,), 0.02)
        _linear(sd, rng, b + "attn.uk", D_MODEL, D_MODEL)
        _linear(sd, rng, b + "attn.uv", D_MODEL, D_MODEL)
        sd[b + "ln_2.weight"] = rng.uniform((D_MODEL,), 0.9, 1.1)
        sd[b + "ln_2.bias"] = rng.normal((D_MODEL,), 0.02)
        sd[b + "mlp.c_fc.weight"] = rng.normal((D_MODEL, 4 * D_MODEL), 0.02)
        sd[b + "mlp.c_fc.bias"] = rng.normal((4 * D_MODEL,), 0.02)
        sd[b + "mlp.c_proj.weight"] = rng.normal((4 * D_MODEL, D_MODEL), 0.02)
        sd[b + "mlp.c_proj.bias"] = rng.normal((D_MODEL,), 0.02)
    sd[g + "ln_f.weight"] = rng.uniform((D_MODEL,), 0.9, 1.1)
    sd[g + "ln_f.bias"] = rng.normal((D_MODEL,), 0.02)
    sd["language_model.gpt_with_lm_head.lm_head.weight"] = wte  # tied (language_model.py:205)
    _linear(sd, rng, "language_model.feature_space_transformation_nn.0", D_MODEL, D_MODEL)
    # x6: the image token carries enough variance that different regions decode to different sentences
    _linear(sd, rng, "language_model.feature_space_transformation_nn.2", D_MODEL, D_MODEL, w_scale=6.0)
    _calibrate_cls_score(sd, seed, profile)
    if profile == "ragged":
        set_eos_bias(sd, 2.6)
    return sd


def set_eos_bias(sd: Dict[str, Tensor], c: float) -> None:
    """Add ``c`` to the EOS logit of every decode step: lm_head has no bias, but
    ``ln_f.bias . wte[v]`` is a per-token constant, so ``ln_f.bias += c * e/|e|^2`` with
    ``e = wte[EOS]`` shifts the EOS logit by exactly ``c`` (others by ~c/32 * N(0,1)).
    c ~ 2.6 makes rows of a random-init decoder finish at different steps (or never);
    c >= 4 makes every row finish early."""
    g = "language_model.gpt_with_lm_head.transformer."
    e = sd[g + "wte.weight"][50256]
    sd[g + "ln_f.bias"] = sd[g + "ln_f.bias"] + c * e / (e * e).sum()


CLS_TEMPERATURE = 2.0


def _calibrate_cls_score(sd: Dict[str, Tensor], seed: int, profile: str) -> None:
    """Fold a per-class affine map into ``cls_score`` so that, on the calibration
    image, every class's logit over the proposals is ~N(0, T^2): random-init heads
    otherwise give one class the arg-max for every box (class_detected would be 1 of
    29).  The per-class mean/std of the raw logits were measured once with the CPU
    oracle by ``tests/golden/calibrate_synth.py`` and are stored in
    ``rgrg_amd/data/synth_calib_seed{seed}.pt`` (240 bytes)."""
    import os
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data", f"synth_calib_seed{seed}.pt")
    if not os.path.exists(path):
        return  # un-calibrated seed: heads stay at their default init
    cal = torch.load(path)
    k = "object_detector.roi_heads.box_predictor.cls_score."
    scale = CLS_TEMPERATURE / cal["std"]
    offset = torch.zeros(30)
    offset[0] = -3.0 * CLS_TEMPERATURE  # background rarely dominates the softmax
    if profile == "ragged":
        offset[1:] = -2.5 * CLS_TEMPERATURE * (torch.arange(29) % 3 == 0).float()  # every 3rd region is rare
    sd[k + "weight"] = sd[k + "weight"] * scale[:, None]
    sd[k + "bias"] = (sd[k + "bias"] - cal["mean"]) * scale + offset


def to_reference_state_dict(sd: Dict[str, Tensor]) -> Dict[str, Tensor]:
    """Expand the canonical dict to the full aliased key set of the reference's
    ``ReportGenerationModel.state_dict()`` (three aliased copies of every GPT-2
    tensor + the pseudo-attention buffers)."""
    out = dict(sd)
    g = "language_model.gpt_with_lm_head.transformer."
    causal = torch.tril(torch.ones((1024, 1024), dtype=torch.uint8)).view(1, 1, 1024, 1024)
    for l in range(N_LAYER):
        out[f"{g}h.{l}.attn.causal_mask"] = causal
        out[f"{g}h.{l}.attn.mask_out_value"] = torch.tensor(-1e4)
    sub = {"ln_1": "0", "attn": "1", "ln_2": "2", "mlp": "3"}
    for k in [k for k in out if k.startswith(g)]:
        rest = k[len(g):]
        out["language_model.gpt." + rest] = out[k]
        parts = rest.split(".")
        if parts[0] == "h":
            out[f"language_model.gpt2_blocks.{parts[1]}.{sub[parts[2]]}." + ".".join(parts[3:])] = out[k]
        elif parts[0] in ("wte", "wpe"):
            out[f"language_model.{parts[0]}.weight"] = out[k]
        elif parts[0] == "ln_f":
            out["language_model.final_layernorm." + parts[1]] = out[k]
    out["language_model.lm_head.weight"] = out["language_model.gpt_with_lm_head.lm_head.weight"]
    return out


def make_images(batch: int, seed: int = 1234) -> Tensor:
    """Synthetic grayscale CXR-like batch ``[B,1,512,512]`` float32, normalised as
    ``generate_reports_for_images.py:129-147`` does: uint8 pixels -> /255 ->
    (x-0.471)/0.302.  Pixels = uniform noise blended with random axis-aligned
    rectangles so that proposals/regions spread over the image."""
    g = torch.Generator().manual_seed(seed)
    imgs = torch.empty((batch, 1, 512, 512), dtype=torch.float32)
    for b in range(batch):
        noise = torch.randint(0, 256, (512, 512), generator=g, dtype=torch.int64).to(torch.float32)
        canvas = torch.full((512, 512), 96.0)
        for _ in range(40):
            x0, y0 = (int(v) for v in torch.randint(0, 448, (2,), generator=g))
            w, h = (int(v) for v in torch.randint(24, 256, (2,), generator=g))
            lvl = float(torch.randint(0, 256, (1,), generator=g))
            canvas[y0:min(512, y0 + h), x0:min(512, x0 + w)] = lvl
        u8 = torch.floor(0.35 * noise + 0.65 * canvas).clamp(0, 255)
        imgs[b, 0] = (u8 / 255.0 - IMAGE_MEAN) / IMAGE_STD
    return imgs
