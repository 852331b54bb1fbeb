"""torch.autocast(float16) - the dtype of the reference's own scripts (generate_reports_for_images.py:108,
test_set_evaluation.py:287, train_full_model.py:172) - runs the reduced-precision HIP path in IEEE fp16 (round 4: until
then it was mapped onto the bf16 kernels): v_mfma_f32_32x32x16_f16 GEMMs / convolutions on fp16 weights and activations,
an fp16 K/V cache, fp16 RoIAlign maps; fp32 accumulation, LayerNorm, softmax, residual stream.  fp16 carries 3 more
mantissa bits than bf16, so the noise-level bounds here are ~8x tighter than in the bf16 tests of the same shape
(tests/test_gpu_parity_gaps.py, tests/test_gpu_parity_r03.py); the oracle's float16 mode (oracle/language_model.py,
bf16=2) is pinned against the REAL reference under torch.autocast("cpu", float16) by tests/golden/lm_autocast_fp16.pt."""
import json
import os
import subprocess
import sys

import pytest
import torch

from conftest import REPO, gpu_model, synth_sd
from oracle import language_model as o_lm
from rgrg_amd import _hip, synth

pytestmark = pytest.mark.gpu
DEV = "cuda:0"

_SCRIPT = r"""
import json, sys, torch
sys.path.insert(0, {repo!r})
sys.path.insert(0, {repo!r} + "/tests")
from conftest import gpu_model, synth_sd
from oracle import language_model as o_lm
S, L = {S}, {L}
g = torch.Generator().manual_seed(35)
feats = torch.randn((S, 1024), generator=g)
m = gpu_model("bench")
with torch.autocast("cuda", dtype=torch.{d1}):
    ids = m.language_model.generate(feats.to("cuda:0"), max_length=L)
hip = m.engine().last_logits(S).cpu()
with torch.autocast("cuda", dtype=torch.{d2}):      # switching the 16-bit type re-converts the weight copies
    ids_b = m.language_model.generate(feats.to("cuda:0"), max_length=L)
with torch.autocast("cuda", dtype=torch.{d1}):       # ... and back: bit-identical to the first run
    ids_again = m.language_model.generate(feats.to("cuda:0"), max_length=L)
ids = ids.cpu()
sd = synth_sd("bench")
T = ids.shape[1] - 1
pos = torch.arange(T)[None, :]
am = torch.ones((S, T), dtype=torch.int64)
with torch.no_grad():
    lo16, _ = o_lm.lm_forward(sd, ids[:, :T], am, feats, None, pos, bf16={ob})
    lo32, _ = o_lm.lm_forward(sd, ids[:, :T], am, feats, None, pos, bf16=False)
lo16, lo32 = lo16[:, -1], lo32[:, -1]
rng = lo32.abs().max().item()
print(json.dumps(dict(
    len=int(ids.shape[1]), keys=T + 1, range=rng, same_after_switching=bool(torch.equal(ids_again.cpu(), ids)),
    differs_from_bf16=bool((ids_b.cpu() != ids).any()),
    err_vs_fp16_oracle=(hip - lo16).abs().max().item(), err_vs_fp32_oracle=(hip - lo32).abs().max().item(),
    argmax_agree_fp16_oracle=(hip.argmax(-1) == lo16.argmax(-1)).float().mean().item(),
    next_token_is_argmax=(ids[:, -1] == hip.argmax(-1)).float().mean().item())))
"""


def test_fp16_decode_100_steps_against_fp16_oracle():
    """The fp16 many-row decode path (fp16-weight MFMA GEMMs, fp16 K/V cache incl. the second key chunk of
    attn_decode_kv16_wave_kernel) for 100 tokens against the ORACLE doing the same fp16 arithmetic, teacher-forced on the
    token history the HIP path chose.  RGRG_SKINNY_MAX_ROWS = 32 in a child process so that 40 rows take the 16-bit path."""
    env = dict(os.environ, RGRG_SKINNY_MAX_ROWS="32")
    res = subprocess.run([sys.executable, "-c", _SCRIPT.format(repo=REPO, S=40, L=100, d1="float16", d2="bfloat16", ob=2)], env=env, capture_output=True, text=True, timeout=1500)
    assert res.returncode == 0, res.stderr[-2000:]
    r = json.loads(res.stdout.strip().splitlines()[-1])
    assert r["len"] == 100 and r["keys"] == 100
    # fp16 quantisation noise after 24 blocks: ~0.2 % of the logit range (the reference's own fp16-vs-fp32 distance in
    # lm_autocast_fp16.pt is 0.14 %); a wrong slot / scale / type shows as O(range), the bf16 kernels as ~1-2 %
    assert r["err_vs_fp16_oracle"] <= 4e-3 * r["range"], r
    assert r["err_vs_fp32_oracle"] <= 5e-3 * r["range"], r
    assert r["argmax_agree_fp16_oracle"] >= 0.95 - 1e-6 and r["next_token_is_argmax"] == 1.0, r
    assert r["same_after_switching"] and r["differs_from_bf16"], r


@pytest.mark.parametrize("S,L,dtype", [(40, 64, "float16"), (64, 40, "float16"), (58, 48, "bfloat16")])
def test_fused_plan_on_16bit_weights_33_to_64_rows(S, L, dtype):
    """33-64 rows under autocast (greedy batch 2; round 6): the fused fragment-direct plan on 16-bit weight fragments
    (skinny_direct.inc W16: v_mfma_f32_16x16x16_{f16,bf16} on weights rounded once and activations rounded on their way into the
    matrix core; fp32 K/V cache, attention, LayerNorm statistics and residual stream) against the oracle in the same 16-bit type,
    teacher-forced on the tokens the HIP path chose.  Same noise-level bounds as the many-sequence path of that type (fp16: above;
    bf16: tests/test_gpu_parity_r03.py); 64 rows = the last row count on this plan."""
    env = {k: v for k, v in os.environ.items() if k not in ("RGRG_SKINNY_MAX_ROWS", "RGRG_SKINNY_MAX_ROWS_16", "RGRG_W16_FUSED")}
    f16 = dtype == "float16"
    script = _SCRIPT.format(repo=REPO, S=S, L=L, d1=dtype, d2="bfloat16" if f16 else "float16", ob=2 if f16 else "True")
    res = subprocess.run([sys.executable, "-c", script], env=env, capture_output=True, text=True, timeout=1500)
    assert res.returncode == 0, res.stderr[-2000:]
    r = json.loads(res.stdout.strip().splitlines()[-1])
    assert r["len"] == L and r["keys"] == L
    tol16, tol32, agree = (4e-3, 5e-3, 0.95) if f16 else (2e-2, 3e-2, 0.85)
    assert r["err_vs_fp16_oracle"] <= tol16 * r["range"], r
    assert r["err_vs_fp32_oracle"] <= tol32 * r["range"], r
    assert r["argmax_agree_fp16_oracle"] >= agree - 1e-6 and r["next_token_is_argmax"] == 1.0, r
    assert r["same_after_switching"] and r["differs_from_bf16"], r


def _every_step_is_the_oracles_argmax(ids, tr, tie):
    agree = tr["top_idx"][:, :, 0] == ids[:, 1:]
    margin = tr["top_val"][:, :, 0] - tr["chosen"]
    return agree, (agree | (margin <= tie))


def test_configs2_fp16_full_size_against_fp16_oracle():
    """BASELINE configs[2] under the REFERENCE'S dtype: 32 images, 'bench' weights (all 29 regions, 127 steps each),
    torch.autocast(float16).  Shapes / BOS / no PAD; 14 of the ~923 rows pinned against the fp16 oracle, teacher-forced:
    last-step logits at the fp16 noise level, >= 97 % of the 127 x 14 chosen tokens are the oracle's arg-max and the rest
    are noise-level ties; generate() == its three stages."""
    m = gpu_model("bench")
    sd = synth_sd("bench")
    images = synth.make_images(32, 1234).to(DEV)
    with torch.autocast("cuda", dtype=torch.float16):
        ids, sel, det, cd = m.generate(images, max_length=128)
        _, _, top, cd2 = m.object_detector(images)
        sel2, feats = m.binary_classifier_region_selection(top, cd2, return_loss=False)
        ids2 = m.language_model.generate(feats, 128)
    S = int(sel.sum())
    assert S >= 900 and ids.shape == (S, 128) and (ids[:, 0] == 50256).all()
    assert torch.equal(sel, sel2) and torch.equal(ids, ids2)
    last = m.engine().last_logits(S).cpu()
    rows = [0, 1, 31, 32, 127, 128, 300, 461, 462, 600, 800, S - 33, S - 2, S - 1]
    idc, fc = ids.cpu(), feats.float().cpu()
    tr = o_lm.teacher_forced_trace(sd, idc[rows], fc[rows], bf16=2)
    rng = tr["last_logits"].abs().max().item()
    err = (last[rows] - tr["last_logits"]).abs().max().item()
    assert err <= 4e-3 * rng, (err, rng)
    agree, ok = _every_step_is_the_oracles_argmax(idc[rows], tr, tie=6e-3 * rng)
    assert ok.all(), (~ok).nonzero().tolist()[:8]
    assert agree.float().mean().item() >= 0.97, agree.float().mean().item()


def test_detector_under_fp16_autocast_close_to_fp32():
    """The detector under torch.autocast(float16): bottlenecks / RPN convolutions / fc6 on v_mfma_f32_32x32x16_f16, fp16
    RoIAlign maps.  Same stage-by-stage comparison as the bf16 test, with bounds 4x tighter (fp16: 11 significand bits)."""
    m = gpu_model("bench")
    images = synth.make_images(2, 1234).to(DEV)
    eng = m.engine()
    t32, t16, tb = {}, {}, {}
    d32, f32_, cd32 = eng.detect(images, t32)
    d16, f16_, cd16 = eng.detect(images, t16, bf16=2)
    db, fb_, cdb = eng.detect(images, tb, bf16=1)
    span = t32["features_nhwc"].abs().max().item()
    err = (t16["features_nhwc"] - t32["features_nhwc"]).abs().max().item()
    err_b = (tb["features_nhwc"] - t32["features_nhwc"]).abs().max().item()
    assert 0.0 < err <= 8e-3 * span and err < 0.5 * err_b, (err, err_b, span)       # really fp16, not the bf16 kernels
    feat32 = t32["features_nhwc"]
    p32, p16 = {}, {}
    eng.roi_heads(feat32, t32["proposals"], t32["offsets"], p32)
    eng.roi_heads(feat32, t32["proposals"], t32["offsets"], p16, bf16=2)
    pspan = p32["pred"].abs().max().item()
    perr = (p16["pred"] - p32["pred"]).abs().max().item()
    assert 0.0 < perr <= 5e-3 * pspan, (perr, pspan)
    # the 8x8 average is formed from the unrounded bins: fp32-accurate (the 16-bit variants of the kernel contract their
    # multiply-adds - like the reference's own GPU kernel -, so not bit-identical to the unfused fp32 variant)
    assert p16["pooled_maps"].dtype == torch.int16
    assert torch.allclose(p16["pooled"], p32["pooled"], rtol=1e-5, atol=1e-6)
    # the stored maps ARE fp16 roundings of fp32-accurate bins: within one fp16 ulp of the rounded fp32 maps, mostly identical
    a16, b16 = p16["pooled_maps"].view(torch.float16).float(), p32["pooled_maps"].to(torch.float16).float()
    assert bool(((a16 - b16).abs() <= 2.0 ** -10 * b16.abs().clamp(min=2.0 ** -14)).all()) and float((a16 == b16).float().mean()) > 0.99
    assert torch.equal(cd16, cd32)
    with torch.autocast("cuda", dtype=torch.float16):
        _, det_a, _, cd_a = m.object_detector(images)
    assert torch.equal(cd_a, cd16) and torch.equal(det_a["top_region_boxes"], d16["top_region_boxes"])
    assert _hip.autocast_mode() == 0
    with torch.autocast("cuda", dtype=torch.float16):
        assert _hip.autocast_mode() == 2
    with torch.autocast("cuda", dtype=torch.bfloat16):
        assert _hip.autocast_mode() == 1
