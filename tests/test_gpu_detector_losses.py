"""ObjectDetector.forward / ReportGenerationModel.forward in eval mode WITH image_targets (the
call of the reference's validation loop, evaluate_model.py:413): target assignment, sampling and the four detector
losses on the HIP path against the CPU oracle.  torchvision's samplers draw with torch.randperm: the oracle (like the
real reference behind the fixture) gets a seeded permutation, its samplers' choices are recorded and replayed on the HIP
path as per-element keys (chosen: 0, everything else: 1 - the device-side sampler takes the smallest keys), so both sides
score the same anchors / proposals.  Match codes are compared bit-exactly, losses to 5e-4 relative."""
import ctypes as C

import pytest
import torch

from conftest import gpu_model, load_golden, synth_sd
from oracle import detector as o_det
from oracle import full_model as o_full
from oracle import tv013
from rgrg_amd import _hip, synth

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _perm(seed):
    g = torch.Generator().manual_seed(seed)
    return lambda n, tag: torch.randperm(n, generator=g)


class _RecordedDraws:
    """Context manager around an oracle run: records which indices tv013.balanced_sample took per (stage, image);
    keys_fn replays them on the HIP path (ObjectDetector.sampler_keys)."""

    def __init__(self):
        self.rec = {}

    def __enter__(self):
        self._orig = tv013.balanced_sample

        def wrap(labels, batch, frac, perm_fn, tag):
            p, q = self._orig(labels, batch, frac, perm_fn, tag)
            self.rec[tag] = (p.clone(), q.clone())
            return p, q
        tv013.balanced_sample = wrap
        return self

    def __exit__(self, *exc):
        tv013.balanced_sample = self._orig

    def keys_fn(self, stage, B, n):
        keys = torch.ones((B, n))
        for b in range(B):
            p, q = self.rec[(stage, b)]
            keys[b, p] = 0.0
            keys[b, q] = 0.0
        return keys.to(DEV)


def _targets(sd, images):
    """Ground truth derived from the detector's own inference output: the box of every detected region, jittered, with
    its class label - so that anchors / proposals really match (positives, low-quality matches, background)."""
    _, det, _, cd = o_det.object_detector_forward(sd, images)
    g = torch.Generator().manual_seed(17)
    targets = []
    for b in range(images.shape[0]):
        keep = cd[b].nonzero().flatten()[::2]                       # every other detected region: some classes have no gt
        boxes = det["top_region_boxes"][b, keep] + torch.randn((keep.numel(), 4), generator=g) * 2.0
        boxes[:, 2:] = torch.maximum(boxes[:, 2:], boxes[:, :2] + 4.0)
        targets.append({"boxes": boxes.clamp(0, 512), "labels": (keep + 1).to(torch.int64)})
    return targets


def test_box_match_kernel_bit_exact_incl_ties_and_empty_images():
    lib = _hip.load()
    g = torch.Generator().manual_seed(5)
    B, G, N = 3, 7, 5000
    xy = torch.rand((B, N, 2), generator=g) * 400
    boxes = torch.cat([xy, xy + 5 + torch.rand((B, N, 2), generator=g) * 150], -1)
    gxy = torch.rand((B, G, 2), generator=g) * 300
    gt = torch.cat([gxy, gxy + 20 + torch.rand((B, G, 2), generator=g) * 180], -1)
    boxes[0, 10] = gt[0, 2]                                           # exact hit
    boxes[0, 11] = boxes[0, 12] = gt[0, 3] + torch.tensor([0., 0., 40., 40.])   # two boxes tie gt 3's best IoU (< 0.7)
    gt_count = torch.tensor([G, 4, 0], dtype=torch.int32)              # image 2 has no ground truth
    box_count = torch.tensor([N, N - 100, N], dtype=torch.int32)
    d_gt, d_gc, d_bx, d_bc = gt.to(DEV).contiguous(), gt_count.to(DEV), boxes.to(DEV).contiguous(), box_count.to(DEV)
    for hi, lo, lq in ((0.7, 0.3, True), (0.5, 0.5, False)):
        matched = torch.empty((B, N), dtype=torch.int32, device=DEV)
        ws = torch.empty((B, G), dtype=torch.int32, device=DEV)
        _hip.check(lib.rgrg_box_match_f32(d_gt.data_ptr(), d_gc.data_ptr(), G, d_bx.data_ptr(), N * 4, d_bc.data_ptr(), B, N, hi, lo,
                                          int(lq), matched.data_ptr(), ws.data_ptr(), torch.cuda.current_stream().cuda_stream),
                   "rgrg_box_match_f32")
        torch.cuda.synchronize()
        for b in range(B):
            n, ng = int(box_count[b]), int(gt_count[b])
            got = matched[b].cpu()
            if ng == 0:
                assert (got == -1).all()
                continue
            ref = tv013.matcher(tv013.box_iou(gt[b, :ng], boxes[b, :n]), hi, lo, lq)
            assert torch.equal(got[:n].to(torch.int64), ref), (b, hi)
            assert (got[n:] == -1).all()


def test_detector_eval_forward_with_targets_matches_oracle():
    m = gpu_model("bench")
    sd = synth_sd("bench")
    images = torch.cat([synth.make_images(1, 1234), synth.make_images(1, 77)], 0)
    targets = _targets(sd, images)
    targets[1] = {"boxes": targets[1]["boxes"][:5], "labels": targets[1]["labels"][:5]}
    with _RecordedDraws() as draws:
        ref_losses, ref_det, ref_top, ref_cd = o_det.object_detector_forward(sd, images, targets=targets, perm_fn=_perm(3))
    det = m.object_detector
    det.sampler_keys = draws.keys_fn
    try:
        losses, dets, top, cd = det(images.to(DEV), [{k: v.to(DEV) for k, v in t.items()} for t in targets])
    finally:
        det.sampler_keys = None
    assert list(losses) == list(ref_losses) == ["loss_classifier", "loss_box_reg", "loss_objectness", "loss_rpn_box_reg"]
    for k in losses:  # 5e-4 relative: the class logits come out of fc6 (K = 131072, fp32, different summation order)
        a, b = float(losses[k]), float(ref_losses[k])
        assert abs(a - b) <= 5e-4 * max(1.0, abs(b)), (k, a, b)
    assert torch.equal(cd.cpu(), ref_cd)
    both = ref_cd
    assert (dets["top_region_boxes"].cpu() - ref_det["top_region_boxes"])[both].abs().max() <= 5e-2
    assert (dets["top_scores"].cpu() - ref_det["top_scores"]).abs().max() <= 1e-4
    err = (top.cpu() - ref_top)[both].abs().max().item()
    assert err <= 1e-3 * ref_top.abs().max().item() + 1e-4, err
    # an image without any ground truth: every sampled proposal is background, box losses of that image vanish
    empty = [{"boxes": torch.zeros((0, 4)), "labels": torch.zeros((0,), dtype=torch.int64)} for _ in range(2)]
    with _RecordedDraws() as draws:
        ref_e, _, _, _ = o_det.object_detector_forward(sd, images, targets=empty, perm_fn=_perm(4))
    det.sampler_keys = draws.keys_fn
    try:
        le, _, _, _ = det(images.to(DEV), [{k: v.to(DEV) for k, v in t.items()} for t in empty])
    finally:
        det.sampler_keys = None
    # the default draws (torch.rand keys on the device): a valid run with finite classification losses
    ld, _, _, _ = det(images.to(DEV), [{k: v.to(DEV) for k, v in t.items()} for t in targets])
    assert all(torch.isfinite(v) for v in ld.values()) and abs(float(ld["loss_objectness"]) - float(ref_losses["loss_objectness"])) < 0.5
    assert float(le["loss_box_reg"]) == 0.0 and float(le["loss_rpn_box_reg"]) == 0.0
    for k in le:
        assert abs(float(le[k]) - float(ref_e[k])) <= 5e-4 * max(1.0, abs(float(ref_e[k]))), k


def test_full_model_eval_forward_accepts_image_targets_like_the_validation_loop():
    """evaluate_model.py:413: model(images, image_targets, input_ids, attention_mask, region_has_sentence,
    region_is_abnormal) in eval mode -> the 8-tuple with the four detector losses in obj_detector_loss_dict."""
    m = gpu_model("bench")
    sd = synth_sd("bench")
    images = synth.make_images(1, 1234)
    targets = _targets(sd, images)
    g = torch.Generator().manual_seed(9)
    T = 12
    input_ids = torch.randint(0, 50000, (29, T), generator=g)
    attention_mask = torch.ones((29, T))
    has_sentence = torch.rand((1, 29), generator=g) > 0.3
    abnormal = torch.rand((1, 29), generator=g) > 0.6
    with _RecordedDraws() as draws:   # the same call through the oracle first: its draws are replayed below
        ref = o_full.forward_eval(sd, images, input_ids, attention_mask, has_sentence, abnormal, image_targets=targets, perm_fn=_perm(6))
    m.object_detector.sampler_keys = draws.keys_fn
    was = m.pretrain_without_lm_model   # the shared test model is built with it set (7-tuple, no LM loss): full model here
    m.pretrain_without_lm_model = False
    try:
        out = m(images.to(DEV), [{k: v.to(DEV) for k, v in t.items()} for t in targets], input_ids.to(DEV), attention_mask.to(DEV),
                has_sentence.to(DEV), abnormal.to(DEV))
    finally:
        m.object_detector.sampler_keys = None
        m.pretrain_without_lm_model = was
    assert isinstance(out, tuple) and len(out) == 8
    loss_dict, l_sel, l_abn, l_lm, dets, cd, sel, pred_abn = out
    assert sorted(loss_dict) == ["loss_box_reg", "loss_classifier", "loss_objectness", "loss_rpn_box_reg"]
    assert all(torch.isfinite(v) for v in loss_dict.values()) and torch.isfinite(l_lm)
    for k in loss_dict:
        assert abs(float(loss_dict[k]) - float(ref[0][k])) <= 5e-4 * max(1.0, abs(float(ref[0][k]))), k
    assert torch.equal(cd.cpu(), ref[5]) and torch.equal(sel.cpu(), ref[6]) and torch.equal(pred_abn.cpu(), ref[7])
    assert abs(float(l_sel) - float(ref[1])) <= 1e-4 and abs(float(l_abn) - float(ref[2])) <= 1e-4
    assert abs(float(l_lm) - float(ref[3])) <= 5e-4 * max(1.0, abs(float(ref[3])))


def test_full_model_eval_forward_with_targets_matches_reference_fixture():
    """The same call against the fixture generated by the REAL reference (tests/golden/make_golden_forward_targets.py)."""
    fx = load_golden("forward_eval_targets_b2.pt")
    m = gpu_model(fx["meta"]["profile"])
    images = torch.cat([synth.make_images(1, s) for s in fx["meta"]["image_seeds"]], 0)
    i, e = fx["inputs"], fx["expected"]
    # the samplers' choices of the reference run = the oracle's with the same seeded permutation (the oracle reproduces the
    # fixture, tests/test_oracle_golden.py): recorded from the oracle's detector pass, replayed as keys
    with _RecordedDraws() as draws:
        o_det.object_detector_forward(synth_sd(fx["meta"]["profile"]), images, targets=i["targets"], perm_fn=_perm(fx["meta"]["perm_seed"]))
    m.object_detector.sampler_keys = draws.keys_fn
    was = m.pretrain_without_lm_model
    m.pretrain_without_lm_model = False
    try:
        out = m(images.to(DEV), [{k: v.to(DEV) for k, v in t.items()} for t in i["targets"]], i["input_ids"].to(DEV),
                i["attention_mask"].to(DEV), i["region_has_sentence"].to(DEV), i["region_is_abnormal"].to(DEV))
    finally:
        m.object_detector.sampler_keys = None
        m.pretrain_without_lm_model = was
    assert list(out[0]) == list(e["obj_detector_loss_dict"])
    for k, v in e["obj_detector_loss_dict"].items():
        assert abs(float(out[0][k]) - float(v)) <= 5e-4 * max(1.0, abs(float(v))), k
    assert abs(float(out[1]) - float(e["classifier_loss_region_selection"])) <= 1e-4
    assert abs(float(out[2]) - float(e["classifier_loss_region_abnormal"])) <= 1e-4
    assert abs(float(out[3]) - float(e["language_model_loss"])) <= 5e-4 * float(e["language_model_loss"])
    assert torch.equal(out[5].cpu(), e["class_detected"]) and torch.equal(out[6].cpu(), e["selected_regions"])
    assert torch.equal(out[7].cpu(), e["predicted_abnormal_regions"])
    assert (out[4]["top_region_boxes"].cpu() - e["top_region_boxes"])[e["class_detected"]].abs().max() <= 5e-2
