"""Golden fixture for the eval-mode ``ReportGenerationModel.forward`` WITH ``image_targets`` - the call of the
reference's validation loop (evaluate_full_model/evaluate_model.py:413): the REAL reference (report_generation_model.py:88-168,
object_detector.py:216-261, custom_rpn.py:53-85, custom_roi_heads.py:210-269) run in the build container on seeded
synthetic weights / images / targets.  torchvision is absent: its training-target arithmetic (Matcher, BoxCoder.encode,
BalancedPositiveNegativeSampler, compute_loss, fastrcnn_loss) is the restatement in oracle/tv013.py behind tv_shim.py, so
this fixture pins the reference-authored glue (which proposals reach the RoI heads, dict order, eval-branch outputs) and
NOT that arithmetic (pinned by hand KATs only, tests/test_oracle_kats.py).  The samplers' torch.randperm draws are
replaced by a seeded generator (``perm_seed`` in the fixture) on both sides.

    python tests/golden/make_golden_forward_targets.py
"""
import os
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, REPO)
sys.path.insert(0, HERE)

import ref_harness  # noqa: E402
import tv_shim  # noqa: E402
from oracle import detector as o_det  # noqa: E402
from oracle import full_model as o_full  # noqa: E402
from rgrg_amd import synth  # noqa: E402

IMG_SEEDS = (1234, 77)
T = 10
PERM_SEED = 11


def seeded_perm(seed):
    g = torch.Generator().manual_seed(seed)
    return lambda n, tag: torch.randperm(n, generator=g)


def make_targets(sd, images):
    """Ground truth from the detector's own inference output (every other detected region, jittered by a few pixels)."""
    _, det, _, cd = o_det.object_detector_forward(sd, images)
    g = torch.Generator().manual_seed(17)
    targets = []
    for b in range(images.shape[0]):
        keep = cd[b].nonzero().flatten()[::2]
        boxes = det["top_region_boxes"][b, keep] + torch.randn((keep.numel(), 4), generator=g) * 2.0
        boxes[:, 2:] = torch.maximum(boxes[:, 2:], boxes[:, :2] + 4.0)
        targets.append({"boxes": boxes.clamp(0, 512), "labels": (keep + 1).to(torch.int64)})
    return targets


def make_inputs(sd):
    images = torch.cat([synth.make_images(1, s) for s in IMG_SEEDS], 0)
    g = torch.Generator().manual_seed(2025)
    n = len(IMG_SEEDS) * 29
    ids = torch.randint(0, 50257, (n, T), generator=g)
    ids[:, 0] = 50256
    mask = torch.ones((n, T), dtype=torch.int64)
    has_sentence = torch.rand((len(IMG_SEEDS), 29), generator=g) < 0.5
    is_abnormal = torch.rand((len(IMG_SEEDS), 29), generator=g) < 0.2
    return images, make_targets(sd, images), ids, mask, has_sentence, is_abnormal


def main():
    model = ref_harness.reference_model()
    model.pretrain_without_lm_model = False
    sd = synth.make_state_dict(0, "bench")
    model.load_state_dict(synth.to_reference_state_dict(sd), strict=True)
    images, targets, ids, mask, has_sentence, is_abnormal = make_inputs(sd)
    tv_shim.PERM_FN = seeded_perm(PERM_SEED)
    with torch.no_grad():
        ref = model(images, [dict(t) for t in targets], ids.clone(), mask.clone(), has_sentence, is_abnormal, return_loss=True)
    ora = o_full.forward_eval(sd, images, ids.clone(), mask.clone(), has_sentence, is_abnormal, image_targets=targets,
                              perm_fn=seeded_perm(PERM_SEED))
    ok = list(ref[0]) == list(ora[0]) == ["loss_classifier", "loss_box_reg", "loss_objectness", "loss_rpn_box_reg"]
    for k in ref[0]:
        d = abs(ref[0][k].item() - ora[0][k].item())
        print(f"{k}: reference {ref[0][k].item():.6f} oracle {ora[0][k].item():.6f} |d| {d:.2e}")
        ok &= d <= 1e-6
    for i, nme in ((1, "selection loss"), (2, "abnormal loss"), (3, "lm loss")):
        d = abs(ref[i].item() - ora[i].item())
        print(f"{nme}: reference {ref[i].item():.6f} oracle {ora[i].item():.6f} |d| {d:.2e}")
        ok &= d <= 1e-5
    for i in (5, 6, 7):
        ok &= torch.equal(ref[i], ora[i])
    ok &= torch.allclose(ref[4]["top_region_boxes"], ora[4]["top_region_boxes"], atol=1e-3)
    out = {"meta": {"torch": str(torch.__version__), "reference": "ttanida/rgrg @ /root/reference", "weights_seed": 0,
                    "profile": "bench", "image_seeds": list(IMG_SEEDS), "perm_seed": PERM_SEED,
                    "oracle_matches_reference": bool(ok)},
           "inputs": {"targets": targets, "input_ids": ids, "attention_mask": mask, "region_has_sentence": has_sentence,
                      "region_is_abnormal": is_abnormal},
           "expected": {"obj_detector_loss_dict": {k: v.clone() for k, v in ref[0].items()},
                        "classifier_loss_region_selection": ref[1].clone(), "classifier_loss_region_abnormal": ref[2].clone(),
                        "language_model_loss": ref[3].clone(), "top_region_boxes": ref[4]["top_region_boxes"],
                        "top_scores": ref[4]["top_scores"], "class_detected": ref[5], "selected_regions": ref[6],
                        "predicted_abnormal_regions": ref[7]}}
    torch.save(out, os.path.join(HERE, "forward_eval_targets_b2.pt"))
    print("saved forward_eval_targets_b2.pt; oracle matches reference:", ok)
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main())
