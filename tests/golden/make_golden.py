"""Generate the golden fixtures in ``tests/golden/*.pt`` by running the REAL
reference (``/root/reference``, imported through ``ref_harness``) on the seeded
synthetic weights/images of ``rgrg_amd.synth``, and check the CPU oracle against
it on the spot.  Build-container only (the reference cannot travel):

    python tests/golden/make_golden.py

Fixtures are data only: seeds/profile (inputs are regenerated from them), and
the reference's outputs.
"""
from __future__ import annotations

import os
import sys
import time

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, REPO)
sys.path.insert(0, HERE)

import ref_harness  # noqa: E402
from oracle import full_model as o_full  # noqa: E402
from oracle import language_model as o_lm  # noqa: E402
from oracle import detector as o_det  # noqa: E402
from rgrg_amd import synth  # noqa: E402

torch.manual_seed(0)


def run_reference_generate(model, images, max_length):
    out = model.generate(images, max_length=max_length, num_beams=1)
    if isinstance(out, int):
        return {"sentinel": out}
    ids, sel, det, cd = out
    return {"output_ids": ids, "selected_regions": sel, "top_region_boxes": det["top_region_boxes"],
            "top_scores": det["top_scores"], "class_detected": cd}


def reference_detector_intermediates(model, images):
    od = model.object_detector
    feat = od.backbone(images)
    il, fd = od._transform_inputs_for_rpn_and_roi(images, feat)
    proposals, _ = od.rpn(il, fd, None)
    _, det, trf, cd = od(images)
    return {"feat_mean": feat.mean(), "feat_std": feat.std(), "feat_sample": feat[:, ::64, ::3, ::3].clone(),
            "proposals": [p.clone() for p in proposals], "top_region_features": trf, "class_detected": cd,
            "top_region_boxes": det["top_region_boxes"], "top_scores": det["top_scores"]}


def check(name, a, b, atol=0.0, rtol=0.0):
    if a.dtype in (torch.bool, torch.int64):
        ok = torch.equal(a, b)
        err = 0.0 if ok else float("nan")
    else:
        err = (a - b).abs().max().item() if a.numel() else 0.0
        ok = torch.allclose(a, b, atol=atol, rtol=rtol)
    print(f"  oracle-vs-reference {name}: {'OK' if ok else 'MISMATCH'} (max abs err {err:.3e})")
    return ok


def main():
    t0 = time.time()
    model = ref_harness.reference_model()
    meta = {"torch": str(torch.__version__), "reference": "ttanida/rgrg @ /root/reference"}
    all_ok = True

    for case, profile, img_seeds, max_length in (("bench_b1_len128", "bench", (1234,), 128),
                                                 ("ragged_b2_len24", "ragged", (77, 5), 24)):
        print(f"[{case}] weights profile={profile} ({time.time() - t0:.0f}s)")
        sd = synth.make_state_dict(0, profile)
        model.load_state_dict(synth.to_reference_state_dict(sd), strict=True)
        images = torch.cat([synth.make_images(1, s) for s in img_seeds], 0)
        with torch.no_grad():
            ref_det = reference_detector_intermediates(model, images)
            ref = run_reference_generate(model, images, max_length)
        # oracle vs reference (pins every reference-authored stage of the restatement)
        o = o_det.object_detector_forward(sd, images, return_intermediates=True)
        ok = check("features", o["_features"][:, ::64, ::3, ::3], ref_det["feat_sample"], 1e-5, 1e-5)
        for i, (a, b) in enumerate(zip(o["_proposals"], ref_det["proposals"])):
            ok &= a.shape == b.shape and check(f"proposals[{i}]", a, b, 1e-4, 1e-5)
        ok &= check("class_detected", o["class_detected"], ref_det["class_detected"])
        ok &= check("top_region_features", o["top_region_features"], ref_det["top_region_features"], 1e-4, 1e-4)
        ok &= check("top_region_boxes", o["detections"]["top_region_boxes"], ref_det["top_region_boxes"], 1e-3, 1e-5)
        ok &= check("top_scores", o["detections"]["top_scores"], ref_det["top_scores"], 1e-5, 1e-4)
        og = o_full.generate(sd, images, max_length)
        ok &= check("selected_regions", og[1], ref["selected_regions"])
        ok &= og[0].shape == ref["output_ids"].shape and check("output_ids", og[0], ref["output_ids"])
        all_ok &= ok
        fx = {"meta": dict(meta, weights_seed=0, profile=profile, image_seeds=list(img_seeds), max_length=max_length,
                           oracle_matches_reference=bool(ok)),
              "detector": ref_det, "generate": ref}
        torch.save(fx, os.path.join(HERE, f"{case}.pt"))
        print(f"  saved {case}.pt  ids shape {tuple(ref['output_ids'].shape)}  S={int(ref['selected_regions'].sum())}")

        if case.startswith("ragged"):
            # language-model-only fixture: first-step logits of the real LanguageModel.forward + short greedy run
            g = torch.Generator().manual_seed(99)
            feats = torch.randn((5, 1024), generator=g)
            with torch.no_grad():
                lm = model.language_model
                ids0 = torch.full((5, 1), 50256, dtype=torch.int64)
                logits, _ = lm.forward(ids0, torch.ones((5, 1), dtype=torch.int64), feats, return_loss=False,
                                       position_ids=torch.zeros((5, 1), dtype=torch.int64), use_cache=True)
                gen = lm.generate(feats, max_length=12, num_beams=1)
            o_ids, o_logits = o_lm.greedy_generate(sd, feats, 12, return_logits=True)
            ok = check("lm step0 logits", o_logits[:, 0], logits[:, 0], 1e-4, 1e-4)
            ok &= o_ids.shape == gen.shape and check("lm ids", o_ids, gen)
            all_ok &= ok
            torch.save({"meta": dict(meta, weights_seed=0, profile=profile, feat_seed=99, max_length=12,
                                     oracle_matches_reference=bool(ok)),
                        "step0_logits_sample": logits[:, 0, ::101].clone(), "step0_argmax": logits[:, 0].argmax(-1),
                        "step0_top2_gap": (lambda t: t[:, 0] - t[:, 1])(logits[:, 0].topk(2, -1).values),
                        "output_ids": gen}, os.path.join(HERE, "lm_only_len12.pt"))

            # every row finishes before max_length: early exit, L' < max_length
            sd_fin = dict(sd)
            # (the ragged profile's EOS bias makes all 5 rows emit EOS by step 15 -> L' = 16 < 40)
            model.load_state_dict(synth.to_reference_state_dict(sd_fin), strict=True)
            with torch.no_grad():
                gen_fin = model.language_model.generate(feats, max_length=40, num_beams=1)
            o_fin = o_lm.greedy_generate(sd_fin, feats, 40)
            ok = o_fin.shape == gen_fin.shape and check("lm all-finish ids", o_fin, gen_fin)
            all_ok &= ok
            print("  all-finish L' =", gen_fin.shape[1])
            assert gen_fin.shape[1] < 40
            torch.save({"meta": dict(meta, weights_seed=0, profile=profile, feat_seed=99, max_length=40,
                                     oracle_matches_reference=bool(ok)), "output_ids": gen_fin},
                       os.path.join(HERE, "lm_only_allfinish.pt"))

            # S == 0 sentinel: selection bias driven to -100
            sd0 = dict(sd)
            sd0["binary_classifier_region_selection.classifier.4.bias"] = torch.tensor([-100.0])
            model.load_state_dict(synth.to_reference_state_dict(sd0), strict=True)
            with torch.no_grad():
                r = model.generate(images[:1], max_length=8, num_beams=1)
            assert r == -1 and o_full.generate(sd0, images[:1], 8) == -1
            print("  S==0 sentinel: reference and oracle both return -1")

    print(f"done in {time.time() - t0:.0f}s; oracle matches reference: {all_ok}")
    return 0 if all_ok else 1


if __name__ == "__main__":
    sys.exit(main())
