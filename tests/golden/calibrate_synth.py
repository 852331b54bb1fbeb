"""Measure the per-class mean/std of the raw ``cls_score`` logits of the seeded
synthetic detector on the calibration image (CPU oracle) and store them for
``rgrg_amd.synth._calibrate_cls_score``.  Run once in the build container:

    python tests/golden/calibrate_synth.py [seed]
"""
import os
import sys

import torch

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
from oracle import detector  # noqa: E402
from rgrg_amd import synth  # noqa: E402


def main(seed: int = 0):
    path = os.path.join(REPO, "rgrg_amd", "data", f"synth_calib_seed{seed}.pt")
    if os.path.exists(path):
        os.remove(path)
    sd = synth.make_state_dict(seed, "bench")  # un-calibrated (file removed)
    out = detector.object_detector_forward(sd, synth.make_images(1, 1234), return_intermediates=True)
    cl = out["_class_logits"]
    torch.save({"mean": cl.mean(0), "std": cl.std(0)}, path)
    print("wrote", path, cl.shape)


if __name__ == "__main__":
    main(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
