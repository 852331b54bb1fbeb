"""Golden fixture for the reduced-precision semantics of the language model: the REAL reference's
``LanguageModel.forward`` under ``torch.autocast(dtype=torch.bfloat16)`` - the way the reference's scripts wrap generation
(generate_reports_for_images.py:108, fp16 on their GPU; bf16 on the CPU of the build container is the closest the real code can
run here) - on seeded synthetic weights.  The oracle's bf16 mode (oracle/language_model.py: ``bf16=True`` rounds weights, GEMM
inputs and the K/V cache where the HIP kernels do) is not the same arithmetic as torch's autocast (which also rounds q, the
attention matmuls and every Linear output), so the comparison is statistical: two correct reduced-precision evaluations
agree to the quantisation noise.  The script prints the three pairwise distances and stores the reference's autocast logits.

    python tests/golden/make_golden_lm_autocast.py
"""
import os
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, REPO)
sys.path.insert(0, HERE)

import ref_harness  # noqa: E402
from oracle import language_model as o_lm  # noqa: E402
from rgrg_amd import synth  # noqa: E402


def main():
    torch.set_num_threads(min(8, os.cpu_count() or 8))
    model = ref_harness.reference_model()
    sd = synth.make_state_dict(0, "bench")
    model.load_state_dict(synth.to_reference_state_dict(sd), strict=True)
    lm = model.language_model.eval()
    g = torch.Generator().manual_seed(7)
    S, T = 6, 24
    ids = torch.randint(0, 50257, (S, T), generator=g)
    ids[:, 0] = 50256
    mask = torch.ones((S, T), dtype=torch.int64)
    feats = torch.randn((S, 1024), generator=g)
    with torch.no_grad():
        ref32, _ = lm(ids.clone(), mask.clone(), feats, return_loss=False, use_cache=True)
        with torch.autocast("cpu", dtype=torch.bfloat16):
            ref16, _ = lm(ids.clone(), mask.clone(), feats, return_loss=False, use_cache=True)
    ref16 = ref16.float()
    o16, _ = o_lm.lm_forward(sd, ids, mask, feats, None, torch.arange(T)[None, :], bf16=True)
    rng = ref32.abs().max().item()

    def dist(a, b):
        return (a - b).abs().max().item() / rng, (a.argmax(-1) == b.argmax(-1)).float().mean().item()
    d_ref = dist(ref16, ref32)
    d_o16 = dist(o16, ref16)
    d_o32 = dist(o16, ref32)
    print(f"logit range {rng:.3f}")
    print(f"reference autocast(bf16) vs reference fp32 : max |d| / range {d_ref[0]:.4f}, arg-max agreement {d_ref[1]:.4f}")
    print(f"bf16 oracle vs reference autocast(bf16)   : max |d| / range {d_o16[0]:.4f}, arg-max agreement {d_o16[1]:.4f}")
    print(f"bf16 oracle vs reference fp32             : max |d| / range {d_o32[0]:.4f}, arg-max agreement {d_o32[1]:.4f}")
    out = {"meta": {"torch": str(torch.__version__), "reference": "ttanida/rgrg @ /root/reference", "weights_seed": 0,
                    "profile": "bench", "autocast": "cpu, bfloat16", "logit_range": rng,
                    "ref16_vs_ref32": d_ref, "oracle16_vs_ref16": d_o16, "oracle16_vs_ref32": d_o32},
           "input_ids": ids, "attention_mask": mask, "feats": feats,
           "ref16_logits_last": ref16[:, -1].clone(), "ref16_argmax": ref16.argmax(-1), "ref32_argmax": ref32.argmax(-1)}
    torch.save(out, os.path.join(HERE, "lm_autocast_bf16.pt"))
    print("saved lm_autocast_bf16.pt")
    return 0


if __name__ == "__main__":
    sys.exit(main())
