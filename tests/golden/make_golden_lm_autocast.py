"""Golden fixtures for the reduced-precision semantics of the language model: the REAL reference's
``LanguageModel.forward`` under ``torch.autocast`` - the way the reference's scripts wrap generation
(generate_reports_for_images.py:108: float16 on their GPU) - run on the CPU of the build container for BOTH 16-bit types:
``lm_autocast_fp16.pt`` (float16: the reference's own mode, round 4) and ``lm_autocast_bf16.pt`` (bfloat16) - on seeded
synthetic weights.  The oracle's 16-bit modes (oracle/language_model.py: ``bf16=1`` bfloat16 / ``bf16=2`` float16 round
weights, GEMM inputs and the K/V cache where the HIP kernels do) are not the same arithmetic as torch's autocast (which also
rounds q, the attention matmuls and every Linear output), so the comparison is statistical: two correct reduced-precision
evaluations agree to the quantisation noise.  The script prints the three pairwise distances and stores the reference's
autocast logits.

    python tests/golden/make_golden_lm_autocast.py [bf16|fp16]    (default: both)
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


def make(kind):
    dtype, mode = (torch.float16, 2) if kind == "fp16" else (torch.bfloat16, 1)
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
        with torch.autocast("cpu", dtype=dtype):
            ref16, _ = lm(ids.clone(), mask.clone(), feats, return_loss=False, use_cache=True)
    ref16 = ref16.float()
    o16, _ = o_lm.lm_forward(sd, ids, mask, feats, None, torch.arange(T)[None, :], bf16=mode)
    rng = ref32.abs().max().item()

    def dist(a, b):
        return (a - b).abs().max().item() / rng, (a.argmax(-1) == b.argmax(-1)).float().mean().item()
    d_ref = dist(ref16, ref32)
    d_o16 = dist(o16, ref16)
    d_o32 = dist(o16, ref32)
    print(f"[{kind}] logit range {rng:.3f}")
    print(f"reference autocast({kind}) vs reference fp32 : max |d| / range {d_ref[0]:.4f}, arg-max agreement {d_ref[1]:.4f}")
    print(f"{kind} oracle vs reference autocast({kind})   : max |d| / range {d_o16[0]:.4f}, arg-max agreement {d_o16[1]:.4f}")
    print(f"{kind} oracle vs reference fp32             : max |d| / range {d_o32[0]:.4f}, arg-max agreement {d_o32[1]:.4f}")
    out = {"meta": {"torch": str(torch.__version__), "reference": "ttanida/rgrg @ /root/reference", "weights_seed": 0,
                    "profile": "bench", "autocast": "cpu, " + ("float16" if kind == "fp16" else "bfloat16"), "logit_range": rng,
                    "ref16_vs_ref32": d_ref, "oracle16_vs_ref16": d_o16, "oracle16_vs_ref32": d_o32},
           "input_ids": ids, "attention_mask": mask, "feats": feats,
           "ref16_logits_last": ref16[:, -1].clone(), "ref16_argmax": ref16.argmax(-1), "ref32_argmax": ref32.argmax(-1)}
    torch.save(out, os.path.join(HERE, f"lm_autocast_{kind}.pt"))
    print(f"saved lm_autocast_{kind}.pt")


def main():
    for kind in (sys.argv[1:] or ["bf16", "fp16"]):
        make(kind)
    return 0


if __name__ == "__main__":
    sys.exit(main())
