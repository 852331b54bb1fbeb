"""Golden fixture for the decoder's training pass: gradients of the REAL reference's
``LanguageModel.forward(return_loss=True)`` (language_model.py:258-399) obtained with ``loss.backward()``, modules in
eval mode (dropout off - the HIP pass is deterministic) and gradients enabled; the oracle's autograd through the
restated forward is checked against them.  Only slices of the 52 M gradient values are stored.

    python tests/golden/make_golden_lm_grads.py
"""
import os
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, REPO)
sys.path.insert(0, HERE)

import ref_harness  # noqa: E402
from make_golden_lm_loss import make_batch  # noqa: E402
from oracle import language_model as o_lm  # noqa: E402
from rgrg_amd import synth  # noqa: E402

PROBES = ["language_model.gpt_with_lm_head.transformer.h.0.attn.uk.weight",
          "language_model.gpt_with_lm_head.transformer.h.0.attn.uv.bias",
          "language_model.gpt_with_lm_head.transformer.h.11.attn.uv.weight",
          "language_model.gpt_with_lm_head.transformer.h.23.attn.uk.bias",
          "language_model.gpt_with_lm_head.transformer.h.23.attn.uv.weight",
          "language_model.feature_space_transformation_nn.0.weight",
          "language_model.feature_space_transformation_nn.0.bias",
          "language_model.feature_space_transformation_nn.2.weight",
          "language_model.feature_space_transformation_nn.2.bias"]


def probe(t):
    return t[::37, ::41].clone() if t.dim() == 2 else t[::7].clone()


def one_case(lm, sd, batch_args, fname):
    ids, mask, feats = make_batch(*batch_args)
    trainable = {k: p for k, p in lm.named_parameters() if p.requires_grad}
    assert sorted("language_model." + k for k in trainable) == sorted(o_lm.trainable_keys()), "trainable set differs"
    for p in trainable.values():
        p.grad = None
    loss = lm(ids.clone(), mask.clone(), feats, return_loss=True)
    loss.backward()
    o_loss, o_grads = o_lm.lm_loss_and_grads(sd, ids, mask, feats)
    ok = abs(o_loss.item() - loss.item()) <= 1e-5
    worst = 0.0
    norms = {}
    for k, p in trainable.items():
        g, og = p.grad, o_grads["language_model." + k]
        rel = (g - og).abs().max().item() / (g.abs().max().item() + 1e-30)
        worst = max(worst, rel)
        norms["language_model." + k] = g.norm().item()
    ok &= worst <= 1e-4
    print(f"{fname}: loss {loss.item():.6f} oracle {o_loss.item():.6f}; worst relative gradient difference oracle vs reference {worst:.2e}")
    out = {"meta": {"torch": str(torch.__version__), "reference": "ttanida/rgrg @ /root/reference", "weights_seed": 0,
                    "profile": "ragged", "batch": f"make_golden_lm_loss.make_batch{batch_args}", "dropout": "off (eval mode)",
                    "oracle_matches_reference": bool(ok)},
           "input_ids": ids, "attention_mask": mask, "feats": feats, "loss": loss.detach().clone(),
           "grad_norms": norms, "probes": {k: probe(trainable[k[len("language_model."):]].grad) for k in PROBES}}
    torch.save(out, os.path.join(HERE, fname))
    print(f"saved {fname}; oracle matches reference:", ok)
    return ok


def main():
    model = ref_harness.reference_model()
    sd = synth.make_state_dict(0, "ragged")
    model.load_state_dict(synth.to_reference_state_dict(sd), strict=True)
    lm = model.language_model
    lm.eval()
    ok = True
    if "--long-only" not in sys.argv:
        ok &= one_case(lm, sd, (3, 6, 11, True), "lm_grads.pt")
    # round 3: a long sequence (more than 256 keys: the HIP pass streams its attention forward and backward there)
    ok &= one_case(lm, sd, (8, 2, 300, True), "lm_grads_t300.pt")
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main())
