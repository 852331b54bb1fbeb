"""Golden fixture for the incremental form of the language model: the REAL reference's
``LanguageModel.forward(input_ids, attention_mask, image_hidden_states, return_loss=False, past_key_values, position_ids,
use_cache=True)`` (language_model.py:258-366, :396-399) called the way its own greedy loop calls it
(prepare_inputs_for_generation, :498-520) - a 3-token prompt, then two single-token calls fed with the returned presents -
run in the build container on seeded synthetic weights; the oracle's ``lm_forward`` is checked against it.

    python tests/golden/make_golden_lm_cached.py
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

LAYERS = (0, 11, 23)   # presents kept in the fixture


def main():
    model = ref_harness.reference_model()
    sd = synth.make_state_dict(0, "ragged")
    model.load_state_dict(synth.to_reference_state_dict(sd), strict=True)
    lm = model.language_model
    g = torch.Generator().manual_seed(21)
    S = 3
    feats = torch.randn((S, 1024), generator=g)
    prompt = torch.randint(0, 50000, (S, 3), generator=g)
    steps = [torch.randint(0, 50000, (S, 1), generator=g) for _ in range(2)]
    calls, ok_all = [], True
    with torch.no_grad():
        r_logits, r_past = lm(prompt, torch.ones((S, 3), dtype=torch.int64), feats, return_loss=False,
                              position_ids=torch.arange(3)[None, :], use_cache=True)
        o_logits, o_past = o_lm.lm_forward(sd, prompt, torch.ones((S, 3), dtype=torch.int64), feats, None, torch.arange(3)[None, :])
        d = (o_logits - r_logits).abs().max().item()
        ok_all &= d <= 2e-4
        print(f"prompt: |dlogits| {d:.2e}")
        calls.append({"input_ids": prompt, "position_ids": torch.arange(3)[None, :], "logits_last": r_logits[:, -1].clone(),
                      "logits_first_probe": r_logits[:, 0, ::97].clone()})
        ntok = 3
        for tok in steps:
            am = torch.ones((S, ntok + 1), dtype=torch.int64)
            pos = torch.full((S, 1), ntok)
            r_logits, r_past = lm(tok, am, feats, return_loss=False, past_key_values=r_past, position_ids=pos, use_cache=True)
            o_logits, o_past = o_lm.lm_forward(sd, tok, am, feats, o_past, pos)
            d = (o_logits - r_logits).abs().max().item()
            ok_all &= d <= 2e-4
            print(f"cached step at {ntok} tokens: |dlogits| {d:.2e}")
            calls.append({"input_ids": tok, "position_ids": pos, "logits_last": r_logits[:, -1].clone()})
            ntok += 1
    dk = max((o_past[l][j] - r_past[l][j]).abs().max().item() for l in LAYERS for j in (0, 1))
    ok_all &= dk <= 1e-5
    print(f"presents (layers {LAYERS}): shape {tuple(r_past[0][0].shape)} |d| {dk:.2e}")
    out = {"meta": {"torch": str(torch.__version__), "reference": "ttanida/rgrg @ /root/reference", "weights_seed": 0,
                    "profile": "ragged", "oracle_matches_reference": bool(ok_all)},
           "feats": feats, "calls": calls,
           "presents": {l: (r_past[l][0].clone(), r_past[l][1].clone()) for l in LAYERS}}
    torch.save(out, os.path.join(HERE, "lm_cached_steps.pt"))
    print("saved lm_cached_steps.pt; oracle matches reference:", ok_all)
    return 0 if ok_all else 1


if __name__ == "__main__":
    sys.exit(main())
