"""Golden fixture for the teacher-forced pass (LM part): the REAL reference's
``LanguageModel.forward(input_ids, attention_mask, image_hidden_states, return_loss=True|False)``
(language_model.py:258-399) run in the build container on seeded synthetic weights, and the oracle's
``lm_teacher_forced`` checked against it.

    python tests/golden/make_golden_lm_loss.py
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


def make_batch(seed, S, T, ragged):
    g = torch.Generator().manual_seed(seed)
    ids = torch.randint(0, 50257, (S, T), generator=g)
    ids[:, 0] = 50256  # sentences start with BOS (train_full_model tokenisation)
    mask = torch.ones((S, T), dtype=torch.int64)
    if ragged:
        lens = torch.randint(2, T + 1, (S,), generator=g)
        lens[0] = T
        for s in range(S):
            mask[s, lens[s]:] = 0
            ids[s, lens[s]:] = 50256  # right padding with the pad token
    feats = torch.randn((S, 1024), generator=g)
    return ids, mask, feats


def main():
    model = ref_harness.reference_model()
    sd = synth.make_state_dict(0, "ragged")
    model.load_state_dict(synth.to_reference_state_dict(sd), strict=True)
    lm = model.language_model
    out = {"meta": {"torch": str(torch.__version__), "reference": "ttanida/rgrg @ /root/reference", "weights_seed": 0,
                    "profile": "ragged"}, "cases": {}}
    ok_all = True
    for name, seed, S, T, ragged in (("ragged_s6_t11", 3, 6, 11, True), ("full_s3_t7", 4, 3, 7, False),
                                     ("ragged_s20_t24", 5, 20, 24, True)):
        ids, mask, feats = make_batch(seed, S, T, ragged)
        with torch.no_grad():
            ref_logits, _presents = lm(ids.clone(), mask.clone(), feats, return_loss=False, use_cache=True)  # logits only come with use_cache
            ids_in = ids.clone()
            ref_loss = lm(ids_in, mask.clone(), feats, return_loss=True)
        o_logits = o_lm.lm_teacher_forced(sd, ids, mask, feats, return_loss=False)
        o_loss = o_lm.lm_teacher_forced(sd, ids, mask, feats, return_loss=True)
        d_logits = (o_logits - ref_logits).abs().max().item()
        d_loss = abs(o_loss.item() - ref_loss.item())
        side_effect = bool((ids_in[mask == 0] == -100).all()) and bool(torch.equal(ids_in[mask != 0], ids[mask != 0]))
        ok = d_logits <= 2e-4 and d_loss <= 1e-5
        ok_all &= ok
        print(f"{name}: loss ref {ref_loss.item():.6f} oracle {o_loss.item():.6f} |dlogits| {d_logits:.2e} "
              f"labels-written-in-place {side_effect} ok={ok}")
        # keep the fixture small: logits of 3 probe rows only
        probes = [(0, 0), (S - 1, T - 1), (S // 2, T // 2)]
        out["cases"][name] = {"input_ids": ids, "attention_mask": mask, "feats": feats, "loss": ref_loss.clone(),
                              "probes": probes, "probe_logits": torch.stack([ref_logits[s, t] for s, t in probes]),
                              "logits_absmax": ref_logits.abs().max().item(), "input_ids_after": ids_in}
    out["meta"]["oracle_matches_reference"] = bool(ok_all)
    torch.save(out, os.path.join(HERE, "lm_teacher_forced.pt"))
    print("saved lm_teacher_forced.pt; oracle matches reference:", ok_all)
    return 0 if ok_all else 1


if __name__ == "__main__":
    sys.exit(main())
