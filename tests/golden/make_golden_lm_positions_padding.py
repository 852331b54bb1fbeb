"""Golden fixture for two argument forms of the REAL reference's ``LanguageModel.forward`` that round 5 still refused
(VERDICT r05 "missing" 2 and 3), run in the build container on seeded synthetic weights, with the oracle checked against it:

  * the teacher-forced pass (``use_cache=False``) with ARBITRARY ``position_ids`` - the reference embeds whatever it is given,
    through the token table (language_model.py:293-307): a [S,T] table of scattered ids and a broadcast [1,T] row;
  * the incremental form (``use_cache=True``) with PADDING: a left-padded prompt (attention_mask zeros in front, as a batched
    generation call would pass them), positions counted from the first real token, then a single-token call with the
    returned presents and the grown mask - the reference adds (1 - mask) * -1e4 to every query's score of a masked key
    (:316-334).

    python tests/golden/make_golden_lm_positions_padding.py
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

LAYERS = (0, 23)


def main():
    model = ref_harness.reference_model()
    sd = synth.make_state_dict(0, "ragged")
    model.load_state_dict(synth.to_reference_state_dict(sd), strict=True)
    lm = model.language_model
    out = {"meta": {"torch": str(torch.__version__), "reference": "ttanida/rgrg @ /root/reference", "weights_seed": 0, "profile": "ragged"},
           "teacher_forced": {}, "cached": {}}
    ok_all = True
    g = torch.Generator().manual_seed(61)
    # ---- teacher-forced pass with position_ids
    for name, S, T, per_row in (("table_s5_t9", 5, 9, True), ("row_s4_t12", 4, 12, False)):
        ids = torch.randint(0, 50257, (S, T), generator=g)
        ids[:, 0] = 50256
        mask = torch.ones((S, T), dtype=torch.int64)
        lens = torch.randint(3, T + 1, (S,), generator=g)
        lens[0] = T
        for s in range(S):
            mask[s, lens[s]:] = 0
        feats = torch.randn((S, 1024), generator=g)
        pos = torch.randint(0, 3000, (S if per_row else 1, T), generator=g)
        with torch.no_grad():
            ref_logits, _ = lm(ids.clone(), mask.clone(), feats, return_loss=False, position_ids=pos, use_cache=True)
            ref_loss = lm(ids.clone(), mask.clone(), feats, return_loss=True, position_ids=pos)
            ref_loss_default = lm(ids.clone(), mask.clone(), feats, return_loss=True)
        o_logits = o_lm.lm_teacher_forced(sd, ids, mask, feats, return_loss=False, position_ids=pos)
        o_loss = o_lm.lm_teacher_forced(sd, ids, mask, feats, return_loss=True, position_ids=pos)
        d_logits, d_loss = (o_logits - ref_logits).abs().max().item(), abs(o_loss.item() - ref_loss.item())
        ok = d_logits <= 2e-4 and d_loss <= 1e-5
        ok_all &= ok
        print(f"teacher forced {name}: loss {ref_loss.item():.6f} (default positions: {ref_loss_default.item():.6f}) oracle {o_loss.item():.6f} "
              f"|dlogits| {d_logits:.2e} ok={ok}")
        probes = [(0, 0), (S - 1, int(lens[S - 1]) - 1), (S // 2, 1)]
        out["teacher_forced"][name] = {"input_ids": ids, "attention_mask": mask, "feats": feats, "position_ids": pos, "loss": ref_loss.clone(),
                                       "loss_default_positions": ref_loss_default.clone(), "probes": probes,
                                       "probe_logits": torch.stack([ref_logits[s, t] for s, t in probes]),
                                       "logits_absmax": ref_logits.abs().max().item()}
    # ---- incremental form with a left-padded prompt
    S, T = 4, 6
    pads = torch.tensor([0, 2, 3, 1])
    prompt = torch.randint(0, 50000, (S, T), generator=g)
    mask = torch.ones((S, T), dtype=torch.int64)
    for s in range(S):
        mask[s, :pads[s]] = 0
        prompt[s, :pads[s]] = 50256
    pos = (torch.cumsum(mask, 1) - 1).clamp(min=0)            # positions counted from the first real token
    feats = torch.randn((S, 1024), generator=g)
    nxt = torch.randint(0, 50000, (S, 1), generator=g)
    mask2 = torch.cat([mask, torch.ones((S, 1), dtype=torch.int64)], dim=1)
    pos2 = pos[:, -1:] + 1
    with torch.no_grad():
        r1, rp = lm(prompt, mask, feats, return_loss=False, position_ids=pos, use_cache=True)
        r2, rp2 = lm(nxt, mask2, feats, return_loss=False, past_key_values=rp, position_ids=pos2, use_cache=True)
        r1_nomask, _ = lm(prompt, torch.ones_like(mask), feats, return_loss=False, position_ids=pos, use_cache=True)
    o1, op = o_lm.lm_forward(sd, prompt, mask, feats, None, pos)
    o2, op2 = o_lm.lm_forward(sd, nxt, mask2, feats, op, pos2)
    d1, d2 = (o1 - r1).abs().max().item(), (o2 - r2).abs().max().item()
    dk = max((op2[l][j] - rp2[l][j]).abs().max().item() for l in LAYERS for j in (0, 1))
    ok = d1 <= 2e-4 and d2 <= 2e-4 and dk <= 1e-5
    ok_all &= ok
    print(f"cached, left-padded prompt: |dlogits| {d1:.2e} / {d2:.2e}, presents |d| {dk:.2e}; the mask matters: "
          f"|logits - unmasked| {(r1 - r1_nomask)[:, -1].abs().max().item():.3f} ok={ok}")
    out["cached"] = {"feats": feats, "prompt": prompt, "mask": mask, "position_ids": pos, "next": nxt, "mask2": mask2, "position_ids2": pos2,
                     "logits_prompt_last": r1[:, -1].clone(), "logits_prompt_probe": r1[:, :, ::97].clone(), "logits_next": r2[:, -1].clone(),
                     "presents": {l: (rp2[l][0].clone(), rp2[l][1].clone()) for l in LAYERS},
                     "unmasked_last_logit_gap": (r1 - r1_nomask)[:, -1].abs().max().item()}
    out["meta"]["oracle_matches_reference"] = bool(ok_all)
    torch.save(out, os.path.join(HERE, "lm_positions_padding.pt"))
    print("saved lm_positions_padding.pt; oracle matches reference:", ok_all)
    return 0 if ok_all else 1


if __name__ == "__main__":
    sys.exit(main())
