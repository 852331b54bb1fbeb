"""Golden fixtures for beam search: the REAL reference's
``LanguageModel.generate(num_beams=4)`` / ``beam_search`` loop (language_model.py:450-475,
:529-607) run in the build container on top of the restated HF-4.19.2 ``BeamSearchScorer``
(``oracle/beam_scorer.py``; the scorer itself is third-party and absent -> unpinned), and the
oracle's ``beam_generate`` checked against it.

    python tests/golden/make_golden_beam.py
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
    model = ref_harness.reference_model()
    sd = synth.make_state_dict(0, "ragged")
    model.load_state_dict(synth.to_reference_state_dict(sd), strict=True)
    g = torch.Generator().manual_seed(99)
    feats = torch.randn((5, 1024), generator=g)
    out = {"meta": {"torch": str(torch.__version__), "reference": "ttanida/rgrg @ /root/reference", "weights_seed": 0,
                    "profile": "ragged", "feat_seed": 99, "num_beams": 4,
                    "scorer": "oracle/beam_scorer.py (restated transformers 4.19.2, unpinned)"}, "cases": {}}
    ok_all = True
    for name, max_length, early in (("early_stop_len20", 20, True), ("no_early_stop_len16", 16, False), ("len40_early", 40, True)):
        with torch.no_grad():
            ref = model.language_model.generate(feats, max_length=max_length, num_beams=4, early_stopping=early)
        ora = o_lm.beam_generate(sd, feats, max_length, 4, early_stopping=early)
        ok = ref.shape == ora.shape and torch.equal(ref, ora)
        ok_all &= ok
        print(f"{name}: reference {tuple(ref.shape)} oracle {tuple(ora.shape)} match={ok}")
        out["cases"][name] = {"max_length": max_length, "early_stopping": early, "sequences": ref}
    out["meta"]["oracle_matches_reference"] = bool(ok_all)
    torch.save(out, os.path.join(HERE, "lm_beam4.pt"))
    print("saved lm_beam4.pt; oracle matches reference:", ok_all)
    # round 3: 16 beams (the widest the HIP path ranks), 3 regions, also with several returned hypotheses per region
    wide = {"meta": dict(out["meta"], num_beams=16), "cases": {}}
    ok_wide = True
    for name, max_length, early, nret in (("beams16_len14_early", 14, True, 1), ("beams16_len10_ret4", 10, False, 4)):
        with torch.no_grad():
            ref = model.language_model.generate(feats[:3], max_length=max_length, num_beams=16, early_stopping=early,
                                                num_return_sequences=nret)
        ora = o_lm.beam_generate(sd, feats[:3], max_length, 16, early_stopping=early, num_return_sequences=nret)
        ok = ref.shape == ora.shape and torch.equal(ref, ora)
        ok_wide &= ok
        print(f"{name}: reference {tuple(ref.shape)} oracle {tuple(ora.shape)} match={ok}")
        wide["cases"][name] = {"max_length": max_length, "early_stopping": early, "num_return_sequences": nret, "sequences": ref}
    wide["meta"]["oracle_matches_reference"] = bool(ok_wide)
    torch.save(wide, os.path.join(HERE, "lm_beam16.pt"))
    print("saved lm_beam16.pt; oracle matches reference:", ok_wide)
    ok_all &= ok_wide
    # round 6: more than 16 beams (the reference has no bound; the HIP path ranks them with its K-round kernels), 2 regions
    wider = {"meta": dict(out["meta"], num_beams="20 / 33"), "cases": {}}
    ok_wider = True
    for name, nb, max_length, early, nret in (("beams20_len12_early", 20, 12, True, 1), ("beams33_len8_ret5", 33, 8, False, 5)):
        with torch.no_grad():
            ref = model.language_model.generate(feats[:2], max_length=max_length, num_beams=nb, early_stopping=early,
                                                num_return_sequences=nret)
        ora = o_lm.beam_generate(sd, feats[:2], max_length, nb, early_stopping=early, num_return_sequences=nret)
        ok = ref.shape == ora.shape and torch.equal(ref, ora)
        ok_wider &= ok
        print(f"{name}: reference {tuple(ref.shape)} oracle {tuple(ora.shape)} match={ok}")
        wider["cases"][name] = {"num_beams": nb, "max_length": max_length, "early_stopping": early, "num_return_sequences": nret, "sequences": ref}
    wider["meta"]["oracle_matches_reference"] = bool(ok_wider)
    torch.save(wider, os.path.join(HERE, "lm_beam_wide.pt"))
    print("saved lm_beam_wide.pt; oracle matches reference:", ok_wider)
    ok_all &= ok_wider
    return 0 if ok_all else 1


if __name__ == "__main__":
    sys.exit(main())
