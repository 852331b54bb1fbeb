"""CPU oracle vs the golden fixtures produced by the REAL reference
(tests/golden/make_golden.py).  Bit-exact: both are torch-CPU fp32 running the same
operation sequence, so any drift of the restatement shows up here."""
import pytest
import torch

from conftest import load_golden
from oracle import detector as o_det
from oracle import full_model as o_full
from oracle import language_model as o_lm
from rgrg_amd import synth


def test_fixtures_were_generated_from_the_reference():
    for name in ("bench_b1_len128.pt", "ragged_b2_len24.pt", "lm_only_len12.pt", "lm_only_allfinish.pt"):
        fx = load_golden(name)
        assert fx["meta"]["oracle_matches_reference"] is True
        assert fx["meta"]["reference"].startswith("ttanida/rgrg")


def test_full_generate_ragged_matches_reference(sd_ragged):
    fx = load_golden("ragged_b2_len24.pt")
    images = torch.cat([synth.make_images(1, s) for s in fx["meta"]["image_seeds"]], 0)
    out = o_det.object_detector_forward(sd_ragged, images, return_intermediates=True)
    d = fx["detector"]
    for a, b in zip(out["_proposals"], d["proposals"]):
        assert torch.equal(a, b)
    assert torch.equal(out["class_detected"], d["class_detected"])
    assert torch.equal(out["top_region_features"], d["top_region_features"])
    assert torch.equal(out["detections"]["top_region_boxes"], d["top_region_boxes"])
    assert torch.equal(out["detections"]["top_scores"], d["top_scores"])
    # ragged on purpose: some regions undetected, some unselected
    assert 0 < int(d["class_detected"].sum()) < d["class_detected"].numel()
    sel, feats, _ = o_full.region_selection(sd_ragged, out["top_region_features"], out["class_detected"])
    g = fx["generate"]
    assert torch.equal(sel, g["selected_regions"]) and 0 < int(sel.sum()) < int(d["class_detected"].sum())
    ids = o_lm.greedy_generate(sd_ragged, feats, fx["meta"]["max_length"])
    assert torch.equal(ids, g["output_ids"])
    finished = (ids[:, 1:] == 50256).any(1)
    assert finished.any() and not finished.all()  # rows finish at different steps, PAD afterwards
    for row in ids[finished]:
        first = int((row[1:] == 50256).nonzero()[0]) + 1
        assert (row[first:] == 50256).all()


def test_lm_only_and_early_exit(sd_ragged):
    g = torch.Generator().manual_seed(99)
    feats = torch.randn((5, 1024), generator=g)
    fx = load_golden("lm_only_len12.pt")
    ids, logits = o_lm.greedy_generate(sd_ragged, feats, 12, return_logits=True)
    assert torch.equal(ids, fx["output_ids"]) and ids.shape == (5, 12)
    assert torch.equal(logits[:, 0, ::101], fx["step0_logits_sample"])
    assert torch.equal(logits[:, 0].argmax(-1), fx["step0_argmax"])
    fin = load_golden("lm_only_allfinish.pt")
    ids = o_lm.greedy_generate(sd_ragged, feats, 40)
    assert torch.equal(ids, fin["output_ids"]) and ids.shape[1] < 40  # all rows emitted EOS -> early exit


def test_empty_selection_returns_minus_one(sd_ragged):
    sd0 = dict(sd_ragged)
    sd0["binary_classifier_region_selection.classifier.4.bias"] = torch.tensor([-100.0])
    feats = torch.zeros((1, 29, 1024))
    sel, f, _ = o_full.region_selection(sd0, feats, torch.ones((1, 29), dtype=torch.bool))
    assert int(sel.sum()) == 0 and f.shape == (0, 1024)


def test_beam_search_oracle_matches_reference_loop(sd_ragged):
    """The reference's own beam_search loop (run over the restated scorer) is reproduced bit-exactly."""
    fx = load_golden("lm_beam4.pt")
    assert fx["meta"]["oracle_matches_reference"] is True
    g = torch.Generator().manual_seed(99)
    feats = torch.randn((5, 1024), generator=g)
    c = fx["cases"]["early_stop_len20"]
    seq = o_lm.beam_generate(sd_ragged, feats, c["max_length"], 4, early_stopping=c["early_stopping"])
    assert torch.equal(seq, c["sequences"])


def test_beam_search_oracle_matches_reference_loop_at_16_beams(sd_ragged):
    """lm_beam16.pt (round 3: the HIP path ranks up to 16 beams): the reference's loop with num_beams=16, with early stopping
    and with 4 returned hypotheses per region."""
    fx = load_golden("lm_beam16.pt")
    assert fx["meta"]["oracle_matches_reference"] is True and fx["meta"]["num_beams"] == 16
    g = torch.Generator().manual_seed(99)
    feats = torch.randn((5, 1024), generator=g)[:3]
    for c in fx["cases"].values():
        seq = o_lm.beam_generate(sd_ragged, feats, c["max_length"], 16, early_stopping=c["early_stopping"],
                                 num_return_sequences=c["num_return_sequences"])
        assert torch.equal(seq, c["sequences"])


def test_beam_search_oracle_matches_reference_loop_beyond_16_beams(sd_ragged):
    """lm_beam_wide.pt (round 6: the reference's loop has no bound on num_beams, and neither has the HIP path now): 20 beams with
    early stopping, 33 beams with 5 returned hypotheses per region."""
    fx = load_golden("lm_beam_wide.pt")
    assert fx["meta"]["oracle_matches_reference"] is True
    g = torch.Generator().manual_seed(99)
    feats = torch.randn((5, 1024), generator=g)[:2]
    c = fx["cases"]["beams20_len12_early"]   # (the 33-beam case is the GPU test's; one case keeps the CPU suite short)
    seq = o_lm.beam_generate(sd_ragged, feats, c["max_length"], c["num_beams"], early_stopping=c["early_stopping"],
                             num_return_sequences=c["num_return_sequences"])
    assert torch.equal(seq, c["sequences"])


def test_beam_scorer_hand_case():
    """Known-answer test of the restated BeamHypotheses (third-party semantics, unpinned otherwise)."""
    from oracle.beam_scorer import BeamHypotheses
    h = BeamHypotheses(num_beams=2, length_penalty=1.0, early_stopping=False)
    h.add(torch.tensor([1, 2, 3, 4]), -4.0)   # score -1.0
    assert not h.is_done(-0.1, 4)
    h.add(torch.tensor([1, 2]), -3.0)         # score -1.5 -> worst
    assert h.worst_score == -1.5 and len(h) == 2
    h.add(torch.tensor([1, 2, 3]), -1.5)      # score -0.5 replaces the worst (-1.5); new worst -1.0
    assert len(h) == 2 and h.worst_score == -1.0
    h.add(torch.tensor([1]), -5.0)            # worse than worst: ignored
    assert len(h) == 2
    assert h.is_done(-8.0, 4) and not h.is_done(-2.0, 4)   # -8/4 = -2 <= -1.0 done ; -2/4 = -0.5 > -1.0 not done


# ------------------------------------------------------------------------- forward(), eval mode
def test_teacher_forced_lm_oracle_matches_reference_forward(sd_ragged):
    """LanguageModel.forward(return_loss=True) of the REAL reference (fixture) vs the restatement: loss to 1e-5
    absolute (loss ~ 11), probe logits to 2e-4; padded label positions are ignored."""
    fx = load_golden("lm_teacher_forced.pt")
    assert fx["meta"]["oracle_matches_reference"] is True
    for name in ("ragged_s6_t11", "full_s3_t7"):
        c = fx["cases"][name]
        loss = o_lm.lm_teacher_forced(sd_ragged, c["input_ids"], c["attention_mask"], c["feats"], return_loss=True)
        assert abs(loss.item() - c["loss"].item()) <= 1e-5, name
        logits = o_lm.lm_teacher_forced(sd_ragged, c["input_ids"], c["attention_mask"], c["feats"], return_loss=False)
        got = torch.stack([logits[s, t] for s, t in c["probes"]])
        assert (got - c["probe_logits"]).abs().max().item() <= 2e-4, name
        # the reference overwrote the padded ids with -100 in place (labels IS input_ids, language_model.py:371-374)
        am = c["attention_mask"].bool()
        assert (c["input_ids_after"][~am] == -100).all() and torch.equal(c["input_ids_after"][am], c["input_ids"][am])
    # a fully padded label row set contributes nothing: masking every label but one leaves that row's NLL
    c = fx["cases"]["full_s3_t7"]
    m1 = torch.zeros_like(c["attention_mask"])
    m1[:, 0] = 1
    m1[1, 1] = 1
    one = o_lm.lm_teacher_forced(sd_ragged, c["input_ids"], m1, c["feats"], return_loss=True)
    assert torch.isfinite(one) and one.item() > 0


def test_eval_forward_oracle_matches_reference(sd_ragged):
    """ReportGenerationModel.forward (eval, image_targets=None) of the REAL reference vs the restatement."""
    fx = load_golden("forward_eval_b2.pt")
    assert fx["meta"]["oracle_matches_reference"] is True
    images = torch.cat([synth.make_images(1, s) for s in fx["meta"]["image_seeds"]], 0)
    i, e = fx["inputs"], fx["expected"]
    out = o_full.forward_eval(sd_ragged, images, i["input_ids"].clone(), i["attention_mask"], i["region_has_sentence"],
                              i["region_is_abnormal"])
    assert out[0] == {}
    for got, key in ((out[1], "classifier_loss_region_selection"), (out[2], "classifier_loss_region_abnormal"),
                     (out[3], "language_model_loss")):
        assert abs(got.item() - e[key].item()) <= 1e-5, key
    assert torch.equal(out[5], e["class_detected"]) and torch.equal(out[6], e["selected_regions"])
    assert torch.equal(out[7], e["predicted_abnormal_regions"])
    assert torch.equal(out[4]["top_region_boxes"], e["top_region_boxes"])


def test_eval_forward_with_image_targets_oracle_matches_reference(sd_bench):
    """ReportGenerationModel.forward(images, image_targets, ...) in eval mode - the validation loop's call
    (evaluate_model.py:413) - of the REAL reference vs the restatement, with the samplers' draws injected
    (tests/golden/make_golden_forward_targets.py)."""
    fx = load_golden("forward_eval_targets_b2.pt")
    assert fx["meta"]["oracle_matches_reference"] is True
    images = torch.cat([synth.make_images(1, s) for s in fx["meta"]["image_seeds"]], 0)
    i, e = fx["inputs"], fx["expected"]
    g = torch.Generator().manual_seed(fx["meta"]["perm_seed"])
    out = o_full.forward_eval(sd_bench, images, i["input_ids"].clone(), i["attention_mask"], i["region_has_sentence"],
                              i["region_is_abnormal"], image_targets=i["targets"], perm_fn=lambda n, tag: torch.randperm(n, generator=g))
    assert list(out[0]) == list(e["obj_detector_loss_dict"]) == ["loss_classifier", "loss_box_reg", "loss_objectness", "loss_rpn_box_reg"]
    for k, v in e["obj_detector_loss_dict"].items():
        assert abs(out[0][k].item() - v.item()) <= 1e-6, k
    for got, key in ((out[1], "classifier_loss_region_selection"), (out[2], "classifier_loss_region_abnormal"),
                     (out[3], "language_model_loss")):
        assert abs(got.item() - e[key].item()) <= 1e-5, key
    assert torch.equal(out[5], e["class_detected"]) and torch.equal(out[6], e["selected_regions"])
    assert torch.equal(out[7], e["predicted_abnormal_regions"]) and torch.equal(out[4]["top_region_boxes"], e["top_region_boxes"])


def test_incremental_forward_oracle_matches_reference_cached_steps(sd_ragged):
    """lm_cached_steps.pt: the REAL reference's forward(use_cache=True) on a prompt and two single-token calls fed with its
    own presents (tests/golden/make_golden_lm_cached.py) - the oracle's lm_forward reproduces logits and presents exactly."""
    fx = load_golden("lm_cached_steps.pt")
    assert fx["meta"]["oracle_matches_reference"] is True
    feats, past, ntok = fx["feats"], None, 0
    for c in fx["calls"]:
        T = c["input_ids"].shape[1]
        am = torch.ones((3, ntok + T), dtype=torch.int64)
        logits, past = o_lm.lm_forward(sd_ragged, c["input_ids"], am, feats, past, c["position_ids"])
        assert torch.equal(logits[:, -1], c["logits_last"])
        if "logits_first_probe" in c:
            assert torch.equal(logits[:, 0, ::97], c["logits_first_probe"])
        ntok += T
    for l, (k, v) in fx["presents"].items():
        assert torch.equal(past[l][0], k) and torch.equal(past[l][1], v)


def test_oracle_matches_reference_with_position_ids_and_with_padding(sd_ragged):
    """lm_positions_padding.pt (tests/golden/make_golden_lm_positions_padding.py, round 6): the REAL reference's teacher-forced
    pass with arbitrary position_ids (language_model.py:293-307) and its incremental form over a left-padded prompt
    (additive -1e4 per masked key, :316-334) - the oracle reproduces both exactly."""
    fx = load_golden("lm_positions_padding.pt")
    assert fx["meta"]["oracle_matches_reference"] is True
    for name, c in fx["teacher_forced"].items():
        loss = o_lm.lm_teacher_forced(sd_ragged, c["input_ids"], c["attention_mask"], c["feats"], return_loss=True, position_ids=c["position_ids"])
        assert torch.equal(loss, c["loss"]), name
        assert abs(float(c["loss"]) - float(c["loss_default_positions"])) > 1e-3, name   # the positions matter
        logits = o_lm.lm_teacher_forced(sd_ragged, c["input_ids"], c["attention_mask"], c["feats"], return_loss=False, position_ids=c["position_ids"])
        assert torch.equal(torch.stack([logits[s, t] for s, t in c["probes"]]), c["probe_logits"]), name
    c = fx["cached"]
    l1, past = o_lm.lm_forward(sd_ragged, c["prompt"], c["mask"], c["feats"], None, c["position_ids"])
    l2, past = o_lm.lm_forward(sd_ragged, c["next"], c["mask2"], c["feats"], past, c["position_ids2"])
    assert torch.equal(l1[:, -1], c["logits_prompt_last"]) and torch.equal(l1[:, :, ::97], c["logits_prompt_probe"])
    assert torch.equal(l2[:, -1], c["logits_next"])
    for l, (k, v) in c["presents"].items():
        assert torch.equal(past[l][0], k) and torch.equal(past[l][1], v)
    assert c["unmasked_last_logit_gap"] > 0.1   # the mask matters


def test_bf16_oracle_is_within_quantisation_noise_of_the_reference_under_autocast(sd_bench):
    """lm_autocast_bf16.pt: the REAL reference's LanguageModel.forward under torch.autocast(bfloat16) (CPU: the closest the
    real code runs here to the fp16 autocast its scripts use).  The bf16 mode of the oracle - the parity target of the HIP
    bf16 path - is a different set of rounding points, so the check is statistical: it is as close to the reference's
    autocast run as that run is to the reference's own fp32 run (both ~1 % of the logit range, arg-max agreement >= 95 %)."""
    fx = load_golden("lm_autocast_bf16.pt")
    ids, mask, feats = fx["input_ids"], fx["attention_mask"], fx["feats"]
    T = ids.shape[1]
    o16, _ = o_lm.lm_forward(sd_bench, ids, mask, feats, None, torch.arange(T)[None, :], bf16=True)
    rng = fx["meta"]["logit_range"]
    d = (o16[:, -1] - fx["ref16_logits_last"]).abs().max().item() / rng
    agree16 = (o16.argmax(-1) == fx["ref16_argmax"]).float().mean().item()
    agree32 = (o16.argmax(-1) == fx["ref32_argmax"]).float().mean().item()
    ref_self = (fx["ref16_argmax"] == fx["ref32_argmax"]).float().mean().item()
    assert d <= 2e-2, d
    assert agree16 >= 0.95 and agree32 >= 0.95 and ref_self >= 0.95, (agree16, agree32, ref_self)
    assert abs(fx["meta"]["oracle16_vs_ref16"][0] - fx["meta"]["ref16_vs_ref32"][0]) <= 1e-2   # same noise level


def test_fp16_oracle_is_within_quantisation_noise_of_the_reference_under_fp16_autocast(sd_bench):
    """lm_autocast_fp16.pt (round 4): the REAL reference's LanguageModel.forward under torch.autocast("cpu", float16) - the
    dtype its own scripts use (generate_reports_for_images.py:108).  The float16 mode of the oracle (bf16=2), parity target of
    the HIP fp16 path, sits within the reference's own fp16-vs-fp32 distance: ~0.15 % of the logit range (8x below bf16)."""
    fx = load_golden("lm_autocast_fp16.pt")
    ids, mask, feats = fx["input_ids"], fx["attention_mask"], fx["feats"]
    T = ids.shape[1]
    o16, _ = o_lm.lm_forward(sd_bench, ids, mask, feats, None, torch.arange(T)[None, :], bf16=2)
    rng = fx["meta"]["logit_range"]
    d = (o16[:, -1] - fx["ref16_logits_last"]).abs().max().item() / rng
    agree16 = (o16.argmax(-1) == fx["ref16_argmax"]).float().mean().item()
    agree32 = (o16.argmax(-1) == fx["ref32_argmax"]).float().mean().item()
    assert fx["meta"]["autocast"] == "cpu, float16" and fx["meta"]["ref16_vs_ref32"][0] <= 3e-3
    assert d <= 3e-3, d
    assert agree16 >= 0.98 and agree32 >= 0.98, (agree16, agree32)
    # and it is NOT the bf16 arithmetic: the bf16 oracle is ~5x further from this fixture
    b16, _ = o_lm.lm_forward(sd_bench, ids, mask, feats, None, torch.arange(T)[None, :], bf16=1)
    assert (b16[:, -1] - fx["ref16_logits_last"]).abs().max().item() / rng >= 3 * d


@pytest.mark.parametrize("name", ["lm_grads.pt", "lm_grads_t300.pt"])
def test_lm_gradients_oracle_matches_reference_autograd(sd_ragged, name):
    """The REAL reference's loss.backward() through the language model (tests/golden/make_golden_lm_grads.py; T = 11 and,
    round 3, T = 300): the oracle's autograd gives the same loss, gradient norms and probe slices."""
    fx = load_golden(name)
    assert fx["meta"]["oracle_matches_reference"] is True
    loss, grads = o_lm.lm_loss_and_grads(sd_ragged, fx["input_ids"], fx["attention_mask"], fx["feats"])
    assert abs(loss.item() - fx["loss"].item()) <= 1e-5
    for k, v in fx["grad_norms"].items():
        assert abs(grads[k].norm().item() - v) <= 1e-4 * v + 1e-9, k
    for k, ref in fx["probes"].items():
        g = grads[k]
        got = g[::37, ::41] if g.dim() == 2 else g[::7]
        assert (got - ref).abs().max().item() <= 1e-5 * g.abs().max().item() + 1e-9, k
