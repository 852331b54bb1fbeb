"""Round 6: the two argument forms of ``LanguageModel.forward`` that still raised NotImplementedError (VERDICT r05 "missing" 2
and 3) against fixtures recorded from the REAL reference (tests/golden/make_golden_lm_positions_padding.py):
arbitrary ``position_ids`` in the teacher-forced pass (src/language_model/language_model.py:293-307) - eval loss, logits, and the
training pass's gradients against autograd through the CPU oracle - and padding inside ``forward(use_cache=True)`` (:316-334)."""
import pytest
import torch

from conftest import gpu_model, load_golden, synth_sd
from oracle import language_model as o_lm

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def test_teacher_forced_pass_embeds_the_position_ids_it_is_given():
    fx = load_golden("lm_positions_padding.pt")
    m = gpu_model(fx["meta"]["profile"])
    lm = m.language_model
    for name, c in fx["teacher_forced"].items():
        ids, am, feats, pos = c["input_ids"].to(DEV), c["attention_mask"].to(DEV), c["feats"].to(DEV), c["position_ids"].to(DEV)
        loss = lm(ids.clone(), am, feats, return_loss=True, position_ids=pos)
        assert abs(float(loss) - float(c["loss"])) <= 2e-4, name
        default = lm(ids.clone(), am, feats, return_loss=True)
        assert abs(float(default) - float(c["loss_default_positions"])) <= 2e-4, name          # None still means arange(T)
        logits = lm.teacher_forced_logits(ids, am, feats, position_ids=pos)
        got = torch.stack([logits[s, t] for s, t in c["probes"]]).cpu()
        assert (got - c["probe_logits"]).abs().max().item() <= 2e-3, name
    # a position outside the token table is an index error, like a token id (torch.nn.Embedding raises in the reference)
    c = fx["teacher_forced"]["row_s4_t12"]
    bad = c["position_ids"].clone()
    bad[0, 3] = 50257
    with pytest.raises(IndexError):
        lm(c["input_ids"].to(DEV), c["attention_mask"].to(DEV), c["feats"].to(DEV), return_loss=True, position_ids=bad.to(DEV))
        torch.cuda.synchronize()
        lm(c["input_ids"].to(DEV), c["attention_mask"].to(DEV), c["feats"].to(DEV), return_loss=True)   # the deferred report surfaces here at the latest
    with pytest.raises(ValueError):
        lm(c["input_ids"].to(DEV), c["attention_mask"].to(DEV), c["feats"].to(DEV), return_loss=True, position_ids=torch.zeros((3, 12), dtype=torch.int64))


def test_training_pass_with_position_ids_matches_autograd_through_the_oracle():
    """Loss and the gradients of uk / uv (layers 0 and 23) and feature_space_transformation_nn of the HIP training pass (fp32,
    dropout off) with a [S,T] position table against torch autograd through the oracle's forward on the CPU."""
    fx = load_golden("lm_positions_padding.pt")
    c = fx["teacher_forced"]["table_s5_t9"]
    m = gpu_model(fx["meta"]["profile"])
    lm = m.language_model
    g = "language_model.gpt_with_lm_head.transformer."
    keys = [g + "h.0.attn.uk.weight", g + "h.23.attn.uv.weight", "language_model.feature_space_transformation_nn.0.weight",
            "language_model.feature_space_transformation_nn.2.bias"]
    ids, am, feats, pos = c["input_ids"], c["attention_mask"], c["feats"], c["position_ids"]
    ref_loss, ref_grads = o_lm.lm_loss_and_grads(synth_sd(fx["meta"]["profile"]), ids, am, feats, position_ids=pos)
    assert abs(float(ref_loss) - float(c["loss"])) <= 1e-5   # the oracle's autograd pass is the reference's loss
    was_training, p_keep = lm.training, lm.dropout_p
    try:
        lm.train()
        lm.dropout_p = 0.0
        for p in lm.trainable_parameters():
            p.grad = None
        loss = lm(ids.to(DEV).clone(), am.to(DEV), feats.to(DEV), return_loss=True, position_ids=pos.to(DEV))
        loss.backward()
        assert abs(float(loss.detach()) - float(ref_loss)) <= 2e-4
        h = lm.gpt.h
        got = [h[0].attn.uk.weight.grad, h[23].attn.uv.weight.grad, lm.feature_space_transformation_nn[0].weight.grad,
               lm.feature_space_transformation_nn[2].bias.grad]
        for k, gp in zip(keys, got):
            want = ref_grads[k]
            rel = (gp.cpu() - want).norm().item() / max(want.norm().item(), 1e-12)
            assert rel <= 2e-3, (k, rel)
    finally:
        lm.dropout_p = p_keep
        lm.train(was_training)
        for p in lm.trainable_parameters():
            p.grad = None


def test_incremental_forward_over_a_left_padded_prompt_matches_reference_fixture():
    """forward(use_cache=True) on a prompt whose first 0-3 positions are padding (attention_mask zeros, positions counted from
    the first real token), then one cached step with the grown mask: logits of both calls and the presents of layers 0 / 23
    within 2e-3 of the REAL reference's; with an all-ones mask the result is measurably different (the mask is applied)."""
    fx = load_golden("lm_positions_padding.pt")
    c = fx["cached"]
    m = gpu_model(fx["meta"]["profile"])
    lm = m.language_model
    feats = c["feats"].to(DEV)
    l1, presents = lm(c["prompt"].to(DEV), c["mask"].to(DEV), feats, return_loss=False, position_ids=c["position_ids"].to(DEV), use_cache=True)
    assert (l1[:, -1].cpu() - c["logits_prompt_last"]).abs().max().item() <= 2e-3
    assert (l1[:, :, ::97].cpu() - c["logits_prompt_probe"]).abs().max().item() <= 2e-3
    l2, presents = lm(c["next"].to(DEV), c["mask2"].to(DEV), feats, return_loss=False, past_key_values=presents,
                      position_ids=c["position_ids2"].to(DEV), use_cache=True)
    assert (l2[:, -1].cpu() - c["logits_next"]).abs().max().item() <= 2e-3
    for l, (k, v) in c["presents"].items():
        assert presents[l][0].shape == k.shape
        assert (presents[l][0].cpu() - k).abs().max().item() <= 2e-3 and (presents[l][1].cpu() - v).abs().max().item() <= 2e-3
    with pytest.raises(ValueError):   # the mask must cover every key of the call (the reference's broadcast fails otherwise)
        lm(c["next"].to(DEV), c["mask"].to(DEV)[:, :3] * 0, feats, return_loss=False, past_key_values=presents, use_cache=True)
    l1_ones, _ = lm(c["prompt"].to(DEV), torch.ones_like(c["mask"]).to(DEV), feats, return_loss=False, position_ids=c["position_ids"].to(DEV),
                    use_cache=True)
    assert (l1_ones[:, -1] - l1[:, -1]).abs().max().item() > 0.1
