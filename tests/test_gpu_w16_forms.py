"""Round 6: the fused decode plan on 16-bit weight fragments (33-64 token rows under torch.autocast, skinny_direct.inc W16) in the call
forms other than greedy generate(): beam search (beam rows read the K/V cache through the ancestor table) and the incremental
``forward(use_cache=True)`` (token / position overrides).  The greedy form is pinned against the 16-bit oracle in
tests/test_gpu_fp16.py; here the same kernels are held to the fp32 path of the SAME call, at the noise level of the 16-bit type."""
import pytest
import torch

from conftest import gpu_model

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _feats(n, seed):
    return torch.randn((n, 1024), generator=torch.Generator().manual_seed(seed)).to(DEV)


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
def test_beam_search_on_the_16bit_fused_plan_stays_close_to_the_fp32_beams(dtype):
    """12 regions x 4 beams = 48 beam rows (two row tiles): under autocast they run the fused plan on 16-bit weights.  Same shapes and
    BOS as the fp32 beams; most tokens equal (a near-tie flipped by 16-bit rounding changes the rest of that hypothesis)."""
    m = gpu_model("bench")
    feats = _feats(12, 41)
    a = m.language_model.generate(feats, max_length=12, num_beams=4, early_stopping=False)
    assert m.engine().fused_row_limit() == 128
    with torch.autocast("cuda", dtype=dtype):
        b = m.language_model.generate(feats, max_length=12, num_beams=4, early_stopping=False)
        assert m.engine().fused_row_limit() == 64          # the decoder is in its 16-bit mode: fused plan up to 64 rows
        b2 = m.language_model.generate(feats, max_length=12, num_beams=4, early_stopping=False)
    assert a.shape == b.shape and (b[:, 0] == 50256).all()
    assert torch.equal(b, b2)                              # deterministic
    agree = (a == b).float().mean().item()
    assert agree >= (0.85 if dtype == torch.float16 else 0.70), agree


def test_incremental_forward_on_the_16bit_fused_plan_close_to_fp32():
    """forward(use_cache=True) over 40 rows under fp16 autocast: a 3-token prompt, then two single-token steps with explicit
    position_ids - logits within the fp16 noise level (5e-3 of the logit range) of the fp32 call sequence, identical arg-max on
    >= 90 % of the rows."""
    m = gpu_model("bench")
    lm = m.language_model
    S = 40
    feats = _feats(S, 43)
    g = torch.Generator().manual_seed(5)
    prompt = torch.randint(0, 50257, (S, 3), generator=g).to(DEV)
    nxt = [torch.randint(0, 50257, (S, 1), generator=g).to(DEV) for _ in range(2)]

    def run():
        out = []
        logits, presents = lm(prompt, torch.ones((S, 3), device=DEV), feats, return_loss=False, use_cache=True)
        out.append(logits[:, -1].float().clone())
        for j, t in enumerate(nxt):
            am = torch.ones((S, 4 + j), device=DEV)
            logits, presents = lm(t, am, feats, return_loss=False, past_key_values=presents, position_ids=torch.full((S, 1), 3 + j, device=DEV),
                                  use_cache=True)
            out.append(logits[:, -1].float().clone())
        return out
    ref = run()
    with torch.autocast("cuda", dtype=torch.float16):
        low = run()
    for r, l in zip(ref, low):
        rng = r.abs().max().item()
        assert (r - l).abs().max().item() <= 5e-3 * rng
        assert (r.argmax(-1) == l.argmax(-1)).float().mean().item() >= 0.90


@pytest.mark.parametrize("S", [33, 64, 65])
def test_row_count_boundaries_of_the_autocast_plans(S):
    """33 = the first row count on the 16-bit-weight fused plan (32 stay bit-exact fp32), 64 its last, 65 the first on the
    many-sequence 16-bit path: greedy ids of 10 tokens under bf16 autocast agree with the fp32 ids on most positions (a flipped near-tie
    changes the rest of that row), deterministic, BOS first; and a 32-row call under autocast IS the fp32 result."""
    m = gpu_model("bench")
    feats = _feats(S, 47)
    ref = m.language_model.generate(feats, max_length=10)
    with torch.autocast("cuda", dtype=torch.bfloat16):
        low = m.language_model.generate(feats, max_length=10)
        low2 = m.language_model.generate(feats, max_length=10)
        exact = m.language_model.generate(feats[:32], max_length=10)
    assert low.shape == ref.shape and (low[:, 0] == 50256).all() and torch.equal(low, low2)
    assert (low == ref).float().mean().item() >= 0.70
    assert torch.equal(exact, m.language_model.generate(feats[:32], max_length=10))
