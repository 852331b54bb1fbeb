"""Round 6: the K-parity ping-pong GEMM (rgrg_amd/csrc/gemm_kp.inc) - the kernel of the four per-layer projections of the
many-sequence 16-bit decode step (GPT2Block c_attn / attn c_proj / c_fc / mlp c_proj, src/language_model/language_model.py:338-366)
- against a float64 product of the same 16-bit operands, against the LDS-DMA kernel it replaces there (same formulas, the K
sum re-associated as even tiles + odd tiles: fp32-noise apart), and for the property the row-range step needs: a row's bits do
not depend on how many rows the launch has."""
import math

import pytest
import torch
import torch.nn.functional as F

from rgrg_amd import _hip

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
T16 = {0: torch.bfloat16, 1: torch.float16}
KP_TILES = {6: "128x128x2", 7: "64x64x4", 8: "128x64x3", 9: "64x128x3", 10: "64x64x3", 12: "pr128x128x4"}


def _stream():
    return torch.cuda.current_stream().cuda_stream


@pytest.mark.parametrize("fp16", [0, 1])
@pytest.mark.parametrize("tile", sorted(KP_TILES))
@pytest.mark.parametrize("M,N,K,mode", [(923, 1024, 1024, "res"), (320, 3072, 1024, "plain"), (37, 256, 4096, "gelu"),
                                        (283, 1000, 1024, "res"), (129, 4096, 1024, "gelu"), (923, 1024, 4096, "res")])
def test_kp_gemm_plain_variants(M, N, K, mode, tile, fp16):
    """Every tile shape of the kernel on ragged row / column tiles (37 rows: one partial tile; N = 1000: a partial column
    tile), the shortest K it takes per stage count and K = 4096, residual and fused gelu."""
    lib = _hip.load()
    t16 = T16[fp16]
    g = torch.Generator().manual_seed(M + 3 * N + K + tile)
    A = torch.randn((M, K), generator=g).to(t16)
    W = (torch.randn((N, K), generator=g) / math.sqrt(K)).to(t16)
    b = torch.randn((N,), generator=g)
    R = torch.randn((M, N), generator=g) if mode == "res" else None
    pre = A.double() @ W.double().t() + b.double()
    if R is not None:
        pre = pre + R.double()
    act = 2 if mode == "gelu" else 0
    ref = F.gelu(pre, approximate="tanh") if act == 2 else pre
    A16, Wb = A.view(torch.int16).to(DEV), W.view(torch.int16).to(DEV)
    y = torch.full((M, N), float("nan"), device=DEV)
    Rd = R.to(DEV) if R is not None else None
    _hip.check(lib.rgrg_debug_linear_bf16_tile(A16.data_ptr(), Wb.data_ptr(), b.to(DEV).data_ptr(), Rd.data_ptr() if Rd is not None else None,
                                               y.data_ptr(), M, N, K, N, act, tile, 0, 0, fp16, _stream()), "kp tile")
    torch.cuda.synchronize()
    err = (y.cpu().double() - ref).abs().max().item()
    assert err <= 2e-5 * math.sqrt(K) + 1e-5 * ref.abs().max().item(), (KP_TILES[tile], err)


@pytest.mark.parametrize("fp16", [0, 1])
def test_kp_gemm_layernorm_fold_variants_follow_the_lds_dma_kernel(fp16):
    """Consumer (c_attn: fp32 out; c_fc: gelu, 16-bit out) and producer (attn_proj K = 1024, mlp_proj K = 4096: fp32 x in place
    of the residual, 16-bit copy, per-row statistics slots) of the folded LayerNorm on the K-parity kernel against the LDS-DMA
    kernel on the same operands: fp32 outputs within 1e-5 of the output range + 2e-6 relative (a re-associated fp32 sum of 16-64
    tile products), 16-bit outputs within one unit in the last place, statistics within 1e-5 relative."""
    lib = _hip.load()
    t16 = T16[fp16]
    g = torch.Generator().manual_seed(5 + fp16)
    M, K = 923, 1024
    A16 = torch.randn((M, K), generator=g).to(t16).view(torch.int16).to(DEV)
    x = A16.view(t16).float().view(M, 16, 64)
    stats = torch.stack([x.sum(2), (x * x).sum(2)], dim=2).contiguous()
    for N, act, out16 in ((3072, 0, False), (4096, 2, True)):
        Wf = (torch.randn((N, K), generator=g) / math.sqrt(K)).to(t16).view(torch.int16).to(DEV)
        sh, cs = torch.randn((N,), generator=g).to(DEV), (torch.randn((N,), generator=g) * 0.1).to(DEV)
        outs = []
        for kp in (0, 1):
            y = torch.empty((M, N), device=DEV)
            y16 = torch.empty((M, N), device=DEV, dtype=torch.int16)
            _hip.check(lib.rgrg_debug_linear_bf16_ln_kp(A16.data_ptr(), Wf.data_ptr(), sh.data_ptr(), None, None if out16 else y.data_ptr(),
                                                        y16.data_ptr() if out16 else None, None, None, stats.data_ptr(), cs.data_ptr(),
                                                        M, N, K, N, act, fp16, kp, _stream()), "consumer")
            outs.append(y16.view(t16).float() if out16 else y)
        torch.cuda.synchronize()
        span = outs[0].abs().max().item()
        tol = (2.0 ** -7 if fp16 == 0 else 2.0 ** -10) * span if out16 else 1e-5 * span
        assert (outs[0] - outs[1]).abs().max().item() <= tol, (N, act)
    for K2 in (1024, 4096):
        A2 = torch.randn((M, K2), generator=g).to(t16).view(torch.int16).to(DEV)
        W2 = (torch.randn((1024, K2), generator=g) / math.sqrt(K2)).to(t16).view(torch.int16).to(DEV)
        R, b2 = torch.randn((M, 1024), generator=g).to(DEV), torch.randn((1024,), generator=g).to(DEV)
        res = []
        for kp in (0, 1):
            y = R.clone()   # in place, as the decoder runs it: the residual stream x is both R and Y
            yb = torch.empty((M, 1024), device=DEV, dtype=torch.int16)
            so = torch.zeros((M, 16, 2), device=DEV)
            _hip.check(lib.rgrg_debug_linear_bf16_ln_kp(A2.data_ptr(), W2.data_ptr(), b2.data_ptr(), y.data_ptr(), y.data_ptr(), None, yb.data_ptr(),
                                                        so.data_ptr(), None, None, M, 1024, K2, 1024, 0, fp16, kp, _stream()), "producer")
            res.append((y, yb.view(t16).float(), so))
        torch.cuda.synchronize()
        span = res[0][0].abs().max().item()
        assert (res[0][0] - res[1][0]).abs().max().item() <= 1e-5 * span, K2
        assert (res[0][1] - res[1][1]).abs().max().item() <= (2.0 ** -7 if fp16 == 0 else 2.0 ** -10) * span, K2
        assert torch.allclose(res[0][2], res[1][2], rtol=1e-5, atol=1e-4 * span), K2
        # the slots are what they claim to be: per-row sums of the fp32 result over the 64-column blocks
        yk = res[1][0].view(M, 16, 64)
        assert torch.allclose(res[1][2][:, :, 0], yk.sum(2), rtol=1e-5, atol=1e-4 * span)
        assert torch.allclose(res[1][2][:, :, 1], (yk * yk).sum(2), rtol=1e-5, atol=1e-3 * span)


def test_kp_gemm_row_bits_do_not_depend_on_the_row_count():
    """The many-sequence step runs as row ranges (decoder.hip run_row_ranges) and must give the bits of the one-range step: the
    K-parity kernel's tile is a function of (N, K) only, so the rows of a 923-row launch equal those of 512- and 320-row
    launches - consumer, producer (statistics included) and plain."""
    lib = _hip.load()
    g = torch.Generator().manual_seed(31)
    M, K = 923, 1024
    A16 = torch.randn((M, K), generator=g).bfloat16().view(torch.int16).to(DEV)
    x = A16.view(torch.bfloat16).float().view(M, 16, 64)
    stats = torch.stack([x.sum(2), (x * x).sum(2)], dim=2).contiguous()
    N = 3072
    Wf = (torch.randn((N, K), generator=g) / math.sqrt(K)).bfloat16().view(torch.int16).to(DEV)
    sh, cs = torch.randn((N,), generator=g).to(DEV), torch.randn((N,), generator=g).to(DEV)
    ys = []
    for m in (923, 512, 320):
        y = torch.empty((m, N), device=DEV)
        _hip.check(lib.rgrg_debug_linear_bf16_ln_kp(A16.data_ptr(), Wf.data_ptr(), sh.data_ptr(), None, y.data_ptr(), None, None, None,
                                                    stats.data_ptr(), cs.data_ptr(), m, N, K, N, 0, 0, 1, _stream()), "consumer")
        ys.append(y)
    torch.cuda.synchronize()
    assert torch.equal(ys[0][:512], ys[1]) and torch.equal(ys[0][:320], ys[2])
    K2 = 4096
    A2 = torch.randn((M, K2), generator=g).bfloat16().view(torch.int16).to(DEV)
    W2 = (torch.randn((1024, K2), generator=g) / math.sqrt(K2)).bfloat16().view(torch.int16).to(DEV)
    R, b2 = torch.randn((M, 1024), generator=g).to(DEV), torch.randn((1024,), generator=g).to(DEV)
    res = []
    for m in (923, 512, 320):
        y = torch.empty((m, 1024), device=DEV)
        yb = torch.empty((m, 1024), device=DEV, dtype=torch.int16)
        so = torch.zeros((m, 16, 2), device=DEV)
        _hip.check(lib.rgrg_debug_linear_bf16_ln_kp(A2.data_ptr(), W2.data_ptr(), b2.data_ptr(), R.data_ptr(), y.data_ptr(), None, yb.data_ptr(),
                                                    so.data_ptr(), None, None, m, 1024, K2, 1024, 0, 0, 1, _stream()), "producer")
        res.append((y, yb, so))
    torch.cuda.synchronize()
    for m, r in ((512, res[1]), (320, res[2])):
        assert all(torch.equal(a[:m], b) for a, b in zip(res[0], r)), m
