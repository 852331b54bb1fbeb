"""Round 5: the 16-bit activation flow of the training pass (BASELINE configs[4]; the reference wraps its training step in
torch.autocast, src/full_model/train_full_model.py:172-237) and the kernels under it - the 256 x 256 ping-pong GEMM with the
training epilogues, the fused residual / dropout / LayerNorm and LayerNorm-backward kernels, the 16-bit cross-entropy
gradient.  The fp32 pass (pinned against the real reference's autograd by tests/golden/lm_grads*.pt) is the yardstick of the
16-bit passes; tolerances are the 16-bit noise levels written next to each check."""
import math

import pytest
import torch
import torch.nn.functional as F

from conftest import gpu_model, synth_sd
from oracle import language_model as o_lm
from rgrg_amd import _hip

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
T16 = {0: torch.bfloat16, 1: torch.float16}


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _gelu_new_grad(x):
    k = math.sqrt(2.0 / math.pi)
    u = k * (x + 0.044715 * x ** 3)
    th = torch.tanh(u)
    return 0.5 * (1 + th) + 0.5 * x * (1 - th * th) * k * (1 + 3 * 0.044715 * x * x)


@pytest.mark.parametrize("fp16", [0, 1])
@pytest.mark.parametrize("tile", [5, 1 + 16 * 2])
@pytest.mark.parametrize("M,N,K,mode", [(2085, 1024, 1024, "res"), (4100, 768, 256, "plain"), (300, 700, 512, "gelu16"),
                                        (2300, 1280, 1024, "pre16"), (2300, 1024, 768, "gbwd"), (513, 257, 2048, "out16")])
def test_pingpong_gemm_and_training_epilogues(M, N, K, mode, tile, fp16):
    """gemm_bf16_pp_kernel (tile 5: 8 waves, 256 x 256, two wave groups alternating between the matrix core and the LDS side)
    and the 128 x 128 LDS-DMA kernel (grouped tile order: 18+ row tiles) against a float64 product of the same 16-bit
    operands: ragged row / column tiles, the shortest K (4 tiles), K = 2048; residual, fused gelu with 16-bit output, the
    pre-activation copy (Ypre16) next to the activated output, the gelu' multiplier (G16)."""
    if K % 256:
        pytest.skip("K must be a multiple of 256")
    lib = _hip.load()
    t16 = T16[fp16]
    g = torch.Generator().manual_seed(M + 3 * N + K)
    A = torch.randn((M, K), generator=g).to(t16)
    W = (torch.randn((N, K), generator=g) / math.sqrt(K)).to(t16)
    b = torch.randn((N,), generator=g)
    R = torch.randn((M, N), generator=g) if mode == "res" else None
    G = torch.randn((M, N), generator=g).to(t16) if mode == "gbwd" else None
    pre = A.double() @ W.double().t() + b.double()
    if R is not None:
        pre = pre + R.double()
    act = 2 if mode in ("gelu16", "pre16") else 0
    ref = F.gelu(pre, approximate="tanh") if act == 2 else pre
    if G is not None:
        ref = pre * _gelu_new_grad(G.double())
    A16, Wb = A.view(torch.int16).to(DEV), W.view(torch.int16).to(DEV)
    out16 = mode in ("gelu16", "pre16", "gbwd", "out16")
    y = torch.full((M, N), float("nan"), device=DEV)
    y16 = torch.zeros((M, N), dtype=torch.int16, device=DEV)
    p16 = torch.zeros((M, N), dtype=torch.int16, device=DEV)
    Gd = G.view(torch.int16).to(DEV) if G is not None else None
    Rd = R.to(DEV) if R is not None else None
    _hip.check(lib.rgrg_debug_linear_bf16_train(A16.data_ptr(), Wb.data_ptr(), b.to(DEV).data_ptr(), Rd.data_ptr() if R is not None else None,
                                                None if out16 else y.data_ptr(), y16.data_ptr() if out16 else None,
                                                p16.data_ptr() if mode == "pre16" else None, Gd.data_ptr() if G is not None else None,
                                                M, N, K, N, act, tile, fp16, _stream()))
    scale = ref.abs().max().item()
    if out16:
        got = y16.cpu().view(t16).double()
        assert (got - ref).abs().max().item() <= 2 ** -7 * scale          # one 16-bit rounding of an fp32-accurate value
    else:
        err = (y.cpu().double() - ref).abs()
        assert bool((err <= 2e-6 * scale + 2e-5 * ref.abs()).all()), err.max().item()   # fp32 accumulation of exact 16-bit products
    if mode == "pre16":
        gotp = p16.cpu().view(t16).double()
        assert (gotp - pre).abs().max().item() <= 2 ** -7 * pre.abs().max().item()


def _lm_train_model():
    import rgrg_amd
    m = rgrg_amd.ReportGenerationModel(pretrain_without_lm_model=False)
    m.load_state_dict(synth_sd("ragged"))
    m.to(torch.device("cuda", 0))
    m.language_model.dropout_p = 0.0
    return m


def _batch(S, T, seed, min_len=2):
    g = torch.Generator().manual_seed(seed)
    ids = torch.randint(0, 50257, (S, T), generator=g)
    lens = torch.randint(min_len, T + 1, (S,), generator=g)
    lens[0] = T
    am = (torch.arange(T)[None, :] < lens[:, None]).to(torch.int64)
    feats = torch.randn((S, 1024), generator=g)
    return ids, am, feats


def _grads(m, ids, am, feats, autocast=None, seed=None):
    lm = m.language_model
    for p in m.parameters():
        p.grad = None
    if seed is not None:
        lm.dropout_seed = seed
    if autocast is None:
        loss = lm(ids.clone().to(DEV), am.to(DEV), feats.to(DEV), return_loss=True)
    else:
        with torch.autocast("cuda", dtype=autocast):
            loss = lm(ids.clone().to(DEV), am.to(DEV), feats.to(DEV), return_loss=True)
    loss.backward()
    return loss.item(), {k: p.grad.detach().clone() for k, p in m.named_parameters() if p.grad is not None}


def _compare(got, ref, cos_min, norm_tol, big_only=True):
    worst = (1.0, 0.0, "")
    for k, r in ref.items():
        if big_only and r.numel() < 1024 * 1024:
            continue
        a, b = got[k].reshape(-1).double(), r.reshape(-1).double()
        cos = (torch.dot(a, b) / (a.norm() * b.norm())).item()
        nr = abs(a.norm().item() / b.norm().item() - 1)
        assert cos >= cos_min and nr <= norm_tol, (k, cos, nr)
        if cos < worst[0]:
            worst = (cos, nr, k)
    return worst


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_train16_with_dropout_follows_the_fp32_pass_with_the_same_masks(dtype):
    """240 token rows (> 128: the 16-bit flow) with dropout 0.25 at all four sites and the SAME seed as an fp32 pass: the
    masks are pure functions of (seed, site, element), recomputed by the fused residual / dropout / LayerNorm kernel, by the
    LayerNorm-backward kernel (the mask of the branch the gradient enters next) and by both attention-backward kernels.  A
    wrong mask anywhere decorrelates the gradients (cosine ~0.75 at p = 0.25); 16-bit noise alone keeps cosine >= 0.99."""
    m = _lm_train_model()
    lm = m.language_model
    lm.train()
    lm.dropout_p = 0.25
    ids, am, feats = _batch(10, 24, 5)
    l32, g32 = _grads(m, ids, am, feats, None, seed=1000)
    l16, g16 = _grads(m, ids, am, feats, dtype, seed=1000)
    lother, gother = _grads(m, ids, am, feats, None, seed=2000)
    assert abs(l16 - l32) <= 3e-2 and l16 != l32
    cos, nr, k = _compare(g16, g32, 0.99, 0.03)
    # the same comparison against OTHER masks fails clearly: the test can see a wrong mask
    a, b = gother[k].reshape(-1).double(), g32[k].reshape(-1).double()
    assert (torch.dot(a, b) / (a.norm() * b.norm())).item() < 0.97
    m.invalidate_engine()


@pytest.mark.parametrize("S,T", [(2, 150), (3, 127), (5, 33), (4, 96)])
def test_train16_attention_paths(S, T):
    """T + 1 <= 128 keys: the 16-bit attention kernels (attn_train16.hip) - 127 tokens = the full 4 key tiles, 33 tokens = 34 keys
    (two keys in the second tile), 96 tokens = a key tile holding only the last key; 150 tokens: the fp32 attention kernels
    with 16-bit outputs.  Dropout 0.2 with the seed of an fp32 pass: the attention-probability masks are recomputed in the
    forward kernel and in both halves of the fused backward kernel.  16-bit noise: cosine >= 0.99, norms within 3 %."""
    m = _lm_train_model()
    lm = m.language_model
    lm.train()
    lm.dropout_p = 0.2
    ids, am, feats = _batch(S, T, 100 + T)
    l32, g32 = _grads(m, ids, am, feats, None, seed=77)
    l16, g16 = _grads(m, ids, am, feats, torch.bfloat16, seed=77)
    assert abs(l16 - l32) <= 3e-2
    _compare(g16, g32, 0.99, 0.03)
    m.invalidate_engine()


def test_fp16_training_gradients_keep_their_small_values():
    """under torch.autocast(float16) d(logits) = (softmax - onehot) / n_scored is 1e-4 .. 1e-9 - fp16's
    subnormal / flush-to-zero range - and GradScaler's scale only arrives after the pass.  The 16-bit flow therefore carries
    an internal power-of-two scale (2^15) from d(logits) to d(uk / uv), removed exactly where the gradients leave 16-bit
    storage.  2320 token rows (n_scored ~ 1700): every large gradient tensor within 3 % in norm and cosine >= 0.99 of the fp32
    pass, bias gradients (sums over few elements: the first to lose small addends) within 5 %."""
    m = _lm_train_model()
    m.language_model.train()
    ids, am, feats = _batch(58, 40, 11, min_len=20)
    l32, g32 = _grads(m, ids, am, feats, None)
    l16, g16 = _grads(m, ids, am, feats, torch.float16)
    assert abs(l16 - l32) <= 3e-2
    _compare(g16, g32, 0.99, 0.03)
    for k, r in g32.items():
        if r.dim() == 1 and r.numel() >= 1024:
            assert abs(g16[k].norm().item() / r.norm().item() - 1) <= 0.05, k
    m.invalidate_engine()


def test_configs4_shape_partition_property_and_oracle_subset():
    """BASELINE configs[4] at its real per-GPU shape: 8 images = 232 sentences x 64 tokens = 14 848 token rows in ONE lm_head /
    cross-entropy chunk, bf16 autocast, dropout off.  The loss is a mean over the scored tokens, so for a partition of the
    sentences into A (4 sentences) and B (228)   n G(full) = n_A G(A) + n_B G(B)   and the same for the loss - three passes of
    the 16-bit flow at 256 / 14 592 / 14 848 rows must agree (cosine >= 0.995, norms within 2 %: each pass has its own 16-bit
    noise), and pass A (256 rows) is checked against torch autograd through the CPU oracle (cosine >= 0.99, norms 3 %)."""
    m = _lm_train_model()
    m.language_model.train()
    S, T, NA = 232, 64, 4
    ids, am, feats = _batch(S, T, 64, min_len=T // 2)

    def scored(a):
        return int(a[:, 1:].sum().item())
    n, nA, nB = scored(am), scored(am[:NA]), scored(am[NA:])
    assert n == nA + nB
    lF, gF = _grads(m, ids, am, feats, torch.bfloat16)
    lA, gA = _grads(m, ids[:NA], am[:NA], feats[:NA], torch.bfloat16)
    lB, gB = _grads(m, ids[NA:], am[NA:], feats[NA:], torch.bfloat16)
    assert abs(lF * n - (lA * nA + lB * nB)) <= 2e-3 * lF * n
    combo = {k: (gA[k] * nA + gB[k] * nB) / n for k in gF}
    _compare(gF, combo, 0.995, 0.02)
    o_loss, o_grads = o_lm.lm_loss_and_grads(synth_sd("ragged"), ids[:NA], am[:NA], feats[:NA])
    assert abs(lA - o_loss.item()) <= 3e-2
    _compare({k: v.cpu() for k, v in gA.items()}, o_grads, 0.99, 0.03)
    m.invalidate_engine()


@pytest.mark.parametrize("fp16", [0, 1])
@pytest.mark.parametrize("M,N,K", [(923, 50257, 1024), (300, 1000, 256), (513, 257, 512)])
def test_pingpong_gemm_argmax_epilogue(M, N, K, fp16):
    """The greedy lm_head of the many-sequence decode step (language_model.py:420-428 `argmax(-1)`): the 256 x 256 kernel leaves
    per row and 256-column tile the maximum and its column instead of the logits.  Against the SAME kernel's logits (tile 5,
    fp32 output): values bit-identical, columns = torch.argmax per tile - including exact ties (duplicated weight rows inside a
    tile, across the lane halves / column quarters / tiles: the first maximum must win), ragged last row and column tiles."""
    lib = _hip.load()
    t16 = T16[fp16]
    g = torch.Generator().manual_seed(M + N + K + fp16)
    A = torch.randn((M, K), generator=g).to(t16)
    W = (torch.randn((N, K), generator=g) / math.sqrt(K)).to(t16)
    b = torch.randn((N,), generator=g) * 0.1
    # exact ties: copies of one weight row (and its shift) at columns that meet at every level of the reduction
    for src, dsts in ((5, (5 + 32, 5 + 64, 5 + 130)), (200, (201,)), (70, (70 + 256 if N > 600 else 71,))):
        for dcol in dsts:
            if dcol < N:
                W[dcol] = W[src]; b[dcol] = b[src]
    A16, Wb = A.view(torch.int16).to(DEV), W.view(torch.int16).to(DEV)
    bd = b.to(DEV)
    # make the tied columns the row maximum for a third of the rows: a large shift on them
    bd_t = bd.clone()
    bd_t[[c for c in (5, 37, 69, 135, 200, 201) if c < N]] += 50.0
    nt = (N + 255) // 256
    for shift in (bd, bd_t):
        y = torch.empty((M, N), device=DEV)
        _hip.check(lib.rgrg_debug_linear_bf16_train(A16.data_ptr(), Wb.data_ptr(), shift.data_ptr(), None, y.data_ptr(), None, None, None,
                                                    M, N, K, N, 0, 5, fp16, _stream()), "logits")
        cv = torch.full((M, nt), float("nan"), device=DEV)
        ci = torch.full((M, nt), -1, device=DEV, dtype=torch.int32)
        _hip.check(lib.rgrg_debug_linear_bf16_argmax(A16.data_ptr(), Wb.data_ptr(), shift.data_ptr(), M, N, K, cv.data_ptr(), ci.data_ptr(),
                                                     fp16, _stream()), "argmax epilogue")
        torch.cuda.synchronize()
        pad = torch.full((M, nt * 256 - N), float("-inf"), device=DEV)
        tiles = torch.cat([y, pad], dim=1).view(M, nt, 256)
        ref_v, ref_i = tiles.max(dim=2)
        first = (tiles == ref_v[..., None]).int().argmax(dim=2) + torch.arange(nt, device=DEV)[None, :] * 256   # first maximum of the tile
        assert torch.equal(cv, ref_v)
        assert torch.equal(ci.long(), first)
        # and the row's token = torch.argmax of the full logits row
        best_tile = (cv == cv.max(dim=1, keepdim=True).values).int().argmax(dim=1)
        assert torch.equal(ci.long().gather(1, best_tile[:, None])[:, 0], y.argmax(dim=1))


def test_a_rows_result_does_not_depend_on_the_tile_or_the_row_count():
    """What lets the many-sequence decode step run as row ranges (decoder.hip run_row_ranges) with bit-identical results: the
    16-bit GEMM of a row gives the same bits on every tile shape / stage count, and inside a launch of 923 rows as inside one of
    512 or 320 rows (the launcher picks other tiles for those) - plain, with gelu, as the CONSUMER of a folded LayerNorm (all 16
    statistics slots filled: their reduction has one order for 64- and 128-row tiles) and as its PRODUCER (16-bit copy + slots)."""
    lib = _hip.load()
    g = torch.Generator().manual_seed(3)
    M, K = 923, 1024
    A16 = torch.randn((M, K), generator=g).bfloat16().view(torch.int16).to(DEV)
    for N, act in ((3072, 0), (4096, 2), (1024, 0)):
        Wb = (torch.randn((N, K), generator=g) / math.sqrt(K)).bfloat16().view(torch.int16).to(DEV)
        b = torch.randn((N,), generator=g).to(DEV)
        outs = []
        for tile in (2 + 64, 2 + 48, 3 + 48, 1 + 32, 4 + 48):   # 64x64x4, 64x64x3, 128x64x3, 128x128x2, 64x128x3
            y = torch.empty((M, N), device=DEV)
            _hip.check(lib.rgrg_debug_linear_bf16_tile(A16.data_ptr(), Wb.data_ptr(), b.data_ptr(), None, y.data_ptr(), M, N, K, N, act, tile,
                                                       0, 0, 0, _stream()), "tile")
            outs.append(y)
        torch.cuda.synchronize()
        assert all(torch.equal(o, outs[0]) for o in outs[1:]), (N, act)
    # consumer of a folded LayerNorm: statistics in all 16 slots of a row
    N = 3072
    Wf = (torch.randn((N, K), generator=g) / math.sqrt(K)).bfloat16().view(torch.int16).to(DEV)
    sh, cs = torch.randn((N,), generator=g).to(DEV), torch.randn((N,), generator=g).to(DEV)
    x = A16.view(torch.bfloat16).float().view(M, 16, 64)
    stats = torch.stack([x.sum(2), (x * x).sum(2)], dim=2).contiguous()   # [M][16][2]
    ys = []
    for m in (923, 512, 320):
        y = torch.empty((m, N), device=DEV)
        _hip.check(lib.rgrg_debug_linear_bf16_ln(A16.data_ptr(), Wf.data_ptr(), sh.data_ptr(), None, y.data_ptr(), None, None, stats.data_ptr(),
                                                 cs.data_ptr(), m, N, K, N, 0, 0, _stream()), "ln consumer")
        ys.append(y)
    torch.cuda.synchronize()
    assert torch.equal(ys[0][:512], ys[1]) and torch.equal(ys[0][:320], ys[2])
    # producer (N = 1024, K = 4096, residual)
    K2 = 4096
    A2 = torch.randn((M, K2), generator=g).bfloat16().view(torch.int16).to(DEV)
    W2 = (torch.randn((1024, K2), generator=g) / math.sqrt(K2)).bfloat16().view(torch.int16).to(DEV)
    R, b2 = torch.randn((M, 1024), generator=g).to(DEV), torch.randn((1024,), generator=g).to(DEV)
    res = []
    for m in (923, 512, 320):
        y = torch.empty((m, 1024), device=DEV)
        yb = torch.empty((m, 1024), device=DEV, dtype=torch.int16)
        so = torch.zeros((m, 16, 2), device=DEV)
        _hip.check(lib.rgrg_debug_linear_bf16_ln(A2.data_ptr(), W2.data_ptr(), b2.data_ptr(), R.data_ptr(), y.data_ptr(), yb.data_ptr(), so.data_ptr(),
                                                 None, None, m, 1024, K2, 1024, 0, 0, _stream()), "ln producer")
        res.append((y, yb, so))
    torch.cuda.synchronize()
    for m, r in ((512, res[1]), (320, res[2])):
        assert all(torch.equal(a[:m], b) for a, b in zip(res[0], r)), m
