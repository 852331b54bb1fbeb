"""The detector's validation-loss bookkeeping as HIP kernels (csrc/det_train.hip: rgrg_balanced_sample - an exact
radix select per image -, rgrg_roi_add_gt_f32, rgrg_roi_gather_samples_f32) against the per-image oracle (oracle/tv013.py:
balanced_sample, select_training_samples).  The oracle draws with an injected permutation; the kernels are driven by
per-element keys built from the oracle's choice (chosen: 0, everything else: 1 - the smallest keys are sampled, lower index
first on ties), so both sides take the same candidates and every integer output is compared bit-exactly.  Without keys the
kernel draws Philox words itself: checked as a valid balanced sample, reproducible under torch.manual_seed, and uniform."""
import pytest
import torch

from conftest import gpu_model
from oracle import tv013

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _perm(seed):
    g = torch.Generator().manual_seed(seed)
    return lambda n, tag: torch.randperm(n, generator=g)


def _engine():
    return gpu_model("ragged").engine()


def _codes(labels):
    """float labels (1 / 0 / -1) -> match codes the RPN sampler derives them from (>= 0 positive, -1 below, -2 between)."""
    m = torch.full(labels.shape, -1, dtype=torch.int32)
    m[labels >= 1] = 3
    m[labels < 0] = -2
    return m.to(DEV)


def _keys(t):
    return lambda stage, B, n: t


def test_balanced_sample_kernel_reproduces_the_oracles_choice_through_keys():
    eng = _engine()
    g = torch.Generator().manual_seed(1)
    B, n = 4, 163840                                              # the RPN's anchor count at 512 x 512
    labels = torch.zeros((B, n))
    labels[0, torch.randperm(n, generator=g)[:40]] = 1.0          # fewer positives than the cap
    labels[1, torch.randperm(n, generator=g)[:500]] = 1.0         # more positives than the cap
    labels[2] = -1.0                                              # nothing but ignored and ...
    labels[2, 7000:7100] = 0.0                                    # ... fewer negatives than the batch
    labels[3, torch.randperm(n, generator=g)[:1000]] = -1.0       # no positives at all
    perm = _perm(5)
    keys = torch.ones((B, n))
    want = torch.zeros((B, n), dtype=torch.uint8)
    for i in range(B):
        p, q = tv013.balanced_sample(labels[i], 256, 0.5, perm, ("rpn", i))
        keys[i, p] = 0.0
        keys[i, q] = 0.0
        want[i, p] = 1
        want[i, q] = 2
    mask, lst, cnt = eng._sample("rpn", _codes(labels), None, None, 256, 128, _keys(keys.to(DEV)))
    assert torch.equal(mask.cpu(), want)
    assert cnt.tolist() == [256, 256, 100, 256]
    for i in range(B):
        idx = want[i].nonzero().flatten()                          # torch.where(pos | neg): ascending
        assert torch.equal(lst[i, :idx.numel()].cpu().to(torch.int64), idx) and bool((lst[i, idx.numel():] == 0).all())
    got = mask.cpu()
    assert (got[0] == 1).sum() == 40 and (got[0] == 2).sum() == 216 and (got[1] == 1).sum() == 128 and (got[1] == 2).sum() == 128
    assert (got[2] == 2).sum() == 100 and (got[2] == 1).sum() == 0 and (got[3] == 2).sum() == 256


@pytest.mark.parametrize("n", [10, 777, 2000, 70001])
def test_balanced_sample_kernel_takes_the_smallest_keys_lower_index_first(n):
    eng = _engine()
    g = torch.Generator().manual_seed(2 + n)
    B = 3
    labels = (torch.rand((B, n), generator=g) < 0.2).float()
    labels[torch.rand((B, n), generator=g) < 0.1] = -1.0
    keys = torch.rand((B, n), generator=g) - 0.25                  # negative keys too
    keys[1] = (keys[1] * 8).round() / 8                            # a handful of distinct values: ties everywhere
    keys[2, ::3] = 0.0
    keys[2, 1::3] = -0.0
    mask, lst, cnt = eng._sample("rpn", _codes(labels), None, None, 512, 128, _keys(keys.to(DEV)))
    m = mask.cpu()
    for i in range(B):
        for cls, cand, k in ((1, labels[i] >= 1, None), (2, labels[i] == 0, None)):
            k_pos = min(int((labels[i] >= 1).sum()), 128)
            k = k_pos if cls == 1 else min(int((labels[i] == 0).sum()), 512 - k_pos)
            idx = cand.nonzero().flatten()
            order = torch.sort(keys[i][idx], stable=True).indices[:k]    # smallest keys, lower index first on ties
            want = torch.zeros(n, dtype=torch.bool)
            want[idx[order]] = True
            assert torch.equal(m[i] == cls, want), (i, cls)
        assert int(cnt[i]) == int((m[i] != 0).sum())
        assert torch.equal(lst[i, :int(cnt[i])].cpu().to(torch.int64), (m[i] != 0).nonzero().flatten())


def test_balanced_sample_kernel_draws_its_own_keys_reproducibly_and_uniformly():
    eng = _engine()
    g = torch.Generator().manual_seed(9)
    B, n = 2, 5000
    labels = (torch.rand((B, n), generator=g) < 0.3).float()
    labels[torch.rand((B, n), generator=g) < 0.1] = -1.0
    codes = _codes(labels)
    torch.manual_seed(123)
    m1, l1, c1 = eng._sample("roi", codes, None, None, 512, 128, None)
    m2, _, _ = eng._sample("roi", codes, None, None, 512, 128, None)           # the next draw differs
    torch.manual_seed(123)
    m3, l3, c3 = eng._sample("roi", codes, None, None, 512, 128, None)          # the seed reproduces it
    assert torch.equal(m1, m3) and torch.equal(l1, l3) and torch.equal(c1, c3) and not torch.equal(m1, m2)
    assert not torch.equal(m1[0][labels[0] >= 1], m1[1][labels[1] >= 1][:int((labels[0] >= 1).sum())])
    for i in range(B):
        pos, neg = (labels[i] >= 1).to(DEV), (labels[i] == 0).to(DEV)
        assert bool(((m1[i] == 1) <= pos).all()) and bool(((m1[i] == 2) <= neg).all())
        assert int((m1[i] == 1).sum()) == 128 and int((m1[i] == 2).sum()) == 384 and int(c1[i]) == 512
    # uniformity: over 400 draws every positive of image 0 is taken with probability 128 / #positive
    hits = torch.zeros(n, device=DEV)
    for _ in range(400):
        m, _, _ = eng._sample("rpn", codes, None, None, 512, 128, None)
        hits += (m[0] == 1)
    npos = int((labels[0] >= 1).sum())
    p = 128 / npos
    f = hits[(labels[0] >= 1).to(DEV)] / 400
    sd = (p * (1 - p) / 400) ** 0.5
    assert abs(float(f.mean()) - p) < 1e-6 and float((f - p).abs().max()) < 5.5 * sd
    assert abs(float(f.std()) - sd) < 0.15 * sd                     # neither clumped nor too regular


def _cpu_match_codes(gt, gt_count, boxes, box_count):
    B, N = boxes.shape[:2]
    out = torch.full((B, N), -1, dtype=torch.int32)
    for b in range(B):
        n, ng = int(box_count[b]), int(gt_count[b])
        if ng:
            out[b, :n] = tv013.matcher(tv013.box_iou(gt[b, :ng], boxes[b, :n]), 0.5, 0.5, False).to(torch.int32)
    return out


def test_select_training_samples_kernels_equal_the_oracle_per_image():
    eng = _engine()
    g = torch.Generator().manual_seed(3)
    B, P, G = 3, 700, 6
    counts = torch.tensor([700, 650, 300], dtype=torch.int32)
    gcount = torch.tensor([6, 0, 3], dtype=torch.int32)            # one image without ground truth
    xy = torch.rand((B, P, 2), generator=g) * 400
    props = torch.cat([xy, xy + 20 + torch.rand((B, P, 2), generator=g) * 100], 2)
    gxy = torch.rand((B, G, 2), generator=g) * 300
    gt = torch.cat([gxy, gxy + 60 + torch.rand((B, G, 2), generator=g) * 100], 2)
    gl = torch.randint(1, 30, (B, G), generator=g)
    for b in range(B):   # proposals near the ground truth so that there are positives
        for k in range(int(gcount[b])):
            props[b, 10 * k:10 * k + 10] = gt[b, k] + torch.randn((10, 4), generator=g) * 3
        props[b, int(counts[b]):] = 0
        gt[b, int(gcount[b]):] = 0
        gl[b, int(gcount[b]):] = 0
    plist = [props[b, :int(counts[b])] for b in range(B)]
    targets = [{"boxes": gt[b, :int(gcount[b])], "labels": gl[b, :int(gcount[b])]} for b in range(B)]
    rec = {}
    orig = tv013.balanced_sample

    def recording(labels, batch, frac, perm_fn, tag):
        p, q = orig(labels, batch, frac, perm_fn, tag)
        rec[tag] = (p, q)
        return p, q
    tv013.balanced_sample = recording
    try:
        o_props, o_labels, o_reg = tv013.select_training_samples(plist, targets, _perm(11))
    finally:
        tv013.balanced_sample = orig
    N = P + G
    keys = torch.ones((B, N))
    for b in range(B):
        p, q = rec[("roi", b)]
        keys[b, p] = 0.0
        keys[b, q] = 0.0
    taps = {}
    props_s, offsets, labels_flat, reg = eng._select_training_samples(
        props.to(DEV), counts.to(DEV), gt.to(DEV), gcount.to(DEV), gl.to(DEV), _keys(keys.to(DEV)), taps=taps)
    # add_gt_proposals and the match codes, bit-exact
    boxes = taps["roi_boxes"].cpu()
    assert taps["roi_box_count"].tolist() == (counts + gcount).tolist()
    for b in range(B):
        c, q = int(counts[b]), int(gcount[b])
        assert torch.equal(boxes[b, :c], props[b, :c]) and torch.equal(boxes[b, c:c + q], gt[b, :q]) and bool((boxes[b, c + q:] == 0).all())
    assert torch.equal(taps["roi_matched"].cpu(), _cpu_match_codes(gt, gcount, boxes, counts + gcount))
    ks = [int(p.shape[0]) for p in o_props]
    assert offsets.dtype == torch.int32 and offsets.tolist() == [0, ks[0], ks[0] + ks[1], sum(ks)]
    assert props_s.shape == (B, 512, 4)
    R = sum(ks)
    for b in range(B):
        assert torch.equal(props_s[b, :ks[b]].cpu(), o_props[b]) and bool((props_s[b, ks[b]:] == 0).all())
    assert torch.equal(labels_flat[:R].cpu(), torch.cat(o_labels)) and bool((labels_flat[R:] == 0).all())
    o = torch.cat(o_reg)
    got = reg[:R].cpu()
    pos = torch.cat(o_labels) > 0                                   # only the positives' targets are ever read by the loss
    fin = torch.isfinite(o).all(1)
    assert bool(pos.any()) and bool((pos <= fin).all())
    # logf / IEEE division on the device against the CPU's: a few ulp
    assert torch.allclose(got[fin], o[fin], rtol=2e-6, atol=2e-6)
    assert bool((reg[R:] == 0).all())


def _philox4x32_10(ctr, key):
    """Philox4x32-10 (Salmon, Moraes, Dror, Shaw: Parallel random numbers - as easy as 1, 2, 3, SC'11), plain Python."""
    M0, M1, W0, W1 = 0xD2511F53, 0xCD9E8D57, 0x9E3779B9, 0xBB67AE85
    c, k = list(ctr), list(key)
    for _ in range(10):
        p0, p1 = M0 * c[0], M1 * c[2]
        c = [(p1 >> 32) ^ c[1] ^ k[0], p1 & 0xffffffff, (p0 >> 32) ^ c[3] ^ k[1], p0 & 0xffffffff]
        k = [(k[0] + W0) & 0xffffffff, (k[1] + W1) & 0xffffffff]
    return c


def test_sampler_keys_are_philox4x32_10_words():
    """The keys the sampler draws when none are injected are word 0 of Philox4x32-10 over counter (index, image, stage, 0)
    and key = the call's 64-bit seed: checked against the generator's published known-answer vector and a plain Python
    evaluation (the keys are left in the caller's work space)."""
    assert _philox4x32_10((0, 0, 0, 0), (0, 0)) == [0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8]        # Random123 kat_vectors
    assert _philox4x32_10((0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344), (0xa4093822, 0x299f31d0)) == \
        [0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1]
    import ctypes as C
    from rgrg_amd import _hip
    lib = _hip.load()
    B, n = 3, 1000
    matched = torch.zeros((B, n), dtype=torch.int32, device=DEV)
    for seed, stage in ((0, 0), (0x0123456789abcdef, 1)):
        ws = torch.zeros((B, n), dtype=torch.int32, device=DEV)
        mask = torch.empty((B, n), dtype=torch.uint8, device=DEV)
        lst = torch.empty((B, 64), dtype=torch.int32, device=DEV)
        cnt = torch.empty((B,), dtype=torch.int32, device=DEV)
        _hip.check(lib.rgrg_balanced_sample(matched.data_ptr(), None, 0, None, None, C.c_uint64(seed), stage, B, n, 64, 32, ws.data_ptr(),
                                            mask.data_ptr(), lst.data_ptr(), cnt.data_ptr(), torch.cuda.current_stream().cuda_stream),
                   "rgrg_balanced_sample")
        got = ws.cpu().to(torch.int64) & 0xffffffff
        for b, i in ((0, 0), (0, 1), (1, 0), (2, 999), (1, 517)):
            want = _philox4x32_10((i, b, stage, 0), (seed & 0xffffffff, seed >> 32))[0]
            assert int(got[b, i]) == want, (seed, stage, b, i, hex(int(got[b, i])), hex(want))
    assert int(got[0, 0]) != 0x6627e8d5                                 # the second call used another seed and stage
