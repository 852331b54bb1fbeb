"""The device-side bookkeeping of the detector's validation losses (rgrg_amd.engine.balanced_sample_mask,
select_training_samples_batched: batched tensor operations without host synchronisation) against the per-image oracle
(oracle/tv013.py: balanced_sample, select_training_samples) on CPU tensors.  The oracle draws with an injected permutation;
the batched code is driven by per-element keys built from the oracle's choice (chosen: 0, everything else: 1), which is
how the GPU tests give both sides the same draws."""
import torch

from oracle import tv013
from rgrg_amd.engine import balanced_sample_mask, select_training_samples_batched


def _perm(seed):
    g = torch.Generator().manual_seed(seed)
    return lambda n, tag: torch.randperm(n, generator=g)


def test_balanced_sample_mask_reproduces_the_oracles_choice_through_keys():
    g = torch.Generator().manual_seed(1)
    B, n = 4, 3000
    labels = torch.zeros((B, n))
    labels[0, torch.randperm(n, generator=g)[:40]] = 1.0          # fewer positives than the cap
    labels[1, torch.randperm(n, generator=g)[:500]] = 1.0         # more positives than the cap
    labels[2] = -1.0                                              # nothing but ignored and ...
    labels[2, :100] = 0.0                                         # ... fewer negatives than the batch
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
    got = balanced_sample_mask(labels, keys, 256, 0.5)
    assert torch.equal(got, want)
    assert (got[0] == 1).sum() == 40 and (got[0] == 2).sum() == 216 and (got[1] == 1).sum() == 128 and (got[1] == 2).sum() == 128
    assert (got[2] == 2).sum() == 100 and (got[2] == 1).sum() == 0 and (got[3] == 2).sum() == 256


def test_balanced_sample_mask_with_random_keys_is_a_valid_balanced_sample():
    g = torch.Generator().manual_seed(2)
    B, n = 3, 2000
    labels = (torch.rand((B, n), generator=g) < 0.2).float()
    labels[torch.rand((B, n), generator=g) < 0.1] = -1.0
    keys = torch.rand((B, n), generator=g)
    m = balanced_sample_mask(labels, keys, 512, 0.25)
    for i in range(B):
        pos, neg = labels[i] >= 1, labels[i] == 0
        assert bool(((m[i] == 1) <= pos).all()) and bool(((m[i] == 2) <= neg).all())
        k_pos = min(int(pos.sum()), 128)
        assert int((m[i] == 1).sum()) == k_pos and int((m[i] == 2).sum()) == min(int(neg.sum()), 512 - k_pos)
        # the sampled positives are exactly the ones with the smallest keys
        assert keys[i][m[i] == 1].max() <= keys[i][pos & (m[i] != 1)].min()
    # ties: lower index first
    t = balanced_sample_mask(torch.ones((1, 10)), torch.zeros((1, 10)), 8, 0.5)
    assert t.tolist() == [[1, 1, 1, 1, 0, 0, 0, 0, 0, 0]]


def _cpu_match(gt, gt_count):
    def match(boxes, box_count):
        B, N = boxes.shape[:2]
        out = torch.full((B, N), -1, dtype=torch.int32)
        for b in range(B):
            n, ng = int(box_count[b]), int(gt_count[b])
            if ng:
                out[b, :n] = tv013.matcher(tv013.box_iou(gt[b, :ng], boxes[b, :n]), 0.5, 0.5, False).to(torch.int32)
        return out
    return match


def test_select_training_samples_batched_equals_the_oracle_per_image():
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
    enc = lambda ref, pr: tv013.box_encode(ref, pr, (10.0, 10.0, 5.0, 5.0))  # noqa: E731
    props_s, offsets, labels_flat, reg = select_training_samples_batched(props, counts, gt, gcount, gl, keys, _cpu_match(gt, gcount), enc)
    ks = [int(p.shape[0]) for p in o_props]
    assert offsets.dtype == torch.int32 and offsets.tolist() == [0, ks[0], ks[0] + ks[1], sum(ks)]
    assert props_s.shape == (B, 512, 4)
    R = sum(ks)
    for b in range(B):
        assert torch.equal(props_s[b, :ks[b]], o_props[b]) and bool((props_s[b, ks[b]:] == 0).all())
    assert torch.equal(labels_flat[:R], torch.cat(o_labels))
    o = torch.cat(o_reg)
    pos = torch.cat(o_labels) > 0                                   # only the positives' targets are ever read by the loss
    assert torch.equal(reg[:R][pos], o[pos])
    fin = torch.isfinite(o).all(1)
    assert torch.equal(reg[:R][fin], o[fin])
