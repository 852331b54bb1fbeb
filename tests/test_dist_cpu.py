"""world_size-2 test of the multi-GPU path on CPU (gloo): image sharding + the single
all_gather of token ids / per-image outputs reproduce the single-process result
(row order, global L', -1 sentinel)."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from rgrg_amd.dist import gather_generate_outputs, shard_bounds

MAXLEN = 24


def _fake_generate(n_images, seed, none_selected=False, length=17):
    """Deterministic stand-in for generate() on a shard (the HIP path needs a GPU)."""
    g = torch.Generator().manual_seed(seed)
    cd = torch.rand((n_images, 29), generator=g) > 0.2
    sel = cd & (torch.rand((n_images, 29), generator=g) > 0.4)
    det = {"top_scores": torch.rand((n_images, 29), generator=g), "top_region_boxes": torch.rand((n_images, 29, 4), generator=g) * 512}
    if none_selected:
        return None, torch.zeros_like(sel), det, cd
    S = int(sel.sum())
    ids = torch.randint(0, 50000, (S, length), generator=g)
    ids[:, 0] = 50256
    return ids, sel, det, cd


def _worker(rank, world, port, case, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        if case == "plain":
            local = _fake_generate(2, 100 + rank, length=17 if rank == 0 else 21)
        elif case == "one_rank_empty":
            local = _fake_generate(2, 100 + rank, none_selected=(rank == 1))
        else:
            local = _fake_generate(2, 100 + rank, none_selected=True)
        out = gather_generate_outputs(*local, MAXLEN, torch.device("cpu"))
        if rank == 0:
            ret["out"] = out if isinstance(out, int) else (out[0], out[1], out[2]["top_scores"], out[2]["top_region_boxes"], out[3])
    finally:
        dist.destroy_process_group()


def _run(case):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(2, port, case, ret), nprocs=2, join=True)
    return ret["out"]


def test_shard_bounds_cover_the_batch_in_order():
    for n, w in ((256, 8), (10, 4), (3, 8), (32, 1)):
        spans = [shard_bounds(n, r, w) for r in range(w)]
        assert spans[0][0] == 0 and spans[-1][1] == n
        assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
        assert max(hi - lo for lo, hi in spans) - min(hi - lo for lo, hi in spans) <= 1


def test_gather_reproduces_single_process_result():
    ids, sel, scores, boxes, cd = _run("plain")
    i0, s0, d0, c0 = _fake_generate(2, 100, length=17)
    i1, s1, d1, c1 = _fake_generate(2, 101, length=21)
    assert torch.equal(sel, torch.cat([s0, s1])) and torch.equal(cd, torch.cat([c0, c1]))
    assert torch.equal(scores, torch.cat([d0["top_scores"], d1["top_scores"]]))       # floats travel bit-exactly
    assert torch.equal(boxes, torch.cat([d0["top_region_boxes"], d1["top_region_boxes"]]))
    assert ids.shape == (int(sel.sum()), 21)                                         # global L' = longest shard
    pad0 = torch.nn.functional.pad(i0, (0, 4), value=50256)
    assert torch.equal(ids, torch.cat([pad0, i1]))                                  # rank-order rows, PAD-extended


def test_gather_with_an_empty_rank_and_all_empty():
    ids, sel, _, boxes, cd = _run("one_rank_empty")
    i0, s0, _, _ = _fake_generate(2, 100)
    _, _, d1, c1 = _fake_generate(2, 101, none_selected=True)
    assert torch.equal(ids, i0) and int(sel[2:].sum()) == 0 and torch.equal(sel[:2], s0)
    assert torch.equal(cd[2:], c1) and torch.equal(boxes[2:], d1["top_region_boxes"])  # detections of the empty rank survive
    assert _run("all_empty") == -1


# ------------------------------------------------------------------------- training: gradient all-reduce (gloo, world 2)
def _grad_worker(rank, world, port, ret):
    from rgrg_amd.dist import allreduce_gradients
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        g = torch.Generator().manual_seed(7)
        shapes = [(300, 40), (40,), (1000, 64), (64,), (5,)]
        params = [torch.nn.Parameter(torch.zeros(s)) for s in shapes]
        for i, p in enumerate(params):
            base = torch.randn(p.shape, generator=g)        # same stream on both ranks
            p.grad = base * (rank + 1) if i != 3 else None   # a parameter without gradient is skipped
        n = allreduce_gradients(params, bucket_bytes=100_000)  # 48 KB + 256 KB + ... -> several buckets
        if rank == 0:
            ret["buckets"] = n
            ret["grads"] = [None if p.grad is None else p.grad.clone() for p in params]
    finally:
        dist.destroy_process_group()


def test_allreduce_gradients_averages_over_ranks_in_buckets():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_grad_worker, args=(2, port, ret), nprocs=2, join=True)
    assert ret["buckets"] >= 2
    g = torch.Generator().manual_seed(7)
    for i, s in enumerate([(300, 40), (40,), (1000, 64), (64,), (5,)]):
        base = torch.randn(s, generator=g)
        if i == 3:
            assert ret["grads"][i] is None
        else:
            assert torch.allclose(ret["grads"][i], base * 1.5, atol=1e-6)   # (1 + 2) / 2


# ------------------------------------------------------------------------- unequal shards / unbounded length (gloo, world 2)
class _FakeModel:
    """The three stage calls generate_sharded makes, with deterministic per-rank outputs (the HIP stages need a GPU)."""

    def __init__(self, seed, length):
        self.seed, self.length, self.language_model = seed, length, self
        self.ids = None

    def object_detector(self, images):
        self.ids, sel, det, cd = _fake_generate(images.shape[0], self.seed, length=self.length)
        self.sel = sel
        return {}, det, torch.zeros((images.shape[0], 29, 1024)), cd

    def binary_classifier_region_selection(self, feats, cd, return_loss=False):
        return self.sel, torch.zeros((int(self.sel.sum()), 1024))

    def generate(self, feats, max_length):
        return self.ids if max_length is None else self.ids[:, :max_length]


def _sharded_worker(rank, world, port, ret):
    from rgrg_amd.dist import generate_sharded
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        lo, hi = shard_bounds(5, rank, world)                 # 3 + 2 images: unequal shards
        model = _FakeModel(200 + rank, 13 if rank == 0 else 19)
        out = generate_sharded(model, torch.zeros((hi - lo, 1, 8, 8)), None, equal_shards=False)  # and no max_length
        if rank == 0:
            ret["out"] = (out[0], out[1], out[2]["top_scores"], out[2]["top_region_boxes"], out[3])
    finally:
        dist.destroy_process_group()


def test_generate_sharded_pads_unequal_shards_and_resolves_unbounded_length():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_sharded_worker, args=(2, port, ret), nprocs=2, join=True)
    ids, sel, scores, boxes, cd = ret["out"]
    i0, s0, d0, c0 = _fake_generate(3, 200, length=13)
    i1, s1, d1, c1 = _fake_generate(2, 201, length=19)
    assert sel.shape == (5, 29) and torch.equal(sel, torch.cat([s0, s1])) and torch.equal(cd, torch.cat([c0, c1]))
    assert torch.equal(boxes, torch.cat([d0["top_region_boxes"], d1["top_region_boxes"]]))
    assert torch.equal(scores, torch.cat([d0["top_scores"], d1["top_scores"]]))
    assert ids.shape == (int(sel.sum()), 19)                                         # L' agreed by the all_reduce(MAX)
    assert torch.equal(ids, torch.cat([torch.nn.functional.pad(i0, (0, 6), value=50256), i1]))


def _bucket_worker(rank, world, port, ret):
    from rgrg_amd.dist import GradBuckets
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        g = torch.Generator().manual_seed(11)
        params = [torch.nn.Parameter(torch.randn(s, generator=g)) for s in [(300, 40), (40,), (1000, 64), (5,)]]
        frozen = torch.nn.Parameter(torch.zeros(7), requires_grad=False)
        gb = GradBuckets(params + [frozen], bucket_bytes=100_000)
        ptrs = [p.grad.data_ptr() for p in params]
        for step in range(2):                                  # autograd accumulates INTO the views, twice
            loss = sum(((rank + 1.0) * (i + 1) * p).sum() for i, p in enumerate(params))
            loss.backward()
        assert [p.grad.data_ptr() for p in params] == ptrs and gb.owns_all_grads()
        n = gb.allreduce()
        # replicas draw different dropout masks: the per-pass seed is offset by the rank (rank 0 keeps the plain counter)
        from types import SimpleNamespace
        from rgrg_amd.language_model import LanguageModel
        ret[f"seed{rank}"] = LanguageModel.pass_dropout_seed(SimpleNamespace(dropout_seed=0x5EED0001))
        if rank == 0:
            ret["n"] = n
            ret["grads"] = [p.grad.clone() for p in params]
            gb.zero()
            ret["zeroed"] = all(float(p.grad.abs().max()) == 0.0 for p in params) and frozen.grad is None
            params[0].grad = None
            ret["detects_lost_view"] = not gb.owns_all_grads()
    finally:
        dist.destroy_process_group()


def test_grad_buckets_allreduce_in_place_on_flat_views():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_bucket_worker, args=(2, port, ret), nprocs=2, join=True)
    assert ret["n"] >= 2 and ret["zeroed"] and ret["detects_lost_view"]
    assert ret["seed0"] == 0x5EED0001 and ret["seed1"] == 0x5EED0001 + (1 << 40)
    for i, gr in enumerate(ret["grads"]):                      # 2 backward passes x mean over ranks of (rank + 1)(i + 1)
        assert torch.allclose(gr, torch.full_like(gr, 2 * 1.5 * (i + 1)), atol=1e-5)


def test_bench_self_spawns_n_ranks_when_no_launcher_is_present():
    """`python bench.py --gpus N` (no torchrun environment) must start by itself.  The launcher
    path - re-execution under torch.distributed.run on 127.0.0.1 with a free port, barrier-bracketed timing, the MAX
    over ranks, generate_sharded's gather and the ONE JSON line from rank 0 - runs here on gloo x 2 with the script's
    CPU stub model (--stub-cpu: no GPU work, never a measurement)."""
    import json
    import os
    import subprocess
    import sys
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR")}
    res = subprocess.run([sys.executable, os.path.join(repo, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
                          "--batch", "3", "--stub-cpu"], env=env, capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stderr[-2000:]
    lines = [l for l in res.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, res.stdout                 # exactly one JSON line, from rank 0
    r = json.loads(lines[0])
    assert r["n_gpus"] == 2 and r["steps"] == 2 and r["warmup"] == 1 and r["scaling"] == "weak"
    assert r["config"]["global_batch"] == 6 and r["config"]["regions_generated"] == 6 * 29
    assert r["value"] > 0 and abs(r["value"] - 6 * 2 / (r["ms_per_step"] * 2e-3)) < 1e-6 * r["value"]


# ------------------------------------------------------------------------- world 8 at the shape of BASELINE configs[3] (gloo)
class _FakeModel8(_FakeModel):
    """Rank 5 selects nothing (its generate() is never called: the reference returns -1 there, generate_sharded must still
    take part in the gather); every rank stops at its own length."""

    def __init__(self, rank):
        super().__init__(300 + rank, 9 + 2 * rank)
        self.none = rank == 5

    def object_detector(self, images):
        self.ids, self.sel, det, cd = _fake_generate(images.shape[0], self.seed, none_selected=self.none, length=self.length)
        return {}, det, torch.zeros((images.shape[0], 29, 1024)), cd


def _world8_worker(rank, world, port, ret):
    from rgrg_amd.dist import generate_sharded
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        lo, hi = shard_bounds(256, rank, world)
        assert hi - lo == 32
        out = generate_sharded(_FakeModel8(rank), torch.zeros((hi - lo, 1, 8, 8)), MAXLEN)
        if rank == 0:
            ret["out"] = (out[0], out[1], out[2]["top_scores"], out[2]["top_region_boxes"], out[3])
    finally:
        dist.destroy_process_group()


def test_generate_sharded_world8_at_the_configs3_shape():
    """configs[3] (256 images over 8 ranks, 32 per rank, ONE gather of token ids) has no hardware run; the
    code path runs here at its real world size and batch on gloo with stub stages: rank-order concatenation of 8 shards, a rank
    without any selected region, eight different early-exit lengths (global L' = the longest), per-image records of every rank."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_world8_worker, args=(8, port, ret), nprocs=8, join=True)
    ids, sel, scores, boxes, cd = ret["out"]
    parts = [_fake_generate(32, 300 + r, none_selected=(r == 5), length=9 + 2 * r) for r in range(8)]
    assert sel.shape == (256, 29) and torch.equal(sel, torch.cat([p[1] for p in parts]))
    assert torch.equal(cd, torch.cat([p[3] for p in parts]))
    assert torch.equal(scores, torch.cat([p[2]["top_scores"] for p in parts]))
    assert torch.equal(boxes, torch.cat([p[2]["top_region_boxes"] for p in parts]))
    L = 9 + 2 * 7
    want = torch.cat([torch.nn.functional.pad(p[0], (0, L - p[0].shape[1]), value=50256) for p in parts if p[0] is not None])
    assert int(sel[5 * 32:6 * 32].sum()) == 0 and ids.shape == (int(sel.sum()), L) and torch.equal(ids, want)


def test_bench_launcher_with_eight_ranks_on_the_stub():
    """`python bench.py --gpus 8` as the driver's scaling run would start it (self-spawned here), 32 images per rank: one JSON
    line from rank 0 with the whole-job aggregate and the configs[3] leg (256 images, one gather) in it."""
    import json
    import subprocess
    import sys
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR")}
    res = subprocess.run([sys.executable, os.path.join(repo, "bench.py"), "--gpus", "8", "--steps", "2", "--warmup", "1",
                          "--batch", "32", "--stub-cpu"], env=env, capture_output=True, text=True, timeout=900)
    assert res.returncode == 0, res.stderr[-2000:]
    lines = [l for l in res.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, res.stdout
    r = json.loads(lines[0])
    assert r["n_gpus"] == 8 and r["scaling"] == "weak" and r["config"]["global_batch"] == 256
    assert r["config"]["regions_generated"] == 256 * 29
    assert r["value"] > 0 and abs(r["value"] - 256 * 2 / (r["ms_per_step"] * 2e-3)) < 1e-6 * r["value"]


# ------------------------------------------------------------------------- reductions issued from the backward pass (gloo, world 2)
def _hooked_bucket_worker(rank, world, port, ret):
    from rgrg_amd.dist import GradBuckets
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        params = [torch.nn.Parameter(torch.zeros(s)) for s in [(300, 40), (40,), (1000, 64), (64,), (7, 9)]]
        gb = GradBuckets(params, bucket_bytes=60000)          # buckets: [p0, p1] [p2] [p3, p4]
        sizes = [b.numel() for b in gb.buckets]
        gb.arm()
        log = {}
        for case in ("natural", "reversed", "missing"):
            gb.zero()
            gb.begin_step()
            scale = float(rank + 1)
            if case == "natural":       # parameters used in order: the backward reaches the LAST bucket first
                order = [0, 1, 2, 3, 4]
            elif case == "reversed":    # used in reverse: bucket 0 completes first and must WAIT for its turn
                order = [4, 3, 2, 1, 0]
            else:                       # rank 1 never touches p4 (a rank without selected regions): its last bucket never completes
                order = [0, 1, 2, 3, 4] if rank == 0 else [0, 1, 2, 3]
            loss = sum(((i + 1) * scale * params[i]).sum() for i in order)
            loss.backward()
            issued_in_backward = list(gb.launch_log)
            n = gb.finish()
            log[case] = (issued_in_backward, list(gb.launch_log), n, [p.grad.clone() for p in params])
        if rank in (0, 1):
            ret[rank] = (sizes, log)
    finally:
        dist.destroy_process_group()


def test_grad_buckets_issue_their_allreduce_from_backward_hooks_in_one_fixed_order():
    """a bucket's all-reduce is issued from the backward pass as soon as the bucket and every bucket in front
    of it in the fixed launch order (last bucket first) are complete - the same order of collectives on every rank, whatever
    order a rank's own backward produces its gradients in and even when a rank produces none for some parameter (then
    finish() issues the rest)."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_hooked_bucket_worker, args=(2, port, ret), nprocs=2, join=True)
    sizes, log0 = ret[0]
    _, log1 = ret[1]
    assert len(sizes) == 3
    for case in ("natural", "reversed", "missing"):
        for log in (log0, log1):
            assert log[case][1] == [2, 1, 0] and log[case][2] == 3            # one order, every bucket, on every rank
    assert log0["natural"][0] == [2, 1, 0] and log1["natural"][0] == [2, 1, 0]  # all issued while the backward was running
    assert log0["reversed"][0] == [2, 1, 0]                                    # ... bucket 0 was complete first and waited
    assert log0["missing"][0] == [2, 1, 0] and log1["missing"][0] == []        # rank 1: nothing before finish()
    for case, p4_ranks in (("natural", 2), ("reversed", 2), ("missing", 1)):
        for log in (log0, log1):
            for i, g in enumerate(log[case][3]):
                want = (i + 1) * 1.5 if (i < 4 or p4_ranks == 2) else (i + 1) * 0.5   # mean over ranks of (rank + 1)(i + 1)
                assert torch.allclose(g, torch.full_like(g, want), atol=1e-6), (case, i)
