"""Round 6 (VERDICT r05 "missing" 1): TWO real ranks through one collective on the one GPU a test box has.  Two processes share
cuda:0, each with its own model replica and HIP engine, each running detector -> selection -> decode on its `shard_bounds` slice
of a 4-image batch; the collective runs on gloo with the payload staged through host memory (``gather_device`` - RCCL cannot
put two ranks on one device).  Until now collectives had only moved stub data between processes (tests/test_dist_cpu.py) or real
data at world 1 (tests/test_gpu_dist.py).  SURVEY.md 8(e); the reference's own multi-GPU code is DDP in train_full_model.py."""
import json
import os
import socket
import subprocess
import sys
import tempfile

import pytest
import torch

pytestmark = pytest.mark.gpu
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r"""
import json, os, sys, torch, torch.distributed as dist
sys.path.insert(0, %(repo)r); sys.path.insert(0, os.path.join(%(repo)r, "tests"))
from conftest import gpu_model
from rgrg_amd import synth
from rgrg_amd.dist import GradBuckets, generate_sharded, shard_bounds
mode, out_path = sys.argv[1], sys.argv[2]
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
cpu = torch.device("cpu")
res = {}
if mode == "generate":
    m = gpu_model("ragged")
    images = synth.make_images(4, 1234).to(dev)
    lo, hi = shard_bounds(4, rank, world)
    out = generate_sharded(m, images[lo:hi], 24, gather_device=cpu)
    out_u = generate_sharded(m, images[0:3] if rank == 0 else images[3:4], 24, equal_shards=False, gather_device=cpu)   # 3 + 1 images
    if rank == 0:
        ref = m.generate(images, max_length=24)
        parts = [m.generate(images[a:b], max_length=24) for a, b in (shard_bounds(4, 0, 2), shard_bounds(4, 1, 2))]
        L = max(p[0].shape[1] for p in parts)
        pad = lambda t: torch.nn.functional.pad(t, (0, L - t.shape[1]), value=50256)
        cat = (torch.cat([pad(p[0]) for p in parts]), torch.cat([p[1] for p in parts]),
               {k: torch.cat([p[2][k] for p in parts]) for k in ("top_region_boxes", "top_scores")}, torch.cat([p[3] for p in parts]))
        res["rows"] = int(out[0].shape[0])
        res["ids_equal_whole_batch"] = bool(torch.equal(out[0], ref[0]))
        res["masks_equal_whole_batch"] = bool(torch.equal(out[1], ref[1]) and torch.equal(out[3], ref[3]))
        res["boxes_max_diff_whole_batch"] = float((out[2]["top_region_boxes"] - ref[2]["top_region_boxes"]).abs().max())
        res["scores_max_diff_whole_batch"] = float((out[2]["top_scores"] - ref[2]["top_scores"]).abs().max())
        res["bitwise_equal_to_per_shard_results"] = bool(torch.equal(out[0], cat[0]) and torch.equal(out[1], cat[1]) and torch.equal(out[3], cat[3])
                                                         and torch.equal(out[2]["top_region_boxes"], cat[2]["top_region_boxes"])
                                                         and torch.equal(out[2]["top_scores"], cat[2]["top_scores"]))
        res["unequal_shards_ids_equal_whole_batch"] = bool(torch.equal(out_u[0], ref[0]) and torch.equal(out_u[1], ref[1]))
        res["devices"] = [str(out[0].device), str(out[2]["top_scores"].device)]
else:
    m = gpu_model("bench")
    m.pretrain_without_lm_model = False
    m.train()
    m.language_model.dropout_p = 0.0
    T = 16
    g = torch.Generator().manual_seed(77)
    images = synth.make_images(4, 4321).to(dev)
    ids = torch.randint(0, 50257, (4 * 29, T), generator=g).to(dev)
    am = torch.ones((4 * 29, T), dtype=torch.int64, device=dev)
    has = torch.ones((4, 29), dtype=torch.bool, device=dev)
    abn = (torch.rand((4, 29), generator=g) < 0.2).to(dev)
    params = m.trainable_parameters()
    buckets = GradBuckets(params)

    def grads_of(a, b):
        buckets.zero()
        out = m(images[a:b], None, ids[a * 29:b * 29].clone(), am[a * 29:b * 29], has[a:b], abn[a:b])
        out[3].backward()                       # the language-model loss: a mean over the shard's token rows
        return torch.cat([f.clone() for f in buckets.buckets]), [o.detach() for o in out[1:4]], int(m.engine().last_train_shape[0])
    lo, hi = shard_bounds(4, rank, world)
    mine, losses, rows_fed = grads_of(lo, hi)
    both = [torch.empty_like(mine, device="cpu") for _ in range(world)]
    dist.all_gather(both, mine.cpu())           # the expected result, collected on the side
    n = buckets.allreduce(average=True, via_host=True)
    torch.cuda.synchronize()
    got = torch.cat([f for f in buckets.buckets]).cpu()
    want = (both[0] + both[1]) / 2
    res_local = {"n_buckets": n, "equals_mean_of_the_two_ranks": bool(torch.equal(got, want)), "differs_from_own": bool(not torch.equal(got, mine.cpu())),
                 "rows_fed": rows_fed}
    if rank == 0:
        union, _, rows_u = grads_of(0, 4)       # one process, the whole batch
        union = union.cpu()
        res_local["all_regions_detected"] = bool(rows_u == 4 * 29 and rows_fed == 2 * 29)
        res_local["cosine_with_union_gradient"] = float(torch.nn.functional.cosine_similarity(got, union, dim=0))
        res_local["relative_error_vs_union_gradient"] = float((got - union).norm() / union.norm())
    res = res_local
dist.barrier()
if rank == 0:
    json.dump(res, open(out_path, "w"))
dist.destroy_process_group()
"""


def _run_two_ranks(mode):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    with tempfile.TemporaryDirectory() as tmp:
        out = os.path.join(tmp, "res.json")
        procs = []
        for rank in range(2):
            env = dict(os.environ, RANK=str(rank), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
            procs.append(subprocess.Popen([sys.executable, "-c", WORKER % {"repo": REPO}, mode, out], env=env, stdout=subprocess.PIPE,
                                          stderr=subprocess.PIPE, text=True))
        errs = []
        for p in procs:
            try:
                _, err = p.communicate(timeout=900)
            except subprocess.TimeoutExpired:
                for q in procs:
                    q.kill()
                raise
            errs.append((p.returncode, err[-3000:]))
        assert all(rc == 0 for rc, _ in errs), errs
        return json.load(open(out))


def test_two_ranks_on_one_gpu_generate_sharded_equals_single_process_generate():
    """4 images -> 2 + 2 (and 3 + 1 with ``equal_shards=False``).  Token ids and both masks are `torch.equal` to what ONE process
    returns for the whole batch.  Boxes / scores: bit-identical to the per-shard single-process results (the gather moves bits),
    and within 1e-4 of the whole-batch run - the detector's fp32 GEMMs pick their split-K factor from the row count, so a
    2-image and a 4-image launch add a pixel's products in different orders."""
    r = _run_two_ranks("generate")
    assert r["rows"] > 29 and r["ids_equal_whole_batch"] and r["masks_equal_whole_batch"], r
    assert r["bitwise_equal_to_per_shard_results"], r
    assert r["boxes_max_diff_whole_batch"] <= 1e-4 * 512 and r["scores_max_diff_whole_batch"] <= 1e-4, r
    assert r["unequal_shards_ids_equal_whole_batch"], r
    assert all(d.startswith("cuda") for d in r["devices"]), r


def test_two_ranks_on_one_gpu_grad_buckets_allreduce_of_real_gradients():
    """One training pass per rank on its 2-image shard (fp32, dropout off, language-model loss), then ``GradBuckets.allreduce``
    staged through the host on gloo: every bucket holds exactly (g_0 + g_1) / 2 of the two ranks' gradients (collected on the
    side with an all_gather), and - all 29 regions detected on every image and every token valid, so both shards weigh the
    same - that mean is the gradient ONE process computes on the 4-image union, up to fp32 summation order."""
    r = _run_two_ranks("train")
    assert r["n_buckets"] >= 2 and r["equals_mean_of_the_two_ranks"] and r["differs_from_own"], r
    if r["all_regions_detected"]:
        assert r["cosine_with_union_gradient"] >= 0.9999 and r["relative_error_vs_union_gradient"] <= 2e-3, r
