"""The RCCL path on the one GPU a test box has: a world_size-1 "nccl" process group in-process.
The real composition detector -> selection -> decode -> all_gather must return exactly what generate() returns, and the
gradient all-reduce over the flat buckets must be the identity."""
import socket

import pytest
import torch
import torch.distributed as dist

from conftest import gpu_model
from rgrg_amd import synth
from rgrg_amd.dist import GradBuckets, allreduce_gradients, generate_sharded

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda", 0)


@pytest.fixture(scope="module")
def rccl_world1():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1, device_id=DEV)
    yield
    dist.destroy_process_group()


def _same(a, b):
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]) and torch.equal(a[3], b[3])
    assert torch.equal(a[2]["top_region_boxes"], b[2]["top_region_boxes"]) and torch.equal(a[2]["top_scores"], b[2]["top_scores"])


def test_generate_sharded_over_rccl_equals_generate(rccl_world1):
    m = gpu_model("ragged")
    images = synth.make_images(2, 1234).to(DEV)
    ref = m.generate(images, max_length=24)
    _same(generate_sharded(m, images, 24), ref)                        # one all_gather
    _same(generate_sharded(m, images, 24, equal_shards=False), ref)    # + the 2-word all_reduce(MAX)
    # bit patterns of the float outputs survive the int64 packing; dtypes are the single-process ones
    out = generate_sharded(m, images, 24)
    assert out[0].dtype == torch.int64 and out[1].dtype == torch.bool and out[2]["top_scores"].dtype == torch.float32


def test_gradient_allreduce_over_rccl_is_the_identity_at_world1(rccl_world1):
    g = torch.Generator().manual_seed(3)
    params = [torch.nn.Parameter(torch.randn(s, generator=g).to(DEV)) for s in [(1024, 1024), (1024,), (49152, 64), (7,)]]
    gb = GradBuckets(params, bucket_bytes=4 << 20)
    sum((p * p).sum() for p in params).backward()
    want = [2 * p.detach().clone() for p in params]
    ptrs = [p.grad.data_ptr() for p in params]
    assert gb.allreduce() == len(gb.buckets) >= 2
    torch.cuda.synchronize()
    assert [p.grad.data_ptr() for p in params] == ptrs                 # reduced in place, views intact
    for p, w in zip(params, want):
        assert torch.equal(p.grad, w)
    loose = [torch.nn.Parameter(torch.zeros(5, device=DEV)) for _ in range(3)]
    for i, p in enumerate(loose):
        p.grad = torch.full((5,), float(i + 1), device=DEV)
    assert allreduce_gradients(loose) == 1
    assert [float(p.grad[0]) for p in loose] == [1.0, 2.0, 3.0]
