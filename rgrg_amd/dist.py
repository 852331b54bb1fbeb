"""Multi-GPU report generation: the path is embarrassingly data-parallel over images
(no cross-image op; BatchNorm uses running statistics), so each rank runs
``generate()`` on its own shard and the ONLY collective is a final gather of the token
ids and the small per-image outputs.  ``torch.distributed`` backend
"nccl" is RCCL over xGMI on ROCm; the CPU tests run the same code on gloo.
"""
from __future__ import annotations

from typing import Dict, Optional, Tuple, Union

import torch
import torch.distributed as dist

PAD_TOKEN_ID = 50256
NUM_REGIONS = 29

GenerateOutput = Union[int, Tuple[torch.Tensor, torch.Tensor, Dict[str, torch.Tensor], torch.Tensor]]


def shard_bounds(n_images: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous shard [lo, hi) of rank ``rank``: rank-order concatenation reproduces the
    single-process row order (row-major over (image, region))."""
    per, rem = divmod(n_images, world)
    lo = rank * per + min(rank, rem)
    return lo, lo + per + (1 if rank < rem else 0)


def generate_sharded(model, images_local: torch.Tensor, max_length: Optional[int], group: Optional[dist.ProcessGroup] = None,
                     equal_shards: bool = True, gather_device: Optional[torch.device] = None) -> GenerateOutput:
    """``generate()`` of the whole (rank-order concatenated) batch: every rank runs the three
    stages on its own image shard, then ONE all_gather assembles the single-process result.

    The collective has a fixed shape, so it needs the same number of images on every rank and a bound on the
    sequence length.  When the caller cannot promise the former (``equal_shards=False``: ``shard_bounds`` of a batch
    that does not divide by the world size) or gives no ``max_length`` (the reference's greedy search then runs until
    every row has emitted EOS), one extra 2-word all_reduce(MAX) agrees on the padded shard size and length first.

    ``gather_device``: where the collectives run (default: the images' device - RCCL).  ``torch.device("cpu")`` stages the
    payload through host memory for a process group whose backend cannot take device tensors (gloo: several ranks on one
    GPU, tests/test_gpu_dist2.py); the returned tensors live on the images' device either way."""
    _, detections, top_region_features, class_detected = model.object_detector(images_local)
    selected, feats = model.binary_classifier_region_selection(top_region_features, class_detected, return_loss=False)
    ids = model.language_model.generate(feats, max_length) if feats.shape[0] > 0 else None
    n_pad = images_local.shape[0]
    if max_length is None or not equal_shards:
        t = torch.tensor([0 if ids is None else ids.shape[1], n_pad], dtype=torch.int64, device=gather_device or images_local.device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
        if max_length is None:
            max_length = max(int(t[0]), 1)
        n_pad = int(t[1])
    return gather_generate_outputs(ids, selected, detections, class_detected, max_length, images_local.device, group, n_pad, gather_device)


def gather_generate_outputs(out_ids: Optional[torch.Tensor], sel: torch.Tensor, det: Dict[str, torch.Tensor],
                            cd: torch.Tensor, max_length: int, device: torch.device,
                            group: Optional[dist.ProcessGroup] = None, n_pad: Optional[int] = None,
                            gather_device: Optional[torch.device] = None) -> GenerateOutput:
    """All-gather the per-rank stage outputs (``out_ids`` is None when the rank selected no
    region) into what a single process would have returned for the concatenated batch.

    One fixed-shape all_gather: ids padded to [29*n_pad, max_length] int64 plus a packed
    per-image record (selected | detected | score bits | box bits | valid | L'), ~31 KB per image, so there
    is exactly one collective; ``n_pad`` >= the local image count pads short shards with invalid images that are
    dropped again after the gather.  The single-process L' is the longest row of the WHOLE batch:
    after the gather the ids are trimmed to the global maximum length.  ``gather_device``: see ``generate_sharded``."""
    world = dist.get_world_size(group)
    n_local_images = sel.shape[0]
    n_pad = n_local_images if n_pad is None else int(n_pad)
    assert n_pad >= n_local_images and max_length is not None and max_length >= 1
    ids = torch.full((NUM_REGIONS * n_pad, max_length), PAD_TOKEN_ID, dtype=torch.int64, device=device)
    meta = torch.zeros((n_pad, NUM_REGIONS * 7 + 2), dtype=torch.int64, device=device)
    n = n_local_images
    meta[:n, 0:29] = sel.to(torch.int64)
    meta[:n, 29:58] = cd.to(torch.int64)
    meta[:n, 58:87] = det["top_scores"].contiguous().view(torch.int32).to(torch.int64)
    meta[:n, 87:203] = det["top_region_boxes"].contiguous().view(n, -1).view(torch.int32).to(torch.int64)
    meta[:n, 203] = 1  # a real image of this rank
    if out_ids is not None:
        ids[: out_ids.shape[0], : out_ids.shape[1]] = out_ids
        meta[:, 204] = out_ids.shape[1]
    payload = torch.cat([ids.view(n_pad, -1), meta], dim=1).contiguous()
    if gather_device is not None and torch.device(gather_device) != payload.device:
        payload = payload.to(gather_device)
    gathered = [torch.empty_like(payload) for _ in range(world)]
    dist.all_gather(gathered, payload, group=group)
    allp = torch.cat(gathered, 0).to(device)
    ids_all = allp[:, : NUM_REGIONS * max_length].reshape(world, n_pad * NUM_REGIONS, max_length)
    meta_all = allp[:, NUM_REGIONS * max_length:]
    valid = meta_all[:, 203].bool()
    sel_pad = meta_all[:, 0:29].bool()
    # each rank's rows are compact (its selected regions first): re-compact over the whole batch - the first n_sel[r] rows of
    # every rank's block, in rank order, picked with ONE mask (one host sync, whatever the world size)
    n_sel = sel_pad.view(world, -1).sum(1)
    keep = torch.arange(n_pad * NUM_REGIONS, device=device)[None, :] < n_sel[:, None]
    rows = ids_all[keep]
    if rows.shape[0] == 0:
        return -1
    L = int(meta_all[:, 204].max())
    m = meta_all[valid]
    n_img = m.shape[0]
    scores = m[:, 58:87].to(torch.int32).view(torch.float32)
    boxes = m[:, 87:203].to(torch.int32).view(torch.float32).view(n_img, NUM_REGIONS, 4)
    return (rows[:, :L].contiguous(), m[:, 0:29].bool(), {"top_region_boxes": boxes, "top_scores": scores}, m[:, 29:58].bool())


class GradBuckets:
    """Persistent flat gradient buckets for the data-parallel training step (SURVEY.md 8(e) "Training"): the ``.grad``
    of every trainable parameter is a VIEW into one of a few flat fp32 buffers (autograd accumulates into the views in
    place), so the all-reduce runs directly on the buffers - no packing copy before and no unpacking copy after the
    collective (2 x 215 MB per step saved against torch.cat + copy_).  Use ``zero()`` (or
    ``optimizer.zero_grad(set_to_none=False)``) between steps so that the views survive."""

    def __init__(self, params, bucket_bytes: int = 64 << 20):
        self.params = [p for p in params if p.requires_grad]
        self.buckets = []
        cur, cur_bytes = [], 0
        groups = []
        for p in self.params:
            nbytes = p.numel() * p.element_size()
            if cur and (cur_bytes + nbytes > bucket_bytes or p.device != cur[0].device or p.dtype != cur[0].dtype):
                groups.append(cur)
                cur, cur_bytes = [], 0
            cur.append(p)
            cur_bytes += nbytes
        if cur:
            groups.append(cur)
        for grp in groups:
            flat = torch.zeros(sum(p.numel() for p in grp), dtype=grp[0].dtype, device=grp[0].device)
            off = 0
            for p in grp:
                n = p.numel()
                view = flat[off:off + n].view_as(p)
                if p.grad is not None:
                    view.copy_(p.grad)
                p.grad = view
                off += n
            self.buckets.append(flat)

        self._owner = {}            # id(param) -> bucket index
        for bi, grp in enumerate(groups):
            for p in grp:
                self._owner[id(p)] = bi
        self._sizes = [len(grp) for grp in groups]
        self._hooks = None
        self._pending = None
        self._works = None
        self._group = None
        self._order = list(reversed(range(len(self.buckets))))   # autograd reaches the LAST parameters first (DDP's bucket order)
        self.launch_log = []        # bucket indices in the order their reduction was issued during the last step

    def zero(self) -> None:
        for flat in self.buckets:
            flat.zero_()

    # ---- reductions issued from the backward pass (opt-in): ``arm()`` once, then per step ``begin_step()`` ... backward ...
    # ``finish()``.  A post-accumulate hook on every parameter counts its bucket down; a bucket's asynchronous all-reduce is
    # issued as soon as the bucket AND every bucket in front of it in the fixed launch order (last bucket first) are complete,
    # so every rank issues the same collectives in the same order whatever its own backward looked like (a rank without
    # selected regions produces no language-model gradients: its buckets are issued by finish()).  What this overlaps on the
    # RGRG training step: the classifier heads' bucket with the language model's backward; the 200 MB of d(uk / uv) come out
    # of the LAST kernel of that backward (rgrg_decoder_lm_loss_grad is one call), so nothing is left to hide them behind.
    def arm(self, group=None) -> None:
        if self._hooks is not None:
            return
        self._group = group
        self._hooks = [p.register_post_accumulate_grad_hook(self._on_grad) for p in self.params]

    def disarm(self) -> None:
        for h in self._hooks or []:
            h.remove()
        self._hooks = None

    def begin_step(self) -> None:
        """One backward pass per ``begin_step()``: a second one (gradient accumulation) would find its buckets reduced already."""
        if not self.owns_all_grads():   # before any collective of the step is issued: raising later would strand the other ranks
            raise RuntimeError("GradBuckets: a .grad no longer points into the flat buckets (use zero() / zero_grad(set_to_none=False))")
        self._pending = list(self._sizes)
        self._works = {}
        self._next = 0
        self.launch_log = []

    def _issue_ready(self, force: bool = False) -> None:
        while self._next < len(self._order):
            bi = self._order[self._next]
            if self._pending[bi] > 0 and not force:
                return
            self._works[bi] = dist.all_reduce(self.buckets[bi], op=dist.ReduceOp.SUM, group=self._group, async_op=True)
            self.launch_log.append(bi)
            self._next += 1

    def _on_grad(self, p) -> None:
        if self._pending is None or not (dist.is_available() and dist.is_initialized()):
            return
        bi = self._owner[id(p)]
        self._pending[bi] -= 1
        if self._pending[bi] < 0:
            raise RuntimeError("GradBuckets: a second backward pass after begin_step() - this bucket's all-reduce has been issued already; "
                               "accumulate micro-steps before begin_step(), or call it once per backward")
        if self._pending[bi] == 0:
            self._issue_ready()

    def finish(self, average: bool = True) -> int:
        """Issue what the hooks could not (buckets with parameters that received no gradient this step), wait, average."""
        if not (dist.is_available() and dist.is_initialized()):
            return 0
        if self._pending is None:
            raise RuntimeError("GradBuckets.finish() without begin_step()")
        owned = self.owns_all_grads()
        self._issue_ready(force=True)   # every rank issues every bucket, whatever happened on this one
        world = dist.get_world_size(self._group)
        for bi, w in self._works.items():
            w.wait()
            if average and world > 1:
                self.buckets[bi].div_(world)
        self._pending = None
        if not owned:   # after the collectives: the other ranks are not left waiting for this one
            raise RuntimeError("GradBuckets: a .grad stopped pointing into the flat buckets during the step (zero_grad(set_to_none=True)?)")
        return len(self._works)

    def owns_all_grads(self) -> bool:
        """False when something replaced a ``.grad`` (e.g. ``zero_grad(set_to_none=True)``)."""
        spans = [(f.data_ptr(), f.data_ptr() + f.numel() * f.element_size()) for f in self.buckets]
        return all(p.grad is not None and any(lo <= p.grad.data_ptr() < hi for lo, hi in spans) for p in self.params)

    def allreduce(self, group=None, average: bool = True, via_host: bool = False) -> int:
        """Sum (average) the buckets over the ranks with asynchronous all-reduces (RCCL over xGMI: a ring moves
        2(N-1)/N x 215 MB per rank per step; large buckets keep the per-link ring efficient).  The HIP backward is one
        fused call per module, so there is nothing to overlap the reduction with except the other buckets.
        ``via_host``: stage every bucket through host memory (a process group that cannot take device tensors - gloo with
        several ranks on one GPU, tests/test_gpu_dist2.py)."""
        if not (dist.is_available() and dist.is_initialized()):
            return 0
        if not self.owns_all_grads():
            raise RuntimeError("GradBuckets: a .grad no longer points into the flat buckets (use zero() / zero_grad(set_to_none=False))")
        world = dist.get_world_size(group)
        bufs = [flat.cpu() for flat in self.buckets] if via_host else self.buckets
        works = [dist.all_reduce(b, op=dist.ReduceOp.SUM, group=group, async_op=True) for b in bufs]
        for w, b, flat in zip(works, bufs, self.buckets):
            w.wait()
            if via_host:
                flat.copy_(b)
            if average and world > 1:
                flat.div_(world)
        return len(self.buckets)


def allreduce_gradients(params, group=None, bucket_bytes: int = 64 << 20, average: bool = True) -> int:
    """Gradient synchronisation for parameters whose gradients are ordinary separate tensors: packed into a few flat
    buckets, all-reduced, unpacked (two extra passes over the gradients - prefer ``GradBuckets``, whose buckets ARE the
    gradients).  Every rank holds a full replica and a shard of the batch; divided by the world size when ``average``
    (the DDP convention).  Returns the number of buckets."""
    if not (dist.is_available() and dist.is_initialized()):
        return 0
    world = dist.get_world_size(group)
    grads = [p.grad for p in params if p.grad is not None]
    if not grads:
        return 0
    buckets, cur, cur_bytes = [], [], 0
    for g in grads:
        nbytes = g.numel() * g.element_size()
        if cur and cur_bytes + nbytes > bucket_bytes:
            buckets.append(cur)
            cur, cur_bytes = [], 0
        cur.append(g)
        cur_bytes += nbytes
    if cur:
        buckets.append(cur)
    pending = []
    for b in buckets:
        flat = torch.cat([g.reshape(-1) for g in b])
        pending.append((dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group, async_op=True), flat, b))
    for work, flat, b in pending:
        work.wait()
        if average and world > 1:
            flat.div_(world)
        off = 0
        for g in b:
            n = g.numel()
            g.copy_(flat[off:off + n].view_as(g))
            off += n
    return len(buckets)
