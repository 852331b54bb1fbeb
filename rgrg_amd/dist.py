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


def generate_sharded(model, images_local: torch.Tensor, max_length: Optional[int], group: Optional[dist.ProcessGroup] = None
                     ) -> GenerateOutput:
    """``generate()`` of the whole (rank-order concatenated) batch: every rank runs the three
    stages on its own image shard, then one all_gather assembles the single-process result.
    Equal shard sizes are required (fixed-shape collective)."""
    _, detections, top_region_features, class_detected = model.object_detector(images_local)
    selected, feats = model.binary_classifier_region_selection(top_region_features, class_detected, return_loss=False)
    ids = model.language_model.generate(feats, max_length) if feats.shape[0] > 0 else None
    return gather_generate_outputs(ids, selected, detections, class_detected, max_length, images_local.device, group)


def gather_generate_outputs(out_ids: Optional[torch.Tensor], sel: torch.Tensor, det: Dict[str, torch.Tensor],
                            cd: torch.Tensor, max_length: int, device: torch.device,
                            group: Optional[dist.ProcessGroup] = None) -> GenerateOutput:
    """All-gather the per-rank stage outputs (``out_ids`` is None when the rank selected no
    region) into what a single process would have returned for the concatenated batch.

    One fixed-shape all_gather: ids padded to [29*n_local, max_length] int64 plus a packed
    per-image record (selected | detected | score bits | box bits), ~31 KB per image, so there
    is exactly one collective.  The single-process L' is the longest row of the WHOLE batch:
    after the gather the ids are trimmed to the global maximum length."""
    world = dist.get_world_size(group)
    n_local_images = sel.shape[0]
    rows = NUM_REGIONS * n_local_images
    ids = torch.full((rows, max_length), PAD_TOKEN_ID, dtype=torch.int64, device=device)
    meta = torch.zeros((n_local_images, NUM_REGIONS * 7 + 1), dtype=torch.int64, device=device)
    meta[:, 0:29] = sel.to(torch.int64)
    meta[:, 29:58] = cd.to(torch.int64)
    meta[:, 58:87] = det["top_scores"].contiguous().view(torch.int32).to(torch.int64)
    meta[:, 87:203] = det["top_region_boxes"].contiguous().view(n_local_images, -1).view(torch.int32).to(torch.int64)
    if out_ids is not None:
        ids[: out_ids.shape[0], : out_ids.shape[1]] = out_ids
        meta[:, -1] = out_ids.shape[1]
    payload = torch.cat([ids.view(n_local_images, -1), meta], dim=1).contiguous()
    gathered = [torch.empty_like(payload) for _ in range(world)]
    dist.all_gather(gathered, payload, group=group)
    allp = torch.cat(gathered, 0)
    n_img = allp.shape[0]
    ids_all = allp[:, : NUM_REGIONS * max_length].reshape(n_img, NUM_REGIONS, max_length)
    meta_all = allp[:, NUM_REGIONS * max_length:]
    sel_all = meta_all[:, 0:29].bool()
    cd_all = meta_all[:, 29:58].bool()
    scores = meta_all[:, 58:87].to(torch.int32).view(torch.float32)
    boxes = meta_all[:, 87:203].to(torch.int32).view(torch.float32).view(n_img, NUM_REGIONS, 4)
    if int(sel_all.sum()) == 0:
        return -1
    # each rank's rows are compact (its selected regions first): re-compact over the whole batch
    out_rows = []
    for r in range(world):
        blk = slice(r * n_local_images, (r + 1) * n_local_images)
        n_sel = int(sel_all[blk].sum())
        out_rows.append(ids_all[blk].reshape(-1, max_length)[:n_sel])
    L = int(meta_all[:, -1].max())
    return torch.cat(out_rows, 0)[:, :L].contiguous(), sel_all, {"top_region_boxes": boxes, "top_scores": scores}, cd_all


def allreduce_gradients(params, group=None, bucket_bytes: int = 64 << 20, average: bool = True) -> int:
    """Gradient synchronisation of the frozen-detector training step (SURVEY.md 8(e) "Training"): every rank holds a
    full replica and a shard of the batch; after ``backward()`` the gradients of the 53.7 M trainable values (215 MB
    fp32) are packed into a few flat buckets, summed with asynchronous all-reduces (RCCL over xGMI; a ring moves
    2(N-1)/N x 215 MB per rank per step) and unpacked, divided by the world size when ``average`` (the DDP
    convention).  The HIP backward is one fused call per module, so there is nothing to overlap the reduction with
    except the other buckets; large buckets keep the per-link ring efficient.  Returns the number of buckets."""
    if not (dist.is_available() and dist.is_initialized()):
        return 0
    world = dist.get_world_size(group)
    grads = [p.grad for p in params if p.grad is not None]
    if world == 1 or not grads:
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
        if average:
            flat.div_(world)
        off = 0
        for g in b:
            n = g.numel()
            g.copy_(flat[off:off + n].view_as(g))
            off += n
    return len(buckets)
