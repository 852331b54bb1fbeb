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


def gather_generate_outputs(local: GenerateOutput, n_local_images: int, max_length: int, device: torch.device,
                            group: Optional[dist.ProcessGroup] = None) -> GenerateOutput:
    """All-gather the per-rank ``generate()`` results into the result a single process
    would have produced for the concatenated batch.

    One fixed-shape all_gather: ids padded to [29*n_local, max_length] int64 plus a packed
    per-image record, so there is exactly one collective (payload ~ 30 KB/image).  The
    single-process L' is the longest row of the WHOLE batch: after the gather the ids are
    trimmed to the global maximum length so shapes stay bit-identical."""
    world = dist.get_world_size(group)
    rows = NUM_REGIONS * n_local_images
    ids = torch.full((rows, max_length), PAD_TOKEN_ID, dtype=torch.int64, device=device)
    # per image: 29 selected | 29 detected | 29 scores (bits) | 116 boxes (bits) -> int64 for a single dtype
    meta = torch.zeros((n_local_images, NUM_REGIONS * 7 + 2), dtype=torch.int64, device=device)
    if isinstance(local, int):
        meta[:, -1] = 1  # "this rank returned -1"
    else:
        out_ids, sel, det, cd = local
        ids[: out_ids.shape[0], : out_ids.shape[1]] = out_ids
        meta[:, 0:29] = sel.to(torch.int64)
        meta[:, 29:58] = cd.to(torch.int64)
        meta[:, 58:87] = det["top_scores"].contiguous().view(torch.int32).to(torch.int64)
        meta[:, 87:203] = det["top_region_boxes"].contiguous().view(n_local_images, -1).view(torch.int32).to(torch.int64)
        meta[:, -2] = out_ids.shape[1]
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
    # rows of each rank are compact (selected regions first): re-compact over the whole batch
    keep = torch.zeros((n_img, NUM_REGIONS), dtype=torch.bool, device=device)
    per_rank_images = n_img // world
    out_rows = []
    for r in range(world):
        blk = slice(r * per_rank_images, (r + 1) * per_rank_images)
        n_sel = int(sel_all[blk].sum())
        flat = ids_all[blk].reshape(-1, max_length)
        out_rows.append(flat[:n_sel])
    ids_cat = torch.cat(out_rows, 0)
    L = int(meta_all[:, -2].max())
    del keep
    return ids_cat[:, :L].contiguous(), sel_all, {"top_region_boxes": boxes, "top_scores": scores}, cd_all
