"""Eval-mode forward() timing at the shape of BASELINE configs[4] (8 images per GPU): the teacher-forced
language-model pass over S = 29*B sentences of T tokens, and the whole ReportGenerationModel.forward.
Usage: python tools/forward_bench.py [B] [T]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rgrg_amd  # noqa: E402
from rgrg_amd import synth  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
T = int(sys.argv[2]) if len(sys.argv) > 2 else 64
dev = torch.device("cuda", 0)
m = rgrg_amd.ReportGenerationModel(pretrain_without_lm_model=False)
m.load_state_dict(synth.make_state_dict(0, "bench"))
m.to(dev).eval()
g = torch.Generator().manual_seed(0)
S = 29 * B
ids = torch.randint(0, 50257, (S, T), generator=g).to(dev)
lens = torch.randint(T // 2, T + 1, (S,), generator=g)
am = (torch.arange(T)[None, :] < lens[:, None]).to(torch.int64).to(dev)
feats = torch.randn((S, 1024), generator=g).to(dev)
images = synth.make_images(B, 1234).to(dev)
has = torch.ones((B, 29), dtype=torch.bool, device=dev)
abn = torch.zeros((B, 29), dtype=torch.bool, device=dev)
FLOP_PER_TOKEN = 2 * 353.453e6  # MACs per token incl. lm_head


def timed(fn, n=3):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n


for name, ctx in (("fp32", None), ("bf16 (autocast)", torch.bfloat16)):
    def lm():
        if ctx is None:
            return m.language_model(ids.clone(), am, feats, return_loss=True)
        with torch.autocast("cuda", dtype=ctx):
            return m.language_model(ids.clone(), am, feats, return_loss=True)

    def full():
        if ctx is None:
            return m(images, None, ids.clone(), am, has, abn, return_loss=True)
        with torch.autocast("cuda", dtype=ctx):
            return m(images, None, ids.clone(), am, has, abn, return_loss=True)

    t_lm = timed(lm)
    loss = lm().item()
    t_full = timed(full)
    tok = S * T
    print(f"{name:16s} B={B} S={S} T={T}: LM teacher-forced pass {t_lm * 1e3:8.1f} ms = {tok / t_lm / 1e3:8.1f} k tokens/s "
          f"= {tok * FLOP_PER_TOKEN / t_lm / 1e12:6.1f} TFLOP/s (loss {loss:.4f}); forward() {t_full * 1e3:8.1f} ms "
          f"= {B / t_full:6.1f} images/s", flush=True)
