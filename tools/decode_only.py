"""The decoder alone on S region-feature rows (no detector, no selection): `LanguageModel.generate` with the bench weights,
for profiling the decode kernels in isolation (rocprofv3 --pmc of the full batch-32 bench crashes the profiler on this image).
Usage: python tools/decode_only.py [S=923] [--dtype bf16|f32] [--max-length 128] [--runs 1]
Prints one JSON line: rows, tokens per row, ms per generate call."""
import argparse
import contextlib
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rgrg_amd  # noqa: E402
from rgrg_amd import synth  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("S", nargs="?", type=int, default=923)
    ap.add_argument("--dtype", choices=("f32", "bf16"), default="bf16")
    ap.add_argument("--max-length", type=int, default=128)
    ap.add_argument("--runs", type=int, default=1)
    ap.add_argument("--warmup", type=int, default=0)
    args = ap.parse_args()
    model = rgrg_amd.ReportGenerationModel(pretrain_without_lm_model=True)
    model.load_state_dict(synth.make_state_dict(0, "bench"))
    model.to("cuda:0").eval()
    g = torch.Generator().manual_seed(99)
    feats = torch.randn((args.S, 1024), generator=g).to("cuda:0")
    ctx = torch.autocast("cuda", dtype=torch.bfloat16) if args.dtype == "bf16" else contextlib.nullcontext()
    ids = None
    with ctx:
        for _ in range(args.warmup):
            ids = model.language_model.generate(feats, max_length=args.max_length)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.runs):
            ids = model.language_model.generate(feats, max_length=args.max_length)
        torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) * 1e3 / args.runs
    print(json.dumps({"rows": args.S, "dtype": args.dtype, "tokens_per_row": int(ids.shape[1]), "ms_per_generate": ms}))


if __name__ == "__main__":
    main()
