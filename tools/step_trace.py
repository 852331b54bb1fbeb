"""True timeline of one many-sequence decode step under concurrency: `rgrg_decoder_trace_step` enqueues the step eagerly with an
event behind every launch on the stream it went to (rocprofv3's kernel trace serialises the queues, so it cannot show how the
row-range chains of decoder.hip run_row_ranges overlap).  Per row range and kernel kind: the mean time from the previous event of
the same chain to this one (= the launch's duration under contention, boundary included), the chain's period per layer, and the
step's total.  Also times whole generate() calls (hipGraph replays).
Usage: python tools/step_trace.py [S=923] [--nkeys 65] [--dtype bf16] [--full]"""
import argparse
import contextlib
import ctypes as C
import json
import os
import sys
import time
from collections import defaultdict

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rgrg_amd  # noqa: E402
from rgrg_amd import _hip, synth  # noqa: E402

KIND = {0: "c_attn", 1: "attention", 2: "attn_proj", 3: "c_fc", 4: "mlp_proj"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("S", nargs="?", type=int, default=923)
    ap.add_argument("--nkeys", type=int, default=65)
    ap.add_argument("--dtype", choices=("f16", "bf16"), default="bf16")
    ap.add_argument("--full", action="store_true")
    ap.add_argument("--parts", action="store_true", help="also rgrg_decoder_time_step_parts (the bench line's GEMM / attention family timings)")
    ap.add_argument("--generate", type=int, default=2, help="timed generate() calls of 128 tokens (0: none)")
    args = ap.parse_args()
    model = rgrg_amd.ReportGenerationModel(pretrain_without_lm_model=True)
    model.load_state_dict(synth.make_state_dict(0, "bench"))
    model.to("cuda:0").eval()
    g = torch.Generator().manual_seed(99)
    feats = torch.randn((args.S, 1024), generator=g).to("cuda:0")
    ctx = torch.autocast("cuda", dtype=torch.bfloat16 if args.dtype == "bf16" else torch.float16)
    with ctx:
        model.language_model.generate(feats, max_length=128)
        torch.cuda.synchronize()
        if args.generate:
            t0 = time.perf_counter()
            for _ in range(args.generate):
                model.language_model.generate(feats, max_length=128)
            torch.cuda.synchronize()
            ms = (time.perf_counter() - t0) * 1e3 / args.generate
            print(json.dumps({"rows": args.S, "ms_per_generate": ms, "ms_per_step": ms / 127}), flush=True)
    eng = model.language_model.engine()
    lib = _hip.load()
    if args.parts:
        p = eng.time_step_parts(args.S, args.nkeys, iters=10)
        print(json.dumps({"parts": {k: (round(v, 4) if isinstance(v, float) else v) for k, v in p.items()},
                          "gemm_tflops": p["gemm_flops"] / (p["ms_gemm"] * 1e-3) / 1e12}), flush=True)
    maxr = 4096
    recs = (C.c_float * (3 * maxr))()
    n = C.c_int(0)
    _hip.check(lib.rgrg_decoder_trace_step(eng._decoder, args.S, args.nkeys, 3, recs, maxr, C.byref(n)), "trace")
    rows = [(int(recs[3 * i]), int(recs[3 * i + 1]), recs[3 * i + 2] * 1e3) for i in range(n.value)]   # (r0, tag, us)
    chains = defaultdict(list)
    for r0, tag, us in rows:
        if tag < 1002:
            chains[r0].append((tag, us))
    end = max(us for _, _, us in rows)
    print(f"eager step at {args.nkeys} keys: {len(rows)} launches, {end:.1f} us from the first launch to the arg-max's end")
    for r0, lst in sorted(chains.items()):
        per = defaultdict(list)
        prev = 0.0
        for tag, us in lst:
            per[KIND.get(tag % 8, "other") if tag < 1000 else {1000: "embedding", 1001: "ln_f"}[tag]].append(us - prev)
            prev = us
        tot = lst[-1][1]
        print(f"  chain of rows {r0}..: ends at {tot:.1f} us ({tot / 24:.1f} us per layer); mean us per launch slot: " +
              ", ".join(f"{k} {sum(v) / len(v):.1f}" for k, v in per.items()))
        if args.full:
            prev = 0.0
            for tag, us in lst[:30]:
                print(f"      +{us:8.1f}  {us - prev:6.1f}  {KIND.get(tag % 8) if tag < 1000 else tag} L{tag // 8 if tag < 1000 else ''}")
                prev = us
    tail = [(tag, us) for r0, tag, us in rows if tag >= 1002]
    last_chain = max(lst[-1][1] for lst in chains.values())
    print("  join .. lm_head .. arg-max: " + ", ".join(f"{ {1002: 'lm_head', 1003: 'arg-max'}[t]} ends {u:.1f}" for t, u in tail) +
          f" (last chain ended at {last_chain:.1f})")


if __name__ == "__main__":
    main()
