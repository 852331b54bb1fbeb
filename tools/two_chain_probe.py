"""Feasibility probe (round 3): do TWO independent decode chains on two HIP streams overlap on one MI355X?
The batch-1 decode step is a chain of ~122 dependent 5-9 us kernels (launch gap + cold-load latency + fp32 MFMA + tail:
HBM sits at 22 % of its roofline), and at batch 32 the attention (HBM bound) and the GEMMs (operand-delivery bound) take
turns.  Two decoders, each with half of the sequences, driven from two host threads (ctypes releases the GIL), against
one decoder with all of them."""
import os
import sys
import threading
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rgrg_amd  # noqa: E402
from rgrg_amd import synth  # noqa: E402


def make():
    m = rgrg_amd.ReportGenerationModel(pretrain_without_lm_model=True)
    m.load_state_dict(synth.make_state_dict(0, "bench"))
    m.to("cuda:0").eval()
    return m


def timed(fn, n=3):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


def main():
    m1, m2 = make(), make()
    g = torch.Generator().manual_seed(1)
    for S, bf16 in ((29, False), (923, True)):
        feats = torch.randn((S, 1024), generator=g).cuda()
        h = (S + 1) // 2
        fa, fb = feats[:h].contiguous(), feats[h:].contiguous()
        ctx = (lambda: torch.autocast("cuda", dtype=torch.bfloat16)) if bf16 else (lambda: torch.autocast("cuda", enabled=False))

        def run(m, f, out, i):
            with ctx():
                out[i] = m.language_model.generate(f, max_length=128)

        def single():
            o = [None]
            run(m1, feats, o, 0)

        def half_a():
            o = [None]
            run(m1, fa, o, 0)

        res = [None, None]

        def both():
            ta = threading.Thread(target=run, args=(m1, fa, res, 0))
            tb = threading.Thread(target=run, args=(m2, fb, res, 1))
            ta.start(); tb.start(); ta.join(); tb.join()

        t_single = timed(single)
        with ctx():
            ref = m1.language_model.generate(feats, max_length=128)
        t_half = timed(half_a)
        t_both = timed(both)
        same = torch.equal(torch.cat([res[0], res[1]]), ref) if not bf16 else None
        print(f"S={S} bf16={bf16}: one decoder {t_single:.1f} ms | half the rows alone {t_half:.1f} ms | two decoders concurrently "
              f"{t_both:.1f} ms  -> x{t_single / t_both:.2f}  ids identical: {same}", flush=True)


if __name__ == "__main__":
    main()
