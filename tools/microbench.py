"""GPU micro-measurements that steer the decode design: cost of a dependent kernel chain
(eager vs hipGraph), decode with/without graph, per-GEMM timings."""
import ctypes as C
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rgrg_amd import _hip  # noqa: E402

lib = _hip.load()
for blocks in (1, 256):
    for mode, name in ((0, "eager/private stream"), (1, "hipGraph replay"), (2, "eager/null stream")):
        for n in (200, 2000):
            us = C.c_float()
            _hip.check(lib.rgrg_debug_chain(n, mode, blocks, C.byref(us)))
            print(f"chain blocks={blocks:4d} n={n:5d} {name:22s}: {us.value:6.2f} us/kernel", flush=True)

if "--decode" in sys.argv:
    import rgrg_amd
    from rgrg_amd import synth
    m = rgrg_amd.ReportGenerationModel(True)
    m.load_state_dict(synth.make_state_dict(0, "bench"))
    m.to("cuda:0").eval()
    eng = m.engine()
    g = torch.Generator().manual_seed(1)
    feats = torch.randn((29, 1024), generator=g).cuda()
    for use_graph in (True, False, True, False):
        eng.greedy_decode(feats, 16, use_graph)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        eng.greedy_decode(feats, 128, use_graph)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        print(f"decode S=29 L=128 graph={use_graph}: {dt*1e3:.1f} ms  ({dt/127*1e6:.0f} us/step)", flush=True)
    print("step parts:", eng.time_step_parts(29, 65, 3))
