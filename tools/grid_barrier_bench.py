"""Latency of a grid-wide barrier among the 256 resident workgroups of a fused decode kernel, per implementation variant
(rgrg_debug_grid_barrier in csrc/runtime.hip).  Usage: python tools/grid_barrier_bench.py [iters=200] [payload_floats=512]"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rgrg_amd import _hip  # noqa: E402

NAMES = {0: "one atomic counter + agent fences", 1: "one atomic counter, no fences", 2: "per-workgroup flags + agent fences",
         3: "flags, coherent payload, no fences", 4: "two-level counters + agent fences", 5: "fences alone (no sync)"}


def main():
    iters = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    payload = int(sys.argv[2]) if len(sys.argv) > 2 else 512
    lib = _hip.load()
    for v in range(6):
        us, stale = C.c_float(), (C.c_uint * 2)()
        _hip.check(lib.rgrg_debug_grid_barrier(v, iters, payload, C.byref(us), stale), "rgrg_debug_grid_barrier")
        print(f"variant {v} {NAMES[v]:40s} {us.value:8.2f} us per round   stale reads {stale[0]}  timeout {stale[1]}", flush=True)
    us, stale = C.c_float(), (C.c_uint * 2)()
    _hip.check(lib.rgrg_debug_grid_barrier(5, iters, 0, C.byref(us), stale), "rgrg_debug_grid_barrier")
    print(f"(loop overhead without payload, fences alone: {us.value:.2f} us)")


if __name__ == "__main__":
    main()
