"""Latency of a grid-wide barrier among the 256 resident workgroups of a fused decode kernel, per implementation variant
(rgrg_debug_grid_barrier in csrc/runtime.hip), and the workgroup -> XCD placement of back-to-back launches.
Usage: python tools/grid_barrier_bench.py [iters=200] [payload_floats=512]"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rgrg_amd import _hip  # noqa: E402

NAMES = {0: "one atomic counter + agent fences", 1: "one atomic counter, no fences", 2: "per-workgroup flags + agent fences",
         3: "flags, coherent payload, no fences", 4: "two-level counters + agent fences", 5: "fences alone (no sync)",
         6: "XCD-hierarchical, sc1 stores, 1 acquire/WG", 7: "XCD-hierarchical, sc1 stores + sc1 loads",
         8: "flat relaxed counter, sc1 stores + sc1 loads"}


def run(lib, v, iters, payload):
    us, stale = C.c_float(), (C.c_uint * 2)()
    _hip.check(lib.rgrg_debug_grid_barrier(v, iters, payload, C.byref(us), stale), "rgrg_debug_grid_barrier")
    return us.value, stale[0], stale[1]


def main():
    iters = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    payload = int(sys.argv[2]) if len(sys.argv) > 2 else 512
    lib = _hip.load()
    print(f"# 256 workgroups x 512 threads, {iters} rounds, {payload} floats published per workgroup and round")
    for v in range(9):
        us, st, to = run(lib, v, iters, payload)
        print(f"variant {v} {NAMES[v]:46s} {us:8.2f} us per round   stale reads {st}  timeout {to}", flush=True)
    us, st, to = run(lib, 5, iters, 0)
    print(f"(loop overhead without payload, fences alone: {us:.2f} us)")
    print("# round-4 variants: barrier alone (no payload), then with the hand-off, then the stale-read check under uneven load")
    for v in (6, 7, 8):
        us0, _, to0 = run(lib, v, iters, 0)
        us1, st1, to1 = run(lib, v, iters, payload)
        us2, st2, to2 = run(lib, v | 0x100, iters, payload)
        print(f"variant {v} {NAMES[v]:46s} barrier only {us0:6.2f} us | with hand-off {us1:6.2f} us, stale {st1} | uneven load: {us2:6.2f} us, "
              f"stale {st2} | timeouts {to0 + to1 + to2}", flush=True)
    # placement: XCC id of every workgroup of 3 back-to-back launches, for grids that are / are not multiples of 8
    for blocks in (256, 464, 29, 100):
        out = (C.c_int * (blocks * 3))()
        _hip.check(lib.rgrg_debug_xcc_map(blocks, 3, out), "rgrg_debug_xcc_map")
        rr = [sum(1 for b in range(blocks) if out[l * blocks + b] == b % 8) for l in range(3)]
        print(f"placement, {blocks} workgroups x 3 launches: workgroups with XCC == b % 8 per launch: {rr}; first 16 of launch 1: "
              f"{[out[blocks + b] for b in range(min(16, blocks))]}", flush=True)


if __name__ == "__main__":
    main()
