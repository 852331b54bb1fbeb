"""Time the 16-bit GEMM kernels on the LARGE shapes (round 5): the training step of BASELINE configs[4] (M = 14 848 token rows:
forward and activation-gradient GEMMs of a GPT-2 block, lm_head and its dgrad) and fc6 of the batch-32 detector
(26 586 x 131 072 x 1024).  Tiles: 1 = 128 x 128 (2 LDS stages), 5 = the 256 x 256 ping-pong kernel, 0 = the launcher's choice;
RGRG_GEMM_GM=-1 in the environment gives the round-4 column-major tile order.  HIP events around back-to-back launches on
the current stream, operands uniform random 16 bit; a few output rows are checked against a float64 product.
Usage: python tools/gemm_big_bench.py [--shapes c_fc,fc6] [--tiles 0,33,5] [--iters 10] [--vendor]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rgrg_amd import _hip  # noqa: E402

M_TRAIN = 14848
SHAPES = {  # name: (M, N, K, act, 16-bit out)
    "c_attn": (M_TRAIN, 3072, 1024, 0, False), "attn_proj": (M_TRAIN, 1024, 1024, 0, False), "c_fc": (M_TRAIN, 4096, 1024, 2, True),
    "mlp_proj": (M_TRAIN, 1024, 4096, 0, False), "c_attn_T": (M_TRAIN, 1024, 3072, 0, False), "mlp_proj_T": (M_TRAIN, 4096, 1024, 0, True),
    "lm_head": (M_TRAIN, 50257, 1024, 0, False), "lm_head_T": (M_TRAIN, 1024, 50432, 0, False),
    "fc6": (26586, 1024, 131072, 1, False), "fc6_b8": (6650, 1024, 131072, 1, False), "lm_head_923": (923, 50257, 1024, 0, False),
}


def timed(call, n):
    for _ in range(2):
        call()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        call()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n


def rand16(shape, scale, g):
    """uniform [-scale, scale) bf16, generated in slabs (fc6's A is 7 GB)."""
    out = torch.empty(shape, dtype=torch.bfloat16, device="cuda")
    rows = max(1, (1 << 28) // shape[1])
    for r0 in range(0, shape[0], rows):
        out[r0:r0 + rows] = ((torch.rand((min(rows, shape[0] - r0), shape[1]), device="cuda", generator=g) * 2 - 1) * scale).to(torch.bfloat16)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--shapes", default="c_attn,attn_proj,c_fc,mlp_proj,c_attn_T,mlp_proj_T,lm_head,lm_head_T")
    ap.add_argument("--tiles", default="33,5", help="tile codes of rgrg_debug_linear_bf16_train (33 = 128x128 with 2 stages, 5 = ping-pong)")
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--vendor", action="store_true", help="also time torch.mm on the same operands (yardstick only)")
    args = ap.parse_args()
    lib = _hip.load()
    st = torch.cuda.current_stream().cuda_stream
    g = torch.Generator(device="cuda").manual_seed(0)
    for name in args.shapes.split(","):
        M, N, K, act, out16 = SHAPES[name]
        A = rand16((M, K), 1.0, g)
        W = rand16((N, K), K ** -0.5, g)
        b = torch.randn((N,), device="cuda")
        Y = torch.empty((M, N), device="cuda", dtype=torch.int16 if out16 else torch.float32)
        rows = torch.tensor([0, 1, M // 3, M // 2 + 17, M - 2, M - 1], device="cuda")
        ref = A[rows].double() @ W.double().t() + b.double()
        if act == 1:
            ref = ref.clamp_min(0)
        if act == 2:
            ref = torch.nn.functional.gelu(ref, approximate="tanh")
        print(f"{name:11s} M={M} N={N} K={K} {'16-bit out' if out16 else 'fp32 out'}", flush=True)
        flop = 2.0 * M * N * K
        if args.vendor:
            Yv = torch.empty((M, N), dtype=torch.bfloat16, device="cuda")
            us = timed(lambda: torch.mm(A, W.t(), out=Yv), args.iters)
            print(f"   vendor torch.mm (bf16 out, no epilogue) {us:9.1f} us {flop / us / 1e6:6.0f} TF/s", flush=True)
        for tile in map(int, args.tiles.split(",")):
            def call():
                _hip.check(lib.rgrg_debug_linear_bf16_train(A.data_ptr(), W.data_ptr(), b.data_ptr(), None, None if out16 else Y.data_ptr(),
                                                            Y.data_ptr() if out16 else None, None, None, M, N, K, N, act, tile, 0, st))
            us = timed(call, args.iters)
            got = (Y[rows].view(torch.bfloat16) if out16 else Y[rows]).double()
            err = ((got - ref).abs().max() / ref.abs().max()).item()
            print(f"   tile {tile:3d}  {us:9.1f} us {flop / us / 1e6:6.0f} TF/s  (rel err of 6 rows {err:.1e})", flush=True)
        del A, W, Y


if __name__ == "__main__":
    main()
