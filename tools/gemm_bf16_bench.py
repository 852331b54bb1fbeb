"""Time the bf16 GEMMs on the decoder's many-sequence shapes (BASELINE configs[2]: M = 923 token rows): the LDS-DMA kernel
(both operands bf16) with its tile shapes x LDS stage counts, optional row-pitch padding, and the register-staged kernel
(fp32 activations) for reference.  HIP events around 20 back-to-back launches on the current
stream, operands uniform random.
Usage: python tools/gemm_bf16_bench.py [M] [--shapes c_fc,lm_head] [--tiles 1,2] [--stages 2,4] [--pads 0,64]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rgrg_amd import _hip  # noqa: E402

SHAPES = [("c_attn", 3072, 1024, 0, False), ("attn_proj", 1024, 1024, 0, True), ("c_fc", 4096, 1024, 2, False),
          ("mlp_proj", 1024, 4096, 0, True), ("lm_head", 50257, 1024, 0, False)]
SHAPE_NAMES = {0: "auto", 1: "128x128", 2: "64x64", 3: "128x64", 4: "64x128", 5: "pp256x256", 6: "kp128x128x2", 7: "kp64x64x4", 8: "kp128x64x3",
               9: "kp64x128x3", 10: "kp64x64x3", 11: "kp-auto", 12: "pr128x128x4", 13: "kp64x64x2"}


def timed(call, n=20):
    for _ in range(3):
        call()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        call()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n


def padded(t, pad):
    """[rows, K] int16 -> a view with row pitch K + pad (elements) inside a larger allocation."""
    rows, K = t.shape
    buf = torch.zeros((rows, K + pad), dtype=t.dtype, device=t.device)
    buf[:, :K] = t
    return buf


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("M", nargs="?", type=int, default=923)
    ap.add_argument("--shapes", default=",".join(s[0] for s in SHAPES))
    ap.add_argument("--tiles", default="0,1,2,3,4", help="shapes: 0 auto, 1 128x128, 2 64x64, 3 128x64, 4 64x128, 5 256x256 ping-pong, 6-11 K-parity ping-pong (gemm_kp.inc)")
    ap.add_argument("--stages", default="2,3,4")
    ap.add_argument("--pads", default="0")
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--vendor", action="store_true",
                    help="also time torch.mm on the same bf16 operands (the vendor library GEMM, bf16 out, no bias / residual): "
                         "a yardstick for what these shapes reach on this GPU, not a code path of the package")
    ap.add_argument("--cold", action="store_true",
                    help="cycle through enough weight copies (> 600 MB) that no launch finds its W in the 256 MB Infinity Cache: the "
                         "regime of a decode step, which streams 0.7 GB of weights between two uses of the same matrix")
    ap.add_argument("--copies-mb", type=int, default=600,
                    help="--cold: total size of the weight copies cycled through (600: past the 256 MB Infinity Cache; ~150: past the 32 MB of L2 "
                         "but inside the Infinity Cache - what a weight prefetch into it would give)")
    args = ap.parse_args()
    M = args.M
    lib = _hip.load()
    st = torch.cuda.current_stream().cuda_stream
    for name, N, K, act, res in SHAPES:
        if name not in args.shapes.split(","):
            continue
        A = torch.rand((M, K), device="cuda") * 2 - 1
        W = (torch.rand((N, K), device="cuda") * 2 - 1) / K ** 0.5
        A16 = torch.empty((M, K), dtype=torch.int16, device="cuda")
        ncopy = max(1, -(-args.copies_mb * 1_000_000 // (N * K * 2))) if args.cold else 1
        Wb = torch.empty((ncopy, N, K), dtype=torch.int16, device="cuda")
        b = torch.randn((N,), device="cuda")
        Y = torch.zeros((M, N), device="cuda")
        _hip.check(lib.rgrg_f32_to_bf16(A.data_ptr(), A16.data_ptr(), M * K, 0, st))
        for c in range(ncopy):
            _hip.check(lib.rgrg_f32_to_bf16(W.data_ptr(), Wb[c].data_ptr(), N * K, 0, st))
        it = [0]

        def wptr(Wx):  # next weight copy
            it[0] += 1
            return Wx[it[0] % ncopy].data_ptr()
        R = Y.data_ptr() if res else None
        ref = (A16.view(torch.bfloat16).float() @ Wb[0].view(torch.bfloat16).float().t() + b)
        print(f"{name:9s} M={M} N={N} K={K}" + (f"  (cold: {ncopy} weight copies)" if args.cold else ""), flush=True)
        us = timed(lambda: _hip.check(lib.rgrg_linear_bf16w_f32(A.data_ptr(), wptr(Wb), b.data_ptr(), R, Y.data_ptr(), M, N, K, N, act, 0, st)), args.iters)
        print(f"   reg-staged (fp32 A)          {us:7.1f} us {2.0 * M * N * K / us / 1e6:5.0f} TF/s", flush=True)
        if args.vendor:
            Ab, Wv, Yb = A16.view(torch.bfloat16), Wb.view(torch.bfloat16), torch.empty((M, N), dtype=torch.bfloat16, device="cuda")

            def vendor():
                it[0] += 1
                torch.mm(Ab, Wv[it[0] % ncopy].t(), out=Yb)
            us = timed(vendor, args.iters)
            print(f"   vendor torch.mm (bf16 out)   {us:7.1f} us {2.0 * M * N * K / us / 1e6:5.0f} TF/s", flush=True)
        for pad in map(int, args.pads.split(",")):
            Ap = padded(A16, pad) if pad else A16
            Wp = torch.stack([padded(Wb[c], pad) for c in range(ncopy)]) if pad else Wb
            for shape in map(int, args.tiles.split(",")):
                for nst in ([0] if shape == 0 or shape >= 5 else list(map(int, args.stages.split(",")))):
                    tile = shape + 16 * nst

                    def call(out=Y, r=R, a=act):
                        _hip.check(lib.rgrg_debug_linear_bf16_tile(Ap.data_ptr(), wptr(Wp), b.data_ptr(), r, out.data_ptr(), M, N, K, N, a, tile,
                                                                   K + pad, K + pad, 0, st))
                    us = timed(call, args.iters)
                    Y2 = torch.empty((M, N), device="cuda")
                    it[0] = -1
                    call(Y2, None, 0)
                    err = (Y2 - ref).abs().max().item()
                    print(f"   {SHAPE_NAMES[shape]:11s} stages {nst} pad {pad:3d}  {us:7.1f} us {2.0 * M * N * K / us / 1e6:5.0f} TF/s  (err {err:.1e})",
                          flush=True)

if __name__ == "__main__":
    main()
