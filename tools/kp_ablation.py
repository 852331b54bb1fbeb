"""Where a slot of the K-parity ping-pong GEMM (csrc/gemm_kp.inc) spends its time: the kernel with its fragment reads (1), its
LDS-DMA requests (2) and its MFMAs (4) switched off in turn (RGRG_PP_DBG, read per launch), cold weights, HIP events around 20
launches.  Usage: python tools/kp_ablation.py [M]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rgrg_amd import _hip  # noqa: E402

CASES = [("mlp_proj", 1024, 4096, 7), ("c_fc", 4096, 1024, 12), ("c_attn", 3072, 1024, 12)]
NAMES = {6: "kp128x128x2", 7: "kp64x64x4", 8: "kp128x64x3", 9: "kp64x128x3", 10: "kp64x64x3", 12: "pr128x128x4"}


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


def main():
    M = int(sys.argv[1]) if len(sys.argv) > 1 else 923
    lib = _hip.load()
    st = torch.cuda.current_stream().cuda_stream
    for name, N, K, tile in CASES:
        ncopy = max(1, -(-600_000_000 // (N * K * 2)))
        A16 = (torch.rand((M, K), device="cuda") * 2 - 1).bfloat16().view(torch.int16)
        Wb = ((torch.rand((ncopy, N, K), device="cuda") * 2 - 1) / K ** 0.5).bfloat16().view(torch.int16)
        b = torch.randn((N,), device="cuda")
        Y = torch.zeros((M, N), device="cuda")
        it = [0]

        def call():
            it[0] += 1
            _hip.check(lib.rgrg_debug_linear_bf16_tile(A16.data_ptr(), Wb[it[0] % ncopy].data_ptr(), b.data_ptr(), None, Y.data_ptr(), M, N, K, N, 0, tile,
                                                       0, 0, 0, st))
        print(f"{name} M={M} N={N} K={K} {NAMES[tile]}", flush=True)
        for dbg, what in ((0, "full"), (1, "no fragment reads"), (2, "no DMA"), (4, "no MFMA"), (3, "MFMA + barriers only"), (6, "reads + barriers only"),
                          (5, "DMA + barriers only"), (7, "barriers only")):
            os.environ["RGRG_PP_DBG"] = str(dbg)
            print(f"   dbg {dbg} {what:24s} {timed(call):7.1f} us", flush=True)
        os.environ["RGRG_PP_DBG"] = "0"
    # the producers as the decode step launches them (K-parity kernel, LayerNorm-producer epilogue: fp32 residual in, fp32 x + 16-bit
    # copy + statistics slots out): the whole launch, the launch without its epilogue, and neither main loop nor epilogue
    for name, N, K in (("attn_proj", 1024, 1024), ("mlp_proj", 1024, 4096)):
        ncopy = max(1, -(-600_000_000 // (N * K * 2)))
        A16 = (torch.rand((M, K), device="cuda") * 2 - 1).bfloat16().view(torch.int16)
        Wb = ((torch.rand((ncopy, N, K), device="cuda") * 2 - 1) / K ** 0.5).bfloat16().view(torch.int16)
        b = torch.randn((N,), device="cuda")
        R = torch.randn((M, N), device="cuda")
        yb, so = torch.empty((M, N), dtype=torch.int16, device="cuda"), torch.zeros((M, 16, 2), device="cuda")
        it = [0]

        def call():
            it[0] += 1
            _hip.check(lib.rgrg_debug_linear_bf16_ln_kp(A16.data_ptr(), Wb[it[0] % ncopy].data_ptr(), b.data_ptr(), R.data_ptr(), R.data_ptr(), None,
                                                        yb.data_ptr(), so.data_ptr(), None, None, M, N, K, N, 0, 0, 1, st))
        print(f"{name} M={M} as a LayerNorm producer (kp64x64x4)", flush=True)
        for dbg, what in ((0, "full"), (8, "no epilogue"), (7, "barriers + epilogue only"), (15, "barriers only, no epilogue")):
            os.environ["RGRG_PP_DBG"] = str(dbg)
            print(f"   dbg {dbg:2d} {what:28s} {timed(call):7.1f} us", flush=True)
        os.environ["RGRG_PP_DBG"] = "0"


if __name__ == "__main__":
    main()
