"""Time the bf16-weight GEMM (rgrg_linear_bf16w_f32) on the decoder's many-sequence shapes, one subprocess per
tile configuration (RGRG_BF16_TILE).  Usage: python tools/gemm_bf16_bench.py [M]"""
import os
import subprocess
import sys

SHAPES = [("c_attn", 3072, 1024, 0, False), ("attn_proj", 1024, 1024, 0, True), ("c_fc", 4096, 1024, 2, False),
          ("mlp_proj", 1024, 4096, 0, True), ("lm_head", 50257, 1024, 0, False)]
CFG = {0: "auto", 1: "128x128/512", 5: "64x64/256"}


def child(M):
    import torch
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from rgrg_amd import _hip
    lib = _hip.load()
    st = torch.cuda.current_stream().cuda_stream
    out = []
    for name, N, K, act, res in SHAPES:
        A = torch.randn((M, K), device="cuda")
        W = torch.randn((N, K), device="cuda") / K ** 0.5
        Wb = torch.empty((N, K), dtype=torch.int16, device="cuda")
        b = torch.randn((N,), device="cuda")
        Y = torch.zeros((M, N), device="cuda")
        _hip.check(lib.rgrg_f32_to_bf16(W.data_ptr(), Wb.data_ptr(), N * K, st))
        call = lambda: _hip.check(lib.rgrg_linear_bf16w_f32(A.data_ptr(), Wb.data_ptr(), b.data_ptr(),
                                                             Y.data_ptr() if res else None, Y.data_ptr(), M, N, K, N, act, st))
        for _ in range(3):
            call()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        n = 20
        e0.record()
        for _ in range(n):
            call()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / n
        out.append(f"{name} {us:7.1f} us {2.0 * M * N * K / us / 1e6:6.0f} TF/s")
    print(f"cfg {CFG[int(os.environ.get('RGRG_BF16_TILE', '0'))]:12s} M={M}: " + " | ".join(out), flush=True)


if __name__ == "__main__":
    if os.environ.get("_GEMM_BENCH_CHILD"):
        child(int(sys.argv[1]))
    else:
        M = sys.argv[1] if len(sys.argv) > 1 else "928"
        for cfg in CFG:
            env = dict(os.environ, RGRG_BF16_TILE=str(cfg), _GEMM_BENCH_CHILD="1")
            subprocess.run([sys.executable, os.path.abspath(__file__), M], env=env, check=False)
