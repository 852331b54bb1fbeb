"""The kernels of the many-sequence 16-bit decode step (BASELINE configs[2]) as a micro-bench for rocprofv3's counter mode, which
does not survive the real decode process on this image (profiles/r05_rocprofv3_pmc_crash_reproducer.md): the four per-layer
projections in the variants and on the kernels the decoder launches (consumer / producer of the folded LayerNorm; LDS-DMA kernel or
K-parity kernel as `decoder.hip linear()` picks them), the lm_head with its arg-max epilogue, at M = 923 rows (the roofline's
one-range launches) and at M = 231 (a row range of the 4-range step), weights cycled through > 600 MB of copies so that no
launch finds its W in the Infinity Cache - and, with --attention, the step's attention launches at 65 keys on the real cache.
Prints a JSON manifest (case -> kernel name pattern, launch grid in threads, algorithmic operand bytes per launch) that
tools/pmc_decode_summary.py joins with the counter CSVs.
Usage (under rocprofv3 --pmc FETCH_SIZE | WRITE_SIZE):  python tools/pmc_decode_kernels.py [--attention] [--manifest out.json]"""
import argparse
import json
import os
import sys

import torch

os.environ.setdefault("RGRG_DECODE_CHAINS", "1")   # --attention: one range, no forked streams under the profiler
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rgrg_amd import _hip  # noqa: E402

KP_MODE = int(os.environ.get("RGRG_GEMM_KP", "2"))   # the decoder's default: producers on the K-parity kernel


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--attention", action="store_true")
    ap.add_argument("--manifest", default=None)
    ap.add_argument("--launches", type=int, default=24)
    args = ap.parse_args()
    lib = _hip.load()
    st = torch.cuda.current_stream().cuda_stream
    manifest = []
    if args.attention:
        import rgrg_amd
        from rgrg_amd import synth
        model = rgrg_amd.ReportGenerationModel(pretrain_without_lm_model=True)
        model.load_state_dict(synth.make_state_dict(0, "bench"))
        model.to("cuda:0").eval()
        S = 923
        feats = torch.randn((S, 1024), generator=torch.Generator().manual_seed(99)).to("cuda:0")
        # the decoder's state (923 sequences, 16-bit cache) from a 2-step EAGER one-range decode: rocprofv3's counter mode segfaults
        # on the hipGraph replays / forked streams of the real generate() (profiles/r05_rocprofv3_pmc_crash_reproducer.md)
        eng = model.language_model.engine()
        eng._get_decoder(S, 128)                              # cache slots for 128 tokens, like the bench's generate()
        eng.greedy_decode(feats, 3, use_graph=False, bf16=1)
        torch.cuda.synchronize()
        _hip.check(lib.rgrg_decoder_attention_only(eng._decoder, S, 65, 2, st))   # 48 launches at 65 keys
        torch.cuda.synchronize()
        manifest.append({"case": "attention_S923_65keys", "kernel": "attn_decode_kv16_wave_kernel", "threads": S * 16 // 4 * 256, "last_launches": 48,
                         "algorithmic_bytes": 2 * S * 1024 * 65 * 2 + S * 3072 * 4 + S * 1024 * 2 + 2 * S * 1024 * 2,
                         "what": "K/V cache rows of 65 keys (bf16) + q|k|v fp32 read + 16-bit output + the new key / value written"})
    else:
        g = torch.Generator().manual_seed(1)
        for M in (923, 231):
            for name, N, K, kind in (("c_attn", 3072, 1024, "cons"), ("attn_proj", 1024, 1024, "prod"), ("c_fc", 4096, 1024, "cons16"),
                                     ("mlp_proj", 1024, 4096, "prod")):
                ncopy = max(1, -(-600_000_000 // (N * K * 2)))
                A16 = torch.randn((M, K), generator=g).bfloat16().view(torch.int16).cuda()
                Wb = (torch.randn((ncopy, N, K), generator=g) / K ** 0.5).bfloat16().view(torch.int16).cuda()
                b = torch.randn((N,), generator=g).cuda()
                kp = 1 if ((kind == "prod" and KP_MODE in (1, 2)) or (kind != "prod" and KP_MODE in (1, 3))) else 0
                if kind == "prod":
                    R = torch.randn((M, N), generator=g).cuda()
                    yb, so = torch.empty((M, N), dtype=torch.int16, device="cuda"), torch.zeros((M, 16, 2), device="cuda")
                    call = lambda i: _hip.check(lib.rgrg_debug_linear_bf16_ln_kp(A16.data_ptr(), Wb[i % ncopy].data_ptr(), b.data_ptr(), R.data_ptr(),  # noqa: E731
                                                                                 R.data_ptr(), None, yb.data_ptr(), so.data_ptr(), None, None, M, N, K, N, 0, 0, kp, st))
                    alg = M * K * 2 + N * K * 2 + M * N * 4 * 2 + M * N * 2 + M * 16 * 8
                    what = "A + W (bf16) + fp32 residual read + fp32 x written + 16-bit copy + statistics slots"
                    kern, wg = ("gemm_bf16_kp_kernel<64, 64, 4", 512) if kp else ("gemm_bf16_glds_kernel<64, 64, 4", 256)
                    tiles = -(-M // 64) * (N // 64)
                else:
                    x = A16.view(torch.bfloat16).float().view(M, 16, 64)
                    stats = torch.stack([x.sum(2), (x * x).sum(2)], dim=2).contiguous()
                    cs = torch.randn((N,), generator=g).cuda()
                    out16 = kind == "cons16"
                    y = torch.empty((M, N), dtype=torch.int16 if out16 else torch.float32, device="cuda")
                    call = lambda i: _hip.check(lib.rgrg_debug_linear_bf16_ln_kp(A16.data_ptr(), Wb[i % ncopy].data_ptr(), b.data_ptr(), None,  # noqa: E731
                                                                                 None if out16 else y.data_ptr(), y.data_ptr() if out16 else None, None, None,
                                                                                 stats.data_ptr(), cs.data_ptr(), M, N, K, N, 2 if out16 else 0, 0, kp, st))
                    alg = M * K * 2 + N * K * 2 + M * N * (2 if out16 else 4) + M * 16 * 8
                    what = "A + W (bf16) + the output (" + ("16 bit" if out16 else "fp32") + ") + statistics slots"
                    if kp:
                        kern, wg, tiles = "gemm_bf16_pr_kernel", 512, -(-M // 128) * (N // 128)
                    else:
                        kern, wg, tiles = "gemm_bf16_glds_kernel", 256, None   # tile picked by the launcher's heuristic: matched by name + case order
                for i in range(args.launches):
                    call(i)
                torch.cuda.synchronize()
                manifest.append({"case": f"{name}_M{M}", "kernel": kern, "threads": None if tiles is None else tiles * wg, "M": M, "N": N, "K": K,
                                 "launches": args.launches, "algorithmic_bytes": alg, "what": what, "kp": kp})
                del Wb
        M, N, K = 923, 50257, 1024
        A16 = torch.randn((M, K), generator=g).bfloat16().view(torch.int16).cuda()
        ncopy = 6
        Wb = (torch.randn((ncopy, N, K), generator=g) / K ** 0.5).bfloat16().view(torch.int16).cuda()
        nt = (N + 255) // 256
        cv, ci = torch.empty((M, nt), device="cuda"), torch.empty((M, nt), dtype=torch.int32, device="cuda")
        for i in range(12):
            _hip.check(lib.rgrg_debug_linear_bf16_argmax(A16.data_ptr(), Wb[i % ncopy].data_ptr(), None, M, N, K, cv.data_ptr(), ci.data_ptr(), 0, st))
        torch.cuda.synchronize()
        manifest.append({"case": "lm_head_argmax_M923", "kernel": "gemm_bf16_pp_kernel", "threads": 4 * nt * 512, "M": M, "N": N, "K": K, "launches": 12,
                         "algorithmic_bytes": M * K * 2 + N * K * 2 + M * nt * 8, "what": "A + W (bf16) + one (maximum, column) pair per row and 256-column tile"})
    txt = json.dumps(manifest, indent=1)
    if args.manifest:
        open(args.manifest, "w").write(txt)
    print(txt)


if __name__ == "__main__":
    main()
