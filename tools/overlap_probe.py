"""Do the per-layer GEMMs of the many-sequence decode step (M = 923, the four projection shapes, cold weights) and an HBM-bound
stream really overlap on one MI355X, and if not, which resource do they share?  Stream A loops the four GEMMs; stream B loops a
read-only reduction over (a) 2 GB (HBM), (b) 96 MB (Infinity-Cache resident), (c) 2 MB (L2 resident: same CU occupancy, no
memory-side traffic).  Reported: the GEMM chain's time per layer alone and beside each stream, and the stream's rate alone and
beside the GEMMs.  Run it with RGRG_DECODE_CHAINS=3 (or fewer): the process has 4 hardware queues (GPU_MAX_HW_QUEUES; 2, 6 and
8 were measured much slower or crashing, profiles/r06_hw_queues_sweep.log) and a 4-range decoder owns 4 streams - the probe's two
streams then share a queue and serialise (profiles/r06_overlap_probe_v4_chunks.log reads "x0.95" for that reason).
Usage: python tools/overlap_probe.py [M] [tile] [attn-only]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rgrg_amd import _hip  # noqa: E402

SHAPES = [("c_attn", 3072, 1024), ("attn_proj", 1024, 1024), ("c_fc", 4096, 1024), ("mlp_proj", 1024, 4096)]


def main():
    M = int(sys.argv[1]) if len(sys.argv) > 1 else 923
    tile = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    lib = _hip.load()
    sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
    ops = []
    for name, N, K in SHAPES:
        ncopy = max(1, -(-300_000_000 // (N * K * 2)))
        A16 = (torch.rand((M, K), device="cuda") * 2 - 1).bfloat16().view(torch.int16)
        Wb = ((torch.rand((ncopy, N, K), device="cuda") * 2 - 1) / K ** 0.5).bfloat16().view(torch.int16)
        b = torch.randn((N,), device="cuda")
        Y = torch.zeros((M, N), device="cuda")
        ops.append((A16, Wb, b, Y, N, K, ncopy))
    it = [0]

    def gemm_layer(st):
        it[0] += 1
        for A16, Wb, b, Y, N, K, ncopy in ops:
            _hip.check(lib.rgrg_debug_linear_bf16_tile(A16.data_ptr(), Wb[it[0] % ncopy].data_ptr(), b.data_ptr(), None, Y.data_ptr(), M, N, K, N, 0, tile,
                                                       0, 0, 0, st))

    def time_gemms(layers=200):
        with torch.cuda.stream(sa):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            for _ in range(10):
                gemm_layer(sa.cuda_stream)
            e0.record()
            for _ in range(layers):
                gemm_layer(sa.cuda_stream)
            e1.record()
        return e0, e1, layers

    bufs = {} if "attn-only" in sys.argv else {"hbm 2 GB": torch.randn((512 * 1024 * 1024,), device="cuda"),
                                                 "mall 96 MB": torch.randn((24 * 1024 * 1024,), device="cuda"), "l2 2 MB": torch.randn((512 * 1024,), device="cuda")}

    def stream_load(buf, reps):
        with torch.cuda.stream(sb):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                buf.sum()
            e1.record()
        return e0, e1

    torch.cuda.synchronize()
    e0, e1, L = time_gemms()
    torch.cuda.synchronize()
    alone = e0.elapsed_time(e1) * 1e3 / L
    print(f"M={M} tile={tile}: GEMM chain alone {alone:.1f} us per layer", flush=True)
    for name, buf in ({} if "attn-only" in sys.argv else bufs).items():
        nbytes = buf.numel() * 4
        reps = max(3, int(12e9 / nbytes) if nbytes > 1e8 else int(40e-3 / 10e-6))   # ~10-40 ms of streaming
        for _ in range(2):
            buf.sum()
        torch.cuda.synchronize()
        s0, s1 = stream_load(buf, reps)
        torch.cuda.synchronize()
        t_alone = s0.elapsed_time(s1)
        # concurrent: start the stream, then the GEMMs
        s0, s1 = stream_load(buf, reps * 2)
        e0, e1, L = time_gemms()
        torch.cuda.synchronize()
        t_gemm = e0.elapsed_time(e1) * 1e3 / L
        print(f"  beside {name:11s}: GEMM chain {t_gemm:6.1f} us per layer (x{t_gemm / alone:.2f}); the stream alone {nbytes * reps / t_alone / 1e9:7.1f} GB/s, "
              f"its launches {t_alone / reps * 1e3:.1f} us; beside the GEMMs (whole span, partly alone) {nbytes * reps * 2 / s0.elapsed_time(s1) / 1e9:7.1f} GB/s",
              flush=True)


    # ... and beside the step's own attention kernel (923 sequences, 65 keys: 246 MB of K/V cache per launch)
    import rgrg_amd
    from rgrg_amd import synth
    model = rgrg_amd.ReportGenerationModel(pretrain_without_lm_model=True)
    model.load_state_dict(synth.make_state_dict(0, "bench"))
    model.to("cuda:0").eval()
    S = 923
    feats = torch.randn((S, 1024), generator=torch.Generator().manual_seed(99)).to("cuda:0")
    with torch.autocast("cuda", dtype=torch.bfloat16):
        model.language_model.generate(feats, max_length=128)
    torch.cuda.synchronize()
    eng = model.language_model.engine()
    kv_bytes = 2 * S * 1024 * 65 * 2

    def attn(iters):
        with torch.cuda.stream(sb):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            _hip.check(lib.rgrg_decoder_attention_only(eng._decoder, S, 65, iters, sb.cuda_stream))
            e1.record()
        return e0, e1
    a0, a1 = attn(20)
    torch.cuda.synchronize()
    t_alone = a0.elapsed_time(a1) / (20 * 24)
    a0, a1 = attn(40)
    e0, e1, L = time_gemms(layers=400)
    torch.cuda.synchronize()
    t_gemm = e0.elapsed_time(e1) * 1e3 / L
    t_att = a0.elapsed_time(a1) / (40 * 24)
    print(f"  beside the attention kernel (S={S}, 65 keys): GEMM chain {t_gemm:6.1f} us per layer (x{t_gemm / alone:.2f}); attention alone "
          f"{t_alone * 1e3:.1f} us per launch = {kv_bytes / t_alone / 1e9:.2f} TB/s, beside the GEMMs (whole span) {t_att * 1e3:.1f} us", flush=True)


if __name__ == "__main__":
    main()
