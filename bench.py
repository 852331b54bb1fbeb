"""bench.py - images/sec of full 29-region report generation (BASELINE.json metric).

A "step" is one ``ReportGenerationModel.generate(images, max_length=128)`` call on a
synthetic 512x512 batch already resident in HBM: detector + region selection + all 127
greedy decode steps (+ the final RCCL gather of token ids when N > 1).  Default workload
is BASELINE configs[1]: batch=1 per GPU, 29 regions, greedy, max_len=128, fp32.

  python bench.py --gpus 1 --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...
  python bench.py --gpus N ...        (no launcher in the environment: re-executes itself under torch.distributed.run
                                       with N ranks on 127.0.0.1 and a free port)

Rank 0 prints ONE JSON line (see DESIGN.md "Measurement").  At N = 1 with the default workload the line also carries
"config2": BASELINE configs[2] (batch 32 under bf16 autocast) timed in the same process after the headline loop, and
"cpu_baseline": the CPU oracle on a bounded sample.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import torch

REPO = os.path.dirname(os.path.abspath(__file__))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)


MFMA_PEAK_TFS = {"bf16": 2500.0, "f32": 157.3}  # MI355X_MICROARCH.md dense peaks (no sparsity)


def pmc_traffic(family, S_dec, dtype):
    """HBM bytes per launch of a kernel family ("gemm" / "attn") from the committed rocprofv3 PMC passes
    (profiles/rNN_pmc_traffic.json, written by tools/pmc_traffic.py / tools/pmc_gemm_step_traffic.py from separate --pmc
    FETCH_SIZE / WRITE_SIZE runs, FETCH_SIZE doubled as the gfx950 note of MI355X_MICROARCH.md prescribes): counters cannot
    be collected from inside the timed process, so the last committed measurement of the same configuration is quoted -
    key <family>_S<sequences>_<dtype>, matched on the dtype and a sequence count within 1 % (the number of regions the
    detector finds on the synthetic batch moves by one or two between builds); None when there is none."""
    import re
    for name in ("r03_pmc_traffic.json", "r02_pmc_traffic.json"):  # newest committed measurement that has the key
        try:
            with open(os.path.join(REPO, "profiles", name)) as f:
                table = json.load(f)
        except Exception:  # noqa: BLE001
            continue
        best = None
        for k, v in table.items():
            m = re.fullmatch(rf"{family}_S(\d+)_{dtype}", k)
            if m and isinstance(v, (int, float)) and abs(int(m.group(1)) - S_dec) <= max(0.01 * S_dec, 0):
                if best is None or abs(int(m.group(1)) - S_dec) < best[0]:
                    best = (abs(int(m.group(1)) - S_dec), float(v))
        if best is not None:
            return best[1]
    return None


def rooflines(eng, S_dec, dtype, max_length):
    """Rooflines of the two kernel families of a decode step (>= 90 % of a generate() call), both timed live with HIP
    events on the decoder's stream (engine.time_step_parts): the projection GEMMs of one step and its 24 single-query
    attention launches at the mid-sequence key count.  <= 128 token rows: the GEMMs are weight-streaming (HBM bound,
    algorithmic bytes = the fp32 weights of the step, each read once); above: MFMA bound (2 M N K flops against the
    dense peak of the compute dtype).  Attention is HBM bound on the K/V cache bytes it must read."""
    nkeys = (2 + (max_length + 1)) // 2
    p = eng.time_step_parts(S_dec, nkeys, iters=3)
    n = max(p["gemm_launches"], 1)
    if S_dec <= 128:
        ach = p["gemm_weight_bytes"] / (p["ms_gemm"] * 1e-3) / 1e9
        gemm = {"bound": "hbm", "kernel": "rgrg_skinny_direct_f32 (+ _half, rgrg_lm_head_wave_f32)", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": ach / HBM_PEAK_GBS, "traffic": pmc_traffic("gemm", S_dec, dtype), "launches_per_decode_step": n,
                "avg_launch_us": 1e3 * p["ms_gemm"] / n, "algorithmic_bytes_per_launch": p["gemm_weight_bytes"] / n,
                "note": "achieved = fp32 weight bytes of the GEMM launches of one decode step (each weight read once) / their duration "
                        "between two HIP events on the decoder stream, launched back to back in step order"}
    else:
        peak = MFMA_PEAK_TFS[dtype]
        ach = p["gemm_flops"] / (p["ms_gemm"] * 1e-3) / 1e12
        gemm = {"bound": "mfma", "kernel": "gemm_bf16_glds_kernel" if dtype == "bf16" else "gemm_f32_kernel", "achieved": ach, "peak": peak,
                "unit": "TFLOP/s", "frac": ach / peak, "traffic": pmc_traffic("gemm", S_dec, dtype), "launches_per_decode_step": n,
                "avg_launch_us": 1e3 * p["ms_gemm"] / n, "algorithmic_flops_per_launch": p["gemm_flops"] / n,
                "note": "achieved = 2 M N K of the GEMM launches of one decode step / their duration between two HIP events on the decoder stream"}
    ach = p["kv_bytes"] / (p["ms_attn"] * 1e-3) / 1e9
    attn = {"bound": "hbm", "kernel": "attn_decode_kv16_wave_kernel" if (dtype == "bf16" and S_dec > 128) else "attn_decode_kernel",
            "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS, "traffic": pmc_traffic("attn", S_dec, dtype),
            "launches_per_decode_step": 24, "avg_launch_us": 1e3 * p["ms_attn"] / 24, "algorithmic_bytes_per_launch": p["kv_bytes"] / 24,
            "keys_per_sequence": nkeys,
            "note": "achieved = K/V cache bytes of the 24 attention launches of one step at the mid-sequence key count / their duration "
                    "between two HIP events on the decoder stream"}
    gemm["ms_per_decode_step"], attn["ms_per_decode_step"] = p["ms_gemm"], p["ms_attn"]
    return (gemm, attn) if p["ms_gemm"] >= p["ms_attn"] else (attn, gemm)


def cpu_baseline(sd, images, max_length, sample_steps=4):
    """The CPU oracle (port of the reference's algorithm; oracle/) timed on this host on a BOUNDED sample of the same
    workload: ONE image through detector + selection (full), then `sample_steps` greedy decode steps; the decode time
    is scaled to the max_length-1 steps of the workload (the per-step cost of the reference's concat-KV decoder grows
    slowly with length, so this slightly flatters the CPU).  As BASELINE.md section 3 prescribes: 1 warm-up + 3 timed
    runs, median, on all PHYSICAL cores (count stated), and 3 more runs with 8 threads (comparable with the survey's
    probe; small fp32 GEMVs do not scale with cores, so this is usually the faster one)."""
    from oracle import detector as o_det
    from oracle import full_model as o_full
    from oracle import language_model as o_lm
    try:
        import psutil
        phys = psutil.cpu_count(logical=False) or os.cpu_count() or 1
    except Exception:  # noqa: BLE001
        phys = os.cpu_count() or 1

    def one_run():
        t0 = time.perf_counter()
        _, _, top, cd = o_det.object_detector_forward(sd, images[:1])
        sel, feats, _ = o_full.region_selection(sd, top, cd)
        t_det = time.perf_counter() - t0
        S, t_dec = int(feats.shape[0]), 0.0
        if S:
            t0 = time.perf_counter()
            o_lm.greedy_generate(sd, feats, sample_steps + 1)
            t_dec = (time.perf_counter() - t0) * (max_length - 1) / sample_steps
        return t_det + t_dec, t_det, t_dec, S

    def median_of(threads, warmup):
        torch.set_num_threads(threads)
        for _ in range(warmup):
            one_run()
        runs = sorted(one_run() for _ in range(3))
        return runs[1]

    # thread counts: all physical cores, 32 and 8 (the survey's probe).  The small fp32 GEMVs of
    # the decode loop do not scale with cores, so `value` is the BEST configuration and every figure is listed.
    by_threads = {}
    best = None
    for i, n in enumerate(sorted({phys, min(32, phys), min(8, phys)}, reverse=True)):
        tot, t_det, t_dec, S = median_of(n, 1 if i == 0 else 0)
        by_threads[str(n)] = 1.0 / tot
        if best is None or tot < best[0]:
            best = (tot, t_det, t_dec, S, n)
    tot, t_det, t_dec, S, n = best
    return {"value": 1.0 / tot, "unit": "images/sec", "cores": n, "kind": "port", "physical_cores": phys,
            "images_per_sec_by_threads": by_threads, "runs": "1 warm-up + 3 timed per thread count, median",
            "sample": f"1 image: detector+selection in full ({t_det:.1f} s) + {sample_steps} of {max_length - 1} greedy decode steps "
                      f"for {S} regions scaled to {max_length - 1} ({t_dec:.1f} s); torch-CPU fp32 oracle; value = the best of "
                      f"the listed thread counts ({n} threads on {phys} physical cores)"}


def _free_port() -> int:
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def respawn_under_launcher(n_gpus: int) -> int:
    """`python bench.py --gpus N` with N > 1 and no launcher environment: re-execute this command line as N ranks of
    one node under torch.distributed.run (rendezvous on 127.0.0.1, a free port).  Returns the launcher's exit code."""
    import subprocess
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n_gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC only on this driver (RCCL needs it)
    env.setdefault("OMP_NUM_THREADS", "8")
    return subprocess.call(cmd, env=env)


class _StubModel:
    """CPU stand-in with the three stage entry points generate_sharded() drives (--stub-cpu: exercises the launcher,
    barrier / max-over-ranks timing, gather and JSON path of this script without a GPU; never a measurement)."""

    class _LM:
        def generate(self, feats, max_length):
            S = feats.shape[0]
            ids = torch.full((S, max_length), 50256, dtype=torch.int64)
            ids[:, 1:] = (torch.arange(S)[:, None] * 7 + torch.arange(max_length - 1)[None, :]) % 50000
            return ids

    def __init__(self):
        self.language_model = self._LM()

    def object_detector(self, images):
        B = images.shape[0]
        det = {"top_region_boxes": torch.zeros((B, 29, 4)), "top_scores": torch.ones((B, 29))}
        return {}, det, torch.zeros((B, 29, 1024)), torch.ones((B, 29), dtype=torch.bool)

    def binary_classifier_region_selection(self, top, cd, return_loss=False):
        return cd.clone(), top.reshape(-1, 1024)


def config2_line(model, synth, max_length):
    """BASELINE configs[2] in the same process: batch 32 under bf16 autocast (bf16-weight MFMA decode GEMMs, bf16 K/V
    cache, hipGraph-captured step), 1 warm-up + 2 timed generate() calls, with its own rooflines."""
    images = synth.make_images(32, 1234).to(next(iter(model.parameters())).device)

    def step():
        with torch.autocast("cuda", dtype=torch.bfloat16):
            return model.generate(images, max_length=max_length, num_beams=1)

    out = step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(2):
        out = step()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    S = 0 if isinstance(out, int) else int(out[0].shape[0])
    res = {"workload": f"full_model.generate() batch=32, 29 regions, greedy max_len={max_length}, bf16 autocast (BASELINE configs[2])",
           "value": 64 / dt, "unit": "images/sec", "ms_per_step": 1e3 * dt / 2, "steps": 2, "warmup": 1, "dtype": "bf16",
           "regions_generated": S, "tokens_per_region": 0 if isinstance(out, int) else int(out[0].shape[1])}
    try:
        res["roofline"], res["roofline_secondary"] = rooflines(model.engine(), max(S, 1), "bf16", max_length)
    except Exception as e:  # noqa: BLE001
        res["roofline"] = {"error": str(e)}
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--batch", type=int, default=1, help="images per GPU per step (BASELINE configs[1]: 1)")
    ap.add_argument("--max-length", type=int, default=128)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-config2", action="store_true", help="skip the batch-32 bf16 leg (BASELINE configs[2]) of the default run")
    ap.add_argument("--dtype", choices=("f32", "bf16"), default="f32",
                    help="bf16: run generate() under torch.autocast(bfloat16) - bf16 MFMA decode GEMMs for > 128 sequences "
                         "(BASELINE configs[2]); not bit-exact, never the default")
    ap.add_argument("--stub-cpu", action="store_true", help=argparse.SUPPRESS)  # tests: launcher / gather path on gloo, no GPU
    args = ap.parse_args()

    if args.gpus > 1 and "RANK" not in os.environ and "WORLD_SIZE" not in os.environ:
        sys.exit(respawn_under_launcher(args.gpus))

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    stub = args.stub_cpu
    if stub:
        dev = torch.device("cpu")
    else:
        assert torch.cuda.is_available(), "bench.py needs an MI355X (the HIP path has no CPU fallback)"
        torch.cuda.set_device(local_rank)
        dev = torch.device("cuda", local_rank)
    # launched by torch.distributed.run (also with one process): one rank per GPU over RCCL ("nccl" on ROCm)
    use_dist = world > 1 or ("RANK" in os.environ and "MASTER_PORT" in os.environ)
    if use_dist:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # RCCL prints a native banner (hostname / library path) on first use: keep stdout clean for the ONE JSON line
        sys.stdout.flush()
        saved_stdout = os.dup(1)
        os.dup2(2, 1)
        try:
            if stub:
                dist.init_process_group("gloo", rank=rank, world_size=world)
            else:
                dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
            dist.barrier()
            if not stub:
                torch.cuda.synchronize()
        finally:
            sys.stdout.flush()
            os.dup2(saved_stdout, 1)
            os.close(saved_stdout)

    from rgrg_amd.dist import generate_sharded
    if stub:
        sd, synth = None, None
        model = _StubModel()
        images_cpu = torch.zeros((args.batch, 1, 8, 8))
        images = images_cpu
    else:
        import rgrg_amd
        from rgrg_amd import synth
        sd = synth.make_state_dict(0, "bench")
        model = rgrg_amd.ReportGenerationModel(pretrain_without_lm_model=True)
        model.load_state_dict(sd)
        model.to(dev).eval()
        images_cpu = synth.make_images(args.batch, 1234)  # same synthetic shard on every rank (weak scaling)
        images = images_cpu.to(dev)

    import contextlib

    def step():
        ctx = torch.autocast("cuda", dtype=torch.bfloat16) if args.dtype == "bf16" and not stub else contextlib.nullcontext()
        with ctx:
            if use_dist:
                return generate_sharded(model, images, args.max_length)
            return model.generate(images, max_length=args.max_length, num_beams=1)

    def barrier():
        if use_dist:
            torch.distributed.barrier()
        if not stub:
            torch.cuda.synchronize()

    out = None
    for _ in range(max(args.warmup, 0)):
        out = step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step()
    barrier()
    dt = time.perf_counter() - t0
    if use_dist:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        dt = float(t.item())

    if rank == 0:
        n_images = args.batch * world * args.steps
        S = 0 if isinstance(out, int) else int(out[0].shape[0])
        Lp = 0 if isinstance(out, int) else int(out[0].shape[1])
        res = {
            "metric": "images/sec full 29-region report gen, 512x512 CXR, greedy max_len=128",
            "value": n_images / dt, "unit": "images/sec", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * dt / args.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": f"full_model.generate() batch={args.batch}/GPU, 29 regions, greedy max_len={args.max_length}, {'fp32' if args.dtype == 'f32' else 'bf16 decode GEMMs (fp32 detector/LN/attention)'}"
                                   + (" (BASELINE configs[1])" if args.batch == 1 and args.max_length == 128 and args.dtype == "f32" else ""),
                       "global_batch": args.batch * world, "regions_generated": S, "tokens_per_region": Lp,
                       "parallelism": f"dp{world} (image shards, one RCCL all_gather of token ids)" if use_dist else "single GPU",
                       "weights": "seeded random init (rgrg_amd.synth, profile bench)", "hipGraph_decode": True},
        }
        if stub:
            res["data"] = "stub (no GPU work: launcher / gather path only)"
            res["config"]["weights"] = "none (stub)"
            print(json.dumps(res), flush=True)
        else:
            # rooflines of the two kernel families of the decode loop, dominant one first (timed live, HIP events)
            try:
                S_dec = max(S // max(world, 1), 1)
                res["roofline"], res["roofline_secondary"] = rooflines(model.engine(), S_dec, args.dtype, args.max_length)
            except Exception as e:  # noqa: BLE001
                res["roofline"] = {"bound": "hbm", "error": str(e)}
            headline = world == 1 and args.batch == 1 and args.dtype == "f32"
            if headline and not args.no_config2:
                try:
                    res["config2"] = config2_line(model, synth, args.max_length)
                except Exception as e:  # noqa: BLE001
                    res["config2"] = {"error": str(e)}
            if world == 1 and not args.no_cpu_baseline:
                res["cpu_baseline"] = cpu_baseline(sd, images_cpu, args.max_length)
            print(json.dumps(res), flush=True)
    if use_dist:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
