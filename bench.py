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
    detector finds on the synthetic batch moves by one or two between builds).  -> (bytes per launch, "file:key") or
    (None, None) when there is none: the quoted number always names the profile it comes from."""
    import re
    for name in ("r06_pmc_traffic.json", "r05_pmc_traffic.json", "r04_pmc_traffic.json", "r03_pmc_traffic.json", "r02_pmc_traffic.json"):  # newest committed measurement that has the key
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
                    best = (abs(int(m.group(1)) - S_dec), float(v), k)
        if best is not None:
            note = table.get(best[2] + "_note")
            return best[1], f"profiles/{name}:{best[2]}" + (f" ({note})" if note else "")
    return None, None


def rooflines(eng, S_dec, dtype, max_length):
    """Rooflines of the two kernel families of a decode step (>= 90 % of a generate() call), both timed live with HIP
    events on the decoder's stream (engine.time_step_parts): the projection GEMMs of one step and its 24 single-query
    attention launches at the mid-sequence key count.  <= 128 token rows: the GEMMs are weight-streaming (HBM bound,
    algorithmic bytes = the fp32 weights of the step, each read once); above: MFMA bound (2 M N K flops against the
    dense peak of the compute dtype).  Attention is HBM bound on the K/V cache bytes it must read."""
    nkeys = (2 + (max_length + 1)) // 2
    # Many-sequence 16-bit step: the step runs as 4 row ranges on forked streams whose launches overlap each other (and the other
    # ranges' attention), so there is no per-launch duration to take from it.  The roofline times every kernel ALONE on the GPU at
    # the step's full row count (one_range) - the figure a serialising profiler gives for a 1-range step (profiles/
    # r06_kernel_trace_summary_b32_bf16_one_range.md) - and quotes the as-launched (concurrent) family times next to it.
    many = S_dec > eng.fused_row_limit()   # the many-sequence path: > 128 rows, or > 64 under autocast (decoder.hip decode_row_limit)
    w16 = not many and dtype != "f32" and S_dec > 32   # 33-64 rows under autocast: the fused plan on 16-bit weights
    ranged = S_dec >= 512 and dtype != "f32"
    p = eng.time_step_parts(S_dec, nkeys, iters=10, one_range=ranged)   # (3 replays read 3-6 % slow: the first one runs on ramping clocks)
    p_conc = eng.time_step_parts(S_dec, nkeys, iters=10) if ranged else None
    n = max(p["gemm_launches"], 1)
    g_traffic, g_src = pmc_traffic("gemm", S_dec, dtype)
    a_traffic, a_src = pmc_traffic("attn", S_dec, dtype)
    if not many:
        ach = p["gemm_weight_bytes"] / (p["ms_gemm"] * 1e-3) / 1e9
        gemm = {"bound": "hbm", "kernel": "rgrg_skinny_direct_f32<.., W16> (16-bit weights)" if w16 else "rgrg_skinny_direct_f32 (+ _half, rgrg_lm_head_wave_f32)", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": ach / HBM_PEAK_GBS, "traffic": g_traffic, "traffic_source": g_src, "launches_per_decode_step": n,
                "avg_launch_us": 1e3 * p["ms_gemm"] / n, "algorithmic_bytes_per_launch": p["gemm_weight_bytes"] / n,
                "note": "achieved = " + ("16-bit" if w16 else "fp32") + " weight bytes of the GEMM launches of one decode step (each weight read once) / their duration "
                        "between two HIP events on the decoder stream, launched back to back in step order"}
    else:
        peak = MFMA_PEAK_TFS["f32" if dtype == "f32" else "bf16"]
        ach = p["gemm_flops"] / (p["ms_gemm"] * 1e-3) / 1e12
        kname = "gemm_f32_kernel"
        if dtype != "f32":
            kname = ("gemm_bf16_glds_kernel (c_attn, c_fc; attn_proj / mlp_proj on 2 / 4 K slices: steps of <= 256 rows) + lm_head" if S_dec <= 256 else
                     "gemm_bf16_glds_kernel (c_attn, c_fc) + gemm_bf16_kp_kernel (attn_proj, mlp_proj) + gemm_bf16_pp_kernel (lm_head)")
        gemm = {"bound": "mfma", "kernel": kname, "achieved": ach, "peak": peak,
                "unit": "TFLOP/s", "frac": ach / peak, "traffic": g_traffic, "traffic_source": g_src, "launches_per_decode_step": n,
                "avg_launch_us": 1e3 * p["ms_gemm"] / n, "algorithmic_flops_per_launch": p["gemm_flops"] / n,
                "note": "achieved = 2 M N K of the GEMM launches of one decode step / their duration between two HIP events on the decoder stream, "
                        "every launch over ALL rows of the step, back to back (each kernel alone on the GPU).  The step itself launches them as "
                        "4 row ranges on forked streams (decoder.hip run_row_ranges) that overlap each other and the other ranges' attention: "
                        "`as_launched_*` = the same launches issued that way, GEMMs only"}
        if p_conc is not None:
            gemm["as_launched_ms_per_decode_step"] = p_conc["ms_gemm"]
            gemm["as_launched_launches_per_decode_step"] = p_conc["gemm_launches"]
    ach = p["kv_bytes"] / (p["ms_attn"] * 1e-3) / 1e9
    attn = {"bound": "hbm", "kernel": "attn_decode_kv16_wave_kernel" if (dtype != "f32" and many) else "attn_decode_kernel",
            "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS, "traffic": a_traffic, "traffic_source": a_src,
            "launches_per_decode_step": 24, "avg_launch_us": 1e3 * p["ms_attn"] / 24, "algorithmic_bytes_per_launch": p["kv_bytes"] / 24,
            "keys_per_sequence": nkeys,
            "note": "achieved = K/V cache bytes of the attention launches of one step at the mid-sequence key count / their duration "
                    "between two HIP events on the decoder stream (24 launches, each over all sequences of the step)"}
    if p_conc is not None:
        attn["as_launched_ms_per_decode_step"] = p_conc["ms_attn"]
    gemm["ms_per_decode_step"], attn["ms_per_decode_step"] = p["ms_gemm"], p["ms_attn"]
    return (gemm, attn) if p["ms_gemm"] >= p["ms_attn"] else (attn, gemm)


def cpu_baseline(sd, images, max_length, sample_steps=4, full_runs=3, all_cores_full_run=False):
    """The CPU oracle (port of the reference's algorithm; oracle/) timed on this host on ONE image of the workload
    (1 warm-up + 3 timed runs, median).  Two stages, so that the default run stays within minutes:
      1. thread-count probe - one short run per candidate thread count (all physical cores, 32, 8): detector + selection in
         full + `sample_steps` greedy decode steps, the decode time extrapolated to the max_length-1 steps; these figures are
         listed as EXTRAPOLATED and only pick the thread count (small fp32 GEMVs do not scale with cores);
      2. at the best thread count: 1 warm-up + `full_runs` FULL runs - detector, selection and all max_length-1 decode steps of
         the 29 regions, nothing scaled - whose median is `value`."""
    from oracle import detector as o_det
    from oracle import full_model as o_full
    from oracle import language_model as o_lm
    try:
        import psutil
        phys = psutil.cpu_count(logical=False) or os.cpu_count() or 1
    except Exception:  # noqa: BLE001
        phys = os.cpu_count() or 1

    def one_run(steps):
        """steps = None: the full max_length-1 steps; else that many, scaled."""
        t0 = time.perf_counter()
        _, _, top, cd = o_det.object_detector_forward(sd, images[:1])
        sel, feats, _ = o_full.region_selection(sd, top, cd)
        t_det = time.perf_counter() - t0
        S, t_dec = int(feats.shape[0]), 0.0
        if S:
            t0 = time.perf_counter()
            if steps is None:
                o_lm.greedy_generate(sd, feats, max_length)
                t_dec = time.perf_counter() - t0
            else:
                o_lm.greedy_generate(sd, feats, steps + 1)
                t_dec = (time.perf_counter() - t0) * (max_length - 1) / steps
        return t_det + t_dec, t_det, t_dec, S

    probe = {}
    best_n, best_t = None, None
    torch.set_num_threads(min(8, phys))
    one_run(1)   # first-touch / lazy initialisation costs stay out of the probe (they used to land on its first candidate)
    for n in sorted({phys, min(32, phys), min(8, phys)}):
        torch.set_num_threads(n)
        tot = one_run(sample_steps)[0]
        probe[str(n)] = 1.0 / tot
        if best_t is None or tot < 0.9 * best_t:   # more threads only for a clear win: a 4-step probe is noisy, and the full runs decide `value`
            best_n, best_t = n, tot
    torch.set_num_threads(best_n)
    one_run(None)  # warm-up, full
    runs = sorted(one_run(None) for _ in range(full_runs))
    tot, t_det, t_dec, S = runs[len(runs) // 2]
    all_cores = None
    if all_cores_full_run and phys != best_n:
        # SURVEY 8(d) asks for the reference path on ALL the node's physical cores: one FULL run (nothing extrapolated) at that
        # thread count - minutes on a 128-core host (small fp32 GEMVs get slower with more threads), hence opt-in
        torch.set_num_threads(phys)
        a_tot, a_det, a_dec, _ = one_run(None)
        torch.set_num_threads(best_n)
        all_cores = {"threads": phys, "images_per_sec": 1.0 / a_tot, "seconds_per_image": a_tot, "seconds_detector_and_selection": a_det,
                     "seconds_decode": a_dec, "runs": "one full run (all decode steps), no warm-up"}
    return {"all_physical_cores_full_run": all_cores, "value": 1.0 / tot, "unit": "images/sec", "cores": best_n, "kind": "port", "physical_cores": phys,
            "runs": f"1 warm-up + {full_runs} full timed runs at {best_n} threads, median (all {max_length - 1} decode steps run, nothing extrapolated)",
            "seconds_per_image": tot, "seconds_detector_and_selection": t_det, "seconds_decode": t_dec,
            "images_per_sec_by_threads_extrapolated": probe,
            "probe": f"one run per thread count with {sample_steps} of {max_length - 1} decode steps scaled: picks the thread count only",
            "sample": f"1 image in full: detector + selection ({t_det:.1f} s) + {max_length - 1} greedy decode steps for {S} regions "
                      f"({t_dec:.1f} s); torch-CPU fp32 oracle on {best_n} threads of {phys} physical cores"}


def detector_rooflines(eng, images, bf16):
    """rooflines of the detector's MFMA work and of RoIAlign, timed live with HIP events on the stream the
    kernels are launched on (the detector runs on torch's current stream, so torch.cuda.Event brackets exactly those
    launches).  MFMA: ResNet-50 trunk + RPN 3x3 + fc6 (99 % of the detector's 332.9 GFLOP / image at 1000 proposals);
    RoIAlign + avg-pool: algorithmic bytes = feature map read once + [R, 64, 2048] maps and [R, 2048] pooled rows written."""
    from rgrg_amd import _hip
    B = images.shape[0]

    def timed(fn, iters=8):   # (3 iterations of a 0.12 ms kernel read 20 % high: clocks still ramping)
        out = None
        for _ in range(3):
            out = None
            out = fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            out = None
            out = fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / iters, out

    st = lambda: torch.cuda.current_stream().cuda_stream  # noqa: E731
    if bf16:
        peak, dt = MFMA_PEAK_TFS["bf16"], "bf16"
        ms_trunk, (feat16, feat) = timed(lambda: eng.backbone16(images))
        ms_rpn, _ = timed(lambda: eng.conv16(feat16, eng.rpn_conv, _hip.ACT_RELU))
        props, counts, offsets = eng.rpn(feat, feat16=feat16)
    else:
        peak, dt = MFMA_PEAK_TFS["f32"], "f32"
        ms_trunk, feat = timed(lambda: eng.backbone(images))
        ms_rpn, _ = timed(lambda: eng.conv(feat, eng.rpn_conv, _hip.ACT_RELU))
        props, counts, offsets = eng.rpn(feat)
    R = int(offsets[-1].item())
    Cf = feat.shape[-1]
    low = bf16 and R > 128
    maps = torch.empty((R, 64, Cf), dtype=torch.int16 if low else torch.float32, device=feat.device)
    pooled = torch.empty((R, Cf), dtype=torch.float32, device=feat.device)

    def roi():
        fn = eng.lib.rgrg_roi_align_avgpool_bf16maps if low else eng.lib.rgrg_roi_align_avgpool_f32
        _hip.check(fn(feat.data_ptr(), props.data_ptr(), offsets.data_ptr(), maps.data_ptr(), pooled.data_ptr(), B, feat.shape[1], feat.shape[2],
                      Cf, props.shape[1], R, 1.0 / 32, *(([0]) if low else []), st()), "roi_align")

    ms_roi, _ = timed(roi)
    if low:
        h6 = torch.empty((R, 1024), device=feat.device)
        wb = eng._fc6_bf16()

        def fc6():
            _hip.check(eng.lib.rgrg_linear_bf16_f32(maps.data_ptr(), wb.data_ptr(), eng.fc6_b.data_ptr(), None, h6.data_ptr(), None, R, 1024,
                                                    64 * Cf, 1024, _hip.ACT_RELU, 0, st()), "fc6")
        ms_fc6, _ = timed(fc6)
    else:
        if maps.dtype != torch.float32:
            maps = maps.float()
        x6 = maps.view(R, 64 * Cf)
        ms_fc6, _ = timed(lambda: eng.linear(x6, eng.fc6_w, eng.fc6_b, _hip.ACT_RELU))
    fl = {"trunk": 2 * 20.942e9 * B, "rpn_conv3x3": 2.0 * 256 * 2048 * 18432 * B, "fc6": 2.0 * R * 131072 * 1024}
    ms = {"trunk": ms_trunk, "rpn_conv3x3": ms_rpn, "fc6": ms_fc6}
    ach = sum(fl.values()) / (sum(ms.values()) * 1e-3) / 1e12
    det = {"bound": "mfma", "kernel": ("gemm_bf16_glds_kernel (implicit-GEMM convolutions + fc6)" if bf16 else "gemm_f32_kernel (implicit-GEMM convolutions + fc6)"),
           "achieved": ach, "peak": peak, "unit": "TFLOP/s", "frac": ach / peak, "traffic": None, "dtype": dt, "images": B, "proposals": R,
           "parts": {k: {"ms": ms[k], "tflops": fl[k] / (ms[k] * 1e-3) / 1e12, "frac": fl[k] / (ms[k] * 1e-3) / 1e12 / peak} for k in fl},
           "note": "achieved = 2 M N K of trunk + RPN 3x3 + fc6 / the sum of their durations between HIP events on the launch stream "
                   "(mean of 3 after 2 warm-ups, each stage timed back to back on resident inputs)"}
    bytes_roi = B * Cf * feat.shape[1] * feat.shape[2] * 4 + R * (Cf * 64 * (2 if low else 4) + Cf * 4)
    ach = bytes_roi / (ms_roi * 1e-3) / 1e9
    ra = {"bound": "hbm", "kernel": "roi_align_avg_kernel" + ("<bf16 maps>" if low else "<fp32 maps>"), "achieved": ach, "peak": HBM_PEAK_GBS,
          "unit": "GB/s", "frac": ach / HBM_PEAK_GBS, "traffic": None, "ms": ms_roi, "proposals": R, "algorithmic_bytes_per_launch": bytes_roi,
          "note": "algorithmic bytes = feature map read once + [R, 64, 2048] maps + [R, 2048] pooled rows written, / the launch's duration "
                  "between HIP events on its stream"}
    return det, ra


def workload_name(batch, max_length, dtype, world):
    """config.workload of the headline line: which BASELINE configuration the per-GPU work is."""
    base = f"full_model.generate() batch={batch}/GPU, 29 regions, greedy max_len={max_length}, "
    if dtype == "f32":
        return base + "fp32" + (" (BASELINE configs[1])" if batch == 1 and max_length == 128 else "")
    tag = ""
    if batch == 32 and max_length == 128:
        tag = " (BASELINE configs[2])" if world == 1 else (f" (BASELINE configs[3]{'' if world == 8 else ': the 8-GPU shape at this N'}: "
                                                          f"batch={32 * world} sharded 32 / GPU, one RCCL all_gather of the token ids)")
    return base + "bf16 autocast: 16-bit detector, 16-bit decode GEMMs + K/V cache, hipGraph-captured decode step" + tag


def generate_leg(model, synth, batch, dtype, max_length, steps, warmup, dev, num_beams=1, early_stopping=False, with_rooflines=True):
    """One more generate() workload inside the same process (N = 1 legs: `config1`, `beam4`): `warmup` + `steps` calls between device
    synchronisations, with the rooflines of its decode-step kernel families and of its detector call."""
    import contextlib
    images = synth.make_images(batch, 1234).to(dev)
    ctx = (lambda: torch.autocast("cuda", dtype={"bf16": torch.bfloat16, "f16": torch.float16}[dtype])) if dtype != "f32" else contextlib.nullcontext

    def step():
        with ctx():
            return model.generate(images, max_length=max_length, num_beams=num_beams, early_stopping=early_stopping)
    out = None
    for _ in range(warmup):
        out = step()
    dt, out = _bracketed(step, steps, False, dev)
    S = 0 if isinstance(out, int) else int(out[0].shape[0])
    res = {"value": steps * batch / dt, "unit": "images/sec", "n_gpus": 1, "ms_per_step": 1e3 * dt / steps, "steps": steps, "warmup": warmup,
           "dtype": dtype, "global_batch": batch, "regions_generated": S, "tokens_per_region": 0 if isinstance(out, int) else int(out[0].shape[1])}
    if with_rooflines:
        try:
            rows = S * num_beams
            res["roofline"], res["roofline_secondary"] = rooflines(model.engine(), max(rows, 1), dtype if (dtype != "f32" and rows > 32) else "f32", max_length)   # <= 32 rows: bit-exact fp32 whatever the autocast state
            res["roofline_detector"], res["roofline_roialign"] = detector_rooflines(model.engine(), images, dtype != "f32")
        except Exception as e:  # noqa: BLE001
            res["roofline"] = {"error": str(e)}
    return res


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


def _bracketed(fn, calls, use_dist, dev):
    """`calls` calls of fn between two barriers (+ device synchronisation) on every rank; -> (seconds, MAX over ranks; last result)."""
    def barrier():
        if use_dist:
            torch.distributed.barrier()
        torch.cuda.synchronize()
    out = None
    barrier()
    t0 = time.perf_counter()
    for _ in range(calls):
        out = fn()
    barrier()
    dt = time.perf_counter() - t0
    if use_dist:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        dt = float(t.item())
    return dt, out


def config4_line(model, synth, world=1, use_dist=False, dev=None, steps=3, T=64):
    """BASELINE configs[4]: one end-to-end training step per call - 8 images per GPU, object detector frozen (its inference
    branch runs), both region classifiers + the language model's trainable tensors (53.66 M values) - forward under bf16
    autocast, HIP backward, gradients ALL-REDUCED over the ranks in place on flat buckets (rgrg_amd.dist.GradBuckets: RCCL
    over xGMI at world > 1), HIP AdamW.  1 warm-up + `steps` timed steps between barriers, MAX over ranks.  Run LAST: it
    updates the weights."""
    from rgrg_amd import optim
    from rgrg_amd.dist import GradBuckets
    dev = dev or next(iter(model.parameters())).device
    B = 8
    S = 29 * B
    g = torch.Generator().manual_seed(1000 + (torch.distributed.get_rank() if use_dist else 0))   # every rank its own shard
    ids = torch.randint(0, 50257, (S, T), generator=g).to(dev)
    lens = torch.randint(T // 2, T + 1, (S,), generator=g)
    am = (torch.arange(T)[None, :] < lens[:, None]).to(torch.int64).to(dev)
    images = synth.make_images(B, 4321).to(dev)
    has = torch.ones((B, 29), dtype=torch.bool, device=dev)
    abn = (torch.rand((B, 29), generator=g) < 0.2).to(dev)
    was_pre, was_training = model.pretrain_without_lm_model, model.training
    model.pretrain_without_lm_model = False
    model.train()
    try:
        params = model.trainable_parameters()
        opt = optim.AdamW(params, lr=5e-5)
        buckets = GradBuckets(params)
        n_buckets = len(buckets.buckets)

        def step():
            buckets.zero()
            with torch.autocast("cuda", dtype=torch.bfloat16):
                out = model(images, None, ids.clone(), am, has, abn)
            (5.0 * out[1] + 5.0 * out[2] + 2.0 * out[3]).backward()     # run_configurations.py:58-61 weights (detector frozen)
            buckets.allreduce()
            opt.step()
            return out

        step()
        dt, out = _bracketed(step, steps, use_dist, dev)
        n_values = sum(p.numel() for p in params)
        # roofline of the step's dominant kernel family (the frozen-weight GEMMs of forward + activation gradients: 46 % of
        # the GPU time of a step): timed live with HIP events on the decoder's stream, launched back to back with the step's
        # own operands and epilogues; `step_frac` = the same flops over the WHOLE step (attention, row kernels, detector,
        # optimizer included)
        roof = None
        try:
            eng = model.engine()
            S_run, T_run = eng.last_train_shape   # the sentences the step really fed (a region that is not detected is dropped)
            t = eng.time_train_gemms(S_run, T_run, 3)
            tf = t["gemm_flops"] / (t["ms_gemm"] * 1e-3) / 1e12
            roof = {"bound": "mfma", "kernel": "gemm_bf16_pp_kernel (256 x 256 ping-pong, all GEMMs of the step at 14 848 rows)",
                    "achieved": tf, "peak": 2500.0, "unit": "TFLOP/s", "frac": tf / 2500.0, "traffic": None,
                    "launches_per_step": t["gemm_launches"], "avg_launch_us": 1e3 * t["ms_gemm"] / t["gemm_launches"],
                    "algorithmic_flops_per_launch": t["gemm_flops"] / t["gemm_launches"], "ms_gemm_per_step": t["ms_gemm"],
                    "step_frac": t["gemm_flops"] / (dt / steps) / 1e12 / 2500.0,
                    "token_rows": S_run * T_run,
                    "note": "achieved = 2 M N K of the frozen-weight GEMMs of one step (forward + activation gradients of 24 blocks, lm_head "
                            "forward + dgrad; M = token_rows) / their duration between two HIP events on the decoder stream"}
        except Exception as e:  # noqa: BLE001
            roof = {"error": str(e)}
        return {"workload": f"end-to-end training step, detector frozen, LM + binary-classifier heads, per-GPU batch=8 (232 sentences x {T} tokens), "
                            f"bf16 autocast, gradient all-reduce over {world} rank(s) on {n_buckets} flat buckets, HIP AdamW (BASELINE configs[4]"
                            + ("" if world == 8 else ": the 8-GPU shape at this N") + ")",
                "metric": "images/sec end-to-end training step", "value": B * world * steps / dt, "unit": "images/sec", "n_gpus": world,
                "ms_per_step": 1e3 * dt / steps, "steps": steps, "warmup": 1, "dtype": "bf16", "scaling": "weak", "global_batch": B * world,
                "trainable_values": n_values, "allreduce_bytes_per_step": 4 * n_values if world > 1 else 0,
                "losses_last_step": [float(o.detach()) for o in out[1:4]], "roofline": roof}
    finally:
        for p in model.trainable_parameters():
            p.grad = None
        model.pretrain_without_lm_model = was_pre
        model.train(was_training)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=6)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=32, help="images per GPU per step (BASELINE configs[2] / configs[3]: 32; configs[1]: --batch 1 --dtype f32)")
    ap.add_argument("--max-length", type=int, default=128)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-cpu-baseline-all-cores", action="store_true",
                    help="cpu_baseline skips its ONE full run of the oracle on all physical cores (~2 minutes on a 128-core host)")
    ap.add_argument("--cpu-baseline-all-cores", action="store_true", help=argparse.SUPPRESS)   # round-5 spelling: now the default
    ap.add_argument("--no-config2", "--no-extra-legs", dest="no_config2", action="store_true",
                    help="skip the secondary legs of the default run: batch 1 in fp32 (BASELINE configs[1], `config1`), the scripts' beam mode "
                         "(`beam4`) and the training step (configs[4], `config4`)")
    ap.add_argument("--train", action="store_true", help="the headline line IS the training step (BASELINE configs[4]) instead of generate()")
    ap.add_argument("--dtype", choices=("f32", "bf16"), default="bf16",
                    help="bf16 (default, BASELINE configs[2] / [3]): generate() under torch.autocast(bfloat16) - 16-bit matrix-core GEMMs, 16-bit K/V "
                         "cache, 16-bit detector; parity there is statistical (README).  f32: the bit-exact path (configs[1] with --batch 1)")
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

    if args.train and not stub:
        # BASELINE configs[4] as the line itself: `--steps` training steps of 8 images per GPU
        res4 = config4_line(model, synth, world, use_dist, dev, steps=args.steps)
        if rank == 0:
            res = {"metric": res4["metric"] + " (BASELINE configs[4]; the headline metric is generate(): run without --train)",
                   "value": res4["value"], "unit": "images/sec", "n_gpus": world, "steps": args.steps, "warmup": 1,
                   "ms_per_step": res4["ms_per_step"], "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "bf16",
                   "data": "synthetic", "config": {"workload": res4["workload"], "global_batch": res4["global_batch"],
                                                   "parallelism": f"dp{world} (replicas, bucketed gradient all-reduce)" if use_dist else "single GPU",
                                                   "weights": "seeded random init (rgrg_amd.synth, profile bench)"},
                   "roofline": res4.get("roofline"), "training": res4}
            print(json.dumps(res), flush=True)
        if use_dist:
            torch.distributed.destroy_process_group()
        return

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

    n_images = args.batch * world * args.steps
    S = 0 if isinstance(out, int) else int(out[0].shape[0])
    Lp = 0 if isinstance(out, int) else int(out[0].shape[1])
    res = {
        "metric": "images/sec full 29-region report gen, 512x512 CXR, greedy max_len=128",
        "value": n_images / dt, "unit": "images/sec", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": 1e3 * dt / args.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": args.dtype, "data": "synthetic",
        "config": {"workload": workload_name(args.batch, args.max_length, args.dtype, world),
                   "global_batch": args.batch * world, "regions_generated": S, "tokens_per_region": Lp,
                   "parallelism": f"dp{world} (image shards, one RCCL all_gather of token ids)" if use_dist else "single GPU",
                   "weights": "seeded random init (rgrg_amd.synth, profile bench)", "hipGraph_decode": True},
    }
    if stub:
        if rank == 0:
            res["data"] = "stub (no GPU work: launcher / gather path only)"
            res["config"]["weights"] = "none (stub)"
            print(json.dumps(res), flush=True)
        if use_dist:
            torch.distributed.destroy_process_group()
        return

    default_workload = args.batch == 32 and args.dtype == "bf16" and args.max_length == 128
    if default_workload:
        res["config"]["note"] = ("headline = 32 images per GPU under bf16 autocast at every N: BASELINE configs[2] at N = 1, configs[3] at N = 8, one weak-"
                                 "scaling curve of ONE workload; the bit-exact fp32 batch-1 path (configs[1]) rides along as `config1`, the scripts' "
                                 "beam mode as `beam4`, the training step (configs[4]) as `config4`.  16-bit parity is statistical (README, DESIGN 7.2): "
                                 "token ids are bit-exact against the reference on the fp32 path only")
    # rooflines of the two kernel families of the decode loop, dominant one first (timed live, HIP events), per rank 0
    if rank == 0:
        try:
            S_dec = max(S // max(world, 1), 1)
            res["roofline"], res["roofline_secondary"] = rooflines(model.engine(), S_dec, args.dtype, args.max_length)
            res["roofline_detector"], res["roofline_roialign"] = detector_rooflines(model.engine(), images, args.dtype == "bf16")
        except Exception as e:  # noqa: BLE001
            res.setdefault("roofline", {"bound": "mfma", "error": str(e)})
    extra = default_workload and not args.no_config2
    if extra and world == 1 and rank == 0:
        try:   # BASELINE configs[1]: batch 1, fp32, greedy, 29 x 128 tokens - the path whose token ids are bit-exact against the reference
            res["config1"] = dict(generate_leg(model, synth, 1, "f32", args.max_length, 5, 1, dev),
                                  workload=workload_name(1, args.max_length, "f32", 1), metric=res["metric"])
        except Exception as e:  # noqa: BLE001
            res["config1"] = {"error": str(e)}
        try:   # what the reference's scripts run: num_beams=4, max_length=300, early_stopping under fp16 autocast
            res["beam4"] = dict(generate_leg(model, synth, 1, "f16", 300, 3, 1, dev, num_beams=4, early_stopping=True),
                                workload="generate_reports_for_images.py:108-114 mode: 1 image, num_beams=4 (116 beam rows), max_length=300, "
                                         "early_stopping=True, torch.autocast(float16): 16-bit detector, and - round 6 - the 116 beam rows on the 16-bit "
                                         "many-sequence decode path (> 64 rows under autocast; the fp32 fused plan before)",
                                metric="images/sec full 29-region report gen, 512x512 CXR, beam search num_beams=4 max_len=300")
        except Exception as e:  # noqa: BLE001
            res["beam4"] = {"error": str(e)}
    if world == 1 and rank == 0 and not args.no_cpu_baseline:
        res["cpu_baseline"] = cpu_baseline(sd, images_cpu, args.max_length, all_cores_full_run=not args.no_cpu_baseline_all_cores)
    if extra:   # last: it updates the weights; every rank takes part (gradient all-reduce at N > 1)
        try:
            res["config4"] = config4_line(model, synth, world, use_dist, dev)
        except Exception as e:  # noqa: BLE001
            res["config4"] = {"error": str(e)}
    if rank == 0:
        print(json.dumps(res), flush=True)
    if use_dist:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
