"""HBM traffic per launch of the batch-32 bf16 decode-step GEMMs from rocprofv3 PMC passes over tools/gemm_bf16_bench.py
(the counter mode of rocprofv3 crashes on this image when the profiled process runs the many-row decoder itself, so the step's
GEMM kernels are measured in the micro-benchmark: same kernels, same shapes, cold weights).

  rocprofv3 --pmc FETCH_SIZE --output-format csv -d F -- python tools/gemm_bf16_bench.py --cold --tiles 0 --iters 6
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d W -- python tools/gemm_bf16_bench.py --cold --tiles 0 --iters 6
  python tools/pmc_gemm_step_traffic.py F W profiles/r03_pmc_traffic.json

Bytes per launch = (2 x FETCH_SIZE + WRITE_SIZE) x 1024 (KiB counters; gfx950 FETCH_SIZE correction of MI355X_MICROARCH.md);
the step mix is 24 x (c_attn + attn_proj + c_fc + mlp_proj) + lm_head = 97 launches, each shape weighted by its launches in the
bench run (equal iteration counts per shape, so the per-instance means are already the step's mix within an instance)."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from pmc_traffic import per_kernel  # noqa: E402

# glds instance -> launches per decode step at M = 923 (launch_glds heuristic, gemm_bf16.hip)
# (round 4: the kernel has two more template arguments - the 16-bit type and the LayerNorm-fold role; the micro-benchmark runs
# the plain bf16 instances, the decode step their producer / consumer variants of the same tile - one more 16-bit store or a
# 256-B statistics read per row on top of the same operand stream)
STEP_MIX = {"<64, 64, 4, false, false, 0>": 48, "<128, 64, 3, false, false, 0>": 24, "<64, 64, 3, false, false, 0>": 24,
            "<128, 128, 2, false, false, 0>": 1}


def main(fetch_dir, write_dir, out):
    fetch, write = per_kernel(fetch_dir, "FETCH_SIZE"), per_kernel(write_dir, "WRITE_SIZE")
    res = {}
    if os.path.exists(out):
        with open(out) as f:
            res = json.load(f)
    tot = n = 0.0
    detail = {}
    for inst, launches in STEP_MIX.items():
        fk = [k for k in fetch if "gemm_bf16_glds_kernel" in k and inst in k]
        wk = [k for k in write if "gemm_bf16_glds_kernel" in k and inst in k]
        assert fk and wk, f"instance {inst} not in the PMC run"
        fb = sum(fetch[k][1] for k in fk) / sum(fetch[k][0] for k in fk)
        wb = sum(write[k][1] for k in wk) / sum(write[k][0] for k in wk)
        b = (2.0 * fb + wb) * 1024.0
        detail[inst] = {"bytes_per_launch": b, "fetch_kib": fb, "write_kib": wb, "launches_per_step": launches}
        tot += b * launches
        n += launches
    res["gemm_S923_bf16"] = tot / n
    res["gemm_S923_bf16_launches"] = int(n)
    res["gemm_S923_bf16_per_instance"] = detail
    res["gemm_S923_bf16_note"] = ("from PMC passes over tools/gemm_bf16_bench.py --cold (same kernels and shapes as the decode step; "
                                  "rocprofv3 --pmc crashes on the many-row decoder process on this image)")
    with open(out, "w") as f:
        json.dump(res, f, indent=1, sort_keys=True)
    print(json.dumps({k: v for k, v in res.items() if k.startswith("gemm_S923")}, indent=1))


if __name__ == "__main__":
    main(*sys.argv[1:4])
