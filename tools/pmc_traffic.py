"""HBM traffic per launch of the decode-step kernel families from two rocprofv3 PMC passes of the SAME bench command
(one with --pmc FETCH_SIZE, one with --pmc WRITE_SIZE; counters are collected in their own runs, never together with a
trace).  Bytes per launch = (2 x FETCH_SIZE + WRITE_SIZE) x 1024: the counters are in KiB, and on gfx950 FETCH_SIZE
reports half of the bytes of a wide coalesced read (MI355X_MICROARCH.md, HBM section).

  pmc_traffic.py <fetch_dir> <write_dir> <cfg> <out.json>     cfg e.g. S29_f32, S928_bf16

Updates profiles/r02_pmc_traffic.json-style files: keys gemm_<cfg>, attn_<cfg> (mean bytes per launch) and
<key>_launches."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

FAMILIES = {"gemm": ("rgrg_skinny_direct", "rgrg_lm_head_wave", "gemm_bf16_glds_kernel", "gemm_bf16w_kernel", "rgrg_skinny_gemm_f32"),
            "attn": ("attn_decode",)}


def per_kernel(d, counter):
    files = glob.glob(f"{d}/**/*counter_collection.csv", recursive=True)
    assert files, f"no counter_collection.csv under {d}"
    agg = defaultdict(lambda: [0, 0.0])
    for f in files:
        with open(f) as fh:
            for r in csv.DictReader(fh):
                if r["Counter_Name"] != counter:
                    continue
                a = agg[r["Kernel_Name"].split("(")[0]]
                a[0] += 1
                a[1] += float(r["Counter_Value"])
    return agg


def main(fetch_dir, write_dir, cfg, out):
    fetch, write = per_kernel(fetch_dir, "FETCH_SIZE"), per_kernel(write_dir, "WRITE_SIZE")
    res = {}
    if os.path.exists(out):
        with open(out) as f:
            res = json.load(f)
    for fam, pats in FAMILIES.items():
        n = fb = wb = 0.0
        for k, (c, tot) in fetch.items():
            if any(p in k for p in pats):
                n += c
                fb += tot
        for k, (c, tot) in write.items():
            if any(p in k for p in pats):
                wb += tot
        if n:
            res[f"{fam}_{cfg}"] = (2.0 * fb + wb) * 1024.0 / n
            res[f"{fam}_{cfg}_launches"] = int(n)
            res[f"{fam}_{cfg}_fetch_kib_per_launch"] = fb / n
            res[f"{fam}_{cfg}_write_kib_per_launch"] = wb / n
    with open(out, "w") as f:
        json.dump(res, f, indent=1, sort_keys=True)
    print(json.dumps({k: v for k, v in res.items() if cfg in k}, indent=1))


if __name__ == "__main__":
    main(*sys.argv[1:5])
