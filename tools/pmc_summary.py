"""Aggregate a rocprofv3 --pmc run (counter_collection.csv) per kernel: launches, mean
counter value per launch.  Usage: pmc_summary.py <dir> <out.md>.
FETCH_SIZE / WRITE_SIZE are reported in KiB; on gfx950 FETCH_SIZE counts 64 B per 128-B
request for wide coalesced reads (MI355X_MICROARCH.md, HBM section) -> x2 correction."""
import csv
import glob
import sys
from collections import defaultdict


def main(d, out=None):
    files = glob.glob(f"{d}/**/*counter_collection.csv", recursive=True)
    assert files, f"no counter_collection.csv under {d}"
    agg = defaultdict(lambda: defaultdict(lambda: [0, 0.0]))
    header = None
    for f in files:
        with open(f) as fh:
            rd = csv.DictReader(fh)
            header = rd.fieldnames
            for r in rd:
                k = r["Kernel_Name"].split("(")[0]
                c = r["Counter_Name"]
                a = agg[k][c]
                a[0] += 1
                a[1] += float(r["Counter_Value"])
    lines = [f"columns: {header}", "", "| kernel | counter | launches | mean per launch | mean bytes per launch (KiB*1024, FETCH x2) |", "|---|---|---|---|---|"]
    for k, cs in sorted(agg.items(), key=lambda kv: -sum(v[1] for v in kv[1].values())):
        for c, (n, tot) in cs.items():
            mean = tot / n
            corr = mean * 1024 * (2 if c == "FETCH_SIZE" else 1) if c in ("FETCH_SIZE", "WRITE_SIZE") else float("nan")
            lines.append(f"| {k[:70]} | {c} | {n} | {mean:.1f} | {corr:.0f} |")
    txt = "\n".join(lines)
    print(txt)
    if out:
        open(out, "w").write(txt + "\n")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else None)
