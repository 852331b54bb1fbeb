"""Summarise a rocprofv3 --kernel-trace output directory: per-kernel count / total / avg
duration, and GPU-busy vs wall span (launch-gap share).  Usage: prof_summary.py <dir> [out.md]"""
import csv
import glob
import sys
from collections import defaultdict


def main(d, out=None):
    files = glob.glob(f"{d}/**/*kernel_trace.csv", recursive=True)
    assert files, f"no kernel_trace.csv under {d}"
    rows = []
    for f in files:
        with open(f) as fh:
            for r in csv.DictReader(fh):
                rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
    rows.sort()
    agg = defaultdict(lambda: [0, 0])
    for s, e, n in rows:
        n = n.split("(")[0]
        agg[n][0] += 1
        agg[n][1] += e - s
    busy = sum(v[1] for v in agg.values())
    span = rows[-1][1] - rows[0][0]
    lines = [f"kernels: {len(rows)}  GPU busy {busy/1e6:.2f} ms  trace span {span/1e6:.2f} ms", "",
             "| kernel | calls | total ms | avg us | % busy |", "|---|---|---|---|---|"]
    for n, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        lines.append(f"| {n[:90]} | {c} | {t/1e6:.3f} | {t/c/1e3:.2f} | {100*t/busy:.1f} |")
    # gap analysis inside the densest 20 ms window (decode loop): busy fraction
    gaps = [rows[i + 1][0] - rows[i][1] for i in range(len(rows) - 1)]
    small = [g for g in gaps if 0 <= g < 50_000]
    if small:
        small.sort()
        lines += ["", f"inter-kernel gaps < 50 us: n={len(small)} median {small[len(small)//2]/1e3:.2f} us "
                      f"mean {sum(small)/len(small)/1e3:.2f} us total {sum(small)/1e6:.2f} ms"]
    txt = "\n".join(lines)
    print(txt)
    if out:
        open(out, "w").write(txt + "\n")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else None)
