"""Timeline of ONE many-sequence decode step from a rocprofv3 --kernel-trace directory: the kernels between two consecutive
argmax_update_kernel launches in the middle of the run, per hardware queue, with start / duration / gap to the previous kernel of
the same queue - where the row-range chains of the step (decoder.hip run_row_ranges) wait, and for what.
Usage: step_timeline.py <dir> [out.md] [--step N] [--full]"""
import csv
import glob
import sys
from collections import defaultdict


def short(n):
    n = n.split("(")[0].replace("void ", "").replace("rgrg::", "")
    return n[:60]


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    full = "--full" in sys.argv
    step = None
    for i, a in enumerate(sys.argv):
        if a == "--step":
            step = int(sys.argv[i + 1])
            args = [x for x in args if x != sys.argv[i + 1]]
    d = args[0]
    out = args[1] if len(args) > 1 else None
    files = glob.glob(f"{d}/**/*kernel_trace.csv", recursive=True)
    assert files, f"no kernel_trace.csv under {d}"
    rows = []
    for f in files:
        with open(f) as fh:
            for r in csv.DictReader(fh):
                rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "?"), r.get("Grid_Size", "?"),
                             r.get("Workgroup_Size", "?")))
    rows.sort()
    marks = [i for i, r in enumerate(rows) if "argmax_update_kernel" in r[2]]
    assert len(marks) > 4, "no decode steps in the trace"
    k = step if step is not None else len(marks) // 2
    lo, hi = marks[k] + 1, marks[k + 1] + 1
    win = rows[lo:hi]
    t0 = rows[marks[k]][1]
    t1 = win[-1][1]
    lines = [f"decode step {k} of {len(marks)}: {len(win)} kernels, {(t1 - t0) / 1e3:.1f} us from the previous arg-max's end to this one's end", ""]
    perq = defaultdict(list)
    for s, e, n, q, g, w in win:
        perq[q].append((s, e, n, g, w))
    # overall busy (union of intervals) and summed kernel time
    ev = sorted((s, e) for s, e, *_ in win)
    union, cs, ce = 0, ev[0][0], ev[0][1]
    for s, e in ev[1:]:
        if s > ce:
            union += ce - cs
            cs, ce = s, e
        else:
            ce = max(ce, e)
    union += ce - cs
    tot = sum(e - s for s, e, *_ in win)
    lines.append(f"sum of kernel durations {tot / 1e3:.1f} us, some kernel running {union / 1e3:.1f} us ({100 * union / (t1 - t0):.0f} % of the step), "
                 f"mean concurrency {tot / union:.2f}")
    agg = defaultdict(lambda: [0, 0])
    for s, e, n, *_ in win:
        agg[short(n)][0] += 1
        agg[short(n)][1] += e - s
    lines += ["", "| kernel | calls | total us | avg us |", "|---|---|---|---|"]
    for n, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        lines.append(f"| {n} | {c} | {t / 1e3:.1f} | {t / c / 1e3:.2f} |")
    for q, lst in sorted(perq.items()):
        busy = sum(e - s for s, e, *_ in lst)
        gaps = [lst[i + 1][0] - lst[i][1] for i in range(len(lst) - 1)]
        lines += ["", f"queue {q}: {len(lst)} kernels, busy {busy / 1e3:.1f} us, gaps between its consecutive kernels: total {sum(gaps) / 1e3:.1f} us, "
                      f"mean {sum(gaps) / max(1, len(gaps)) / 1e3:.2f} us, max {max(gaps or [0]) / 1e3:.2f} us"]
        if full:
            prev = None
            for s, e, n, g, w in lst[:80]:
                gap = (s - prev) / 1e3 if prev is not None else 0.0
                lines.append(f"  +{(s - t0) / 1e3:8.1f} us  dur {(e - s) / 1e3:6.2f}  gap {gap:6.2f}  {short(n)}  grid {g}/{w}")
                prev = e
    txt = "\n".join(lines)
    print(txt)
    if out:
        open(out, "w").write(txt + "\n")


if __name__ == "__main__":
    main()
