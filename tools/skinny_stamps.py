"""Phase stamps of the five kernels of a batch-1 (29-sequence) decode layer and of lm_head' (csrc/skinny_direct.inc, SKS macros):
where the 6-9 us of each launch go.  Thread 0 of every workgroup stamps the 100 MHz real-time clock (s_memrealtime, common to
all XCDs; 10 ns resolution) at

    0 first instruction | 1 activation streams landed (s_waitcnt vmcnt(8): only the 8 weight chunks outstanding) | 2 first weight
    chunk landed (vmcnt(7)) | 3 wave 0's MFMA chain complete | 4 every wave at the reduction barrier | 5 reduction + epilogue
    done, stores issued | 7 stores acknowledged (vmcnt(0))

(attention: 1 = keys consumed, 3 = partial sums in LDS, 4 = barrier, 5 = output store issued; lm_head': 1 = rows staged +
statistics, 2 = first 8 KiB weight group consumed, 3 = first tile finished, 5 = wave 0's last tile finished).  The stamps only
exist in a measurement build (-DRGRG_SKINNY_STAMPS): they stay in registers until the kernel's end, and the two explicit waits
in front of stamps 1 / 2 delay the epilogue-operand requests by one L2 latency - the tool prints the step time of both builds.

  python tools/skinny_stamps.py --build      (CPU container: compiles rgrg_amd/lib/librgrg_hip_stamps.so)
  python tools/skinny_stamps.py [out.md]     (GPU box: one 29-row greedy generate per build, the stamps of the LAST decode step)
"""
import os
import statistics
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
STAMP_LIB = os.path.join(REPO, "rgrg_amd", "lib", "librgrg_hip_stamps.so")
CHILD = r"""
import sys, time, torch
sys.path.insert(0, %r)
import rgrg_amd.build as b
if len(sys.argv) > 1:
    b.LIB_PATH = sys.argv[1]
    b.is_stale = lambda: False
import rgrg_amd
from rgrg_amd import synth
m = rgrg_amd.ReportGenerationModel(pretrain_without_lm_model=True)
m.load_state_dict(synth.make_state_dict(0, "bench"))
m.to("cuda:0").eval()
g = torch.Generator().manual_seed(99)
feats = torch.randn((29, 1024), generator=g).to("cuda:0")
eng = m.engine()
for _ in range(2):
    eng.greedy_decode(feats, 128)
torch.cuda.synchronize()
t = time.perf_counter()
for _ in range(3):
    eng.greedy_decode(feats, 128)
torch.cuda.synchronize()
print("MS_PER_GENERATE", (time.perf_counter() - t) / 3 * 1e3)
eng.close()
""" % REPO

PHASES = {
    "gemm": [(0, 1, "entry -> activation streams landed"), (1, 2, "-> first weight chunk landed (HBM)"),
             (2, 3, "-> wave 0's MFMA chain complete (8 chunks landed, 32-64 MFMAs)"), (3, 4, "-> every wave at the reduction barrier"),
             (4, 5, "-> 8-way LDS reduction + epilogue, stores issued"), (5, 7, "-> stores acknowledged")],
    "attention": [(0, 1, "entry -> q/k/v + cache rows landed, scores + weighted sums issued"), (1, 3, "-> partial sums written to LDS"),
                  (3, 4, "-> every wave at the merge barrier"), (4, 5, "-> 16-way merge, output store issued"), (5, 7, "-> stores acknowledged")],
    "lm_head'": [(0, 1, "entry -> 128 KiB of rows staged in LDS + statistics (first 16 KiB of weights requested at entry)"),
                 (1, 2, "-> first 8 KiB weight group consumed"), (2, 3, "-> first tile finished (512 MFMAs + epilogue)"),
                 (3, 5, "-> wave 0's last tile finished (~3 tiles per wave)"), (5, 7, "-> stores acknowledged")],
}


def build():
    sys.path.insert(0, REPO)
    import rgrg_amd.build as b
    os.environ["RGRG_HIPCC_FLAGS"] = (os.environ.get("RGRG_HIPCC_FLAGS", "") + " -DRGRG_SKINNY_STAMPS").strip()
    b.LIB_PATH = STAMP_LIB
    b.HASH_PATH = STAMP_LIB + ".srchash"
    print(b.build_library(force=True))
    os.remove(b.HASH_PATH)


def run_child(lib, trace):
    env = dict(os.environ)
    if trace:
        env["RGRG_SKINNY_TRACE"] = trace
    out = subprocess.run([sys.executable, "-c", CHILD] + ([lib] if lib else []), env=env, check=True, capture_output=True, text=True).stdout
    return float([ln for ln in out.splitlines() if ln.startswith("MS_PER_GENERATE")][0].split()[1])


def parse(path):
    slots = []
    with open(path) as f:
        for ln in f:
            if ln.startswith("slot"):
                _, _, name, wgs = ln.split()
                slots.append((name, []))
            else:
                r = [int(x) for x in ln.split()]
                if r[0]:   # workgroups that returned before their flush (lm_head' past the last tile) leave zeros
                    slots[-1][1].append(r)
    return slots


def us(ticks):
    return ticks / 100.0


def main():
    if "--build" in sys.argv:
        return build()
    out_md = sys.argv[1] if len(sys.argv) > 1 else os.path.join(REPO, "gpurun_out", "skinny_phase_stamps.md")
    if not os.path.exists(STAMP_LIB):
        raise SystemExit("build the measurement library first: python tools/skinny_stamps.py --build")
    trace = "/tmp/skinny_stamps.txt"
    ms_plain = run_child(None, None)
    ms_stamped = run_child(STAMP_LIB, trace)
    slots = parse(trace)
    L = []
    L.append("# Phase stamps of the batch-1 decode step (29 sequences, fp32, last step of a 128-token generate: 128 keys)\n")
    L.append(f"`tools/skinny_stamps.py`; 100 MHz `s_memrealtime` stamps of thread 0 of every workgroup, {len(slots)} launches of one hipGraph replay.  "
             f"One generate (127 steps + prefill): {ms_plain:.2f} ms with the product library, {ms_stamped:.2f} ms with the stamped build "
             f"(+{(ms_stamped / ms_plain - 1) * 100:.1f} %: the perturbation of the stamps).\n")
    t_first = min(r[0] for r in slots[0][1])
    t_last = max(r[7] for r in slots[-1][1])
    L.append(f"First workgroup start of the step's first launch -> last store acknowledgement of lm_head': {us(t_last - t_first):.1f} us "
             f"(+ argmax_update and the graph boundary = one step).\n")
    kinds = []
    for name, _ in slots:
        if name not in kinds:
            kinds.append(name)
    summary = []
    for kind in kinds:
        idx = [i for i, (n, _) in enumerate(slots) if n == kind]
        if kind == "c_attn'":
            idx = [i for i in idx if i != 0]   # layer 0 reads the embedding rows instead of the combine streams
        ph = PHASES["attention"] if kind == "attention" else PHASES["lm_head'"] if kind == "lm_head'" else PHASES["gemm"]
        bound, skew_med, skew_max, span, wg_med, wg_max, nwg = [], [], [], [], [], [], []
        seg = {p: [] for p in ph}
        for i in idx:
            rows = slots[i][1]
            s0 = min(r[0] for r in rows)
            if i > 0:
                bound.append(s0 - max(r[7] for r in slots[i - 1][1]))
            sk = [r[0] - s0 for r in rows]
            skew_med.append(statistics.median(sk)); skew_max.append(max(sk))
            span.append(max(r[7] for r in rows) - s0)
            tot = [r[7] - r[0] for r in rows]
            wg_med.append(statistics.median(tot)); wg_max.append(max(tot)); nwg.append(len(rows))
            for p in ph:
                seg[p].append(statistics.median([r[p[1]] - r[p[0]] for r in rows]))
        mean = lambda v: sum(v) / len(v) if v else float("nan")   # noqa: E731
        L.append(f"\n## {kind}  ({len(idx)} launches, {int(mean(nwg))} workgroups)\n")
        L.append("| segment | us (median over workgroups, mean over launches) |\n|---|---|")
        L.append(f"| previous launch's last store acknowledged -> first workgroup starts (launch boundary) | {us(mean(bound)):.2f} |")
        L.append(f"| workgroup start skew: median / last workgroup after the first | {us(mean(skew_med)):.2f} / {us(mean(skew_max)):.2f} |")
        for p in ph:
            L.append(f"| {p[2]} | {us(mean(seg[p])):.2f} |")
        L.append(f"| one workgroup, entry -> stores acknowledged: median / slowest | {us(mean(wg_med)):.2f} / {us(mean(wg_max)):.2f} |")
        L.append(f"| launch span (first start -> last acknowledgement) | {us(mean(span)):.2f} |")
        summary.append((kind, len(idx), us(mean(bound)), us(mean(span))))
    L.append("\n## Per-step sum\n")
    L.append("| kernel | launches | boundary us | span us | (boundary + span) x launches |\n|---|---|---|---|---|")
    tot = 0.0
    for kind, n, b, s in summary:
        n_all = n + 1 if kind == "c_attn'" else n
        L.append(f"| {kind} | {n_all} | {b:.2f} | {s:.2f} | {(b + s) * n_all:.1f} |")
        tot += (b + s) * n_all
    L.append(f"| sum | | | | {tot:.1f} |")
    text = "\n".join(L) + "\n"
    os.makedirs(os.path.dirname(out_md), exist_ok=True)
    with open(out_md, "w") as f:
        f.write(text)
    print(text)


if __name__ == "__main__":
    main()
