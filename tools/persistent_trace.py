"""Where the time of a persistent decode launch goes (csrc/persistent.inc, RGRG_PK_TRACE): thread 0 of each of the 256
workgroups stamps the 100 MHz real-time clock (s_memrealtime: common to all XCDs) at the start of the launch, then per seam: phase arithmetic done (wave 0) | its
write-through stores acknowledged | every wave of the workgroup at the seam | barrier released; and at the end of the launch.
Runs one 29-row generate per mode in a child process (the stamps of the LAST launch of the call are dumped when the decoder
is destroyed), prints per segment the mean / max over workgroups in microseconds (clock from the launch's own span).
Usage: python tools/persistent_trace.py [modes=1,2,3] [clock_mhz=100]"""
import os
import struct
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r"""
import sys, torch
sys.path.insert(0, %r)
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
eng.close()
""" % REPO

NAMES = {1: ["c_fc'", "mlp_proj"], 2: ["attn_proj'", "c_fc'", "mlp_proj"], 3: ["attention", "attn_proj'", "c_fc'", "mlp_proj"],
         4: ["c_attn'", "attention", "attn_proj'", "c_fc'", "mlp_proj"]}


def main():
    modes = [int(m) for m in (sys.argv[1] if len(sys.argv) > 1 else "1,2,3").split(",")]
    mhz = float(sys.argv[2]) if len(sys.argv) > 2 else 100.0
    for mode in modes:
        path = f"/tmp/pk_trace_{mode}.bin"
        env = dict(os.environ, RGRG_PERSISTENT=str(mode), RGRG_PK_TRACE=path)
        subprocess.run([sys.executable, "-c", CHILD], env=env, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        raw = open(path, "rb").read()
        st = struct.unpack(f"<{len(raw) // 8}Q", raw)
        nph = len(NAMES[mode])
        n = 1 + 4 * (nph - 1) + 1
        rows = [st[b * 64:b * 64 + n] for b in range(256)]
        t0 = min(r[0] for r in rows)
        us = lambda c: c / mhz  # noqa: E731
        print(f"== mode {mode}: one launch = {' | '.join(NAMES[mode])}; {n} stamps per workgroup; span {us(max(r[-1] for r in rows) - t0):.2f} us "
              f"(first start .. last end, at {mhz:.0f} MHz)")
        print(f"   launch skew: workgroups start {us(max(r[0] for r in rows) - t0):.2f} us apart")
        labels = []
        for i in range(nph - 1):
            labels += [f"{NAMES[mode][i]}: arithmetic (incl. operand waits)", "   stores acknowledged (vmcnt 0)", "   every wave of the workgroup drained",
                       "   arrive, prefetch issue, wait for the release"]
        labels.append(f"{NAMES[mode][-1]}: arithmetic (incl. operand waits)")
        for k, lab in enumerate(labels):
            seg = [r[k + 1] - r[k] for r in rows]
            act = [s for s in seg]
            print(f"   {lab:52s} mean {us(sum(act) / len(act)):6.2f}  min {us(min(act)):6.2f}  max {us(max(act)):6.2f} us")
        # time between the LAST workgroup's arrival and the release seen by the median workgroup, per barrier
        for i in range(nph - 1):
            arr = [r[1 + 4 * i + 2] for r in rows]
            rel = sorted(r[1 + 4 * i + 3] for r in rows)
            print(f"   barrier {i}: last arrival -> median release {us(rel[128] - max(arr)):5.2f} us; arrival spread {us(max(arr) - min(arr)):5.2f} us")


if __name__ == "__main__":
    main()
