"""Round 6 experiment (negative, kernel side removed again - the diff is kept as profiles/r06_attn_confine_experiment.patch, the
results as profiles/r06_attn_confine_experiment.log): does CONFINING the HBM-bound attention launches of the many-sequence decode step to a fixed subset of the
CUs (RGRG_ATTN_CONFINE="shift,mask,limit": workgroups whose (HW_ID >> shift) & mask >= limit leave, the rest take (sequence, head)
items from a device-wide ticket) let the 16-bit GEMMs of the other row ranges run undisturbed beside it?  (Unconfined, the GEMM chain
runs x1.99 slower beside attention and the step is the SUM of its kernels: profiles/r06_overlap_probe*.log.)
Without arguments: prints the HW_ID / XCC_ID placement histogram of 4096 one-wave workgroups (which bits number the CUs).
With `step`: times generate() at 923 rows (ms per decode step) under the current environment."""
import collections
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rgrg_amd import _hip  # noqa: E402


def placement():
    lib = _hip.load()
    n = 4096
    out = torch.zeros((n, 2), dtype=torch.int32, device="cuda")
    for _ in range(2):
        _hip.check(lib.rgrg_debug_hw_ids(out.data_ptr(), n, torch.cuda.current_stream().cuda_stream))
    torch.cuda.synchronize()
    o = out.cpu().numpy().astype("uint32")
    hw, xcc = o[:, 0], o[:, 1] & 0xf
    fields = {"wave[3:0]": (0, 0xf), "simd[5:4]": (4, 3), "pipe[7:6]": (6, 3), "cu[11:8]": (8, 0xf), "sh[12]": (12, 1), "se[15:13]": (13, 7),
              "tg[19:16]": (16, 0xf), "vm[23:20]": (20, 0xf), "queue[26:24]": (24, 7), "state[29:27]": (27, 7), "me[31:30]": (30, 3)}
    for k, (sh, m) in fields.items():
        c = collections.Counter(((hw >> sh) & m).tolist())
        print(f"{k:14s}", dict(sorted(c.items())))
    print("xcc           ", dict(sorted(collections.Counter(xcc.tolist()).items())))
    cus = collections.Counter(zip(xcc.tolist(), ((hw >> 13) & 7).tolist(), ((hw >> 12) & 1).tolist(), ((hw >> 8) & 0xf).tolist()))
    print(f"distinct (xcc, se, sh, cu): {len(cus)}; per xcc: {dict(sorted(collections.Counter(k[0] for k in cus).items()))}")
    print("(se, sh, cu) of xcc 0:", sorted(k[1:] for k in cus if k[0] == 0))


def step():
    import rgrg_amd
    from rgrg_amd import synth
    model = rgrg_amd.ReportGenerationModel(pretrain_without_lm_model=True)
    model.load_state_dict(synth.make_state_dict(0, "bench"))
    model.to("cuda:0").eval()
    S = 923
    feats = torch.randn((S, 1024), generator=torch.Generator().manual_seed(99)).to("cuda:0")
    with torch.autocast("cuda", dtype=torch.bfloat16):
        ids0 = model.language_model.generate(feats, max_length=128)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            ids = model.language_model.generate(feats, max_length=128)
        torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) * 1e3 / 3
    print(f"RGRG_ATTN_CONFINE={os.environ.get('RGRG_ATTN_CONFINE', '-')} RGRG_DECODE_CHAINS={os.environ.get('RGRG_DECODE_CHAINS', '-')}: "
          f"{ms:.1f} ms per generate, {ms / 127:.3f} ms per step; checksum {int(ids.sum())} (same as first call: {bool(torch.equal(ids, ids0))})", flush=True)
    eng = model.language_model.engine()
    p = eng.time_step_parts(S, 65, iters=10, one_range=True)
    print(f"   alone, one range: GEMMs {p['ms_gemm']:.3f} ms, attention {p['ms_attn']:.3f} ms per step")


if __name__ == "__main__":
    step() if "step" in sys.argv else placement()
