"""Batch-1 decode (29 region rows, fp32, 127 steps) per mode of the persistent decode kernel (csrc/persistent.inc;
RGRG_PERSISTENT: 0 launch chain, 1 c_fc' + mlp_proj per launch, 2 attn_proj' .. mlp_proj, 3 attention .. mlp_proj, 4 one launch
per layer, 5 one launch per step).  One process, one model;
the decoder is re-created per setting.  Prints per setting: ms per generate call (median / min), whether the token ids equal
the launch chain's, and the largest difference of the last step's logits.
Usage: python tools/persistent_sweep.py [modes=0,1,2,3,4,5,0] [runs=5] [S=29]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rgrg_amd  # noqa: E402
from rgrg_amd import synth  # noqa: E402


def main():
    settings = [int(s) for s in (sys.argv[1] if len(sys.argv) > 1 else "0,1,2,3,4,5,0").split(",")]
    runs = int(sys.argv[2]) if len(sys.argv) > 2 else 5
    S = int(sys.argv[3]) if len(sys.argv) > 3 else 29
    model = rgrg_amd.ReportGenerationModel(pretrain_without_lm_model=True)
    model.load_state_dict(synth.make_state_dict(0, "bench"))
    model.to("cuda:0").eval()
    eng = model.engine()
    g = torch.Generator().manual_seed(99)
    feats = torch.randn((S, 1024), generator=g).to("cuda:0")
    ref_ids = ref_logits = None
    for mode in settings:
        os.environ["RGRG_PERSISTENT"] = str(mode)
        eng.close()
        eng._decoder_caps = (0, 0)
        try:
            ids = None
            for _ in range(2):
                ids = eng.greedy_decode(feats, 128)
            torch.cuda.synchronize()
            ts = []
            for _ in range(runs):
                t0 = time.perf_counter()
                ids = eng.greedy_decode(feats, 128)
                torch.cuda.synchronize()
                ts.append((time.perf_counter() - t0) * 1e3)
            ts.sort()
            logits = eng.last_logits(S)
        except Exception as e:  # noqa: BLE001
            print(f"mode {mode}: FAILED {e}", flush=True)
            continue
        if ref_ids is None:
            ref_ids, ref_logits = ids.clone(), logits.clone()
        same = bool(torch.equal(ids, ref_ids))
        dl = float((logits - ref_logits).abs().max())
        print(f"mode {mode}: generate {ts[len(ts) // 2]:8.3f} ms (min {ts[0]:.3f})  ids equal launch chain: {same}  "
              f"max |logit diff| {dl:.3e}", flush=True)


if __name__ == "__main__":
    main()
