"""Achieved K/V-cache bandwidth of the many-sequence attention kernel against the key count (923 sequences, 16-bit cache with 128
slots per (sequence, head): at n keys a launch reads the first n * 128 B of every 16-KiB region of the K and of the V plane).
Usage: python tools/attn_keys_sweep.py [S=923]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rgrg_amd  # noqa: E402
from rgrg_amd import synth  # noqa: E402

S = int(sys.argv[1]) if len(sys.argv) > 1 else 923
model = rgrg_amd.ReportGenerationModel(pretrain_without_lm_model=True)
model.load_state_dict(synth.make_state_dict(0, "bench"))
model.to("cuda:0").eval()
feats = torch.randn((S, 1024), generator=torch.Generator().manual_seed(99)).to("cuda:0")
with torch.autocast("cuda", dtype=torch.bfloat16):
    model.language_model.generate(feats, max_length=128)
torch.cuda.synchronize()
eng = model.language_model.engine()
for nkeys in (9, 17, 33, 49, 65, 73, 81, 97, 113, 128):
    p = eng.time_step_parts(S, nkeys, iters=10, one_range=True)
    print(f"keys {nkeys:4d}: {p['ms_attn'] / 24 * 1e3:6.1f} us per launch, {p['kv_bytes'] / 24 / 1e6:6.1f} MB -> {p['kv_bytes'] / (p['ms_attn'] * 1e-3) / 1e12:5.2f} TB/s", flush=True)
