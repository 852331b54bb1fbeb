"""One training step at the shape of BASELINE configs[4] (8 images per GPU, object detector frozen): forward() in
train mode (detector inference + both classifier losses + teacher-forced LM loss), backward on the HIP kernels, HIP AdamW
over the 53.66 M trainable values.  Usage: python tools/train_bench.py [B] [T] [steps] [fp32|bf16|both]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rgrg_amd  # noqa: E402
from rgrg_amd import optim, synth  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
T = int(sys.argv[2]) if len(sys.argv) > 2 else 64
STEPS = int(sys.argv[3]) if len(sys.argv) > 3 else 3
MODES = sys.argv[4] if len(sys.argv) > 4 else "both"
dev = torch.device("cuda", 0)
m = rgrg_amd.ReportGenerationModel(pretrain_without_lm_model=False)
m.load_state_dict(synth.make_state_dict(0, "bench"))
m.to(dev).train()
g = torch.Generator().manual_seed(0)
S = 29 * B
ids = torch.randint(0, 50257, (S, T), generator=g).to(dev)
lens = torch.randint(T // 2, T + 1, (S,), generator=g)
am = (torch.arange(T)[None, :] < lens[:, None]).to(torch.int64).to(dev)
images = synth.make_images(B, 1234).to(dev)
has = torch.ones((B, 29), dtype=torch.bool, device=dev)
abn = (torch.rand((B, 29), generator=g) < 0.2).to(dev)
opt = optim.AdamW(m.trainable_parameters(), lr=5e-5)
FWD_FLOP_PER_TOKEN = 2 * 353.453e6  # SURVEY 8(d); the backward here is activation gradients only (~1x forward)


def step(low):
    opt.zero_grad()
    if low:
        with torch.autocast("cuda", dtype=torch.bfloat16):
            out = m(images, None, ids.clone(), am, has, abn)
    else:
        out = m(images, None, ids.clone(), am, has, abn)
    total = 5.0 * out[1] + 5.0 * out[2] + 2.0 * out[3]
    total.backward()
    opt.step()
    return [o.item() for o in out[1:]]


for name, low in (("fp32", False), ("bf16 autocast", True)):
    if MODES != "both" and MODES != name.split()[0]:
        continue
    losses = step(low)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(STEPS):
        losses = step(low)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / STEPS
    tok = S * T
    print(f"train step [{name}] B={B} S={S} T={T}: {dt * 1e3:.1f} ms/step = {B / dt:.1f} images/s = {tok / dt / 1e3:.1f} k tokens/s; "
          f"decoder fwd+bwd GEMM work {2 * tok * FWD_FLOP_PER_TOKEN / dt / 1e12:.1f} TFLOP/s; losses {losses}", flush=True)
