"""The reference scripts' mode for a kernel trace: 1 image, 4 beams, max_length 300, early stopping, fp16 autocast - one warm-up
call of 8 tokens and ONE full call (tools/prof_summary.py reads the rocprofv3 --kernel-trace CSVs)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rgrg_amd  # noqa: E402
from rgrg_amd import synth  # noqa: E402

m = rgrg_amd.ReportGenerationModel(True)
m.load_state_dict(synth.make_state_dict(0, "bench"))
m.to("cuda:0").eval()
img = synth.make_images(1, 1234).cuda()
with torch.no_grad(), torch.autocast("cuda", dtype=torch.float16):
    m.generate(img, max_length=8, num_beams=4, early_stopping=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out = m.generate(img, max_length=300, num_beams=4, early_stopping=True)
    torch.cuda.synchronize()
print(f"beam4 fp16: {(time.perf_counter() - t0) * 1e3:.1f} ms, ids {tuple(out[0].shape)}")
