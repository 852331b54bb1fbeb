"""Time full report generation with the reference's shipped settings (num_beams=4, max_length=300,
early_stopping=True; generate_reports_for_images.py:27-28,108-114) on the synthetic bench image."""
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
for nb, L in ((4, 64), (4, 300), (1, 300)):
    m.generate(img, max_length=8, num_beams=nb, early_stopping=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out = m.generate(img, max_length=L, num_beams=nb, early_stopping=True)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"num_beams={nb} max_length={L}: {dt*1e3:.1f} ms, ids {tuple(out[0].shape)}, {dt/(out[0].shape[1]-1)*1e6:.0f} us/step", flush=True)
