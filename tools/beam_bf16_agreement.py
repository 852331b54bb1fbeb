"""Agreement of beam search under the opt-in bf16 path with the fp32 path (40 items x 4 beams = 160 rows)."""
import torch, sys
sys.path.insert(0, __import__("os").path.join(__import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))), "tests"))
from test_gpu_generate import gpu_model, DEV
m = gpu_model("ragged")
g = torch.Generator().manual_seed(5)
feats = torch.randn((40, 1024), generator=g).to(DEV)
a = m.language_model.generate(feats, max_length=14, num_beams=4, early_stopping=False)
with torch.autocast("cuda", dtype=torch.bfloat16):
    b = m.language_model.generate(feats, max_length=14, num_beams=4, early_stopping=False)
L = min(a.shape[1], b.shape[1])
print("beam shapes", a.shape, b.shape, "token agree", (a[:, :L] == b[:, :L]).float().mean().item(), "rows equal", (a[:, :L] == b[:, :L]).all(1).float().mean().item())
