"""Times the detector's validation-loss bookkeeping kernels (csrc/det_train.hip) at the reference's sizes: the RPN side
(match 163 840 anchors -> sample 256 -> loss) and the RoI side (add_gt -> match -> sample 512 -> gather) per batch size.
HIP-event timing of whole chains on the engine's stream; prints one line per case."""
import sys
import os

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rgrg_amd import synth  # noqa: E402
from rgrg_amd.report_generation_model import ReportGenerationModel  # noqa: E402

DEV = "cuda:0"


def timed(fn, iters=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3


def main():
    m = ReportGenerationModel()
    m.load_state_dict(synth.make_state_dict(0, "bench"))
    m = m.to(DEV).eval()
    eng = m.engine()
    g = torch.Generator().manual_seed(0)
    for B in (1, 8, 32):
        G = 29
        gxy = torch.rand((B, G, 2), generator=g) * 350
        gt = torch.cat([gxy, gxy + 40 + torch.rand((B, G, 2), generator=g) * 120], 2).to(DEV).contiguous()
        gcount = torch.full((B,), G, dtype=torch.int32, device=DEV)
        gl = torch.randint(1, 30, (B, G), generator=g).to(DEV)
        A = eng.anchors.shape[0]
        head = torch.randn((B, 32, 32, eng.num_anchors * 5), generator=g).to(DEV)
        xy = torch.rand((B, 1000, 2), generator=g) * 400
        props = torch.cat([xy, xy + 20 + torch.rand((B, 1000, 2), generator=g) * 100], 2).to(DEV).contiguous()
        counts = torch.full((B,), 1000, dtype=torch.int32, device=DEV)
        mm = eng._match(gt, gcount, eng.anchors, 0, None, A, 0.7, 0.3, True)
        t_match = timed(lambda: eng._match(gt, gcount, eng.anchors, 0, None, A, 0.7, 0.3, True))
        t_sample = timed(lambda: eng._sample("rpn", mm, None, None, 256, 128, None))
        t_rpn = timed(lambda: eng._rpn_losses(head, gt, gcount, None))
        t_roi = timed(lambda: eng._select_training_samples(props, counts, gt, gcount, gl, None))
        print(f"B={B:3d}  rpn: match {t_match:7.1f} us, sample(163840 -> 256) {t_sample:7.1f} us, match+sample+loss {t_rpn:7.1f} us"
              f" | roi: add_gt+match+sample(1029 -> 512)+gather {t_roi:7.1f} us")


if __name__ == "__main__":
    main()
