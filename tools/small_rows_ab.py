"""Small decode batches under torch.autocast, end to end (detector + selection + decode): the scripts' own mode (1 image, 4 beams x
29 regions = 116 beam rows, fp16, max_length 300, early stopping - generate_reports_for_images.py:108-114) and greedy bf16 batches
of 2 and 4 images (58 / 115 rows).  33-128 decoder rows are the range between the one-row-tile fp32 fused plan (batch-1 greedy,
bit-exact) and the many-sequence 16-bit path; A/B the plans with RGRG_W16_FUSED=0/1 (decoder.hip decode_row_limit).
Usage: python tools/small_rows_ab.py [calls=3]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rgrg_amd  # noqa: E402
from rgrg_amd import synth  # noqa: E402

CALLS = int(sys.argv[1]) if len(sys.argv) > 1 else 3


def timed(fn):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(CALLS):
        out = fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) * 1e3 / CALLS, out


def main():
    model = rgrg_amd.ReportGenerationModel(pretrain_without_lm_model=False)
    model.load_state_dict(synth.make_state_dict(0, "bench"))
    model.to("cuda:0").eval()
    print(f"== RGRG_W16_FUSED={os.environ.get('RGRG_W16_FUSED', '(default 1)')} RGRG_SKINNY_MAX_ROWS_16={os.environ.get('RGRG_SKINNY_MAX_ROWS_16', '(default)')}")
    img1 = synth.make_images(1, 1234).cuda()
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.float16):
        ms, out = timed(lambda: model.generate(img1, max_length=300, num_beams=4, early_stopping=True))
    ids = out[0] if isinstance(out, tuple) else out
    print(f"beam4 fp16 1 image: {ms:.1f} ms, ids {tuple(ids.shape)}", flush=True)
    for b in (2, 4):
        imgs = synth.make_images(b, 1234).cuda()
        with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
            ms, out = timed(lambda: model.generate(imgs, max_length=128))
        ids = out[0] if isinstance(out, tuple) else out
        print(f"greedy bf16 batch {b}: {ms:.1f} ms = {b / ms * 1e3:.2f} images/s, rows {ids.shape[0]}", flush=True)


if __name__ == "__main__":
    main()
