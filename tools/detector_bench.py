"""Per-stage timing of the detector kernels with HIP events on the stream the kernels
are launched on, against their rooflines (fp32 MFMA 157.3 TF/s, bf16 MFMA 2500 TF/s, HBM 8 TB/s).
Usage: detector_bench.py [batch] [bf16]     (bf16: the path torch.autocast selects - bottlenecks / RPN / fc6 on the bf16 matrix core)"""
import math
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rgrg_amd  # noqa: E402
from rgrg_amd import _hip, synth  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
LOW = len(sys.argv) > 2 and sys.argv[2] == "bf16"
m = rgrg_amd.ReportGenerationModel(True)
m.load_state_dict(synth.make_state_dict(0, "bench"))
m.to("cuda:0").eval()
eng = m.engine()
images = synth.make_images(B, 1234).cuda()


def timed(fn, iters=5):
    """Mean ms per call.  Three warm-up calls first, every result dropped before the next call: round 1 kept the previous
    result alive across iterations, so at batch 8 every iteration of the trunk (~1 GB of intermediate activations) missed
    torch's caching allocator and paid hipMalloc + a device synchronisation - the "22 ms / 0.096 of peak" trunk figure of
    profiles/r01_detector_stages_b8.md was that artefact (the whole detector took 26.8 ms in the same run, fc6 alone 16.9)."""
    out = None
    for _ in range(3):
        out = None
        out = fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        out = None
        out = fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters, out


rows = []
st = lambda: torch.cuda.current_stream().cuda_stream  # noqa: E731
if LOW:
    PEAK = 2500.0
    ms, (feat16, feat) = timed(lambda: eng.backbone16(images))
    rows.append(("ResNet-50 trunk (fp32 stem + 52 bf16 implicit-GEMM convs, BN/ReLU/residual fused)", ms, 2 * 20.942e9 * B / (ms * 1e-3) / 1e12, "TF/s", PEAK))
    ms, t = timed(lambda: eng.conv16(feat16, eng.rpn_conv, _hip.ACT_RELU))
    rows.append(("RPN 3x3 conv 2048->2048 (+bias, ReLU), bf16", ms, 2 * 256 * 2048 * 18432 * B / (ms * 1e-3) / 1e12, "TF/s", PEAK))
    ms, head = timed(lambda: eng.conv16(t, eng.rpn_head, _hip.ACT_NONE, out_f32=True))
    rows.append(("RPN 1x1 heads fused N=800, bf16", ms, 2 * 256 * 800 * 2048 * B / (ms * 1e-3) / 1e12, "TF/s", PEAK))
    ms, (props, counts, offsets) = timed(lambda: eng.rpn(feat, feat16=feat16))
else:
    PEAK = 157.3
    ms, feat = timed(lambda: eng.backbone(images))
    rows.append(("ResNet-50 trunk (53 convs, NHWC implicit GEMM + BN/ReLU/residual)", ms, 2 * 20.942e9 * B / (ms * 1e-3) / 1e12, "TF/s", PEAK))
    ms, t = timed(lambda: eng.conv(feat, eng.rpn_conv, _hip.ACT_RELU))
    rows.append(("RPN 3x3 conv 2048->2048 (+bias, ReLU)", ms, 2 * 256 * 2048 * 18432 * B / (ms * 1e-3) / 1e12, "TF/s", PEAK))
    ms, head = timed(lambda: eng.conv(t, eng.rpn_head, _hip.ACT_NONE))
    rows.append(("RPN 1x1 heads fused N=800", ms, 2 * 256 * 800 * 2048 * B / (ms * 1e-3) / 1e12, "TF/s", PEAK))
    ms, (props, counts, offsets) = timed(lambda: eng.rpn(feat))
rows.append(("RPN head + proposals (top-1000, decode, NMS) per call", ms, float("nan"), "-", float("nan")))
R = int(offsets[-1].item())
Cf = feat.shape[-1]
maps = torch.empty((R, 64, Cf), dtype=torch.int16 if LOW else torch.float32, device="cuda")
pooled = torch.empty((R, Cf), device="cuda")


def roi():
    fn = eng.lib.rgrg_roi_align_avgpool_bf16maps if LOW else eng.lib.rgrg_roi_align_avgpool_f32
    _hip.check(fn(feat.data_ptr(), props.data_ptr(), offsets.data_ptr(), maps.data_ptr(), pooled.data_ptr(), B, 16, 16, Cf, 1000, R, 1.0 / 32,
                  *([0] if LOW else []), st()))


ms, _ = timed(roi)
bytes_roi = B * 2048 * 256 * 4 + R * (2048 * 64 * (2 if LOW else 4) + 2048 * 4)
rows.append((f"RoIAlign 8x8 + avgpool ({R} RoIs, {'bf16' if LOW else 'fp32'} maps)", ms, bytes_roi / (ms * 1e-3) / 1e9, "GB/s", 8000.0))
if LOW:
    h6 = torch.empty((R, 1024), device="cuda")
    wb = eng._fc6_bf16()

    def fc6():
        _hip.check(eng.lib.rgrg_linear_bf16_f32(maps.data_ptr(), wb.data_ptr(), eng.fc6_b.data_ptr(), None, h6.data_ptr(), None, R, 1024, 64 * Cf, 1024,
                                                _hip.ACT_RELU, 0, st()))
    ms, _ = timed(fc6)
else:
    x6 = maps.view(R, 64 * Cf)
    ms, h6 = timed(lambda: eng.linear(x6, eng.fc6_w, eng.fc6_b, _hip.ACT_RELU))
rows.append((f"fc6 [{R}x131072]x[131072x1024] (+bias, ReLU)", ms, 2 * R * 131072 * 1024 / (ms * 1e-3) / 1e12, "TF/s", PEAK))
ms, h7 = timed(lambda: eng.linear(h6, eng.fc7_w, eng.fc7_b, _hip.ACT_RELU))
rows.append(("fc7 1024->1024 (fp32)", ms, 2 * R * 1024 * 1024 / (ms * 1e-3) / 1e12, "TF/s", 157.3))
ms, _ = timed(lambda: eng.detect(images, bf16=LOW))
rows.append(("whole detector (detect(): trunk..dim_reduction, 1 host sync)", ms, float("nan"), "-", float("nan")))
print(f"batch {B}, {R} proposals after NMS, {'bf16 (autocast) path' if LOW else 'fp32 path'}")
print("| stage | ms | achieved | unit | peak | frac |")
print("|---|---|---|---|---|---|")
for name, ms, ach, unit, peak in rows:
    frac = ach / peak if not math.isnan(ach) else float("nan")
    print(f"| {name} | {ms:.3f} | {ach:.1f} | {unit} | {peak} | {frac:.3f} |")
