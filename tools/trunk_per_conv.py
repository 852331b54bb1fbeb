"""Per-convolution table of the ResNet-50 trunk under autocast: the 52 implicit-GEMM convolutions of the 16
bottlenecks (+ the fp32 stem) at batch B in the 16-bit NHWC flow, each timed alone with HIP events on its launch stream on the
operands it sees in the trunk: M = B * OH * OW, N = Cout, K = KH * KW * Cin, algorithmic bytes (16-bit input read once + weights +
output [+ residual]), flops, which roofline bounds it (the ridge of the 16-bit matrix core against HBM is 2500 TF/s / 8 TB/s =
312 flop/B), microseconds, achieved TF/s or GB/s, and the fraction of ITS bound.
Usage: python tools/trunk_per_conv.py [batch=32] [out.md]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rgrg_amd  # noqa: E402
from rgrg_amd import _hip, synth  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
OUT = sys.argv[2] if len(sys.argv) > 2 else None
PEAK_TF, PEAK_GB = 2500.0, 8000.0


def timed(fn, iters=20):
    for _ in range(3):
        o = fn()
        del o
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        o = fn()
        del o
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters


def main():
    m = rgrg_amd.ReportGenerationModel(True)
    m.load_state_dict(synth.make_state_dict(0, "bench"))
    m.to("cuda:0").eval()
    eng = m.engine()
    images = synth.make_images(B, 1234).cuda()
    rows = []
    # the fp32 front end, for completeness (one input channel: ~1 % of the flops)
    H = W = 512
    x = images.reshape(B, H, W).contiguous()
    y = torch.empty((B, H // 2, W // 2, 64), dtype=torch.float32, device="cuda")
    st = lambda: torch.cuda.current_stream().cuda_stream  # noqa: E731
    us = timed(lambda: _hip.check(eng.lib.rgrg_stem_conv7x7_f32(_hip.ptr(x), _hip.ptr(eng.stem_w), _hip.ptr(eng.stem_scale), _hip.ptr(eng.stem_shift),
                                                               _hip.ptr(y), B, H, W, st())))
    fl, by = 2.0 * B * 256 * 256 * 64 * 49, B * (512 * 512 * 4 + 256 * 256 * 64 * 4)
    rows.append(("stem conv1 7x7/2 (fp32, direct)", B * 256 * 256, 64, 49, fl, by, us))
    p = torch.empty((B, H // 4, W // 4, 64), dtype=torch.float32, device="cuda")
    _hip.check(eng.lib.rgrg_stem_conv7x7_f32(_hip.ptr(x), _hip.ptr(eng.stem_w), _hip.ptr(eng.stem_scale), _hip.ptr(eng.stem_shift), _hip.ptr(y), B, H, W, st()))
    us = timed(lambda: _hip.check(eng.lib.rgrg_maxpool3x3s2_nhwc_f32(_hip.ptr(y), _hip.ptr(p), B, H // 2, W // 2, 64, st())))
    rows.append(("maxpool 3x3/2 (fp32)", B * 128 * 128, 64, 9, 0.0, B * (256 * 256 * 64 * 4 + 128 * 128 * 64 * 4), us))
    x16 = eng._act16(p.shape)
    _hip.check(eng.lib.rgrg_f32_to_bf16(_hip.ptr(p), _hip.ptr(x16), p.numel(), 0, st()))
    del y, p
    names = []
    for li, n in enumerate((3, 4, 6, 3), start=1):
        names += [f"layer{li}.{i}" for i in range(n)]
    for name, blk in zip(names, eng.blocks):
        def conv_row(tag, src, spec, act, residual=None):
            o = eng.conv16(src, spec, act, residual16=residual)
            Bn, OH, OW, Cout = o.shape
            Cin = src.shape[-1]
            M, N, K = Bn * OH * OW, Cout, spec.kh * spec.kw * Cin
            us = timed(lambda: eng.conv16(src, spec, act, residual16=residual))
            fl = 2.0 * M * N * K
            by = src.numel() * 2 + N * K * 2 + M * N * 2 + (M * N * 2 if residual is not None else 0)
            rows.append((f"{name}.{tag} {spec.kh}x{spec.kw} {Cin}->{Cout}" + (" /2" if spec.stride == 2 else "") + (" +res" if residual is not None else ""),
                         M, N, K, fl, by, us))
            return o
        o = conv_row("conv1", x16, blk["c1"], _hip.ACT_RELU)
        o = conv_row("conv2", o, blk["c2"], _hip.ACT_RELU)
        idt = conv_row("downsample", x16, blk["ds"], _hip.ACT_NONE) if "ds" in blk else x16
        x16 = conv_row("conv3", o, blk["c3"], _hip.ACT_RELU, residual=idt)
    ridge = PEAK_TF * 1e12 / (PEAK_GB * 1e9)
    lines = [f"ResNet-50 trunk under bf16 autocast, batch {B}, one MI355X; every convolution alone between HIP events (20 launches after 3 warm-ups), "
             f"operands as in the trunk.  Bound: arithmetic intensity (flops / algorithmic bytes) against the ridge {ridge:.0f} flop/B "
             f"(2500 TF/s dense 16-bit MFMA, 8 TB/s HBM).", "",
             "| conv | M | N | K | GFLOP | MB | flop/B | bound | us | achieved | frac of its bound | us at its bound |", "|---|---|---|---|---|---|---|---|---|---|---|---|"]
    tot_us = tot_floor = 0.0
    agg = {"mfma": [0.0, 0.0], "hbm": [0.0, 0.0]}
    for name, M, N, K, fl, by, us in rows:
        ai = fl / by if by else 0.0
        bound = "mfma" if ai >= ridge else "hbm"
        floor = max(fl / (PEAK_TF * 1e12), by / (PEAK_GB * 1e9)) * 1e6
        if bound == "mfma":
            ach, frac = f"{fl / (us * 1e-6) / 1e12:.0f} TF/s", fl / (us * 1e-6) / 1e12 / PEAK_TF
        else:
            ach, frac = f"{by / (us * 1e-6) / 1e9:.0f} GB/s", by / (us * 1e-6) / 1e9 / PEAK_GB
        tot_us += us
        tot_floor += floor
        agg[bound][0] += us
        agg[bound][1] += floor
        lines.append(f"| {name} | {M} | {N} | {K} | {fl / 1e9:.2f} | {by / 1e6:.1f} | {ai:.0f} | {bound} | {us:.1f} | {ach} | {frac:.3f} | {floor:.1f} |")
    lines += ["", f"sum of the launches {tot_us / 1e3:.3f} ms; at their own bounds {tot_floor / 1e3:.3f} ms ({tot_floor / tot_us:.3f}); "
                  f"MFMA-bound convs {agg['mfma'][0] / 1e3:.3f} ms (floor {agg['mfma'][1] / 1e3:.3f}), HBM-bound ones {agg['hbm'][0] / 1e3:.3f} ms (floor {agg['hbm'][1] / 1e3:.3f})"]
    us_trunk = timed(lambda: eng.backbone16(images), iters=8)
    lines.append(f"whole trunk call (backbone16, incl. the 16-bit <-> fp32 conversions at its ends): {us_trunk / 1e3:.3f} ms")
    txt = "\n".join(lines)
    print(txt)
    if OUT:
        open(OUT, "w").write(txt + "\n")


if __name__ == "__main__":
    main()
