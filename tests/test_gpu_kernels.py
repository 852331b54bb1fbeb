"""Parity of every C-ABI entry point (librgrg_hip.so, called through ctypes) against the
CPU oracle on identical seeded inputs.  Tolerances are written next to each check:
integer / index / mask outputs are bit-exact, fp32 outputs are compared with the stated
absolute+relative bound (different summation order of fp32 MFMA vs the CPU BLAS)."""
import ctypes as C
import math

import pytest
import torch
import torch.nn.functional as F

from conftest import gpu_model, synth_sd
from oracle import detector as o_det
from oracle import full_model as o_full
from oracle import tv013
from rgrg_amd import _hip, synth

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _stream():
    return torch.cuda.current_stream().cuda_stream


def close(a, b, rtol, atol_frac, what):
    """|a-b| <= atol_frac*max|b| + rtol*|b| elementwise; reports the worst offender."""
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    assert a.shape == b.shape, f"{what}: shape {tuple(a.shape)} vs {tuple(b.shape)}"
    tol = atol_frac * b.abs().max() + rtol * b.abs()
    err = (a - b).abs()
    bad = err > tol
    assert not bad.any(), (f"{what}: {int(bad.sum())}/{bad.numel()} out of tolerance, max abs err "
                           f"{err.max().item():.3e} (ref max {b.abs().max().item():.3e})")


@pytest.fixture(scope="module")
def eng():
    return gpu_model("bench").engine()


@pytest.fixture(scope="module")
def oracle_bench():
    """CPU oracle intermediates for two synthetic images (computed once)."""
    sd = synth_sd("bench")
    images = torch.cat([synth.make_images(1, 1234), synth.make_images(1, 77)], 0)
    out = o_det.object_detector_forward(sd, images, return_intermediates=True)
    obj, reg = tv013.rpn_head(sd, "object_detector.rpn.head.", out["_features"])
    out["_rpn_obj"], out["_rpn_reg"], out["_images"] = obj, reg, images
    return out


# ------------------------------------------------------------------------- GEMM / conv
@pytest.mark.parametrize("M,N,K,act,res,splitk", [
    (29, 1024, 1024, 0, True, 1), (823, 1024, 4096, 1, False, 4), (100, 150, 1024, 0, False, 1),
    (58, 1, 128, 0, False, 1), (300, 800, 2048, 2, False, 1), (1, 1024, 2048, 0, False, 2), (257, 129, 96, 1, True, 1)])
def test_linear_f32(eng, M, N, K, act, res, splitk):
    g = torch.Generator().manual_seed(M * 7 + N)
    A = torch.randn((M, K), generator=g)
    W = torch.randn((N, K), generator=g) / math.sqrt(K)
    b = torch.randn((N,), generator=g)
    R = torch.randn((M, N), generator=g) if res else None
    ref = A.double() @ W.double().t() + b.double()
    if res:
        ref = ref + R.double()
    ref = {0: lambda x: x, 1: F.relu, 2: lambda x: F.gelu(x, approximate="tanh")}[act](ref)
    y = eng.linear(A.to(DEV), W.to(DEV), b.to(DEV), act, R.to(DEV) if res else None, splitk=splitk)
    close(y, ref, 2e-5, 2e-6, f"linear {M}x{N}x{K}")


@pytest.mark.parametrize("B,H,Cin,Cout,k,stride,pad,res", [
    (2, 32, 64, 256, 1, 1, 0, True), (2, 16, 64, 64, 3, 1, 1, False), (1, 32, 128, 128, 3, 2, 1, False),
    (2, 16, 256, 512, 1, 2, 0, False), (1, 16, 2048, 160, 1, 1, 0, False)])
def test_conv2d_nhwc_f32(eng, B, H, Cin, Cout, k, stride, pad, res):
    from rgrg_amd.engine import _Conv
    g = torch.Generator().manual_seed(B * 100 + Cin + k)
    x = torch.randn((B, Cin, H, H), generator=g)
    w = torch.randn((Cout, Cin, k, k), generator=g) / math.sqrt(Cin * k * k)
    scale = torch.rand((Cout,), generator=g) + 0.5
    shift = torch.randn((Cout,), generator=g)
    ref = F.conv2d(x.double(), w.double(), stride=stride, padding=pad) * scale.double().view(1, -1, 1, 1) + shift.double().view(1, -1, 1, 1)
    R = None
    if res:
        R = torch.randn(ref.shape, generator=g)
        ref = ref + R.double()
    ref = F.relu(ref)
    conv = _Conv(w.to(DEV), scale.to(DEV), shift.to(DEV), stride, pad)
    y = eng.conv(x.permute(0, 2, 3, 1).contiguous().to(DEV), conv, _hip.ACT_RELU,
                 R.permute(0, 2, 3, 1).contiguous().to(DEV) if res else None)
    close(y.permute(0, 3, 1, 2), ref, 2e-5, 2e-6, f"conv k{k} s{stride}")


@pytest.mark.parametrize("B,H,Cin,Cout,k,stride,pad,res", [
    (2, 32, 64, 256, 1, 1, 0, True), (2, 16, 64, 64, 3, 1, 1, False), (1, 32, 128, 128, 3, 2, 1, False),
    (2, 16, 256, 512, 1, 2, 0, False), (1, 16, 2048, 160, 1, 1, 0, False), (3, 16, 2048, 2048, 3, 1, 1, False),
    (1, 24, 64, 64, 1, 1, 0, False), (2, 9, 128, 192, 3, 2, 1, True)])
def test_conv2d_nhwc_bf16(eng, B, H, Cin, Cout, k, stride, pad, res):
    """Implicit-GEMM convolution on the LDS-DMA bf16 kernel (the detector under torch.autocast): 1x1 / 3x3, stride 1 / 2,
    padding through the zero line in front of the image, K = 64 (a single K tile) up to 18 432 (the RPN conv), bf16 residual,
    ragged row / column tiles.  Against a float64 convolution of the SAME bf16-rounded operands (BatchNorm scale folded into
    the weights before rounding, as the engine does): fp32 output to summation-order accuracy, bf16 output to one rounding."""
    from rgrg_amd.engine import _Conv
    g = torch.Generator().manual_seed(B * 100 + Cin + k + Cout)
    x = torch.randn((B, Cin, H, H), generator=g).bfloat16()
    w = torch.randn((Cout, Cin, k, k), generator=g) / math.sqrt(Cin * k * k)
    scale = torch.rand((Cout,), generator=g) + 0.5
    shift = torch.randn((Cout,), generator=g)
    wf = (w * scale.view(-1, 1, 1, 1)).bfloat16()
    ref = F.conv2d(x.double(), wf.double(), stride=stride, padding=pad) + shift.double().view(1, -1, 1, 1)
    R = None
    if res:
        R = torch.randn(ref.shape, generator=g).bfloat16()
        ref = ref + R.double()
    ref = F.relu(ref)
    conv = _Conv(w.to(DEV), scale.to(DEV), shift.to(DEV), stride, pad)
    x16 = eng._act16((B, H, H, Cin))
    x16.copy_(x.permute(0, 2, 3, 1).contiguous().view(torch.int16).to(DEV))
    r16 = None
    if res:
        r16 = eng._act16(tuple(R.permute(0, 2, 3, 1).shape))
        r16.copy_(R.permute(0, 2, 3, 1).contiguous().view(torch.int16).to(DEV))
    y32 = eng.conv16(x16, conv, _hip.ACT_RELU, r16, out_f32=True)
    close(y32.permute(0, 3, 1, 2), ref, 2e-5, 4e-6, f"bf16 conv k{k} s{stride} (fp32 out)")
    y16 = eng.conv16(x16, conv, _hip.ACT_RELU, r16)
    got = y16.view(torch.bfloat16).float().permute(0, 3, 1, 2)
    close(got, ref, 2 ** -8, 4e-6, f"bf16 conv k{k} s{stride} (bf16 out)")


def test_stem_and_maxpool(eng):
    g = torch.Generator().manual_seed(3)
    x = torch.randn((2, 1, 64, 96), generator=g)
    w = torch.randn((64, 1, 7, 7), generator=g) / 7.0
    scale, shift = torch.rand((64,), generator=g) + 0.5, torch.randn((64,), generator=g)
    ref = F.relu(F.conv2d(x.double(), w.double(), stride=2, padding=3) * scale.double().view(1, -1, 1, 1) + shift.double().view(1, -1, 1, 1))
    y = torch.empty((2, 32, 48, 64), device=DEV)
    wt = w.reshape(64, 49).t().contiguous().to(DEV)
    xd, sc_d, sh_d = x.to(DEV), scale.to(DEV), shift.to(DEV)  # keep alive: the kernel reads them asynchronously
    _hip.check(eng.lib.rgrg_stem_conv7x7_f32(xd.data_ptr(), wt.data_ptr(), sc_d.data_ptr(), sh_d.data_ptr(), y.data_ptr(),
                                             2, 64, 96, _stream()))
    close(y.permute(0, 3, 1, 2), ref, 1e-5, 1e-6, "stem conv7x7+bn+relu")
    p = torch.empty((2, 16, 24, 64), device=DEV)
    _hip.check(eng.lib.rgrg_maxpool3x3s2_nhwc_f32(y.data_ptr(), p.data_ptr(), 2, 32, 48, 64, _stream()))
    refp = F.max_pool2d(y.permute(0, 3, 1, 2).cpu(), 3, 2, 1)
    assert torch.equal(p.permute(0, 3, 1, 2).cpu(), refp)  # max is exact


def test_backbone_vs_oracle(eng, oracle_bench):
    feat = eng.backbone(oracle_bench["_images"].to(DEV)).permute(0, 3, 1, 2)
    # 53 stacked convs in fp32 with a different summation order: 1e-4 of the map's max + 1e-3 relative
    close(feat, oracle_bench["_features"], 1e-3, 1e-4, "ResNet-50 trunk features")


# ------------------------------------------------------------------------- RPN proposals
def _run_proposals(eng, head, B):
    props = torch.empty((B, 1000, 4), device=DEV)
    counts = torch.empty((B,), dtype=torch.int32, device=DEV)
    offsets = torch.empty((B + 1,), dtype=torch.int32, device=DEV)
    _hip.check(eng.lib.rgrg_rpn_proposals_f32(head.data_ptr(), eng.anchors.data_ptr(), props.data_ptr(), counts.data_ptr(),
                                              offsets.data_ptr(), B, 256, 160, 1000, 1000, 0.7, 1e-3, 512.0, 512.0, _stream()))
    return props.cpu(), counts.cpu(), offsets.cpu()


def _head_nhwc(obj, reg):
    B = obj.shape[0]
    return torch.cat([obj.permute(0, 2, 3, 1).reshape(B, 256, 160), reg.permute(0, 2, 3, 1).reshape(B, 256, 640)], 2).contiguous()


def _oracle_proposals(obj, reg):
    B = obj.shape[0]
    anchors = tv013.grid_anchors((512, 512), (16, 16))
    objectness = tv013.permute_and_flatten(obj, 1).reshape(B, -1)
    deltas = tv013.permute_and_flatten(reg, 4).reshape(-1, 4)
    proposals = tv013.box_decode(deltas, anchors.repeat(B, 1), (1.0, 1.0, 1.0, 1.0)).view(B, -1, 4)
    return tv013.filter_proposals(proposals, objectness, (512, 512))[0]


def test_anchors_match_oracle(eng):
    assert torch.equal(eng.anchors.cpu(), tv013.grid_anchors((512, 512), (16, 16)))


def test_rpn_proposals_on_oracle_head(eng, oracle_bench):
    obj, reg = oracle_bench["_rpn_obj"], oracle_bench["_rpn_reg"]
    props, counts, offsets = _run_proposals(eng, _head_nhwc(obj, reg).to(DEV), 2)
    ref = oracle_bench["_proposals"]
    assert counts.tolist() == [p.shape[0] for p in ref]          # same survivors (integer: exact)
    assert offsets.tolist() == [0, ref[0].shape[0], ref[0].shape[0] + ref[1].shape[0]]
    for b in range(2):
        n = ref[b].shape[0]
        # coordinates go through expf (device vs host libm differ by <= 1 ulp): 1e-4 px absolute
        assert (props[b, :n] - ref[b]).abs().max() <= 1e-4, (props[b, :n] - ref[b]).abs().max()
        assert (props[b, n:] == 0).all()


def test_rpn_proposals_ties_small_boxes_and_heavy_overlap(eng):
    g = torch.Generator().manual_seed(11)
    obj = torch.zeros((1, 160, 16, 16))                       # massive ties: lower index must win
    obj.view(-1)[torch.randperm(40960, generator=g)[:300]] = 1.0
    reg = torch.randn((1, 640, 16, 16), generator=g) * 0.05   # near-duplicate boxes -> long NMS chains
    reg.view(1, 160, 4, 16, 16)[:, ::7, 2] = -20.0            # exp(-20)*w < 1e-3 -> dropped as too small
    props, counts, _ = _run_proposals(eng, _head_nhwc(obj, reg).to(DEV), 1)
    ref = _oracle_proposals(obj, reg)[0]
    assert counts[0] == ref.shape[0]
    assert (props[0, :ref.shape[0]] - ref).abs().max() <= 1e-4


# ------------------------------------------------------------------------- RoIAlign + avg pool
def _pad_props(plist):
    B = len(plist)
    props = torch.zeros((B, 1000, 4))
    offs = [0]
    for b, p in enumerate(plist):
        props[b, :p.shape[0]] = p
        offs.append(offs[-1] + p.shape[0])
    return props, torch.tensor(offs, dtype=torch.int32)


def _run_roi(eng, feat_nchw, plist):
    props, offs = _pad_props(plist)
    R = int(offs[-1])
    B, Cc, FH, FW = feat_nchw.shape
    out = torch.empty((R, 64, Cc), device=DEV)
    pooled = torch.empty((R, Cc), device=DEV)
    fn = feat_nchw.permute(0, 2, 3, 1).contiguous().to(DEV)
    props_d, offs_d = props.to(DEV), offs.to(DEV)
    _hip.check(eng.lib.rgrg_roi_align_avgpool_f32(fn.data_ptr(), props_d.data_ptr(), offs_d.data_ptr(), out.data_ptr(),
                                                  pooled.data_ptr(), B, FH, FW, Cc, 1000, R, 1.0 / 32, _stream()))
    return out.cpu().view(R, 8, 8, Cc).permute(0, 3, 1, 2), pooled.cpu()


def test_roi_align_on_oracle_features(eng, oracle_bench):
    feat, plist = oracle_bench["_features"], [p[:200] for p in oracle_bench["_proposals"]]
    out, pooled = _run_roi(eng, feat, plist)
    rois = torch.cat([torch.cat([torch.full((p.shape[0], 1), float(i)), p], 1) for i, p in enumerate(plist)], 0)
    ref = tv013.roi_align(feat, rois, 1.0 / 32, 8, 2)
    # same operation order with FP contraction off -> bit-exact
    assert torch.equal(out, ref), f"max abs diff {(out - ref).abs().max().item():.3e}"
    close(pooled, F.avg_pool2d(ref, 8).flatten(1), 1e-6, 1e-6, "8x8 average pool")


def test_roi_align_edge_boxes(eng):
    g = torch.Generator().manual_seed(5)
    feat = torch.randn((1, 256, 16, 16), generator=g)
    boxes = torch.tensor([[0.0, 0.0, 512.0, 512.0], [500.0, 500.0, 512.0, 512.0], [100.0, 100.0, 100.5, 100.2],
                          [0.0, 0.0, 1e-3, 1e-3], [31.9, 64.0, 480.1, 96.0], [200.0, 0.0, 232.0, 512.0]])
    out, pooled = _run_roi(eng, feat, [boxes])
    rois = torch.cat([torch.zeros((boxes.shape[0], 1)), boxes], 1)
    ref = tv013.roi_align(feat, rois, 1.0 / 32, 8, 2)
    assert torch.equal(out, ref), f"max abs diff {(out - ref).abs().max().item():.3e}"


@pytest.mark.parametrize("FH,FW,C", [(8, 8, 128), (12, 20, 64), (20, 12, 192)])
def test_roi_align_other_map_sizes_and_partial_chunks(eng, FH, FW, C):
    """Maps that are not 16 x 16 take the kernel variant with a run-time row stride (the 16-wide one has its tap offsets
    as instruction immediates); RoI counts around the 32-RoI chunk size (1, 31, 33 per image, an empty image in between),
    boxes on every border and samples outside the map (dead samples).  Bit-exact like the 16 x 16 case."""
    g = torch.Generator().manual_seed(FH * 100 + FW)
    feat = torch.randn((4, C, FH, FW), generator=g)
    W, H = FW * 32.0, FH * 32.0

    def boxes(n):
        xy = torch.rand((n, 2), generator=g) * torch.tensor([W, H]) * 0.9
        wh = torch.rand((n, 2), generator=g) * torch.tensor([W, H]) * 0.6 + 1.0
        b = torch.cat([xy, torch.minimum(xy + wh, torch.tensor([W, H]))], 1)
        b[0] = torch.tensor([0.0, 0.0, W, H])                      # the whole image: last row / column taps
        if n > 2:
            b[1] = torch.tensor([W - 3.0, H - 3.0, W, H])          # bottom-right corner
            b[2] = torch.tensor([-40.0, -40.0, 20.0, 20.0])        # starts outside: samples below -1 are dead
        return b
    plist = [boxes(33), torch.zeros((0, 4)), boxes(1), boxes(31)]
    out, pooled = _run_roi(eng, feat, plist)
    rois = torch.cat([torch.cat([torch.full((p.shape[0], 1), float(i)), p], 1) for i, p in enumerate(plist)], 0)
    ref = tv013.roi_align(feat, rois, 1.0 / 32, 8, 2)
    assert torch.equal(out, ref), f"max abs diff {(out - ref).abs().max().item():.3e}"
    close(pooled, F.avg_pool2d(ref, 8).flatten(1), 1e-6, 1e-6, "8x8 average pool")


# ------------------------------------------------------------------------- top-1 per class
def test_top1_per_class_on_oracle_inputs(eng, oracle_bench):
    plist = oracle_bench["_proposals"]
    props, offs = _pad_props(plist)
    pred = torch.cat([oracle_bench["_class_logits"], oracle_bench["_box_regression"]], 1).contiguous()
    pooled = oracle_bench["_pooled_avg"].contiguous()
    B = 2
    cd = torch.zeros((B, 29), dtype=torch.uint8, device=DEV)
    sc = torch.zeros((B, 29), device=DEV)
    bx = torch.zeros((B, 29, 4), device=DEV)
    ft = torch.zeros((B, 29, 2048), device=DEV)
    pred_d, props_d, offs_d, pooled_d = pred.to(DEV), props.to(DEV), offs.to(DEV), pooled.to(DEV)
    _hip.check(eng.lib.rgrg_top1_per_class_f32(pred_d.data_ptr(), 150, props_d.data_ptr(), offs_d.data_ptr(),
                                               pooled_d.data_ptr(), cd.data_ptr(), sc.data_ptr(), bx.data_ptr(), ft.data_ptr(),
                                               B, 2048, 1000, 512.0, 512.0, _stream()))
    ref_cd, ref_ft, ref_bx, ref_sc = o_det.top_region_postprocess(pooled, oracle_bench["_box_regression"],
                                                                  oracle_bench["_class_logits"], plist, [(512, 512)] * 2)
    assert torch.equal(cd.cpu().bool(), ref_cd)
    assert torch.equal(ft.cpu(), ref_ft)                         # pure gather: exact (=> same arg-max boxes)
    close(sc, ref_sc, 1e-5, 1e-6, "top scores (softmax through expf)")
    assert (bx.cpu() - ref_bx).abs().max() <= 1e-3               # pixels; decode goes through expf


def test_top1_per_class_undetected_and_empty_image(eng):
    g = torch.Generator().manual_seed(2)
    n0 = 50
    logits = torch.randn((n0, 30), generator=g)
    logits[:, 5] = -50.0                                         # region 4 never wins -> undetected, score 0, index 0
    deltas = torch.randn((n0, 120), generator=g) * 0.1
    boxes0 = torch.rand((n0, 2), generator=g) * 300
    plist = [torch.cat([boxes0, boxes0 + 50 + torch.rand((n0, 2), generator=g) * 100], 1), torch.zeros((0, 4))]
    props, offs = _pad_props(plist)
    pooled = torch.randn((n0, 256), generator=g)
    pred = torch.cat([logits, deltas], 1).contiguous()
    cd = torch.ones((2, 29), dtype=torch.uint8, device=DEV)
    sc, bx, ft = torch.ones((2, 29), device=DEV), torch.ones((2, 29, 4), device=DEV), torch.ones((2, 29, 256), device=DEV)
    pred_d, props_d, offs_d, pooled_d = pred.to(DEV), props.to(DEV), offs.to(DEV), pooled.to(DEV)
    _hip.check(eng.lib.rgrg_top1_per_class_f32(pred_d.data_ptr(), 150, props_d.data_ptr(), offs_d.data_ptr(),
                                               pooled_d.data_ptr(), cd.data_ptr(), sc.data_ptr(), bx.data_ptr(), ft.data_ptr(),
                                               2, 256, 1000, 512.0, 512.0, _stream()))
    ref_cd, ref_ft, ref_bx, ref_sc = o_det.top_region_postprocess(pooled, deltas, logits, plist[:1], [(512, 512)])
    assert torch.equal(cd.cpu()[0].bool(), ref_cd[0]) and not ref_cd[0, 4]
    assert torch.equal(ft.cpu()[0], ref_ft[0]) and torch.equal(ft.cpu()[0, 4], pooled[0])
    assert sc.cpu()[0, 4] == 0.0
    assert (bx.cpu()[0] - ref_bx[0]).abs().max() <= 1e-3
    assert not cd.cpu()[1].any() and (sc.cpu()[1] == 0).all() and (ft.cpu()[1] == 0).all()  # image without proposals


# ------------------------------------------------------------------------- selection
def test_select_regions_and_gather(eng):
    g = torch.Generator().manual_seed(8)
    n = 29 * 70  # > 1024: exercises the chunked ordered compaction
    logits = torch.randn((n,), generator=g) - 1.0
    logits[::13] = -1.0                                          # threshold is strict: exactly -1 is NOT selected
    det = (torch.rand((n,), generator=g) > 0.3)
    sel = torch.empty((n,), dtype=torch.uint8, device=DEV)
    rows = torch.full((n,), -1, dtype=torch.int32, device=DEV)
    cnt = torch.zeros((1,), dtype=torch.int32, device=DEV)
    logits_d, det_d = logits.to(DEV), det.to(torch.uint8).to(DEV)
    _hip.check(eng.lib.rgrg_select_regions_f32(logits_d.data_ptr(), det_d.data_ptr(), -1.0,
                                               sel.data_ptr(), rows.data_ptr(), cnt.data_ptr(), n, _stream()))
    ref = (logits > -1) & det
    assert torch.equal(sel.cpu().bool(), ref)
    S = int(cnt.item())
    assert S == int(ref.sum()) and torch.equal(rows.cpu()[:S].long(), ref.nonzero().flatten())
    src = torch.randn((n, 64), generator=g)
    dst = torch.empty((S, 64), device=DEV)
    src_d = src.to(DEV)
    _hip.check(eng.lib.rgrg_gather_rows_f32(src_d.data_ptr(), rows.data_ptr(), dst.data_ptr(), S, 64, _stream()))
    assert torch.equal(dst.cpu(), src[ref])


def test_selection_head_vs_oracle(eng, oracle_bench):
    sd = synth_sd("bench")
    trf, cd = oracle_bench["top_region_features"], oracle_bench["class_detected"]
    taps = {}
    sel, feats = eng.select(trf.to(DEV), cd.to(DEV), taps)
    ref_sel, ref_feats, ref_logits = o_full.region_selection(sd, trf, cd)
    close(taps["selection_logits"], ref_logits, 1e-5, 1e-6, "selection logits")
    assert torch.equal(sel.cpu(), ref_sel) and torch.equal(feats.cpu(), ref_feats)


# ------------------------------------------------------------------------- detector end to end
def test_detector_end_to_end_vs_oracle(eng, oracle_bench):
    taps = {}
    det, top, cd = eng.detect(oracle_bench["_images"].to(DEV), taps)
    assert taps["counts"].cpu().tolist() == [p.shape[0] for p in oracle_bench["_proposals"]]
    assert torch.equal(cd.cpu(), oracle_bench["class_detected"])
    # region boxes: pixels, tolerance 0.05 px (fp32 convs -> fc6 -> deltas -> exp); scores 1e-4 absolute
    assert (det["top_region_boxes"].cpu() - oracle_bench["detections"]["top_region_boxes"]).abs().max() <= 5e-2
    assert (det["top_scores"].cpu() - oracle_bench["detections"]["top_scores"]).abs().max() <= 1e-4
    close(top, oracle_bench["top_region_features"], 1e-3, 1e-4, "top_region_features")


# ------------------------------------------------------------------------- bf16-weight GEMM (opt-in path)
T16 = {0: torch.bfloat16, 1: torch.float16}   # the `fp16` argument of the 16-bit entry points -> torch dtype


@pytest.mark.parametrize("fp16", [0, 1])
def test_f32_to_bf16_is_round_to_nearest_even(eng, fp16):
    """fp32 -> bf16 / fp16 bits exactly as torch rounds (nearest even; fp16: subnormals, overflow to inf beyond 65504),
    and the widening back is exact."""
    g = torch.Generator().manual_seed(17)
    x = torch.randn((4099,), generator=g) * 37.0
    x[:8] = torch.tensor([1.0 + 2 ** -8, 1.0 + 3 * 2 ** -8, -0.0, 65504.0, 65520.0, 1e-7, -3e-6, 1.0 + 2 ** -11])  # exact ties, edges
    xd = x.to(DEV)
    out = torch.empty((4099,), dtype=torch.int16, device=DEV)
    _hip.check(eng.lib.rgrg_f32_to_bf16(xd.data_ptr(), out.data_ptr(), 4099, fp16, _stream()))
    assert torch.equal(out.cpu(), x.to(T16[fp16]).view(torch.int16))
    back = torch.empty((4099,), dtype=torch.float32, device=DEV)
    _hip.check(eng.lib.rgrg_bf16_to_f32(out.data_ptr(), back.data_ptr(), 4099, fp16, _stream()))
    assert torch.equal(back.cpu(), x.to(T16[fp16]).float())


@pytest.mark.parametrize("fp16", [0, 1])
@pytest.mark.parametrize("M,N,K,act,res", [(300, 512, 1024, 0, True), (928, 1024, 4096, 2, False), (129, 50257, 1024, 0, False)])
def test_linear_bf16w(eng, M, N, K, act, res, fp16):
    """16-bit MFMA path (v_mfma_f32_32x32x16_bf16 / _f16): products of rounded operands are exact in fp32, so the only
    difference to a float64 reference on the SAME rounded operands is the fp32 summation order."""
    t16 = T16[fp16]
    g = torch.Generator().manual_seed(M + N)
    A = torch.randn((M, K), generator=g)
    W = torch.randn((N, K), generator=g) / math.sqrt(K)
    b = torch.randn((N,), generator=g)
    R = torch.randn((M, N), generator=g) if res else None
    ref = A.to(t16).double() @ W.to(t16).double().t() + b.double()
    if res:
        ref = ref + R.double()
    ref = {0: lambda x: x, 2: lambda x: F.gelu(x, approximate="tanh")}[act](ref)
    Ad, Wd, bd = A.to(DEV), W.to(DEV), b.to(DEV)
    Rd = R.to(DEV) if res else None
    Wb = torch.empty((N, K), dtype=torch.int16, device=DEV)
    _hip.check(eng.lib.rgrg_f32_to_bf16(Wd.data_ptr(), Wb.data_ptr(), N * K, fp16, _stream()))
    y = torch.empty((M, N), device=DEV)
    _hip.check(eng.lib.rgrg_linear_bf16w_f32(Ad.data_ptr(), Wb.data_ptr(), bd.data_ptr(), Rd.data_ptr() if res else None,
                                             y.data_ptr(), M, N, K, N, act, fp16, _stream()))
    close(y, ref, 2e-5, 2e-6, f"16-bit-weight linear {M}x{N}x{K} ({t16})")


@pytest.mark.parametrize("M,N,K,act,res,out16", [(923, 3072, 1024, 0, False, False), (923, 1024, 4096, 0, True, False),
                                                   (923, 4096, 1024, 2, False, True), (129, 50257, 1024, 0, False, False),
                                                   (70, 192, 256, 0, True, False), (300, 1000, 768, 2, False, False),
                                                   (1, 64, 256, 0, False, False)])
@pytest.mark.parametrize("fp16", [0, 1])
def test_linear_bf16_lds_dma_kernel(eng, M, N, K, act, res, out16, fp16):
    """Round-3 LDS-DMA GEMM (both operands bf16 in HBM, `buffer_load ... lds`, 4 stages across raw barriers, XOR-swizzled
    LDS rows): against a float64 reference on the same bf16 operands.  Shapes: the four decode projections at the
    configs[2] row count (923: ragged last row tile; 128x128 and 64x64 tile paths), the vocabulary edge (50257 columns:
    ragged last column tile), tiny / odd shapes, K = 256 (the shortest pipeline: prologue + tail only), bf16 output."""
    g = torch.Generator().manual_seed(M * 7 + N)
    A = torch.randn((M, K), generator=g)
    W = torch.randn((N, K), generator=g) / math.sqrt(K)
    b = torch.randn((N,), generator=g)
    R = torch.randn((M, N), generator=g) if res else None
    t16 = T16[fp16]
    ref = A.to(t16).double() @ W.to(t16).double().t() + b.double()
    if res:
        ref = ref + R.double()
    ref = {0: lambda x: x, 2: lambda x: F.gelu(x, approximate="tanh")}[act](ref)
    A16 = A.to(t16).view(torch.int16).to(DEV)
    Wb = W.to(t16).view(torch.int16).to(DEV)
    bd = b.to(DEV)
    Rd = R.to(DEV) if res else None
    y = torch.empty((M, N), device=DEV)
    y16 = torch.empty((M, N), dtype=torch.int16, device=DEV)
    _hip.check(eng.lib.rgrg_linear_bf16_f32(A16.data_ptr(), Wb.data_ptr(), bd.data_ptr(), Rd.data_ptr() if res else None,
                                            None if out16 else y.data_ptr(), y16.data_ptr() if out16 else None, M, N, K, N, act,
                                            fp16, _stream()))
    if out16:
        assert torch.equal(y16.cpu().view(t16), ref.float().to(t16)) or \
            (y16.cpu().view(t16).double() - ref).abs().max().item() <= 2 ** -7 * ref.abs().max().item()
    else:
        close(y, ref, 2e-5, 2e-6, f"16-bit LDS-DMA linear {M}x{N}x{K} ({t16})")
    tol = (2e-5, 2e-6) if act == 0 else (2e-5, 4e-6)  # GELU on the hardware exp2 / rcp (~1e-7 relative to tanhf)
    for tile in [shape + 16 * nst for shape in (1, 2, 3, 4) for nst in (2, 3, 4)]:   # every tile shape x stage count
        y2 = torch.empty((M, N), device=DEV)
        _hip.check(eng.lib.rgrg_debug_linear_bf16_tile(A16.data_ptr(), Wb.data_ptr(), bd.data_ptr(), Rd.data_ptr() if res else None,
                                                       y2.data_ptr(), M, N, K, N, act, tile, 0, 0, fp16, _stream()))
        close(y2, ref, tol[0], tol[1], f"bf16 LDS-DMA linear {M}x{N}x{K} tile {tile}")


@pytest.mark.parametrize("fp16", [0, 1])
@pytest.mark.parametrize("M,N2,act", [(923, 3072, 0), (923, 4096, 2), (131, 192, 0), (64, 64, 2), (4100, 192, 0), (16400, 64, 0)])
def test_layernorm_folded_gemm_pair_against_layernorm_then_linear(eng, M, N2, act, fp16):
    """The LayerNorm folded around the 16-bit decode GEMMs (round 4; transformers GPT2Block ln_1 -> c_attn, ln_2 -> c_fc):
      producer  x = A16 W1^T + b1 + R (N = 1024) also stores x as 16 bit and per-row (sum, sum of squares) slots per 64 columns;
      consumer  y = act(LN(x; gain, beta) W2^T + b2) computed as act(rstd (x16 Wg^T - mean colsum) + shift), Wg = round16(gain o W2).
    Checked piece by piece: the fold vectors against torch (rounded weights bit-exact), the slots against exact row sums of
    the producer's own fp32 output, the 16-bit copy bit-exact, the consumer against a float64 evaluation of its OWN formula on
    the same rounded operands (2e-5), and against LayerNorm -> round -> matmul (what the unfolded path and the oracle do) at the
    16-bit noise level.  Rows with a large mean and an outlier column are included (the fold rounds x, not LN(x)).  The row
    counts pick every producer tile: 64 x 64 (two waves of a row block meet in LDS), 128 x 64 at 4100 rows (two row blocks per
    wave), 128 x 128 at 16 400 rows (a wave covers 64 columns itself)."""
    g = torch.Generator().manual_seed(M + N2 + 7 * fp16)
    t16 = T16[fp16]
    K1, D = 1024, 1024
    A = torch.randn((M, K1), generator=g)
    W1 = torch.randn((D, K1), generator=g) / math.sqrt(K1)
    b1 = torch.randn((D,), generator=g)
    R = torch.randn((M, D), generator=g) * 2.0
    R[::3] += 1.5                    # rows with mean / std ~ 0.6
    R[:, 77] += 40.0                 # an outlier feature, as GPT-2 residual streams have
    gain = 1.0 + 0.2 * torch.randn((D,), generator=g)
    beta = 0.3 * torch.randn((D,), generator=g)
    W2 = torch.randn((N2, D), generator=g) / math.sqrt(D)
    b2 = torch.randn((N2,), generator=g)
    dev = lambda t: t.to(DEV).contiguous()  # noqa: E731
    # --- weight side
    wg = torch.empty((N2, D), dtype=torch.int16, device=DEV)
    cs = torch.empty((N2,), device=DEV)
    sh = torch.empty((N2,), device=DEV)
    dW2, dgain, dbeta, db2 = dev(W2), dev(gain), dev(beta), dev(b2)
    _hip.check(eng.lib.rgrg_debug_ln_fold16(dW2.data_ptr(), dgain.data_ptr(), dbeta.data_ptr(), db2.data_ptr(), wg.data_ptr(),
                                            cs.data_ptr(), sh.data_ptr(), N2, D, fp16, _stream()))
    wg_ref = (W2 * gain[None, :]).to(t16)
    assert torch.equal(wg.cpu().view(t16), wg_ref)
    close(cs, wg_ref.double().sum(1), 1e-6, 1e-5, "column sums of the rounded scaled weights")
    close(sh, b2.double() + W2.double() @ beta.double(), 1e-6, 1e-5, "folded shift")
    # --- producer
    A16, W1b = dev(A.to(t16).view(torch.int16)), dev(W1.to(t16).view(torch.int16))
    x = torch.empty((M, D), device=DEV)
    x16 = torch.empty((M, D), dtype=torch.int16, device=DEV)
    slots = torch.full((M, 16, 2), float("nan"), device=DEV)
    dR, db1 = dev(R), dev(b1)
    _hip.check(eng.lib.rgrg_debug_linear_bf16_ln(A16.data_ptr(), W1b.data_ptr(), db1.data_ptr(), dR.data_ptr(), x.data_ptr(), x16.data_ptr(),
                                                 slots.data_ptr(), None, None, M, D, K1, D, 0, fp16, _stream()))
    x_ref = A.to(t16).double() @ W1.to(t16).double().t() + b1.double() + R.double()
    close(x, x_ref, 2e-5, 2e-6, "producer output")
    xc = x.cpu()
    assert torch.equal(x16.cpu().view(t16), xc.to(t16))
    blocks = xc.double().view(M, 16, 64)
    close(slots[:, :, 0], blocks.sum(2), 1e-6, 1e-4, "per-block row sums")
    close(slots[:, :, 1], (blocks * blocks).sum(2), 1e-6, 1e-3, "per-block row sums of squares")
    # --- consumer
    y = torch.empty((M, N2), device=DEV)
    _hip.check(eng.lib.rgrg_debug_linear_bf16_ln(x16.data_ptr(), wg.data_ptr(), sh.data_ptr(), None, y.data_ptr(), None, None,
                                                 slots.data_ptr(), cs.data_ptr(), M, N2, D, N2, act, fp16, _stream()))
    actf = {0: lambda v: v, 2: lambda v: F.gelu(v, approximate="tanh")}[act]
    mean = xc.double().mean(1, keepdim=True)
    rstd = 1.0 / torch.sqrt(xc.double().var(1, unbiased=False, keepdim=True) + 1e-5)
    own = actf(rstd * (xc.to(t16).double() @ wg_ref.double().t() - mean * wg_ref.double().sum(1)[None, :])
               + (b2.double() + W2.double() @ beta.double())[None, :])
    close(y, own, 5e-5, 2e-5, "consumer against its own formula in float64")
    ln = F.layer_norm(xc, (D,), gain, beta, 1e-5)
    unfolded = actf(ln.to(t16).double() @ W2.to(t16).double().t() + b2.double())
    span = unfolded.abs().max().item()
    err = (y.cpu().double() - unfolded).abs().max().item()
    assert err <= (4e-3 if fp16 else 3e-2) * span, (err, span)     # two roundings of the same magnitude at different places


# ------------------------------------------------------------------------- image preprocessing
@pytest.mark.parametrize("h,w", [(3056, 2544), (2544, 3056), (1024, 1024), (1536, 1536), (768, 512), (512, 512), (700, 513),
                                 (300, 200), (200, 300), (256, 256), (511, 3), (37, 41)])
def test_preprocess_matches_the_restated_opencv_pipeline(h, w):
    """rgrg_preprocess_u8_f32 vs oracle/preprocess.py (OpenCV INTER_AREA + albumentations pad/normalize restated; the
    third-party originals are absent -> unpinned): the rounded 8-bit pixel must be identical, so the float output is
    bit-identical.  Shapes cover the general table path (5.97x, portrait and landscape), the integer 2x2 and 3x3 fast
    paths, a 1.5x scale, no resize, an odd size, and images SMALLER than 512 px (round 3: LongestMaxSize enlarges them and
    OpenCV emulates INTER_AREA with its fixed-point bilinear path - 1.7x, exact 2x, a 3-pixel-wide strip, a tiny image)."""
    import numpy as np
    from oracle import preprocess as P
    from rgrg_amd.preprocess import preprocess_image
    rng = np.random.default_rng(h * 10007 + w)
    img = rng.integers(0, 256, (h, w), dtype=np.uint8)
    img[: h // 3] = (np.arange(w) % 256).astype(np.uint8)  # smooth part: exercises .5 rounding cases less randomly
    ref = torch.from_numpy(P.get_image_tensor_from_array(img))
    out = preprocess_image(img, DEV).cpu()
    assert out.shape == (1, 1, 512, 512) and out.dtype == torch.float32
    diff = (out - ref).abs()
    # one 8-bit level = 1/(0.302*255) = 0.013: any rounding disagreement would show as >= 0.013
    assert diff.max().item() == 0.0, (diff.max().item(), int((diff > 0).sum()))
