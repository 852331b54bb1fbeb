"""Host-side driver of the HIP kernels: weight repacking at load time and the launch
sequence of the detector / selection head / decoder.

torch is used for device memory (tensors as buffers), the current stream and the
one-off layout transforms of the weights; every FLOP of the hot path runs in
``librgrg_hip.so``.  The engine refuses to run without a GPU.
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, List, Optional, Tuple

import torch

from . import _hip
from .constants import (ANCHOR_RATIOS, ANCHOR_SIZES, IMAGE_INPUT_SIZE, NUM_REGIONS, RESNET50_LAYERS,
                        RPN_NMS_THRESH, RPN_POST_NMS_TOP_N, RPN_PRE_NMS_TOP_N, SELECTION_LOGIT_THRESHOLD)

Tensor = torch.Tensor
BN_EPS = 1e-5


def _require_gpu(device: torch.device) -> None:
    if device.type != "cuda" or not torch.cuda.is_available():
        raise _hip.RgrgHipError("rgrg_amd runs only on an AMD GPU through librgrg_hip.so: move the model to "
                                "'cuda' (there is no CPU fallback)")


def _on_engine_device(cls):
    """Every engine entry point runs with the engine's device current: librgrg_hip.so allocates, creates streams and
    launches on the CURRENT device, so a model on cuda:1 must not be driven while cuda:0 is current (single-process
    multi-GPU, generate_sharded without set_device)."""
    import functools

    def wrap(fn):
        @functools.wraps(fn)
        def inner(self, *a, **k):
            if torch.cuda.current_device() == self.device.index:   # the common case: no hipGetDevice / hipSetDevice pair
                return fn(self, *a, **k)
            with torch.cuda.device(self.device):
                return fn(self, *a, **k)
        return inner

    for name, fn in list(vars(cls).items()):
        if isinstance(fn, (staticmethod, classmethod)):
            continue
        if callable(fn) and not name.startswith("__") and name != "_s":
            setattr(cls, name, wrap(fn))
    return cls


def pick_splitk(M: int, N: int, K: int) -> int:
    """Split the K loop over workgroups when the output has too few tiles to fill the 256
    CUs (each split still streams >= 4 K tiles of 32).  Large outputs (fc6: 823x1024 with
    K = 131072) get the 128x128 tile (the C side picks it when tiles*splitk >= 192), which
    halves the operand re-reads of the 64x64 tile; small ones split the 64x64 grid."""
    def grow(tiles: int, target: int, cap: int) -> int:
        sk = 1
        while sk * 2 <= cap and tiles * sk < target and K % (32 * sk * 2) == 0 and K // (32 * sk * 2) >= 4:
            sk *= 2
        return sk

    if M >= 512 and N >= 512:
        return grow(((M + 127) // 128) * ((N + 127) // 128), 192, 16)
    tiles = ((M + 63) // 64) * ((N + 63) // 64)
    if tiles >= 256:
        return 1
    return grow(tiles, 256, 16)


def grid_anchors(image_size: int, grid: int) -> Tensor:
    """AnchorGenerator of object_detector.py:78-81 (torchvision 0.13.1 semantics):
    160 anchors per cell, index = (y*grid + x)*160 + ratio*10 + size, base anchors
    rounded half-to-even, stride = image // grid, no half-stride offset.  CPU fp32."""
    scales = torch.tensor(ANCHOR_SIZES, dtype=torch.float32)
    h_r = torch.sqrt(torch.tensor(ANCHOR_RATIOS, dtype=torch.float32))
    w_r = 1.0 / h_r
    ws = (w_r[:, None] * scales[None, :]).reshape(-1)
    hs = (h_r[:, None] * scales[None, :]).reshape(-1)
    base = (torch.stack([-ws, -hs, ws, hs], dim=1) / 2).round()
    stride = image_size // grid
    s = torch.arange(0, grid, dtype=torch.int32) * stride
    yy, xx = torch.meshgrid(s, s, indexing="ij")
    shifts = torch.stack((xx.reshape(-1), yy.reshape(-1), xx.reshape(-1), yy.reshape(-1)), dim=1).to(torch.float32)
    return (shifts.view(-1, 1, 4) + base.view(1, -1, 4)).reshape(-1, 4).contiguous()


class _Conv:
    """One conv (+BN) layer in kernel layout: w [Cout][KH][KW][Cin], scale/shift [Cout]."""

    def __init__(self, w: Tensor, scale: Optional[Tensor], shift: Optional[Tensor], stride: int, pad: int):
        self.cout, self.cin, self.kh, self.kw = w.shape
        self.w = w.permute(0, 2, 3, 1).contiguous()
        self.scale, self.shift, self.stride, self.pad = scale, shift, stride, pad


def _bn_affine(sd: Dict[str, Tensor], p: str) -> Tuple[Tensor, Tensor]:
    # eval BatchNorm2d as PyTorch's CPU kernel applies it: alpha = w / sqrt(var+eps), beta = b - mean*alpha
    invstd = 1.0 / torch.sqrt(sd[p + "running_var"] + BN_EPS)
    alpha = sd[p + "weight"] * invstd
    beta = sd[p + "bias"] - sd[p + "running_mean"] * alpha
    return alpha.contiguous(), beta.contiguous()


@_on_engine_device
class HipEngine:
    def __init__(self, state_dict: Dict[str, Tensor], device: torch.device):
        _require_gpu(device)
        if device.index is None:
            device = torch.device("cuda", torch.cuda.current_device())
        self.device = device
        self.lib = _hip.load()
        self._decoder = None
        self._kv = None        # the decoder's K/V cache: a torch tensor (caller-owned memory of rgrg_decoder_create_with_cache)
        self._cached = None
        self._init(state_dict)

    def _init(self, state_dict: Dict[str, Tensor]) -> None:
        device = self.device
        arch = C.create_string_buffer(64)
        _hip.check(self.lib.rgrg_device_arch(device.index, arch, 64), "rgrg_device_arch")
        self.arch = arch.value.decode()
        sd = {k: v.detach().to(device=device, dtype=torch.float32) for k, v in state_dict.items()
              if v.dtype.is_floating_point}
        self._decoder = None
        self._decoder_caps = (0, 0)
        # a sub-model used stand-alone only brings its own keys: pack what is there
        self.has_detector = "object_detector.backbone.0.weight" in sd
        self.has_selection = "binary_classifier_region_selection.classifier.0.weight" in sd
        self.has_decoder = "language_model.gpt_with_lm_head.transformer.wte.weight" in sd
        if self.has_detector:
            self._pack_detector(sd)
        self.sel = self.abn = None
        if self.has_selection:
            self._pack_selection(sd)
        a = "binary_classifier_region_abnormal.classifier."  # forward() only
        if a + "0.weight" in sd:
            self.abn = [(sd[a + f"{i}.weight"].contiguous(), sd[a + f"{i}.bias"].contiguous()) for i in (0, 2, 4)]
        if self.has_decoder:
            self._pack_decoder(sd)

    def _s(self) -> int:
        """The caller's stream ON THE ENGINE'S DEVICE (a ``void*`` for the C ABI)."""
        return torch.cuda.current_stream(self.device).cuda_stream

    # ------------------------------------------------------------------ packing
    def _pack_detector(self, sd):
        bb = "object_detector.backbone."
        w = sd[bb + "0.weight"]  # [64,1,7,7]
        self.stem_w = w.reshape(64, 49).t().contiguous()  # [49][64]
        self.stem_scale, self.stem_shift = _bn_affine(sd, bb + "1.")
        self.blocks: List[Dict[str, _Conv]] = []
        for li, (_planes, blocks, stride) in enumerate(RESNET50_LAYERS):
            for b in range(blocks):
                p = f"{bb}{4 + li}.{b}."
                s = stride if b == 0 else 1
                blk = {"c1": _Conv(sd[p + "conv1.weight"], *_bn_affine(sd, p + "bn1."), 1, 0),
                       "c2": _Conv(sd[p + "conv2.weight"], *_bn_affine(sd, p + "bn2."), s, 1),
                       "c3": _Conv(sd[p + "conv3.weight"], *_bn_affine(sd, p + "bn3."), 1, 0)}
                if (p + "downsample.0.weight") in sd:
                    blk["ds"] = _Conv(sd[p + "downsample.0.weight"], *_bn_affine(sd, p + "downsample.1."), s, 0)
                self.blocks.append(blk)
        rp = "object_detector.rpn.head."
        self.rpn_conv = _Conv(sd[rp + "conv.0.0.weight"], None, sd[rp + "conv.0.0.bias"].contiguous(), 1, 1)
        # objectness (160) and deltas (640) 1x1 convs fused into one N=800 GEMM
        w_head = torch.cat([sd[rp + "cls_logits.weight"], sd[rp + "bbox_pred.weight"]], 0)
        b_head = torch.cat([sd[rp + "cls_logits.bias"], sd[rp + "bbox_pred.bias"]], 0).contiguous()
        self.rpn_head = _Conv(w_head, None, b_head, 1, 0)
        self.num_anchors = sd[rp + "cls_logits.weight"].shape[0]
        self.anchors = grid_anchors(IMAGE_INPUT_SIZE, 16).to(self.device)
        rh = "object_detector.roi_heads."
        # RoIAlign output is [roi][bin][channel]; torchvision flattens [channel][bin] -> permute fc6's K axis
        w6 = sd[rh + "box_head.fc6.weight"]
        self.fc6_w = w6.view(w6.shape[0], 2048, 64).permute(0, 2, 1).reshape(w6.shape[0], -1).contiguous()
        self.fc6_b = sd[rh + "box_head.fc6.bias"].contiguous()
        self.fc7_w = sd[rh + "box_head.fc7.weight"].contiguous()
        self.fc7_b = sd[rh + "box_head.fc7.bias"].contiguous()
        self.pred_w = torch.cat([sd[rh + "box_predictor.cls_score.weight"], sd[rh + "box_predictor.bbox_pred.weight"]], 0).contiguous()
        self.pred_b = torch.cat([sd[rh + "box_predictor.cls_score.bias"], sd[rh + "box_predictor.bbox_pred.bias"]], 0).contiguous()
        self.dimred_w = sd[rh + "dim_reduction.weight"].contiguous()
        self.dimred_b = sd[rh + "dim_reduction.bias"].contiguous()

    def _pack_selection(self, sd):
        c = "binary_classifier_region_selection.classifier."
        self.sel = [(sd[c + f"{i}.weight"].contiguous(), sd[c + f"{i}.bias"].contiguous()) for i in (0, 2, 4)]

    def _pack_decoder(self, sd):
        g = "language_model.gpt_with_lm_head.transformer."
        f = "language_model.feature_space_transformation_nn."
        keep: List[Tensor] = []

        def T(t: Tensor) -> Tensor:  # HF Conv1D [in,out] -> [out,in]
            t = t.t().contiguous()
            keep.append(t)
            return t

        def K(t: Tensor) -> Tensor:
            t = t.contiguous()
            keep.append(t)
            return t

        n_layer = 0
        while f"{g}h.{n_layer}.ln_1.weight" in sd:
            n_layer += 1
        self.n_layer = n_layer
        layers = (_hip.DecoderLayerWeights * n_layer)()
        uk = []
        ub = []
        for l in range(n_layer):
            b = f"{g}h.{l}."
            lw = layers[l]
            lw.ln1_g, lw.ln1_b = K(sd[b + "ln_1.weight"]).data_ptr(), K(sd[b + "ln_1.bias"]).data_ptr()
            lw.ln2_g, lw.ln2_b = K(sd[b + "ln_2.weight"]).data_ptr(), K(sd[b + "ln_2.bias"]).data_ptr()
            lw.c_attn_w, lw.c_attn_b = T(sd[b + "attn.c_attn.weight"]).data_ptr(), K(sd[b + "attn.c_attn.bias"]).data_ptr()
            lw.attn_proj_w, lw.attn_proj_b = T(sd[b + "attn.c_proj.weight"]).data_ptr(), K(sd[b + "attn.c_proj.bias"]).data_ptr()
            lw.c_fc_w, lw.c_fc_b = T(sd[b + "mlp.c_fc.weight"]).data_ptr(), K(sd[b + "mlp.c_fc.bias"]).data_ptr()
            lw.mlp_proj_w, lw.mlp_proj_b = T(sd[b + "mlp.c_proj.weight"]).data_ptr(), K(sd[b + "mlp.c_proj.bias"]).data_ptr()
            uk += [sd[b + "attn.uk.weight"], sd[b + "attn.uv.weight"]]
            ub += [sd[b + "attn.uk.bias"], sd[b + "attn.uv.bias"]]
        dw = _hip.DecoderWeights()
        dw.n_layer, dw.d_model, dw.n_head = n_layer, 1024, 16
        wte = K(sd[g + "wte.weight"])
        dw.vocab = wte.shape[0]
        dw.wte = wte.data_ptr()
        dw.lnf_g, dw.lnf_b = K(sd[g + "ln_f.weight"]).data_ptr(), K(sd[g + "ln_f.bias"]).data_ptr()
        self._ukv_w, self._ukv_b = K(torch.cat(uk, 0)), K(torch.cat(ub, 0))  # [uk_0; uv_0; uk_1; ...] stacked Linear
        self._fst = [K(sd[f + "0.weight"]), K(sd[f + "0.bias"]), K(sd[f + "2.weight"]), K(sd[f + "2.bias"])]
        dw.fst0_w, dw.fst0_b, dw.fst2_w, dw.fst2_b = (t.data_ptr() for t in self._fst)
        dw.ukv_w, dw.ukv_b = self._ukv_w.data_ptr(), self._ukv_b.data_ptr()
        dw.layers = layers
        self._dec_weights, self._dec_layers, self._dec_keep = dw, layers, keep
        self.vocab = dw.vocab

    def close(self):
        if self._decoder is not None:
            # a token-id error of the last teacher-forced pass that has not been reported yet outlives the decoder
            pending = C.c_int(0)
            if self.lib.rgrg_decoder_take_id_error(self._decoder, C.byref(pending)) == 0 and pending.value:
                self._pending_id_error = True
            self.lib.rgrg_decoder_destroy(self._decoder)
            # the K/V cache is a torch tensor: presents of forward(use_cache=True) are views of it and keep it alive by
            # themselves after the decoder is gone (reading them is never a use-after-free)
            self._decoder = None
            self._kv = None
            self._cached = None
            # the decoder's own work space is raw hipMalloc memory: hand torch's now unused cache blocks (up to tens of GB) back to
            # the device so that the next decoder can allocate
            try:
                torch.cuda.empty_cache()
            except Exception:  # noqa: BLE001  (interpreter shutdown)
                pass

    def __del__(self):
        try:
            self.close()
        except Exception:  # noqa: BLE001
            pass

    # ------------------------------------------------------------------ op wrappers
    def linear(self, x: Tensor, w: Tensor, b: Optional[Tensor], act: int = _hip.ACT_NONE,
               residual: Optional[Tensor] = None, splitk: Optional[int] = None) -> Tensor:
        M, K = x.shape
        N = w.shape[0]
        y = torch.empty((M, N), dtype=torch.float32, device=x.device)
        sk = pick_splitk(M, N, K) if splitk is None else splitk
        ws = torch.empty((sk, M, N), dtype=torch.float32, device=x.device) if sk > 1 else None
        _hip.check(self.lib.rgrg_linear_f32(_hip.ptr(x), _hip.ptr(w), None, _hip.ptr(b), _hip.ptr(residual), _hip.ptr(y),
                                            M, N, K, N, act, sk, _hip.ptr(ws), self._s()), "rgrg_linear_f32")
        return y

    def conv(self, x: Tensor, c: _Conv, act: int, residual: Optional[Tensor] = None) -> Tensor:
        B, H, W, Cin = x.shape
        assert Cin == c.cin
        OH = (H + 2 * c.pad - c.kh) // c.stride + 1
        OW = (W + 2 * c.pad - c.kw) // c.stride + 1
        y = torch.empty((B, OH, OW, c.cout), dtype=torch.float32, device=x.device)
        M, N, K = B * OH * OW, c.cout, c.kh * c.kw * Cin
        sk = pick_splitk(M, N, K)
        ws = torch.empty((sk, M, N), dtype=torch.float32, device=x.device) if sk > 1 else None
        _hip.check(self.lib.rgrg_conv2d_nhwc_f32(_hip.ptr(x), _hip.ptr(c.w), _hip.ptr(c.scale), _hip.ptr(c.shift),
                                                 _hip.ptr(residual), _hip.ptr(y), B, H, W, Cin, c.cout, c.kh, c.kw,
                                                 c.stride, c.pad, act, sk, _hip.ptr(ws), self._s()),
                   "rgrg_conv2d_nhwc_f32")
        return y

    # ------------------------------------------------------------------ detector
    def backbone(self, images: Tensor) -> Tensor:
        """ResNet-50 trunk, [B,1,H,W] -> NHWC [B,H/32,W/32,2048]."""
        B, Cc, H, W = images.shape
        assert Cc == 1
        x = images.reshape(B, H, W).contiguous()
        y = torch.empty((B, H // 2, W // 2, 64), dtype=torch.float32, device=x.device)
        _hip.check(self.lib.rgrg_stem_conv7x7_f32(_hip.ptr(x), _hip.ptr(self.stem_w), _hip.ptr(self.stem_scale),
                                                  _hip.ptr(self.stem_shift), _hip.ptr(y), B, H, W, self._s()), "stem")
        p = torch.empty((B, H // 4, W // 4, 64), dtype=torch.float32, device=x.device)
        _hip.check(self.lib.rgrg_maxpool3x3s2_nhwc_f32(_hip.ptr(y), _hip.ptr(p), B, H // 2, W // 2, 64, self._s()), "maxpool")
        x = p
        for blk in self.blocks:
            o = self.conv(x, blk["c1"], _hip.ACT_RELU)
            o = self.conv(o, blk["c2"], _hip.ACT_RELU)
            idt = self.conv(x, blk["ds"], _hip.ACT_NONE) if "ds" in blk else x
            x = self.conv(o, blk["c3"], _hip.ACT_RELU, residual=idt)
        return x

    # ------------------------------------------------------------------ detector under torch.autocast (bf16 matrix core)
    def _act16(self, shape) -> Tensor:
        """bf16 activation buffer (int16 bits) with the 128 zero elements in front of it that rgrg_conv2d_nhwc_bf16 reads
        for padding taps; the returned view keeps the whole allocation alive."""
        n = 1
        for d in shape:
            n *= int(d)
        buf = torch.empty((n + 128,), dtype=torch.int16, device=self.device)
        buf[:128].zero_()
        return buf[128:].view(*shape)

    def _conv16_weights(self, c: _Conv, f16: int = 0):
        """[Cout][KH][KW][Cin] as 16-bit values (f16 = 0: bf16, 1: fp16) with the eval-BatchNorm scale folded in (alpha o W,
        then rounded once) + the f32 shift; one copy per type, made on first use."""
        key = "w16_f16" if f16 else "w16"
        if getattr(c, key, None) is None:
            w = c.w if c.scale is None else c.w * c.scale.view(-1, 1, 1, 1)
            w16 = torch.empty(w.shape, dtype=torch.int16, device=w.device)
            _hip.check(self.lib.rgrg_f32_to_bf16(_hip.ptr(w.contiguous()), _hip.ptr(w16), w.numel(), f16, self._s()), "rgrg_f32_to_bf16")
            setattr(c, key, w16)
        return getattr(c, key)

    def conv16(self, x16: Tensor, c: _Conv, act: int, residual16: Optional[Tensor] = None, out_f32: bool = False, f16: int = 0) -> Tensor:
        """nn.Conv2d + eval BatchNorm2d + residual + ReLU on 16-bit (bf16 / fp16) NHWC activations (rgrg_conv2d_nhwc_bf16)."""
        B, H, W, Cin = x16.shape
        assert Cin == c.cin and Cin % 64 == 0
        OH = (H + 2 * c.pad - c.kh) // c.stride + 1
        OW = (W + 2 * c.pad - c.kw) // c.stride + 1
        y = torch.empty((B, OH, OW, c.cout), dtype=torch.float32, device=x16.device) if out_f32 else self._act16((B, OH, OW, c.cout))
        _hip.check(self.lib.rgrg_conv2d_nhwc_bf16(_hip.ptr(x16), _hip.ptr(self._conv16_weights(c, f16)), _hip.ptr(c.shift), _hip.ptr(residual16),
                                                  _hip.ptr(y) if out_f32 else None, None if out_f32 else _hip.ptr(y), B, H, W, Cin, c.cout,
                                                  c.kh, c.kw, c.stride, c.pad, act, f16, self._s()), "rgrg_conv2d_nhwc_bf16")
        return y

    def backbone16(self, images: Tensor, f16: int = 0):
        """The trunk under autocast: stem + max-pool in fp32 (one input channel: 1 % of the FLOPs), the 16 bottlenecks as
        bf16 implicit GEMMs with fp32 accumulation, activations stored as bf16 -> (feat16 [B,16,16,2048] bf16,
        feat fp32 NHWC for RoIAlign / the caller)."""
        B, Cc, H, W = images.shape
        assert Cc == 1
        x = images.reshape(B, H, W).contiguous()
        y = torch.empty((B, H // 2, W // 2, 64), dtype=torch.float32, device=x.device)
        _hip.check(self.lib.rgrg_stem_conv7x7_f32(_hip.ptr(x), _hip.ptr(self.stem_w), _hip.ptr(self.stem_scale),
                                                  _hip.ptr(self.stem_shift), _hip.ptr(y), B, H, W, self._s()), "stem")
        p = torch.empty((B, H // 4, W // 4, 64), dtype=torch.float32, device=x.device)
        _hip.check(self.lib.rgrg_maxpool3x3s2_nhwc_f32(_hip.ptr(y), _hip.ptr(p), B, H // 2, W // 2, 64, self._s()), "maxpool")
        x16 = self._act16(p.shape)
        _hip.check(self.lib.rgrg_f32_to_bf16(_hip.ptr(p), _hip.ptr(x16), p.numel(), f16, self._s()), "rgrg_f32_to_bf16")
        del y, p
        for blk in self.blocks:
            o = self.conv16(x16, blk["c1"], _hip.ACT_RELU, f16=f16)
            o = self.conv16(o, blk["c2"], _hip.ACT_RELU, f16=f16)
            idt = self.conv16(x16, blk["ds"], _hip.ACT_NONE, f16=f16) if "ds" in blk else x16
            x16 = self.conv16(o, blk["c3"], _hip.ACT_RELU, residual16=idt, f16=f16)
        feat = torch.empty(x16.shape, dtype=torch.float32, device=x16.device)
        _hip.check(self.lib.rgrg_bf16_to_f32(_hip.ptr(x16), _hip.ptr(feat), x16.numel(), f16, self._s()), "rgrg_bf16_to_f32")
        return x16, feat

    def rpn(self, feat: Tensor, return_head: bool = False, feat16: Optional[Tensor] = None, f16: int = 0):
        """RPN head + proposal filtering -> proposals [B,1000,4], counts [B], offsets [B+1] (device)
        (+ the raw head output [B,FH,FW,800] = objectness | deltas when ``return_head``: the RPN losses read it)."""
        B, FH, FW, _ = feat.shape
        if feat16 is not None:  # autocast: the 3x3 conv (20 GFLOP / image) and the fused 1x1 heads on the bf16 matrix core
            t = self.conv16(feat16, self.rpn_conv, _hip.ACT_RELU, f16=f16)
            head = self.conv16(t, self.rpn_head, _hip.ACT_NONE, out_f32=True, f16=f16)
        else:
            t = self.conv(feat, self.rpn_conv, _hip.ACT_RELU)
            head = self.conv(t, self.rpn_head, _hip.ACT_NONE)  # [B,FH,FW,800]
        props = torch.empty((B, RPN_POST_NMS_TOP_N, 4), dtype=torch.float32, device=feat.device)
        counts = torch.empty((B,), dtype=torch.int32, device=feat.device)
        offsets = torch.empty((B + 1,), dtype=torch.int32, device=feat.device)
        size = float(IMAGE_INPUT_SIZE)
        _hip.check(self.lib.rgrg_rpn_proposals_f32(_hip.ptr(head), _hip.ptr(self.anchors), _hip.ptr(props),
                                                   _hip.ptr(counts), _hip.ptr(offsets), B, FH * FW, self.num_anchors,
                                                   RPN_PRE_NMS_TOP_N, RPN_POST_NMS_TOP_N, RPN_NMS_THRESH, 1e-3, size,
                                                   size, self._s()), "rgrg_rpn_proposals_f32")
        if return_head:
            return props, counts, offsets, head
        return props, counts, offsets

    def _fc6_bf16(self, f16: int = 0) -> Tensor:
        """16-bit copy (bf16 / fp16) of the (K-permuted) fc6 weight, made on first use of the autocast path."""
        key = "fc6_wb_f16" if f16 else "fc6_wb"
        if getattr(self, key, None) is None:
            wb = torch.empty(self.fc6_w.shape, dtype=torch.int16, device=self.fc6_w.device)
            _hip.check(self.lib.rgrg_f32_to_bf16(_hip.ptr(self.fc6_w), _hip.ptr(wb), self.fc6_w.numel(), f16, self._s()),
                       "rgrg_f32_to_bf16")
            setattr(self, key, wb)
        return getattr(self, key)

    def roi_heads(self, feat: Tensor, props: Tensor, offsets: Tensor, taps: Optional[dict] = None, bf16=False):
        B, FH, FW, Cf = feat.shape
        R = int(offsets[-1].item())  # host sync #1: number of RoIs sizes the box-head launches
        dev = feat.device
        cd = torch.zeros((B, NUM_REGIONS), dtype=torch.uint8, device=dev)
        scores = torch.zeros((B, NUM_REGIONS), dtype=torch.float32, device=dev)
        boxes = torch.zeros((B, NUM_REGIONS, 4), dtype=torch.float32, device=dev)
        feats = torch.zeros((B, NUM_REGIONS, Cf), dtype=torch.float32, device=dev)
        if R > 0:
            low = bool(bf16) and R > 128
            f16 = 1 if int(bf16) == 2 else 0
            pooled = torch.empty((R, Cf), dtype=torch.float32, device=dev)
            scale = 2.0 ** round(__import__("math").log2(FH / IMAGE_INPUT_SIZE))
            if low:
                # torch.autocast in the reference runs box_head in half precision: RoIAlign stores its [R, 64, C] maps as
                # bf16 (half of the 524 KB per RoI it writes and fc6 re-reads) and fc6 (81 % of the detector's FLOPs,
                # custom_roi_heads.py:235) runs on the LDS-DMA bf16 GEMM, fp32 accumulate / bias / ReLU
                pooled_maps = torch.empty((R, 64, Cf), dtype=torch.int16, device=dev)
                _hip.check(self.lib.rgrg_roi_align_avgpool_bf16maps(_hip.ptr(feat), _hip.ptr(props), _hip.ptr(offsets),
                                                                    _hip.ptr(pooled_maps), _hip.ptr(pooled), B, FH, FW, Cf,
                                                                    props.shape[1], R, scale, f16, self._s()), "rgrg_roi_align")
                h = torch.empty((R, self.fc6_w.shape[0]), dtype=torch.float32, device=dev)
                _hip.check(self.lib.rgrg_linear_bf16_f32(_hip.ptr(pooled_maps), _hip.ptr(self._fc6_bf16(f16)), _hip.ptr(self.fc6_b), None,
                                                         _hip.ptr(h), None, R, self.fc6_w.shape[0], 64 * Cf, self.fc6_w.shape[0],
                                                         _hip.ACT_RELU, f16, self._s()), "rgrg_linear_bf16_f32")
            else:
                pooled_maps = torch.empty((R, 64, Cf), dtype=torch.float32, device=dev)
                _hip.check(self.lib.rgrg_roi_align_avgpool_f32(_hip.ptr(feat), _hip.ptr(props), _hip.ptr(offsets),
                                                               _hip.ptr(pooled_maps), _hip.ptr(pooled), B, FH, FW, Cf,
                                                               props.shape[1], R, scale, self._s()), "rgrg_roi_align")
                h = self.linear(pooled_maps.view(R, 64 * Cf), self.fc6_w, self.fc6_b, _hip.ACT_RELU)
            h = self.linear(h, self.fc7_w, self.fc7_b, _hip.ACT_RELU)
            pred = self.linear(h, self.pred_w, self.pred_b)  # [R,150]: 30 class logits | 120 deltas
            size = float(IMAGE_INPUT_SIZE)
            _hip.check(self.lib.rgrg_top1_per_class_f32(_hip.ptr(pred), pred.shape[1], _hip.ptr(props), _hip.ptr(offsets),
                                                        _hip.ptr(pooled), _hip.ptr(cd), _hip.ptr(scores), _hip.ptr(boxes),
                                                        _hip.ptr(feats), B, Cf, props.shape[1], size, size, self._s()),
                       "rgrg_top1_per_class_f32")
            if taps is not None:
                taps.update(pooled_maps=pooled_maps, pooled=pooled, pred=pred)
        top = self.linear(feats.view(B * NUM_REGIONS, Cf), self.dimred_w, self.dimred_b).view(B, NUM_REGIONS, -1)
        return cd.bool(), scores, boxes, top

    def roi_align_boxes(self, feat_nhwc: Tensor, boxes: List[Tensor]) -> Tuple[Tensor, Tensor]:
        """RoIAlign(8x8)+avg on caller-supplied boxes (one [n_i,4] xyxy tensor per image):
        -> (maps [N,64,C] bin-major, pooled [N,C]).  Selection-based generation path
        (evaluate_bbox_variations.py:92-109)."""
        B, FH, FW, Cf = feat_nhwc.shape
        assert len(boxes) == B
        maxn = max(1, max(int(b.shape[0]) for b in boxes))
        props = torch.zeros((B, maxn, 4), dtype=torch.float32, device=feat_nhwc.device)
        offs = [0]
        for i, b in enumerate(boxes):
            props[i, : b.shape[0]] = b.to(torch.float32)
            offs.append(offs[-1] + int(b.shape[0]))
        offsets = torch.tensor(offs, dtype=torch.int32, device=feat_nhwc.device)
        R = offs[-1]
        maps = torch.empty((R, 64, Cf), dtype=torch.float32, device=feat_nhwc.device)
        pooled = torch.empty((R, Cf), dtype=torch.float32, device=feat_nhwc.device)
        scale = 2.0 ** round(__import__("math").log2(FH / IMAGE_INPUT_SIZE))
        _hip.check(self.lib.rgrg_roi_align_avgpool_f32(_hip.ptr(feat_nhwc), _hip.ptr(props), _hip.ptr(offsets), _hip.ptr(maps),
                                                       _hip.ptr(pooled), B, FH, FW, Cf, maxn, R, scale, self._s()), "rgrg_roi_align")
        return maps, pooled

    def detect(self, images: Tensor, taps: Optional[dict] = None, bf16=False, targets=None, keys_fn=None):
        """ObjectDetector.forward in eval mode: -> (detections, top_region_features, class_detected) or, with
        ``targets`` (list of {"boxes" [n,4], "labels" [n]} per image), (losses, detections, top_region_features,
        class_detected) where - as in the reference - the RoI heads then run on the SAMPLED training proposals.
        bf16 (opt-in through torch.autocast, like the reference's scripts; a mode: False / 0 fp32, True / 1 bfloat16, 2 float16 =
        _hip.autocast_mode()): bottlenecks, RPN convs and fc6 on the 16-bit matrix core of that type with fp32 accumulation, RoIAlign
        maps stored in it; stem, proposals / NMS, fc7, predictor, post-processing stay fp32."""
        _require_gpu(images.device)
        images = images.to(torch.float32)
        feat16 = None
        f16 = 1 if int(bf16) == 2 else 0
        if bf16:
            feat16, feat = self.backbone16(images, f16)
        else:
            feat = self.backbone(images)
        if targets is None:
            props, counts, offsets = self.rpn(feat, feat16=feat16, f16=f16)
            cd, scores, boxes, top = self.roi_heads(feat, props, offsets, taps, bf16)
            if taps is not None:
                taps.update(features_nhwc=feat, proposals=props, counts=counts, offsets=offsets)
            return {"top_region_boxes": boxes, "top_scores": scores}, top, cd
        # The samplers' draws: of an image's positives (negatives) the ones with the SMALLEST keys are taken, lower index
        # first on ties - a uniformly random subset for i.i.d. keys, like torchvision's positive[randperm(|positive|)[:k]].
        # Default: Philox words generated in the kernel from a seed drawn from torch's CPU generator (torch.manual_seed
        # governs it, no device round trip); keys_fn(stage, B, n) -> fp32 [B, n] injects the keys (tests).
        props, counts, offsets, head = self.rpn(feat, return_head=True, feat16=feat16, f16=f16)
        gt, gt_count, gt_labels = self._pad_targets(targets, images.device)
        loss_obj, loss_rpn_box = self._rpn_losses(head, gt, gt_count, keys_fn, taps)
        props_s, offsets_s, labels_s, reg_s = self._select_training_samples(props, counts, gt, gt_count, gt_labels, keys_fn, taps=taps)
        t2 = {} if taps is None else taps
        cd, scores, boxes, top = self.roi_heads(feat, props_s, offsets_s, t2, bf16)
        pred = t2.get("pred")
        R = 0 if pred is None else int(pred.shape[0])   # known on the host since roi_heads sized its launches (its one sync)
        if R == 0:
            # no sampled RoI at all: cross_entropy / smooth_l1 over empty tensors are nan in the reference as well
            out2 = torch.full((2,), float("nan"), dtype=torch.float32, device=images.device)
        else:
            out2 = torch.empty((2,), dtype=torch.float32, device=images.device)
            _hip.check(self.lib.rgrg_fastrcnn_loss_f32(_hip.ptr(pred), pred.shape[1], 30, _hip.ptr(labels_s), _hip.ptr(reg_s), R,
                                                       _hip.ptr(out2), self._s()), "rgrg_fastrcnn_loss_f32")
        # the reference's dict order: RoI-head losses, then the RPN's (object_detector.py:240-242)
        losses = {"loss_classifier": out2[0], "loss_box_reg": out2[1], "loss_objectness": loss_obj, "loss_rpn_box_reg": loss_rpn_box}
        if taps is not None:
            taps.update(features_nhwc=feat, proposals=props_s, offsets=offsets_s, sampled_labels=labels_s[:R], sampled_reg_targets=reg_s[:R])
        return losses, {"top_region_boxes": boxes, "top_scores": scores}, top, cd

    # ------------------------------------------------------------------ detector targets / losses (eval forward with image_targets)
    @staticmethod
    def _pad_targets(targets, dev):
        """list of {"boxes", "labels"} -> gt [B,G,4] fp32 and labels [B,G] int64 (zero padded), gt_count int32 [B].  The sizes
        come from the tensors' shapes (host metadata): one cat + one scatter, no device read-back."""
        B = len(targets)
        sizes = [int(t["boxes"].shape[0]) for t in targets]
        G = max(1, max(sizes))
        gt = torch.zeros((B, G, 4), dtype=torch.float32, device=dev)
        lab = torch.zeros((B, G), dtype=torch.int64, device=dev)
        if sum(sizes):
            row = torch.repeat_interleave(torch.arange(B), torch.tensor(sizes)).to(dev)
            col = torch.cat([torch.arange(n) for n in sizes]).to(dev)
            gt[row, col] = torch.cat([t["boxes"].to(device=dev, dtype=torch.float32).reshape(-1, 4) for t in targets])
            lab[row, col] = torch.cat([t["labels"].to(device=dev, dtype=torch.int64).reshape(-1) for t in targets])
        return gt, torch.tensor(sizes, dtype=torch.int32, device=dev), lab

    def _match(self, gt, gt_count, boxes, box_image_stride, box_count, N, high, low, allow_low_quality) -> Tensor:
        B, G = gt.shape[:2]
        matched = torch.empty((B, N), dtype=torch.int32, device=gt.device)
        ws = torch.empty((B, G), dtype=torch.int32, device=gt.device)
        _hip.check(self.lib.rgrg_box_match_f32(_hip.ptr(gt), _hip.ptr(gt_count), G, _hip.ptr(boxes), box_image_stride,
                                               None if box_count is None else _hip.ptr(box_count), B, N, high, low,
                                               1 if allow_low_quality else 0, _hip.ptr(matched), _hip.ptr(ws), self._s()),
                   "rgrg_box_match_f32")
        return matched

    def _sample(self, stage: str, matched: Tensor, gt_labels, box_count, batch: int, max_pos: int, keys_fn):
        """BalancedPositiveNegativeSampler on the device (rgrg_balanced_sample): -> mask uint8 [B,n] (1 sampled positive,
        2 sampled negative), list int32 [B,batch] (sampled indices ascending), count int32 [B]."""
        B, n = matched.shape
        dev = matched.device
        keys = None if keys_fn is None else keys_fn(stage, B, n).to(device=dev, dtype=torch.float32).contiguous()
        seed = 0 if keys is not None else int(torch.empty((), dtype=torch.int64).random_().item())
        ws = torch.empty((B, n), dtype=torch.int32, device=dev)
        mask = torch.empty((B, n), dtype=torch.uint8, device=dev)
        lst = torch.empty((B, batch), dtype=torch.int32, device=dev)
        cnt = torch.empty((B,), dtype=torch.int32, device=dev)
        G = 0 if gt_labels is None else gt_labels.shape[1]
        _hip.check(self.lib.rgrg_balanced_sample(_hip.ptr(matched), _hip.ptr(gt_labels), G, _hip.ptr(box_count), _hip.ptr(keys), seed,
                                                 {"rpn": 0, "roi": 1}[stage], B, n, batch, max_pos, _hip.ptr(ws), _hip.ptr(mask),
                                                 _hip.ptr(lst), _hip.ptr(cnt), self._s()), "rgrg_balanced_sample")
        return mask, lst, cnt

    def _rpn_losses(self, head: Tensor, gt: Tensor, gt_count: Tensor, keys_fn, taps=None):
        """RegionProposalNetwork.assign_targets_to_anchors + encode + compute_loss (custom_rpn.py:74-83;
        object_detector.py:84-96: fg 0.7 / bg 0.3, low-quality matches, 256 anchors per image, half positive): match ->
        sample -> loss, three HIP entry points on the stream, no torch arithmetic and no host synchronisation; the loss
        kernel derives label and regression target of the <= 256 sampled anchors per image itself."""
        B, FH, FW, ld = head.shape
        A = self.anchors.shape[0]
        m = self._match(gt, gt_count, self.anchors, 0, None, A, 0.7, 0.3, True)
        mask, lst, cnt = self._sample("rpn", m, None, None, 256, 128, keys_fn)
        out2 = torch.empty((2,), dtype=torch.float32, device=head.device)
        _hip.check(self.lib.rgrg_rpn_loss_sampled_f32(_hip.ptr(head), ld, self.num_anchors, _hip.ptr(m), _hip.ptr(gt), gt.shape[1],
                                                      _hip.ptr(self.anchors), _hip.ptr(mask), _hip.ptr(lst), _hip.ptr(cnt), B, A, 256,
                                                      _hip.ptr(out2), self._s()), "rgrg_rpn_loss_sampled_f32")
        if taps is not None:
            taps.update(rpn_matched=m, rpn_sampled=mask, rpn_sampled_list=lst, rpn_sampled_count=cnt)
        return out2[0], out2[1]

    def _select_training_samples(self, props: Tensor, counts: Tensor, gt: Tensor, gt_count: Tensor, gt_labels: Tensor, keys_fn,
                                 K: int = 512, taps=None):
        """RoIHeads.select_training_samples (custom_roi_heads.py:225-226; object_detector.py:118-123: the ground-truth boxes
        join the proposals, IoU 0.5, K = 512 per image, a quarter positive, BoxCoder weights (10,10,5,5)) for a whole batch
        with static shapes and no host synchronisation: add_gt -> match -> sample -> gather, HIP kernels throughout.
        props [B,P,4] with counts [B] valid rows, gt [B,G,4] / gt_labels [B,G] with gt_count [B] valid rows -> proposals
        [B,K,4] (the sampled ones of an image first, in index order, zero padded), offsets int32 [B+1] (device), labels
        int64 [B*K] and regression targets [B*K,4] in RoI order = image-major compaction (the first offsets[B] rows are
        meaningful, the rest zero)."""
        B, P = props.shape[:2]
        G = gt.shape[1]
        N = P + G
        dev = props.device
        props = props.to(torch.float32).contiguous()
        boxes = torch.empty((B, N, 4), dtype=torch.float32, device=dev)
        box_count = torch.empty((B,), dtype=torch.int32, device=dev)
        _hip.check(self.lib.rgrg_roi_add_gt_f32(_hip.ptr(props), _hip.ptr(counts), P, _hip.ptr(gt), _hip.ptr(gt_count), G, B,
                                                _hip.ptr(boxes), _hip.ptr(box_count), self._s()), "rgrg_roi_add_gt_f32")
        m = self._match(gt, gt_count, boxes, N * 4, box_count, N, 0.5, 0.5, False)
        mask, lst, cnt = self._sample("roi", m, gt_labels, box_count, K, K // 4, keys_fn)
        props_s = torch.empty((B, K, 4), dtype=torch.float32, device=dev)
        offsets = torch.empty((B + 1,), dtype=torch.int32, device=dev)
        labels_flat = torch.zeros((B * K,), dtype=torch.int64, device=dev)
        reg = torch.zeros((B * K, 4), dtype=torch.float32, device=dev)
        _hip.check(self.lib.rgrg_roi_gather_samples_f32(_hip.ptr(boxes), _hip.ptr(m), _hip.ptr(gt), _hip.ptr(gt_labels), _hip.ptr(gt_count),
                                                        G, _hip.ptr(lst), _hip.ptr(cnt), B, N, K, 10.0, 10.0, 5.0, 5.0, _hip.ptr(props_s),
                                                        _hip.ptr(offsets), _hip.ptr(labels_flat), _hip.ptr(reg), self._s()),
                   "rgrg_roi_gather_samples_f32")
        if taps is not None:
            taps.update(roi_boxes=boxes, roi_box_count=box_count, roi_matched=m, roi_sampled=mask, roi_sampled_list=lst)
        return props_s, offsets, labels_flat, reg

    # ------------------------------------------------------------------ selection
    def classifier_logits(self, mlp, x: Tensor) -> Tensor:
        """The 1024-512-128-1 ReLU MLP shared by both region classifiers: x [n,1024] -> logits [n]."""
        h = self.linear(x, *mlp[0], act=_hip.ACT_RELU)
        h = self.linear(h, *mlp[1], act=_hip.ACT_RELU)
        return self.linear(h, *mlp[2]).view(-1)

    def bce_masked(self, logits: Tensor, mask: Tensor, target: Tensor, pos_weight: float) -> Tensor:
        """BCEWithLogitsLoss(pos_weight)(logits[mask], target[mask]) -> float32 scalar tensor."""
        n = logits.numel()
        m = mask.reshape(-1).to(torch.uint8).contiguous()
        t = target.reshape(-1).to(torch.uint8).contiguous()
        loss = torch.empty((), dtype=torch.float32, device=logits.device)
        _hip.check(self.lib.rgrg_bce_with_logits_masked_f32(_hip.ptr(logits), _hip.ptr(m), _hip.ptr(t), float(pos_weight), n,
                                                            _hip.ptr(loss), self._s()), "rgrg_bce_with_logits_masked_f32")
        return loss

    def _transpose_pad(self, x: Tensor, rows_padded: int) -> Tensor:
        R, Cc = x.shape
        out = torch.empty((Cc, rows_padded), dtype=torch.float32, device=x.device)
        _hip.check(self.lib.rgrg_transpose_pad_f32(_hip.ptr(x), _hip.ptr(out), R, Cc, rows_padded, self._s()), "rgrg_transpose_pad_f32")
        return out

    def _colsum(self, x: Tensor) -> Tensor:
        out = torch.empty((x.shape[1],), dtype=torch.float32, device=x.device)
        _hip.check(self.lib.rgrg_colsum_f32(_hip.ptr(x), _hip.ptr(out), x.shape[0], x.shape[1], self._s()), "rgrg_colsum_f32")
        return out

    def classifier_loss_grad(self, mlp, x: Tensor, mask: Tensor, target: Tensor, pos_weight: float):
        """Loss of one region classifier (1024-512-128-1 ReLU MLP + BCEWithLogits(pos_weight) over ``mask``) and its
        gradients w.r.t. the six parameter tensors, all on the HIP kernels: -> (loss, logits, [dW0, db0, dW2, db2, dW4, db4])."""
        n = x.shape[0]
        npad = (n + 31) // 32 * 32
        (W0, b0), (W2, b2), (W4, b4) = mlp
        h1 = self.linear(x, W0, b0, act=_hip.ACT_RELU)
        h2 = self.linear(h1, W2, b2, act=_hip.ACT_RELU)
        logits = self.linear(h2, W4, b4).view(-1)
        loss = self.bce_masked(logits, mask, target, pos_weight)
        m = mask.reshape(-1).to(torch.uint8).contiguous()
        t = target.reshape(-1).to(torch.uint8).contiguous()
        dlog = torch.zeros((n, 32), dtype=torch.float32, device=x.device)  # column 0 = d logits, 31 columns of K padding
        _hip.check(self.lib.rgrg_bce_with_logits_masked_backward_f32(_hip.ptr(logits), _hip.ptr(m), _hip.ptr(t), float(pos_weight), n,
                                                                     1.0, _hip.ptr(dlog), 32, self._s()),
                   "rgrg_bce_with_logits_masked_backward_f32")
        dlogT = self._transpose_pad(dlog, npad)                       # [32, npad], row 0 = d logits
        dW4 = self.linear(dlogT[:1].contiguous(), self._transpose_pad(h2, npad), None)          # [1,128]
        db4 = self._colsum(dlog)[:1].contiguous()
        dh2 = self.linear(dlog, self._transpose_pad(W4, 32), None)   # [n,128] = dlog W4
        _hip.check(self.lib.rgrg_relu_backward_f32(_hip.ptr(dh2), _hip.ptr(h2), dh2.numel(), self._s()), "rgrg_relu_backward_f32")
        dW2 = self.linear(self._transpose_pad(dh2, npad), self._transpose_pad(h1, npad), None)   # [128,512]
        db2 = self._colsum(dh2)
        dh1 = self.linear(dh2, self._transpose_pad(W2, W2.shape[0]), None)                        # [n,512] = dh2 W2
        _hip.check(self.lib.rgrg_relu_backward_f32(_hip.ptr(dh1), _hip.ptr(h1), dh1.numel(), self._s()), "rgrg_relu_backward_f32")
        dW0 = self.linear(self._transpose_pad(dh1, npad), self._transpose_pad(x, npad), None)    # [512,1024]
        db0 = self._colsum(dh1)
        return loss, logits, [dW0, db0, dW2, db2, dW4, db4]

    def abnormal(self, top_region_features: Tensor, class_detected: Tensor, region_is_abnormal: Tensor, pos_weight: float):
        """BinaryClassifierRegionAbnormal.forward, eval: -> (loss, predicted_abnormal_regions bool [B,29])."""
        if self.abn is None:
            raise RuntimeError("the loaded state dict has no binary_classifier_region_abnormal.* weights")
        B, Rg, D = top_region_features.shape
        x = top_region_features.reshape(B * Rg, D).contiguous().to(torch.float32)
        logits = self.classifier_logits(self.abn, x)
        loss = self.bce_masked(logits, class_detected, region_is_abnormal, pos_weight)
        n = B * Rg
        ones = torch.ones((n,), dtype=torch.uint8, device=x.device)
        pred = torch.empty((n,), dtype=torch.uint8, device=x.device)
        rows = torch.empty((n,), dtype=torch.int32, device=x.device)
        cnt = torch.empty((1,), dtype=torch.int32, device=x.device)
        _hip.check(self.lib.rgrg_select_regions_f32(_hip.ptr(logits), _hip.ptr(ones), SELECTION_LOGIT_THRESHOLD, _hip.ptr(pred),
                                                    _hip.ptr(rows), _hip.ptr(cnt), n, self._s()), "rgrg_select_regions_f32")
        return loss, pred.view(B, Rg).bool()

    def select(self, top_region_features: Tensor, class_detected: Tensor, taps: Optional[dict] = None):
        B, Rg, D = top_region_features.shape
        x = top_region_features.reshape(B * Rg, D).contiguous().to(torch.float32)
        logits = self.classifier_logits(self.sel, x)
        n = B * Rg
        det = class_detected.reshape(-1).to(torch.uint8).contiguous()
        sel = torch.empty((n,), dtype=torch.uint8, device=x.device)
        rows = torch.empty((n,), dtype=torch.int32, device=x.device)
        nsel = torch.empty((1,), dtype=torch.int32, device=x.device)
        _hip.check(self.lib.rgrg_select_regions_f32(_hip.ptr(logits), _hip.ptr(det), SELECTION_LOGIT_THRESHOLD,
                                                    _hip.ptr(sel), _hip.ptr(rows), _hip.ptr(nsel), n, self._s()),
                   "rgrg_select_regions_f32")
        S = int(nsel.item())  # host sync #2 (the reference syncs here too: report_generation_model.py:260)
        feats = torch.empty((S, D), dtype=torch.float32, device=x.device)
        if S > 0:
            _hip.check(self.lib.rgrg_gather_rows_f32(_hip.ptr(x), _hip.ptr(rows), _hip.ptr(feats), S, D, self._s()),
                       "rgrg_gather_rows_f32")
        if taps is not None:
            taps.update(selection_logits=logits.view(B, Rg))
        return sel.view(B, Rg).bool(), feats

    # ------------------------------------------------------------------ decoder
    def _get_decoder(self, S: int, max_len: int):
        cap_s, cap_l = self._decoder_caps
        if self._decoder is None or S > cap_s or max_len > cap_l:
            self.close()
            cap_s = max(32, ((S + 31) // 32) * 32, cap_s)
            cap_l = max(max_len, cap_l)
            h = C.c_void_p()
            nbytes = int(self.lib.rgrg_decoder_kv_cache_bytes(self.n_layer, cap_s, cap_l))
            kv = torch.zeros((self.n_layer, 2, cap_s, 16, cap_l + 1, 64), dtype=torch.float32, device=self.device)
            assert kv.numel() * 4 == nbytes
            torch.cuda.current_stream(self.device).synchronize()   # the zero fill is complete before the decoder's own stream uses the cache
            _hip.check(self.lib.rgrg_decoder_create_with_cache(C.byref(self._dec_weights), cap_s, cap_l, _hip.ptr(kv), nbytes, C.byref(h)),
                       "rgrg_decoder_create_with_cache")
            self._decoder, self._decoder_caps, self._kv = h, (cap_s, cap_l), kv
        return self._decoder

    def greedy_decode(self, feats: Tensor, max_length: Optional[int], use_graph: bool = True, bf16=False) -> Tensor:
        """LanguageModel.generate(num_beams=1): feats [S,1024] -> int64 [S, L'].  bf16 = precision mode (opt-in through
        torch.autocast: False / 0 fp32, True / 1 bfloat16, 2 float16): lets the > 128-sequence path use 16-bit-weight MFMA GEMMs
        and a 16-bit K/V cache of that type (not bit-exact)."""
        _require_gpu(feats.device)
        S = feats.shape[0]
        limit = int(max_length) if max_length else 1024  # reference has no bound when None; positions stop at 1024
        dec = self._get_decoder(S, limit)
        self._cached = None   # the K/V cache and the step counter are rewritten: presents of an earlier forward(use_cache=True) are stale
        _hip.check(self.lib.rgrg_decoder_set_precision(dec, int(bf16)), "rgrg_decoder_set_precision")
        feats = feats.to(torch.float32).contiguous()
        out = torch.empty((S, limit), dtype=torch.int64, device=feats.device)
        out_len = C.c_int(0)
        _hip.check(self.lib.rgrg_decoder_generate(dec, _hip.ptr(feats), S, limit, _hip.ptr(out), limit,
                                                  C.byref(out_len), 1 if use_graph else 0, self._s()),
                   "rgrg_decoder_generate")
        return out[:, :out_len.value].contiguous()

    def beam_search(self, feats: Tensor, max_length: int, num_beams: int, early_stopping: bool = False,
                    length_penalty: float = 1.0, bf16=False, num_return_sequences: int = 1) -> Tensor:
        """LanguageModel.generate(num_beams>1): feats [S,1024] -> int64 [S * num_return_sequences, L]."""
        _require_gpu(feats.device)
        S = feats.shape[0]
        limit = int(max_length)
        dec = self._get_decoder(S * num_beams, limit)
        self._cached = None   # as in greedy_decode
        _hip.check(self.lib.rgrg_decoder_set_precision(dec, int(bf16)), "rgrg_decoder_set_precision")
        feats = feats.to(torch.float32).contiguous()
        out = torch.empty((S * int(num_return_sequences), limit), dtype=torch.int64, device=feats.device)
        out_len = C.c_int(0)
        _hip.check(self.lib.rgrg_decoder_beam_search(dec, _hip.ptr(feats), S, int(num_beams), limit, 1 if early_stopping else 0,
                                                     float(length_penalty), int(num_return_sequences), _hip.ptr(out), limit,
                                                     C.byref(out_len), self._s()),
                   "rgrg_decoder_beam_search")
        return out[:, :out_len.value].contiguous()

    def _raise_pending_id_error(self) -> None:
        """An unreported token-id error of a decoder that has been re-created since (close())."""
        if getattr(self, "_pending_id_error", False):
            self._pending_id_error = False
            raise IndexError("index out of range in self: a token id of a previous teacher-forced pass was outside "
                             f"[0, {self.vocab})")

    @staticmethod
    def _check_ids(rc: int, what: str) -> None:
        """Token ids are validated on the device: an out-of-range id poisons that pass's loss (NaN) and is reported by
        the NEXT decoder call as torch.nn.Embedding's IndexError (the reference fails in the same call; checking there
        would cost a device-to-host round trip per call)."""
        try:
            _hip.check(rc, what)
        except _hip.RgrgHipError as e:
            if "index out of range in self" in str(e):
                raise IndexError(str(e)) from None
            raise

    def _set_lm_positions(self, dec, position_ids: Optional[Tensor], S: int, T: int, device):
        """``position_ids`` of a teacher-forced pass ([S,T], or anything that views as [1,T]; None = arange(T)): handed to the
        decoder for its next pass; the tensor is returned so that the caller keeps it alive across the launch."""
        if position_ids is None:
            _hip.check(self.lib.rgrg_decoder_set_lm_positions(dec, None, 0), "rgrg_decoder_set_lm_positions")
            return None
        pos = position_ids.reshape(-1, T).to(device=device, dtype=torch.int64).contiguous()
        if pos.shape[0] not in (1, S):
            raise ValueError(f"position_ids has {pos.shape[0]} rows, expected 1 or {S}")
        _hip.check(self.lib.rgrg_decoder_set_lm_positions(dec, _hip.ptr(pos), pos.numel()), "rgrg_decoder_set_lm_positions")
        return pos

    def lm_forward(self, feats: Tensor, input_ids: Tensor, attention_mask: Optional[Tensor], want_logits: bool = False,
                   want_loss: bool = True, bf16=False, position_ids: Optional[Tensor] = None) -> Tuple[Optional[Tensor], Optional[Tensor]]:
        """LanguageModel.forward without cache (teacher forcing): feats [S,1024], input_ids int64 [S,T],
        attention_mask [S,T] -> (logits f32 [S,T,V] or None, loss f32 scalar tensor or None)."""
        _require_gpu(feats.device)
        S, T = input_ids.shape
        if feats.shape[0] != S:
            raise ValueError(f"image_hidden_states has {feats.shape[0]} rows, input_ids {S}")
        if T > 1023:  # T + 1 keys: the 1024 positions of GPT-2's causal-mask buffer bound the reference as well
            raise NotImplementedError("the teacher-forced pass supports sequences of up to 1023 tokens")
        # token ids are range-checked on the device (no host sync here): see embed_seq_ln_kernel
        dec = self._get_decoder(S, 2)
        self._raise_pending_id_error()
        _hip.check(self.lib.rgrg_decoder_set_precision(dec, int(bf16)), "rgrg_decoder_set_precision")
        feats = feats.to(torch.float32).contiguous()
        ids = input_ids.to(torch.int64).contiguous()
        am = None if attention_mask is None else attention_mask.to(torch.float32).contiguous()
        logits = torch.empty((S, T, self.vocab), dtype=torch.float32, device=feats.device) if want_logits else None
        loss = torch.empty((), dtype=torch.float32, device=feats.device) if want_loss else None
        _pos = self._set_lm_positions(dec, position_ids, S, T, feats.device)  # noqa: F841 (kept alive across the launch)
        self._check_ids(self.lib.rgrg_decoder_lm_forward(dec, _hip.ptr(feats), _hip.ptr(ids), None if am is None else _hip.ptr(am),
                                                         S, T, None if logits is None else _hip.ptr(logits),
                                                         None if loss is None else _hip.ptr(loss), self._s()),
                        "rgrg_decoder_lm_forward")
        return logits, loss

    def lm_loss_grad(self, feats: Tensor, input_ids: Tensor, attention_mask: Optional[Tensor], loss_scale: float = 1.0,
                     bf16=False, dropout_p: float = 0.0, dropout_seed: int = 0, position_ids: Optional[Tensor] = None):
        """Teacher-forced loss and its gradients w.r.t. the trainable decoder weights (rgrg_decoder_lm_loss_grad):
        -> (loss, {"ukv_w" [L*2*1024,1024], "ukv_b", "fst0_w", "fst0_b", "fst2_w", "fst2_b"}).  bf16=True (torch.autocast,
        as the reference's training loop uses): the frozen-weight GEMMs of forward and backward run on the bf16 MFMA
        when there are more than 128 token rows; LayerNorm, softmax, cross entropy and the weight gradients stay fp32."""
        _require_gpu(feats.device)
        S, T = input_ids.shape
        if feats.shape[0] != S:
            raise ValueError(f"image_hidden_states has {feats.shape[0]} rows, input_ids {S}")
        if T > 1023:  # T + 1 keys: the 1024 positions of GPT-2's causal-mask buffer bound the reference as well
            raise NotImplementedError("the training pass supports sequences of up to 1023 tokens")
        # token ids are range-checked on the device (no host sync here): see embed_seq_ln_kernel
        dec = self._get_decoder(S, 2)
        self._raise_pending_id_error()
        _hip.check(self.lib.rgrg_decoder_set_precision(dec, int(bf16)), "rgrg_decoder_set_precision")
        feats = feats.detach().to(torch.float32).contiguous()
        ids = input_ids.to(torch.int64).contiguous()
        am = None if attention_mask is None else attention_mask.to(torch.float32).contiguous()
        dev, LD = feats.device, self.n_layer * 2 * 1024
        g = {"ukv_w": torch.empty((LD, 1024), dtype=torch.float32, device=dev), "ukv_b": torch.empty((LD,), dtype=torch.float32, device=dev),
             "fst0_w": torch.empty((1024, 1024), dtype=torch.float32, device=dev), "fst0_b": torch.empty((1024,), dtype=torch.float32, device=dev),
             "fst2_w": torch.empty((1024, 1024), dtype=torch.float32, device=dev), "fst2_b": torch.empty((1024,), dtype=torch.float32, device=dev)}
        loss = torch.empty((), dtype=torch.float32, device=dev)
        self.last_train_shape = (int(S), int(T))   # (sentences, tokens) of the last training pass: bench.py times its GEMMs at this shape
        _pos = self._set_lm_positions(dec, position_ids, S, T, dev)  # noqa: F841 (kept alive across the launch)
        self._check_ids(self.lib.rgrg_decoder_lm_loss_grad(dec, _hip.ptr(feats), _hip.ptr(ids), None if am is None else _hip.ptr(am), S, T,
                                                           float(loss_scale), float(dropout_p), int(dropout_seed) & (2 ** 64 - 1), _hip.ptr(loss),
                                                           _hip.ptr(g["ukv_w"]), _hip.ptr(g["ukv_b"]),
                                                           _hip.ptr(g["fst0_w"]), _hip.ptr(g["fst0_b"]), _hip.ptr(g["fst2_w"]),
                                                           _hip.ptr(g["fst2_b"]), self._s()), "rgrg_decoder_lm_loss_grad")
        return loss, g

    def dropout_mask(self, seed: int, layer: int, site: int, p: float, shape) -> Tensor:
        """The training pass's dropout mask of one site (0 or 1/(1-p)); sites: 0 embedding, 1 attention probabilities
        [S,16,T,T+1], 2 attn c_proj output, 3 mlp c_proj output ([S*T,1024])."""
        out = torch.empty(shape, dtype=torch.float32, device=self.device)
        row_len = int(shape[-1]) if site == 1 else 0   # attention masks: the key pitch of the index is a multiple of 4
        _hip.check(self.lib.rgrg_dropout_mask_f32(int(seed) & (2 ** 64 - 1), layer * 4 + site, float(p), out.numel(), row_len,
                                                  _hip.ptr(out), self._s()), "rgrg_dropout_mask_f32")
        return out

    def sync_trainable(self, state_dict: Dict[str, Tensor]) -> None:
        """Copy the (optimizer-updated) trainable decoder parameters into the engine's buffers and rebuild the
        kernel-side layouts derived from them."""
        g = "language_model.gpt_with_lm_head.transformer."
        f = "language_model.feature_space_transformation_nn."
        D = 1024
        with torch.no_grad():
            for l in range(self.n_layer):
                b = f"{g}h.{l}.attn."
                self._ukv_w[(2 * l) * D:(2 * l + 1) * D].copy_(state_dict[b + "uk.weight"])
                self._ukv_w[(2 * l + 1) * D:(2 * l + 2) * D].copy_(state_dict[b + "uv.weight"])
                self._ukv_b[(2 * l) * D:(2 * l + 1) * D].copy_(state_dict[b + "uk.bias"])
                self._ukv_b[(2 * l + 1) * D:(2 * l + 2) * D].copy_(state_dict[b + "uv.bias"])
            for t, k in zip(self._fst, ("0.weight", "0.bias", "2.weight", "2.bias")):
                if t.data_ptr() != state_dict[f + k].data_ptr():
                    t.copy_(state_dict[f + k])
        if self._decoder is not None:
            _hip.check(self.lib.rgrg_decoder_refresh_trainable(self._decoder, self._s()), "rgrg_decoder_refresh_trainable")

    def cache_tokens_that_fit(self, S: int, want: int, budget_fraction: float = 0.5) -> int:
        """How many token slots a K/V cache for S rows may have so that the allocation (24 x 2 x rows x 16 x (L + 1) x 64 x 4 B)
        stays within ``budget_fraction`` of the free device memory: the reference's 1024 positions for a few dozen rows (6.4 GB at
        32 rows), fewer for hundreds of rows (928 rows x 1024 slots would be 187 GB, and the decoder never shrinks)."""
        rows = max(32, ((S + 31) // 32) * 32)
        per_slot = self.n_layer * 2 * rows * 16 * 64 * 4
        free, _total = torch.cuda.mem_get_info(self.device)
        # blocks torch's caching allocator holds but does not use (e.g. a replaced cache) can serve the new cache tensor
        free += max(0, torch.cuda.memory_reserved(self.device) - torch.cuda.memory_allocated(self.device))
        fit = int(free * budget_fraction) // per_slot - 1
        if fit < 2:
            raise _hip.RgrgHipError(f"a K/V cache for {S} rows does not fit: {free / 2 ** 30:.1f} GiB free, {per_slot / 2 ** 20:.0f} MiB per token slot")
        return max(2, min(int(want), fit))

    def forward_cached(self, feats: Optional[Tensor], input_ids: Tensor, past_len: int, cache_len: int = 1024,
                       position_ids: Optional[Tensor] = None, adopt_past=None, attention_mask: Optional[Tensor] = None):
        """LanguageModel.forward(use_cache=True[, past_key_values]) over the decoder's K/V cache (rgrg_decoder_forward_cached):
        feeds input_ids [S,T] into cache slots past_len + 1 .. past_len + T -> (logits f32 [S,T,V], presents) where presents is
        the reference's tuple of 24 (key, value) pairs, each a VIEW [S,16,1 + past_len + T,64] of the cache.  ``position_ids``
        [S,T] (or [1,T]): the embedding positions (wte[position], language_model.py:293-307), any values inside the table; None:
        past_len + j.  ``adopt_past``: a FOREIGN past_key_values (24 pairs of [S,16,1 + past_len,64] tensors that are not views of
        this decoder's cache, e.g. clones or another model's presents): copied into the cache first.  ``cache_len`` = token slots
        to provide for when the cache is created: the reference's 1024 positions, clipped to what half of the free device memory
        holds for this many rows.  ``attention_mask`` [S, past_len + T] (None: all ones): keys whose entry is 0 get the
        reference's additive -1e4 for every query (language_model.py:316-334); the image key is never masked."""
        S, T = input_ids.shape
        # torch.nn.Embedding raises on an id outside the vocabulary in the same call; so does this path (one read-back: it is
        # the incremental API, not the generate loop)
        idc = input_ids.to(device=self.device)
        if bool(((idc < 0) | (idc >= self.vocab)).any()):
            raise IndexError(f"index out of range in self: a token id is outside [0, {self.vocab})")
        pos = None
        if position_ids is not None:
            pos = position_ids.to(device=self.device, dtype=torch.int64).reshape(-1, T)
            if pos.shape[0] not in (1, S):
                raise ValueError(f"position_ids has {pos.shape[0]} rows, input_ids {S}")
            if bool(((pos < 0) | (pos >= self.vocab)).any()):   # positions index wte as well (the quirk, :307)
                raise IndexError(f"index out of range in self: a position id is outside [0, {self.vocab})")
            pos = pos.expand(S, T).contiguous()
        if past_len == 0 or adopt_past is not None:
            # an adopted past that holds only the image slot (shape [.., 1, 64]): the supplied key / value of slot 0 is used and the
            # image is ignored, like the reference does with any past_key_values (language_model.py:162-166)
            if past_len == 0 and adopt_past is None and (feats is None or feats.shape[0] != S):
                raise ValueError("image_hidden_states [S,1024] is needed when past_key_values is None")
            if adopt_past is not None:
                feats = None
                for k, v in adopt_past:
                    if not (torch.is_tensor(k) and torch.is_tensor(v) and k.is_floating_point() and v.is_floating_point()):
                        raise TypeError("past_key_values must hold floating-point tensors")
            _require_gpu(feats.device if feats is not None else self.device)
            want = max(self.cache_tokens_that_fit(S, cache_len), past_len + T)
            if self._decoder is not None and S <= self._decoder_caps[0]:
                want = min(want, max(self._decoder_caps[1], past_len + T))   # an existing decoder is reused as is; it grows only when it must
            dec = self._get_decoder(S, want)
            if adopt_past is not None:
                if len(adopt_past) != self.n_layer:
                    raise ValueError(f"past_key_values has {len(adopt_past)} layers, the model {self.n_layer}")
                for l, (k, v) in enumerate(adopt_past):
                    if tuple(k.shape) != (S, 16, 1 + past_len, 64) or tuple(v.shape) != tuple(k.shape):
                        raise ValueError(f"past_key_values[{l}] has shape {tuple(k.shape)}, expected {(S, 16, 1 + past_len, 64)}")
                    self._kv[l, 0, :S, :, :1 + past_len].copy_(k.to(device=self.device, dtype=torch.float32))
                    self._kv[l, 1, :S, :, :1 + past_len].copy_(v.to(device=self.device, dtype=torch.float32))
            self._cached = {"S": S, "tokens": past_len}
        else:
            c = getattr(self, "_cached", None)
            if self._decoder is None or c is None or c["S"] != S or c["tokens"] != past_len:
                raise NotImplementedError("past_key_values must be the presents returned by the previous forward(use_cache=True) "
                                          "of this model (the cache lives in the HIP decoder)")
            dec = self._decoder
        if past_len + T > self._decoder_caps[1]:
            # the chain outgrows the cache: a larger decoder (at least twice the slots, at most what the memory budget holds) takes
            # over the cached keys / values; presents handed out so far keep their (old) tensor and would be adopted by copy
            need = past_len + T
            fit = self.cache_tokens_that_fit(S, max(cache_len, need))
            if need > fit:
                raise NotImplementedError(f"a K/V cache of {need} tokens for {S} rows does not fit in half of the free device memory "
                                          f"({fit} tokens do)")
            old = self._kv
            dec = self._get_decoder(S, min(max(2 * self._decoder_caps[1], need), fit))
            if past_len:
                self._kv[:, :, :S, :, :1 + past_len].copy_(old[:, :, :S, :, :1 + past_len])
                torch.cuda.current_stream(self.device).synchronize()   # the decoder's own stream reads the cache next
            del old
            self._cached = {"S": S, "tokens": past_len}
        _hip.check(self.lib.rgrg_decoder_set_precision(dec, 0), "rgrg_decoder_set_precision")
        ids = input_ids.to(device=self.device, dtype=torch.int64).contiguous()
        f = None if (past_len or feats is None) else feats.to(torch.float32).contiguous()
        logits = torch.empty((S, T, self.vocab), dtype=torch.float32, device=self.device)
        am = None
        if attention_mask is not None:
            am = attention_mask.to(device=self.device, dtype=torch.float32).reshape(S, -1).contiguous()
            if am.shape[1] != past_len + T:   # the reference broadcasts [S,1,1,1+L] against scores [S,16,T,1+past+T]
                raise ValueError(f"attention_mask covers {am.shape[1]} tokens, the keys of this call are {past_len} cached + {T} new")
        _hip.check(self.lib.rgrg_decoder_forward_cached(dec, _hip.ptr(f), _hip.ptr(ids), _hip.ptr(pos), _hip.ptr(am), S, T, int(past_len),
                                                        _hip.ptr(logits), self._s()), "rgrg_decoder_forward_cached")
        self._cached["tokens"] = past_len + T
        return logits, self._cache_views(dec, S, 1 + past_len + T)

    def _cache_views(self, dec, S: int, n_keys: int):
        """The reference's ``presents``: 24 (key, value) pairs [S, 16, n_keys, 64], views of the cache tensor."""
        return tuple((self._kv[l, 0, :S, :, :n_keys], self._kv[l, 1, :S, :, :n_keys]) for l in range(self.n_layer))

    def owns_cache(self, past_key_values) -> Optional[int]:
        """Number of tokens cached if ``past_key_values`` are this decoder's own presents (as returned by the previous
        forward_cached), else None."""
        c = getattr(self, "_cached", None)
        if self._decoder is None or c is None or past_key_values is None:
            return None
        try:
            k0 = past_key_values[0][0]
            if (self._kv is not None and k0.data_ptr() == self._kv.data_ptr() and k0.shape[0] == c["S"] and k0.shape[-2] == 1 + c["tokens"]
                    and len(past_key_values) == self.n_layer):
                return c["tokens"]
        except Exception:  # noqa: BLE001
            return None
        return None

    def aliases_cache(self, past_key_values) -> bool:
        """True when the tensors live in this decoder's cache allocation (views of it) - valid only as its current presents."""
        try:
            st = past_key_values[0][0].untyped_storage().data_ptr()
            return self._kv is not None and st == self._kv.untyped_storage().data_ptr()
        except Exception:  # noqa: BLE001
            return False

    def last_logits(self, S: int) -> Tensor:
        dst = torch.empty((S, self.vocab), dtype=torch.float32, device=self.device)
        _hip.check(self.lib.rgrg_decoder_copy_last_logits(self._decoder, _hip.ptr(dst), S, self._s()), "copy_last_logits")
        return dst

    def time_train_gemms(self, S: int, T: int, iters: int = 3) -> Dict[str, float]:
        """The frozen-weight GEMMs of one 16-bit training step (forward + activation gradients + lm_head), launched back to
        back between two HIP events on the decoder's stream, on the work space of the last training pass of this shape
        (bench.py: roofline of BASELINE configs[4])."""
        ms, fl, n = C.c_float(0), C.c_double(0), C.c_int(0)
        _hip.check(self.lib.rgrg_decoder_time_train_gemms(self._decoder, S, T, iters, C.byref(ms), C.byref(fl), C.byref(n)),
                   "rgrg_decoder_time_train_gemms")
        return {"ms_gemm": ms.value, "gemm_flops": fl.value, "gemm_launches": n.value}

    def fused_row_limit(self) -> int:
        """Token rows up to which the decoder of the last generate() runs the fused plan in its precision mode (above: many-sequence path)."""
        return int(self.lib.rgrg_decoder_row_limit(self._decoder))

    def time_step_parts(self, S: int, nkeys: int, iters: int = 3, one_range: bool = False) -> Dict[str, float]:
        """Per decode step, measured with HIP events on the decoder's stream (bench.py roofline): ms spent in the
        projection GEMM launches and in the 24 attention launches (at ``nkeys`` keys), with the algorithmic flops /
        weight bytes / K/V bytes of one step.  Uses the decoder of the last generate() (its precision mode).  ``one_range``:
        every launch covers all S rows (the kernels alone on the GPU) instead of the step's concurrent row ranges."""
        mg, ma, fl, wb, kv, n = C.c_float(0), C.c_float(0), C.c_double(0), C.c_double(0), C.c_double(0), C.c_int(0)
        _hip.check(self.lib.rgrg_decoder_time_step_parts(self._decoder, S, nkeys, iters, int(bool(one_range)), C.byref(mg), C.byref(ma), C.byref(fl),
                                                         C.byref(wb), C.byref(kv), C.byref(n)), "rgrg_decoder_time_step_parts")
        return {"ms_gemm": mg.value / iters, "ms_attn": ma.value / iters, "gemm_flops": fl.value, "gemm_weight_bytes": wb.value,
                "kv_bytes": kv.value, "gemm_launches": n.value}
