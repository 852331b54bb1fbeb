"""AdamW on the HIP kernel ``rgrg_adamw_step_f32`` (torch.optim.AdamW semantics: decoupled weight decay, bias
correction), the optimizer of ``src/full_model/train_full_model.py:409``.  Same constructor / ``step`` /
``zero_grad`` / ``param_groups`` / ``state`` surface as ``torch.optim.AdamW`` (it IS a ``torch.optim.Optimizer``),
so the reference's loop, LR scheduler and checkpoint code keep working; only fp32 CUDA parameters are supported.
"""
from __future__ import annotations

import torch

from . import _hip


class AdamW(torch.optim.Optimizer):
    def __init__(self, params, lr: float = 1e-3, betas=(0.9, 0.999), eps: float = 1e-8, weight_decay: float = 1e-2):
        if lr < 0 or eps < 0 or not 0 <= betas[0] < 1 or not 0 <= betas[1] < 1 or weight_decay < 0:
            raise ValueError("invalid AdamW hyper-parameter")
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        self._lib = None

    @torch.no_grad()
    def step(self, closure=None, grad_scale: float = 1.0):
        """``grad_scale`` multiplies every gradient inside the kernel (1 / AMP scale, 1 / accumulation steps)."""
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        for group in self.param_groups:
            b1, b2 = group["betas"]
            if self._multi_step(group, b1, b2, grad_scale):
                continue
            for p in group["params"]:
                if p.grad is None:
                    continue
                if not (p.is_cuda and p.dtype == torch.float32 and p.is_contiguous()):
                    raise _hip.RgrgHipError("rgrg_amd.optim.AdamW needs contiguous fp32 parameters on the GPU (no CPU fallback)")
                if self._lib is None:
                    self._lib = _hip.load()
                g = p.grad if p.grad.is_contiguous() else p.grad.contiguous()
                st = self.state[p]
                if not st:
                    st["step"] = 0
                    st["exp_avg"] = torch.zeros_like(p, memory_format=torch.contiguous_format)
                    st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.contiguous_format)
                st["step"] += 1
                with torch.cuda.device(p.device):  # the kernel launches on the current device: the parameter's
                    _hip.check(self._lib.rgrg_adamw_step_f32(p.data_ptr(), g.data_ptr(), st["exp_avg"].data_ptr(),
                                                             st["exp_avg_sq"].data_ptr(), p.numel(), float(group["lr"]), float(b1),
                                                             float(b2), float(group["eps"]), float(group["weight_decay"]),
                                                             int(st["step"]), float(grad_scale),
                                                             torch.cuda.current_stream(p.device).cuda_stream), "rgrg_adamw_step_f32")
                # the kernel wrote p behind torch's back: bump the version counter so that engines caching derived
                # layouts (LanguageModel.sync_trainable_if_stale) and autograd's checks notice
                torch._C._increment_version(p)
        return loss

    def _multi_step(self, group, b1, b2, grad_scale) -> bool:
        """All tensors of the group in ceil(n / 64) launches per distinct step count (rgrg_adamw_multi_step_f32) when they are
        contiguous fp32 tensors on one device - the normal case; the per-tensor loop above remains for anything else.  Tensors
        whose step counts differ (a parameter that received no gradient in an earlier step - e.g. a rank without selected
        regions) are partitioned by count, one multi launch per partition; zero-element tensors are skipped."""
        ps = [p for p in group["params"] if p.grad is not None and p.numel() > 0]
        if len(ps) < 2:
            return False
        dev = ps[0].device
        for p in ps:   # every record is validated before the first launch: a later partition cannot fail after an earlier one ran
            if not (p.is_cuda and p.device == dev and p.dtype == torch.float32 and p.is_contiguous() and p.grad.is_contiguous()
                    and p.grad.dtype == torch.float32 and p.grad.device == dev and p.grad.numel() == p.numel()):
                return False
        for p in ps:
            st = self.state[p]
            if not st:
                st["step"] = 0
                st["exp_avg"] = torch.zeros_like(p, memory_format=torch.contiguous_format)
                st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.contiguous_format)
        if self._lib is None:
            self._lib = _hip.load()
        by_step = {}
        for p in ps:
            by_step.setdefault(int(self.state[p]["step"]), []).append(p)
        for step0, part in sorted(by_step.items()):
            rows = torch.tensor([(p.data_ptr(), p.grad.data_ptr(), self.state[p]["exp_avg"].data_ptr(), self.state[p]["exp_avg_sq"].data_ptr(),
                                  p.numel()) for p in part], dtype=torch.int64)   # host table: the records travel in the kernel arguments
            with torch.cuda.device(dev):
                _hip.check(self._lib.rgrg_adamw_multi_step_f32(rows.data_ptr(), len(part), float(group["lr"]), float(b1), float(b2),
                                                               float(group["eps"]), float(group["weight_decay"]), step0 + 1, float(grad_scale),
                                                               torch.cuda.current_stream(dev).cuda_stream), "rgrg_adamw_multi_step_f32")
            for p in part:
                self.state[p]["step"] = step0 + 1
                torch._C._increment_version(p)
        return True
