"""Engine ownership shared by the host-side modules.

The HIP engine (packed weights + decoder object) is built lazily from the parameters of
the outermost module (``ReportGenerationModel``; or a sub-model used stand-alone) and is
dropped whenever the parameters may have changed (``load_state_dict``, ``.to()``...).
"""
from __future__ import annotations

from typing import Dict

import torch.nn as nn
from torch import Tensor

from .engine import HipEngine


class EngineOwner(nn.Module):
    _engine_prefix = ""  # key prefix this module has inside ReportGenerationModel.state_dict()

    def _adopt(self, child: "EngineOwner") -> None:
        """Make ``child`` use this module's engine (plain attribute: no module cycle)."""
        object.__setattr__(child, "_engine_root", self)

    def _root(self) -> "EngineOwner":
        return self.__dict__.get("_engine_root") or self

    def _full_state_dict(self) -> Dict[str, Tensor]:
        return {self._engine_prefix + k: v for k, v in self.state_dict().items()}

    def engine(self) -> HipEngine:
        root = self._root()
        eng = root.__dict__.get("_engine")
        dev = next(root.parameters()).device
        if eng is None or eng.device != dev:
            if eng is not None:
                eng.close()
            eng = HipEngine(root._full_state_dict(), dev)
            root.__dict__["_engine"] = eng
        return eng

    def invalidate_engine(self) -> None:
        root = self._root()
        eng = root.__dict__.pop("_engine", None)
        if eng is not None:
            eng.close()

    def _apply(self, fn, *a, **k):
        self.invalidate_engine()
        return super()._apply(fn, *a, **k)

    def load_state_dict(self, state_dict, strict: bool = True, **kw):
        self.invalidate_engine()
        return super().load_state_dict(state_dict, strict=strict, **kw)
