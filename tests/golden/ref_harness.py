"""Import the real reference modules (``/root/reference``) in the BUILD
container.  Used only by ``make_golden.py``; never on the GPU box (the
reference does not exist there).

Stubs installed (none of them carries hot-path arithmetic the reference
authors):
  * ``torchinfo``                         - imported, unused on the path
  * ``transformers.generation_beam_search`` - removed in transformers 5.x; the
    stub module carries ``oracle/beam_scorer.py`` (restated 4.19.2 BeamSearchScorer)
    so that the reference's own ``beam_search`` loop can run
  * ``GPT2LMHeadModel.from_pretrained``   - no network: returns a random-init
    gpt2-medium skeleton whose weights are then overwritten by load_state_dict
  * ``torchvision``                       - ``tv_shim`` over ``oracle/tv013.py``
"""
from __future__ import annotations

import os
import sys
import types

REFERENCE_ROOT = os.environ.get("RGRG_REFERENCE_ROOT", "/root/reference")


def install_stubs():
    here = os.path.dirname(os.path.abspath(__file__))
    repo = os.path.dirname(os.path.dirname(here))
    for p in (repo, here, REFERENCE_ROOT):
        if p not in sys.path:
            sys.path.insert(0, p)
    ti = types.ModuleType("torchinfo")
    ti.summary = lambda *a, **k: None
    sys.modules.setdefault("torchinfo", ti)
    import transformers  # noqa: F401
    gb = types.ModuleType("transformers.generation_beam_search")
    from oracle.beam_scorer import BeamSearchScorer  # restated 4.19.2 class: the installed 5.x no longer ships it
    gb.BeamSearchScorer = BeamSearchScorer
    sys.modules.setdefault("transformers.generation_beam_search", gb)
    from transformers import GPT2Config, GPT2LMHeadModel

    def _from_pretrained(*a, **k):
        cfg = GPT2Config(vocab_size=50257, n_positions=1024, n_embd=1024, n_layer=24, n_head=16,
                         activation_function="gelu_new", layer_norm_epsilon=1e-5)
        return GPT2LMHeadModel(cfg)

    GPT2LMHeadModel.from_pretrained = staticmethod(_from_pretrained)
    import tv_shim
    tv_shim.install()


def reference_model():
    """Real ``ReportGenerationModel`` (reference code), eval mode, random init."""
    install_stubs()
    from src.full_model.report_generation_model import ReportGenerationModel
    m = ReportGenerationModel(pretrain_without_lm_model=True)
    m.eval()
    return m
