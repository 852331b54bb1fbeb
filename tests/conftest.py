import os
import sys

import pytest
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

GOLDEN = os.path.join(REPO, "tests", "golden")

# The CPU oracle is small-GEMV work: on the GPU box's 128 cores torch's default (one thread per core) makes it ~6x SLOWER
# than 8 threads (bench.py cpu_baseline: 0.010 vs 0.066 images/s), and the golden fixtures were generated with 8 threads (the
# build container) - pin that at import (the child processes of some GPU tests import this module too), for speed and for
# an identical summation order in the oracle's GEMMs.
torch.set_num_threads(min(8, os.cpu_count() or 8))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    return torch.load(os.path.join(GOLDEN, name), weights_only=True)


_SD_CACHE = {}


def synth_sd(profile):
    """Seeded synthetic state dict (CPU fp32), cached per session."""
    if profile not in _SD_CACHE:
        from rgrg_amd import synth
        _SD_CACHE[profile] = synth.make_state_dict(0, profile)
    return _SD_CACHE[profile]


@pytest.fixture(scope="session")
def sd_bench():
    return synth_sd("bench")


@pytest.fixture(scope="session")
def sd_ragged():
    return synth_sd("ragged")


_MODEL_CACHE = {}


def gpu_model(profile):
    """ReportGenerationModel on cuda:0 with the synthetic weights loaded (HIP engine behind it)."""
    if profile not in _MODEL_CACHE:
        import rgrg_amd
        _MODEL_CACHE.clear()  # one resident model at a time
        m = rgrg_amd.ReportGenerationModel(pretrain_without_lm_model=True)
        m.load_state_dict(synth_sd(profile))
        m.to(torch.device("cuda", 0))
        m.eval()
        _MODEL_CACHE[profile] = m
    return _MODEL_CACHE[profile]
