"""CPU checks of the drop-in boundary: the C-ABI library builds for gfx950, loads and
exports every symbol include/rgrg_hip.h declares (no compute without a GPU); the host
modules mirror the reference's state-dict keys, signatures and error behaviour; the
product refuses to compute without its HIP path."""
import inspect
import os
import re

import pytest
import torch

import rgrg_amd
from conftest import REPO
from rgrg_amd import _hip, build, synth
from rgrg_amd.report_generation_model import expand_alias_keys


def declared_functions():
    text = open(os.path.join(REPO, "include", "rgrg_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(rgrg_[a-z0-9_]+)\s*\(", text)))


def test_library_builds_loads_and_exports_the_declared_abi():
    path = build.build_library()
    assert os.path.exists(path)
    lib = _hip.load()
    names = declared_functions()
    assert len(names) >= 17
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/rgrg_hip.h but not exported"
    assert sorted(_hip.SIGNATURES) == names  # the ctypes table covers exactly the header
    assert lib.rgrg_abi_version() == _hip.ABI_VERSION


def test_gfx950_code_object_is_embedded():
    blob = open(build.LIB_PATH, "rb").read()
    assert b"gfx950" in blob and b"rgrg_skinny_gemm_f32" in blob


@pytest.fixture(scope="module")
def model():
    m = rgrg_amd.ReportGenerationModel(pretrain_without_lm_model=True)
    m.eval()
    return m


def test_module_tree_and_state_dict_keys_match_the_reference(model):
    for attr in ("object_detector", "binary_classifier_region_selection", "binary_classifier_region_abnormal", "language_model"):
        assert hasattr(model, attr)
    od = model.object_detector
    assert hasattr(od, "backbone") and hasattr(od.roi_heads, "dim_reduction") and hasattr(od, "_transform_inputs_for_rpn_and_roi")
    keys = set(model.state_dict())
    assert len(keys) == 1662  # the real reference's count under the same stubs (tests/golden/make_golden.py loads strict)
    for k in ("object_detector.backbone.0.weight", "object_detector.backbone.7.2.bn3.running_var",
              "object_detector.rpn.head.conv.0.0.weight", "object_detector.roi_heads.box_head.fc6.weight",
              "binary_classifier_region_selection.loss_fn.pos_weight",
              "language_model.gpt_with_lm_head.transformer.h.23.attn.uk.weight", "language_model.gpt.h.0.mlp.c_fc.weight",
              "language_model.gpt2_blocks.5.1.causal_mask", "language_model.gpt2_blocks.5.3.c_proj.bias",
              "language_model.wte.weight", "language_model.lm_head.weight", "language_model.final_layernorm.bias",
              "language_model.feature_space_transformation_nn.2.weight"):
        assert k in keys, k
    assert tuple(model.state_dict()["language_model.gpt.h.0.attn.c_attn.weight"].shape) == (1024, 3072)  # Conv1D [in,out]
    assert tuple(model.state_dict()["object_detector.roi_heads.box_head.fc6.weight"].shape) == (1024, 131072)


def test_generate_signature_matches_reference(model):
    sig = inspect.signature(model.generate)
    assert list(sig.parameters) == ["images", "max_length", "num_beams", "num_beam_groups", "do_sample",
                                    "num_return_sequences", "early_stopping"]
    assert sig.parameters["max_length"].default is None and sig.parameters["num_beams"].default == 1
    lsig = inspect.signature(model.language_model.generate)
    assert list(lsig.parameters)[0] == "image_hidden_states"


def test_load_state_dict_accepts_reference_checkpoints(model, sd_bench):
    assert not model.load_state_dict(sd_bench).missing_keys  # canonical family only
    full = synth.to_reference_state_dict(sd_bench)
    r = model.load_state_dict(full)  # the reference's full aliased key set
    assert not r.missing_keys and not r.unexpected_keys
    legacy = dict(full)  # pre-0.13 RPN head names (generate_reports_for_images.py:156-159)
    legacy["object_detector.rpn.head.conv.weight"] = legacy.pop("object_detector.rpn.head.conv.0.0.weight")
    legacy["object_detector.rpn.head.conv.bias"] = legacy.pop("object_detector.rpn.head.conv.0.0.bias")
    model.load_state_dict(legacy)
    only_blocks = {k: v for k, v in full.items() if not (k.startswith("language_model.gpt.") or k.startswith("language_model.gpt_with_lm_head.transformer.h."))}
    model.load_state_dict(only_blocks)  # gpt2_blocks.* family alone is enough
    bad = dict(sd_bench)
    bad.pop("object_detector.roi_heads.box_head.fc7.bias")
    bad["object_detector.bogus"] = torch.zeros(1)
    with pytest.raises(RuntimeError) as e:
        model.load_state_dict(bad)
    assert "fc7.bias" in str(e.value) and "bogus" in str(e.value)  # reported, not silently dropped
    w = model.language_model.gpt2_blocks[3][1].uk.weight
    assert torch.equal(w, sd_bench["language_model.gpt_with_lm_head.transformer.h.3.attn.uk.weight"])
    assert set(expand_alias_keys(sd_bench)) >= set(k for k in full if not k.endswith(("causal_mask", "mask_out_value")))


def test_generation_mode_errors_match_reference(model):
    lm = model.language_model
    f = torch.zeros(2, 1024)
    with pytest.raises(ValueError, match="num_beam_groups"):
        lm.generate(f, 8, num_beams=1, num_beam_groups=2)
    with pytest.raises(ValueError, match="num_return_sequences has to be 1"):
        lm.generate(f, 8, num_return_sequences=2)
    with pytest.raises(NotImplementedError, match="Multinomial"):
        lm.generate(f, 8, do_sample=True)
    with pytest.raises(ValueError, match="max_length has to be set"):
        lm.generate(f, None, num_beams=4)
    with pytest.raises(ValueError, match="num_return_sequences"):
        lm.generate(f, 8, num_beams=2, num_return_sequences=3)
    with pytest.raises(NotImplementedError):
        lm.generate(f, 8, num_beams=4, num_beam_groups=2)
    with pytest.raises(_hip.RgrgHipError, match="no CPU fallback"):   # any beam count reaches the HIP path (round 6: > 16 beams too)
        lm.generate(f, 8, num_beams=17)


def test_image_targets_are_validated_like_the_reference(model):
    """object_detector.py:133-162: boxes must be [N, 4] tensors with positive width and height (checked before any GPU work)."""
    images = torch.zeros(1, 1, 512, 512)
    with pytest.raises(AssertionError, match="positive height and width"):   # torch._assert in the reference
        model.object_detector(images, [{"boxes": torch.tensor([[10.0, 10.0, 10.0, 50.0]]), "labels": torch.tensor([1])}])
    with pytest.raises(AssertionError, match="shape"):
        model.object_detector(images, [{"boxes": torch.zeros(4), "labels": torch.tensor([1])}])
    ok = torch.tensor([[10.0, 10.0, 40.0, 50.0]])
    with pytest.raises(TypeError, match="float type"):                          # torchvision RoIHeads.check_targets
        model.object_detector(images, [{"boxes": ok.to(torch.int32), "labels": torch.tensor([1])}])
    with pytest.raises(TypeError, match="int64"):
        model.object_detector(images, [{"boxes": ok, "labels": torch.tensor([1], dtype=torch.int32)}])
    with pytest.raises(IndexError, match="out of bounds"):                      # the class logits have 30 columns
        model.object_detector(images, [{"boxes": ok, "labels": torch.tensor([30])}])
    # a negative label passes validation (torchvision / the reference raise nothing: such a box is ignored by the sampler);
    # on this CPU-only model the call then stops at the missing GPU, not at the targets
    with pytest.raises(_hip.RgrgHipError, match="no CPU fallback"):
        model.object_detector(images, [{"boxes": ok, "labels": torch.tensor([-1])}])


def test_no_cpu_fallback(model):
    with pytest.raises(_hip.RgrgHipError, match="no CPU fallback"):
        model.generate(torch.zeros(1, 1, 512, 512), max_length=4)
    with pytest.raises(_hip.RgrgHipError):
        model.language_model.generate(torch.zeros(2, 1024), 4)
    ids, am = torch.zeros((2, 5), dtype=torch.int64), torch.ones((2, 5), dtype=torch.int64)
    with pytest.raises(_hip.RgrgHipError):
        model.language_model(ids, am, torch.zeros(2, 1024), return_loss=True)   # eval forward: HIP only as well
    model.language_model.train()
    with pytest.raises(_hip.RgrgHipError):
        model.language_model(ids, am, torch.zeros(2, 1024), return_loss=True)   # training pass: HIP only too
    model.language_model.eval()


def test_product_never_imports_the_oracle():
    pkg = os.path.join(REPO, "rgrg_amd")
    for root, _, files in os.walk(pkg):
        for fn in files:
            if fn.endswith(".py"):
                src = open(os.path.join(root, fn)).read()
                assert "import oracle" not in src and "from oracle" not in src, fn


def test_training_and_preprocessing_host_modules_fail_loudly_without_a_gpu(model):
    """The training / preprocessing host code has no CPU path either, and mirrors the reference's trainable set."""
    import numpy as np
    from rgrg_amd import optim
    from rgrg_amd.preprocess import preprocess_image
    ps = model.trainable_parameters()
    assert len(ps) == 112 and sum(p.numel() for p in ps) == 52480000 + 2 * 590593  # decoder uk/uv/fst + two classifiers
    assert all(p.requires_grad for p in ps)
    frozen = dict(model.named_parameters())["language_model.gpt_with_lm_head.transformer.h.0.attn.c_attn.weight"]
    assert not frozen.requires_grad                                     # GPT-2 itself is frozen (language_model.py:207-213)
    p = torch.nn.Parameter(torch.zeros(8))
    p.grad = torch.ones(8)
    with pytest.raises(_hip.RgrgHipError, match="no CPU fallback"):
        optim.AdamW([p], lr=1e-3).step()
    with pytest.raises(ValueError):
        optim.AdamW([p], lr=-1.0)
    img = np.zeros((1024, 768), np.uint8)
    with pytest.raises(_hip.RgrgHipError, match="no CPU fallback"):
        preprocess_image(img, "cpu")
    with pytest.raises(ValueError):
        preprocess_image(img.astype(np.float32), "cpu")
    lm = model.language_model
    ids, am = torch.zeros((2, 5), dtype=torch.int64), torch.ones((2, 5), dtype=torch.int64)
    with pytest.raises(_hip.RgrgHipError, match="no CPU fallback"):          # the incremental form runs on the HIP decoder too
        lm(ids, am, torch.zeros(2, 1024), return_loss=False, use_cache=True)
    with pytest.raises(NotImplementedError, match="return_loss=False"):
        lm(ids, am, torch.zeros(2, 1024), return_loss=True, use_cache=True)
    with pytest.raises(_hip.RgrgHipError, match="no CPU fallback"):          # round 6: position_ids are embedded as given - on the GPU
        lm(ids, am, torch.zeros(2, 1024), return_loss=True, position_ids=torch.ones((2, 5), dtype=torch.int64))
    assert lm(ids, am, torch.zeros(2, 1024), return_loss=False) is None   # language_model.py:396-399
    assert lm.dropout_p == 0.1                                            # GPT-2's train-mode dropout, as in the reference
