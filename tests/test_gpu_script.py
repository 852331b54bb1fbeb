"""SURVEY 8(a) row a17 on the hardware: the drop-in script (rgrg_amd/generate_reports_for_images.py, counterpart of
src/full_model/generate_reports_for_images.py:107-167) driven the way its main() drives it - image FILE -> get_image_tensor
(decode on the host, INTER_AREA resize + pad + normalise on the GPU) -> get_report_for_image (model.generate with 4 beams,
early stopping, under the reference's autocast wrapper) -> token ids -> text -> de-duplicated report -> report file - against
the CPU oracle on the oracle's own preprocessing of the same pixels."""
import json

import numpy as np
import pytest
import torch

from conftest import gpu_model, synth_sd
from oracle import full_model as o_full
from oracle import preprocess as o_pre
from rgrg_amd import generate_reports_for_images as script
from rgrg_amd.bpe import GPT2ByteDecoder, bytes_to_unicode

pytestmark = pytest.mark.gpu


def _write_png(path, h, w, seed):
    """A synthetic 8-bit gray radiograph-like image: smooth background + rectangles + noise."""
    from PIL import Image
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:h, 0:w]
    img = 90 + 60 * np.sin(xx / w * 3.1) * np.cos(yy / h * 2.3) + rng.normal(0, 12, (h, w))
    for _ in range(12):
        y0, x0 = int(rng.integers(0, h - 40)), int(rng.integers(0, w - 40))
        img[y0:y0 + int(rng.integers(20, h // 3)), x0:x0 + int(rng.integers(20, w // 3))] += float(rng.integers(-50, 90))
    arr = np.clip(img, 0, 255).astype(np.uint8)
    Image.fromarray(arr, mode="L").save(path)
    return arr


def _synthetic_vocab():
    """A GPT-2-shaped vocabulary (50257 entries over the byte-level alphabet) in which every id decodes: token i = " w<i>",
    id 13 = "." (sentence ends for the de-duplication), <|endoftext|> last."""
    space = bytes_to_unicode()[ord(" ")]
    vocab = {("." if i == 13 else f"{space}w{i}"): i for i in range(50256)}
    vocab["<|endoftext|>"] = 50256
    return vocab


def test_script_path_image_file_to_report_matches_the_oracle(tmp_path, monkeypatch):
    model = gpu_model("ragged")
    sd = synth_sd("ragged")
    png = str(tmp_path / "cxr.png")
    arr = _write_png(png, 620, 500, seed=5)                           # portrait, not a multiple of anything: general INTER_AREA table path
    tensor = script.get_image_tensor(png)
    ref_tensor = torch.from_numpy(o_pre.get_image_tensor_from_array(arr))
    assert tensor.shape == (1, 1, 512, 512) and tensor.is_cuda and torch.equal(tensor.cpu(), ref_tensor)   # bit-identical preprocessing
    monkeypatch.setattr(script, "MAX_NUM_TOKENS_GENERATE", 40)        # bounds the CPU oracle's beam search (the reference's 300 only bounds)
    # (1) fp32 (no autocast): ids bit-exact against the oracle's 4-beam search with early stopping
    monkeypatch.setattr(script, "AUTOCAST_DTYPE", None)
    ids = script.get_report_for_image(model, tensor, None, None, None)
    ref = o_full.generate(sd, ref_tensor, 40, num_beams=4, early_stopping=True)
    assert not isinstance(ref, int), "the synthetic image selected no region: pick another seed"
    assert ids.dtype == torch.int64 and ids.shape == ref[0].shape and torch.equal(ids.cpu(), ref[0])
    # (2) ids -> text -> report with a tokenizer object (the byte-level decoder on a synthetic GPT-2-shaped vocabulary)
    tok = GPT2ByteDecoder(_synthetic_vocab())
    report = script.get_report_for_image(model, tensor, tok, None, None)
    sents = tok.batch_decode(ref[0], skip_special_tokens=True, clean_up_tokenization_spaces=True)
    assert isinstance(report, str) and report == script.convert_generated_sentences_to_report(sents, None, None) and len(report) > 0
    # (3) the reference's wrapper: torch.autocast(float16) - reduced-precision detector; same contract, ids of the same form
    monkeypatch.setattr(script, "AUTOCAST_DTYPE", torch.float16)
    ids16 = script.get_report_for_image(model, tensor, None, None, None)
    assert ids16.dtype == torch.int64 and ids16.dim() == 2 and ids16.shape[1] <= 40 and bool((ids16[:, 0] == 50256).all())
    assert abs(ids16.shape[0] - ids.shape[0]) <= 2                    # the selection may flip on a borderline region, not more
    # (4) main(): checkpoint -> model -> per-image loop -> report file (get_model stubbed: a 1.6 GB checkpoint file is not worth writing)
    monkeypatch.setattr(script, "AUTOCAST_DTYPE", None)
    monkeypatch.setattr(script, "get_model", lambda path: model)
    out_txt = tmp_path / "reports.txt"
    script.main("unused.pt", [png, png], str(out_txt), tokenizer=tok)
    text = out_txt.read_text()
    assert text.count("Image path: " + png) == 2 and text.count("Generated report: " + report) == 2
    json.dumps(report)  # plain text


def test_script_returns_empty_report_when_nothing_is_selected(tmp_path, monkeypatch):
    """report_generation_model.py:260-261 -> -1; the script's per-image result is then the empty string."""
    model = gpu_model("ragged")
    png = str(tmp_path / "blank.png")
    _write_png(png, 300, 300, seed=9)
    monkeypatch.setattr(model, "generate", lambda *a, **k: -1)
    assert script.get_report_for_image(model, script.get_image_tensor(png), None, None, None) == ""
