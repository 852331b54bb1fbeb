"""Drop-in counterpart of ``src/full_model/generate_reports_for_images.py`` (ttanida/rgrg).

Same module-level constants and function names (``get_model``, ``get_image_tensor``,
``get_report_for_image``, ``remove_duplicate_generated_sentences``,
``convert_generated_sentences_to_report``, ``write_generated_reports_to_txt``); the model is
``rgrg_amd.ReportGenerationModel`` (HIP path).  Differences, all forced by what the image lacks:
  * cv2 / albumentations / spaCy / evaluate(bertscore) / the GPT-2 tokenizer files are not
    installed and cannot be downloaded, so they are imported lazily and are pluggable: pass
    your own ``tokenizer`` / ``sentence_tokenizer`` / ``bert_score`` objects (same duck types
    as the reference uses); without them the script still produces the token ids.
  * like the reference (generate_reports_for_images.py:108), generation runs under
    ``torch.autocast(device_type="cuda", dtype=AUTOCAST_DTYPE)`` with ``AUTOCAST_DTYPE = torch.float16``: the detector's
    convolutions / fc6 and the many-row decode GEMMs then run on the 16-bit matrix core with fp32 accumulation (one image
    with 4 beams = 116 decode rows stays on the exact-fp32 weight-streaming plan).  Set ``AUTOCAST_DTYPE = None`` for an
    fp32 run.  Beam search (NUM_BEAMS = 4, max_length 300, early_stopping) runs on the HIP decoder like greedy does.
"""
from __future__ import annotations

from collections import defaultdict

import torch

from .constants import BERTSCORE_SIMILARITY_THRESHOLD, IMAGE_INPUT_SIZE, mean, std  # noqa: F401
from .report_generation_model import ReportGenerationModel

device = torch.device("cuda" if torch.cuda.is_available() else "cpu")

MAX_NUM_TOKENS_GENERATE = 300
NUM_BEAMS = 4
AUTOCAST_DTYPE = torch.float16   # the reference's wrapper (:108); None = fp32


def write_generated_reports_to_txt(images_paths, generated_reports, generated_reports_txt_path):
    """One block per image, separated by a 30-character rule (same file format as the reference script)."""
    rule = "=" * 30
    blocks = [f"Image path: {path}\nGenerated report: {report}\n\n{rule}\n\n" for path, report in zip(images_paths, generated_reports)]
    with open(generated_reports_txt_path, "w") as handle:
        handle.write("".join(blocks))


def remove_duplicate_generated_sentences(generated_report, bert_score, sentence_tokenizer):
    """Exact de-duplication always; BERTScore soft de-duplication (threshold 0.9, keep the
    longer sentence) when a ``bert_score`` object is supplied (generate_reports_for_images.py:42-97)."""
    if sentence_tokenizer is not None:
        gen_sents = [sent.text for sent in sentence_tokenizer(generated_report).sents]
    else:
        gen_sents = [s.strip() + "." for s in generated_report.split(".") if s.strip()]
    gen_sents = list(dict.fromkeys(gen_sents))
    to_remove = defaultdict(list)

    def marked(sent):
        return any(sent in lst for lst in to_remove.values())

    if bert_score is not None:
        for i in range(len(gen_sents)):
            s1 = gen_sents[i]
            for j in range(i + 1, len(gen_sents)):
                if marked(s1):
                    break
                s2 = gen_sents[j]
                if marked(s2):
                    continue
                res = bert_score.compute(lang="en", predictions=[s1], references=[s2], model_type="distilbert-base-uncased")
                if res["f1"][0] > BERTSCORE_SIMILARITY_THRESHOLD:
                    if len(s1) > len(s2):
                        to_remove[s1].append(s2)
                    else:
                        to_remove[s2].append(s1)
    return " ".join(s for s in gen_sents if not marked(s))


def convert_generated_sentences_to_report(generated_sents_for_selected_regions, bert_score, sentence_tokenizer):
    return remove_duplicate_generated_sentences(" ".join(generated_sents_for_selected_regions), bert_score, sentence_tokenizer)


def get_report_for_image(model, image_tensor, tokenizer, bert_score, sentence_tokenizer):
    import contextlib
    ctx = torch.autocast(device_type="cuda", dtype=AUTOCAST_DTYPE) if AUTOCAST_DTYPE is not None else contextlib.nullcontext()
    with ctx:
        output = model.generate(image_tensor.to(device, non_blocking=True), max_length=MAX_NUM_TOKENS_GENERATE,
                                num_beams=NUM_BEAMS, early_stopping=True)
    if isinstance(output, int):  # -1: no region both detected and selected
        return ""
    output_ids, _, _, _ = output
    if tokenizer is None:
        return output_ids
    sents = tokenizer.batch_decode(output_ids, skip_special_tokens=True, clean_up_tokenization_spaces=True)
    return convert_generated_sentences_to_report(sents, bert_score, sentence_tokenizer)


def get_image_tensor(image_path):
    """LongestMaxSize(512, INTER_AREA) -> centred zero PadIfNeeded(512,512) -> Normalize(mean, std) -> [1,1,512,512]
    (generate_reports_for_images.py:129-147), computed on the GPU (``rgrg_amd.preprocess``); only the file decoding is host
    code (cv2 when installed, else PIL)."""
    from .preprocess import preprocess_image, read_gray_image
    return preprocess_image(read_gray_image(image_path), device)


def get_model(checkpoint_path):
    checkpoint = torch.load(checkpoint_path, map_location=torch.device("cpu"))
    model = ReportGenerationModel(pretrain_without_lm_model=True)
    model.load_state_dict(checkpoint["model"])  # accepts both the 0.13 and the pre-0.13 RPN head key names
    model.to(device, non_blocking=True)
    model.eval()
    del checkpoint
    return model


def main(checkpoint_path, images_paths, generated_reports_txt_path, tokenizer=None, bert_score=None, sentence_tokenizer=None):
    model = get_model(checkpoint_path)
    generated_reports = []
    for image_path in images_paths:
        image_tensor = get_image_tensor(image_path)
        generated_reports.append(get_report_for_image(model, image_tensor, tokenizer, bert_score, sentence_tokenizer))
    write_generated_reports_to_txt(images_paths, [r if isinstance(r, str) else r.tolist() for r in generated_reports],
                                   generated_reports_txt_path)
