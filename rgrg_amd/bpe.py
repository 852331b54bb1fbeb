"""GPT-2 byte-level BPE *decoding* (ids -> text) for the report text of ``get_report_for_image``
(src/full_model/generate_reports_for_images.py:107-127: ``tokenizer.batch_decode(output_ids, skip_special_tokens=True,
clean_up_tokenization_spaces=True)``).  Only decoding is needed on the generate path.  The
vocabulary file (``vocab.json`` of healx/gpt-2-pubmed-medium = GPT-2's) is not shipped and cannot be downloaded here:
pass its path.  Restated from the published GPT-2 ``encoder.py`` / transformers ``tokenization_gpt2`` (``bytes_to_unicode``,
``convert_tokens_to_string``, ``clean_up_tokenization``); checked in tests against the installed transformers on a
synthetic vocabulary."""
from __future__ import annotations

import json
from typing import Dict, Iterable, List


def bytes_to_unicode() -> Dict[int, str]:
    """The reversible byte <-> printable-unicode table of GPT-2's byte-level BPE."""
    bs = list(range(ord("!"), ord("~") + 1)) + list(range(ord("¡"), ord("¬") + 1)) + list(range(ord("®"), ord("ÿ") + 1))
    cs = bs[:]
    n = 0
    for b in range(256):
        if b not in bs:
            bs.append(b)
            cs.append(256 + n)
            n += 1
    return dict(zip(bs, (chr(c) for c in cs)))


def clean_up_tokenization(s: str) -> str:
    for a, b in ((" .", "."), (" ?", "?"), (" !", "!"), (" ,", ","), (" ' ", "'"), (" n't", "n't"), (" 'm", "'m"), (" 's", "'s"),
                 (" 've", "'ve"), (" 're", "'re")):
        s = s.replace(a, b)
    return s


class GPT2ByteDecoder:
    """Duck type of the tokenizer object the script uses: ``batch_decode`` / ``decode`` only."""

    def __init__(self, vocab, eos_token: str = "<|endoftext|>"):
        if isinstance(vocab, str):
            with open(vocab, encoding="utf-8") as f:
                vocab = json.load(f)
        self.decoder = {int(i): t for t, i in vocab.items()}
        self.byte_decoder = {c: b for b, c in bytes_to_unicode().items()}
        self.special_ids = {i for i, t in self.decoder.items() if t == eos_token}

    def decode(self, ids: Iterable[int], skip_special_tokens: bool = False, clean_up_tokenization_spaces: bool = True) -> str:
        toks = [self.decoder[int(i)] for i in ids if not (skip_special_tokens and int(i) in self.special_ids)]
        text = bytearray(self.byte_decoder[c] for c in "".join(toks)).decode("utf-8", errors="replace")
        return clean_up_tokenization(text) if clean_up_tokenization_spaces else text

    def batch_decode(self, sequences, skip_special_tokens: bool = False, clean_up_tokenization_spaces: bool = True) -> List[str]:
        rows = sequences.tolist() if hasattr(sequences, "tolist") else sequences
        return [self.decode(r, skip_special_tokens, clean_up_tokenization_spaces) for r in rows]
