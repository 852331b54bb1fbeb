"""Restatement of ``transformers==4.19.2`` ``BeamSearchScorer`` / ``BeamHypotheses``
(``transformers/generation_beam_search.py``).  TEST INFRASTRUCTURE ONLY.

Third-party dependency of the reference (``environment.yml:48``; used at
``src/language_model/language_model.py:8,457-464,570-578,597-605``) that is absent from
``/root/reference`` and was removed from the installed transformers 5.x, so it is restated
from the published 4.19.2 algorithm: PARITY UNPINNED against a real 4.19.2 install.  The
reference's own ``beam_search`` loop IS run on top of this class to produce the golden
fixtures (``tests/golden/make_golden.py``), which pins the loop itself.

Scores are handled exactly like HF does: tensor entries are float32, everything that goes
through ``.item()`` (hypothesis scores, ``worst_score``, the ``is_done`` test) is Python
float (double) arithmetic.
"""
from __future__ import annotations

from typing import List, Optional, Tuple

import torch


class BeamHypotheses:
    def __init__(self, num_beams: int, length_penalty: float, early_stopping: bool):
        self.length_penalty = length_penalty
        self.early_stopping = early_stopping
        self.num_beams = num_beams
        self.beams: List[Tuple[float, torch.Tensor]] = []
        self.worst_score = 1e9

    def __len__(self):
        return len(self.beams)

    def add(self, hyp: torch.Tensor, sum_logprobs: float):
        score = sum_logprobs / (hyp.shape[-1] ** self.length_penalty)
        if len(self) < self.num_beams or score > self.worst_score:
            self.beams.append((score, hyp))
            if len(self) > self.num_beams:
                ranked = sorted([(s, idx) for idx, (s, _) in enumerate(self.beams)])
                del self.beams[ranked[0][1]]
                self.worst_score = ranked[1][0]
            else:
                self.worst_score = min(score, self.worst_score)

    def is_done(self, best_sum_logprobs: float, cur_len: int) -> bool:
        if len(self) < self.num_beams:
            return False
        if self.early_stopping:
            return True
        cur_score = best_sum_logprobs / cur_len ** self.length_penalty
        return self.worst_score >= cur_score


class BeamSearchScorer:
    def __init__(self, batch_size: int, num_beams: int, device=None, length_penalty: float = 1.0,
                 do_early_stopping: bool = False, num_beam_hyps_to_keep: int = 1, num_beam_groups: int = 1, **kwargs):
        self.num_beams = num_beams
        self.device = device
        self.length_penalty = length_penalty
        self.do_early_stopping = do_early_stopping
        self.num_beam_hyps_to_keep = num_beam_hyps_to_keep
        self.num_beam_groups = num_beam_groups
        self.group_size = num_beams // num_beam_groups
        self._beam_hyps = [BeamHypotheses(num_beams, length_penalty, do_early_stopping) for _ in range(batch_size)]
        self._done = torch.tensor([False for _ in range(batch_size)], dtype=torch.bool)
        if not isinstance(num_beams, int) or num_beams <= 1:
            raise ValueError(f"`num_beams` has to be an integer strictly greater than 1, but is {num_beams}.")

    @property
    def is_done(self) -> bool:
        return bool(self._done.all())

    def process(self, input_ids, next_scores, next_tokens, next_indices, pad_token_id: Optional[int] = None,
                eos_token_id: Optional[int] = None, beam_indices=None):
        cur_len = input_ids.shape[-1]
        batch_size = len(self._beam_hyps)
        if not (batch_size == (input_ids.shape[0] // self.group_size)):
            raise ValueError("batch size of input_ids does not match the scorer")
        next_beam_scores = torch.zeros((batch_size, self.group_size), dtype=next_scores.dtype)
        next_beam_tokens = torch.zeros((batch_size, self.group_size), dtype=next_tokens.dtype)
        next_beam_indices = torch.zeros((batch_size, self.group_size), dtype=next_indices.dtype)
        for batch_idx, beam_hyp in enumerate(self._beam_hyps):
            if self._done[batch_idx]:
                next_beam_scores[batch_idx, :] = 0
                next_beam_tokens[batch_idx, :] = pad_token_id
                next_beam_indices[batch_idx, :] = 0
                continue
            beam_idx = 0
            for rank, (tok, score, index) in enumerate(zip(next_tokens[batch_idx], next_scores[batch_idx], next_indices[batch_idx])):
                batch_beam_idx = batch_idx * self.group_size + index
                if (eos_token_id is not None) and (tok.item() == eos_token_id):
                    if rank >= self.group_size:  # an EOS ranked below the top num_beams is dropped
                        continue
                    beam_hyp.add(input_ids[batch_beam_idx].clone(), score.item())
                else:
                    next_beam_scores[batch_idx, beam_idx] = score
                    next_beam_tokens[batch_idx, beam_idx] = tok
                    next_beam_indices[batch_idx, beam_idx] = batch_beam_idx
                    beam_idx += 1
                if beam_idx == self.group_size:
                    break
            if beam_idx < self.group_size:
                raise ValueError(f"At most {self.group_size} tokens in {next_tokens[batch_idx]} can be equal to `eos_token_id`")
            self._done[batch_idx] = self._done[batch_idx] or beam_hyp.is_done(next_scores[batch_idx].max().item(), cur_len)
        return {"next_beam_scores": next_beam_scores.view(-1), "next_beam_tokens": next_beam_tokens.view(-1),
                "next_beam_indices": next_beam_indices.view(-1)}

    def finalize(self, input_ids, final_beam_scores, final_beam_tokens, final_beam_indices, max_length: int,
                 pad_token_id: Optional[int] = None, eos_token_id: Optional[int] = None, beam_indices=None):
        batch_size = len(self._beam_hyps)
        for batch_idx, beam_hyp in enumerate(self._beam_hyps):
            if self._done[batch_idx]:
                continue
            for beam_id in range(self.num_beams):
                batch_beam_idx = batch_idx * self.num_beams + beam_id
                beam_hyp.add(input_ids[batch_beam_idx], final_beam_scores[batch_beam_idx].item())
        sent_lengths = input_ids.new_zeros(batch_size * self.num_beam_hyps_to_keep)
        best = []
        best_scores = torch.zeros(batch_size * self.num_beam_hyps_to_keep, dtype=torch.float32)
        for i, beam_hyp in enumerate(self._beam_hyps):
            sorted_hyps = sorted(beam_hyp.beams, key=lambda x: x[0])
            for j in range(self.num_beam_hyps_to_keep):
                best_score, best_hyp = sorted_hyps.pop()
                sent_lengths[self.num_beam_hyps_to_keep * i + j] = len(best_hyp)
                best.append(best_hyp)
                best_scores[i * self.num_beam_hyps_to_keep + j] = best_score
        sent_max_len = min(int(sent_lengths.max().item()) + 1, max_length)
        decoded = input_ids.new_zeros((batch_size * self.num_beam_hyps_to_keep, sent_max_len))
        if sent_lengths.min().item() != sent_lengths.max().item():
            decoded.fill_(pad_token_id)
        for i, hypo in enumerate(best):
            decoded[i, : sent_lengths[i]] = hypo
            if sent_lengths[i] < max_length:
                decoded[i, sent_lengths[i]] = eos_token_id
        return {"sequences": decoded, "sequence_scores": best_scores}
