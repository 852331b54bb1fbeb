"""CPU restatement of the reference's GPT-2 decoder with pseudo self-attention
and its greedy loop.  TEST INFRASTRUCTURE ONLY (see ``oracle/__init__.py``).

Follows ``src/language_model/language_model.py`` of ttanida/rgrg:
``Conv1DWithTrainedWeights`` :11-29, ``GPT2PseudoAttention`` :32-180,
``LanguageModel.forward`` :258-399, ``generate`` :401-447,
``prepare_inputs_for_generation`` :498-520, ``greedy_search`` :609-652.
HF ``GPT2MLP``/``NewGELUActivation``/``LayerNorm(eps=1e-5)`` (transformers
4.19.2, third-party) are restated from their published definitions.
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Tuple

import torch
import torch.nn.functional as F

Tensor = torch.Tensor
SD = Dict[str, Tensor]

N_LAYER, N_HEAD, D_MODEL, HEAD_DIM, VOCAB = 24, 16, 1024, 64, 50257
BOS = EOS = PAD = 50256  # language_model.py:200-202
LN_EPS = 1e-5
MASK_VALUE = -1e4  # language_model.py:70


def gelu_new(x: Tensor) -> Tensor:
    return 0.5 * x * (1.0 + torch.tanh(math.sqrt(2.0 / math.pi) * (x + 0.044715 * torch.pow(x, 3.0))))


def _r16(t: Tensor, mode=True) -> Tensor:
    """Round to the 16-bit type of the reduced-precision mode and back (what a 16-bit store / operand conversion does; round
    to nearest even): ``mode`` True / 1 = bfloat16, 2 = float16 (IEEE half - the dtype of the reference's own autocast,
    generate_reports_for_images.py:108)."""
    return t.to(torch.float16 if mode == 2 else torch.bfloat16).to(torch.float32)


def conv1d(sd: SD, p: str, x: Tensor, bf16: bool = False) -> Tensor:
    """HF Conv1D / Conv1DWithTrainedWeights: weight is [in,out] (:22,27).  ``bf16``: emulation of the reference's
    autocast GEMM (both operands rounded to bf16, fp32 accumulation, fp32 bias) - used to check the opt-in bf16 path."""
    w = sd[p + "weight"]
    if bf16:
        x, w = _r16(x, bf16), _r16(w, bf16)
    return torch.addmm(sd[p + "bias"], x.reshape(-1, x.shape[-1]), w).view(*x.shape[:-1], w.shape[-1])


def _heads(t: Tensor) -> Tensor:
    return t.view(*t.shape[:-1], N_HEAD, HEAD_DIM).permute(0, 2, 1, 3)


def pseudo_attention(sd: SD, p: str, x: Tensor, img: Tensor, add_mask: Tensor,
                     past: Optional[Tuple[Tensor, Tensor]], drop_probs: Optional[Tensor] = None,
                     drop_out: Optional[Tensor] = None, bf16: bool = False) -> Tuple[Tensor, Tuple[Tensor, Tensor]]:
    """GPT2PseudoAttention.forward (:124-180).  x [S,T,1024]; img [S,1024] (already
    through feature_space_transformation_nn); add_mask [S,1,1,1+T_total].  ``bf16``: bf16 GEMM operands and a bf16
    K/V cache (the reference under autocast), scores / softmax / accumulation in fp32."""
    q, k, v = conv1d(sd, p + "c_attn.", x, bf16).split(D_MODEL, dim=2)
    if bf16:
        k, v = _r16(k, bf16), _r16(v, bf16)
    if past is None:
        k_img = F.linear(img[:, None, :], sd[p + "uk.weight"], sd[p + "uk.bias"])
        v_img = F.linear(img[:, None, :], sd[p + "uv.weight"], sd[p + "uv.bias"])
        if bf16:
            k_img, v_img = _r16(k_img, bf16), _r16(v_img, bf16)
        K = _heads(torch.cat((k_img, k), dim=1))
        V = _heads(torch.cat((v_img, v), dim=1))
    else:
        K = torch.cat((past[0], _heads(k)), dim=-2)
        V = torch.cat((past[1], _heads(v)), dim=-2)
    Q = _heads(q)
    w = torch.matmul(Q, K.transpose(-1, -2)) / (HEAD_DIM ** 0.5)
    ql, kl = Q.shape[-2], K.shape[-2]
    # causal mask rows [kl-ql, kl), cols [0, kl): the image column is never masked (:96-99)
    causal = torch.tril(torch.ones((kl, kl), dtype=torch.bool))[kl - ql:kl, :kl]
    w = torch.where(causal, w, torch.tensor(MASK_VALUE, dtype=w.dtype))
    w = F.softmax(w + add_mask, dim=-1)
    if drop_probs is not None:  # attn_dropout (:116) with an explicit mask (0 or 1/(1-p)), train mode only
        w = w * drop_probs
    o = torch.matmul(w, V).permute(0, 2, 1, 3).reshape(x.shape[0], ql, D_MODEL)
    a = conv1d(sd, p + "c_proj.", o, bf16)
    if drop_out is not None:    # resid_dropout (:178)
        a = a * drop_out.view_as(a)
    return a, (K, V)


def lm_forward(sd: SD, input_ids: Tensor, attention_mask: Tensor, image_hidden_states: Tensor,
               past: Optional[List[Tuple[Tensor, Tensor]]], position_ids: Tensor, p: str = "language_model.",
               drop_masks: Optional[Dict[Tuple[int, int], Tensor]] = None, bf16: bool = False, return_hidden: bool = False):
    """LanguageModel.forward(return_loss=False, use_cache=True) (:258-366).  ``bf16``: the decoder blocks and lm_head
    with bf16 GEMM operands / bf16 K/V cache (LayerNorm, residual stream, softmax in fp32) - the arithmetic of the
    build's opt-in bf16 path, for checking it against something other than itself."""
    g = p + "gpt_with_lm_head.transformer."
    f = p + "feature_space_transformation_nn."
    img = F.linear(F.relu(F.linear(image_hidden_states, sd[f + "0.weight"], sd[f + "0.bias"])),
                   sd[f + "2.weight"], sd[f + "2.bias"])  # :284
    wte = sd[g + "wte.weight"]
    x = wte[input_ids] + wte[position_ids]  # quirk: positions are embedded with wte, not wpe (:307)
    dm = drop_masks or {}  # train mode: explicit dropout masks {(layer, site): 0 | 1/(1-p)}; sites as in csrc/common.h
    if (0, 0) in dm:
        x = x * dm[(0, 0)].view_as(x)  # self.drop (:311)
    S = input_ids.shape[0]
    am = torch.cat((torch.ones((S, 1), dtype=torch.int64), attention_mask), dim=-1)[:, None, None, :]
    add_mask = (1.0 - am.to(x.dtype)) * -10000.0  # :325-334
    presents = []
    for l in range(N_LAYER):
        b = f"{g}h.{l}."
        h = F.layer_norm(x, (D_MODEL,), sd[b + "ln_1.weight"], sd[b + "ln_1.bias"], LN_EPS)
        a, present = pseudo_attention(sd, b + "attn.", h, img, add_mask, None if past is None else past[l],
                                      dm.get((l, 1)), dm.get((l, 2)), bf16)
        x = a + x
        h = F.layer_norm(x, (D_MODEL,), sd[b + "ln_2.weight"], sd[b + "ln_2.bias"], LN_EPS)
        h = conv1d(sd, b + "mlp.c_proj.", gelu_new(conv1d(sd, b + "mlp.c_fc.", h, bf16)), bf16)
        if (l, 3) in dm:
            h = h * dm[(l, 3)].view_as(h)  # GPT2MLP dropout
        x = h + x
        presents.append(present)
    x = F.layer_norm(x, (D_MODEL,), sd[g + "ln_f.weight"], sd[g + "ln_f.bias"], LN_EPS)
    if return_hidden:  # the caller applies lm_head itself (row chunks: [S,T,50257] does not fit for many rows)
        return x, presents
    lmw = sd[p + "gpt_with_lm_head.lm_head.weight"]
    logits = F.linear(_r16(x, bf16), _r16(lmw, bf16)) if bf16 else F.linear(x, lmw)  # [S,T,50257]
    return logits, presents


@torch.no_grad()
def teacher_forced_trace(sd: SD, ids: Tensor, image_hidden_states: Tensor, bf16: bool = False, topk: int = 1,
                         rows_per_chunk: int = 8, p: str = "language_model."):
    """ONE teacher-forced pass of the restated forward over token histories ``ids`` [S, L] (leading BOS included) that
    some decoder produced: position t sees ids[:, :t+1] exactly as the incremental greedy / beam loop did (positions =
    arange, attention mask of ones - what prepare_inputs_for_generation :498-520 builds during generation).  Returns,
    for the L-1 predicting positions: the logit of the token that was actually chosen next (``chosen`` [S, L-1]), the
    ``topk`` largest logits and their ids (``top_val`` / ``top_idx`` [S, L-1, topk], ties: lower id first like
    torch.argmax) and the full logits of the LAST predicting position (``last_logits`` [S, 50257]).  The lm_head runs
    over chunks of rows so that the [S, L, 50257] logits never exist at once.  Cheap way to pin EVERY step of a long
    or wide decode (hundreds of sequences, > 100 tokens) against the oracle: O(one forward), not O(L) cached steps."""
    S, L = ids.shape
    T = L - 1
    pos = torch.arange(T, dtype=torch.long)[None, :]
    am = torch.ones((S, T), dtype=torch.int64)
    x, _ = lm_forward(sd, ids[:, :T], am, image_hidden_states, None, pos, p, bf16=bf16, return_hidden=True)
    lmw = sd[p + "gpt_with_lm_head.lm_head.weight"]
    if bf16:
        x, lmw = _r16(x, bf16), _r16(lmw, bf16)
    chosen = torch.empty((S, T))
    top_val = torch.empty((S, T, topk))
    top_idx = torch.empty((S, T, topk), dtype=torch.int64)
    last = torch.empty((S, VOCAB))
    for r0 in range(0, S, rows_per_chunk):
        lg = F.linear(x[r0:r0 + rows_per_chunk], lmw)  # [r, T, V]
        chosen[r0:r0 + rows_per_chunk] = lg.gather(-1, ids[r0:r0 + rows_per_chunk, 1:, None]).squeeze(-1)
        if topk == 1:
            v, i = lg.max(-1, keepdim=True)
        else:
            v, i = torch.topk(lg, topk, dim=-1)
        top_val[r0:r0 + rows_per_chunk], top_idx[r0:r0 + rows_per_chunk] = v, i
        last[r0:r0 + rows_per_chunk] = lg[:, -1]
    return {"chosen": chosen, "top_val": top_val, "top_idx": top_idx, "last_logits": last}


@torch.no_grad()
def lm_teacher_forced(sd: SD, input_ids: Tensor, attention_mask: Tensor, image_hidden_states: Tensor,
                      return_loss: bool = True, p: str = "language_model.", position_ids: Optional[Tensor] = None):
    """LanguageModel.forward(..., past_key_values=None, position_ids, use_cache=False) in eval mode
    (:258-399): positions default to arange(T) (:298-300), given ones ([S,T] or [1,T]) are embedded as they are (:293-307);
    with ``return_loss`` the labels are ``input_ids``
    with attention_mask == 0 positions set to -100, shifted one to the left, CrossEntropyLoss(ignore_index=-100)
    (:368-396).  Returns the scalar loss, or the logits [S,T,50257]."""
    S, T = input_ids.shape
    pos = torch.arange(T, dtype=torch.long)[None, :] if position_ids is None else position_ids.view(-1, T)
    logits, _ = lm_forward(sd, input_ids, attention_mask.to(torch.int64) if attention_mask.dtype == torch.bool else attention_mask,
                           image_hidden_states, None, pos, p)
    if not return_loss:
        return logits
    labels = input_ids.clone()
    labels[~attention_mask.to(torch.bool)] = -100
    return F.cross_entropy(logits[:, :-1, :].reshape(-1, VOCAB), labels[:, 1:].reshape(-1), ignore_index=-100)


def trainable_keys(p: str = "language_model.") -> List[str]:
    """The decoder tensors the reference trains (language_model.py:207-213 freezes GPT-2 before uk/uv and
    feature_space_transformation_nn are created, :50-57,:230-236)."""
    g = p + "gpt_with_lm_head.transformer."
    keys = []
    for l in range(N_LAYER):
        keys += [f"{g}h.{l}.attn.{n}.{wb}" for n in ("uk", "uv") for wb in ("weight", "bias")]
    return keys + [p + f"feature_space_transformation_nn.{i}.{wb}" for i in (0, 2) for wb in ("weight", "bias")]


def lm_loss_and_grads(sd: SD, input_ids: Tensor, attention_mask: Tensor, image_hidden_states: Tensor, p: str = "language_model.",
                      drop_masks: Optional[Dict[Tuple[int, int], Tensor]] = None, position_ids: Optional[Tensor] = None):
    """``loss = LanguageModel.forward(return_loss=True); loss.backward()``: torch autograd through the restated forward.
    Dropout off (modules in eval mode, gradients enabled) unless explicit ``drop_masks`` are given.
    Returns (loss, {key: grad})."""
    sd2 = dict(sd)
    keys = trainable_keys(p)
    for k in keys:
        sd2[k] = sd[k].detach().clone().requires_grad_(True)
    with torch.enable_grad():
        S, T = input_ids.shape
        pos = torch.arange(T, dtype=torch.long)[None, :] if position_ids is None else position_ids.view(-1, T)
        logits, _ = lm_forward(sd2, input_ids, attention_mask, image_hidden_states, None, pos, p, drop_masks)
        labels = input_ids.clone()
        labels[~attention_mask.to(torch.bool)] = -100
        loss = F.cross_entropy(logits[:, :-1, :].reshape(-1, VOCAB), labels[:, 1:].reshape(-1), ignore_index=-100)
        loss.backward()
    return loss.detach(), {k: sd2[k].grad for k in keys}


@torch.no_grad()
def greedy_generate(sd: SD, image_hidden_states: Tensor, max_length: Optional[int], p: str = "language_model.",
                    return_logits: bool = False):
    """LanguageModel.generate(num_beams=1) -> greedy_search (:401-447, :609-652).
    Returns int64 [S, L'] incl. the leading BOS; L' = 1 + number of forward passes."""
    S = image_hidden_states.shape[0]
    ids = torch.full((S, 1), BOS, dtype=torch.int64)
    attn = torch.ones((S, 1), dtype=torch.int64)
    unfinished = torch.ones((S,), dtype=torch.int64)
    past, cur_len, all_logits = None, 1, []
    while True:
        pos = attn.long().cumsum(-1) - 1  # :509-512
        pos.masked_fill_(attn == 0, 1)
        inp = ids if past is None else ids[:, -1:]
        if past is not None:
            pos = pos[:, -1:]
        logits, past = lm_forward(sd, inp, attn, image_hidden_states, past, pos, p)
        nxt_logits = logits[:, -1, :]
        if return_logits:
            all_logits.append(nxt_logits.clone())
        nxt = torch.argmax(nxt_logits, dim=-1)
        nxt = nxt * unfinished + PAD * (1 - unfinished)
        ids = torch.cat([ids, nxt[:, None]], dim=-1)
        attn = torch.cat([attn, attn.new_ones((S, 1))], dim=-1)
        cur_len += 1
        unfinished = unfinished * (nxt != EOS).long()
        if unfinished.max() == 0 or (max_length and cur_len >= max_length):
            break
    if return_logits:
        return ids, torch.stack(all_logits, dim=1)
    return ids


@torch.no_grad()
def beam_generate(sd: SD, image_hidden_states: Tensor, max_length: int, num_beams: int, early_stopping: bool = False,
                  num_return_sequences: int = 1, p: str = "language_model.", return_trace: bool = False):
    """LanguageModel.generate(num_beams>1) -> beam_search (:450-475, :481-496, :529-607) on top of
    the restated HF-4.19.2 BeamSearchScorer (length_penalty 1.0).  Returns int64 [S, L]."""
    from .beam_scorer import BeamSearchScorer
    S = image_hidden_states.shape[0]
    scorer = BeamSearchScorer(batch_size=S, num_beams=num_beams, length_penalty=1.0, do_early_stopping=early_stopping,
                              num_beam_hyps_to_keep=num_return_sequences)
    expand = torch.arange(S).view(-1, 1).repeat(1, num_beams).view(-1)  # _expand_inputs_for_generation (:481-490)
    ids = torch.full((S, 1), BOS, dtype=torch.int64).index_select(0, expand)
    attn = torch.ones((S, 1), dtype=torch.int64).index_select(0, expand)
    beam_scores = torch.zeros((S, num_beams), dtype=torch.float)
    beam_scores[:, 1:] = -1e9
    beam_scores = beam_scores.view(-1)
    past, cur_len, trace = None, 1, []
    while True:
        pos = attn.long().cumsum(-1) - 1
        pos.masked_fill_(attn == 0, 1)
        inp = ids if past is None else ids[:, -1:]
        if past is not None:
            pos = pos[:, -1:]
        # step 0: uk/uv of the S image features are repeat_interleave'd to S*num_beams rows (:145-150)
        logits, past = _lm_forward_beams(sd, inp, attn, image_hidden_states, past, pos, num_beams, p)
        scores = F.log_softmax(logits[:, -1, :], dim=-1) + beam_scores[:, None]
        V = scores.shape[-1]
        scores, tokens = torch.topk(scores.view(S, num_beams * V), 2 * num_beams, dim=1, largest=True, sorted=True)
        indices = torch.div(tokens, V, rounding_mode="floor")
        tokens = tokens % V
        if return_trace:
            trace.append((scores.clone(), tokens.clone(), indices.clone()))
        out = scorer.process(ids, scores, tokens, indices, pad_token_id=PAD, eos_token_id=EOS)
        beam_scores, beam_tok, beam_idx = out["next_beam_scores"], out["next_beam_tokens"], out["next_beam_indices"]
        ids = torch.cat([ids[beam_idx, :], beam_tok.unsqueeze(-1)], dim=-1)
        attn = torch.cat([attn, attn.new_ones((attn.shape[0], 1))], dim=-1)
        past = [(k.index_select(0, beam_idx), v.index_select(0, beam_idx)) for k, v in past]  # _reorder_cache (:492-496)
        cur_len += 1
        if scorer.is_done or (max_length and cur_len >= max_length):
            break
    seq = scorer.finalize(ids, beam_scores, tokens, indices, pad_token_id=PAD, eos_token_id=EOS, max_length=max_length)["sequences"]
    return (seq, trace) if return_trace else seq


def _lm_forward_beams(sd, input_ids, attention_mask, image_hidden_states, past, position_ids, num_beams, p):
    """lm_forward for S*num_beams token rows and S image rows: the image key/value of a row's
    batch item is repeated for its beams (GPT2PseudoAttention.forward :145-150)."""
    if past is not None or num_beams == 1:
        rep = image_hidden_states.repeat_interleave(num_beams, dim=0) if num_beams > 1 else image_hidden_states
        return lm_forward(sd, input_ids, attention_mask, rep, past, position_ids, p)
    # step 0: identical arithmetic on repeated rows (uk/uv are row-wise linear maps)
    return lm_forward(sd, input_ids, attention_mask, image_hidden_states.repeat_interleave(num_beams, dim=0), None, position_ids, p)
