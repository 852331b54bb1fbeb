"""Host-side mirror of ``src/language_model/language_model.py`` (ttanida/rgrg).

``LanguageModel`` keeps the reference's module tree - and therefore its state-dict keys,
including the three aliased copies of every GPT-2 tensor (``gpt_with_lm_head.transformer.*``,
``gpt.*``, ``gpt2_blocks.N.{0,1,2,3}.*``) and the pseudo-attention buffers - but holds
parameters only.  ``generate`` (language_model.py:401-479) keeps the reference's mode
checks and exceptions; greedy and beam search run on the HIP decoder.
"""
from __future__ import annotations

from typing import Optional

import torch
import torch.nn as nn

from . import _hip
from ._owner import EngineOwner
from .constants import BOS_TOKEN_ID, EOS_TOKEN_ID, PAD_TOKEN_ID

N_LAYER, D_MODEL, VOCAB, N_POS = 24, 1024, 50257, 1024


class Conv1DWithTrainedWeights(nn.Module):
    """HF Conv1D layout: weight [in,out] (language_model.py:11-29)."""

    def __init__(self, nin: int, nout: int):
        super().__init__()
        self.weight = nn.Parameter(torch.zeros(nin, nout), requires_grad=False)
        self.bias = nn.Parameter(torch.zeros(nout), requires_grad=False)


class GPT2PseudoAttention(nn.Module):
    def __init__(self):
        super().__init__()
        self.c_attn = Conv1DWithTrainedWeights(D_MODEL, 3 * D_MODEL)
        self.c_proj = Conv1DWithTrainedWeights(D_MODEL, D_MODEL)
        self.embed_dim, self.num_heads, self.head_dim, self.split_size = D_MODEL, 16, 64, D_MODEL
        # buffers of the reference state dict (language_model.py:60-70); the decode kernels need neither:
        # a single-query causal row and the all-ones padding mask are no-ops in greedy generation
        self.register_buffer("causal_mask", torch.tril(torch.ones((N_POS, N_POS), dtype=torch.uint8)).view(1, 1, N_POS, N_POS))
        self.register_buffer("mask_out_value", torch.tensor(-1e4))
        self.uk = nn.Linear(D_MODEL, D_MODEL)
        self.uv = nn.Linear(D_MODEL, D_MODEL)


class GPT2MLP(nn.Module):
    def __init__(self):
        super().__init__()
        self.c_fc = Conv1DWithTrainedWeights(D_MODEL, 4 * D_MODEL)
        self.c_proj = Conv1DWithTrainedWeights(4 * D_MODEL, D_MODEL)


class GPT2Block(nn.Module):
    def __init__(self):
        super().__init__()
        self.ln_1 = nn.LayerNorm(D_MODEL, eps=1e-5)
        self.attn = GPT2PseudoAttention()
        self.ln_2 = nn.LayerNorm(D_MODEL, eps=1e-5)
        self.mlp = GPT2MLP()


class GPT2Transformer(nn.Module):
    def __init__(self):
        super().__init__()
        self.wte = nn.Embedding(VOCAB, D_MODEL)
        self.wpe = nn.Embedding(N_POS, D_MODEL)
        self.drop = nn.Dropout(0.1)
        self.h = nn.ModuleList(GPT2Block() for _ in range(N_LAYER))
        self.ln_f = nn.LayerNorm(D_MODEL, eps=1e-5)


class GPT2LMHeadSkeleton(nn.Module):
    def __init__(self):
        super().__init__()
        self.transformer = GPT2Transformer()
        self.lm_head = nn.Linear(D_MODEL, VOCAB, bias=False)
        self.lm_head.weight = self.transformer.wte.weight  # tied


class _TeacherForcedLoss(torch.autograd.Function):
    """Bridges the HIP training pass into autograd: forward runs loss + gradients in one call
    (rgrg_decoder_lm_loss_grad); backward hands the gradients of the 100 trainable tensors (uk/uv of every layer,
    feature_space_transformation_nn) to autograd, scaled by the incoming gradient."""

    @staticmethod
    def forward(ctx, lm, input_ids, attention_mask, feats, position_ids, *params):
        low = _hip.autocast_mode()
        lm.dropout_seed += 1  # a fresh counter-based stream per pass
        loss, g = lm.engine().lm_loss_grad(feats, input_ids, attention_mask, bf16=low, dropout_p=float(lm.dropout_p),
                                           dropout_seed=lm.pass_dropout_seed(), position_ids=position_ids)
        D, grads = 1024, []
        for l in range(len(lm.gpt.h)):  # same order as LanguageModel.trainable_parameters()
            grads += [g["ukv_w"][(2 * l) * D:(2 * l + 1) * D], g["ukv_b"][(2 * l) * D:(2 * l + 1) * D],
                      g["ukv_w"][(2 * l + 1) * D:(2 * l + 2) * D], g["ukv_b"][(2 * l + 1) * D:(2 * l + 2) * D]]
        grads += [g["fst0_w"], g["fst0_b"], g["fst2_w"], g["fst2_b"]]
        ctx.grads = grads
        return loss

    @staticmethod
    def backward(ctx, grad_out):
        return (None, None, None, None, None) + tuple(g * grad_out for g in ctx.grads)


class LanguageModel(EngineOwner):
    _engine_prefix = "language_model."

    def __init__(self):
        super().__init__()
        self.checkpoint = "healx/gpt-2-pubmed-medium"  # not downloaded: weights come from load_state_dict
        self.bos_token_id, self.eos_token_id, self.pad_token_id = BOS_TOKEN_ID, EOS_TOKEN_ID, PAD_TOKEN_ID
        self.gpt_with_lm_head = GPT2LMHeadSkeleton()
        for p in self.gpt_with_lm_head.parameters():
            p.requires_grad = False
        # the reference freezes GPT-2 BEFORE it swaps in GPT2PseudoAttention (language_model.py:207-213, :50-57), so
        # uk / uv (and feature_space_transformation_nn below) are the trainable decoder tensors
        for b in self.gpt_with_lm_head.transformer.h:
            for p in (b.attn.uk.weight, b.attn.uk.bias, b.attn.uv.weight, b.attn.uv.bias):
                p.requires_grad = True
        # the same aliases the reference creates (language_model.py:215-227)
        self.gpt = self.gpt_with_lm_head.transformer
        self.lm_head = self.gpt_with_lm_head.lm_head
        self.wte, self.wpe, self.drop = self.gpt.wte, self.gpt.wpe, self.gpt.drop
        self.final_layernorm = self.gpt.ln_f
        self.gpt2_blocks = nn.ModuleList(nn.ModuleList([b.ln_1, b.attn, b.ln_2, b.mlp]) for b in self.gpt.h)
        self.feature_space_transformation_nn = nn.Sequential(nn.Linear(1024, 1024), nn.ReLU(), nn.Linear(1024, 1024))
        # train-mode dropout of GPT-2 (embd / attention / residual / MLP, all 0.1 in the gpt2-medium config the reference
        # loads).  Counter-based (seed, site, element) masks: same distribution as the reference's, not the same draws.
        # Set dropout_p = 0.0 for a deterministic training pass.
        self.dropout_p = 0.1
        self.dropout_seed = 0x5EED0000

    def forward(self, input_ids: torch.LongTensor, attention_mask: torch.FloatTensor, image_hidden_states: torch.FloatTensor,
                return_loss: bool = False, past_key_values=None, position_ids: Optional[torch.LongTensor] = None,
                use_cache: Optional[bool] = False):
        """Teacher-forced pass of language_model.py:258-399 in eval mode (LM part):
        ``return_loss=True`` -> the scalar language-modelling loss (float32 tensor); ``position_ids`` ([S,T] or [1,T]; default
        arange(T)) are embedded as given - through the token table, the reference's quirk (:293-307).  Like the reference, the
        positions of ``input_ids`` whose ``attention_mask`` is 0 are overwritten with -100 IN PLACE (:371-374),
        and ``return_loss=False, use_cache=False`` returns None (:396-399).  The incremental
        ``use_cache=True`` form belongs to the reference's own generate loop; here ``generate()`` owns the cache.
        In ``train()`` mode with gradients enabled the returned loss carries a ``grad_fn`` (HIP backward pass)."""
        if past_key_values is not None or use_cache:
            return self._forward_cached(input_ids, attention_mask, image_hidden_states, return_loss, past_key_values, position_ids)
        if not return_loss:
            return None
        ids2 = input_ids.view(-1, input_ids.shape[-1])
        am2 = attention_mask.view(ids2.shape[0], -1)
        if self.training and torch.is_grad_enabled():
            # training pass: loss with a grad_fn; loss.backward() fills .grad of uk/uv/feature_space_transformation_nn
            # (what the reference trains in the decoder).  GPT-2's four dropout sites are active with self.dropout_p
            # (0.1 as in the reference; 0 = deterministic), bf16 GEMMs under torch.autocast.
            self.sync_trainable_if_stale()
            loss = _TeacherForcedLoss.apply(self, ids2, am2, image_hidden_states, position_ids, *self.trainable_parameters())
        else:
            self.sync_trainable_if_stale()
            low = _hip.autocast_mode()
            _, loss = self.engine().lm_forward(image_hidden_states, ids2, am2, want_logits=False, want_loss=True, bf16=low,
                                               position_ids=position_ids)
        ids2[~am2.to(torch.bool)] = -100  # the reference's in-place label write (labels IS input_ids)
        return loss

    @torch.no_grad()
    def _forward_cached(self, input_ids, attention_mask, image_hidden_states, return_loss, past_key_values, position_ids):
        """The incremental form (language_model.py:258-366, :396-399): ``forward(..., use_cache=True)`` returns
        ``(lm_logits [S,T,50257], presents)``; ``presents`` - 24 (key, value) pairs [S,16,1+tokens,64], the image key / value
        in slot 0 - are VIEWS of the HIP decoder's pre-allocated cache (a torch tensor), and feeding them back as
        ``past_key_values`` continues on that cache (the reference concatenates new tensors every step).  Round 4:
        ``position_ids`` may be anything the embedding table holds ([S,T] or [1,T]; default, as in the reference :293-304,
        arange(past_length, past_length + T) where past_length counts the image key when a past is given), and a FOREIGN
        ``past_key_values`` (clones, tensors of another model instance) is copied into the cache and continued from.  Still
        raised: stale views of this decoder's own cache (generate() or another call chain has rewritten it since), train mode, a
        loss.  Round 6: an ``attention_mask`` with zeros ([S, past + T], padding inside the prompt or the past) adds the
        reference's -1e4 to the scores of the masked keys (:316-334)."""
        if return_loss or self.training:
            raise NotImplementedError("forward(use_cache=True) is the generation form: eval mode, return_loss=False")
        self.sync_trainable_if_stale()
        ids2 = input_ids.view(-1, input_ids.shape[-1])
        S, T = ids2.shape
        eng = self.engine()
        adopt = None
        if past_key_values is None:
            past = 0
        else:
            past = eng.owns_cache(past_key_values)
            if past is None:
                if eng.aliases_cache(past_key_values):
                    raise NotImplementedError("past_key_values must be the presents returned by the previous forward(use_cache=True) of "
                                              "this model, or independent tensors: these are stale views of the decoder's cache (it has "
                                              "been rewritten since)")
                adopt = past_key_values
                past = int(past_key_values[0][0].shape[-2]) - 1   # slot 0 = the image key
        am = None
        if attention_mask is not None and not bool((attention_mask.reshape(S, -1) != 0).all()):
            am = attention_mask   # padding inside the prompt / the past: the reference's additive -1e4 per masked key (:316-334)
        if position_ids is None and past_key_values is not None:
            # the reference's default counts the image key: arange(past_length, past_length + T), past_length = keys in the cache
            position_ids = torch.arange(past + 1, past + 1 + T).view(1, T)
        return eng.forward_cached(image_hidden_states if (past == 0 and adopt is None) else None, ids2, past, position_ids=position_ids,
                                  adopt_past=adopt, attention_mask=am)

    def trainable_parameters(self):
        """uk/uv of every layer, then feature_space_transformation_nn: the language-model tensors the reference
        leaves trainable (language_model.py:207-213 freezes the rest)."""
        ps = []
        for b in self.gpt.h:
            ps += [b.attn.uk.weight, b.attn.uk.bias, b.attn.uv.weight, b.attn.uv.bias]
        f = self.feature_space_transformation_nn
        return ps + [f[0].weight, f[0].bias, f[2].weight, f[2].bias]

    def pass_dropout_seed(self) -> int:
        """Seed of the current training pass: the per-pass counter, offset per data-parallel rank so that replicas
        draw different masks; rank 0 / no process group = the counter itself."""
        rank = 0
        if torch.distributed.is_available() and torch.distributed.is_initialized():
            rank = torch.distributed.get_rank()
        return int(self.dropout_seed) + (rank << 40)

    def sync_trainable_if_stale(self) -> None:
        """Push optimizer-updated parameters into the engine (cheap version check: torch bumps ``_version`` on every
        in-place update)."""
        ver = tuple(p._version for p in self.trainable_parameters())
        root = self._root()
        eng_existed = root.__dict__.get("_engine") is not None
        eng = self.engine()
        if eng_existed and root.__dict__.get("_trainable_version") != ver:
            eng.sync_trainable(root._full_state_dict())
        root.__dict__["_trainable_version"] = ver

    @torch.no_grad()
    def teacher_forced_logits(self, input_ids: torch.LongTensor, attention_mask: torch.FloatTensor,
                              image_hidden_states: torch.FloatTensor, position_ids: Optional[torch.LongTensor] = None) -> torch.FloatTensor:
        """lm_logits [S,T,50257] of the same pass (what the reference returns next to ``presents``, :398-399)."""
        logits, _ = self.engine().lm_forward(image_hidden_states, input_ids, attention_mask, want_logits=True, want_loss=False,
                                             position_ids=position_ids)
        return logits

    @torch.no_grad()
    def generate(self, image_hidden_states: torch.FloatTensor, max_length: Optional[int] = None, num_beams: int = 1,
                 num_beam_groups: int = 1, do_sample: bool = False, num_return_sequences: int = 1,
                 early_stopping: bool = False) -> torch.LongTensor:
        """Same contract as language_model.py:401-479: int64 [S, L'] incl. the leading BOS."""
        # same mode table, exceptions and messages as the reference's generate() (language_model.py:422-479)
        if image_hidden_states.is_cuda:
            self.sync_trainable_if_stale()  # weights changed by an optimizer step since the engine packed them
        single_group = num_beam_groups == 1
        if num_beam_groups > num_beams:
            raise ValueError("'num_beam_groups' has to be smaller or equal to 'num_beams'")
        if num_beams > 1 and not single_group and do_sample is True:
            raise ValueError("Diverse beam search cannot be used in sampling mode. Make sure that 'do_sample' is set to 'False'.")
        if num_beams == 1 and single_group:
            if do_sample is True:
                raise NotImplementedError("Multinomial sampling is not implemented.")
            if num_return_sequences > 1:
                raise ValueError(f"num_return_sequences has to be 1, but is {num_return_sequences} when doing greedy search.")
            # like the reference's scripts, callers may wrap generate() in torch.autocast: a reduced-precision
            # autocast dtype opts the many-sequence decode GEMMs into the bf16 MFMA path
            low = _hip.autocast_mode()
            return self.engine().greedy_decode(image_hidden_states, max_length, bf16=low)
        if num_beams > 1 and single_group:
            if do_sample is True:
                raise NotImplementedError("Beam-search multinomial sampling is not implemented.")
            if num_return_sequences > num_beams:
                raise ValueError("'num_return_sequences' has to be smaller or equal to 'num_beams'.")
            if max_length is None:
                raise ValueError("max_length has to be set for beam generation.")
            # length_penalty = 1.0 as in the reference (language_model.py:461)
            low = _hip.autocast_mode()
            return self.engine().beam_search(image_hidden_states, max_length, num_beams, early_stopping, 1.0, bf16=low,
                                             num_return_sequences=num_return_sequences)
        raise NotImplementedError("Diverse beam-search decoding is not implemented.")
