"""The reference's Python surface for the hot path, backed by the HIP engines.

Same constructor arguments, ``forward`` signature, output-dict keys, sub-module names and
error behaviour as reference ``models/model.py`` (``PreferenceGuidedCaptioningModel`` :740-853,
``VisionEncoder.forward`` :166-243, ``TextEncoder.forward`` :402-474, ``CaptionDecoder.forward``
:561-619).  Tensors returned are plain device tensors: gradients are produced by the explicit
backward schedules in ``steps.py`` (``DPOStep`` / ``ContrastiveStep``), not by autograd.

Not implemented (out of the hot path): LoRA adapters.  Caption generation (row 14)
runs on the same kernels without a KV cache: ``CaptionDecoder.generate`` / ``generate_token_ids`` / ``generate_captions``.
``forward`` evaluates the reference's eval-mode arithmetic (no dropout); the training steps in ``steps.py`` apply
``model.dropout`` at the reference's train-mode sites (fused, counter-based, replayed in the backward - DESIGN.md).
"""
from __future__ import annotations

import logging
from typing import Any, Dict, List, Optional

import torch

from . import hip
from .arch import ModelArch, make_arch
from .engine import (BF16, F32, I32, I64, CaptionDecoderEngine, ProjHead, SeqBatch, TextTowerEngine, VisionTower,
                     Workspace, make_seq_batch)
from .params import ParamStore

logger = logging.getLogger(__name__)


def _default_device():
    if not torch.cuda.is_available():
        raise RuntimeError("pgca_amd needs an MI355X (torch.cuda.is_available() is False); there is no CPU fallback")
    return torch.device("cuda", torch.cuda.current_device())


class ModuleView:
    """What the reference exposes as ``nn.Module`` attributes (``vision_model``, ``projection``, ``text_model``,
    ``lm_model``, ``cross_attention`` ... - reference tests/test_model.py:30-35,74-88,112-118,220-226 touch them): a
    read-only view of the parameters under one prefix of the flat store, with ``requires_grad`` telling whether their
    segment trains (the reference's ``_freeze_backbone``, model.py:150-164,354-368)."""

    def __init__(self, store, prefix: str):
        self._store, self._prefix = store, prefix.rstrip(".") + "."

    def named_parameters(self):
        out = []
        for seg in self._store.segments.values():
            for n in seg.index:
                if n.startswith(self._prefix):
                    t = seg.w(n)
                    t.requires_grad_(bool(seg.trainable))
                    out.append((n[len(self._prefix):], t))
        return out

    def parameters(self):
        return [t for _, t in self.named_parameters()]

    def state_dict(self):
        return {n: t.detach() for n, t in self.named_parameters()}


class DecoderOutput(dict):
    """``CaptionDecoder.forward`` result: HF's ``CausalLMOutputWithCrossAttentions`` is read both as ``out.logits``
    (reference model.py:836, tests/test_model.py:241-243) and as ``out["logits"]``."""

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e


class VisionEncoder:
    """CLIP ViT (frozen on request, reference model.py:150-164) + trainable projection head (reference model.py:64-243)."""

    def __init__(self, owner: "PreferenceGuidedCaptioningModel"):
        self._o = owner
        self.projection_dim = owner.projection_dim
        self.feature_dim = owner.arch.vit.hidden
        self.tower = VisionTower(owner.store, owner.arch.vit, owner.ws)
        self.freeze_backbone = not self.tower.trainable
        self.head = ProjHead(owner.store, "vision_encoder.projection", owner.arch.vit.hidden, owner.arch.proj_dim,
                             owner.ws, "vhead")
        self.vision_model = ModuleView(owner.store, "vision_encoder.vision_model")
        self.projection = ModuleView(owner.store, "vision_encoder.projection")
        self._all = ModuleView(owner.store, "vision_encoder")

    def parameters(self):
        return self._all.parameters()

    def named_parameters(self):
        return self._all.named_parameters()

    def forward(self, pixel_values: torch.Tensor, attention_mask: Optional[torch.Tensor] = None) -> Dict[str, torch.Tensor]:
        if pixel_values.dim() != 4:
            raise ValueError(f"Expected pixel_values to be 4D tensor (B, C, H, W), got {pixel_values.dim()}D")
        if pixel_values.size(1) != 3:
            raise ValueError(f"Expected 3 channels (RGB), got {pixel_values.size(1)} channels")
        try:
            px = pixel_values.to(self._o.device, F32).contiguous()
            feats, pooled, pooled_bf = self.tower.forward(px)
        except Exception as e:  # same wrapping as reference model.py:232-234
            raise RuntimeError(f"Vision encoding failed: {e}") from e
        emb = self.head.forward(pooled_bf, px.shape[0], save=False)
        return {"features": feats, "embeddings": emb, "pooled_output": pooled}

    __call__ = forward


class TextEncoder:
    """GPT-2 text tower + projection head (reference model.py:246-474)."""

    def __init__(self, owner: "PreferenceGuidedCaptioningModel"):
        self._o = owner
        self.projection_dim = owner.projection_dim
        self.feature_dim = owner.arch.gpt.hidden
        self.engine = TextTowerEngine(owner.store, owner.arch, owner.ws, "text")
        self.freeze_backbone = not owner.store.segments["text_tower"].trainable
        self.text_model = ModuleView(owner.store, "text_encoder.text_model")
        self.projection = ModuleView(owner.store, "text_encoder.projection")
        self.tokenizer = None     # GPT-2 vocab/merges files are not bundled; attach a tokenizer to decode / encode text
        self._all = ModuleView(owner.store, "text_encoder")

    def parameters(self):
        return self._all.parameters()

    def named_parameters(self):
        return self._all.named_parameters()

    def forward(self, input_ids: torch.Tensor, attention_mask: torch.Tensor,
                return_hidden_states: bool = False) -> Dict[str, torch.Tensor]:
        if input_ids.dim() != 2:
            raise ValueError(f"Expected input_ids to be 2D tensor (B, seq_len), got {input_ids.dim()}D")
        if attention_mask.dim() != 2:
            raise ValueError(f"Expected attention_mask to be 2D tensor (B, seq_len), got {attention_mask.dim()}D")
        if input_ids.shape != attention_mask.shape:
            raise ValueError(
                f"input_ids shape {input_ids.shape} doesn't match attention_mask shape {attention_mask.shape}")
        try:
            dev = self._o.device
            ids = input_ids.to(dev, I64).contiguous()
            mask = (attention_mask != 0).to(I32).to(dev).contiguous()
            if return_hidden_states and self.freeze_backbone:
                raise NotImplementedError("hidden_states of a frozen text tower are not kept (its layers run in place)")
            feats, pooled, emb = self.engine.forward(ids, mask, save=bool(return_hidden_states))
        except Exception as e:  # reference model.py:458-460
            raise RuntimeError(f"Text encoding failed: {e}") from e
        result = {"features": feats, "embeddings": emb, "pooled_output": pooled}
        if return_hidden_states:
            # HF GPT2Model(output_hidden_states=True) (modeling_gpt2.py:596-634): the input embeddings, the output of every
            # block but the last, then the last block's output AFTER ln_f (== features)
            B, S = ids.shape
            sv = self.engine.trunk.saved
            hs = [sv[li]["hin"].view(B, S, -1).clone() for li in range(len(self.engine.trunk.layers))]
            result["hidden_states"] = tuple(hs + [feats.clone()])
            self.engine.trunk.saved = self.engine.saved = None
        return result

    __call__ = forward


class CaptionDecoder:
    """GPT-2 caption decoder with the collapsed 1-token cross-attention (reference model.py:477-619)."""

    def __init__(self, owner: "PreferenceGuidedCaptioningModel"):
        self._o = owner
        self.hidden_size = owner.arch.gpt.hidden
        self.vocab_size = owner.arch.dec_vocab
        self.vision_feature_dim = owner.projection_dim
        self.engine = CaptionDecoderEngine(owner.store, owner.arch, owner.ws, "pol")
        self.lm_model = ModuleView(owner.store, "caption_decoder.lm_model")
        self.vision_projection = ModuleView(owner.store, "caption_decoder.vision_projection")
        self.cross_attention = ModuleView(owner.store, "caption_decoder.cross_attention")
        self.attention_norm = ModuleView(owner.store, "caption_decoder.attention_norm")
        self.tokenizer = None
        self._all = ModuleView(owner.store, "caption_decoder")

    def parameters(self):
        return self._all.parameters()

    def named_parameters(self):
        return self._all.named_parameters()

    def forward(self, vision_features: torch.Tensor, input_ids: Optional[torch.Tensor] = None,
                attention_mask: Optional[torch.Tensor] = None, labels: Optional[torch.Tensor] = None,
                use_cache: bool = False, return_dict: bool = True) -> Dict[str, torch.Tensor]:
        dev = self._o.device
        if input_ids is None:
            # generation mode (reference model.py:611-617): the LM runs on the projected vision vector alone -> [B, 1, V]
            vf = vision_features.to(dev, F32).contiguous()
            pv = self.engine.prefix_embedding(vf)
            empty = torch.zeros(vf.shape[0], 0, dtype=I64, device=dev)
            return DecoderOutput(logits=self.engine.next_token_logits(pv, empty)[:, None, :].contiguous(), loss=None)
        if attention_mask is None:
            attention_mask = torch.ones_like(input_ids)
        sb = make_seq_batch(input_ids, attention_mask, dev, pack=False)
        logits = self.engine.logits(vision_features.to(dev, F32).contiguous(), sb)
        loss = None
        if labels is not None:  # HF ForCausalLMLoss: mean CE over all shifted positions (modeling_gpt2.py:700-716)
            B, S, V = logits.shape
            full = make_seq_batch(labels, torch.ones_like(labels), dev, pack=False)
            tok = torch.empty(full.n_rows, dtype=F32, device=dev)
            hip.logits_logprob(logits, V, V, full.row_map, full.targets, full.n_rows, tok)
            loss = -tok.mean()
        return DecoderOutput(logits=logits, loss=loss)

    __call__ = forward

    # -- generation --------------------------------------------------------------------------------
    EOS_CHECK = 8    # tokens between two "has every sequence finished" host read-backs

    @staticmethod
    def _process_scores(scores: torch.Tensor, ids: torch.Tensor, repetition_penalty: float, warp: bool,
                        temperature: float, top_p: float) -> torch.Tensor:
        """HF's logits processors in HF's order (generation/logits_process.py): RepetitionPenaltyLogitsProcessor, then -
        when sampling - TemperatureLogitsWarper and TopPLogitsWarper (keeps the smallest set with mass >= top_p, at
        least one token).  ``scores`` are raw logits for greedy / sampling and log-probabilities for beam search."""
        if repetition_penalty != 1.0 and ids.shape[1]:
            seen = torch.gather(scores, 1, ids)
            seen = torch.where(seen < 0, seen * repetition_penalty, seen / repetition_penalty)
            scores = scores.scatter(1, ids, seen)
        if warp:
            if temperature != 1.0:
                scores = scores / float(temperature)
            if top_p < 1.0:
                srt, idx = torch.sort(scores, dim=-1, descending=False)
                cum = torch.softmax(srt, dim=-1).cumsum(dim=-1)
                rm = cum <= (1.0 - float(top_p))
                rm[:, -1] = False
                scores = scores.masked_fill(rm.scatter(1, idx, rm), float("-inf"))
        return scores

    @torch.no_grad()
    def generate(self, vision_features: torch.Tensor, max_length: int = 50, num_beams: int = 4,
                 temperature: float = 1.0, do_sample: bool = True, top_p: float = 0.9,
                 repetition_penalty: float = 1.1, pad_token_id: Optional[int] = None,
                 eos_token_id: Optional[int] = None, generator: Optional[torch.Generator] = None,
                 use_cache: bool = True, **kwargs) -> torch.Tensor:
        """Reference ``CaptionDecoder.generate`` (model.py:621-678): HF ``generate`` started from the single embedding
        ``vision_projection(vision_features)``.  Same arguments; returns the generated ids ``[B, <= max_length - 1]``
        (int64; HF counts the prefix embedding as one of the ``max_length`` positions, generation/utils.py
        ``_prepare_generated_length``), padded with ``pad_token_id`` after ``eos_token_id``.

        ``num_beams == 1``: greedy / nucleus sampling on the processed logits.  ``num_beams > 1``: HF's beam search
        restated step for step (generation/utils.py ``_beam_search``: log-softmax, processors on the log-probabilities,
        ``2 * num_beams`` candidates, finished-hypothesis pool with ``length_penalty`` 1.0 and the ``early_stopping=False``
        heuristic; ``do_sample`` draws the candidates with ``torch.multinomial`` as HF's beam-sample does).  Both pinned
        against the reference's own ``generate`` on ``tests/golden/generation.npz`` (the sampling modes only in
        distribution: torch's RNG stream differs).  Token ids default to the decoder tokenizer's (model.py:509-511:
        [PAD] = vocab, [EOS] = vocab + 2).  ``use_cache`` (HF's default): every step feeds ONE position through the trunk
        against per-layer K/V buffers (``engine.GptTrunk.decode_step``); ``False`` recomputes the prefix each step (the
        cross-check)."""
        if kwargs:
            raise TypeError(f"unsupported generation arguments: {sorted(kwargs)}")
        eng, dev = self.engine, self._o.device
        base = self._o.arch.gpt.base_vocab
        pad = base if pad_token_id is None else int(pad_token_id)
        eos = base + 2 if eos_token_id is None else int(eos_token_id)
        emb = vision_features.to(dev, F32).contiguous()
        B, nb = emb.shape[0], max(1, int(num_beams))
        L = int(max_length) - 1            # tokens to generate: the prefix embedding occupies one position
        if L < 1:
            raise ValueError(f"max_length={max_length} leaves no room for a generated token (the prefix counts as one)")
        pv = eng.prefix_embedding(emb)
        if nb > 1:
            return self._beam_search(pv, B, nb, L, pad, eos, float(temperature), bool(do_sample), float(top_p),
                                     float(repetition_penalty), generator, use_cache)
        # ---- greedy / sampling (HF _sample): preallocated ids, one EOS read-back every EOS_CHECK tokens
        ids_buf = torch.full((B, L), pad, dtype=I64, device=dev)
        done = torch.zeros(B, dtype=torch.bool, device=dev)
        n = 0
        logits = eng.decode_begin(pv, L) if use_cache else None
        while True:
            ids = ids_buf[:, :n]
            if not use_cache:
                logits = eng.next_token_logits(pv, ids)
            scores = self._process_scores(logits.clone(), ids, repetition_penalty, do_sample, temperature, top_p)
            if do_sample:
                nxt = torch.multinomial(torch.softmax(scores, dim=-1), 1, generator=generator)[:, 0]
            else:
                nxt = scores.argmax(dim=-1)
            nxt = torch.where(done, torch.full_like(nxt, pad), nxt)
            ids_buf[:, n] = nxt
            n += 1
            done = done | (nxt == eos)
            if n == L or ((n % self.EOS_CHECK == 0) and bool(done.all())):   # the only host read-back of the loop
                break
            if use_cache:
                logits = eng.decode_advance(nxt)
        # columns generated after every sequence had finished are all [PAD]: trimming them equals stopping at once
        alive = (ids_buf[:, :n] != pad).any(dim=0)
        last_tok = int(alive.nonzero().max()) + 1 if bool(alive.any()) else 1
        return ids_buf[:, :max(1, min(n, last_tok))]

    def _beam_search(self, pv, B, nb, L, pad, eos, temperature, do_sample, top_p, repetition_penalty, generator,
                     use_cache) -> torch.Tensor:
        """HF ``GenerationMixin._beam_search`` (transformers 5.x, generation/utils.py) for ``length_penalty`` 1.0,
        ``early_stopping`` False, one returned sequence: state tensors and update rules carry HF's names."""
        eng, dev = self.engine, self._o.device
        R, K2 = B * nb, 2 * nb                                   # beams_to_keep = 2 * num_beams (one EOS id)
        pvr = pv.repeat_interleave(nb, dim=0)
        running_sequences = torch.full((B, nb, L), pad, dtype=I64, device=dev)
        sequences = running_sequences.clone()
        running_beam_scores = torch.zeros(B, nb, device=dev)
        running_beam_scores[:, 1:] = -1e9
        beam_scores = torch.full((B, nb), -1e9, device=dev)
        is_sent_finished = torch.zeros(B, nb, dtype=torch.bool, device=dev)
        unsat = torch.ones(B, 1, dtype=torch.bool, device=dev)   # is_early_stop_heuristic_unsatisfied
        top_num_beam_mask = torch.cat([torch.ones(nb, dtype=torch.bool), torch.zeros(nb, dtype=torch.bool)]).to(dev)
        gen_len = torch.zeros(B, nb, dtype=I64, device=dev)      # tokens of each finished hypothesis (HF: beam_indices)
        ar = torch.arange(B, device=dev)[:, None]
        cur = 0
        logits = eng.decode_begin(pvr, L) if use_cache else None
        while True:
            flat = running_sequences.view(R, L)[:, :cur]
            if not use_cache:
                logits = eng.next_token_logits(pvr, flat)
            log_probs = torch.log_softmax(logits.float(), dim=-1)
            log_probs = self._process_scores(log_probs, flat, repetition_penalty, do_sample, temperature, top_p)
            V = log_probs.shape[1]
            acc = (log_probs.view(B, nb, V) + running_beam_scores[:, :, None]).view(B, nb * V)
            if do_sample:                                        # beam-sample: candidates drawn, then ranked by score
                idx = torch.multinomial(torch.softmax(acc, dim=-1), num_samples=K2, generator=generator)
                topk_log_probs = torch.gather(acc, 1, idx)
            else:
                topk_log_probs, idx = torch.topk(acc, k=K2)
            src_beam = idx // V
            topk_running = running_sequences[ar, src_beam]      # [B, 2nb, L]
            topk_ids = idx % V
            topk_running[:, :, cur] = topk_ids
            hits = (topk_ids == eos) | (cur + 1 >= L)            # EosTokenCriteria | MaxLengthCriteria
            # e. the num_beams best unfinished candidates keep running
            run_lp = topk_log_probs + hits.float() * -1.0e9
            nxt_idx = torch.topk(run_lp, k=nb)[1]
            running_sequences = topk_running[ar, nxt_idx]
            running_beam_scores = torch.gather(run_lp, 1, nxt_idx)
            beam_src = torch.gather(src_beam, 1, nxt_idx)        # the beam each running sequence continues
            # f. finished pool: only candidates ranked inside the first num_beams may finish
            just = hits & top_num_beam_mask[None, :]
            fin_lp = topk_log_probs / float(cur + 1)             # length_penalty 1.0
            fin_lp = fin_lp + (~unsat).float() * -1.0e9 + (~just).float() * -1.0e9
            m_seq = torch.cat([sequences, topk_running], dim=1)
            m_sc = torch.cat([beam_scores, fin_lp], dim=1)
            m_fin = torch.cat([is_sent_finished, just], dim=1)
            m_len = torch.cat([gen_len, torch.full((B, K2), cur + 1, dtype=I64, device=dev)], dim=1)
            keep = torch.topk(m_sc, k=nb)[1]
            sequences, beam_scores = m_seq[ar, keep], torch.gather(m_sc, 1, keep)
            is_sent_finished, gen_len = torch.gather(m_fin, 1, keep), torch.gather(m_len, 1, keep)
            cur += 1
            # g. stop? (_check_early_stop_heuristic with early_stopping=False, _beam_search_has_unfinished_sequences)
            best_run = running_beam_scores[:, :1] / float(cur)
            worst_fin = torch.where(is_sent_finished, beam_scores.min(dim=1, keepdim=True)[0],
                                    torch.full_like(beam_scores, -1.0e9))
            unsat = unsat & (best_run > worst_fin).any(dim=-1, keepdim=True)
            # HF reads this flag back every step; here every EOS_CHECK steps: once no batch item can improve, the
            # finished pool is closed to new entries (the -1e9 terms above), so the extra steps change nothing
            if cur >= L or (cur % self.EOS_CHECK == 0 and not bool(unsat.any() & ~hits.all())):
                break
            tok = torch.gather(topk_ids, 1, nxt_idx).view(R)
            if use_cache:
                flat_src = (beam_src + ar * nb).view(R)
                eng.decode_reorder(flat_src)
                logits = eng.decode_advance(tok)
        out_len = max(1, int(gen_len[:, 0].max()))
        return sequences[:, 0, :out_len].contiguous()


class PreferenceGuidedCaptioningModel:
    """Drop-in for reference ``PreferenceGuidedCaptioningModel`` (model.py:681-853) on MI355X."""

    def __init__(self, vision_model: str = "openai/clip-vit-base-patch32", text_model: str = "microsoft/DialoGPT-medium",
                 projection_dim: int = 512, temperature: float = 0.07, dropout: float = 0.1,
                 freeze_vision_backbone: bool = False, freeze_text_backbone: bool = False,
                 lora_config: Optional[Dict[str, Any]] = None, *, device=None, seed: int = 0,
                 arch: Optional[ModelArch] = None) -> None:
        if lora_config:
            raise NotImplementedError("LoRA adapters are disabled in every shipped reference config and not supported")
        self.projection_dim = projection_dim
        self.temperature = temperature
        self.dropout = dropout
        self.device = torch.device(device) if device is not None else _default_device()
        self.arch = arch if arch is not None else make_arch(vision_model, text_model, projection_dim)
        self.projection_dim = self.arch.proj_dim      # an explicit ``arch`` carries its own projection width
        frozen = (["vit"] if freeze_vision_backbone else []) + (["text_tower"] if freeze_text_backbone else [])
        self.store = ParamStore(self.arch, self.device, seed=seed, frozen=frozen)
        for seg in self.store.segments.values():
            seg.ensure_bf16()
            if seg.trainable:
                seg.ensure_train_state()
        hip.load()
        self.ws = Workspace(self.device)
        self.vision_encoder = VisionEncoder(self)
        self.text_encoder = TextEncoder(self)
        self.caption_decoder = CaptionDecoder(self)
        self.training = True
        self.logger = logger

    # -- nn.Module-like conveniences ---------------------------------------------------------------
    def train(self, mode: bool = True):
        self.training = mode
        return self

    def eval(self):
        return self.train(False)

    def to(self, *a, **k):
        return self

    def parameters(self):
        return [seg.w(n) for seg in self.store.segments.values() for n in seg.index]

    def named_parameters(self):
        return [(n, seg.w(n)) for seg in self.store.segments.values() for n in seg.index]

    def state_dict(self):
        return self.store.state_dict(aliases=True)

    def load_state_dict(self, sd, strict: bool = True):
        """Strict like ``nn.Module.load_state_dict``: every parameter of this model must be present with its shape
        (the reference's duplicate registrations - ``clip_model.*``, ``lm_head.weight`` - and its unused CLIP text
        tower are accepted and ignored)."""
        return self.store.load_state_dict(sd, strict=strict)

    def sync_bf16(self) -> None:
        """Refresh the bf16 mirrors after the f32 masters were edited outside the optimiser."""
        for seg in self.store.segments.values():
            seg.ensure_bf16()

    # -- forward -----------------------------------------------------------------------------------
    def forward(self, images: torch.Tensor, caption_ids: Optional[torch.Tensor] = None,
                caption_mask: Optional[torch.Tensor] = None, labels: Optional[torch.Tensor] = None,
                mode: str = "contrastive") -> Dict[str, torch.Tensor]:
        outputs: Dict[str, torch.Tensor] = {}
        vision_outputs = self.vision_encoder(images)
        image_embeddings = vision_outputs["embeddings"]
        B, P = image_embeddings.shape
        if mode in ("contrastive", "dual") and caption_ids is not None:
            text_outputs = self.text_encoder(caption_ids, caption_mask)
            img_n = torch.empty(B, P, dtype=F32, device=self.device)
            txt_n = torch.empty(B, P, dtype=F32, device=self.device)
            hip.l2norm_fwd(image_embeddings, B, P, img_n)
            hip.l2norm_fwd(text_outputs["embeddings"], B, P, txt_n)
            outputs.update({"image_embeddings": img_n, "text_embeddings": txt_n,
                            "vision_features": vision_outputs["features"], "text_features": text_outputs["features"]})
        if mode in ("generation", "dual"):
            dec = self.caption_decoder(vision_features=image_embeddings, input_ids=caption_ids,
                                       attention_mask=caption_mask, labels=labels, return_dict=True)
            outputs.update({"logits": dec["logits"],
                            "generation_loss": dec["loss"] if dec["loss"] is not None
                            else torch.tensor(0.0, device=self.device)})
        return outputs

    __call__ = forward

    def sequence_logprobs(self, images: torch.Tensor, caption_ids: torch.Tensor, caption_mask: torch.Tensor,
                          reduce: str = "mean") -> torch.Tensor:
        """Fast entry: per-sequence log-prob without materialising [B,S,V] logits."""
        emb = self.vision_encoder(images)["embeddings"]
        sb = make_seq_batch(caption_ids, caption_mask, self.device)
        return self.caption_decoder.engine.sequence_logprobs(emb, sb, reduce, save=False).clone()

    def compute_similarity(self, images, captions, caption_mask) -> torch.Tensor:
        out = self(images=images, caption_ids=captions, caption_mask=caption_mask, mode="contrastive")
        from .components import TemperatureScaledSimilarity
        # reference model.py:947-953: no clamp on the model's own temperature
        sim = TemperatureScaledSimilarity(self.temperature, min_temp=0.0, max_temp=float("inf"))
        return sim(out["image_embeddings"], out["text_embeddings"])

    @torch.no_grad()
    def generate_token_ids(self, images: torch.Tensor, **gen_kwargs) -> torch.Tensor:
        """Images -> generated caption token ids (the tensor half of reference ``generate_captions``, model.py:883-901)."""
        was = self.training
        self.eval()
        try:
            emb = self.vision_encoder(images)["embeddings"]
            return self.caption_decoder.generate(vision_features=emb, **gen_kwargs)
        finally:
            self.train(was)

    def generate_captions(self, images: torch.Tensor, max_length: int = 50, num_beams: int = 4, temperature: float = 1.0,
                          do_sample: bool = True, top_p: float = 0.9, **kwargs) -> List[str]:
        """Reference ``generate_captions`` (model.py:855-923).  Decoding to text needs the decoder's GPT-2 tokenizer
        (vocab.json / merges.txt are not available offline): attach one as ``model.caption_decoder.tokenizer``."""
        tok = getattr(self.caption_decoder, "tokenizer", None)
        if tok is None:
            raise RuntimeError("no tokenizer attached (model.caption_decoder.tokenizer); use generate_token_ids() for ids")
        ids = self.generate_token_ids(images, max_length=max_length, num_beams=num_beams, temperature=temperature,
                                      do_sample=do_sample, top_p=top_p, **kwargs)
        return [tok.decode(row.tolist(), skip_special_tokens=True).strip() for row in ids.cpu()]
