"""ORACLE - test infrastructure only, never the product path.

CPU restatement (plain PyTorch, fp32) of the reference algorithm on the hot path:
Stage-1 NT-Xent step and Stage-2 DPO step of
A-SHOJAEI/preference-guided-image-captioning-alignment.  Only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
this module, and only as the checker / the timed CPU baseline.  The product
package (``pgca_amd``) never imports it and has no CPU fallback.

Parity pinning: the reference's own tests hold no numeric vectors for this path
(shape/finite checks only), so this restatement is pinned against outputs of the
reference itself, produced in the build container by ``oracle/make_golden.py``
(which imports the reference's ``models/model.py`` + ``models/components.py`` and
HF modules built from local configs) and committed under ``tests/golden/``.
``tests/test_oracle_golden.py`` checks every function below against them.

The transformer arithmetic the reference delegates to ``transformers`` 5.15.0 /
``torch`` is restated from the published algorithms, with the call sites cited.
All functions are functional: weights come in as a ``dict`` keyed by the
reference's ``state_dict`` names.
"""
from __future__ import annotations

import math
from typing import Dict, Optional, Tuple

import torch
import torch.nn.functional as F

SD = Dict[str, torch.Tensor]


# ----------------------------------------------------------------------------- dropout (train mode)
# The reference's six nn.Dropout sites draw from torch's global Philox stream, which no other implementation can
# reproduce bit for bit.  The MI355X path uses a counter-based mask: element `idx` of site `seed` is dropped iff
# lowbias32(idx * 0x9E3779B1 + seed) < p * 2^32.  This restates that function so that train-mode parity (same
# masks on both sides) is testable; with p = 0 everything below is the reference's eval-mode arithmetic.
_M32 = 0xFFFFFFFF
KIND_EMBD, KIND_ATTN, KIND_RESID_ATTN, KIND_RESID_MLP, KIND_VPROJ, KIND_XATTN, KIND_HEAD = range(7)
TOWER_DECODER, TOWER_TEXT, TOWER_VHEAD, TOWER_THEAD = range(4)


def _hash32_int(x: int) -> int:
    x &= _M32
    x ^= x >> 16
    x = (x * 0x7FEB352D) & _M32
    x ^= x >> 15
    x = (x * 0x846CA68B) & _M32
    x ^= x >> 16
    return x


def site_seed(base_seed: int, step: int, tower: int, layer: int, kind: int) -> int:
    return _hash32_int(_hash32_int(base_seed * 0x9E3779B1 + step) ^ ((tower << 24) | ((layer & 0xFFFF) << 8) | kind))


def dropout_multiplier(seed: int, p: float, numel: int) -> torch.Tensor:
    """f32 tensor of 0 / 1/(1-p) for elements 0..numel-1 of the site with this seed.  One hash per PAIR of elements
    (index >> 1): the even element reads the low 16 bits, the odd one the high 16; dropped when below
    (p * 2^32) >> 16 - the function csrc/common.h implements (``Drop::mul``)."""
    idx = torch.arange(numel, dtype=torch.int64)
    x = ((idx >> 1) * 0x9E3779B1 + seed) & _M32
    x = x ^ (x >> 16)
    x = (x * 0x7FEB352D) & _M32
    x = x ^ (x >> 15)
    x = (x * 0x846CA68B) & _M32
    x = x ^ (x >> 16)
    bits = torch.where((idx & 1) == 1, x >> 16, x & 0xFFFF)
    keep = bits >= (min(_M32, int(p * 4294967296.0)) >> 16)
    return keep.to(torch.float32) / (1.0 - p)


class Dropper:
    """mult(tower, layer, kind, shape) -> multiplier tensor (or None when p == 0)."""

    def __init__(self, base_seed: int, p: float, step: int = 0, p_gpt: Optional[float] = None):
        # p: model.dropout at the reference's own sites (model.py:139,341,524,531); p_gpt: HF GPT-2's internal
        # embd / attn / resid dropouts (GPT2Config default 0.1, never overridden by the reference); None = same as p
        self.base_seed, self.p, self.step = base_seed, p, step
        self.p_gpt = p if p_gpt is None else p_gpt

    def mult(self, tower: int, layer: int, kind: int, shape) -> Optional[torch.Tensor]:
        p = self.p_gpt if kind in (KIND_EMBD, KIND_ATTN, KIND_RESID_ATTN, KIND_RESID_MLP) else self.p
        if p <= 0.0:
            return None
        n = 1
        for d in shape:
            n *= int(d)
        return dropout_multiplier(site_seed(self.base_seed, self.step, tower, layer, kind), p, n).view(*shape)


def _apply(x: torch.Tensor, m: Optional[torch.Tensor]) -> torch.Tensor:
    return x if m is None else x * m


# ----------------------------------------------------------------------------- primitives
def layer_norm(x: torch.Tensor, sd: SD, prefix: str, eps: float = 1e-5) -> torch.Tensor:
    return F.layer_norm(x, (x.shape[-1],), sd[prefix + ".weight"], sd[prefix + ".bias"], eps)


def gelu_new(x: torch.Tensor) -> torch.Tensor:
    """HF ``NewGELUActivation`` (transformers/activations.py:59-66), GPT-2's MLP activation."""
    return 0.5 * x * (1.0 + torch.tanh(math.sqrt(2.0 / math.pi) * (x + 0.044715 * torch.pow(x, 3.0))))


def quick_gelu(x: torch.Tensor) -> torch.Tensor:
    """HF ``QuickGELUActivation`` (transformers/activations.py:117-123), CLIP's MLP activation."""
    return x * torch.sigmoid(1.702 * x)


def linear(x: torch.Tensor, sd: SD, prefix: str) -> torch.Tensor:
    """``nn.Linear``: weight ``[out, in]``."""
    return x @ sd[prefix + ".weight"].t() + sd[prefix + ".bias"]


def conv1d(x: torch.Tensor, sd: SD, prefix: str) -> torch.Tensor:
    """HF ``Conv1D`` (GPT-2): weight ``[in, out]`` (modeling_gpt2.py:106-107,203)."""
    return x @ sd[prefix + ".weight"] + sd[prefix + ".bias"]


def projection_head(x: torch.Tensor, sd: SD, prefix: str, drop: Optional[Dropper] = None,
                    tower: int = TOWER_VHEAD) -> torch.Tensor:
    """Linear -> ReLU -> Dropout -> Linear -> LayerNorm.  Reference model.py:136-142 (vision), :338-344 (text)."""
    h = torch.relu(linear(x, sd, prefix + ".0"))
    if drop is not None:
        h = _apply(h, drop.mult(tower, 0, KIND_HEAD, h.shape))
    h = linear(h, sd, prefix + ".3")
    return layer_norm(h, sd, prefix + ".4", 1e-5)


# ----------------------------------------------------------------------------- ViT (frozen image tower)
def vit_forward(sd: SD, prefix: str, pixel_values: torch.Tensor, heads: int, patch: int,
                eps: float = 1e-5) -> Tuple[torch.Tensor, torch.Tensor]:
    """HF ``CLIPVisionTransformer`` forward as called at reference model.py:222-230.

    embeddings (modeling_clip.py:200-218): bias-free Conv2d(k=s=patch) == per-patch
    GEMM; class token prepended; learned positions added.  ``pre_layrnorm``; pre-LN
    encoder layers with ``quick_gelu`` MLP (:353-384); softmax attention, no mask
    (:259-277); ``post_layernorm`` on the class token -> ``pooler_output`` (:643-646).
    Returns ``(last_hidden_state [B,T,H], pooled [B,H])``.
    """
    w = sd[prefix + ".embeddings.patch_embedding.weight"]  # [H,3,p,p]
    hid = w.shape[0]
    b, c, hh, ww = pixel_values.shape
    gh, gw = hh // patch, ww // patch
    # im2col: [B, gh*gw, 3*p*p] with the (c, ky, kx) order of the conv weight
    cols = (pixel_values.reshape(b, c, gh, patch, gw, patch)
            .permute(0, 2, 4, 1, 3, 5).reshape(b, gh * gw, c * patch * patch))
    patches = cols @ w.reshape(hid, -1).t()
    cls = sd[prefix + ".embeddings.class_embedding"].expand(b, 1, hid)
    x = torch.cat([cls, patches], dim=1) + sd[prefix + ".embeddings.position_embedding.weight"][None]
    x = layer_norm(x, sd, prefix + ".pre_layrnorm", eps)
    t = x.shape[1]
    dh = hid // heads
    n_layers = 0
    while f"{prefix}.encoder.layers.{n_layers}.layer_norm1.weight" in sd:
        n_layers += 1
    for i in range(n_layers):
        p = f"{prefix}.encoder.layers.{i}"
        r = x
        y = layer_norm(x, sd, p + ".layer_norm1", eps)
        q = linear(y, sd, p + ".self_attn.q_proj").view(b, t, heads, dh).transpose(1, 2)
        k = linear(y, sd, p + ".self_attn.k_proj").view(b, t, heads, dh).transpose(1, 2)
        v = linear(y, sd, p + ".self_attn.v_proj").view(b, t, heads, dh).transpose(1, 2)
        att = torch.softmax((q @ k.transpose(-1, -2)) * dh ** -0.5, dim=-1)
        y = (att @ v).transpose(1, 2).reshape(b, t, hid)
        x = r + linear(y, sd, p + ".self_attn.out_proj")
        r = x
        y = layer_norm(x, sd, p + ".layer_norm2", eps)
        y = linear(quick_gelu(linear(y, sd, p + ".mlp.fc1")), sd, p + ".mlp.fc2")
        x = r + y
    pooled = layer_norm(x[:, 0, :], sd, prefix + ".post_layernorm", eps)
    return x, pooled


def vision_encoder_forward(sd: SD, pixel_values: torch.Tensor, heads: int, patch: int,
                           drop: Optional[Dropper] = None) -> Dict[str, torch.Tensor]:
    """Reference ``VisionEncoder.forward`` (model.py:166-243)."""
    if pixel_values.dim() != 4:
        raise ValueError(f"Expected pixel_values to be 4D tensor (B, C, H, W), got {pixel_values.dim()}D")
    if pixel_values.size(1) != 3:
        raise ValueError(f"Expected 3 channels (RGB), got {pixel_values.size(1)} channels")
    feats, pooled = vit_forward(sd, "vision_encoder.vision_model", pixel_values, heads, patch)
    emb = projection_head(pooled, sd, "vision_encoder.projection", drop, TOWER_VHEAD)
    return {"features": feats, "embeddings": emb, "pooled_output": pooled}


# ----------------------------------------------------------------------------- GPT-2 trunk
def gpt2_trunk(sd: SD, prefix: str, hidden: torch.Tensor, attention_mask: Optional[torch.Tensor],
               heads: int, eps: float = 1e-5, drop: Optional[Dropper] = None, tower: int = TOWER_DECODER) -> torch.Tensor:
    """HF ``GPT2Model`` from ``inputs_embeds`` onward (modeling_gpt2.py:568-622):
    ``+ wpe(arange(S))`` irrespective of padding (:571-577), additive causal AND
    key-padding mask (:583), pre-LN blocks (:283-310) with ``gelu_new`` MLP, ``ln_f``.
    Dropout sites are identity (eval / p=0).  ``hidden`` is ``[B,S,H]`` *before* the
    position add.
    """
    b, s, hid = hidden.shape
    dh = hid // heads
    x = hidden + sd[prefix + ".wpe.weight"][:s][None]
    if drop is not None:  # self.drop (modeling_gpt2.py:604)
        x = _apply(x, drop.mult(tower, 0, KIND_EMBD, x.shape))
    allowed = torch.tril(torch.ones(s, s, dtype=torch.bool, device=hidden.device))[None, None]
    if attention_mask is not None:
        allowed = allowed & (attention_mask[:, None, None, :] != 0)
    bias = torch.zeros(allowed.shape, dtype=x.dtype, device=x.device).masked_fill(~allowed, torch.finfo(x.dtype).min)
    n_layers = 0
    while f"{prefix}.h.{n_layers}.ln_1.weight" in sd:
        n_layers += 1
    for i in range(n_layers):
        p = f"{prefix}.h.{i}"
        r = x
        y = layer_norm(x, sd, p + ".ln_1", eps)
        qkv = conv1d(y, sd, p + ".attn.c_attn")
        q, k, v = qkv.split(hid, dim=2)
        q = q.view(b, s, heads, dh).transpose(1, 2)
        k = k.view(b, s, heads, dh).transpose(1, 2)
        v = v.view(b, s, heads, dh).transpose(1, 2)
        att = torch.softmax((q @ k.transpose(-1, -2)) * dh ** -0.5 + bias, dim=-1)
        if drop is not None:  # attn_dropout (modeling_gpt2.py:66)
            att = _apply(att, drop.mult(tower, i, KIND_ATTN, att.shape))
        y = (att @ v).transpose(1, 2).reshape(b, s, hid)
        y = conv1d(y, sd, p + ".attn.c_proj")
        if drop is not None:  # resid_dropout (modeling_gpt2.py:224)
            y = _apply(y, drop.mult(tower, i, KIND_RESID_ATTN, y.shape))
        x = r + y
        r = x
        y = layer_norm(x, sd, p + ".ln_2", eps)
        y = conv1d(gelu_new(conv1d(y, sd, p + ".mlp.c_fc")), sd, p + ".mlp.c_proj")
        if drop is not None:  # mlp.dropout (modeling_gpt2.py:242)
            y = _apply(y, drop.mult(tower, i, KIND_RESID_MLP, y.shape))
        x = r + y
    return layer_norm(x, sd, prefix + ".ln_f", eps)


def text_encoder_forward(sd: SD, input_ids: torch.Tensor, attention_mask: torch.Tensor,
                         heads: int, drop: Optional[Dropper] = None) -> Dict[str, torch.Tensor]:
    """Reference ``TextEncoder.forward`` (model.py:402-474): GPT2Model -> masked mean
    pool with ``clamp(min=1)`` divisor (:449-456) -> projection head (:463)."""
    if input_ids.dim() != 2:
        raise ValueError(f"Expected input_ids to be 2D tensor (B, seq_len), got {input_ids.dim()}D")
    if attention_mask.dim() != 2:
        raise ValueError(f"Expected attention_mask to be 2D tensor (B, seq_len), got {attention_mask.dim()}D")
    if input_ids.shape != attention_mask.shape:
        raise ValueError(
            f"input_ids shape {input_ids.shape} doesn't match attention_mask shape {attention_mask.shape}")
    p = "text_encoder.text_model"
    feats = gpt2_trunk(sd, p, sd[p + ".wte.weight"][input_ids], attention_mask, heads, drop=drop, tower=TOWER_TEXT)
    m = attention_mask.unsqueeze(-1).to(feats.dtype)
    pooled = (feats * m).sum(dim=1) / torch.clamp(attention_mask.sum(dim=1, keepdim=True), min=1)
    emb = projection_head(pooled.float(), sd, "text_encoder.projection", drop, TOWER_THEAD)
    return {"features": feats.float(), "embeddings": emb, "pooled_output": pooled.float()}


def cross_attention_one_key(sd: SD, prefix: str, query: torch.Tensor, kv: torch.Tensor,
                            heads: int, drop: Optional[Dropper] = None) -> torch.Tensor:
    """``nn.MultiheadAttention(H, heads, batch_first=True)`` with key/value length 1
    (reference model.py:528-533,594-598), written out in full: per-head softmax over
    a single key.  ``query [B,S,H]``, ``kv [B,1,H]``.
    """
    h = query.shape[-1]
    w, bias = sd[prefix + ".in_proj_weight"], sd[prefix + ".in_proj_bias"]
    q = query @ w[:h].t() + bias[:h]
    k = kv @ w[h:2 * h].t() + bias[h:2 * h]
    v = kv @ w[2 * h:].t() + bias[2 * h:]
    b, s, _ = q.shape
    dh = h // heads
    q = q.view(b, s, heads, dh).transpose(1, 2)
    k = k.view(b, 1, heads, dh).transpose(1, 2)
    v = v.view(b, 1, heads, dh).transpose(1, 2)
    att = torch.softmax((q @ k.transpose(-1, -2)) / math.sqrt(dh), dim=-1)  # [B,heads,S,1] == 1
    if drop is not None:  # nn.MultiheadAttention(dropout=p) drops attention WEIGHTS: each (b, head, s) is 0 or 1/(1-p)
        att = _apply(att, drop.mult(TOWER_DECODER, 0, KIND_XATTN, att.shape))
    o = (att @ v).transpose(1, 2).reshape(b, s, h)
    return o @ sd[prefix + ".out_proj.weight"].t() + sd[prefix + ".out_proj.bias"]


def caption_decoder_hidden(sd: SD, vision_embeddings: torch.Tensor, input_ids: torch.Tensor,
                           attention_mask: torch.Tensor, heads: int, xattn_heads: int = 8,
                           drop: Optional[Dropper] = None) -> torch.Tensor:
    """Reference ``CaptionDecoder.forward`` up to ``ln_f`` (model.py:583-610):
    ``tanh(Linear(emb))`` -> 1-token cross-attention over ``wte(ids)`` -> residual +
    ``attention_norm`` -> GPT-2 trunk via ``inputs_embeds``."""
    p = "caption_decoder"
    pv = torch.tanh(linear(vision_embeddings.float(), sd, p + ".vision_projection.0"))
    if drop is not None:  # vision_projection Dropout (model.py:524)
        pv = _apply(pv, drop.mult(TOWER_DECODER, 0, KIND_VPROJ, pv.shape))
    pv = pv.unsqueeze(1)
    te = sd[p + ".lm_model.transformer.wte.weight"][input_ids].float()
    att = cross_attention_one_key(sd, p + ".cross_attention", te, pv, xattn_heads, drop)
    te = layer_norm(te + att, sd, p + ".attention_norm", 1e-5)
    return gpt2_trunk(sd, p + ".lm_model.transformer", te, attention_mask, heads, drop=drop, tower=TOWER_DECODER)


def caption_decoder_logits(sd: SD, vision_embeddings: torch.Tensor, input_ids: torch.Tensor,
                           attention_mask: torch.Tensor, heads: int, drop: Optional[Dropper] = None) -> torch.Tensor:
    """... + tied LM head (modeling_gpt2.py:638-644,700): ``logits [B,S,V]``."""
    h = caption_decoder_hidden(sd, vision_embeddings, input_ids, attention_mask, heads, drop=drop)
    return h @ sd["caption_decoder.lm_model.transformer.wte.weight"].t()


def generate_step_logits(sd: SD, vision_embeddings: torch.Tensor, ids: torch.Tensor, heads: int) -> torch.Tensor:
    """Next-token logits of reference ``CaptionDecoder.generate`` (model.py:653-675): HF ``generate`` is started from
    ``inputs_embeds = vision_projection(vision_features)[:, None]`` - no cross-attention, no attention_norm - and then
    feeds ``wte`` of the tokens it produced; ``[B, V]`` for the position after ``ids`` ``[B, t]``."""
    p = "caption_decoder"
    pv = torch.tanh(linear(vision_embeddings.float(), sd, p + ".vision_projection.0")).unsqueeze(1)
    emb = torch.cat([pv, sd[p + ".lm_model.transformer.wte.weight"][ids].float()], dim=1)
    h = gpt2_trunk(sd, p + ".lm_model.transformer", emb, None, heads)
    return h[:, -1] @ sd[p + ".lm_model.transformer.wte.weight"].t()


def process_scores(scores: torch.Tensor, ids: torch.Tensor, repetition_penalty: float = 1.0, warp: bool = False,
                   temperature: float = 1.0, top_p: float = 1.0) -> torch.Tensor:
    """HF's processors in HF's order (transformers generation/logits_process.py: RepetitionPenaltyLogitsProcessor,
    TemperatureLogitsWarper, TopPLogitsWarper with ``min_tokens_to_keep=1``), pinned against the installed classes by
    tests/test_oracle_golden.py."""
    if repetition_penalty != 1.0 and ids.shape[1]:
        seen = torch.gather(scores, 1, ids)
        scores = scores.scatter(1, ids, torch.where(seen < 0, seen * repetition_penalty, seen / repetition_penalty))
    if warp:
        if temperature != 1.0:
            scores = scores / temperature
        if top_p < 1.0:
            srt, idx = torch.sort(scores, descending=False)
            rm = torch.softmax(srt, dim=-1).cumsum(dim=-1) <= (1 - top_p)
            rm[..., -1:] = False
            scores = scores.masked_fill(rm.scatter(1, idx, rm), float("-inf"))
    return scores


def generate_greedy(sd: SD, vision_embeddings: torch.Tensor, max_length: int, heads: int, pad: int, eos: int,
                    repetition_penalty: float = 1.0):
    """``num_beams=1, do_sample=False`` of the reference's ``CaptionDecoder.generate`` (model.py:657-675 -> HF ``generate``
    from ``inputs_embeds``): the prefix embedding counts as one of the ``max_length`` positions, so at most
    ``max_length - 1`` tokens are produced (generation/utils.py ``_prepare_generated_length``).  Returns
    (ids [B, <= max_length - 1], per-step top-2 logit margins) - the margins tell a bf16 comparison where a tie could flip."""
    b = vision_embeddings.shape[0]
    ids = torch.zeros(b, 0, dtype=torch.long)
    done = torch.zeros(b, dtype=torch.bool)
    margins = []
    for _ in range(max_length - 1):
        logits = process_scores(generate_step_logits(sd, vision_embeddings, ids, heads).clone(), ids, repetition_penalty)
        top2 = logits.topk(2, dim=-1).values
        margins.append(top2[:, 0] - top2[:, 1])
        nxt = torch.where(done, torch.full((b,), pad), logits.argmax(dim=-1))
        ids = torch.cat([ids, nxt[:, None]], dim=1)
        done = done | (nxt == eos)
        if bool(done.all()):
            break
    return ids, torch.stack(margins, dim=1)


def generate_beam_search(sd: SD, vision_embeddings: torch.Tensor, max_length: int, num_beams: int, heads: int, pad: int,
                         eos: int, repetition_penalty: float = 1.0):
    """``num_beams > 1, do_sample=False``: HF ``_beam_search`` (transformers 5.x generation/utils.py) with its defaults
    ``length_penalty=1.0``, ``early_stopping=False``: log-softmax first, processors on the log-probabilities,
    ``2 * num_beams`` candidates per step, a finished-hypothesis pool fed only from the first ``num_beams`` ranks, the
    best-possible-running-score stop heuristic.  Returns (ids [B, len], min top-(2 nb) gap per step) - the gap says where
    a bf16 comparison may legitimately reorder candidates."""
    B, nb = vision_embeddings.shape[0], num_beams
    L, K2 = max_length - 1, 2 * num_beams
    emb = vision_embeddings.repeat_interleave(nb, dim=0)
    run = torch.full((B, nb, L), pad, dtype=torch.long)
    seqs = run.clone()
    run_sc = torch.zeros(B, nb)
    run_sc[:, 1:] = -1e9
    fin_sc = torch.full((B, nb), -1e9)
    fin = torch.zeros(B, nb, dtype=torch.bool)
    unsat = torch.ones(B, 1, dtype=torch.bool)
    mask = torch.cat([torch.ones(nb, dtype=torch.bool), torch.zeros(nb, dtype=torch.bool)])
    glen = torch.zeros(B, nb, dtype=torch.long)
    ar = torch.arange(B)[:, None]
    gaps = []
    cur = 0
    while True:
        flat = run.view(B * nb, L)[:, :cur]
        lp = torch.log_softmax(generate_step_logits(sd, emb, flat, heads).float(), dim=-1)
        lp = process_scores(lp, flat, repetition_penalty)
        V = lp.shape[1]
        acc = (lp.view(B, nb, V) + run_sc[:, :, None]).view(B, nb * V)
        top, idx = torch.topk(acc, k=K2 + 1)
        gaps.append((top[:, :-1] - top[:, 1:]).min(dim=1).values)
        top, idx = top[:, :K2], idx[:, :K2]
        src, tok = idx // V, idx % V
        cand = run[ar, src]
        cand[:, :, cur] = tok
        hits = (tok == eos) | (cur + 1 >= L)
        rlp = top + hits.float() * -1e9
        ni = torch.topk(rlp, k=nb)[1]
        run, run_sc = cand[ar, ni], torch.gather(rlp, 1, ni)
        just = hits & mask[None, :]
        flp = top / float(cur + 1) + (~unsat).float() * -1e9 + (~just).float() * -1e9
        m_seq, m_sc = torch.cat([seqs, cand], 1), torch.cat([fin_sc, flp], 1)
        m_fin = torch.cat([fin, just], 1)
        m_len = torch.cat([glen, torch.full((B, K2), cur + 1, dtype=torch.long)], 1)
        keep = torch.topk(m_sc, k=nb)[1]
        seqs, fin_sc, fin, glen = m_seq[ar, keep], torch.gather(m_sc, 1, keep), torch.gather(m_fin, 1, keep), \
            torch.gather(m_len, 1, keep)
        cur += 1
        worst = torch.where(fin, fin_sc.min(dim=1, keepdim=True)[0], torch.full_like(fin_sc, -1e9))
        unsat = unsat & (run_sc[:, :1] / float(cur) > worst).any(dim=-1, keepdim=True)
        if cur >= L or not bool(unsat.any() & ~hits.all()):
            break
    return seqs[:, 0, :max(1, int(glen[:, 0].max()))], torch.stack(gaps, dim=1)


def model_forward(sd: SD, images: torch.Tensor, caption_ids: torch.Tensor, caption_mask: torch.Tensor,
                  mode: str, vit_heads: int, patch: int, gpt_heads: int) -> Dict[str, torch.Tensor]:
    """Reference ``PreferenceGuidedCaptioningModel.forward`` (model.py:794-853)."""
    out: Dict[str, torch.Tensor] = {}
    vis = vision_encoder_forward(sd, images, vit_heads, patch)
    img_emb = vis["embeddings"]
    if mode in ("contrastive", "dual") and caption_ids is not None:
        txt = text_encoder_forward(sd, caption_ids, caption_mask, gpt_heads)
        out["image_embeddings"] = F.normalize(img_emb, p=2, dim=-1)
        out["text_embeddings"] = F.normalize(txt["embeddings"], p=2, dim=-1)
        out["vision_features"] = vis["features"]
        out["text_features"] = txt["features"]
    if mode in ("generation", "dual"):
        out["logits"] = caption_decoder_logits(sd, img_emb, caption_ids, caption_mask, gpt_heads)
    return out


# ----------------------------------------------------------------------------- losses
def nt_xent(image_embeddings: torch.Tensor, text_embeddings: torch.Tensor, temperature: float) -> torch.Tensor:
    """Reference ``ContrastiveLoss.forward`` (model.py:984-1000): inputs already
    L2-normalised; ``S = I T^t / tau``; mean CE over rows and over columns, averaged."""
    sim = image_embeddings @ text_embeddings.t() / temperature
    labels = torch.arange(sim.shape[0], device=sim.device)
    return (F.cross_entropy(sim, labels) + F.cross_entropy(sim.t(), labels)) / 2


def nt_xent_components(vision: torch.Tensor, text: torch.Tensor, temperature: float,
                       min_temp: float = 0.1, max_temp: float = 2.0) -> torch.Tensor:
    """Reference ``components.ContrastiveLoss`` (components.py:61-83,129-145): normalises
    internally and clamps tau to [0.1, 2.0]."""
    t = min(max(temperature, min_temp), max_temp)
    return nt_xent(F.normalize(vision, p=2, dim=-1), F.normalize(text, p=2, dim=-1), t)


def gather_indices(labels: torch.Tensor) -> torch.Tensor:
    """The int64 gather index the reference builds: ``labels[:, 1:]`` (model.py:1070,
    components.py:341) - logits row t-1 scores token t."""
    return labels[..., 1:].contiguous()


def token_logprobs(logits: torch.Tensor, labels: torch.Tensor) -> torch.Tensor:
    """shift -> log_softmax -> gather (model.py:1069-1079 / components.py:340-354): ``[B,S-1]``."""
    lp = F.log_softmax(logits[..., :-1, :], dim=-1)
    return lp.gather(dim=-1, index=gather_indices(labels).unsqueeze(-1)).squeeze(-1)


def sequence_logprob_mean(logits: torch.Tensor, labels: torch.Tensor, mask: torch.Tensor) -> torch.Tensor:
    """Reference ``PreferenceLoss._compute_log_probs`` (model.py:1052-1085): length-mean,
    divisor ``sum(mask[:,1:])`` (no clamp: NaN when a caption has <= 1 real token)."""
    sm = mask[..., 1:]
    return (token_logprobs(logits, labels) * sm).sum(dim=-1) / sm.sum(dim=-1)


def sequence_logprob_sum(logits: torch.Tensor, labels: torch.Tensor,
                         mask: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Reference ``compute_sequence_logprobs`` (components.py:321-362): length-sum."""
    tl = token_logprobs(logits, labels)
    sm = mask[:, 1:] if mask is not None else torch.ones_like(tl)
    return (tl * sm).sum(dim=1)


def preference_loss(pref_logits, rej_logits, pref_labels, rej_labels, pref_mask, rej_mask,
                    beta: float) -> torch.Tensor:
    """Reference ``PreferenceLoss.forward`` (model.py:1038-1050): the trainer's 2-forward,
    reference-free Stage-2 loss (trainer.py:596-603)."""
    lw = sequence_logprob_mean(pref_logits, pref_labels, pref_mask)
    ll = sequence_logprob_mean(rej_logits, rej_labels, rej_mask)
    return -F.logsigmoid(beta * (lw - ll)).mean()


def dpo_loss(policy_chosen, policy_rejected, ref_chosen=None, ref_rejected=None, beta: float = 0.1,
             reference_free: bool = False, label_smoothing: float = 0.0):
    """Reference ``DPOPreferenceLoss.forward`` (components.py:192-249): 4-forward DPO."""
    pol = policy_chosen - policy_rejected
    if reference_free or ref_chosen is None:
        ref = torch.zeros_like(pol)
    else:
        ref = ref_chosen - ref_rejected
    z = beta * (pol - ref)
    if label_smoothing > 0:
        loss = F.binary_cross_entropy_with_logits(z, (1.0 - label_smoothing) * torch.ones_like(z))
    else:
        loss = -F.logsigmoid(z).mean()
    with torch.no_grad():
        metrics = {
            "dpo_loss": float(loss),
            "reward_margin": float((pol - ref).mean()),
            "reward_accuracy": float((pol > ref).float().mean()),
            "policy_chosen_logprob": float(policy_chosen.mean()),
            "policy_rejected_logprob": float(policy_rejected.mean()),
        }
    return loss, metrics


# ----------------------------------------------------------------------------- optimiser step
def cosine_warmup_lr(base_lr: float, step: int, warmup: int, total: int) -> float:
    """``get_cosine_schedule_with_warmup`` multiplier (transformers/optimization.py;
    called at reference trainer.py:285-289), num_cycles = 0.5."""
    if step < warmup:
        return base_lr * step / max(1, warmup)
    prog = (step - warmup) / max(1, total - warmup)
    return base_lr * max(0.0, 0.5 * (1.0 + math.cos(math.pi * 2.0 * 0.5 * prog)))


def clip_coefficient(total_norm: float, max_norm: float) -> float:
    """``torch.nn.utils.clip_grad_norm_`` scaling (called via accelerator at reference
    trainer.py:511-515,619-623): ``min(1, max_norm / (norm + 1e-6))``."""
    return min(1.0, max_norm / (total_norm + 1e-6))


def adamw_step(p: torch.Tensor, g: torch.Tensor, m: torch.Tensor, v: torch.Tensor, step: int, lr: float,
               wd: float = 0.01, b1: float = 0.9, b2: float = 0.999, eps: float = 1e-8) -> None:
    """``torch.optim.AdamW`` single-tensor update as configured at reference
    trainer.py:275-281 (decoupled decay on every parameter, LN/bias included)."""
    p.mul_(1.0 - lr * wd)
    m.mul_(b1).add_(g, alpha=1.0 - b1)
    v.mul_(b2).addcmul_(g, g, value=1.0 - b2)
    bc1 = 1.0 - b1 ** step
    bc2 = 1.0 - b2 ** step
    denom = (v.sqrt() / math.sqrt(bc2)).add_(eps)
    p.addcdiv_(m, denom, value=-lr / bc1)
