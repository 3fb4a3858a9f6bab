"""Flat HBM parameter store keyed by the reference's ``state_dict`` names.

Every tower's parameters live in ONE contiguous fp32 buffer per segment (master
weights), mirrored by a bf16 buffer of identical layout that the MFMA GEMMs read,
plus - for trainable segments - flat fp32 gradient and AdamW moment buffers.
Contiguity is what makes the fused clip/AdamW kernels single launches and the
RCCL gradient all-reduce a handful of large contiguous buckets.

Tensor names and shapes follow what the reference registers
(reference ``models/model.py:126-142,311-344,505-535`` and the HF modules they
instantiate), so a reference checkpoint's ``model_state_dict`` maps 1:1:
GPT-2 ``Conv1D`` weights are ``[in, out]``, ``nn.Linear`` weights ``[out, in]``,
``lm_head.weight`` is tied to ``transformer.wte.weight``.
"""
from __future__ import annotations

import math
from collections import OrderedDict
from dataclasses import dataclass
from typing import Dict, Iterable, List, Optional, Tuple

import torch

from .arch import GptArch, ModelArch, VitArch

ALIGN = 64  # elements; 256 B in fp32, 128 B in bf16


def _round_up(n: int, m: int) -> int:
    return (n + m - 1) // m * m


@dataclass
class Spec:
    name: str
    shape: Tuple[int, ...]
    init: str  # "normal:<std>" | "zeros" | "ones" | "uniform:<bound>"
    pad_rows: int = 0  # allocate (and keep zero) this many extra rows after the tensor

    @property
    def numel(self) -> int:
        return int(math.prod(self.shape))

    @property
    def alloc(self) -> int:
        return self.numel + self.pad_rows * int(math.prod(self.shape[1:]))


def _ln(prefix: str, n: int) -> List[Spec]:
    return [Spec(prefix + ".weight", (n,), "ones"), Spec(prefix + ".bias", (n,), "zeros")]


def _linear(prefix: str, out_f: int, in_f: int, std: Optional[float] = None) -> List[Spec]:
    if std is None:
        b = 1.0 / math.sqrt(in_f)
        return [Spec(prefix + ".weight", (out_f, in_f), f"uniform:{b}"),
                Spec(prefix + ".bias", (out_f,), f"uniform:{b}")]
    return [Spec(prefix + ".weight", (out_f, in_f), f"normal:{std}"),
            Spec(prefix + ".bias", (out_f,), "zeros")]


def _conv1d(prefix: str, in_f: int, out_f: int, std: float) -> List[Spec]:
    return [Spec(prefix + ".weight", (in_f, out_f), f"normal:{std}"),
            Spec(prefix + ".bias", (out_f,), "zeros")]


def vit_specs(prefix: str, a: VitArch) -> List[Spec]:
    """HF ``CLIPVisionTransformer`` parameters (modeling_clip.py: embeddings :138-157,
    encoder layer :353-384, vision model :594-650)."""
    s: List[Spec] = []
    e = prefix + ".embeddings"
    s.append(Spec(e + ".class_embedding", (a.hidden,), "normal:0.02"))
    s.append(Spec(e + ".patch_embedding.weight", (a.hidden, 3, a.patch, a.patch), "normal:0.02"))
    s.append(Spec(e + ".position_embedding.weight", (a.tokens, a.hidden), "normal:0.02"))
    s += _ln(prefix + ".pre_layrnorm", a.hidden)
    for i in range(a.layers):
        p = f"{prefix}.encoder.layers.{i}"
        # q,k,v are laid out back to back so one [3H,H] GEMM serves all three
        s += _linear(p + ".self_attn.q_proj", a.hidden, a.hidden, 0.02)[:1]
        s += _linear(p + ".self_attn.k_proj", a.hidden, a.hidden, 0.02)[:1]
        s += _linear(p + ".self_attn.v_proj", a.hidden, a.hidden, 0.02)[:1]
        s += _linear(p + ".self_attn.q_proj", a.hidden, a.hidden, 0.02)[1:]
        s += _linear(p + ".self_attn.k_proj", a.hidden, a.hidden, 0.02)[1:]
        s += _linear(p + ".self_attn.v_proj", a.hidden, a.hidden, 0.02)[1:]
        s += _linear(p + ".self_attn.out_proj", a.hidden, a.hidden, 0.02)
        s += _ln(p + ".layer_norm1", a.hidden)
        s += _linear(p + ".mlp.fc1", a.mlp, a.hidden, 0.02)
        s += _linear(p + ".mlp.fc2", a.hidden, a.mlp, 0.02)
        s += _ln(p + ".layer_norm2", a.hidden)
    s += _ln(prefix + ".post_layernorm", a.hidden)
    return s


def head_specs(prefix: str, in_f: int, proj: int) -> List[Spec]:
    """``nn.Sequential(Linear, ReLU, Dropout, Linear, LayerNorm)`` (reference model.py:136-142)."""
    return (_linear(prefix + ".0", proj, in_f) + _linear(prefix + ".3", proj, proj)
            + _ln(prefix + ".4", proj))


VOCAB_TILE = 128  # the tied LM head contracts over the vocabulary in 128-row GEMM tiles


def gpt2_specs(prefix: str, a: GptArch, vocab: int, pad_vocab: bool = False) -> List[Spec]:
    """HF ``GPT2Model`` parameters (modeling_gpt2.py: attention :84-110, MLP :229-243,
    block :246-262, model :486-500).  With ``pad_vocab`` the embedding table is followed by
    zero rows up to a multiple of 128 so the LM-head dgrad GEMM (K = vocab) reads zeros."""
    pad = (_round_up(vocab, VOCAB_TILE) - vocab) if pad_vocab else 0
    s: List[Spec] = [Spec(prefix + ".wte.weight", (vocab, a.hidden), "normal:0.02", pad_rows=pad),
                     Spec(prefix + ".wpe.weight", (a.n_pos, a.hidden), "normal:0.02")]
    proj_std = 0.02 / math.sqrt(2 * a.layers)
    for i in range(a.layers):
        p = f"{prefix}.h.{i}"
        s += _ln(p + ".ln_1", a.hidden)
        s += _conv1d(p + ".attn.c_attn", a.hidden, 3 * a.hidden, 0.02)
        s += _conv1d(p + ".attn.c_proj", a.hidden, a.hidden, proj_std)
        s += _ln(p + ".ln_2", a.hidden)
        s += _conv1d(p + ".mlp.c_fc", a.hidden, a.inner, 0.02)
        s += _conv1d(p + ".mlp.c_proj", a.inner, a.hidden, proj_std)
    s += _ln(prefix + ".ln_f", a.hidden)
    return s


def decoder_extra_specs(prefix: str, a: ModelArch) -> List[Spec]:
    """vision_projection / cross_attention / attention_norm (reference model.py:521-535)."""
    h = a.gpt.hidden
    xav = math.sqrt(6.0 / (h + 3 * h))
    return (_linear(prefix + ".vision_projection.0", h, a.proj_dim)
            + [Spec(prefix + ".cross_attention.in_proj_weight", (3 * h, h), f"uniform:{xav}"),
               Spec(prefix + ".cross_attention.in_proj_bias", (3 * h,), "zeros")]
            + [Spec(prefix + ".cross_attention.out_proj.weight", (h, h), f"uniform:{1.0 / math.sqrt(h)}"),
               Spec(prefix + ".cross_attention.out_proj.bias", (h,), "zeros")]
            + _ln(prefix + ".attention_norm", h))


SEGMENT_ORDER = ("vit", "vision_head", "text_tower", "text_head", "decoder")


def model_specs(a: ModelArch) -> "OrderedDict[str, List[Spec]]":
    segs: "OrderedDict[str, List[Spec]]" = OrderedDict()
    segs["vit"] = vit_specs("vision_encoder.vision_model", a.vit)
    segs["vision_head"] = head_specs("vision_encoder.projection", a.vit.hidden, a.proj_dim)
    segs["text_tower"] = gpt2_specs("text_encoder.text_model", a.gpt, a.text_vocab)
    segs["text_head"] = head_specs("text_encoder.projection", a.gpt.hidden, a.proj_dim)
    segs["decoder"] = (gpt2_specs("caption_decoder.lm_model.transformer", a.gpt, a.dec_vocab, pad_vocab=True)
                       + decoder_extra_specs("caption_decoder", a))
    return segs


class Segment:
    """One contiguous fp32 buffer + views; optional bf16 mirror / grad / AdamW moments."""

    def __init__(self, name: str, specs: List[Spec], device: torch.device, trainable: bool):
        self.name = name
        self.specs = specs
        self.trainable = trainable
        self.index: Dict[str, Tuple[int, Tuple[int, ...]]] = OrderedDict()
        off = 0
        for sp in specs:
            if sp.name in self.index:
                raise ValueError(f"duplicate parameter {sp.name}")
            self.index[sp.name] = (off, sp.shape)
            off += _round_up(sp.alloc, ALIGN)
        self.numel = off
        self.device = device
        self.fp32 = torch.zeros(off, dtype=torch.float32, device=device)
        self.bf16: Optional[torch.Tensor] = None
        self.bf16_version = 0  # bumped whenever the mirror is refreshed from the masters (derived copies key on it)
        self.grad: Optional[torch.Tensor] = None
        self.exp_avg: Optional[torch.Tensor] = None
        self.exp_avg_sq: Optional[torch.Tensor] = None

    def _view(self, flat: torch.Tensor, name: str) -> torch.Tensor:
        off, shape = self.index[name]
        n = int(math.prod(shape))
        return flat[off:off + n].view(shape)

    def w(self, name: str) -> torch.Tensor:
        return self._view(self.fp32, name)

    def wb(self, name: str) -> torch.Tensor:
        if self.bf16 is None:
            raise RuntimeError(f"segment {self.name}: bf16 mirror not materialised")
        return self._view(self.bf16, name)

    def g(self, name: str) -> torch.Tensor:
        if self.grad is None:
            raise RuntimeError(f"segment {self.name}: no gradient buffer (frozen?)")
        return self._view(self.grad, name)

    def padded(self, flat: torch.Tensor, name: str) -> torch.Tensor:
        """View including the zero pad rows (``[rows + pad_rows, cols]``)."""
        off, shape = self.index[name]
        sp = next(s for s in self.specs if s.name == name)
        rows = shape[0] + sp.pad_rows
        return flat[off:off + rows * int(math.prod(shape[1:]))].view((rows,) + tuple(shape[1:]))

    def ensure_bf16(self) -> None:
        if self.bf16 is None:
            self.bf16 = torch.empty(self.numel, dtype=torch.bfloat16, device=self.device)
        self.bf16.copy_(self.fp32)
        self.bf16_version += 1

    def ensure_train_state(self) -> None:
        if not self.trainable:
            raise RuntimeError(f"segment {self.name} is frozen")
        if self.grad is None:
            self.grad = torch.zeros(self.numel, dtype=torch.float32, device=self.device)
            self.exp_avg = torch.zeros_like(self.grad)
            self.exp_avg_sq = torch.zeros_like(self.grad)


class ParamStore:
    """All segments of one model replica."""

    def __init__(self, arch: ModelArch, device, seed: Optional[int] = 0,
                 frozen: Iterable[str] = ("vit",), segments: Optional[Iterable[str]] = None):
        self.arch = arch
        self.device = torch.device(device)
        specs = model_specs(arch)
        wanted = list(segments) if segments is not None else list(specs)
        frozen = set(frozen)
        self.segments: "OrderedDict[str, Segment]" = OrderedDict()
        for name in SEGMENT_ORDER:
            if name in wanted:
                self.segments[name] = Segment(name, specs[name], self.device, name not in frozen)
        self._owner: Dict[str, Segment] = {}
        for seg in self.segments.values():
            for n in seg.index:
                self._owner[n] = seg
        if seed is not None:
            self.init_random(seed)

    # -- lookup -----------------------------------------------------------------
    def seg_of(self, name: str) -> Segment:
        return self._owner[name]

    def w(self, name: str) -> torch.Tensor:
        return self._owner[name].w(name)

    def wb(self, name: str) -> torch.Tensor:
        return self._owner[name].wb(name)

    def g(self, name: str) -> torch.Tensor:
        return self._owner[name].g(name)

    def names(self) -> List[str]:
        return list(self._owner)

    # -- init / io --------------------------------------------------------------
    def init_random(self, seed: int) -> None:
        """Seeded init on the CPU generator (identical on every box), then upload."""
        gen = torch.Generator(device="cpu").manual_seed(seed)
        for seg in self.segments.values():
            host = torch.zeros(seg.numel, dtype=torch.float32)
            for sp in seg.specs:
                off, _ = seg.index[sp.name]
                v = host[off:off + sp.numel]
                kind, _, arg = sp.init.partition(":")
                if kind == "normal":
                    v.normal_(0.0, float(arg), generator=gen)
                elif kind == "uniform":
                    v.uniform_(-float(arg), float(arg), generator=gen)
                elif kind == "ones":
                    v.fill_(1.0)
                elif kind != "zeros":
                    raise ValueError(sp.init)
            seg.fp32.copy_(host)
            if seg.bf16 is not None:
                seg.ensure_bf16()

    def state_dict(self, aliases: bool = True) -> "OrderedDict[str, torch.Tensor]":
        """Views (not copies) under the reference's key names.

        With ``aliases`` the two duplicate registrations the reference's
        ``state_dict`` carries are added: the ViT under ``vision_encoder.clip_model.``
        (model.py:126-127 keeps the whole CLIPModel *and* its ``vision_model``)
        and the tied ``lm_head.weight`` (modeling_gpt2.py:638-644).
        """
        out: "OrderedDict[str, torch.Tensor]" = OrderedDict()
        for seg in self.segments.values():
            for n in seg.index:
                out[n] = seg.w(n)
                if aliases and n.startswith("vision_encoder.vision_model."):
                    out["vision_encoder.clip_model." + n[len("vision_encoder."):]] = seg.w(n)
        wte = "caption_decoder.lm_model.transformer.wte.weight"
        if aliases and wte in out:
            out["caption_decoder.lm_model.lm_head.weight"] = out[wte]
        return out

    def load_state_dict(self, sd: Dict[str, torch.Tensor], strict: bool = False) -> List[str]:
        """Copy matching tensors in; returns the list of our names that were missing."""
        missing = []
        for n, seg in self._owner.items():
            if n in sd:
                t = sd[n]
                if tuple(t.shape) != tuple(seg.index[n][1]):
                    raise ValueError(f"{n}: shape {tuple(t.shape)} != {seg.index[n][1]}")
                seg.w(n).copy_(t.to(torch.float32))
            else:
                missing.append(n)
        if strict and missing:
            raise KeyError(f"missing keys: {missing[:5]} ...")
        for seg in self.segments.values():
            if seg.bf16 is not None:
                seg.ensure_bf16()
        return missing

    def trainable_segments(self) -> List[Segment]:
        return [s for s in self.segments.values() if s.trainable]

    def num_params(self) -> int:
        return sum(sp.numel for seg in self.segments.values() for sp in seg.specs)
