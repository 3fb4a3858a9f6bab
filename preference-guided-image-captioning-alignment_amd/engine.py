"""Forward/backward schedules of the towers on the HIP kernels (no autograd, no CPU fallback).

Each class owns the launch order of one tower and the activations its backward needs.
Activations live in a reusable ``Workspace`` sized once per batch geometry (288 GB of HBM:
everything a backward needs is kept resident, nothing is recomputed except the LM-head
logits tiles, which never reach HBM in the forward).

Residual streams are f32 (as under the reference's autocast, where LayerNorm outputs and the
residual adds stay f32); GEMM operands are bf16 with f32 MFMA accumulation.

Reference call sites replaced are cited per method.
"""
from __future__ import annotations

import os
from dataclasses import dataclass
from typing import Dict, List, Optional, Tuple

import torch

from . import hip
from .arch import GptArch, ModelArch, VitArch
from .params import ParamStore, Segment, VOCAB_TILE

F32, BF16, I32, I64 = torch.float32, torch.bfloat16, torch.int32, torch.int64
ATTN_BWD_MAX_S = 512   # include/pgca_hip.h PGCA_ATTN_MAX_S


class Workspace:
    """Named, reusable device buffers (allocated on first use, reused every step)."""

    # packed (variable-length) batches change their row count from step to step: a buffer that has to GROW is given this
    # much head-room, so the first few batches settle the sizes and later ones only re-slice
    GROW = 1.125

    def __init__(self, device):
        self.device = torch.device(device)
        self.bufs: Dict[str, torch.Tensor] = {}

    def get(self, key: str, shape, dtype, zero: bool = False) -> torch.Tensor:
        shape = tuple(int(s) for s in shape)
        n = 1
        for s in shape:
            n *= s
        t = self.bufs.get(key)
        if t is None or t.dtype != dtype or t.numel() < n:
            grow = t is not None and t.dtype == dtype
            t = torch.empty(max(int(n * self.GROW) if grow else n, 1), dtype=dtype, device=self.device)
            self.bufs[key] = t
        v = t[:n].view(shape)
        if zero:
            v.zero_()
        return v

    def nbytes(self) -> int:
        return sum(t.numel() * t.element_size() for t in self.bufs.values())


# ------------------------------------------------------------------------------------------ dropout plan
KIND_EMBD, KIND_ATTN, KIND_RESID_ATTN, KIND_RESID_MLP, KIND_VPROJ, KIND_XATTN, KIND_HEAD = range(7)
TOWER_DECODER, TOWER_TEXT, TOWER_VHEAD, TOWER_THEAD = range(4)
_M32 = 0xFFFFFFFF


def _hash32(x: int) -> int:
    x &= _M32
    x ^= x >> 16
    x = (x * 0x7FEB352D) & _M32
    x ^= x >> 15
    x = (x * 0x846CA68B) & _M32
    x ^= x >> 16
    return x


class DropoutPlan:
    """Seeds of the reference's train-mode dropout sites (reference model.py:139,341,524,531 and GPT-2's
    embd/attn/resid dropouts, modeling_gpt2.py:66,224,242,604).  Masks are a pure function of
    (base seed, step, tower, layer, site kind, element index), so every backward kernel replays the forward's
    mask from the same triple instead of reading a stored one.  ``site`` returns the ``(seed, threshold,
    scale)`` triple the kernels take, or None when dropout is off (p == 0 or eval).

    Two probabilities, as in the reference: ``p`` (``model.dropout``, configs/default.yaml:22) at the reference's own
    sites - projection heads, ``vision_projection``, cross-attention weights - and ``p_gpt`` at GPT-2's internal
    embd / attn / resid sites, which HF fixes at 0.1 in ``GPT2Config`` whatever ``model.dropout`` says (the reference
    never overrides them, model.py:311-312,505-506).  ``p_gpt=None`` means "same as p" (kernel tests, the bench)."""

    GPT_KINDS = (KIND_EMBD, KIND_ATTN, KIND_RESID_ATTN, KIND_RESID_MLP)

    def __init__(self, p: float = 0.0, base_seed: int = 0, p_gpt: Optional[float] = None):
        self.p, self.base_seed, self.step, self.active = float(p), int(base_seed), 0, True
        self.p_gpt = float(p) if p_gpt is None else float(p_gpt)

    def _p(self, kind: int) -> float:
        return self.p_gpt if kind in self.GPT_KINDS else self.p

    def _site(self, step: int, tower: int, layer: int, kind: int):
        seed = _hash32(_hash32(self.base_seed * 0x9E3779B1 + step) ^ ((tower << 24) | ((layer & 0xFFFF) << 8) | kind))
        return hip.drop_args(seed, self._p(kind))

    @property
    def on(self) -> bool:
        return self.active and (self.p > 0.0 or self.p_gpt > 0.0)

    def site(self, tower: int, layer: int, kind: int):
        if not self.on:
            return None
        return self._site(self.step, tower, layer, kind)

    def bind(self, tower: int):
        """Closure over the CURRENT step: the backward of this forward replays exactly these seeds."""
        if not self.on:
            return None
        step = self.step
        return lambda layer, kind: self._site(step, tower, layer, kind)


# ------------------------------------------------------------------------------------------ batches
PACK_PAD = 64   # packed row counts are rounded up to the K tile of the weight-gradient GEMMs (zero filler rows)


@dataclass
class RowPack:
    """Packed (variable-length) row layout of a right-padded [Bq, S] batch: only positions ``t < len[b]`` (``len`` = 1 +
    last unmasked position) own a row, sequence after sequence, followed by ``Mp - n`` zero filler rows that round the
    row count up to ``PACK_PAD``.  Nothing a padded position computes reaches a loss term or a gradient (causal AND
    key-padding mask, masked mean, masked loss: reference model.py:449-456,1069-1083), so the
    trunk runs on these rows only.  Positions, key masks, log-sum-exps and every dropout index stay keyed on the padded
    (b, t), so a packed launch reproduces the padded one token for token."""
    cu: torch.Tensor        # [nseq+1] int32: first row of each sequence; cu[Bq] = n, cu[nseq] = Mp (filler "sequences")
    row_ids: torch.Tensor   # [Mp] int32: padded position b*S + t of each row, -1 for filler rows
    mask: torch.Tensor      # [nseq, S] int32 key mask; the extra rows (the filler's) are ones
    lens: torch.Tensor      # [Bq] int32
    n: int                  # rows that hold a token
    Mp: int                 # rows incl. filler (multiple of PACK_PAD)
    Bq: int
    S: int

    @property
    def nseq(self) -> int:
        """Sequences of an attention launch over the packed rows: the batch's plus the filler pseudo-sequences (at most
        S rows each, so they obey the kernels' length bound)."""
        return self.Bq + n_filler_seqs(self.S)


def n_filler_seqs(S: int) -> int:
    return (PACK_PAD - 1 + S - 1) // S


@dataclass
class SeqBatch:
    """Token batch + the integer index work of the loss, prepared once per batch.

    ``row_map[r]`` is the flat (b*S + t) hidden-state row whose logits score token
    ``targets[r] = ids[b, t+1]``; only rows with ``mask[b, t+1] == 1`` are kept (the shift and the
    mask product of reference model.py:1069-1083 / components.py:340-357).  Rows are sorted by
    sequence, so a sequence's rows are contiguous.
    """
    ids: torch.Tensor          # [Bq, S] int64 (device)
    mask: torch.Tensor         # [Bq, S] int32 (device)
    row_map: torch.Tensor      # [Mc] int32
    targets: torch.Tensor      # [Mc] int64  (bit-exact copy of labels[:, 1:] at the kept positions)
    seq_of_row: torch.Tensor   # [Mc] int32
    counts: torch.Tensor       # [Bq] int32  = sum(mask[:, 1:])
    n_rows: int
    Bq: int
    S: int
    pack: Optional[RowPack] = None          # packed row layout of the same batch
    row_map_packed: Optional[torch.Tensor] = None   # [Mc] int32: row_map in packed row numbers


class PendingRowPack:
    """Row-pack index preparation issued on the current stream; ``result()`` waits for the two row counts."""

    def __init__(self, mask1, lens, cu, row_ids, n_host, event, Bq, S):
        self._f, self._n, self._ev, self.Bq, self.S = (mask1, lens, cu, row_ids), n_host, event, Bq, S

    def result(self) -> RowPack:
        self._ev.synchronize()
        n, Mp = int(self._n[0]), int(self._n[1])
        mask1, lens, cu, row_ids = self._f
        return RowPack(cu=cu, row_ids=row_ids[:max(Mp, 1)], mask=mask1, lens=lens, n=n, Mp=Mp, Bq=self.Bq, S=self.S)


def _issue_pack(mask1: torch.Tensor, Bq: int, S: int, counts=None, row_map=None):
    dev = mask1.device
    cap = (Bq * S + PACK_PAD - 1) // PACK_PAD * PACK_PAD
    lens = torch.empty(Bq, dtype=I32, device=dev)
    cu = torch.empty(Bq + n_filler_seqs(S) + 1, dtype=I32, device=dev)
    row_ids = torch.empty(cap, dtype=I32, device=dev)
    n_dev = torch.empty(2, dtype=I32, device=dev)
    hip.seq_pack_prepare(mask1, Bq, S, PACK_PAD, lens, cu, row_ids, n_dev, counts=counts, row_map=row_map)
    return lens, cu, row_ids, n_dev


def prepare_row_pack_async(mask32: torch.Tensor) -> PendingRowPack:
    """mask32 [Bq, S] int32 on the device (0/1) -> the packed row layout (Stage 1: no log-prob index work)."""
    Bq, S = mask32.shape
    mask1 = torch.empty(Bq + n_filler_seqs(S), S, dtype=I32, device=mask32.device)
    mask1[:Bq].copy_(mask32)
    lens, cu, row_ids, n_dev = _issue_pack(mask1, Bq, S)
    n_host = torch.empty(2, dtype=I32, pin_memory=True)
    n_host.copy_(n_dev, non_blocking=True)
    ev = torch.cuda.Event()
    ev.record()
    return PendingRowPack(mask1, lens, cu, row_ids, n_host, ev, Bq, S)


def make_row_pack(mask32: torch.Tensor) -> RowPack:
    return prepare_row_pack_async(mask32).result()


class PendingSeqBatch:
    """Index preparation issued on the current stream; ``result()`` waits for the one number the host needs (how many
    rows are scored - it sizes the LM-head launch) and hands out the views."""

    def __init__(self, ids, mask1, row_map, targets, seq_of_row, counts, n_host, event, Bq, S, pack=None):
        self._f = (ids, mask1, row_map, targets, seq_of_row, counts)
        self._n, self._ev, self.Bq, self.S, self._pack = n_host, event, Bq, S, pack

    def result(self) -> "SeqBatch":
        self._ev.synchronize()
        n = int(self._n[0])
        if n <= 0:
            raise ValueError("no scored token in the batch (every caption has <= 1 real token)")
        ids, mask1, row_map, targets, seq_of_row, counts = self._f
        pack, rmp = None, None
        if self._pack is not None:
            lens, cu, row_ids, rmp = self._pack
            npk, Mp = int(self._n[1]), int(self._n[2])
            pack = RowPack(cu=cu, row_ids=row_ids[:Mp], mask=mask1, lens=lens, n=npk, Mp=Mp, Bq=self.Bq, S=self.S)
            rmp = rmp[:n]
        return SeqBatch(ids=ids, mask=mask1[:self.Bq], row_map=row_map[:n], targets=targets[:n],
                        seq_of_row=seq_of_row[:n], counts=counts, n_rows=n, Bq=self.Bq, S=self.S, pack=pack,
                        row_map_packed=rmp)


def prepare_seq_batch_async(ids: torch.Tensor, mask: torch.Tensor, device, pack: bool = True) -> PendingSeqBatch:
    """``ids`` / ``mask`` [Bq, S] on the host (pinned: the copy is asynchronous) or on the device.  Everything -
    the int32 key mask, the shifted gather indices, the compaction, the per-sequence counts and (``pack``) the packed
    row layout - runs on the current HIP stream (``pgca_seq_batch_prepare`` / ``pgca_seq_pack_prepare``); nothing is
    computed on the host."""
    dev = torch.device(device)
    ids_d = ids.detach().to(dev, I64, non_blocking=True).contiguous()
    mask_d = mask.detach().to(dev, I64, non_blocking=True).contiguous()
    Bq, S = ids_d.shape
    if S < 2:
        raise ValueError("sequences need at least 2 positions (the first token is never scored)")
    cap = Bq * (S - 1)
    mask1 = torch.empty(Bq + n_filler_seqs(S), S, dtype=I32, device=dev)   # + the key-mask rows of the packed filler
    counts = torch.empty(Bq, dtype=I32, device=dev)
    row_map = torch.empty(cap, dtype=I32, device=dev)
    targets = torch.empty(cap, dtype=I64, device=dev)
    seq_of_row = torch.empty(cap, dtype=I32, device=dev)
    n_dev = torch.empty(3, dtype=I32, device=dev)
    hip.seq_batch_prepare(ids_d, mask_d, Bq, S, counts, mask1, row_map, targets, seq_of_row, n_dev)
    pk = None
    if pack:
        rmp = row_map.clone()
        lens, cu, row_ids, n2 = _issue_pack(mask1, Bq, S, counts=counts, row_map=rmp)
        n_dev[1:].copy_(n2)
        pk = (lens, cu, row_ids, rmp)
    n_host = torch.empty(3, dtype=I32, pin_memory=True)
    n_host.copy_(n_dev, non_blocking=True)
    ev = torch.cuda.Event()
    ev.record()
    return PendingSeqBatch(ids_d, mask1, row_map, targets, seq_of_row, counts, n_host, ev, Bq, S, pk)


def make_seq_batch(ids: torch.Tensor, mask: torch.Tensor, device, pack: bool = True) -> SeqBatch:
    """Token batch + index work of the loss, prepared on the device (one host wait: the row counts)."""
    return prepare_seq_batch_async(ids, mask, device, pack).result()


# ------------------------------------------------------------------------------------------ helpers
def _bias_grad(ws: Workspace, M: int, N: int, ld: int, gout: torch.Tensor, x_bf16=None, x_f32=None) -> None:
    nb = hip.colsum_blocks(M)
    part = ws.get("colsum_part", (nb, N), F32)
    hip.colsum(M, N, ld, part, x_bf16=x_bf16, x_f32=x_f32)
    hip.colsum_finish(part, nb, N, gout, accumulate=True)


def _ln_param_grads(part: torch.Tensor, nb: int, H: int, *outs: torch.Tensor) -> None:
    """part [len(outs), nb, H] partial planes -> accumulated into the given gradient views (one launch)."""
    hip.colsum_finish4(part, len(outs), nb, H, outs, accumulate=True)


class _P:
    """Resolved views of one parameter: f32 master, bf16 mirror, gradient (if trainable)."""
    __slots__ = ("w", "b", "g")

    def __init__(self, seg: Segment, name: str):
        self.w = seg.w(name)
        self.b = seg.wb(name) if seg.bf16 is not None else None
        self.g = seg.g(name) if seg.grad is not None else None


# ------------------------------------------------------------------------------------------ GPT-2 trunk
class GptTrunk:
    """GPT-2 blocks + ln_f on an f32 residual stream (HF GPT2Block, modeling_gpt2.py:246-310).

    Weights are Conv1D ``[in, out]``: forward is an NN GEMM, dgrad NT, wgrad TN - all from the
    one bf16 mirror.  The attn / resid dropout sites take the triple ``drop(layer, kind)`` yields (train mode) and are
    replayed in the backward; without ``drop`` they are the identity (eval semantics).
    """

    GELU_PAIR = os.environ.get("PGCA_GELU_PAIR", "1") != "0"   # A/B switch: 0 = evaluate the sigmoid again in the backward

    def __init__(self, store: ParamStore, prefix: str, arch: GptArch, ws: Workspace, tag: str):
        self.arch, self.ws, self.tag = arch, ws, tag
        seg = store.seg_of(prefix + ".ln_f.weight")
        self.seg = seg
        self.layers = []
        for i in range(arch.layers):
            p = f"{prefix}.h.{i}"
            self.layers.append({k: _P(seg, p + n) for k, n in (
                ("ln1w", ".ln_1.weight"), ("ln1b", ".ln_1.bias"), ("wqkv", ".attn.c_attn.weight"),
                ("bqkv", ".attn.c_attn.bias"), ("wo", ".attn.c_proj.weight"), ("bo", ".attn.c_proj.bias"),
                ("ln2w", ".ln_2.weight"), ("ln2b", ".ln_2.bias"), ("wfc", ".mlp.c_fc.weight"),
                ("bfc", ".mlp.c_fc.bias"), ("wpr", ".mlp.c_proj.weight"), ("bpr", ".mlp.c_proj.bias"))})
        self.lnf_w, self.lnf_b = _P(seg, prefix + ".ln_f.weight"), _P(seg, prefix + ".ln_f.bias")
        self.saved: Optional[dict] = None
        # flat-buffer range [start, end) of each layer's parameters (for bucketed gradient all-reduce)
        starts = [seg.index[f"{prefix}.h.{i}.ln_1.weight"][0] for i in range(arch.layers)]
        starts.append(seg.index[prefix + ".ln_f.weight"][0])
        self.layer_ranges = [(starts[i], starts[i + 1]) for i in range(arch.layers)]
        self.grad_hook = None  # callable(layer_index) fired when a layer's gradients are complete
        self.overlap_wgrad = True   # weight-gradient launches on a side stream (see backward)
        self._wgrad_stream: Optional[torch.cuda.Stream] = None

    def _buf(self, name, shape, dtype, zero=False):
        return self.ws.get(f"{self.tag}.{name}", shape, dtype, zero)

    def forward(self, h0: torch.Tensor, mask: Optional[torch.Tensor], Bq: int, S: int, save: bool,
                drop=None, pack: Optional[RowPack] = None) -> torch.Tensor:
        """h0 [Bq*S, H] f32 (already including positions) -> residual stream after the last block.
        ``drop(layer, kind)`` (optional) yields the dropout triple of a site (train mode).
        ``pack``: h0 holds the packed rows [pack.Mp, H] instead (``RowPack``): every kernel runs on those rows only, the
        attention takes the sequence offsets, and the dropout sites hash the padded position of each row."""
        a, H, I = self.arch, self.arch.hidden, self.arch.inner
        if S > a.n_pos:
            raise ValueError(f"sequence length {S} exceeds GPT-2's {a.n_pos} learned positions")
        if save and S > ATTN_BWD_MAX_S:
            raise ValueError(f"training needs sequence length <= {ATTN_BWD_MAX_S} (attention backward keeps the query "
                             f"gradients of every 128-token block in registers); got {S}")
        M = Bq * S
        cu = rows = None
        nseq = Bq
        if pack is not None:   # the filler rows are one more (unmasked) sequence of the attention launch
            M, cu, rows, mask, nseq = pack.Mp, pack.cu, pack.row_ids, pack.mask, pack.nseq
        L = len(self.layers)
        sv = {"M": M, "Bq": Bq, "S": S, "mask": mask, "drop": drop, "pack": pack} if save else None
        dsite = (lambda li, kind: drop(li, kind)) if drop is not None else (lambda li, kind: None)
        h = h0
        for li, P in enumerate(self.layers):
            k = f"l{li}." if save else ""
            ln1 = self._buf(k + "ln1", (M, H), BF16)
            m1 = self._buf(k + "m1", (M,), F32)
            r1 = self._buf(k + "r1", (M,), F32)
            hip.layernorm_fwd(h, M, H, P["ln1w"].w, P["ln1b"].w, a.eps, y_bf16=ln1, mean=m1, rstd=r1)
            qkv = self._buf(k + "qkv", (M, 3 * H), BF16)
            hip.gemm(ln1, P["wqkv"].b, M, 3 * H, H, hip.NN, bias=P["bqkv"].w, out_bf16=qkv)
            att = self._buf(k + "att", (M, H), BF16)
            lse = self._buf(k + "lse", (nseq, a.heads, S), F32)
            hip.attention_fwd(qkv, mask, nseq, S, a.heads, True, att, lse, drop=dsite(li, KIND_ATTN), cu=cu)
            hm = self._buf(k + "hm", (M, H), F32) if save else h
            hip.gemm(att, P["wo"].b, M, H, H, hip.NN, bias=P["bo"].w, residual=h, out_f32=hm,
                     drop=dsite(li, KIND_RESID_ATTN), drop_rows=rows)
            ln2 = self._buf(k + "ln2", (M, H), BF16)
            m2 = self._buf(k + "m2", (M,), F32)
            r2 = self._buf(k + "r2", (M,), F32)
            hip.layernorm_fwd(hm, M, H, P["ln2w"].w, P["ln2b"].w, a.eps, y_bf16=ln2, mean=m2, rstd=r2)
            act = self._buf(k + "act", (M, I), BF16)
            # training: the epilogue has the sigmoid of gelu_new in registers and leaves gelu_new'(pre) for the backward, whose
            # data-gradient GEMM then only multiplies (EPI_MUL_AUX) instead of evaluating the sigmoid a second time
            pre = self._buf(k + "pre", (M, I), BF16) if save else None
            hip.gemm(ln2, P["wfc"].b, M, I, H, hip.NN,
                     epilogue=hip.EPI_GELU_NEW_D if (save and self.GELU_PAIR) else hip.EPI_GELU_NEW,
                     bias=P["bfc"].w, out_bf16=act, aux_out=pre)
            hn = self._buf(f"l{li + 1}.hin", (M, H), F32) if save else hm
            hip.gemm(act, P["wpr"].b, M, H, I, hip.NN, bias=P["bpr"].w, residual=hm, out_f32=hn,
                     drop=dsite(li, KIND_RESID_MLP), drop_rows=rows)
            if save:
                sv[li] = dict(hin=h, ln1=ln1, m1=m1, r1=r1, qkv=qkv, att=att, lse=lse, hm=hm, ln2=ln2, m2=m2, r2=r2,
                              act=act, pre=pre)
            h = hn
        if save:
            sv["hL"] = h
            self.saved = sv
        return h

    # -- incremental decoding (generation) ---------------------------------------------------------
    SKINNY_ROWS = 32   # up to this many sequences a decode step runs on pgca_gemm_skinny (faster than the tile GEMMs up to 32 rows: DESIGN 5)

    def decode_cache(self, R: int, smax: int) -> dict:
        """Per-layer K/V cache of an incremental decode over ``R`` sequences of at most ``smax`` positions: one resident
        ``[R * smax, 3H]`` bf16 buffer per layer in the SAME q|k|v row layout the attention kernel reads (row r*smax + t
        holds position t of sequence r), plus the shared attention-output rows and the offsets ``cu[r] = r * smax``."""
        a, H = self.arch, self.arch.hidden
        if smax > a.n_pos:
            raise ValueError(f"{smax} positions exceed GPT-2's {a.n_pos} learned positions")
        L = len(self.layers)
        kv_all = self._buf("gen.kv", (L, R * smax, 3 * H), BF16)     # ONE allocation: a beam reorder is one gather
        kv = [kv_all[li] for li in range(L)]
        att = self._buf("gen.att", (R * smax, H), BF16)
        cu = self.ws.bufs.get(self.tag + ".gen.cu")
        if cu is None or cu.numel() != R + 1 or int(getattr(self, "_gen_smax", -1)) != smax:
            cu = (torch.arange(R + 1, dtype=I32, device=self.ws.device) * smax).contiguous()
            self.ws.bufs[self.tag + ".gen.cu"] = cu
            self._gen_smax = smax
        return dict(kv=kv, kv_all=kv_all, att=att, cu=cu, R=R, smax=smax)

    def decode_step(self, x: torch.Tensor, st: dict, t: int) -> torch.Tensor:
        """One position of an incremental forward: ``x`` [R, H] f32 is the input embedding (+ position) of position
        ``t`` of every sequence; keys / values of positions < t come from the cache ``st`` (HF ``use_cache``,
        modeling_gpt2.py:144-226 with ``layer_past``).  Updates ``x`` in place to the residual stream after the last
        block and returns it.  The query rows of the cached positions ride along in the attention launch (their outputs
        are rewritten with the values they already had): at caption lengths (<= 128: one key tile) that costs nothing
        next to the weight reads of the R-row GEMMs."""
        a, H, I = self.arch, self.arch.hidden, self.arch.inner
        R, smax = st["R"], st["smax"]
        if not 0 <= t < smax:
            raise ValueError(f"position {t} outside the cache ({smax})")
        y = self._buf("gen.y", (R, H), BF16)
        act = self._buf("gen.act", (R, I), BF16)
        att = st["att"]
        att_t = att.view(-1)[t * H:]
        L = len(self.layers)
        if R <= min(self.SKINNY_ROWS, hip.SKINNY_MAX_M) and I <= 8192 and 3 * H <= 8192:
            # a handful of rows: every product is one pass over its weights spread over all CUs (pgca_gemm_skinny), and
            # each LayerNorm rides on the finish pass of the product before it
            need = max(hip.gemm_skinny_workspace(R, n, k) for n, k in ((3 * H, H), (H, H), (I, H), (H, I))) // 4
            sc = self._buf("gen.scratch", (need,), F32)
            P0 = self.layers[0]
            hip.layernorm_fwd(x, R, H, P0["ln1w"].w, P0["ln1b"].w, a.eps, y_bf16=y)
            for li, P in enumerate(self.layers):
                kv = st["kv"][li]
                hip.gemm_skinny(y, P["wqkv"].b, R, 3 * H, H, sc, bias=P["bqkv"].w, out_bf16=kv.view(-1)[t * 3 * H:],
                                ld_out_bf16=smax * 3 * H)
                hip.attention_fwd(kv, None, R, t + 1, a.heads, True, att, None, cu=st["cu"])
                hip.gemm_skinny(att_t, P["wo"].b, R, H, H, sc, lda=smax * H, bias=P["bo"].w, residual=x, out_f32=x,
                                ln=(P["ln2w"].w, P["ln2b"].w, a.eps), ln_out=y)
                hip.gemm_skinny(y, P["wfc"].b, R, I, H, sc, bias=P["bfc"].w, act=hip.EPI_GELU_NEW, out_bf16=act)
                nxt = (self.layers[li + 1]["ln1w"].w, self.layers[li + 1]["ln1b"].w) if li + 1 < L else \
                    (self.lnf_w.w, self.lnf_b.w)
                hip.gemm_skinny(act, P["wpr"].b, R, H, I, sc, bias=P["bpr"].w, residual=x, out_f32=x,
                                ln=(nxt[0], nxt[1], a.eps), ln_out=y)
            st["lnf_done"] = True      # y holds ln_f(x): the LM head's operand
            return x
        st["lnf_done"] = False
        for li, P in enumerate(self.layers):
            kv = st["kv"][li]
            hip.layernorm_fwd(x, R, H, P["ln1w"].w, P["ln1b"].w, a.eps, y_bf16=y)
            hip.gemm(y, P["wqkv"].b, R, 3 * H, H, hip.NN, bias=P["bqkv"].w, out_bf16=kv.view(-1)[t * 3 * H:],
                     ld_out_bf16=smax * 3 * H)
            hip.attention_fwd(kv, None, R, t + 1, a.heads, True, att, None, cu=st["cu"])
            hip.gemm(att_t, P["wo"].b, R, H, H, hip.NN, lda=smax * H, bias=P["bo"].w, residual=x, out_f32=x)
            hip.layernorm_fwd(x, R, H, P["ln2w"].w, P["ln2b"].w, a.eps, y_bf16=y)
            hip.gemm(y, P["wfc"].b, R, I, H, hip.NN, epilogue=hip.EPI_GELU_NEW, bias=P["bfc"].w, out_bf16=act)
            hip.gemm(act, P["wpr"].b, R, H, I, hip.NN, bias=P["bpr"].w, residual=x, out_f32=x)
        return x

    def decode_reorder(self, st: dict, src: torch.Tensor) -> None:
        """Beam search: sequence r continues from cached sequence ``src[r]`` (HF ``_reorder_cache``)."""
        R, smax = st["R"], st["smax"]
        v = st["kv_all"].view(len(self.layers), R, smax, -1)[:, :, :st.get("t", smax - 1) + 1]   # the filled positions
        v.copy_(v.index_select(1, src))

    def top_drop(self):
        """Dropout triple the producer of ``g_bf`` (the ln_f backward) must apply: last layer's mlp dropout."""
        drop = self.saved.get("drop") if self.saved else None
        return drop(len(self.layers) - 1, KIND_RESID_MLP) if drop is not None else None

    def backward(self, g: torch.Tensor, g_bf: torch.Tensor) -> torch.Tensor:
        """g / g_bf: dL/d(stream after last block) as f32 and bf16 [M, H]. Returns dL/dh0 (f32).
        Weight/bias/LN gradients are ACCUMULATED into the segment's flat gradient buffer.
        In train mode g_bf must already carry the mask of the last layer's mlp dropout (``top_drop()``).

        The grouped weight-gradient launch of a block (192 workgroups of 128 KiB LDS: 192 of the 256 CUs for ~0.85 ms)
        goes to a SIDE stream: nothing in the data-gradient chain of the following blocks depends on it, so their kernels
        run on the 64 CUs it leaves idle instead of waiting behind it.  Its operands that the next block would overwrite
        (``dpre``, ``dqkv``) alternate between two buffers by layer parity; events order the rest (see below)."""
        sv = self.saved
        assert sv is not None, "forward(save=True) must precede backward"
        a, H, I = self.arch, self.arch.hidden, self.arch.inner
        M, Bq, S = sv["M"], sv["Bq"], sv["S"]
        pack: Optional[RowPack] = sv.get("pack")
        cu, rows, nseq = (pack.cu, pack.row_ids, pack.nseq) if pack is not None else (None, None, Bq)
        ws = self.ws
        nb = hip.layernorm_bwd_blocks(M)
        part4 = ws.get("ln_part", (4, nb, H), F32)   # planes: dgamma, dbeta, sum(add_to), sum(dx_out)
        part, partx = part4[:2], part4[2:]
        drop = sv.get("drop")
        dsite = (lambda li, kind: drop(li, kind)) if drop is not None else (lambda li, kind: None)
        L = len(self.layers)
        main = torch.cuda.current_stream()
        side = None
        if self.overlap_wgrad and g.is_cuda:
            if self._wgrad_stream is None:
                self._wgrad_stream = torch.cuda.Stream()
            side = self._wgrad_stream
        done: Dict[int, torch.cuda.Event] = {}      # layer -> its weight gradients are complete

        def settle(layer: int) -> None:
            """Main stream may not pass this point before the weight-gradient launch of ``layer`` has finished."""
            ev = done.pop(layer, None)
            if ev is not None:
                main.wait_event(ev)
                if self.grad_hook is not None:      # the layer's gradient range is complete only now
                    self.grad_hook(layer)

        for li in range(L - 1, -1, -1):
            P, s = self.layers[li], sv[li]
            par = "1" if (li & 1) else "0"
            settle(li + 2)   # that launch read dpre / dqkv of this parity
            # ---- MLP: h_out = hm + c_proj(gelu(c_fc(ln2(hm))))
            dpre = self._buf("dpre" + par, (M, I), BF16)
            # the same GEMM leaves the column sums of dpre per 64-row block: mlp.c_fc's bias gradient without a second
            # pass over the M x 4H matrix
            nbr = (M + 63) // 64
            dpre_cs = self._buf("dpre_colsum", (nbr, I), F32)
            # s["pre"] holds gelu_new'(pre-activation) when the forward ran EPI_GELU_NEW_D, else the pre-activation
            hip.gemm(g_bf, P["wpr"].b, M, I, H, hip.NT, epilogue=hip.EPI_MUL_AUX if self.GELU_PAIR else hip.EPI_DGELU_NEW,
                     aux_in=s["pre"], out_bf16=dpre, colsum_part=dpre_cs)
            hip.colsum_finish(dpre_cs, nbr, I, P["bfc"].g, accumulate=True)
            wgrads = [(s["act"], g_bf, I, H, M, P["wpr"].g)]   # the layer's four weight gradients go out together
            dln = self._buf("dln", (M, H), BF16)
            hip.gemm(dpre, P["wfc"].b, M, H, I, hip.NT, out_bf16=dln)
            wgrads.append((s["ln2"], dpre, H, I, M, P["wfc"].g))
            g2 = self._buf("g_b" if (li & 1) else "g_a", (M, H), F32)
            g2_bf = self._buf("gbf_b" if (li & 1) else "gbf_a", (M, H), BF16)
            hip.layernorm_bwd(s["hm"], M, H, P["ln2w"].w, s["m2"], s["r2"], g2, dy_bf16=dln, add_to=g, dx_bf16=g2_bf,
                              part=part, part_extra=partx, drop_add=dsite(li, KIND_RESID_MLP),
                              drop_dx=dsite(li, KIND_RESID_ATTN), drop_rows=rows)
            # the same pass summed g (bias gradient of mlp.c_proj) and g2 (bias gradient of attn.c_proj)
            _ln_param_grads(part4, nb, H, P["ln2w"].g, P["ln2b"].g, P["bpr"].g, P["bo"].g)
            # ---- attention: hm = hin + c_proj(attn(c_attn(ln1(hin))))
            datt = self._buf("datt", (M, H), BF16)
            hip.gemm(g2_bf, P["wo"].b, M, H, H, hip.NT, out_bf16=datt)
            wgrads.append((s["att"], g2_bf, H, H, M, P["wo"].g))
            dqkv = self._buf("dqkv" + par, (M, 3 * H), BF16)
            hip.attention_bwd(s["qkv"], s["att"], datt, s["lse"], sv["mask"], nseq, S, a.heads, True, dqkv,
                              drop=dsite(li, KIND_ATTN), cu=cu)
            hip.gemm(dqkv, P["wqkv"].b, M, H, 3 * H, hip.NT, out_bf16=dln)
            wgrads.append((s["ln1"], dqkv, H, 3 * H, M, P["wqkv"].g))
            # one launch, whole K per tile: no split-K atomics (gemm256_group_tn_kernel)
            if side is not None:
                side.wait_stream(main)              # operands above are complete
                with torch.cuda.stream(side):
                    hip.gemm_wgrad_group(wgrads)
                    ev = torch.cuda.Event()
                    ev.record(side)
                done[li] = ev
            else:
                hip.gemm_wgrad_group(wgrads)
            # (moving this HBM-bound column sum to the side stream as well measured neutral: it stays here)
            _bias_grad(ws, M, 3 * H, 3 * H, P["bqkv"].g, x_bf16=dqkv)
            g3 = self._buf("g_c" if (li & 1) else "g_d", (M, H), F32)
            g3_bf = self._buf("gbf_c" if (li & 1) else "gbf_d", (M, H), BF16)
            settle(li + 1)   # that launch read its incoming g_bf from the buffer g3_bf is about to overwrite
            hip.layernorm_bwd(s["hin"], M, H, P["ln1w"].w, s["m1"], s["r1"], g3, dy_bf16=dln, add_to=g2,
                              dx_bf16=g3_bf, part=part,
                              drop_dx=dsite(li - 1, KIND_RESID_MLP) if li > 0 else None, drop_rows=rows)
            _ln_param_grads(part, nb, H, P["ln1w"].g, P["ln1b"].g)
            g, g_bf = g3, g3_bf
            if side is None and self.grad_hook is not None:
                self.grad_hook(li)
        for layer in sorted(done, reverse=True):   # layers 1 and 0: join the side stream before anything reads the gradients
            settle(layer)
        return g


# ------------------------------------------------------------------------------------------ projection heads
class ProjHead:
    """Linear -> ReLU -> Dropout (fused into the first GEMM's epilogue when ``drop`` is given) -> Linear -> LayerNorm
    (reference model.py:136-142,338-344)."""

    def __init__(self, store: ParamStore, prefix: str, in_dim: int, proj: int, ws: Workspace, tag: str):
        seg = store.seg_of(prefix + ".0.weight")
        self.seg, self.ws, self.tag, self.inn, self.P = seg, ws, tag, in_dim, proj
        self.w0, self.b0 = _P(seg, prefix + ".0.weight"), _P(seg, prefix + ".0.bias")
        self.w3, self.b3 = _P(seg, prefix + ".3.weight"), _P(seg, prefix + ".3.bias")
        self.lnw, self.lnb = _P(seg, prefix + ".4.weight"), _P(seg, prefix + ".4.bias")
        self.saved = None

    def forward(self, x_bf: torch.Tensor, B: int, save: bool, drop=None) -> torch.Tensor:
        """x_bf [B, in] bf16 -> embeddings [B, P] f32 (un-normalised).  ``drop``: triple of the head's Dropout."""
        ws, t, Pd = self.ws, self.tag, self.P
        h1 = ws.get(t + ".h1", (B, Pd), BF16)
        hip.gemm(x_bf, self.w0.b, B, Pd, self.inn, hip.NT, epilogue=hip.EPI_RELU, bias=self.b0.w, out_bf16=h1,
                 drop=drop)
        h2 = ws.get(t + ".h2", (B, Pd), F32)
        hip.gemm(h1, self.w3.b, B, Pd, Pd, hip.NT, bias=self.b3.w, out_f32=h2)
        emb = ws.get(t + ".emb", (B, Pd), F32)
        mean, rstd = ws.get(t + ".mean", (B,), F32), ws.get(t + ".rstd", (B,), F32)
        hip.layernorm_fwd(h2, B, Pd, self.lnw.w, self.lnb.w, 1e-5, y_f32=emb, mean=mean, rstd=rstd)
        if save:
            self.saved = dict(x=x_bf, h1=h1, h2=h2, mean=mean, rstd=rstd, B=B, drop=drop)
        return emb

    def backward(self, demb: torch.Tensor, need_dx: bool) -> Optional[torch.Tensor]:
        s, ws, t, Pd = self.saved, self.ws, self.tag, self.P
        B = s["B"]
        nb = hip.layernorm_bwd_blocks(B)
        part = ws.get("ln_part_small", (2, nb, Pd), F32)
        dh2 = ws.get(t + ".dh2", (B, Pd), F32)
        dh2_bf = ws.get(t + ".dh2bf", (B, Pd), BF16)
        hip.layernorm_bwd(s["h2"], B, Pd, self.lnw.w, s["mean"], s["rstd"], dh2, dy_f32=demb, dx_bf16=dh2_bf, part=part)
        _ln_param_grads(part, nb, Pd, self.lnw.g, self.lnb.g)
        # nn.Linear [out,in]: dW = dY^t X (TN), dX = dY W (NN)
        hip.gemm(dh2_bf, s["h1"], Pd, Pd, B, hip.TN, lda=Pd, ldb=Pd, out_f32=self.w3.g, accumulate=True)
        _bias_grad(ws, B, Pd, Pd, self.b3.g, x_f32=dh2)
        dh1 = ws.get(t + ".dh1", (B, Pd), BF16)
        hip.gemm(dh2_bf, self.w3.b, B, Pd, Pd, hip.NN, epilogue=hip.EPI_DRELU, aux_in=s["h1"], out_bf16=dh1,
                 drop=s["drop"])
        hip.gemm(dh1, s["x"], Pd, self.inn, B, hip.TN, lda=Pd, ldb=self.inn, out_f32=self.w0.g, accumulate=True)
        _bias_grad(ws, B, Pd, Pd, self.b0.g, x_bf16=dh1)
        if not need_dx:
            return None
        dx = ws.get(t + ".dx", (B, self.inn), F32)
        hip.gemm(dh1, self.w0.b, B, self.inn, Pd, hip.NN, out_f32=dx)
        return dx


# ------------------------------------------------------------------------------------------ ViT
class VisionTower:
    """CLIP vision transformer (HF CLIPVisionTransformer, modeling_clip.py:200-218,353-384,594-650) as called at
    reference model.py:222-230.  Run ONCE per image per step.

    Frozen (``freeze_vision_backbone=True``, configs/default.yaml:23 - every shipped config) it is forward-only and keeps no
    activations.  Left trainable (the constructor default, reference model.py:150-164 only freezes on request) ``forward(...,
    save=True)`` keeps each layer's activations and ``backward(dpooled)`` accumulates the gradients of every tower
    parameter: nn.Linear weights are ``[out, in]``, so forward is NT, the data gradient NN and the weight gradient TN with
    the output gradient as the A operand."""

    def __init__(self, store: ParamStore, arch: VitArch, ws: Workspace):
        self.arch, self.ws = arch, ws
        p = "vision_encoder.vision_model"
        seg = store.seg_of(p + ".post_layernorm.weight")
        self.seg = seg
        self.trainable = seg.grad is not None
        G = (lambda n: seg.g(n)) if self.trainable else (lambda n: None)
        self._g = G
        self._names = dict(cls=p + ".embeddings.class_embedding", pos=p + ".embeddings.position_embedding.weight",
                           patch=p + ".embeddings.patch_embedding.weight")
        self.cls = seg.w(self._names["cls"])
        self.pos = seg.w(self._names["pos"])
        self._wpatch_name = self._names["patch"]
        # K of the patch-embedding GEMM: 3*P*P, padded with zero columns to the GEMM's 64-deep tile when it is not a
        # multiple of 8 (ViT-L/14: 588 -> 640); the padded bf16 weight is a private copy refreshed when the mirror changes
        self.patch_k = arch.patch_dim if arch.patch_dim % 8 == 0 else (arch.patch_dim + 63) // 64 * 64
        self._wpatch_pad, self._wpatch_ver = None, -1
        self.pre = (seg.w(p + ".pre_layrnorm.weight"), seg.w(p + ".pre_layrnorm.bias"))
        self.post = (seg.w(p + ".post_layernorm.weight"), seg.w(p + ".post_layernorm.bias"))
        self.pre_g = (G(p + ".pre_layrnorm.weight"), G(p + ".pre_layrnorm.bias"))
        self.post_g = (G(p + ".post_layernorm.weight"), G(p + ".post_layernorm.bias"))
        H = arch.hidden
        self.layers = []
        for i in range(arch.layers):
            q = f"{p}.encoder.layers.{i}"
            wq_off = seg.index[q + ".self_attn.q_proj.weight"][0]
            bq_off = seg.index[q + ".self_attn.q_proj.bias"][0]
            assert seg.index[q + ".self_attn.v_proj.weight"][0] == wq_off + 2 * H * H, "q,k,v must be contiguous"
            assert seg.index[q + ".self_attn.v_proj.bias"][0] == bq_off + 2 * H
            L = dict(
                wqkv=seg.bf16[wq_off:wq_off + 3 * H * H].view(3 * H, H), bqkv=seg.fp32[bq_off:bq_off + 3 * H],
                wo=seg.wb(q + ".self_attn.out_proj.weight"), bo=seg.w(q + ".self_attn.out_proj.bias"),
                ln1=(seg.w(q + ".layer_norm1.weight"), seg.w(q + ".layer_norm1.bias")),
                ln2=(seg.w(q + ".layer_norm2.weight"), seg.w(q + ".layer_norm2.bias")),
                w1=seg.wb(q + ".mlp.fc1.weight"), b1=seg.w(q + ".mlp.fc1.bias"),
                w2=seg.wb(q + ".mlp.fc2.weight"), b2=seg.w(q + ".mlp.fc2.bias"))
            if self.trainable:
                L.update(
                    g_wqkv=seg.grad[wq_off:wq_off + 3 * H * H].view(3 * H, H), g_bqkv=seg.grad[bq_off:bq_off + 3 * H],
                    g_wo=G(q + ".self_attn.out_proj.weight"), g_bo=G(q + ".self_attn.out_proj.bias"),
                    g_ln1=(G(q + ".layer_norm1.weight"), G(q + ".layer_norm1.bias")),
                    g_ln2=(G(q + ".layer_norm2.weight"), G(q + ".layer_norm2.bias")),
                    g_w1=G(q + ".mlp.fc1.weight"), g_b1=G(q + ".mlp.fc1.bias"),
                    g_w2=G(q + ".mlp.fc2.weight"), g_b2=G(q + ".mlp.fc2.bias"))
            self.layers.append(L)
        self.saved: Optional[dict] = None

    def _patch_weight(self) -> torch.Tensor:
        """[H, patch_k] bf16 view of the conv weight (zero-padded copy when 3*P*P is not a multiple of 8)."""
        w = self.seg.wb(self._wpatch_name).view(self.arch.hidden, self.arch.patch_dim)
        if self.patch_k == self.arch.patch_dim:
            return w
        if self._wpatch_pad is None or self._wpatch_ver != self.seg.bf16_version:
            if self._wpatch_pad is None:
                self._wpatch_pad = torch.zeros(self.arch.hidden, self.patch_k, dtype=BF16, device=self.ws.device)
            self._wpatch_pad[:, :self.arch.patch_dim].copy_(w)
            self._wpatch_ver = self.seg.bf16_version
        return self._wpatch_pad

    def forward(self, pixels: torch.Tensor, save: bool = False) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
        """pixels [B,3,I,I] f32 -> (features [B,T,H] f32, pooled [B,H] f32, pooled bf16).
        ``save`` (trainable tower only): keep what ``backward`` needs."""
        a, ws = self.arch, self.ws
        save = bool(save) and self.trainable
        B = pixels.shape[0]
        H, T, G, D = a.hidden, a.tokens, a.grid, a.patch_dim
        if save and T > ATTN_BWD_MAX_S:
            raise ValueError(f"a trainable vision tower needs <= {ATTN_BWD_MAX_S} tokens per image; got {T}")
        M = B * T
        Kp = self.patch_k
        cols = ws.get("vit.cols", (B * G * G, Kp), BF16)
        hip.patchify(pixels, B, a.image, a.patch, cols, ld_out=Kp)
        pe = ws.get("vit.pe", (B * G * G, H), F32)
        hip.gemm(cols, self._patch_weight(), B * G * G, H, Kp, hip.NT, out_f32=pe)
        x0 = ws.get("vit.x0", (M, H), F32)
        hip.vit_assemble(pe, self.cls, self.pos, B, T, H, x0)
        x = ws.get("vit.x", (M, H), F32)
        sv = None
        if save:
            sv = dict(B=B, M=M, cols=cols, x0=x0, m0=ws.get("vit.m0", (M,), F32), r0=ws.get("vit.r0", (M,), F32))
            hip.layernorm_fwd(x0, M, H, self.pre[0], self.pre[1], a.eps, y_f32=x, mean=sv["m0"], rstd=sv["r0"])
        else:
            hip.layernorm_fwd(x0, M, H, self.pre[0], self.pre[1], a.eps, y_f32=x)
        for li, Lw in enumerate(self.layers):
            k = f"vit.l{li}." if save else "vit."
            y = ws.get(k + "y", (M, H), BF16)
            qkv = ws.get(k + "qkv", (M, 3 * H), BF16)
            att = ws.get(k + "att", (M, H), BF16)
            act = ws.get(k + "act", (M, a.mlp), BF16)
            if not save:   # in place on the residual stream, nothing kept
                hip.layernorm_fwd(x, M, H, Lw["ln1"][0], Lw["ln1"][1], a.eps, y_bf16=y)
                hip.gemm(y, Lw["wqkv"], M, 3 * H, H, hip.NT, bias=Lw["bqkv"], out_bf16=qkv)
                hip.attention_fwd(qkv, None, B, T, a.heads, False, att, None)
                hip.gemm(att, Lw["wo"], M, H, H, hip.NT, bias=Lw["bo"], residual=x, out_f32=x)
                hip.layernorm_fwd(x, M, H, Lw["ln2"][0], Lw["ln2"][1], a.eps, y_bf16=y)
                hip.gemm(y, Lw["w1"], M, a.mlp, H, hip.NT, epilogue=hip.EPI_QUICK_GELU, bias=Lw["b1"], out_bf16=act)
                hip.gemm(act, Lw["w2"], M, H, a.mlp, hip.NT, bias=Lw["b2"], residual=x, out_f32=x)
                continue
            s = dict(xin=x, y1=y, qkv=qkv, att=att, act=act,
                     m1=ws.get(k + "m1", (M,), F32), r1=ws.get(k + "r1", (M,), F32),
                     m2=ws.get(k + "m2", (M,), F32), r2=ws.get(k + "r2", (M,), F32),
                     lse=ws.get(k + "lse", (B, a.heads, T), F32), xm=ws.get(k + "xm", (M, H), F32),
                     y2=ws.get(k + "y2", (M, H), BF16), pre=ws.get(k + "pre", (M, a.mlp), BF16))
            hip.layernorm_fwd(x, M, H, Lw["ln1"][0], Lw["ln1"][1], a.eps, y_bf16=y, mean=s["m1"], rstd=s["r1"])
            hip.gemm(y, Lw["wqkv"], M, 3 * H, H, hip.NT, bias=Lw["bqkv"], out_bf16=qkv)
            hip.attention_fwd(qkv, None, B, T, a.heads, False, att, s["lse"])
            hip.gemm(att, Lw["wo"], M, H, H, hip.NT, bias=Lw["bo"], residual=x, out_f32=s["xm"])
            hip.layernorm_fwd(s["xm"], M, H, Lw["ln2"][0], Lw["ln2"][1], a.eps, y_bf16=s["y2"], mean=s["m2"], rstd=s["r2"])
            hip.gemm(s["y2"], Lw["w1"], M, a.mlp, H, hip.NT, epilogue=hip.EPI_QUICK_GELU, bias=Lw["b1"], out_bf16=act,
                     aux_out=s["pre"])
            xn = ws.get(f"vit.l{li + 1}.xin", (M, H), F32)
            hip.gemm(act, Lw["w2"], M, H, a.mlp, hip.NT, bias=Lw["b2"], residual=s["xm"], out_f32=xn)
            sv[li] = s
            x = xn
        cls_rows = ws.bufs.get("vit.clsrows")
        if cls_rows is None or cls_rows.numel() != B:
            cls_rows = (torch.arange(B, dtype=I32, device=ws.device) * T).contiguous()
            ws.bufs["vit.clsrows"] = cls_rows
        pooled = ws.get("vit.pooled", (B, H), F32)
        pooled_bf = ws.get("vit.pooledbf", (B, H), BF16)
        if save:
            sv.update(xL=x, cls_rows=cls_rows, mp=ws.get("vit.mp", (B,), F32), rp=ws.get("vit.rp", (B,), F32))
            hip.layernorm_fwd(x, B, H, self.post[0], self.post[1], a.eps, row_map=cls_rows, y_f32=pooled,
                              y_bf16=pooled_bf, mean=sv["mp"], rstd=sv["rp"])
        else:
            hip.layernorm_fwd(x, B, H, self.post[0], self.post[1], a.eps, row_map=cls_rows, y_f32=pooled,
                              y_bf16=pooled_bf)
        self.saved = sv
        return x.view(B, T, H), pooled, pooled_bf

    def backward(self, dpooled: torch.Tensor) -> None:
        """dpooled f32 [B, H] = dL/d(pooler_output).  Accumulates the gradient of every tower parameter (the ``features``
        output carries no gradient on the reference's training paths: trainer.py:467-478,578-603 use the embeddings)."""
        sv = self.saved
        if sv is None:
            raise RuntimeError("VisionTower.backward needs forward(save=True) on a trainable tower")
        a, ws = self.arch, self.ws
        B, M = sv["B"], sv["M"]
        H, T, I = a.hidden, a.tokens, a.mlp
        nb = hip.layernorm_bwd_blocks(M)
        part = ws.get("vit.ln_part", (2, nb, H), F32)
        # post_layernorm saw the CLS rows only: their gradients are scattered into an otherwise zero stream gradient
        g = ws.get("vit.g_a", (M, H), F32)
        g.zero_()
        nbp = hip.layernorm_bwd_blocks(B)
        partp = ws.get("vit.ln_part_cls", (2, nbp, H), F32)
        hip.layernorm_bwd(sv["xL"], B, H, self.post[0], sv["mp"], sv["rp"], g, dy_f32=dpooled, row_map=sv["cls_rows"],
                          part=partp)
        _ln_param_grads(partp, nbp, H, self.post_g[0], self.post_g[1])
        g_bf = ws.get("vit.gbf_a", (M, H), BF16)
        hip.cast_bf16(g, g_bf, M * H)
        dpre = ws.get("vit.dpre", (M, I), BF16)
        dln = ws.get("vit.dln", (M, H), BF16)
        datt = ws.get("vit.datt", (M, H), BF16)
        dqkv = ws.get("vit.dqkv", (M, 3 * H), BF16)
        for li in range(len(self.layers) - 1, -1, -1):
            Lw, s = self.layers[li], sv[li]
            # ---- MLP: x_out = xm + fc2(quick_gelu(fc1(ln2(xm))))
            hip.gemm(g_bf, Lw["w2"], M, I, H, hip.NN, epilogue=hip.EPI_DQUICK_GELU, aux_in=s["pre"], out_bf16=dpre)
            hip.gemm(g_bf, s["act"], H, I, M, hip.TN, lda=H, ldb=I, out_f32=Lw["g_w2"], accumulate=True)
            _bias_grad(ws, M, H, H, Lw["g_b2"], x_f32=g)
            hip.gemm(dpre, Lw["w1"], M, H, I, hip.NN, out_bf16=dln)
            hip.gemm(dpre, s["y2"], I, H, M, hip.TN, lda=I, ldb=H, out_f32=Lw["g_w1"], accumulate=True)
            _bias_grad(ws, M, I, I, Lw["g_b1"], x_bf16=dpre)
            g2 = ws.get("vit.g_b", (M, H), F32)
            g2_bf = ws.get("vit.gbf_b", (M, H), BF16)
            hip.layernorm_bwd(s["xm"], M, H, Lw["ln2"][0], s["m2"], s["r2"], g2, dy_bf16=dln, add_to=g, dx_bf16=g2_bf,
                              part=part)
            _ln_param_grads(part, nb, H, Lw["g_ln2"][0], Lw["g_ln2"][1])
            # ---- attention: xm = xin + out_proj(attn(qkv(ln1(xin))))
            hip.gemm(g2_bf, Lw["wo"], M, H, H, hip.NN, out_bf16=datt)
            hip.gemm(g2_bf, s["att"], H, H, M, hip.TN, lda=H, ldb=H, out_f32=Lw["g_wo"], accumulate=True)
            _bias_grad(ws, M, H, H, Lw["g_bo"], x_f32=g2)
            hip.attention_bwd(s["qkv"], s["att"], datt, s["lse"], None, B, T, a.heads, False, dqkv)
            hip.gemm(dqkv, Lw["wqkv"], M, H, 3 * H, hip.NN, out_bf16=dln)
            hip.gemm(dqkv, s["y1"], 3 * H, H, M, hip.TN, lda=3 * H, ldb=H, out_f32=Lw["g_wqkv"], accumulate=True)
            _bias_grad(ws, M, 3 * H, 3 * H, Lw["g_bqkv"], x_bf16=dqkv)
            hip.layernorm_bwd(s["xin"], M, H, Lw["ln1"][0], s["m1"], s["r1"], g, dy_bf16=dln, add_to=g2, dx_bf16=g_bf,
                              part=part)
            _ln_param_grads(part, nb, H, Lw["g_ln1"][0], Lw["g_ln1"][1])
        # ---- pre_layrnorm, embeddings
        dx0 = ws.get("vit.g_b", (M, H), F32)
        hip.layernorm_bwd(sv["x0"], M, H, self.pre[0], sv["m0"], sv["r0"], dx0, dy_f32=g, part=part)
        _ln_param_grads(part, nb, H, self.pre_g[0], self.pre_g[1])
        Np, D, Kp = B * (T - 1), a.patch_dim, self.patch_k
        dpe = ws.get("vit.dpe", (Np, H), BF16)
        hip.vit_assemble_bwd(dx0, B, T, H, dpe, self._g(self._names["cls"]), self._g(self._names["pos"]))
        # conv weight [H, 3, P, P] == [H, D]: dW = dpe^t cols (the zero pad columns of ``cols`` are not read)
        hip.gemm(dpe, sv["cols"], H, D, Np, hip.TN, lda=H, ldb=Kp, out_f32=self._g(self._names["patch"]).view(H, D),
                 accumulate=True)


# ------------------------------------------------------------------------------------------ caption decoder
class CaptionDecoderEngine:
    """Reference ``CaptionDecoder.forward`` (model.py:583-610) + tied LM head + log-prob gather
    (model.py:1069-1083 / components.py:340-362), fused so the [B,S,V] logits never reach HBM.

    The 1-token cross-attention is collapsed: softmax over one key is 1, so
    ``attended[b, s, :] = W_o (W_v pv_b + b_v) + b_o`` for every s; q/k rows of ``in_proj`` receive
    exactly zero gradient, as in the reference.
    """

    # rows of dlogits materialised at a time in the LM-head backward (bf16 workspace, 0.8 GB): 32 x 197 tiles = 24.6 rounds
    # of the 256 CUs (a 4096-row chunk ends on a 31 %-full 13th round)
    LM_CHUNK = int(os.environ.get("PGCA_LM_CHUNK", "8192"))

    def __init__(self, store: ParamStore, arch: ModelArch, ws: Workspace, tag: str):
        self.arch, self.ws, self.tag = arch, ws, tag
        g = arch.gpt
        p = "caption_decoder"
        t = p + ".lm_model.transformer"
        seg = store.seg_of(t + ".wte.weight")
        self.seg = seg
        self.trunk = GptTrunk(store, t, g, ws, tag + ".trunk")
        self.wte, self.wpe = _P(seg, t + ".wte.weight"), _P(seg, t + ".wpe.weight")
        self.V = arch.dec_vocab
        self.Vp = (self.V + VOCAB_TILE - 1) // VOCAB_TILE * VOCAB_TILE
        self.wte_pad_bf = seg.padded(seg.bf16, t + ".wte.weight")          # [Vp, H], pad rows are zero
        self.vp_w, self.vp_b = _P(seg, p + ".vision_projection.0.weight"), _P(seg, p + ".vision_projection.0.bias")
        self.inw, self.inb = _P(seg, p + ".cross_attention.in_proj_weight"), _P(seg, p + ".cross_attention.in_proj_bias")
        self.ow, self.ob = _P(seg, p + ".cross_attention.out_proj.weight"), _P(seg, p + ".cross_attention.out_proj.bias")
        self.anw, self.anb = _P(seg, p + ".attention_norm.weight"), _P(seg, p + ".attention_norm.bias")
        self.saved = None

    def _buf(self, name, shape, dtype, zero=False):
        return self.ws.get(f"{self.tag}.{name}", shape, dtype, zero)

    # -- forward --------------------------------------------------------------------------------
    def _prefix(self, emb: torch.Tensor, Bq: int, drop=None):
        """emb [Bq, P] f32 -> the cross-attention term of the decoder input (+ saved intermediates).

        Eval / p = 0: one collapsed vector per sequence, ``attended [Bq, H]``.  Train mode: the attention-weight
        dropout of nn.MultiheadAttention makes the term per (sequence, head, position), so the per-head vectors
        ``U [Bq, 8, H]`` are produced instead and the embedding kernel applies the replayable head weights."""
        H, Pd, XH = self.arch.gpt.hidden, self.arch.proj_dim, self.arch.xattn_heads
        dh = H // XH
        emb_bf = self._buf("emb_bf", (Bq, Pd), BF16)
        hip.cast_bf16(emb, emb_bf, Bq * Pd)
        pv = self._buf("pv", (Bq, H), BF16)
        pv_raw = self._buf("pv_raw", (Bq, H), BF16) if drop is not None else None
        hip.gemm(emb_bf, self.vp_w.b, Bq, H, Pd, hip.NT, epilogue=hip.EPI_TANH, bias=self.vp_b.w, out_bf16=pv,
                 aux_out=pv_raw, drop=drop(0, KIND_VPROJ) if drop is not None else None)
        vv = self._buf("vv", (Bq, H), BF16)
        hip.gemm(pv, self.inw.b[2 * H:], Bq, H, H, hip.NT, bias=self.inb.w[2 * H:], out_bf16=vv)
        if drop is None:
            att = self._buf("attended", (Bq, H), F32)
            hip.gemm(vv, self.ow.b, Bq, H, H, hip.NT, bias=self.ob.w, out_f32=att)
            return dict(emb_bf=emb_bf, pv=pv, pv_raw=pv, vv=vv, att=att, U=None)
        U = self._buf("U", (Bq, XH, H), F32)
        for h in range(XH):  # U[:, h, :] = vv[:, head h] @ W_o[:, head h]^t
            hip.gemm(vv[:, h * dh:], self.ow.b[:, h * dh:], Bq, H, dh, hip.NT, lda=H, ldb=H, out_f32=U[:, h],
                     ld_out_f32=XH * H)
        return dict(emb_bf=emb_bf, pv=pv, pv_raw=pv_raw, vv=vv, att=None, U=U)

    def hidden(self, emb: torch.Tensor, sb: SeqBatch, save: bool, drop=None, packed: bool = False) -> torch.Tensor:
        """Residual stream after the last block, [Bq*S, H] f32 - or, ``packed``, the rows of ``sb.pack`` only
        ([pack.Mp, H]: the padded positions are never computed).  ``drop(layer, kind)``: train-mode dropout sites."""
        a = self.arch.gpt
        H, Bq, S, XH = a.hidden, sb.Bq, sb.S, self.arch.xattn_heads
        pack = sb.pack if packed else None
        if packed and pack is None:
            raise ValueError("packed=True needs a SeqBatch prepared with pack=True")
        M = pack.Mp if pack is not None else Bq * S
        pk = dict(row_ids=pack.row_ids, n_rows=M) if pack is not None else {}
        pf = self._prefix(emb, Bq, drop)
        h0 = self._buf("h0", (M, H), F32)
        m0, r0 = self._buf("m0", (M,), F32), self._buf("r0", (M,), F32)
        if drop is None:
            hip.embed_fwd(sb.ids, Bq, S, H, self.wte.w, self.wpe.w, h0, attended=pf["att"], gamma=self.anw.w,
                          beta=self.anb.w, eps=1e-5, mean=m0, rstd=r0, **pk)
        else:
            hip.embed_fwd(sb.ids, Bq, S, H, self.wte.w, self.wpe.w, h0, attended=self.ob.w, att_stride=0,
                          gamma=self.anw.w, beta=self.anb.w, eps=1e-5, mean=m0, rstd=r0, U=pf["U"], xheads=XH,
                          drop_x=drop(0, KIND_XATTN), drop_e=drop(0, KIND_EMBD), **pk)
        hL = self.trunk.forward(h0, sb.mask, Bq, S, save, drop, pack=pack)
        if save:
            self.saved = dict(sb=sb, m0=m0, r0=r0, hL=hL, drop=drop, pack=pack, **pf)
        return hL

    def token_logprobs(self, hL: torch.Tensor, sb: SeqBatch, save: bool, packed: bool = False) -> torch.Tensor:
        """ln_f on the scored rows only, then the fused LM head: tok_lp [Mc] f32."""
        a = self.arch.gpt
        H, Mc = a.hidden, sb.n_rows
        hf = self._buf("hf", (Mc, H), BF16)
        mf, rf = self._buf("mf", (Mc,), F32), self._buf("rf", (Mc,), F32)
        hip.layernorm_fwd(hL, Mc, H, self.trunk.lnf_w.w, self.trunk.lnf_b.w, a.eps,
                          row_map=sb.row_map_packed if packed else sb.row_map, y_bf16=hf, mean=mf, rstd=rf)
        nparts = 2 * (self.Vp // 128)
        smax = self._buf("smax", (Mc, nparts), F32)
        ssum = self._buf("ssum", (Mc, nparts), F32)
        tval = self._buf("tval", (Mc,), F32)
        hip.gemm(hf, self.wte.b, Mc, self.V, H, hip.NT, epilogue=hip.EPI_ROWSTATS, targets=sb.targets, stat_max=smax,
                 stat_sum=ssum, stat_ld=nparts, target_val=tval)
        lse = self._buf("lse", (Mc,), F32)
        tok = self._buf("tok_lp", (Mc,), F32)
        hip.rowstats_combine(smax, ssum, nparts, 2 * ((self.V + 127) // 128), tval, Mc, lse=lse, out_logprob=tok)
        if save:
            self.saved.update(hf=hf, mf=mf, rf=rf, lse=lse)
        return tok

    def sequence_logprobs(self, emb: torch.Tensor, sb: SeqBatch, reduce: str, save: bool, drop=None,
                          packed: Optional[bool] = None) -> torch.Tensor:
        """seq_lp [Bq]: 'sum' (components.py:357-362) or 'mean' (model.py:1082-1083).  ``packed`` (default: whenever the
        batch carries a packed layout) runs the trunk on the real tokens' rows only."""
        packed = (sb.pack is not None) if packed is None else bool(packed)
        hL = self.hidden(emb, sb, save, drop, packed)
        tok = self.token_logprobs(hL, sb, save, packed)
        out = self._buf("seq_lp", (sb.Bq,), F32)
        hip.seq_reduce(tok, sb.seq_of_row, sb.n_rows, sb.Bq, sb.counts, 1 if reduce == "mean" else 0, out)
        if save:
            self.saved["reduce"] = reduce
        return out

    def logits(self, emb: torch.Tensor, sb: SeqBatch) -> torch.Tensor:
        """Materialised logits [Bq, S, V] f32 for API compatibility (``mode='generation'``)."""
        a = self.arch.gpt
        H, M = a.hidden, sb.Bq * sb.S
        hL = self.hidden(emb, sb, False)
        hf = self._buf("hf_full", (M, H), BF16)
        hip.layernorm_fwd(hL, M, H, self.trunk.lnf_w.w, self.trunk.lnf_b.w, a.eps, y_bf16=hf)
        out = torch.empty(M, self.V, dtype=F32, device=self.ws.device)
        hip.gemm(hf, self.wte.b, M, self.V, H, hip.NT, out_f32=out)
        return out.view(sb.Bq, sb.S, self.V)

    # -- generation (reference CaptionDecoder.generate, model.py:621-678) --------------------------
    def prefix_embedding(self, emb: torch.Tensor) -> torch.Tensor:
        """``vision_projection(vision_features)`` (Linear + Tanh, eval): the single input embedding HF's ``generate`` is
        started from (model.py:653-655) - [B, H] f32."""
        B = emb.shape[0]
        H, Pd = self.arch.gpt.hidden, self.arch.proj_dim
        emb_bf = self._buf("gen.emb_bf", (B, Pd), BF16)
        hip.cast_bf16(emb.contiguous(), emb_bf, B * Pd)
        pv = torch.empty(B, H, dtype=F32, device=self.ws.device)
        hip.gemm(emb_bf, self.vp_w.b, B, H, Pd, hip.NT, epilogue=hip.EPI_TANH, bias=self.vp_b.w, out_f32=pv)
        return pv

    def next_token_logits(self, pv: torch.Tensor, ids: torch.Tensor) -> torch.Tensor:
        """Logits of the next token given the prefix embedding ``pv`` [B, H] and the tokens generated so far ``ids``
        [B, t] (t >= 0): GPT-2 on ``[pv, wte(ids)] + wpe`` (``inputs_embeds`` path of HF generate: no cross-attention,
        no attention_norm - the reference feeds the projected vision vector straight to ``lm_model.generate``), causal
        attention, ``ln_f`` + tied LM head on the last position only.  This is the CACHE-FREE form - the whole prefix is
        recomputed - kept as the cross-check of the K/V-cache path (``decode_begin`` / ``decode_advance``) that
        ``generate`` runs."""
        a = self.arch.gpt
        B, t = ids.shape
        H, S = a.hidden, t + 1
        full = torch.zeros(B, S, dtype=I64, device=self.ws.device)
        if t:
            full[:, 1:] = ids
        h0 = self._buf("gen.h0", (B * S, H), F32)
        hip.embed_fwd(full, B, S, H, self.wte.w, self.wpe.w, h0)          # wte[id] + wpe[s]; row 0 is replaced below
        h0.view(B, S, H)[:, 0, :] = pv + self.wpe.w[0]
        hL = self.trunk.forward(h0, None, B, S, save=False)
        rows = (torch.arange(B, dtype=I32, device=self.ws.device) * S + (S - 1)).contiguous()
        hf = self._buf("gen.hf", (B, H), BF16)
        hip.layernorm_fwd(hL, B, H, self.trunk.lnf_w.w, self.trunk.lnf_b.w, a.eps, row_map=rows, y_bf16=hf)
        ldv = (self.V + 3) // 4 * 4
        out = torch.empty(B, ldv, dtype=F32, device=self.ws.device)
        hip.gemm(hf, self.wte.b, B, self.V, H, hip.NT, out_f32=out, ld_out_f32=ldv)
        return out[:, :self.V]

    # -- incremental decoding with a K/V cache (what HF generate does with use_cache=True, reference model.py:657-670) ----
    def decode_begin(self, pv: torch.Tensor, max_positions: int) -> torch.Tensor:
        """Start an incremental decode from the prefix embeddings ``pv`` [R, H] (position 0); returns the logits of the
        first token [R, V].  ``max_positions`` = 1 + the most tokens that will be fed back."""
        R = pv.shape[0]
        self._dec = self.trunk.decode_cache(R, int(max_positions))
        self._dec["t"] = 0
        x = self._buf("gen.x", (R, self.arch.gpt.hidden), F32)
        torch.add(pv, self.wpe.w[0], out=x)
        return self._decode_logits(x)

    # One position of the decode is ~180 small launches (24 layers x 7 kernels); issued one by one from Python that is
    # 3.4 ms per token whatever the batch (measured: 165 ms per 49-token caption at batch 1, cache or no cache).  Each
    # position's launch sequence is therefore captured ONCE into a HIP graph and replayed (one graph per (rows, cache
    # length, position): pointer offsets and the filled length are baked into the kernels' arguments).
    use_graphs = True

    def _advance_eager(self, tok: torch.Tensor, t: int) -> torch.Tensor:
        st = self._dec
        x = self._buf("gen.x", (st["R"], self.arch.gpt.hidden), F32)
        torch.index_select(self.wte.w, 0, tok, out=x)
        x.add_(self.wpe.w[t])
        return self._decode_logits(x)

    def decode_advance(self, tokens: torch.Tensor) -> torch.Tensor:
        """Feed the tokens chosen for the current position ([R] int64); returns the next token's logits [R, V]."""
        st = self._dec
        st["t"] += 1
        t = st["t"]
        if not (self.use_graphs and tokens.is_cuda):
            return self._advance_eager(tokens, t)
        tok_in = self._buf("gen.tok_in", (st["R"],), I64)
        tok_in.copy_(tokens)
        # graphs hold raw pointers: they are only valid while every buffer they touch is the allocation they captured
        stamp = (st["kv"][0].data_ptr(), st["kv"][-1].data_ptr(), st["att"].data_ptr(), tok_in.data_ptr(),
                 self.seg.bf16.data_ptr())
        cache = self.__dict__.setdefault("_graphs", {})
        if cache.get("stamp") != stamp:
            cache.clear()
            cache["stamp"] = stamp
        key = (st["R"], st["smax"], t)
        g = cache.get(key)
        if g is not None:
            g[0].replay()
            return g[1]
        out = self._advance_eager(tok_in, t)             # this call's result (also sizes every buffer)
        try:
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):                # records the same launches; nothing executes
                out_g = self._advance_eager(tok_in, t)
            cache[key] = (graph, out_g)
        except Exception:  # noqa: BLE001 - capture is an optimisation; the eager HIP launches above are the product path
            self.use_graphs = False
            torch.cuda.synchronize()
        return out

    def decode_reorder(self, src: torch.Tensor) -> None:
        self.trunk.decode_reorder(self._dec, src)

    def _decode_logits(self, x: torch.Tensor) -> torch.Tensor:
        a, st = self.arch.gpt, self._dec
        R, H = st["R"], a.hidden
        self.trunk.decode_step(x, st, st["t"])
        if st.get("lnf_done"):           # the last product's finish pass already ran ln_f
            hf = self.trunk._buf("gen.y", (R, H), BF16)
        else:
            hf = self._buf("gen.hf", (R, H), BF16)
            hip.layernorm_fwd(x, R, H, self.trunk.lnf_w.w, self.trunk.lnf_b.w, a.eps, y_bf16=hf)
        ldv = (self.V + 3) // 4 * 4
        out = self._buf("gen.logits", (R, ldv), F32)
        hip.gemm(hf, self.wte.b, R, self.V, H, hip.NT, out_f32=out, ld_out_f32=ldv)
        return out[:, :self.V]

    # -- backward -------------------------------------------------------------------------------
    def backward(self, dseq: torch.Tensor) -> torch.Tensor:
        """dseq [Bq] = dLoss/dseq_lp.  Accumulates all decoder gradients; returns dLoss/demb [Bq, P] f32."""
        s = self.saved
        sb: SeqBatch = s["sb"]
        a = self.arch.gpt
        H, Pd, Bq, S, Mc = a.hidden, self.arch.proj_dim, sb.Bq, sb.S, sb.n_rows
        pack: Optional[RowPack] = s.get("pack")
        M = pack.Mp if pack is not None else Bq * S
        cu, rows = (pack.cu, pack.row_ids) if pack is not None else (None, None)
        ws = self.ws
        rs = self._buf("row_scale", (Mc,), F32)
        # d tok_lp / d logits = onehot - softmax; the DLOGITS epilogue computes the cross-entropy form
        # row_scale * (softmax - onehot), so it is fed -dLoss/dtok_lp (mode bit 2 negates)
        hip.row_scale(dseq, sb.seq_of_row, sb.counts, Mc, (1 if s["reduce"] == "mean" else 0) | 2, rs)
        # LM head: dlogits recomputed tile by tile in row chunks
        dhf = self._buf("dhf32", (Mc, H), F32, zero=True)  # f32 + accumulate: lets the K = vocab dgrad GEMM split K
        ck = min(self.LM_CHUNK, Mc)
        dl = self._buf("dlogits", (ck, self.Vp), BF16)
        for r0 in range(0, Mc, ck):
            n = min(ck, Mc - r0)
            hfc = s["hf"][r0:r0 + n]
            hip.gemm(hfc, self.wte.b, n, self.V, H, hip.NT, epilogue=hip.EPI_DLOGITS, targets=sb.targets[r0:r0 + n],
                     row_lse=s["lse"][r0:r0 + n], row_scale=rs[r0:r0 + n], out_bf16=dl, ld_out_bf16=self.Vp,
                     out_cols=self.Vp)
            hip.gemm(dl, self.wte_pad_bf, n, H, self.Vp, hip.NN, lda=self.Vp, ldb=H, out_f32=dhf[r0:r0 + n], accumulate=True)
            hip.gemm(dl, hfc, self.V, H, n, hip.TN, lda=self.Vp, ldb=H, out_f32=self.wte.g, accumulate=True)
        # ln_f backward scatters to the scored rows; every other row of the stream gets zero gradient
        g = self.trunk._buf("g_top", (M, H), F32, zero=True)
        g_bf = self.trunk._buf("gbf_top", (M, H), BF16, zero=True)
        nb = hip.layernorm_bwd_blocks(Mc)
        part = ws.get("ln_part_f", (2, nb, H), F32)
        hip.layernorm_bwd(s["hL"], Mc, H, self.trunk.lnf_w.w, s["mf"], s["rf"], g, dy_f32=dhf,
                          row_map=sb.row_map_packed if pack is not None else sb.row_map, dx_bf16=g_bf, part=part,
                          drop_dx=self.trunk.top_drop(), drop_rows=rows)
        _ln_param_grads(part, nb, H, self.trunk.lnf_w.g, self.trunk.lnf_b.g)
        g0 = self.trunk.backward(g, g_bf)
        # embedding + attention_norm + 1-key cross-attention
        drop = s["drop"]
        XH = self.arch.xattn_heads
        dh = H // XH
        datt = self._buf("datt", (Bq, H), F32, zero=True)
        nbe = hip.embed_bwd_blocks(Bq, S)
        parte = ws.get("ln_part_e", (2, nbe, H), F32)
        dvv = self._buf("dvv", (Bq, H), BF16)
        if drop is None:
            hip.embed_bwd(g0, sb.ids, sb.mask, Bq, S, H, self.wte.g, self.wpe.g, wte=self.wte.w, attended=s["att"],
                          gamma=self.anw.w, mean=s["m0"], rstd=s["r0"], dattended=datt, part=parte, cu=cu)
            _ln_param_grads(parte, nbe, H, self.anw.g, self.anb.g)
            datt_bf = self._buf("datt_bf", (Bq, H), BF16)
            hip.cast_bf16(datt, datt_bf, Bq * H)
            # attended = vv W_o^t + b_o
            hip.gemm(datt_bf, s["vv"], H, H, Bq, hip.TN, lda=H, ldb=H, out_f32=self.ow.g, accumulate=True)
            _bias_grad(ws, Bq, H, H, self.ob.g, x_f32=datt)
            hip.gemm(datt_bf, self.ow.b, Bq, H, H, hip.NN, out_bf16=dvv)
        else:
            dU = self._buf("dU", (Bq, XH, H), F32, zero=True)
            hip.embed_bwd(g0, sb.ids, sb.mask, Bq, S, H, self.wte.g, self.wpe.g, wte=self.wte.w, attended=self.ob.w,
                          att_stride=0, gamma=self.anw.w, mean=s["m0"], rstd=s["r0"], dattended=datt, part=parte,
                          U=s["U"], dU=dU, xheads=XH, drop_x=drop(0, KIND_XATTN), drop_e=drop(0, KIND_EMBD), cu=cu)
            _ln_param_grads(parte, nbe, H, self.anw.g, self.anb.g)
            _bias_grad(ws, Bq, H, H, self.ob.g, x_f32=datt)        # d b_o = sum over rows of d e
            dU_bf = self._buf("dU_bf", (Bq, XH, H), BF16)
            hip.cast_bf16(dU, dU_bf, Bq * XH * H)
            for h in range(XH):  # U[:, h] = vv[:, head h] W_o[:, head h]^t
                hip.gemm(dU_bf[:, h], s["vv"][:, h * dh:], H, dh, Bq, hip.TN, lda=XH * H, ldb=H,
                         out_f32=self.ow.g[:, h * dh:], ld_out_f32=H, accumulate=True)
                hip.gemm(dU_bf[:, h], self.ow.b[:, h * dh:], Bq, dh, H, hip.NN, lda=XH * H, ldb=H,
                         out_bf16=dvv[:, h * dh:], ld_out_bf16=H)
        # vv = pv W_v^t + b_v   (value rows of in_proj; q/k rows keep zero gradient)
        hip.gemm(dvv, s["pv"], H, H, Bq, hip.TN, lda=H, ldb=H, out_f32=self.inw.g[2 * H:], accumulate=True)
        _bias_grad(ws, Bq, H, H, self.inb.g[2 * H:], x_bf16=dvv)
        dpv = self._buf("dpv", (Bq, H), BF16)
        hip.gemm(dvv, self.inw.b[2 * H:], Bq, H, H, hip.NN, epilogue=hip.EPI_DTANH, aux_in=s["pv_raw"], out_bf16=dpv,
                 drop=drop(0, KIND_VPROJ) if drop is not None else None)
        # pv = tanh(emb W_vp^t + b_vp)
        hip.gemm(dpv, s["emb_bf"], H, Pd, Bq, hip.TN, lda=H, ldb=Pd, out_f32=self.vp_w.g, accumulate=True)
        _bias_grad(ws, Bq, H, H, self.vp_b.g, x_bf16=dpv)
        demb = self._buf("demb", (Bq, Pd), F32)
        hip.gemm(dpv, self.vp_w.b, Bq, Pd, H, hip.NN, out_f32=demb)
        return demb


# ------------------------------------------------------------------------------------------ text tower (Stage 1)
class TextTowerEngine:
    """Reference ``TextEncoder.forward`` (model.py:437-474): GPT2Model -> masked mean pool -> head."""

    def __init__(self, store: ParamStore, arch: ModelArch, ws: Workspace, tag: str):
        self.arch, self.ws, self.tag = arch, ws, tag
        t = "text_encoder.text_model"
        seg = store.seg_of(t + ".wte.weight")
        self.seg = seg
        self.trunk = GptTrunk(store, t, arch.gpt, ws, tag + ".trunk")
        self.wte, self.wpe = _P(seg, t + ".wte.weight"), _P(seg, t + ".wpe.weight")
        self.head = ProjHead(store, "text_encoder.projection", arch.gpt.hidden, arch.proj_dim, ws, tag + ".head")
        self.saved = None

    def _buf(self, name, shape, dtype, zero=False):
        return self.ws.get(f"{self.tag}.{name}", shape, dtype, zero)

    def forward(self, ids: torch.Tensor, mask: torch.Tensor, save: bool, drop=None, head_drop=None,
                pack: Optional[RowPack] = None):
        """ids int64 [B,S], mask int32 [B,S] (device) -> (features [B,S,H] f32, pooled [B,H] f32, emb [B,P] f32).
        ``pack`` (the training steps): the tower runs on the packed rows of the real tokens only; ``features`` are then
        the packed rows [pack.Mp, H] (the padded [B,S,H] surface of ``text_features`` is the un-packed call's)."""
        a = self.arch.gpt
        B, S = ids.shape
        H = a.hidden
        M = pack.Mp if pack is not None else B * S
        cu = pack.cu if pack is not None else None
        pk = dict(row_ids=pack.row_ids, n_rows=M) if pack is not None else {}
        h0 = self._buf("h0", (M, H), F32)
        hip.embed_fwd(ids, B, S, H, self.wte.w, self.wpe.w, h0, drop_e=drop(0, KIND_EMBD) if drop is not None else None,
                      **pk)
        hL = self.trunk.forward(h0, mask, B, S, save and self.seg.grad is not None, drop, pack=pack)
        feats = self._buf("feats", (M, H), F32)
        mf, rf = self._buf("mf", (M,), F32), self._buf("rf", (M,), F32)
        hip.layernorm_fwd(hL, M, H, self.trunk.lnf_w.w, self.trunk.lnf_b.w, a.eps, y_f32=feats, mean=mf, rstd=rf)
        pooled = self._buf("pooled", (B, H), F32)
        hip.masked_mean_fwd(feats, mask, B, S, H, pooled, cu=cu)
        pooled_bf = self._buf("pooled_bf", (B, H), BF16)
        hip.cast_bf16(pooled, pooled_bf, B * H)
        emb = self.head.forward(pooled_bf, B, save, head_drop)
        if save:
            self.saved = dict(ids=ids, mask=mask, hL=hL, mf=mf, rf=rf, B=B, S=S, drop=drop, pack=pack)
        return (feats if pack is not None else feats.view(B, S, H)), pooled, emb

    def backward(self, demb: torch.Tensor) -> None:
        s, a, ws = self.saved, self.arch.gpt, self.ws
        B, S, H = s["B"], s["S"], a.hidden
        pack: Optional[RowPack] = s.get("pack")
        M = pack.Mp if pack is not None else B * S
        cu, rows = (pack.cu, pack.row_ids) if pack is not None else (None, None)
        if self.seg.grad is None:      # freeze_text_backbone=True (reference model.py:354-368): only the head trains
            self.head.backward(demb, need_dx=False)
            return
        dpooled = self.head.backward(demb, need_dx=True)
        dfeats = self._buf("dfeats", (M, H), F32)
        hip.masked_mean_bwd(dpooled, s["mask"], B, S, H, dfeats, cu=cu)
        if pack is not None and pack.Mp > pack.n:
            dfeats[pack.n:].zero_()    # filler rows: zero gradient
        g = self.trunk._buf("g_top", (M, H), F32)
        g_bf = self.trunk._buf("gbf_top", (M, H), BF16)
        nb = hip.layernorm_bwd_blocks(M)
        part = ws.get("ln_part_f", (2, nb, H), F32)
        hip.layernorm_bwd(s["hL"], M, H, self.trunk.lnf_w.w, s["mf"], s["rf"], g, dy_f32=dfeats, dx_bf16=g_bf, part=part,
                          drop_dx=self.trunk.top_drop(), drop_rows=rows)
        _ln_param_grads(part, nb, H, self.trunk.lnf_w.g, self.trunk.lnf_b.g)
        g0 = self.trunk.backward(g, g_bf)
        drop = s["drop"]
        hip.embed_bwd(g0, s["ids"], s["mask"], B, S, H, self.wte.g, self.wpe.g,
                      drop_e=drop(0, KIND_EMBD) if drop is not None else None, cu=cu)


# ------------------------------------------------------------------------------------------ NT-Xent (Stage 1 loss)
class NTXentEngine:
    """Reference ``ContrastiveLoss.forward`` (model.py:984-1000) on F.normalize'd embeddings
    (model.py:828-829), fused: the [N, N] similarity matrix is never materialised in f32 - each
    128x128 tile yields per-row (max, sum exp) partials (ROWSTATS epilogue) for the rows of S
    (image->text) and of S^t (text->image).

    With data-parallel global negatives, rank r owns rows ``[off, off+B)`` of S and of S^t: it
    scores its B images against all N texts and its B texts against all N images, so its local
    gradient is complete given the gathered embeddings and the gathered row/column log-sum-exps
    (2N floats) - no gradient collective.
    """

    def __init__(self, ws: Workspace, proj_dim: int, temperature: float, tag: str = "ntx"):
        self.ws, self.P, self.tau, self.tag = ws, proj_dim, float(temperature), tag
        if proj_dim % 8:
            raise ValueError("projection_dim must be a multiple of 8 (K alignment of the similarity GEMM)")
        self.saved = None
        self._tg_off, self._tgN_key, self._consts = None, None, {}

    def _buf(self, name, shape, dtype, zero=False):
        return self.ws.get(f"{self.tag}.{name}", shape, dtype, zero)

    def normalize(self, emb: torch.Tensor, which: str):
        B, P = emb.shape
        y, n = self._buf(which + ".y", (B, P), F32), self._buf(which + ".n", (B,), F32)
        hip.l2norm_fwd(emb, B, P, y, n)
        return y, n

    def _stats(self, a_bf, b_all_bf, B, N, targets, which):
        nparts = 2 * ((N + 127) // 128)
        smax, ssum = self._buf(which + ".smax", (B, nparts), F32), self._buf(which + ".ssum", (B, nparts), F32)
        tval, lse = self._buf(which + ".tval", (B,), F32), self._buf(which + ".lse", (B,), F32)
        hip.gemm(a_bf, b_all_bf, B, N, 3 * self.P, hip.NT, epilogue=hip.EPI_ROWSTATS, alpha=1.0 / self.tau,
                 targets=targets, stat_max=smax, stat_sum=ssum, stat_ld=nparts, target_val=tval)
        hip.rowstats_combine(smax, ssum, nparts, nparts, tval, B, lse=lse)
        return lse, tval

    def forward(self, img_n: torch.Tensor, txt_n: torch.Tensor, img_all: Optional[torch.Tensor] = None,
                txt_all: Optional[torch.Tensor] = None, offset: int = 0):
        """Normalised local embeddings [B,P] f32 (+ gathered [N,P] when data parallel).
        Returns (local loss contribution already divided by 2N, lse_r [B], lse_c [B]).

        The reference evaluates the similarity in fp32 (the loss runs outside autocast, model.py:988-990), and at
        tau = 0.07 a bf16-rounded cosine is a 3e-2 error on the logit.  The operands therefore go to the MFMA GEMM
        as hi/lo bf16 pairs concatenated along K (``pgca_split_bf16``): [hi|hi|lo] . [hi|lo|hi] over K = 3P sums
        hi.hi + hi.lo + lo.hi in f32 - the logits are good to ~1e-5 relative at 3x the (negligible) flops."""
        B, P = img_n.shape
        img_all = img_n if img_all is None else img_all
        txt_all = txt_n if txt_all is None else txt_all
        N = img_all.shape[0]
        bf = {}
        Np = (N + 7) // 8 * 8  # gathered tables are zero-padded to a multiple of 8 rows (K of the dgrad GEMMs)
        for k, t, rows, pattern in (("i", img_n, B, 0), ("t", txt_n, B, 0), ("ia", img_all, Np, 1), ("ta", txt_all, Np, 1)):
            bf[k] = self._buf("bf3." + k, (rows, 3 * P), BF16)
            hip.split_bf16(t, t.shape[0], P, rows, pattern, bf[k])
        tg = self.ws.bufs.get(self.tag + ".targets")
        if tg is None or tg.numel() != B or self._tg_off != offset:
            tg = (torch.arange(B, dtype=I64, device=self.ws.device) + offset).contiguous()
            self.ws.bufs[self.tag + ".targets"] = tg
            self._tg_off = offset
        lse_r, diag = self._stats(bf["i"], bf["ta"], B, N, tg, "r")
        lse_c, _ = self._stats(bf["t"], bf["ia"], B, N, tg, "c")
        loss = self._buf("loss", (1,), F32)
        hip.ntxent_loss(lse_r, lse_c, diag, B, N, loss)
        self.saved = dict(bf=bf, B=B, N=N, offset=offset, lse_r=lse_r, lse_c=lse_c, tg=tg)
        return loss, lse_r, lse_c

    def _const(self, name: str, n: int, value: float) -> torch.Tensor:
        """[n] f32 filled with ``value`` (row_scale operand of the DLOGITS epilogue); refilled only when it changes."""
        key = (n, float(value))
        t = self._buf(name, (n,), F32)
        if self._consts.get(name) != key:
            t.fill_(value)
            self._consts[name] = key
        return t

    def backward(self, lse_r_all: Optional[torch.Tensor] = None, lse_c_all: Optional[torch.Tensor] = None,
                 loss_scale: float = 1.0):
        """Returns (d loss / d img_n, d loss / d txt_n) [B,P] f32 for the LOCAL rows."""
        s, P = self.saved, self.P
        B, N, off, bf = s["B"], s["N"], s["offset"], s["bf"]
        lse_r_all = s["lse_r"] if lse_r_all is None else lse_r_all
        lse_c_all = s["lse_c"] if lse_c_all is None else lse_c_all
        c = loss_scale / (2.0 * N)
        inv_tau = 1.0 / self.tau
        Np = (N + 7) // 8 * 8
        Bp = (B + 7) // 8 * 8
        K3 = 3 * P
        cB = self._const("cB", B, c)
        cN = self._const("cN", N, c)
        tgN = self.ws.bufs.get(self.tag + ".targetsN")
        if tgN is None or tgN.numel() != N or (off, B) != self._tgN_key:
            j = torch.arange(N, dtype=I64, device=self.ws.device) - off
            tgN = torch.where((j >= 0) & (j < B), j, torch.full_like(j, -1)).contiguous()
            self.ws.bufs[self.tag + ".targetsN"] = tgN
            self._tgN_key = (off, B)
        # G1[i in loc, j in all] = c (p^r_ij - d_ij);  G2[j in loc, i in all] = c (p^c_ij - d_ij)
        # Each G leaves the DLOGITS epilogue as a hi + lo bf16 pair: at initialisation the embeddings of a batch are
        # nearly parallel, P is nearly uniform and G.T is a difference of nearly equal vectors - a G rounded to bf16
        # (2^-9) puts ~10 % noise on that difference (measured: head-gradient cosine 0.984), the pair keeps it at f32 level
        def G(name, shape):
            return self._buf(name, shape, BF16), self._buf(name + "_lo", shape, BF16)
        g1, g1l = G("g1", (B, Np))
        g2, g2l = G("g2", (B, Np))
        hip.gemm(bf["i"], bf["ta"], B, N, K3, hip.NT, epilogue=hip.EPI_DLOGITS, alpha=inv_tau, targets=s["tg"],
                 row_lse=s["lse_r"], row_scale=cB, out_bf16=g1, ld_out_bf16=Np, out_cols=Np, aux_out=g1l, ld_aux=Np)
        hip.gemm(bf["t"], bf["ia"], B, N, K3, hip.NT, epilogue=hip.EPI_DLOGITS, alpha=inv_tau, targets=s["tg"],
                 row_lse=s["lse_c"], row_scale=cB, out_bf16=g2, ld_out_bf16=Np, out_cols=Np, aux_out=g2l, ld_aux=Np)
        # G1'[i in all, j in loc] = c (p^r_ij - d_ij);  G2'[j in all, i in loc] = c (p^c_ij - d_ij)
        # (operand roles swapped: [hi|lo|hi] . [hi|hi|lo] is the same three-term sum)
        g1t, g1tl = G("g1t", (N, Bp))
        g2t, g2tl = G("g2t", (N, Bp))
        hip.gemm(bf["ia"], bf["t"], N, B, K3, hip.NT, epilogue=hip.EPI_DLOGITS, alpha=inv_tau, targets=tgN,
                 row_lse=lse_r_all, row_scale=cN, out_bf16=g1t, ld_out_bf16=Bp, out_cols=Bp, aux_out=g1tl, ld_aux=Bp)
        hip.gemm(bf["ta"], bf["i"], N, B, K3, hip.NT, epilogue=hip.EPI_DLOGITS, alpha=inv_tau, targets=tgN,
                 row_lse=lse_c_all, row_scale=cN, out_bf16=g2t, ld_out_bf16=Bp, out_cols=Bp, aux_out=g2tl, ld_aux=Bp)
        dI = self._buf("dI", (B, P), F32)
        dT = self._buf("dT", (B, P), F32)
        # dI_loc = (1/tau) (G1 T_all + G2'^t T_all);  dT_loc = (1/tau) (G2 I_all + G1'^t I_all); the gathered
        # tables enter as hi + lo (columns [0,P) and [P,2P) of the [hi|lo|hi] image, row stride 3P), the G's likewise
        first = True
        for part in (0, 1):
            ta, ia = bf["ta"][:, part * P:], bf["ia"][:, part * P:]
            for a1, a2t, b2, b1t in ((g1, g2t, g2, g1t), (g1l, g2tl, g2l, g1tl)):
                hip.gemm(a1, ta, B, P, Np, hip.NN, lda=Np, ldb=K3, alpha=inv_tau, out_f32=dI, accumulate=not first)
                hip.gemm(a2t, ta, B, P, N, hip.TN, lda=Bp, ldb=K3, alpha=inv_tau, out_f32=dI, accumulate=True)
                hip.gemm(b2, ia, B, P, Np, hip.NN, lda=Np, ldb=K3, alpha=inv_tau, out_f32=dT, accumulate=not first)
                hip.gemm(b1t, ia, B, P, N, hip.TN, lda=Bp, ldb=K3, alpha=inv_tau, out_f32=dT, accumulate=True)
                first = False
        return dI, dT
