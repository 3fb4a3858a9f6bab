"""One training step of each stage on the HIP engines, and the fused optimiser.

* ``DPOStep``         Stage 2 (reference trainer.py:575-628): ViT once per image, policy forward on
                      chosen+rejected batched along M, optional frozen reference policy (4-forward DPO,
                      components.py:192-249) or the trainer's reference-free 2-forward PreferenceLoss
                      (model.py:1038-1050), backward through the policy only.
* ``ContrastiveStep`` Stage 1 (trainer.py:464-520): frozen ViT + head, GPT-2 text tower + head,
                      F.normalize, NT-Xent, backward.
* ``FusedOptimizer``  AdamW + global-norm clip + cosine warm-up (trainer.py:275-289,511-520) with the
                      whole decision chain on the device: no ``.item()`` in the step.
"""
from __future__ import annotations

from typing import Dict, Iterable, List, Optional

import torch

from . import hip
from .engine import (BF16, F32, I32, I64, KIND_HEAD, TOWER_DECODER, TOWER_TEXT, TOWER_THEAD, TOWER_VHEAD,
                     CaptionDecoderEngine, DropoutPlan, NTXentEngine, ProjHead, RowPack, SeqBatch, TextTowerEngine,
                     VisionTower, Workspace, make_row_pack, make_seq_batch)
from .params import ParamStore, Segment


class FusedOptimizer:
    """AdamW over flat segments with device-side clip / NaN-skip / LR schedule.

    ``ctrl`` (f32[8] on the device) carries total_norm, finite flag, clip factor, lr, bias
    corrections and the optimiser / scheduler step counters between kernels (pgca_hip.h).
    ``grad_scale`` folds the data-parallel mean (1/world) and the accumulation mean (1/accum)
    into the same pass that clips.
    """

    def __init__(self, segments: Iterable[Segment], lr: float, weight_decay: float = 0.01, betas=(0.9, 0.999),
                 eps: float = 1e-8, max_grad_norm: Optional[float] = 1.0, warmup_steps: int = 0,
                 total_steps: int = 1, sched_stride: int = 1):
        self.segments: List[Segment] = list(segments)
        for s in self.segments:
            s.ensure_train_state()
            if s.bf16 is None:
                s.ensure_bf16()
        self.lr, self.wd, self.betas, self.eps = float(lr), float(weight_decay), betas, float(eps)
        self.max_norm = float(max_grad_norm) if max_grad_norm else 0.0
        self.warmup, self.total, self.stride = int(warmup_steps), max(1, int(total_steps)), int(sched_stride)
        dev = self.segments[0].device
        self.ctrl = torch.zeros(8, dtype=F32, device=dev)
        self.nblocks = [hip.sqnorm_blocks(s.numel) for s in self.segments]
        self.part = torch.zeros(sum(self.nblocks), dtype=F32, device=dev)
        self.coef = torch.ones(2, dtype=F32, device=dev)   # [clip factor, norm] of clip_partial()

    def zero_grad(self) -> None:
        for s in self.segments:
            s.grad.zero_()

    def _sqnorms(self) -> None:
        off = 0
        for s, nb in zip(self.segments, self.nblocks):
            hip.sqnorm(s.grad, s.numel, self.part[off:off + nb])
            off += nb

    def clip_partial(self) -> None:
        """The reference's ``clip_grad_norm_`` on a PARTIALLY accumulated (local) gradient - it clips after every
        micro-batch (trainer.py:511-515,619-623).  Norm, factor and scaling stay on the device."""
        if self.max_norm <= 0:
            return
        self._sqnorms()
        hip.clip_coef(self.part, self.part.numel(), self.max_norm, self.coef)
        for s in self.segments:
            hip.scale_dev(s.grad, s.numel, self.coef)

    def step(self, grad_scale: float = 1.0, gate: Optional[torch.Tensor] = None) -> None:
        """``gate``: 1-element device tensor (the loss of the micro-batch closing the accumulation group); a
        non-finite value skips the step on the device, as the reference drops that group (pgca_step_control)."""
        self._sqnorms()
        hip.step_control(self.part, self.part.numel(), self.max_norm, self.lr, self.warmup, self.total, self.stride,
                         self.betas[0], self.betas[1], grad_scale, self.ctrl, gate=gate)
        for s in self.segments:
            hip.adamw(s.fp32, s.grad, s.exp_avg, s.exp_avg_sq, s.bf16, s.numel, self.ctrl, self.wd, self.betas[0],
                      self.betas[1], self.eps, grad_scale)

    def state(self) -> Dict[str, float]:
        """One D2H copy (for logging only)."""
        c = self.ctrl.tolist()
        return dict(grad_norm=c[0], finite=bool(c[1]), clip=c[2], lr=c[3], step=int(c[6]), sched_step=int(c[7]))

    def state_dict(self) -> dict:
        return {"ctrl": self.ctrl.clone(), "exp_avg": [s.exp_avg.clone() for s in self.segments],
                "exp_avg_sq": [s.exp_avg_sq.clone() for s in self.segments]}

    def load_state_dict(self, sd: dict) -> None:
        self.ctrl.copy_(sd["ctrl"])
        for s, m, v in zip(self.segments, sd["exp_avg"], sd["exp_avg_sq"]):
            s.exp_avg.copy_(m)
            s.exp_avg_sq.copy_(v)


class ReferencePolicy:
    """pi_ref of DPO: a frozen snapshot of every module on the generation path (vision head,
    vision_projection, cross_attention, attention_norm, lm_model) taken at Stage-2 start
    (SURVEY 8a row A7).  The frozen ViT output is shared with the policy."""

    def __init__(self, policy_store: ParamStore, ws: Workspace):
        arch = policy_store.arch
        self.store = ParamStore(arch, policy_store.device, seed=None, frozen=("vision_head", "decoder"),
                                segments=("vision_head", "decoder"))
        self.refresh(policy_store)
        self.head = ProjHead(self.store, "vision_encoder.projection", arch.vit.hidden, arch.proj_dim, ws, "ref.vhead")
        self.dec = CaptionDecoderEngine(self.store, arch, ws, "ref")

    def refresh(self, policy_store: ParamStore) -> None:
        for name, seg in self.store.segments.items():
            seg.fp32.copy_(policy_store.segments[name].fp32)
            seg.ensure_bf16()


class DPOStep:
    """Stage-2 micro-step: loss + gradients (accumulated into the flat gradient buffers)."""

    def __init__(self, store: ParamStore, ws: Workspace, vit: VisionTower, vhead: ProjHead,
                 dec: CaptionDecoderEngine, beta: float = 0.1, reference_free: bool = False,
                 label_smoothing: float = 0.0, reduce: Optional[str] = None, ref: Optional[ReferencePolicy] = None,
                 ref_side_stream: bool = False, dropout: Optional[DropoutPlan] = None, packed: bool = True):
        self.store, self.ws, self.vit, self.vhead, self.dec = store, ws, vit, vhead, dec
        self.beta, self.reference_free, self.ls = float(beta), bool(reference_free), float(label_smoothing)
        # trainer parity: 2-forward == PreferenceLoss (length-mean); 4-forward == DPOPreferenceLoss (length-sum)
        self.reduce = reduce or ("mean" if reference_free else "sum")
        self.ref = ref
        if not self.reference_free and ref is None:
            raise ValueError("4-forward DPO needs a ReferencePolicy (or pass reference_free=True)")
        dev = ws.device
        self.loss = torch.zeros(1, dtype=F32, device=dev)
        self.metrics = torch.zeros(4, dtype=F32, device=dev)
        self._ref_stream = None
        self.ref_side_stream = bool(ref_side_stream)
        # train-mode dropout of the policy (the frozen reference policy always runs without dropout)
        self.dropout = dropout if dropout is not None else DropoutPlan(0.0)
        # both decoder trunks (policy and reference) run on the rows of the real tokens only (engine.RowPack); False
        # computes every padded position like the reference does (same scored log-probs, loss and gradients)
        self.packed = bool(packed)

    @staticmethod
    def prepare(batch: dict, device) -> dict:
        """Collate-time preparation of one reference batch dict (loader.py:487-497 contract):
        chosen and rejected are stacked along the sequence-batch dimension."""
        ids = torch.cat([batch["preferred_ids"], batch["rejected_ids"]], dim=0)
        mask = torch.cat([batch["preferred_mask"], batch["rejected_mask"]], dim=0)
        if not ids.is_cuda and batch["preferred_ids"].is_pinned():   # keep the H2D copies asynchronous
            ids, mask = ids.pin_memory(), mask.pin_memory()
        image = batch["image"].to(device, F32, non_blocking=True)
        return {"image": image, "seq": make_seq_batch(ids, mask, device)}

    def forward(self, images: torch.Tensor, sb: SeqBatch, save: bool = True):
        B = images.shape[0]
        assert sb.Bq == 2 * B
        P = self.store.arch.proj_dim
        _, _, pooled_bf = self.vit.forward(images, save)   # keeps activations only when the tower is trainable
        plan = self.dropout
        plan.active = bool(save)          # save == training forward; evaluation forwards run without dropout
        emb = self.vhead.forward(pooled_bf, B, save, plan.site(TOWER_VHEAD, 0, KIND_HEAD))
        emb2 = self.ws.get("dpo.emb2", (2 * B, P), F32)
        emb2[:B].copy_(emb)
        emb2[B:].copy_(emb)
        ref_lp = None
        packed = self.packed and sb.pack is not None and sb.pack.Mp > 0
        if not self.reference_free:
            # the frozen reference policy runs on its own HIP stream, concurrently with the policy forward:
            # the two kernel sequences are independent, so one's store-bound epilogues and tile tails are
            # filled by the other's MFMA main loops
            main = torch.cuda.current_stream()
            if not self.ref_side_stream:
                self._ref_stream = main
            elif self._ref_stream is None or self._ref_stream is main:
                self._ref_stream = torch.cuda.Stream()
            if self._ref_stream is not main:
                self._ref_stream.wait_stream(main)
            with torch.cuda.stream(self._ref_stream):
                remb = self.ref.head.forward(pooled_bf, B, False)
                remb2 = self.ws.get("dpo.remb2", (2 * B, P), F32)
                remb2[:B].copy_(remb)
                remb2[B:].copy_(remb)
                ref_lp = self.ref.dec.sequence_logprobs(remb2, sb, self.reduce, False, packed=packed)
        pol = self.dec.sequence_logprobs(emb2, sb, self.reduce, save, plan.bind(TOWER_DECODER), packed=packed)
        if ref_lp is not None and self._ref_stream is not torch.cuda.current_stream():
            torch.cuda.current_stream().wait_stream(self._ref_stream)
        return pol, ref_lp

    def loss_and_grads(self, images: torch.Tensor, sb: SeqBatch, loss_scale: float = 1.0) -> torch.Tensor:
        """Returns the (unscaled) loss as a 1-element device tensor; gradients of ``loss_scale * loss``
        are accumulated (loss_scale = 1/accumulation_steps, accelerate/accelerator.py:2840)."""
        B = images.shape[0]
        pol, ref_lp = self.forward(images, sb, True)
        self.dropout.step += 1             # next micro-step draws fresh masks
        dseq = self.ws.get("dpo.dseq", (2 * B,), F32)
        hip.dpo_loss(pol[:B], pol[B:], None if ref_lp is None else ref_lp[:B], None if ref_lp is None else ref_lp[B:],
                     B, self.beta, self.ls, self.loss, dseq[:B], dseq[B:], self.metrics)
        if loss_scale != 1.0:
            dseq.mul_(loss_scale)
        demb2 = self.dec.backward(dseq)
        demb = self.ws.get("dpo.demb", (B, self.store.arch.proj_dim), F32)
        torch.add(demb2[:B], demb2[B:], out=demb)
        dpooled = self.vhead.backward(demb, need_dx=self.vit.trainable)
        if self.vit.trainable:            # freeze_vision_backbone=False (reference model.py:150-164)
            self.vit.backward(dpooled)
        return self.loss

    @torch.no_grad()
    def loss_only(self, images: torch.Tensor, sb: SeqBatch) -> torch.Tensor:
        B = images.shape[0]
        pol, ref_lp = self.forward(images, sb, False)
        hip.dpo_loss(pol[:B], pol[B:], None if ref_lp is None else ref_lp[:B], None if ref_lp is None else ref_lp[B:],
                     B, self.beta, self.ls, self.loss, None, None, self.metrics)
        return self.loss


class ContrastiveStep:
    """Stage-1 micro-step.  ``dp`` (optional) provides ``all_gather_rows`` / ``all_gather_vec`` /
    ``rank`` / ``world`` for global negatives; without it negatives are local (as the reference)."""

    def __init__(self, store: ParamStore, ws: Workspace, vit: VisionTower, vhead: ProjHead, text: TextTowerEngine,
                 temperature: float, dp=None, global_negatives: bool = False, dropout: Optional[DropoutPlan] = None,
                 packed: bool = True):
        self.store, self.ws, self.vit, self.vhead, self.text = store, ws, vit, vhead, text
        self.packed = bool(packed)   # text tower on the real tokens' rows only (engine.RowPack)
        self.dropout = dropout if dropout is not None else DropoutPlan(0.0)
        self.ntx = NTXentEngine(ws, store.arch.proj_dim, temperature)
        self.dp = dp if (dp is not None and global_negatives and dp.world > 1) else None

    _processors: dict = {}

    @staticmethod
    def _device_images(batch: dict, device) -> torch.Tensor:
        """A loader that hands over DECODED images (``image`` uint8 [B, H, W, 3]) gets the reference's image transform on
        the device (input.GpuImageProcessor, bit-exact with the host path): ``augment`` true -> the training transform
        with the random augmentations (data/preprocessing.py:52-70; draws from ``augment_params`` or the torch global RNG,
        as torchvision in a loader worker), otherwise the validation transform (:44-48).  ``image_size`` defaults to 224."""
        from .input import GpuImageProcessor
        size = int(batch.get("image_size", 224))
        key = (str(device), size)
        proc = ContrastiveStep._processors.get(key)
        if proc is None:
            proc = ContrastiveStep._processors[key] = GpuImageProcessor(size, device=device)
        if batch.get("augment", False):
            return proc.process_train_batch(batch["image"], batch.get("augment_params"))
        return proc.process_batch(batch["image"])

    @staticmethod
    def prepare(batch: dict, device) -> dict:
        mask = (batch["caption_mask"] != 0).to(I32).to(device, non_blocking=True).contiguous()
        if batch["image"].dtype == torch.uint8:
            image = ContrastiveStep._device_images(batch, device)
        else:
            image = batch["image"].to(device, F32, non_blocking=True)
        return {"image": image,
                "ids": batch["caption_ids"].to(device, I64, non_blocking=True),
                "mask": mask, "pack": make_row_pack(mask)}

    def _pack(self, mask, pack: Optional[RowPack]) -> Optional[RowPack]:
        if not self.packed:
            return None
        pack = pack if pack is not None else make_row_pack(mask.contiguous())   # (one host wait; prepare() avoids it)
        return pack if pack.Mp > 0 else None     # a batch of empty captions: nothing to pack, the padded path handles it

    def forward(self, images, ids, mask, save: bool = True, pack: Optional[RowPack] = None):
        """``pack`` (``prepare()['pack']``): packed row layout of ``mask``; the text tower then computes the real tokens
        only and ``text_features`` are its packed rows (the training steps do not read them)."""
        B = images.shape[0]
        feats_v, _, pooled_bf = self.vit.forward(images, save)
        plan = self.dropout
        plan.active = bool(save)
        iemb = self.vhead.forward(pooled_bf, B, save, plan.site(TOWER_VHEAD, 0, KIND_HEAD))
        feats_t, _, temb = self.text.forward(ids, mask, save, plan.bind(TOWER_TEXT), plan.site(TOWER_THEAD, 0, KIND_HEAD),
                                             pack=pack)
        img_n, in_norm = self.ntx.normalize(iemb, "i")
        txt_n, tn_norm = self.ntx.normalize(temb, "t")
        return dict(image_embeddings=img_n, text_embeddings=txt_n, vision_features=feats_v, text_features=feats_t,
                    _in=in_norm, _tn=tn_norm)

    def loss_and_grads(self, images, ids, mask, loss_scale: float = 1.0, pack: Optional[RowPack] = None) -> torch.Tensor:
        B = images.shape[0]
        o = self.forward(images, ids, mask, True, self._pack(mask, pack))
        self.dropout.step += 1
        img_n, txt_n = o["image_embeddings"], o["text_embeddings"]
        if self.dp is None:
            loss, _, _ = self.ntx.forward(img_n, txt_n)
            dI, dT = self.ntx.backward(loss_scale=loss_scale)
        else:
            ia, ta = self.dp.all_gather_rows(img_n), self.dp.all_gather_rows(txt_n)
            loss, lr, lc = self.ntx.forward(img_n, txt_n, ia, ta, offset=self.dp.rank * B)
            lra, lca = self.dp.all_gather_vec(lr), self.dp.all_gather_vec(lc)
            # every rank holds 1/world of the summed loss; DP averaging of gradients divides by world,
            # so scale local gradients by world to get d(global loss)/d(local embeddings)
            dI, dT = self.ntx.backward(lra, lca, loss_scale=loss_scale * self.dp.world)
            loss = self.dp.all_reduce_sum(loss)
        P = self.store.arch.proj_dim
        di = self.ws.get("s1.di", (B, P), F32)
        dt = self.ws.get("s1.dt", (B, P), F32)
        hip.l2norm_bwd(dI, img_n, o["_in"], B, P, di)
        hip.l2norm_bwd(dT, txt_n, o["_tn"], B, P, dt)
        dpooled = self.vhead.backward(di, need_dx=self.vit.trainable)
        if self.vit.trainable:
            self.vit.backward(dpooled)
        self.text.backward(dt)
        return loss

    @torch.no_grad()
    def loss_only(self, images, ids, mask, pack: Optional[RowPack] = None) -> torch.Tensor:
        o = self.forward(images, ids, mask, False, self._pack(mask, pack))
        loss, _, _ = self.ntx.forward(o["image_embeddings"], o["text_embeddings"])
        return loss
