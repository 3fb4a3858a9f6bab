"""Two-stage training loop with the reference trainer's semantics on the HIP step engines.

Mirrors ``PreferenceGuidedTrainer`` (reference training/trainer.py:159-222, 296-433, 435-652): same
constructor, ``train_stage1 / train_stage2 / train / load_checkpoint``, same config keys, same step
semantics:

* loss scaled by 1/accumulation (accelerate/accelerator.py:2840), optimizer step on every
  ``accum``-th micro-batch or the last batch of the loader; Stage-1's accumulation value governs both
  stages because the reference builds ONE Accelerator from it (scripts/train.py:317-322);
* AdamW(lr, wd=0.01, betas=(0.9, 0.999), eps=1e-8) on every parameter that receives gradients,
  cosine schedule with warm-up, total steps = len(loader) // accum * epochs, scheduler advanced
  ``world`` times per optimiser step (accelerate/scheduler.py:54-82);
* non-finite loss / gradients skip the update (trainer.py:481-508, 606-613) - decided ON THE DEVICE
  from the all-reduced global norm, so every rank takes the same decision without a host sync;
* ``global_step`` counts micro-batches (trainer.py:525, 633).

Differences (documented in DESIGN.md): no ``.item()`` per micro-batch (losses are read from the device every
``logging_steps``), validation loss is averaged across ranks, gradients are clipped once per optimiser step
unless ``mi355x.clip_every_micro_step`` asks for the reference's per-micro-step clipping.
"""
from __future__ import annotations

import logging
import math
import time
from pathlib import Path
from typing import Any, Dict, Iterable, List, Optional

import torch

from .dist import DataParallel, OverlappedTrunkReducer
from .engine import DropoutPlan
from .model import PreferenceGuidedCaptioningModel
from .steps import ContrastiveStep, DPOStep, FusedOptimizer, ReferencePolicy


class PreferenceGuidedTrainer:
    def __init__(self, model: PreferenceGuidedCaptioningModel, config, train_loader_stage1, val_loader_stage1,
                 train_loader_stage2=None, val_loader_stage2=None, accelerator=None) -> None:
        self.model, self.config = model, config
        self.train_loader_stage1, self.val_loader_stage1 = train_loader_stage1, val_loader_stage1
        self.train_loader_stage2, self.val_loader_stage2 = train_loader_stage2, val_loader_stage2
        self.accelerator = accelerator  # accepted for signature compatibility; DP is handled by pgca_amd.dist
        self.dp = DataParallel(bucket_elems=int(config.get("mi355x.allreduce_bucket_elems", 64 * 1024 * 1024)))
        self.device = model.device
        self.logger = logging.getLogger(__name__)
        # one accumulation value for both stages, as the reference's single Accelerator
        self.accum = int(getattr(accelerator, "gradient_accumulation_steps", None)
                         or config.get("training.stage1.gradient_accumulation_steps", 4))
        self.temperature = float(config.get("model.temperature", 0.07))
        self.beta = float(config.get("training.stage2.dpo_beta", 0.1))
        self.current_stage, self.global_step, self.epoch = 1, 0, 0
        self.best_val_loss, self.patience_counter = float("inf"), 0
        self.output_dir = Path(config.get("paths.output_dir", "./outputs"))
        self.checkpoint_dir = self.output_dir / "checkpoints"
        if self.is_main_process:
            self.checkpoint_dir.mkdir(parents=True, exist_ok=True)
        self.history: List[Dict[str, Any]] = []

    @property
    def is_main_process(self) -> bool:
        return self.dp.rank == 0

    def _dropout_plan(self, stage: int) -> DropoutPlan:
        """model.dropout (configs/default.yaml:22) at the reference's train-mode sites; per-rank, per-stage seed."""
        seed = int(self.config.get("training.seed", 42)) * 1000 + stage * 100 + self.dp.rank
        return DropoutPlan(float(getattr(self.model, "dropout", 0.0) or 0.0), base_seed=seed)

    # ------------------------------------------------------------------ optimiser
    def _setup_optimizer(self, stage: int, num_training_steps: int) -> FusedOptimizer:
        sc = self.config.get(f"training.stage{stage}")
        names = ("vision_head", "text_tower", "text_head") if stage == 1 else ("vision_head", "decoder")
        segs = [self.model.store.segments[n] for n in names if self.model.store.segments[n].trainable]
        return FusedOptimizer(segs, lr=sc["learning_rate"], weight_decay=sc.get("weight_decay", 0.01),
                              betas=(0.9, 0.999), eps=1e-8, max_grad_norm=sc.get("max_grad_norm"),
                              warmup_steps=sc.get("warmup_steps", 0), total_steps=num_training_steps,
                              sched_stride=self.dp.world)

    # ------------------------------------------------------------------ epoch loops
    def _run_epoch(self, loader, opt: FusedOptimizer, micro_step, reducer: Optional[OverlappedTrunkReducer],
                   extra_segments, stage_cfg: Dict[str, Any], stage: int) -> float:
        self.model.train()
        n_batches = len(loader)
        log_every = int(stage_cfg.get("logging_steps", 100))
        clip_micro = bool(self.config.get("mi355x.clip_every_micro_step", False))
        loss_sum = torch.zeros(1, dtype=torch.float32, device=self.device)
        finite_cnt = torch.zeros(1, dtype=torch.float32, device=self.device)
        opt.zero_grad()
        for step, batch in enumerate(loader):
            boundary = ((step + 1) % self.accum == 0) or (step + 1 == n_batches)
            if reducer is not None:
                reducer.arm() if boundary else reducer.disarm()
            loss = micro_step(batch, 1.0 / self.accum)
            ok = torch.isfinite(loss).to(torch.float32)
            loss_sum += torch.nan_to_num(loss, nan=0.0, posinf=0.0, neginf=0.0) * ok
            finite_cnt += ok
            self.global_step += 1
            if clip_micro and not boundary and opt.max_norm > 0:
                self._clip_partial(opt)
            if boundary:
                if reducer is not None:
                    reducer.finish(other_segments=extra_segments)
                else:
                    self.dp.all_reduce_grads(opt.segments)
                opt.step(grad_scale=1.0 / self.dp.world)
                opt.zero_grad()
            if self.global_step % log_every == 0 and self.is_main_process:
                st = opt.state()
                self._log_metrics({"step_loss": float(loss), "learning_rate": st["lr"], "global_step": self.global_step,
                                   "grad_norm": st["grad_norm"], "stage": stage})
        n = float(finite_cnt)
        skipped = n_batches - int(n)
        if skipped:
            self.logger.warning(f"Epoch had {skipped} NaN batches out of {n_batches} total")
        return float(loss_sum) / n if n > 0 else 0.0

    def _clip_partial(self, opt: FusedOptimizer) -> None:
        """Reference quirk: ``clip_grad_norm_`` runs on every micro-step (trainer.py:511-515,619-623), i.e. on
        the partially accumulated gradient."""
        norm = math.sqrt(sum(float((s.grad.double() ** 2).sum()) for s in opt.segments))
        c = min(1.0, opt.max_norm / (norm + 1e-6))
        if c < 1.0:
            for s in opt.segments:
                s.grad.mul_(c)

    @torch.no_grad()
    def _validate(self, loader, loss_only) -> float:
        self.model.eval()
        tot = torch.zeros(2, dtype=torch.float32, device=self.device)
        for batch in loader:
            loss = loss_only(batch)
            if bool(torch.isfinite(loss)):
                tot[0] += loss.reshape(())
                tot[1] += 1
        self.dp.all_reduce_sum(tot)  # the reference logs rank 0's value only; here every rank agrees
        return float(tot[0] / tot[1]) if float(tot[1]) > 0 else float("inf")

    # ------------------------------------------------------------------ stages
    def train_stage1(self) -> Dict[str, List[float]]:
        self.logger.info("Starting Stage 1: Contrastive Learning")
        self.current_stage = 1
        sc = self.config.get_stage1_config()
        steps_per_epoch = len(self.train_loader_stage1) // sc.get("gradient_accumulation_steps", 1)
        opt = self._setup_optimizer(1, steps_per_epoch * sc["num_epochs"])
        m = self.model
        step = ContrastiveStep(m.store, m.ws, m.vision_encoder.tower, m.vision_encoder.head, m.text_encoder.engine,
                               self.temperature, dp=self.dp,
                               global_negatives=bool(self.config.get("mi355x.stage1.global_negatives", False)),
                               dropout=self._dropout_plan(1))
        reducer = OverlappedTrunkReducer(self.dp, m.text_encoder.engine.trunk,
                                         group=int(self.config.get("mi355x.allreduce_layer_group", 4)))
        extra = [s for s in opt.segments if s is not reducer.seg]

        def micro(batch, scale):
            p = ContrastiveStep.prepare(batch, self.device)
            return step.loss_and_grads(p["image"], p["ids"], p["mask"], loss_scale=scale)

        def val(batch):
            p = ContrastiveStep.prepare(batch, self.device)
            return step.loss_only(p["image"], p["ids"], p["mask"])

        return self._train_loop(1, sc, opt, self.train_loader_stage1, self.val_loader_stage1, micro, val, reducer, extra)

    def train_stage2(self) -> Dict[str, List[float]]:
        if self.train_loader_stage2 is None:
            self.logger.warning("No stage 2 data loader provided, skipping stage 2")
            return {}
        self.logger.info("Starting Stage 2: Preference Optimization")
        self.current_stage = 2
        sc = self.config.get_stage2_config()
        steps_per_epoch = len(self.train_loader_stage2) // sc.get("gradient_accumulation_steps", 1)
        opt = self._setup_optimizer(2, steps_per_epoch * sc["num_epochs"])
        m = self.model
        reference_free = bool(self.config.get("mi355x.dpo.reference_free", True))
        ref = None if reference_free else ReferencePolicy(m.store, m.ws)
        step = DPOStep(m.store, m.ws, m.vision_encoder.tower, m.vision_encoder.head, m.caption_decoder.engine,
                       beta=self.beta, reference_free=reference_free,
                       label_smoothing=float(self.config.get("mi355x.dpo.label_smoothing", 0.0)), ref=ref,
                       dropout=self._dropout_plan(2))
        reducer = OverlappedTrunkReducer(self.dp, m.caption_decoder.engine.trunk,
                                         group=int(self.config.get("mi355x.allreduce_layer_group", 4)))
        extra = [s for s in opt.segments if s is not reducer.seg]

        def micro(batch, scale):
            p = DPOStep.prepare(batch, self.device)
            return step.loss_and_grads(p["image"], p["seq"], loss_scale=scale)

        def val(batch):
            p = DPOStep.prepare(batch, self.device)
            return step.loss_only(p["image"], p["seq"])

        return self._train_loop(2, sc, opt, self.train_loader_stage2,
                                self.val_loader_stage2 or self.train_loader_stage2, micro, val, reducer, extra)

    def _train_loop(self, stage, sc, opt, train_loader, val_loader, micro, val, reducer, extra):
        metrics: Dict[str, List[float]] = {"train_loss": [], "val_loss": [], "learning_rates": []}
        for epoch in range(sc["num_epochs"]):
            self.epoch = epoch
            t0 = time.time()
            train_loss = self._run_epoch(train_loader, opt, micro, reducer, extra, sc, stage)
            val_loss = self._validate(val_loader, val)
            lr = opt.state()["lr"]
            metrics["train_loss"].append(train_loss)
            metrics["val_loss"].append(val_loss)
            metrics["learning_rates"].append(lr)
            self._log_metrics({"epoch": epoch, "stage": stage, "train_loss": train_loss, "val_loss": val_loss,
                               "learning_rate": lr, "epoch_seconds": time.time() - t0})
            should_stop = self._check_early_stopping(val_loss, sc)
            if self.is_main_process:
                self._save_checkpoint(epoch, opt, val_loss, stage)
            if should_stop:
                self.logger.info(f"Early stopping triggered at epoch {epoch}")
                break
        self.logger.info(f"Completed Stage {stage} training")
        return metrics

    def train(self) -> Dict[str, Any]:
        out = {"stage1": self.train_stage1()}
        if self.train_loader_stage2 is not None:
            self.best_val_loss, self.patience_counter = float("inf"), 0
            out["stage2"] = self.train_stage2()
        return out

    # ------------------------------------------------------------------ bookkeeping
    def _log_metrics(self, m: Dict[str, Any]) -> None:
        self.history.append(m)
        if self.is_main_process:
            self.logger.info(" ".join(f"{k}={v:.6g}" if isinstance(v, float) else f"{k}={v}" for k, v in m.items()))

    def _check_early_stopping(self, val_loss: float, sc: Dict[str, Any]) -> bool:
        patience = sc.get("early_stopping_patience", 3)  # read from the stage dict like the reference (trainer.py:825)
        if val_loss < self.best_val_loss:
            self.patience_counter = 0
            return False
        self.patience_counter += 1
        return self.patience_counter >= patience

    def _save_checkpoint(self, epoch: int, opt: FusedOptimizer, val_loss: float, stage: int) -> None:
        """Same dict layout and file names as reference trainer.py:770-813 (key names of ``model_state_dict``
        are the reference's, including the duplicated ViT and the tied lm_head)."""
        ck = {"epoch": epoch, "stage": stage, "global_step": self.global_step,
              "model_state_dict": {k: v.detach().cpu().clone() for k, v in self.model.state_dict().items()},
              "optimizer_state_dict": {k: ([t.cpu() for t in v] if isinstance(v, list) else v.cpu())
                                       for k, v in opt.state_dict().items()},
              "scheduler_state_dict": {"sched_step": opt.state()["sched_step"]},
              "val_loss": val_loss, "config": self.config.config}
        torch.save(ck, self.checkpoint_dir / f"checkpoint_stage{stage}_epoch{epoch}.pt")
        if val_loss < self.best_val_loss:
            self.best_val_loss = val_loss
            torch.save(ck, self.checkpoint_dir / f"best_model_stage{stage}.pt")

    def load_checkpoint(self, path: str) -> None:
        ck = torch.load(path, map_location="cpu", weights_only=False)
        self.model.load_state_dict(ck["model_state_dict"])
        self.model.sync_bf16()
        self.epoch, self.current_stage = ck.get("epoch", 0), ck.get("stage", 1)
        self.global_step = ck.get("global_step", 0)
        self.logger.info(f"Loaded checkpoint from {path}")
