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
* a micro-batch whose loss is not finite contributes no gradient (the reference skips its backward,
  trainer.py:481-489, 606-613); when it is the micro-batch that CLOSES an accumulation group the whole group is
  dropped without an optimiser / scheduler step (the reference's ``zero_grad()`` is only real there); non-finite
  gradients skip the step (trainer.py:494-508).  All of it is decided ON THE DEVICE (gradient seeds zeroed by
  ``pgca_dpo_loss``, ``gate`` + all-reduced norm in ``pgca_step_control``): every rank takes the same decision and the
  step loop never reads a loss back;
* ``global_step`` counts micro-batches (trainer.py:525, 633).

Differences (documented in DESIGN.md): no ``.item()`` per micro-batch (losses are read from the device every
``logging_steps``), so ``global_step`` also counts the skipped micro-batches; validation loss is averaged across
ranks; early stopping is decided identically on every rank; gradients are clipped once per optimiser step unless
``mi355x.clip_every_micro_step`` asks for the reference's per-micro-step clipping; ``load_checkpoint`` also restores
the optimiser / scheduler / dropout-stream state the reference forgets (trainer.py:836-853) and resumes after the
saved epoch.
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
from .input import BatchPrefetcher
from .model import PreferenceGuidedCaptioningModel
from .steps import ContrastiveStep, DPOStep, FusedOptimizer, ReferencePolicy


class PreferenceGuidedTrainer:
    def __init__(self, model: PreferenceGuidedCaptioningModel, config, train_loader_stage1, val_loader_stage1,
                 train_loader_stage2=None, val_loader_stage2=None, accelerator=None) -> None:
        self.model, self.config = model, config
        self.train_loader_stage1, self.val_loader_stage1 = train_loader_stage1, val_loader_stage1
        self.train_loader_stage2, self.val_loader_stage2 = train_loader_stage2, val_loader_stage2
        self.accelerator = accelerator  # accepted for signature compatibility; DP is handled by pgca_amd.dist
        self.dp = DataParallel(bucket_elems=int(config.get("mi355x.allreduce_bucket_elems", 64 * 1024 * 1024)),
                               compress_bf16=bool(config.get("mi355x.allreduce_bf16", False)))
        self.device = model.device
        self.logger = logging.getLogger(__name__)
        # one accumulation value for both stages, as the reference's single Accelerator
        self.accum = int(getattr(accelerator, "gradient_accumulation_steps", None)
                         or config.get("training.stage1.gradient_accumulation_steps", 4))
        self.temperature = float(config.get("model.temperature", 0.07))
        self.beta = float(config.get("training.stage2.dpo_beta", 0.1))
        self.current_stage, self.global_step, self.epoch = 1, 0, 0
        self.best_val_loss, self.patience_counter = float("inf"), 0
        self._resume: Optional[Dict[str, Any]] = None   # optimiser / scheduler / dropout state of a loaded checkpoint
        self.output_dir = Path(config.get("paths.output_dir", "./outputs"))
        self.checkpoint_dir = self.output_dir / "checkpoints"
        if self.is_main_process:
            self.checkpoint_dir.mkdir(parents=True, exist_ok=True)
        self.history: List[Dict[str, Any]] = []

    @property
    def is_main_process(self) -> bool:
        return self.dp.rank == 0

    def _dropout_plan(self, stage: int) -> DropoutPlan:
        """model.dropout (configs/default.yaml:22) at the reference's own train-mode sites and HF GPT-2's fixed 0.1 at
        its internal embd / attn / resid sites (``mi355x.gpt2_pdrop``); per-rank, per-stage seed."""
        seed = int(self.config.get("training.seed", 42)) * 1000 + stage * 100 + self.dp.rank
        plan = DropoutPlan(float(getattr(self.model, "dropout", 0.0) or 0.0), base_seed=seed,
                           p_gpt=float(self.config.get("mi355x.gpt2_pdrop", 0.1)))
        if self._resume is not None and self._resume.get("stage") == stage:
            plan.step = int(self._resume.get("dropout_step", 0))
        self._plan = plan
        return plan

    def _check_loader(self, loader, what: str) -> None:
        """Every rank must see the same number of batches: the loop's collectives are per batch."""
        n = len(loader)
        lo = self.dp.all_reduce_max_scalar(-float(n), self.device)
        hi = self.dp.all_reduce_max_scalar(float(n), self.device)
        if -lo != hi:
            raise RuntimeError(f"{what}: ranks hold between {int(-lo)} and {int(hi)} batches; shard the dataset evenly "
                               "(scripts/train.py truncates to a multiple of the world size)")

    # ------------------------------------------------------------------ optimiser
    def _setup_optimizer(self, stage: int, num_training_steps: int) -> FusedOptimizer:
        sc = self.config.get(f"training.stage{stage}")
        # the CLIP tower joins when it was left trainable (reference AdamW(model.parameters()), trainer.py:275-281)
        names = ("vit", "vision_head", "text_tower", "text_head") if stage == 1 else ("vit", "vision_head", "decoder")
        segs = [self.model.store.segments[n] for n in names if self.model.store.segments[n].trainable]
        opt = FusedOptimizer(segs, lr=sc["learning_rate"], weight_decay=sc.get("weight_decay", 0.01),
                             betas=(0.9, 0.999), eps=1e-8, max_grad_norm=sc.get("max_grad_norm"),
                             warmup_steps=sc.get("warmup_steps", 0), total_steps=num_training_steps,
                             sched_stride=self.dp.world)
        if self._resume is not None and self._resume.get("stage") == stage and self._resume.get("optimizer"):
            opt.load_state_dict({k: ([t.to(self.device) for t in v] if isinstance(v, list) else v.to(self.device))
                                 for k, v in self._resume["optimizer"].items()})
            self.logger.info(f"Restored optimiser / scheduler state of stage {stage} "
                             f"(step {opt.state()['step']}, schedule position {opt.state()['sched_step']})")
        return opt

    # ------------------------------------------------------------------ epoch loops
    def _feed(self, loader, prepare):
        """Prepared batches: pinned staging + H2D + device-side index preparation ``mi355x.prefetch_depth`` batches
        ahead on a side stream (input.BatchPrefetcher); depth 0 prepares inline."""
        depth = int(self.config.get("mi355x.prefetch_depth", 2))
        if depth > 0 and self.device.type == "cuda":
            return BatchPrefetcher(loader, prepare, self.device, depth)
        return (prepare(b, self.device) for b in loader)

    def _run_epoch(self, loader, opt: FusedOptimizer, prepare, micro_step, reducer: Optional[OverlappedTrunkReducer],
                   extra_segments, stage_cfg: Dict[str, Any], stage: int) -> float:
        self.model.train()
        n_batches = len(loader)
        log_every = int(stage_cfg.get("logging_steps", 100))
        clip_micro = bool(self.config.get("mi355x.clip_every_micro_step", False))
        loss_sum = torch.zeros(1, dtype=torch.float32, device=self.device)
        finite_cnt = torch.zeros(1, dtype=torch.float32, device=self.device)
        opt.zero_grad()
        for step, batch in enumerate(self._feed(loader, prepare)):
            boundary = ((step + 1) % self.accum == 0) or (step + 1 == n_batches)
            if reducer is not None:
                reducer.arm() if boundary else reducer.disarm()
            loss = micro_step(batch, 1.0 / self.accum)
            ok = torch.isfinite(loss).to(torch.float32)
            loss_sum += torch.nan_to_num(loss, nan=0.0, posinf=0.0, neginf=0.0) * ok
            finite_cnt += ok
            self.global_step += 1
            if clip_micro and not boundary:
                opt.clip_partial()
            if boundary:
                if reducer is not None:
                    reducer.finish(other_segments=extra_segments)
                else:
                    self.dp.all_reduce_grads(opt.segments)
                # a non-finite loss on the closing micro-batch drops the group (pgca_step_control gate); under data
                # parallelism the gate must be the same on every rank: a NaN anywhere makes the summed value NaN
                gate = loss if self.dp.world == 1 else self.dp.all_reduce_sum(loss.clone())
                opt.step(grad_scale=1.0 / self.dp.world, gate=gate)
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

    @torch.no_grad()
    def _validate(self, loader, prepare, loss_only) -> float:
        self.model.eval()
        tot = torch.zeros(2, dtype=torch.float32, device=self.device)
        for batch in self._feed(loader, prepare):
            loss = loss_only(batch).reshape(())
            ok = torch.isfinite(loss)                       # decided on the device: no read-back per batch
            tot[0] += torch.where(ok, loss, torch.zeros_like(loss))
            tot[1] += ok.to(tot.dtype)
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
                               dropout=self._dropout_plan(1), packed=bool(self.config.get("mi355x.packed_rows", True)))
        reducer = OverlappedTrunkReducer(self.dp, m.text_encoder.engine.trunk,
                                         group=int(self.config.get("mi355x.allreduce_layer_group", 4)))
        extra = [s for s in opt.segments if s is not reducer.seg]

        def micro(p, scale):
            return step.loss_and_grads(p["image"], p["ids"], p["mask"], loss_scale=scale, pack=p.get("pack"))

        def val(p):
            return step.loss_only(p["image"], p["ids"], p["mask"], pack=p.get("pack"))

        return self._train_loop(1, sc, opt, self.train_loader_stage1, self.val_loader_stage1, ContrastiveStep.prepare,
                                micro, val, reducer, extra)

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
                       dropout=self._dropout_plan(2), packed=bool(self.config.get("mi355x.packed_rows", True)))
        reducer = OverlappedTrunkReducer(self.dp, m.caption_decoder.engine.trunk,
                                         group=int(self.config.get("mi355x.allreduce_layer_group", 4)))
        extra = [s for s in opt.segments if s is not reducer.seg]

        def micro(p, scale):
            return step.loss_and_grads(p["image"], p["seq"], loss_scale=scale)

        def val(p):
            return step.loss_only(p["image"], p["seq"])

        return self._train_loop(2, sc, opt, self.train_loader_stage2,
                                self.val_loader_stage2 or self.train_loader_stage2, DPOStep.prepare, micro, val, reducer,
                                extra)

    def _train_loop(self, stage, sc, opt, train_loader, val_loader, prepare, micro, val, reducer, extra):
        metrics: Dict[str, List[float]] = {"train_loss": [], "val_loss": [], "learning_rates": []}
        self._check_loader(train_loader, f"stage {stage} training loader")
        self._check_loader(val_loader, f"stage {stage} validation loader")
        first = 0
        if self._resume is not None and self._resume.get("stage") == stage:
            first = int(self._resume.get("epoch", -1)) + 1   # the reference restarts at epoch 0 (trainer.py:848)
            self._resume = None
        for epoch in range(first, sc["num_epochs"]):
            self.epoch = epoch
            t0 = time.time()
            train_loss = self._run_epoch(train_loader, opt, prepare, micro, reducer, extra, sc, stage)
            val_loss = self._validate(val_loader, prepare, val)
            lr = opt.state()["lr"]
            metrics["train_loss"].append(train_loss)
            metrics["val_loss"].append(val_loss)
            metrics["learning_rates"].append(lr)
            self._log_metrics({"epoch": epoch, "stage": stage, "train_loss": train_loss, "val_loss": val_loss,
                               "learning_rate": lr, "epoch_seconds": time.time() - t0})
            # every rank runs the same bookkeeping on the same (rank-reduced) val_loss, so all of them leave the loop
            # together; only the file writes are rank 0's (the reference updates best_val_loss inside the rank-0-only
            # save and hangs the other ranks in the next collective)
            should_stop = self._check_early_stopping(val_loss, sc)
            improved = val_loss < self.best_val_loss
            self._save_checkpoint(epoch, opt, opt, val_loss, stage)      # rank 0 writes (and updates best_val_loss)
            if improved:
                self.best_val_loss = val_loss                            # ... every other rank follows
            if should_stop:
                self.logger.info(f"Early stopping triggered at epoch {epoch}")
                break
        self.logger.info(f"Completed Stage {stage} training")
        return metrics

    def train(self) -> Dict[str, Any]:
        """Reference ``train`` (trainer.py:855-884): both stages, then the same result dict.  As in the reference
        ``best_val_loss`` / ``patience_counter`` carry over from Stage 1 into Stage 2 (a Stage-2 checkpoint only counts as
        "best" below the best Stage-1 loss); ``mi355x.reset_best_val_between_stages: true`` starts Stage 2 afresh."""
        stage1_metrics = self.train_stage1()
        if bool(self.config.get("mi355x.reset_best_val_between_stages", False)):
            self.best_val_loss, self.patience_counter = float("inf"), 0
        stage2_metrics = self.train_stage2()
        results = {"stage1_metrics": stage1_metrics, "stage2_metrics": stage2_metrics,
                   "best_val_loss": self.best_val_loss, "total_steps": self.global_step}
        if self.is_main_process:
            self._log_metrics({"final_best_val_loss": self.best_val_loss, "total_training_steps": self.global_step})
        self.logger.info("Training completed successfully")
        return results

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

    def _save_checkpoint(self, epoch: int, optimizer, scheduler, val_loss: float, stage: int) -> None:
        """Reference ``_save_checkpoint`` (trainer.py:770-813): same signature, dict layout and file names (the key names of
        ``model_state_dict`` are the reference's, including the duplicated ViT and the tied lm_head); rank 0 only; a
        ``val_loss`` below ``best_val_loss`` updates it and also writes ``best_model_stage{stage}.pt``.  The fused optimiser
        is its own scheduler.  ``mi355x_state`` is extra: what a bit-faithful resume needs beyond the reference's keys (the
        reference's consumers ignore unknown keys)."""
        if not self.is_main_process:
            return

        def cpu(x):
            if isinstance(x, torch.Tensor):
                return x.detach().cpu()
            if isinstance(x, dict):
                return {k: cpu(v) for k, v in x.items()}
            if isinstance(x, (list, tuple)):
                return type(x)(cpu(v) for v in x)
            return x

        sched = ({"sched_step": scheduler.state()["sched_step"]} if isinstance(scheduler, FusedOptimizer)
                 else (scheduler.state_dict() if scheduler is not None else {}))
        ck = {"epoch": epoch, "stage": stage, "global_step": self.global_step,
              "model_state_dict": {k: v.detach().cpu().clone() for k, v in self.model.state_dict().items()},
              "optimizer_state_dict": cpu(optimizer.state_dict()), "scheduler_state_dict": cpu(sched),
              "val_loss": val_loss, "config": self.config.config,
              "mi355x_state": {"dropout_step": int(getattr(self, "_plan", DropoutPlan()).step),
                               "best_val_loss": min(self.best_val_loss, val_loss),
                               "patience_counter": self.patience_counter}}
        self.checkpoint_dir.mkdir(parents=True, exist_ok=True)
        path = self.checkpoint_dir / f"checkpoint_stage{stage}_epoch{epoch}.pt"
        torch.save(ck, path)
        if val_loss < self.best_val_loss:
            self.best_val_loss = val_loss
            torch.save(ck, self.checkpoint_dir / f"best_model_stage{stage}.pt")
            self.logger.info(f"Saved best model with val_loss: {val_loss:.4f}")
        self.logger.info(f"Saved checkpoint: {path}")

    def load_checkpoint(self, path: str) -> None:
        """Reference trainer.py:836-853 (model + epoch / global_step / stage / best_val_loss) plus what it forgets:
        the AdamW moments, the step and schedule counters and the dropout stream position are restored when the next
        ``train_stage{stage}`` builds its optimiser, and that stage continues after the saved epoch.  Loading is
        strict, as ``nn.Module.load_state_dict`` is: a checkpoint of another architecture raises."""
        ck = torch.load(path, map_location="cpu", weights_only=False)
        self.model.load_state_dict(ck["model_state_dict"], strict=True)
        self.model.sync_bf16()
        self.epoch, self.current_stage = ck.get("epoch", 0), ck.get("stage", 1)
        self.global_step = ck.get("global_step", 0)
        extra = ck.get("mi355x_state", {})
        self.best_val_loss = float(extra.get("best_val_loss", ck.get("val_loss", float("inf"))))
        self.patience_counter = int(extra.get("patience_counter", 0))
        self._resume = {"stage": self.current_stage, "epoch": self.epoch, "optimizer": ck.get("optimizer_state_dict"),
                        "dropout_step": extra.get("dropout_step", 0)}
        self.logger.info(f"Loaded checkpoint from {path}")
