"""TEST INFRASTRUCTURE - CPU restatement of the reference trainer's inner step loops
(reference training/trainer.py:464-539 Stage 1, :575-647 Stage 2) as they behave under the
``accelerate.Accelerator`` the reference wraps them in (accelerate 1.14.0):

* ``accumulate()`` makes micro-batch ``step`` a sync boundary when ``(step + 1) % accum == 0`` or it is the last
  batch of the loader (accelerator.py:1229-1236); ``optimizer.step / zero_grad`` and ``scheduler.step`` are no-ops
  off the boundary; ``backward`` divides the loss by ``accum`` (accelerator.py:2840);
* a non-finite loss skips the micro-batch's backward, and the ``zero_grad()`` the reference then calls is only real on
  a boundary - there it drops the whole accumulated group (no optimiser / scheduler step);
* Stage 1 additionally checks every gradient for non-finite values (trainer.py:494-508) with the same consequence;
* ``clip_grad_norm_`` has no sync check (accelerator.py:2946-3007), so the reference clips the PARTIALLY accumulated
  gradient after every micro-batch; ``clip_every_micro_step=False`` restates the MI355X path's documented default
  (one clip per optimiser step) instead;
* the cosine schedule advances ``world`` positions per optimiser step (scheduler.py:54-82).

Only ``tests/`` import this module.  Parity pinning: the model / loss arithmetic underneath is
``oracle/restatement.py``, pinned against the imported reference by ``tests/golden``; the LOOP is pinned against a run of
the reference's own ``_train_epoch_stage1`` / ``_train_epoch_stage2`` under a real ``accelerate.Accelerator``
(``tests/golden/trainer_epoch.npz``, written by ``oracle/make_trainer_golden.py``: accumulation 2, a NaN micro-batch
opening and one closing a group, a trailing partial group, two epochs) by ``tests/test_trainer_oracle_cpu.py``.
"""
from __future__ import annotations

import math
from typing import Callable, Dict, List, Sequence

import torch

from . import restatement as R


def run_epochs(sd: Dict[str, torch.Tensor], train: Sequence[str], batches: List[dict],
               loss_fn: Callable[[Dict[str, torch.Tensor], dict], torch.Tensor], *, stage: int, accum: int,
               epochs: int, lr: float, warmup: int, total_steps: int, max_norm: float, weight_decay: float = 0.01,
               clip_every_micro_step: bool = True, world: int = 1) -> dict:
    """Runs ``epochs`` passes over ``batches`` updating ``sd[n] for n in train`` in place.
    Returns the per-micro-batch losses, optimiser / scheduler counters and the epoch means the reference logs."""
    for n in train:
        sd[n].requires_grad_(True)
    params = [sd[n] for n in train]
    grads = [torch.zeros_like(p) for p in params]
    m = [torch.zeros_like(p) for p in params]
    v = [torch.zeros_like(p) for p in params]
    opt_step, sched_step, global_step = 0, 0, 0
    losses, epoch_means, lrs, lr_in_force = [], [], [], []

    def zero():
        for g in grads:
            g.zero_()

    for _ in range(epochs):
        total, count = 0.0, 0
        zero()                                                   # optimizer.zero_grad() (trainer.py:452,564)
        for step, b in enumerate(batches):
            boundary = ((step + 1) % accum == 0) or (step + 1 == len(batches))
            loss = loss_fn(sd, b)
            lv = float(loss.detach())
            losses.append(lv)
            # what optimizer.param_groups[0]["lr"] reads during this micro-batch: the schedule at the current position
            lr_in_force.append(R.cosine_warmup_lr(lr, sched_step, warmup, total_steps))
            if not math.isfinite(lv):                            # trainer.py:481-489 / 606-613
                if boundary:
                    zero()
                continue
            gs = torch.autograd.grad(loss / accum, params, allow_unused=True)
            for g, dg in zip(grads, gs):
                if dg is not None:
                    g.add_(dg)
            if stage == 1 and not all(bool(torch.isfinite(g).all()) for g in grads):   # trainer.py:494-508
                if boundary:
                    zero()
                continue
            if max_norm and (clip_every_micro_step or boundary):  # trainer.py:511-515 / 619-623
                norm = math.sqrt(sum(float((g.double() ** 2).sum()) for g in grads))
                c = R.clip_coefficient(norm, max_norm)
                if c < 1.0:
                    for g in grads:
                        g.mul_(c)
            if boundary:                                         # optimizer.step(); scheduler.step(); zero_grad()
                opt_step += 1
                lr_t = R.cosine_warmup_lr(lr, sched_step, warmup, total_steps)
                lrs.append(lr_t)
                with torch.no_grad():
                    for p, g, mm, vv in zip(params, grads, m, v):
                        R.adamw_step(p, g, mm, vv, opt_step, lr_t, wd=weight_decay)
                sched_step += world
                zero()
            total += lv
            count += 1
            global_step += 1
        epoch_means.append(total / count if count else 0.0)
    return {"losses": losses, "epoch_means": epoch_means, "opt_steps": opt_step, "sched_step": sched_step,
            "global_step": global_step, "lrs": lrs, "lr_in_force": lr_in_force, "exp_avg": dict(zip(train, m)), "exp_avg_sq": dict(zip(train, v))}


def stage2_loss(arch, beta: float):
    """The trainer's Stage-2 loss (trainer.py:578-603): two generation forwards + ``PreferenceLoss``."""
    def f(sd, b):
        kw = (arch.vit.heads, arch.vit.patch, arch.gpt.heads)
        lw = R.model_forward(sd, b["image"], b["preferred_ids"], b["preferred_mask"], "generation", *kw)["logits"]
        ll = R.model_forward(sd, b["image"], b["rejected_ids"], b["rejected_mask"], "generation", *kw)["logits"]
        return R.preference_loss(lw, ll, b["preferred_ids"], b["rejected_ids"], b["preferred_mask"],
                                 b["rejected_mask"], beta)
    return f


def stage1_loss(arch, tau: float):
    """The trainer's Stage-1 loss (trainer.py:467-478): contrastive forward + ``ContrastiveLoss``."""
    def f(sd, b):
        o = R.model_forward(sd, b["image"], b["caption_ids"], b["caption_mask"], "contrastive", arch.vit.heads,
                            arch.vit.patch, arch.gpt.heads)
        return R.nt_xent(o["image_embeddings"], o["text_embeddings"], tau)
    return f
