"""The oracle's restatement of the reference trainer's step loop (oracle/trainer_restatement.py) on a toy problem:
its accumulate() / NaN-skip / clip / schedule semantics are checked against closed-form expectations and against
torch.optim.AdamW + transformers' cosine schedule wired as reference trainer.py:275-289 does."""
import math

import torch

from oracle import trainer_restatement as TR


def _toy(nan_at=None, n=6):
    batches = [{"x": torch.tensor([float(i + 1)]), "nan": i == nan_at} for i in range(n)]

    def loss_fn(sd, b):
        base = (sd["w"] * b["x"]).sum() ** 2 + sd["b"].sum()
        return base * float("nan") if b["nan"] else base
    return batches, loss_fn


def _run(nan_at, clip, accum=2, n=6):
    sd = {"w": torch.tensor([0.5, -0.25]), "b": torch.tensor([0.1])}
    batches, loss_fn = _toy(nan_at, n)
    out = TR.run_epochs(sd, ["w", "b"], batches, loss_fn, stage=2, accum=accum, epochs=1, lr=1e-2, warmup=1,
                        total_steps=n // accum, max_norm=1.0, clip_every_micro_step=clip)
    return sd, out


def test_boundaries_and_counters():
    _, o = _run(None, True)
    assert o["opt_steps"] == 3 and o["sched_step"] == 3 and o["global_step"] == 6
    _, o = _run(None, True, accum=4)            # trailing partial group still steps (last batch of the loader)
    assert o["opt_steps"] == 2


def test_nan_off_boundary_keeps_the_partner_and_on_boundary_drops_the_group():
    sd_a, a = _run(0, True)                      # NaN opens group 0: batch 1 alone is stepped
    assert a["opt_steps"] == 3 and a["global_step"] == 5 and math.isnan(a["losses"][0])
    sd_b, b = _run(1, True)                      # NaN closes group 0: the group is dropped, no step, no schedule advance
    assert b["opt_steps"] == 2 and b["sched_step"] == 2 and b["global_step"] == 5
    assert not torch.equal(sd_a["w"], sd_b["w"])


def test_matches_torch_adamw_with_per_micro_batch_clipping():
    from transformers import get_cosine_schedule_with_warmup
    sd, o = _run(None, True)
    w = torch.nn.Parameter(torch.tensor([0.5, -0.25]))
    b = torch.nn.Parameter(torch.tensor([0.1]))
    opt = torch.optim.AdamW([w, b], lr=1e-2, weight_decay=0.01, betas=(0.9, 0.999), eps=1e-8)
    sch = get_cosine_schedule_with_warmup(opt, num_warmup_steps=1, num_training_steps=3)
    batches, _ = _toy()
    for step, bt in enumerate(batches):
        loss = ((w * bt["x"]).sum() ** 2 + b.sum()) / 2          # accelerator.backward divides by accum
        loss.backward()
        torch.nn.utils.clip_grad_norm_([w, b], 1.0)               # every micro-batch (the reference quirk)
        if (step + 1) % 2 == 0:
            opt.step()
            sch.step()
            opt.zero_grad()
    assert torch.allclose(sd["w"].detach(), w.detach(), atol=1e-7) and torch.allclose(sd["b"].detach(), b.detach(), atol=1e-7)
    _, o2 = _run(None, False)                    # one clip per optimiser step: different first moments (AdamW's update is
    differs = any(not torch.allclose(o2["exp_avg"][k], o["exp_avg"][k], rtol=1e-4) for k in ("w", "b"))   # scale-free,
    assert differs                                # so compare the moments, not the weights
