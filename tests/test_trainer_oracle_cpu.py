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


# ------------------------------------------------------------------------------------------------------------------
# Pinned against the reference trainer itself: tests/golden/trainer_epoch.npz holds a run of the reference's own
# _train_epoch_stage2 / _train_epoch_stage1 (training/trainer.py:575-647, :464-539) under a real accelerate.Accelerator
# (oracle/make_trainer_golden.py).  The restatement must reproduce it micro-batch by micro-batch.
import json  # noqa: E402

import numpy as np  # noqa: E402
import pytest  # noqa: E402


@pytest.mark.parametrize("stage", [2, 1])
def test_loop_restatement_reproduces_the_reference_trainer(golden, stage):
    from pgca_amd.arch import tiny_arch
    from pgca_amd.params import ParamStore
    g = golden("trainer_epoch")
    meta = json.loads(str(g["meta"]))
    arch = tiny_arch()
    store = ParamStore(arch, "cpu", seed=int(meta["seed"]), frozen=())
    sd = {k: v.clone() for k, v in store.state_dict(aliases=False).items()}
    pre = f"s{stage}_"
    n = int(g[pre + "n_batches"])
    keys = ("image", "caption_ids", "caption_mask") if stage == 1 else \
        ("image", "preferred_ids", "preferred_mask", "rejected_ids", "rejected_mask")
    batches = [{k: torch.from_numpy(g[f"{pre}batch{i}_{k}"]) for k in keys} for i in range(n)]
    trained = json.loads(str(g[pre + "trained_names"]))
    trained = [t for t in trained if t in sd]                  # (aliases of the tower keys are the same tensors)
    loss_fn = TR.stage1_loss(arch, meta["tau"]) if stage == 1 else TR.stage2_loss(arch, meta["beta"])
    out = TR.run_epochs(sd, trained, batches, loss_fn, stage=stage, accum=meta["accum"], epochs=meta["epochs"],
                        lr=meta["lr"], warmup=meta["warmup"], total_steps=(n // meta["accum"]) * meta["epochs"],
                        max_norm=meta["max_norm"], clip_every_micro_step=True)
    want = g[pre + "losses"]
    got = np.array(out["losses"])
    assert np.array_equal(np.isnan(got), np.isnan(want)), (got, want)          # the NaN micro-batches, in place
    ok = ~np.isnan(want)
    # every later loss depends on every earlier update: a wrong boundary / skip / clip / schedule rule shows up here
    np.testing.assert_allclose(got[ok], want[ok], rtol=0, atol=2e-5)
    np.testing.assert_allclose(np.array(out["lr_in_force"]), g[pre + "lrs"], rtol=1e-6, atol=1e-12)
    np.testing.assert_allclose(np.array(out["epoch_means"]), g[pre + "epoch_means"], atol=2e-5)
    assert out["global_step"] == int(g[pre + "global_step"])
    assert out["opt_steps"] == int(g[pre + "opt_steps"])
    for name in trained:
        t = sd[name].detach().double().flatten()
        s_want = g["s%d_final_sum::%s" % (stage, name)]
        assert abs(float(t.sum()) - s_want[0]) <= 1e-4 * max(1.0, s_want[1]), name
        np.testing.assert_allclose(t[::max(1, t.numel() // 64)][:64].numpy(), g["s%d_final_sample::%s" % (stage, name)],
                                   atol=2e-5, err_msg=name)
