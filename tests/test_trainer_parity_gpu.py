"""Trainer step semantics and resume (row N3): the HIP trainer's epoch loop against the oracle's
restatement of reference trainer.py:464-539,575-647 under accelerate's accumulate() - accumulation = 2 over 8
micro-batches, one of which holds a caption with a single real token (its length-mean log-prob is 0/0, so the
reference skips that batch) - once with the reference's per-micro-batch clipping and once with the path's default."""
import math
import os

import numpy as np
import pytest
import torch

from oracle import trainer_restatement as TR
from pgca_amd import REPO_ROOT

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


class ListLoader(list):
    """A loader is anything iterable with a length (the trainer only needs that)."""


def _pairs(arch, n, B, S, seed, nan_at=None):
    g = torch.Generator().manual_seed(seed)
    out = []
    for i in range(n):
        lens = torch.randint(4, S + 1, (2 * B,), generator=g)
        if i == nan_at:
            lens[1] = 1                                   # one caption with <= 1 real token: Σ mask[1:] = 0
        ids = torch.randint(0, arch.gpt.base_vocab, (2 * B, S), generator=g)
        mask = (torch.arange(S)[None] < lens[:, None]).long()
        out.append({"image": torch.randn(B, 3, arch.vit.image, arch.vit.image, generator=g),
                    "preferred_ids": ids[:B], "rejected_ids": ids[B:], "preferred_mask": mask[:B],
                    "rejected_mask": mask[B:]})
    return ListLoader(out)


def _config(tmp_path, accum, epochs, lr, clip_micro):
    from pgca_amd.config import Config
    cfg = Config(os.path.join(REPO_ROOT, "configs", "default.yaml"))
    cfg.set("paths.output_dir", str(tmp_path))
    for st in ("stage1", "stage2"):
        cfg.set(f"training.{st}.num_epochs", epochs)
        cfg.set(f"training.{st}.warmup_steps", 2)
        cfg.set(f"training.{st}.learning_rate", lr)
        cfg.set(f"training.{st}.gradient_accumulation_steps", accum)
        cfg.set(f"training.{st}.max_grad_norm", 1.0)
    cfg.set("mi355x.clip_every_micro_step", clip_micro)
    cfg.set("model.dropout", 0.0)
    cfg.set("mi355x.gpt2_pdrop", 0.0)        # the oracle comparison is deterministic arithmetic: dropout off
    return cfg


def _model(seed=21):
    from pgca_amd.arch import tiny_arch
    from pgca_amd.model import PreferenceGuidedCaptioningModel
    return PreferenceGuidedCaptioningModel(freeze_vision_backbone=True, arch=tiny_arch(), dropout=0.0, seed=seed, device=DEV)


def _delta_cos(a, b, a0):
    da, db = (a.double() - a0.double()).flatten(), (b.double() - a0.double()).flatten()
    return float(da @ db / (da.norm() * db.norm() + 1e-300))


@pytest.mark.parametrize("clip_micro,nan_at", [(True, 2), (False, 2), (True, 3), (False, None)])
def test_stage2_epoch_matches_reference_step_semantics(tmp_path, clip_micro, nan_at):
    """nan_at = 2: the NaN micro-batch opens an accumulation group (its partner still steps);
    nan_at = 3: it CLOSES a group, which the reference then drops without an optimiser / scheduler step."""
    from pgca_amd.trainer import PreferenceGuidedTrainer
    accum, epochs, lr, B, S, n = 2, 1, 2e-3, 2, 16, 8
    model = _model()
    arch = model.arch
    train = _pairs(arch, n, B, S, seed=5, nan_at=nan_at)
    val = _pairs(arch, 2, B, S, seed=6)
    cfg = _config(tmp_path, accum, epochs, lr, clip_micro)
    names = [k for seg in ("vision_head", "decoder") for k in model.store.segments[seg].index]
    sd = {k: v.detach().cpu().clone() for k, v in model.store.state_dict(aliases=False).items()}
    init = {k: sd[k].clone() for k in names}
    ref = TR.run_epochs(sd, names, list(train), TR.stage2_loss(arch, 0.1), stage=2, accum=accum, epochs=epochs, lr=lr,
                        warmup=2, total_steps=(n // accum) * epochs, max_norm=1.0, clip_every_micro_step=clip_micro)
    tr = PreferenceGuidedTrainer(model, cfg, train, val, train, val)
    out = tr.train_stage2()
    # bookkeeping: which groups stepped
    hist = [h for h in tr.history if "epoch" in h]
    expect_steps = n // accum - (1 if nan_at == 3 else 0)
    assert ref["opt_steps"] == expect_steps
    seg = model.store.segments["decoder"]
    ck = torch.load(tmp_path / "checkpoints" / "checkpoint_stage2_epoch0.pt", map_location="cpu", weights_only=False)
    ctrl = ck["optimizer_state_dict"]["ctrl"]
    assert int(ctrl[6]) == ref["opt_steps"] and int(ctrl[7]) == ref["sched_step"]
    # epoch mean of the finite micro-batch losses (what the reference returns, trainer.py:544,652)
    assert abs(out["train_loss"][0] - ref["epoch_means"][0]) <= 5e-3, (out["train_loss"], ref["epoch_means"])
    assert len(hist) == 1
    if nan_at is not None:
        assert not math.isfinite(ref["losses"][nan_at])
    # parameters: same update direction everywhere it matters, same magnitude
    for k in names:
        a, b = model.store.w(k).detach().cpu(), sd[k].detach()
        d_ref = float((b.double() - init[k].double()).norm())
        if d_ref < 1e-7 * max(1.0, float(init[k].norm())):        # q/k rows of the collapsed cross-attention etc.
            continue
        c = _delta_cos(a, b, init[k])
        assert c >= 0.90, f"{k}: update cosine {c}"
        assert abs(float((a.double() - init[k].double()).norm()) / d_ref - 1) <= 0.1, k
    # the big tensors individually tighter
    wte = "caption_decoder.lm_model.transformer.wte.weight"
    assert _delta_cos(model.store.w(wte).cpu(), sd[wte].detach(), init[wte]) >= 0.97
    del seg


def test_nan_microbatch_contributes_nothing(tmp_path):
    """The skipped micro-batch must leave no trace: a run with the NaN batch == a run where that batch is replaced by
    another NaN batch with different (finite-part) content.  Bitwise, gradients included."""
    from pgca_amd.trainer import PreferenceGuidedTrainer
    res = []
    for seed_nan in (100, 200):
        model = _model()
        arch = model.arch
        train = _pairs(arch, 4, 2, 16, seed=5)
        bad = _pairs(arch, 1, 2, 16, seed=seed_nan, nan_at=0)[0]
        train[0] = bad
        cfg = _config(tmp_path, 2, 1, 1e-3, False)
        tr = PreferenceGuidedTrainer(model, cfg, train, _pairs(arch, 1, 2, 16, seed=6), train, _pairs(arch, 1, 2, 16, seed=6))
        tr.train_stage2()
        res.append(model.store.segments["decoder"].fp32.clone())
    assert torch.equal(res[0], res[1])


def test_resume_restores_optimizer_scheduler_and_continues(tmp_path):
    """train 2 epochs (4 optimiser steps)  ==  train 1 epoch, save, NEW model + trainer, load, train the 2nd epoch.
    AdamW moments, step / schedule counters, best_val_loss and the dropout stream position all travel in the
    checkpoint (the reference restores the model and counters only, trainer.py:836-853).  Train-mode dropout is ON."""
    from pgca_amd.config import Config
    from pgca_amd.trainer import PreferenceGuidedTrainer

    def cfg_for(d, epochs):
        c = _config(d, 2, epochs, 1e-3, False)
        c.set("model.dropout", 0.1)
        c.set("mi355x.gpt2_pdrop", 0.1)
        return c

    def fresh():
        from pgca_amd.arch import tiny_arch
        from pgca_amd.model import PreferenceGuidedCaptioningModel
        return PreferenceGuidedCaptioningModel(freeze_vision_backbone=True, arch=tiny_arch(), dropout=0.1, seed=21, device=DEV)

    m1 = fresh()
    arch = m1.arch
    train, val = _pairs(arch, 4, 2, 16, seed=5), _pairs(arch, 2, 2, 16, seed=6)
    d1 = tmp_path / "straight"
    t1 = PreferenceGuidedTrainer(m1, cfg_for(d1, 2), train, val, train, val)
    o1 = t1.train_stage2()
    # interrupted run: same 2-epoch schedule (total steps), stopped after epoch 0 by loading its checkpoint elsewhere
    m2 = fresh()
    d2 = tmp_path / "resumed"
    c2 = cfg_for(d2, 2)
    t2 = PreferenceGuidedTrainer(m2, c2, train, val, train, val)
    t2.load_checkpoint(str(d1 / "checkpoints" / "checkpoint_stage2_epoch0.pt"))
    assert t2.best_val_loss == pytest.approx(o1["val_loss"][0])
    o2 = t2.train_stage2()
    assert len(o2["train_loss"]) == 1                                 # only epoch 1 ran
    assert o2["train_loss"][0] == pytest.approx(o1["train_loss"][1], abs=2e-6)
    assert o2["learning_rates"][0] == pytest.approx(o1["learning_rates"][1], rel=1e-6)
    for name in ("vision_head", "decoder"):
        a, b = m1.store.segments[name], m2.store.segments[name]
        # embedding / LM-head gradients are summed with f32 atomics (order not fixed), everything else is bitwise
        np.testing.assert_allclose(b.fp32.cpu().numpy(), a.fp32.cpu().numpy(), rtol=0, atol=2e-6)
        np.testing.assert_allclose(b.exp_avg.cpu().numpy(), a.exp_avg.cpu().numpy(), rtol=0, atol=2e-6)
    ck1 = torch.load(d1 / "checkpoints" / "checkpoint_stage2_epoch1.pt", map_location="cpu", weights_only=False)
    ck2 = torch.load(d2 / "checkpoints" / "checkpoint_stage2_epoch1.pt", map_location="cpu", weights_only=False)
    assert torch.equal(ck1["optimizer_state_dict"]["ctrl"][6:], ck2["optimizer_state_dict"]["ctrl"][6:])
    assert ck1["mi355x_state"]["dropout_step"] == ck2["mi355x_state"]["dropout_step"] == 8
    # strict loading: a checkpoint with a missing tensor raises
    bad = dict(ck1["model_state_dict"])
    bad.pop("caption_decoder.attention_norm.weight")
    with pytest.raises(KeyError):
        m2.load_state_dict(bad)


def test_stage1_with_frozen_text_tower_trains_the_heads_only(tmp_path):
    """``freeze_text_backbone=True`` (reference model.py:354-368): Stage 1 leaves the GPT-2 tower untouched and trains the
    two projection heads; their gradients equal the oracle's with the towers treated as constants."""
    from oracle import restatement as R
    from pgca_amd.arch import tiny_arch
    from pgca_amd.model import PreferenceGuidedCaptioningModel
    from pgca_amd.steps import ContrastiveStep
    from pgca_amd.trainer import PreferenceGuidedTrainer
    arch = tiny_arch()
    model = PreferenceGuidedCaptioningModel(freeze_vision_backbone=True, freeze_text_backbone=True, arch=arch, dropout=0.0,
                                            seed=1, device=DEV)
    assert model.store.segments["text_tower"].grad is None
    g = torch.Generator().manual_seed(9)
    img = torch.randn(4, 3, arch.vit.image, arch.vit.image, generator=g)
    ids = torch.randint(0, arch.gpt.base_vocab, (4, 16), generator=g)
    mask = (torch.arange(16)[None] < torch.tensor([16, 7, 11, 3])[:, None]).long()
    step = ContrastiveStep(model.store, model.ws, model.vision_encoder.tower, model.vision_encoder.head,
                           model.text_encoder.engine, temperature=0.5)
    p = ContrastiveStep.prepare({"image": img, "caption_ids": ids, "caption_mask": mask}, model.device)
    for s_ in model.store.trainable_segments():
        s_.grad.zero_()
    loss = float(step.loss_and_grads(p["image"], p["ids"], p["mask"]))
    sd = {k: v.detach().cpu().clone().requires_grad_(".projection." in k)
          for k, v in model.store.state_dict(aliases=False).items()}
    ie = R.vision_encoder_forward(sd, img, arch.vit.heads, arch.vit.patch)["embeddings"]
    te = R.text_encoder_forward(sd, ids, mask, arch.gpt.heads)["embeddings"]
    n = torch.nn.functional.normalize
    ref = R.nt_xent(n(ie, dim=-1), n(te, dim=-1), 0.5)
    ref.backward()
    assert abs(loss - float(ref)) <= 5e-3
    for name in ("text_encoder.projection.0.weight", "text_encoder.projection.3.weight", "text_encoder.projection.4.weight",
                 "vision_encoder.projection.0.weight"):
        a, b = model.store.g(name).double().flatten().cpu(), sd[name].grad.double().flatten()
        assert float((a @ b) / (a.norm() * b.norm())) >= 0.99, name
    # the trainer runs Stage 1 on such a model and moves the heads only
    before = {k: s_.fp32.clone() for k, s_ in model.store.segments.items()}
    batches = [{"image": img, "caption_ids": ids, "caption_mask": mask}] * 2
    tr = PreferenceGuidedTrainer(model, _config(tmp_path, 1, 1, 1e-3, False), ListLoader(batches), ListLoader(batches))
    tr.train_stage1()
    assert torch.equal(model.store.segments["text_tower"].fp32, before["text_tower"])
    assert torch.equal(model.store.segments["vit"].fp32, before["vit"])
    assert not torch.equal(model.store.segments["text_head"].fp32, before["text_head"])
    assert not torch.equal(model.store.segments["vision_head"].fp32, before["vision_head"])
