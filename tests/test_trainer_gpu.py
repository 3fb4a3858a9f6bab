"""Reference-surface objects on the GPU: loss classes vs the reference's outputs, trainer loops, CLI dry run."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from pgca_amd import REPO_ROOT

pytestmark = pytest.mark.gpu
T = torch.from_numpy
DEV = "cuda:0"


def test_loss_classes_match_reference_outputs(golden):
    from pgca_amd.losses import ContrastiveLoss, DPOPreferenceLoss, PreferenceLoss, compute_sequence_logprobs
    g = golden("nt_xent")
    for b in (2, 8, 64):
        for tau in (0.07, 0.5):
            k = f"b{b}_t{tau}"
            loss = ContrastiveLoss(temperature=tau)(T(g[k + "_img"]).to(DEV), T(g[k + "_txt"]).to(DEV))
            # hi/lo split bf16 operands (f32-grade logits): well inside the stated 5e-3 at both temperatures
            assert abs(float(loss) - float(g[k + "_loss"])) <= 2e-4, k
    g = golden("logprob_dpo")
    lw, ll = T(g["logits_w"]).to(DEV), T(g["logits_l"]).to(DEV)
    iw, il, mw, ml = (T(g[k]).to(DEV) for k in ("ids_w", "ids_l", "mask_w", "mask_l"))
    np.testing.assert_allclose(compute_sequence_logprobs(lw, iw, mw).cpu().numpy(), g["seq_sum_w"], atol=1e-4)
    np.testing.assert_allclose(compute_sequence_logprobs(lw, iw, None).cpu().numpy(), g["seq_sum_w_nomask"], atol=1e-4)
    pl = PreferenceLoss(beta=0.1)
    np.testing.assert_allclose(pl._compute_log_probs(ll, il, ml).cpu().numpy(), g["seq_mean_l"], atol=1e-5)
    assert abs(float(pl(lw, ll, iw, il, mw, ml)) - float(g["pref_loss"])) <= 1e-5
    pc, pr, rc, rr = (T(g[k]).to(DEV) for k in ("dpo_pc", "dpo_pr", "dpo_rc", "dpo_rr"))
    for name, kw in (("std", {}), ("ls", {"label_smoothing": 0.1}), ("rf", {"reference_free": True})):
        loss, metrics = DPOPreferenceLoss(beta=0.1, **kw)(pc, pr, rc, rr)
        assert abs(float(loss) - float(g[f"dpo_{name}_loss"])) <= 1e-5
        ref = json.loads(str(g[f"dpo_{name}_metrics"]))
        assert set(metrics) == set(ref)
        for k, v in ref.items():
            assert abs(metrics[k] - v) <= 1e-4, (name, k)


class _DS(torch.utils.data.Dataset):
    def __init__(self, n, arch, S, pairs, seed):
        g = torch.Generator().manual_seed(seed)
        self.pairs = pairs
        self.img = torch.randn(n, 3, arch.vit.image, arch.vit.image, generator=g)
        k = 2 if pairs else 1
        ids = torch.randint(0, arch.gpt.base_vocab, (n, k, S), generator=g)
        lens = torch.randint(4, S + 1, (n, k), generator=g)
        self.mask = (torch.arange(S)[None, None] < lens[..., None]).long()
        self.ids = torch.where(self.mask.bool(), ids, torch.full_like(ids, arch.gpt.base_vocab))

    def __len__(self):
        return self.img.shape[0]

    def __getitem__(self, i):
        if self.pairs:
            return {"image": self.img[i], "preferred_ids": self.ids[i, 0], "preferred_mask": self.mask[i, 0],
                    "rejected_ids": self.ids[i, 1], "rejected_mask": self.mask[i, 1]}
        return {"image": self.img[i], "caption_ids": self.ids[i, 0], "caption_mask": self.mask[i, 0]}


def test_trainer_two_stages_tiny(tmp_path):
    from torch.utils.data import DataLoader
    from pgca_amd.arch import tiny_arch
    from pgca_amd.config import Config
    from pgca_amd.model import PreferenceGuidedCaptioningModel
    from pgca_amd.trainer import PreferenceGuidedTrainer
    cfg = Config(os.path.join(REPO_ROOT, "configs", "default.yaml"))
    cfg.set("paths.output_dir", str(tmp_path))
    for st in ("stage1", "stage2"):
        cfg.set(f"training.{st}.num_epochs", 2)
        cfg.set(f"training.{st}.warmup_steps", 1)
        cfg.set(f"training.{st}.learning_rate", 1e-3)
        cfg.set(f"training.{st}.gradient_accumulation_steps", 2)
    cfg.set("mi355x.gpt2_pdrop", 0.0)     # a "loss must fall in 6 steps" property: no dropout noise (its parity: test_e2e_gpu)
    arch = tiny_arch()
    model = PreferenceGuidedCaptioningModel(freeze_vision_backbone=True, arch=arch, dropout=0.0, seed=1, device=DEV)
    before = {n: s.fp32.clone() for n, s in model.store.segments.items()}
    mk = lambda pairs, seed, n: DataLoader(_DS(n, arch, 16, pairs, seed), batch_size=4, shuffle=False, drop_last=True)  # noqa
    tr = PreferenceGuidedTrainer(model, cfg, mk(False, 1, 20), mk(False, 2, 8), mk(True, 3, 20), mk(True, 4, 8))
    assert tr.accum == 2
    out = tr.train()
    assert set(out) == {"stage1_metrics", "stage2_metrics", "best_val_loss", "total_steps"}    # reference trainer.py:869-874
    assert out["total_steps"] == tr.global_step and out["best_val_loss"] == tr.best_val_loss
    s1, s2 = out["stage1_metrics"], out["stage2_metrics"]
    assert len(s1["train_loss"]) == 2 and len(s2["train_loss"]) == 2
    assert all(np.isfinite(s1["train_loss"] + s1["val_loss"] + s2["train_loss"] + s2["val_loss"]))
    assert s1["train_loss"][1] < s1["train_loss"][0]            # same 20 samples twice: NT-Xent must fall
    assert s2["train_loss"][1] <= s2["train_loss"][0] + 1e-3
    assert tr.global_step == 2 * 5 + 2 * 5                       # micro-batches, as the reference counts them
    # stage 1 trains text tower + heads; stage 2 trains decoder + vision head; the ViT never moves
    assert torch.equal(model.store.segments["vit"].fp32, before["vit"])
    for n in ("text_tower", "text_head", "vision_head", "decoder"):
        assert not torch.equal(model.store.segments[n].fp32, before[n]), n
    ck = torch.load(tmp_path / "checkpoints" / "checkpoint_stage2_epoch1.pt", map_location="cpu", weights_only=False)
    assert {"epoch", "stage", "global_step", "model_state_dict", "optimizer_state_dict", "scheduler_state_dict",
            "val_loss", "config"} <= set(ck)
    assert "caption_decoder.lm_model.lm_head.weight" in ck["model_state_dict"]
    assert "vision_encoder.clip_model.vision_model.post_layernorm.weight" in ck["model_state_dict"]
    assert (tmp_path / "checkpoints" / "best_model_stage1.pt").exists()
    tr.load_checkpoint(str(tmp_path / "checkpoints" / "checkpoint_stage2_epoch1.pt"))
    assert tr.global_step == ck["global_step"]


def test_train_cli_dry_run():
    env = dict(os.environ, PGCA_CFG_MODEL__TEXT_MODEL="gpt2", PYTHONPATH=REPO_ROOT)
    p = subprocess.run([sys.executable, os.path.join(REPO_ROOT, "scripts", "train.py"), "--dry-run", "--stage", "2",
                        "--synthetic-samples", "32"], capture_output=True, text=True, env=env, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    assert "Dry run completed successfully" in p.stderr
