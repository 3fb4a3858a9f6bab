"""Generate ``tests/golden/trainer_epoch.npz`` from the REFERENCE TRAINER itself.

Build container only (needs ``/root/reference`` and ``accelerate``; nothing here travels to the GPU box except the
``.npz``).  Imports the reference's ``training/trainer.py`` (behind stub ``mlflow`` / ``peft`` modules: neither is
installed, neither is used by the step loops), builds the reference model on the tiny geometry with OUR seeded weights
(``make_golden.build_reference_model``), wraps it in a real ``accelerate.Accelerator`` on CPU with
``gradient_accumulation_steps=2`` and runs the reference's own ``_train_epoch_stage2`` / ``_train_epoch_stage1``
(reference training/trainer.py:575-647, :464-539) over a fixed list of micro-batches that contains a NaN micro-batch
OPENING an accumulation group, one CLOSING a group, and an odd batch count (a trailing partial group).

Recorded per stage: the micro-batches, per-micro-batch loss values (forward hook on the reference loss module), the
learning rate in force at each micro-batch, the epoch means the trainer returns, ``global_step``, the optimiser's step
count, and every trainable parameter after the epochs.  ``tests/test_trainer_oracle_cpu.py`` holds
``oracle/trainer_restatement.py`` to these numbers; ``tests/test_trainer_parity_gpu.py`` compares the MI355X trainer with
that restatement, so the pin carries through.

Dropout: every dropout probability of the reference model is set to 0 for this run (torch's Philox stream cannot be
replayed by a restatement); ``model.train()`` is still what the reference calls.

    python oracle/make_trainer_golden.py
"""
from __future__ import annotations

import importlib.util
import json
import os
import sys
import tempfile
import types

import numpy as np
import torch
import torch.nn as nn

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
PKG = "/root/reference/src/preference_guided_image_captioning_alignment/"
OUT = os.path.join(ROOT, "tests", "golden")

ACCUM, EPOCHS, B, S = 2, 2, 2, 16
LR, WARMUP, MAX_NORM, BETA, TAU = 1e-3, 2, 1.0, 0.1, 0.5


def import_reference_trainer():
    """reference models/model.py + training/trainer.py as modules of a synthetic package ``refsrc`` (the package's own
    ``__init__`` files pull in data/ and evaluation/, which need torchvision / nltk - not on this path)."""
    # transformers probes for an installed ``peft`` while importing its schedulers: let it conclude "absent" first
    from transformers import get_cosine_schedule_with_warmup, get_linear_schedule_with_warmup  # noqa: F401
    import accelerate  # noqa: F401
    peft = types.ModuleType("peft")
    peft.LoraConfig = type("LoraConfig", (), {"__init__": lambda self, **kw: None})
    peft.get_peft_model = lambda m, c: m
    sys.modules.setdefault("peft", peft)
    mlflow = types.ModuleType("mlflow")
    for fn in ("set_experiment", "start_run", "log_params", "log_metric", "log_metrics", "end_run"):
        setattr(mlflow, fn, lambda *a, **k: None)
    sys.modules.setdefault("mlflow", mlflow)
    for name, sub in (("refsrc", ""), ("refsrc.models", "models"), ("refsrc.utils", "utils"),
                      ("refsrc.training", "training")):
        m = types.ModuleType(name)
        m.__path__ = [os.path.join(PKG, sub)]
        sys.modules[name] = m

    def load(name, rel):
        spec = importlib.util.spec_from_file_location(name, os.path.join(PKG, rel))
        mod = importlib.util.module_from_spec(spec)
        sys.modules[name] = mod
        spec.loader.exec_module(mod)
        return mod

    load("refsrc.models.components", "models/components.py")
    model = load("refsrc.models.model", "models/model.py")
    config = load("refsrc.utils.config", "utils/config.py")
    trainer = load("refsrc.training.trainer", "training/trainer.py")
    return model, config, trainer


def write_config(tmp: str) -> str:
    import yaml
    cfg = {
        "data": {"image_size": 32, "max_caption_length": S, "num_workers": 0},
        "model": {"vision_model": "tiny", "text_model": "tiny", "projection_dim": 32, "temperature": TAU},
        "training": {
            "stage1": {"num_epochs": EPOCHS, "batch_size": B, "learning_rate": LR, "warmup_steps": WARMUP,
                       "gradient_accumulation_steps": ACCUM, "max_grad_norm": MAX_NORM, "weight_decay": 0.01},
            "stage2": {"num_epochs": EPOCHS, "batch_size": B, "learning_rate": LR, "warmup_steps": WARMUP,
                       "gradient_accumulation_steps": ACCUM, "max_grad_norm": MAX_NORM, "weight_decay": 0.01,
                       "dpo_beta": BETA}},
        "evaluation": {}, "targets": {}, "hardware": {"mixed_precision": "no"},
        "paths": {"output_dir": os.path.join(tmp, "out")}, "logging": {},
    }
    path = os.path.join(tmp, "cfg.yaml")
    with open(path, "w") as fh:
        yaml.safe_dump(cfg, fh)
    return path


def make_batches(arch, stage: int):
    """5 micro-batches of B samples (accumulation 2 -> groups {0,1}, {2,3}, {4}): micro-batch 2 (opens a group) and
    micro-batch 1 of the SECOND list position... see ``nan_at``."""
    g = torch.Generator().manual_seed(100 + stage)
    n = 7                                        # groups {0,1} {2,3} {4,5} {6}: odd count -> trailing partial group
    nan_at = (2, 5)                              # 2 opens group {2,3}; 5 closes group {4,5}
    out = []
    for i in range(n):
        img = torch.randn(B, 3, arch.vit.image, arch.vit.image, generator=g)
        b = {"image": img}
        names = ("caption",) if stage == 1 else ("preferred", "rejected")
        for name in names:
            ids = torch.randint(0, arch.gpt.base_vocab, (B, S), generator=g)
            lens = torch.randint(3, S + 1, (B,), generator=g)
            if stage == 2 and i in nan_at and name == "preferred":
                lens[0] = 1                      # one real token -> 0/0 in PreferenceLoss (model.py:1082-1083) -> NaN
            mask = (torch.arange(S)[None] < lens[:, None]).long()
            b[name + "_ids"] = torch.where(mask.bool(), ids, torch.full_like(ids, arch.gpt.base_vocab))
            b[name + "_mask"] = mask
        if stage == 1 and i in nan_at:
            img[0, 0, 0, 0] = float("nan")       # a NaN pixel -> NaN embeddings -> NaN contrastive loss
        if stage == 2:
            b["preference_score"] = torch.ones(B)
        out.append(b)
    return out, nan_at


class ListDataset(torch.utils.data.Dataset):
    def __init__(self, batches):
        self.items = [{k: v[j] for k, v in b.items()} for b in batches for j in range(B)]

    def __len__(self):
        return len(self.items)

    def __getitem__(self, i):
        return self.items[i]


def run_stage(stage: int, refmodel, refconfig, reftrainer, tmp: str) -> dict:
    from accelerate import Accelerator

    from oracle.make_golden import build_reference_model
    from pgca_amd.arch import tiny_arch
    from pgca_amd.params import ParamStore
    arch = tiny_arch()
    store = ParamStore(arch, "cpu", seed=4321, frozen=())
    sd0 = {k: v.clone() for k, v in store.state_dict().items()}
    model = build_reference_model(refmodel, arch, sd0)
    for m in model.modules():                     # no dropout anywhere (see module docstring)
        if isinstance(m, nn.Dropout):
            m.p = 0.0
        if isinstance(m, nn.MultiheadAttention):
            m.dropout = 0.0
    for gm in (model.text_encoder.text_model, model.caption_decoder.lm_model):
        gm.config.attn_pdrop = gm.config.resid_pdrop = gm.config.embd_pdrop = 0.0
        for mod in gm.modules():
            if hasattr(mod, "attn_dropout") and isinstance(mod.attn_dropout, nn.Dropout):
                mod.attn_dropout.p = 0.0
    model.temperature = TAU
    # the reference freezes the CLIP tower in every shipped config (configs/default.yaml:23)
    for p in model.vision_encoder.clip_model.parameters():
        p.requires_grad_(False)
    batches, nan_at = make_batches(arch, stage)
    loader = torch.utils.data.DataLoader(ListDataset(batches), batch_size=B, shuffle=False)
    cfg = refconfig.Config(write_config(tmp))
    acc = Accelerator(gradient_accumulation_steps=ACCUM, mixed_precision="no", cpu=True)
    tr = reftrainer.PreferenceGuidedTrainer(model, cfg, loader, loader, loader, loader, accelerator=acc)
    stage_cfg = cfg.get(f"training.stage{stage}")
    steps_per_epoch = len(loader) // ACCUM
    optimizer, scheduler = tr._setup_optimizer_and_scheduler(stage, steps_per_epoch * EPOCHS)
    train_loader = acc.prepare(loader)
    losses, lrs = [], []
    loss_mod = tr.contrastive_loss if stage == 1 else tr.preference_loss
    def record(_mod, _inp, out_):       # returns None: the hook must not replace the loss
        losses.append(float(out_.detach()))
        lrs.append(float(optimizer.param_groups[0]["lr"]))

    loss_mod.register_forward_hook(record)
    means = []
    for ep in range(EPOCHS):
        tr.epoch = ep
        fn = tr._train_epoch_stage1 if stage == 1 else tr._train_epoch_stage2
        means.append(float(fn(train_loader, optimizer, scheduler, stage_cfg)))
    inner = optimizer.optimizer if hasattr(optimizer, "optimizer") else optimizer
    opt_steps = max((int(st["step"]) for st in inner.state.values() if "step" in st), default=0)
    out = {"losses": np.array(losses, dtype=np.float64), "lrs": np.array(lrs, dtype=np.float64),
           "epoch_means": np.array(means, dtype=np.float64), "global_step": np.int64(tr.global_step),
           "opt_steps": np.int64(opt_steps), "nan_at": np.array(nan_at, dtype=np.int64),
           "n_batches": np.int64(len(batches))}
    for i, b in enumerate(batches):
        for k, v in b.items():
            out[f"batch{i}_{k}"] = v.numpy().copy()
    unwrapped = acc.unwrap_model(tr.model)
    final = unwrapped.state_dict()
    trained = [n for n, p in unwrapped.named_parameters() if p.requires_grad and n in sd0
               and not torch.equal(final[n], sd0[n])]
    out["trained_names"] = json.dumps(trained)
    # parameters that moved, kept compact: per-tensor sums + a strided sample (the fixture stays small)
    for n in trained:
        t = final[n].detach().double().flatten()
        out["final_sum::" + n] = np.array([float(t.sum()), float(t.abs().sum())])
        out["final_sample::" + n] = t[::max(1, t.numel() // 64)][:64].numpy().copy()
    return out


def main():
    import accelerate
    import transformers
    refmodel, refconfig, reftrainer = import_reference_trainer()
    out = {"meta": json.dumps({"torch": torch.__version__, "transformers": transformers.__version__,
                               "accelerate": accelerate.__version__, "seed": 4321, "accum": ACCUM, "epochs": EPOCHS,
                               "B": B, "S": S, "lr": LR, "warmup": WARMUP, "max_norm": MAX_NORM, "beta": BETA, "tau": TAU,
                               "reference": "training/trainer.py _train_epoch_stage1/_train_epoch_stage2 under "
                                            "accelerate.Accelerator(gradient_accumulation_steps=2, cpu=True)"})}
    with tempfile.TemporaryDirectory() as tmp:
        cwd = os.getcwd()
        os.chdir(tmp)                              # the trainer creates ./outputs-style directories
        try:
            for stage in (2, 1):
                for k, v in run_stage(stage, refmodel, refconfig, reftrainer, tmp).items():
                    out[f"s{stage}_{k}"] = v
        finally:
            os.chdir(cwd)
    path = os.path.join(OUT, "trainer_epoch.npz")
    np.savez_compressed(path, **out)
    print(path, os.path.getsize(path), "bytes")
    for s in (2, 1):
        print(f"stage {s}: losses", np.round(out[f"s{s}_losses"], 5).tolist(), "global_step", int(out[f"s{s}_global_step"]),
              "opt_steps", int(out[f"s{s}_opt_steps"]), "means", out[f"s{s}_epoch_means"].tolist())


if __name__ == "__main__":
    main()
