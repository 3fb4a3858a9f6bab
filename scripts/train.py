#!/usr/bin/env python3
"""Train the two stages on MI355X with the reference's command line
(reference scripts/train.py:253-294: --config --resume --stage {1,2} --output-dir --log-level --dry-run).

Data loading is outside the hot path: batches come from seeded synthetic datasets with the reference's
batch-dict contract (loader.py:252-258,487-497; the reference's own dummy loader, scripts/train.py:194-241),
sized by --synthetic-samples.  Launch with torch.distributed.run for data parallelism.
"""
import argparse
import logging
import os
import random
import sys

import numpy as np
import torch
from torch.utils.data import DataLoader, Dataset

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from pgca_amd.config import Config  # noqa: E402
from pgca_amd.dist import DataParallel  # noqa: E402


class SyntheticPairs(Dataset):
    """Image + caption(s) with right padding; preference pairs when ``pairs`` is set."""

    def __init__(self, n, image_size, seq_len, vocab, pad_id, pairs, seed):
        g = torch.Generator().manual_seed(seed)
        self.n, self.pairs, self.image_size, self.seed = n, pairs, image_size, seed
        k = 2 if pairs else 1
        ids = torch.randint(0, vocab, (n, k, seq_len), generator=g)
        lens = torch.randint(min(16, seq_len), seq_len + 1, (n, k), generator=g)
        self.mask = (torch.arange(seq_len)[None, None] < lens[..., None]).long()
        self.ids = torch.where(self.mask.bool(), ids, torch.full_like(ids, pad_id))

    def __len__(self):
        return self.n

    def __getitem__(self, i):
        g = torch.Generator().manual_seed(self.seed * 1000003 + i)
        img = torch.randn(3, self.image_size, self.image_size, generator=g)
        if self.pairs:
            return {"image": img, "preferred_ids": self.ids[i, 0], "preferred_mask": self.mask[i, 0],
                    "rejected_ids": self.ids[i, 1], "rejected_mask": self.mask[i, 1],
                    "preference_score": torch.tensor(1.0)}
        return {"image": img, "caption_ids": self.ids[i, 0], "caption_mask": self.mask[i, 0]}


def set_random_seeds(seed: int) -> None:
    random.seed(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)
    torch.cuda.manual_seed_all(seed)


def shard(ds, dp):
    """Rank r takes items r, r + world, ... of the first len // world * world items: every rank holds the same
    number of samples, hence the same number of batches (the step loop's collectives are per batch)."""
    n = len(ds) // dp.world * dp.world
    return torch.utils.data.Subset(ds, list(range(dp.rank, n, dp.world)))


def main():
    ap = argparse.ArgumentParser(description="Train preference-guided captioning model (MI355X path)")
    ap.add_argument("--config", type=str, default=os.path.join(os.path.dirname(__file__), "..", "configs", "default.yaml"))
    ap.add_argument("--resume", type=str, default=None)
    ap.add_argument("--stage", type=int, choices=[1, 2], default=None)
    ap.add_argument("--output-dir", type=str, default=None)
    ap.add_argument("--log-level", type=str, default="INFO", choices=["DEBUG", "INFO", "WARNING", "ERROR"])
    ap.add_argument("--dry-run", action="store_true")
    ap.add_argument("--synthetic-samples", type=int, default=256)
    args = ap.parse_args()

    logging.basicConfig(level=getattr(logging, args.log_level), format="%(asctime)s %(name)s %(levelname)s %(message)s")
    log = logging.getLogger("train")
    cfg = Config(args.config)
    if args.output_dir:
        cfg.set("paths.output_dir", args.output_dir)
    seed = int(cfg.get("training.seed", 42))
    set_random_seeds(seed)
    dp = DataParallel.init_from_env()
    if torch.cuda.is_available():
        torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")))

    from pgca_amd.model import PreferenceGuidedCaptioningModel
    from pgca_amd.trainer import PreferenceGuidedTrainer
    mc = cfg.get_model_config()
    model = PreferenceGuidedCaptioningModel(
        vision_model=mc.get("vision_model", "openai/clip-vit-base-patch32"),
        text_model=mc.get("text_model", "microsoft/DialoGPT-medium"), projection_dim=mc.get("projection_dim", 512),
        temperature=mc.get("temperature", 0.07), dropout=mc.get("dropout", 0.1),
        freeze_vision_backbone=mc.get("freeze_vision_backbone", False),
        freeze_text_backbone=mc.get("freeze_text_backbone", False), lora_config=mc.get("lora_config"), seed=seed)
    log.info(f"model: {model.store.num_params() / 1e6:.1f} M parameters on {model.device}")

    S, I = int(cfg.get("data.max_caption_length", 128)), int(cfg.get("data.image_size", 224))
    vocab = model.arch.gpt.base_vocab
    n = args.synthetic_samples

    def loaders(pairs, bs, s):
        tr = shard(SyntheticPairs(int(n * 0.8), I, S, vocab, vocab, pairs, s), dp)
        va = shard(SyntheticPairs(max(bs * dp.world, int(n * 0.1)), I, S, vocab, vocab, pairs, s + 1), dp)
        return (DataLoader(tr, batch_size=bs, shuffle=True, drop_last=True, generator=torch.Generator().manual_seed(s)),
                DataLoader(va, batch_size=bs, shuffle=False, drop_last=True))

    tl1, vl1 = loaders(False, int(cfg.get("training.stage1.batch_size", 8)), seed)
    tl2, vl2 = loaders(True, int(cfg.get("training.stage2.batch_size", 8)), seed + 7)
    trainer = PreferenceGuidedTrainer(model, cfg, tl1, vl1, tl2, vl2)
    if args.resume:
        trainer.load_checkpoint(args.resume)
    if args.dry_run:
        log.info("Dry run completed successfully")
        return
    try:
        if args.stage == 1:
            res = trainer.train_stage1()
        elif args.stage == 2:
            res = trainer.train_stage2()
        else:
            res = trainer.train()
        log.info(f"Training completed: {res}")
    except KeyboardInterrupt:
        log.info("Training interrupted by user")


if __name__ == "__main__":
    main()
