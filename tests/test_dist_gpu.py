"""Data-parallel Stage-2 step on the GPU with 2 ranks (both on cuda:0, gloo transport - the box has one GPU;
RCCL is exercised by the driver's multi-GPU bench): all-reduced, 1/world-scaled gradients and the resulting
AdamW update must equal the single-process step on the concatenated batch."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _batch(n, S, vocab, seed):
    g = torch.Generator().manual_seed(seed)
    lens = torch.randint(4, S + 1, (2 * n,), generator=g)
    ids = torch.randint(0, vocab, (2 * n, S), generator=g)
    mask = (torch.arange(S)[None] < lens[:, None]).long()
    return {"image": torch.randn(n, 3, 64, 64, generator=g), "preferred_ids": ids[:n], "rejected_ids": ids[n:],
            "preferred_mask": mask[:n], "rejected_mask": mask[n:]}


def _run_step(batch, dp, train_vit=False):
    from pgca_amd.arch import tiny_arch
    from pgca_amd.dist import OverlappedTrunkReducer
    from pgca_amd.model import PreferenceGuidedCaptioningModel
    from pgca_amd.steps import DPOStep, FusedOptimizer
    model = PreferenceGuidedCaptioningModel(freeze_vision_backbone=not train_vit, arch=tiny_arch(), seed=5,
                                            device="cuda:0")
    step = DPOStep(model.store, model.ws, model.vision_encoder.tower, model.vision_encoder.head,
                   model.caption_decoder.engine, beta=0.1, reference_free=True)
    segs = [model.store.segments["vision_head"], model.store.segments["decoder"]]
    if train_vit:      # the CLIP tower left trainable: its flat gradient buffer is one more all-reduced segment
        segs.insert(0, model.store.segments["vit"])
    opt = FusedOptimizer(segs, lr=1e-3, max_grad_norm=1.0, total_steps=10)
    red = OverlappedTrunkReducer(dp, model.caption_decoder.engine.trunk, group=1) if dp else None
    if red:
        red.arm()
    p = DPOStep.prepare(batch, model.device)
    opt.zero_grad()
    loss = float(step.loss_and_grads(p["image"], p["seq"]))
    if red:
        red.finish(other_segments=segs[:-1])
    world = dp.world if dp else 1
    grads = [s.grad.clone().cpu() / world for s in segs]
    opt.step(grad_scale=1.0 / world)
    torch.cuda.synchronize()
    return loss, grads, [s.fp32.clone().cpu() for s in segs], opt.state()


def _worker(rank, world, port, out_dir, train_vit=False):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from pgca_amd.dist import DataParallel
        dp = DataParallel(bucket_elems=50000)
        full = _batch(8, 16, 509, seed=77)
        lo, hi = dp.shard(8)
        mine = {k: v[lo:hi] for k, v in full.items()}
        loss, grads, params, st = _run_step(mine, dp, train_vit)
        torch.save((rank, loss, grads, params, st), os.path.join(out_dir, f"rank{rank}.pt"))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("train_vit", [False, True])
def test_two_rank_step_equals_single_process_step(tmp_path, train_vit):
    world = 2
    ctx = mp.get_context("spawn")
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, str(tmp_path), train_vit)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=300)
        assert p.exitcode == 0
    res = [torch.load(tmp_path / f"rank{r}.pt", weights_only=False) for r in range(world)]
    loss1, grads1, params1, st1 = _run_step(_batch(8, 16, 509, seed=77), None, train_vit)
    assert abs(0.5 * (res[0][1] + res[1][1]) - loss1) <= 1e-4            # mean of the rank losses == global mean
    for r in range(world):
        _, _, grads, params, st = res[r]
        for a, b in zip(grads, grads1):
            cosv = float((a.double() @ b.double()) / (a.double().norm() * b.double().norm()))
            assert cosv >= 0.9999 and abs(float(a.norm()) / float(b.norm()) - 1) <= 1e-3
        for a, b in zip(params, params1):                                   # same AdamW update on every rank
            assert float((a - b).abs().max()) <= 2e-3 * 1e-3 + 1e-6 or torch.allclose(a, b, atol=2e-4)
        assert abs(st["grad_norm"] - st1["grad_norm"]) <= 2e-3 * st1["grad_norm"] and st["step"] == 1
    for a, b in zip(res[0][3], res[1][3]):
        assert torch.equal(a, b)                                             # replicas stay bit-identical


# ----------------------------------------------------------------------------------------------- Stage 1
def _batch_s1(n, S, vocab, seed):
    g = torch.Generator().manual_seed(seed)
    lens = torch.randint(4, S + 1, (n,), generator=g)
    ids = torch.randint(0, vocab, (n, S), generator=g)
    mask = (torch.arange(S)[None] < lens[:, None]).long()
    return {"image": torch.randn(n, 3, 64, 64, generator=g), "caption_ids": ids, "caption_mask": mask}


def _run_step_s1(batch, dp, tau=0.07):
    from pgca_amd.arch import tiny_arch
    from pgca_amd.dist import OverlappedTrunkReducer
    from pgca_amd.model import PreferenceGuidedCaptioningModel
    from pgca_amd.steps import ContrastiveStep, FusedOptimizer
    model = PreferenceGuidedCaptioningModel(freeze_vision_backbone=True, arch=tiny_arch(), seed=5, device="cuda:0")
    step = ContrastiveStep(model.store, model.ws, model.vision_encoder.tower, model.vision_encoder.head,
                           model.text_encoder.engine, temperature=tau, dp=dp, global_negatives=True)
    segs = [model.store.segments[n] for n in ("vision_head", "text_head", "text_tower")]
    opt = FusedOptimizer(segs, lr=1e-3, max_grad_norm=1.0, total_steps=10)
    red = OverlappedTrunkReducer(dp, model.text_encoder.engine.trunk, group=1) if dp else None
    if red:
        red.arm()
    p = ContrastiveStep.prepare(batch, model.device)
    opt.zero_grad()
    loss = float(step.loss_and_grads(p["image"], p["ids"], p["mask"]))
    if red:
        red.finish(other_segments=segs[:2])
    world = dp.world if dp else 1
    grads = [s.grad.clone().cpu() / world for s in segs]
    opt.step(grad_scale=1.0 / world)
    torch.cuda.synchronize()
    return loss, grads, [s.fp32.clone().cpu() for s in segs], opt.state()


def _worker_s1(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from pgca_amd.dist import DataParallel
        dp = DataParallel(bucket_elems=50000)
        full = _batch_s1(12, 16, 509, seed=78)
        lo, hi = dp.shard(12)
        loss, grads, params, st = _run_step_s1({k: v[lo:hi] for k, v in full.items()}, dp)
        torch.save((rank, loss, grads, params, st), os.path.join(out_dir, f"s1rank{rank}.pt"))
    finally:
        dist.destroy_process_group()


def test_two_rank_stage1_global_negatives_equals_single_process(tmp_path):
    """Stage-1 step with global negatives (all-gather of embeddings + of 2N log-sum-exps, no gradient collective for
    the loss) on 2 ranks == the single-process step whose local batch IS the concatenated batch
    (reference model.py:984-1000 on all N rows)."""
    world = 2
    ctx = mp.get_context("spawn")
    port = _free_port()
    procs = [ctx.Process(target=_worker_s1, args=(r, world, port, str(tmp_path))) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=300)
        assert p.exitcode == 0
    res = [torch.load(tmp_path / f"s1rank{r}.pt", weights_only=False) for r in range(world)]
    loss1, grads1, params1, st1 = _run_step_s1(_batch_s1(12, 16, 509, seed=78), None)
    for r in range(world):
        _, loss, grads, params, st = res[r]
        assert abs(loss - loss1) <= 2e-4, (loss, loss1)           # every rank reports the GLOBAL loss
        for a, b in zip(grads, grads1):
            cosv = float((a.double() @ b.double()) / (a.double().norm() * b.double().norm()))
            assert cosv >= 0.9999 and abs(float(a.norm()) / float(b.norm()) - 1) <= 2e-3, cosv
        assert abs(st["grad_norm"] - st1["grad_norm"]) <= 2e-3 * st1["grad_norm"] and st["step"] == 1
    for a, b in zip(res[0][3], res[1][3]):
        assert torch.equal(a, b)                                   # replicas stay bit-identical


# ----------------------------------------------------------------------------------------------- early stopping
def _worker_es(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from pgca_amd import REPO_ROOT
        from pgca_amd.arch import tiny_arch
        from pgca_amd.config import Config
        from pgca_amd.model import PreferenceGuidedCaptioningModel
        from pgca_amd.trainer import PreferenceGuidedTrainer
        cfg = Config(os.path.join(REPO_ROOT, "configs", "default.yaml"))
        cfg.set("paths.output_dir", os.path.join(out_dir, f"out{rank}"))
        cfg.set("training.stage2.num_epochs", 6)
        cfg.set("training.stage2.learning_rate", 0.0)            # nothing improves: val loss is constant
        cfg.set("training.stage2.early_stopping_patience", 2)
        cfg.set("training.stage2.gradient_accumulation_steps", 1)
        cfg.set("training.stage1.gradient_accumulation_steps", 1)
        model = PreferenceGuidedCaptioningModel(freeze_vision_backbone=True, arch=tiny_arch(), dropout=0.0, seed=5,
                                                device="cuda:0")
        mk = lambda seed: [_batch(2, 16, 509, seed=seed + 10 * rank + i) for i in range(2)]  # noqa: E731
        tr = PreferenceGuidedTrainer(model, cfg, [], [], mk(1), mk(50))
        out = tr.train_stage2()
        torch.save((rank, len(out["train_loss"]), out["val_loss"], tr.best_val_loss), os.path.join(out_dir, f"es{rank}.pt"))
    finally:
        dist.destroy_process_group()


def test_two_ranks_leave_the_loop_together_on_early_stopping(tmp_path):
    """The reference updates best_val_loss inside the rank-0-only checkpoint save, so only rank 0 ever reaches its
    patience and the others block in the next all-reduce.  Here every rank tracks it on the rank-reduced val loss:
    with a constant val loss and patience 2 both ranks stop after epoch 2 (epoch 0 sets the best)."""
    world = 2
    ctx = mp.get_context("spawn")
    port = _free_port()
    procs = [ctx.Process(target=_worker_es, args=(r, world, port, str(tmp_path))) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=300)
        assert p.exitcode == 0, "a rank hung or failed"
    res = [torch.load(tmp_path / f"es{r}.pt", weights_only=False) for r in range(world)]
    assert res[0][1] == res[1][1] == 3                             # epochs run: 0 (best), 1, 2 -> stop
    assert res[0][2] == res[1][2]                                  # identical (rank-reduced) validation losses
    assert res[0][3] == res[1][3] == res[0][2][0]
