#!/usr/bin/env python3
"""Benchmark of the hot path: Stage-2 DPO preference-pairs/sec (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" is one full optimizer step on one batch of B preference pairs per GPU:
ViT-B/32 forward (once per image) -> policy forward on chosen+rejected -> frozen reference-policy
forward on chosen+rejected (4 decoder forwards per pair) -> fused LM-head log-prob gather -> DPO loss
-> backward through the policy only -> (RCCL gradient all-reduce, N > 1) -> global-norm clip + AdamW
+ cosine warm-up.  Inputs are synthetic and resident in HBM before the timed region.

Output: ONE JSON line on rank 0 with the contract's keys plus
  "roofline":     dominant kernel (forward NN MFMA GEMM) TFLOP/s from HIP events on its launch stream
  "cpu_baseline": the oracle restatement timed on this box's host cores on a bounded sample.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

# dmabuf IPC (RCCL across the ranks of a node needs it on this pool); must be in place before the HIP runtime starts
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import torch  # noqa: E402

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

PEAK_BF16_TFLOPS = 2500.0  # dense MFMA bf16, MI355X_MICROARCH.md chip table


def synthetic_batch(B: int, S: int, vocab: int, pad_id: int, seed: int):
    """randn images; ids ~ U[0, 50257); valid length ~ U{16..S}; right padding."""
    g = torch.Generator().manual_seed(seed)
    images = torch.randn(B, 3, 224, 224, generator=g)
    out = {"image": images}
    for name in ("preferred", "rejected"):
        ids = torch.randint(0, vocab, (B, S), generator=g)
        lens = torch.randint(16, S + 1, (B,), generator=g)
        mask = (torch.arange(S)[None] < lens[:, None]).long()
        out[name + "_ids"] = torch.where(mask.bool(), ids, torch.full_like(ids, pad_id))
        out[name + "_mask"] = mask
    return out


class GemmProbe:
    """Collects HIP-event pairs around every forward (NN, plain/GELU epilogue) decoder GEMM launch."""

    def __init__(self, layout):
        self.layout, self.events, self.enabled = layout, [], False

    def want(self, layout, epilogue, plan):
        # plan = schedule*1e6 + tile*100 + splits: keep launches of gemm256s_kernel<0,1> (schedule 6 = the
        # phase-staggered loop the NN layout runs by default, tile 256)
        return self.enabled and layout == self.layout and plan // 100 == 60256

    def add(self, e0, e1, flops):
        self.events.append((e0, e1, flops))

    def summary(self):
        if not self.events:
            return None
        ms = sum(e0.elapsed_time(e1) for e0, e1, _ in self.events)
        fl = sum(f for _, _, f in self.events)
        return {"launches": len(self.events), "avg_us": 1e3 * ms / len(self.events), "tflops": fl / (ms * 1e-3) / 1e12}


def host_cores() -> int:
    """Cores this process may actually use (cgroup/affinity share, capped at 16 = the 1-GPU box share)."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    return max(1, min(n, 16))


def log(msg: str) -> None:
    if int(os.environ.get("RANK", "0")) == 0:
        print(f"[bench {time.strftime('%H:%M:%S')}] {msg}", file=sys.stderr, flush=True)


def cpu_baseline(model, arch, S, beta, pairs):
    """Oracle (CPU restatement, fp32) on the host cores: one 4-forward DPO step, fwd + bwd."""
    from oracle import restatement as R
    torch.set_num_threads(host_cores())
    sd = {k: v.detach().cpu().clone() for k, v in model.store.state_dict(aliases=False).items()
          if not k.startswith("text_encoder.")}
    for k, v in sd.items():
        if k.startswith(("caption_decoder", "vision_encoder.projection")):
            v.requires_grad_()
    b = synthetic_batch(pairs, S, arch.gpt.base_vocab, arch.gpt.base_vocab, seed=99)

    def step():
        lps = {}
        for tag, req in (("pol", True), ("ref", False)):
            with torch.set_grad_enabled(req):
                for name in ("preferred", "rejected"):
                    logits = R.model_forward(sd, b["image"], b[name + "_ids"], b[name + "_mask"], "generation",
                                             arch.vit.heads, arch.vit.patch, arch.gpt.heads)["logits"]
                    lps[tag, name] = R.sequence_logprob_sum(logits, b[name + "_ids"], b[name + "_mask"])
        loss, _ = R.dpo_loss(lps["pol", "preferred"], lps["pol", "rejected"], lps["ref", "preferred"],
                             lps["ref", "rejected"], beta=beta)
        loss.backward()
        for v in sd.values():
            v.grad = None
        return float(loss)

    # B = 8, one warm-up then three timed optimizer-free steps.  The warm-up runs on ONE pair (it is
    # there to page in the weights and the allocator, which one pair does), so the whole leg stays near 25-30 s of host time
    full = b
    b = {k: v[:1] for k, v in full.items()}
    step()
    b = full
    times = []
    for _ in range(3):
        t0 = time.time()
        step()
        times.append(time.time() - t0)
    dt = sum(times) / len(times)
    return {"value": pairs / dt, "unit": "pairs/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"3 timed optimizer-free 4-forward DPO steps (fwd+bwd) of {pairs} pairs after a 1-pair warm-up, S={S}, "
                      f"fp32, full depth, oracle/restatement.py; {', '.join(f'{t:.1f}' for t in times)} s"}


PMC_FILE = "profiles/r03_pmc_summary.json"


def pmc_from_profile(pairs_per_gpu: int, kernel_substr: str):
    """HBM bytes per launch and matrix-pipe busy fraction of the dominant kernel from COMMITTED rocprofv3 PMC passes of this
    command (made by tools/pmc_summary.py: the counters cannot be collected from inside the process, and FETCH_SIZE /
    WRITE_SIZE need separate passes).  They are NOT measured in this run: the line says which file and which commit they
    come from, and they are only reported for the workload the passes were taken on."""
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), PMC_FILE)
    try:
        with open(path) as fh:
            d = json.load(fh)
        if d.get("pairs_per_gpu") != pairs_per_gpu:
            return None
        for k in d["kernels"]:
            if kernel_substr in k["kernel"]:
                return k
    except (OSError, KeyError, ValueError):
        pass
    return None


def self_launch(n: int, argv, timeout_s: float) -> int:
    """``python bench.py --gpus N`` started as ONE plain process (how the driver starts ``--gpus 1``): spawn the N ranks
    here, as fresh children of a parent that has not touched the GPU (no HIP call, no ``torch.cuda.*``) - one
    ``torch.distributed.run`` child that starts one rank per GPU, the launch the reference gets from Accelerate
    (reference scripts/train.py:317-322).  Rank 0's single JSON line is forwarded on stdout; the exit code is non-zero if
    any rank fails or the job exceeds ``timeout_s``."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr",
           "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + list(argv)
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC: RCCL needs it on this pool
    env.setdefault("OMP_NUM_THREADS", "8")
    log(f"self-launch: {n} ranks via torch.distributed.run on 127.0.0.1:{port}")
    proc = subprocess.Popen(cmd, stdout=subprocess.PIPE, env=env, text=True, start_new_session=True)
    try:
        out, _ = proc.communicate(timeout=timeout_s)
    except subprocess.TimeoutExpired:
        import signal
        os.killpg(proc.pid, signal.SIGKILL)      # exactly the process group started above
        proc.wait()
        print(f"[bench] self-launched job exceeded {timeout_s:.0f} s and was killed", file=sys.stderr, flush=True)
        return 124
    lines = [ln for ln in out.splitlines() if ln.startswith("{") and '"metric"' in ln]
    if proc.returncode != 0:
        print(f"[bench] a rank failed (torch.distributed.run exit code {proc.returncode})", file=sys.stderr, flush=True)
        return proc.returncode or 1
    if len(lines) != 1:
        print(f"[bench] expected ONE result line from rank 0, got {len(lines)}", file=sys.stderr, flush=True)
        return 1
    print(lines[0], flush=True)
    return 0


def dry_launch(args) -> None:
    """``--dry-launch``: every rank joins the process group (RCCL, or gloo under PGCA_BENCH_BACKEND=gloo), proves the group
    with one all-reduce and rank 0 prints a result-shaped line - the launcher, the rendezvous and the rc propagation
    without a model or a GPU kernel (tests/test_bench_launch_cpu.py)."""
    import torch.distributed as dist
    backend = os.environ.get("PGCA_BENCH_BACKEND", "nccl")
    world, rank = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0"))
    if world > 1:
        if backend == "nccl":
            torch.cuda.set_device(int(os.environ.get("PGCA_BENCH_DEVICE", os.environ.get("LOCAL_RANK", "0"))))
        dist.init_process_group(backend=backend)
    if args.dry_fail_rank == rank:
        raise SystemExit(3)
    t = torch.ones(1, device="cuda" if backend == "nccl" and world > 1 else "cpu")
    if world > 1:
        dist.all_reduce(t)
        dist.barrier()
    if rank == 0:
        print(json.dumps({"metric": "dry launch (no model, no kernel)", "value": 0.0, "unit": "pairs/s",
                          "n_gpus": args.gpus, "dry_launch": True, "ranks_summed": int(t.item()),
                          "process_group": {"world_size": dist.get_world_size() if world > 1 else 1,
                                            "backend": dist.get_backend() if world > 1 else "none"}}), flush=True)
    if world > 1:
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--dry-launch", action="store_true",
                    help="launcher rehearsal: ranks rendezvous, all-reduce once, rank 0 prints a result-shaped line")
    ap.add_argument("--dry-fail-rank", type=int, default=-1, help="--dry-launch: this rank exits with code 3")
    ap.add_argument("--launch-timeout", type=float, default=3000.0, help="self-launched job limit in seconds")
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--pairs-per-gpu", type=int, default=512,
                    help="preference pairs per GPU per optimizer step (rounds 1-2 ran 128 / 256 on padded rows; with packed "
                         "rows 512 pairs are ~75 k token rows per GEMM and ~90 of the 288 GB: +5 %% pairs/s over 256)")
    ap.add_argument("--padded", action="store_true",
                    help="A/B: run the trunks on all B x S padded positions like the reference (round-2 behaviour) instead "
                         "of the packed rows of the real tokens")
    ap.add_argument("--seq-len", type=int, default=128)
    ap.add_argument("--vision-model", default="openai/clip-vit-base-patch32")
    ap.add_argument("--text-model", default="gpt2-medium")
    ap.add_argument("--reference-free", action="store_true", help="2-forward trainer path instead of 4-forward DPO")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-pairs", type=int, default=8, help="pairs per CPU-baseline step (BASELINE.md section 3: 8)")
    ap.add_argument("--no-probe", action="store_true")
    ap.add_argument("--dropout", type=float, default=0.1, help="train-mode dropout of the policy (reference: 0.1)")
    ap.add_argument("--with-input-path", action="store_true",
                    help="feed every step from HOST batches through the input path (pinned staging, async H2D, device-side "
                         "index preparation, prefetch depth 2) instead of batches resident in HBM; the contract's `value` "
                         "is the resident number - this flag reports what the loop costs with a real feeder")
    ap.add_argument("--allreduce-bf16", action="store_true",
                    help="N > 1: bf16-compressed gradient all-reduce (half the xGMI bytes; mi355x.allreduce_bf16)")
    ap.add_argument("--no-wgrad-overlap", action="store_true",
                    help="A/B: keep the grouped weight-gradient launches on the main stream (round-1 behaviour)")
    ap.add_argument("--ref-side-stream", action="store_true",
                    help="run the frozen reference-policy forward on a second HIP stream, concurrently with the policy's "
                         "(+1.4 %% pairs/s; not the default because two GEMMs sharing the chip make the per-launch "
                         "duration behind `roofline` meaningless - DESIGN section 5)")
    ap.add_argument("--set", action="append", default=[], metavar="NAME=VALUE",
                    help="pgca_set_option dispatch knob (e.g. gemm_stagger=8); repeatable")
    ap.add_argument("--trainable-vision", action="store_true",
                    help="secondary line: leave the CLIP tower trainable (reference default freeze_vision_backbone=False; the "
                         "headline config freezes it as configs/default.yaml does)")
    ap.add_argument("--stage", type=int, default=2, choices=(1, 2),
                    help="2 (default): the headline Stage-2 DPO step; 1: secondary line, Stage-1 NT-Xent step "
                         "(global negatives over the ranks when N > 1)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # plain `python bench.py --gpus N`: this process becomes the launcher (nothing below has touched the GPU)
        raise SystemExit(self_launch(args.gpus, sys.argv[1:], args.launch_timeout))
    if args.dry_launch:
        return dry_launch(args)

    from pgca_amd import hip
    from pgca_amd.arch import make_arch
    from pgca_amd.dist import DataParallel, OverlappedTrunkReducer
    from pgca_amd.engine import DropoutPlan
    from pgca_amd.model import PreferenceGuidedCaptioningModel
    from pgca_amd.steps import ContrastiveStep, DPOStep, FusedOptimizer, ReferencePolicy

    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: the launcher's rank count and --gpus disagree")
    # rehearsal hooks (one-GPU box): PGCA_BENCH_DEVICE pins every rank to one card, PGCA_BENCH_BACKEND=gloo
    # swaps the transport; the driver's runs use neither (one rank per GPU over RCCL)
    dev_index = int(os.environ.get("PGCA_BENCH_DEVICE", local_rank))
    torch.cuda.set_device(dev_index)
    dp = DataParallel.init_from_env(os.environ.get("PGCA_BENCH_BACKEND", "nccl"))
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    dp.compress_bf16 = bool(args.allreduce_bf16)

    for kv in args.set:
        name, _, val = kv.partition("=")
        hip.set_option(name, int(val))
    B, S, beta = args.pairs_per_gpu, args.seq_len, 0.1
    log(f"init: world={world} pairs/gpu={B} S={S}")
    arch = make_arch(args.vision_model, args.text_model, 512)
    model = PreferenceGuidedCaptioningModel(args.vision_model, args.text_model, 512, temperature=0.5,
                                            freeze_vision_backbone=not args.trainable_vision, device=dev, seed=42)
    stage1 = args.stage == 1
    if stage1:
        ref = None
        step = ContrastiveStep(model.store, model.ws, model.vision_encoder.tower, model.vision_encoder.head,
                               model.text_encoder.engine, temperature=0.5, dp=dp, global_negatives=True,
                               dropout=DropoutPlan(args.dropout, base_seed=42 + dp.rank), packed=not args.padded)
        segs = [model.store.segments["vision_head"], model.store.segments["text_head"],
                model.store.segments["text_tower"]]
        trunk = model.text_encoder.engine.trunk
    else:
        ref = None if args.reference_free else ReferencePolicy(model.store, model.ws)
        step = DPOStep(model.store, model.ws, model.vision_encoder.tower, model.vision_encoder.head,
                       model.caption_decoder.engine, beta=beta, reference_free=args.reference_free, ref=ref,
                       ref_side_stream=args.ref_side_stream, dropout=DropoutPlan(args.dropout, base_seed=42 + dp.rank),
                       packed=not args.padded)
        segs = [model.store.segments["vision_head"], model.store.segments["decoder"]]
        trunk = model.caption_decoder.engine.trunk
    if args.trainable_vision:
        segs.insert(0, model.store.segments["vit"])
    opt = FusedOptimizer(segs, lr=5e-5 if stage1 else 1e-5, weight_decay=0.01, max_grad_norm=1.0, warmup_steps=500,
                         total_steps=100000, sched_stride=dp.world)
    trunk.overlap_wgrad = not args.no_wgrad_overlap
    reducer = OverlappedTrunkReducer(dp, trunk, group=4)
    reducer.arm()
    log(f"model + optimizer ready ({model.store.num_params() / 1e6:.1f} M params)")

    nbatch = 4  # distinct resident batches, cycled
    raw = [synthetic_batch(B, S, arch.gpt.base_vocab, arch.gpt.base_vocab, seed=1234 + dp.rank + 1000 * i)
           for i in range(nbatch)]
    if stage1:
        raw = [{"image": r["image"], "caption_ids": r["preferred_ids"], "caption_mask": r["preferred_mask"]} for r in raw]
    prep = ContrastiveStep.prepare if stage1 else DPOStep.prepare
    feeder = None
    if args.with_input_path:
        from pgca_amd.input import BatchPrefetcher

        class _Cycle:
            def __init__(self, n):
                self.n = n

            def __len__(self):
                return self.n

            def __iter__(self):
                return (raw[i % nbatch] for i in range(self.n))

        feeder = iter(BatchPrefetcher(_Cycle(args.warmup + args.steps), prep, dev, depth=2))
        batches = None
    else:
        batches = [prep(r, dev) for r in raw]

    probe = GemmProbe(hip.NN)
    if not args.no_probe and dp.rank == 0:
        hip.gemm_probe = probe

    last = [None]

    def one_step(i):
        p = next(feeder) if feeder is not None else batches[i % nbatch]
        opt.zero_grad()
        if stage1:
            last[0] = step.loss_and_grads(p["image"], p["ids"], p["mask"], pack=p.get("pack"))
        else:
            last[0] = step.loss_and_grads(p["image"], p["seq"])
        reducer.finish(other_segments=segs[:-1])
        opt.step(grad_scale=1.0 / dp.world)

    log("batches resident; warm-up")
    for i in range(args.warmup):
        one_step(i)
        torch.cuda.synchronize()
        log(f"warm-up step {i} done, loss={float(last[0]):.5f}")
    torch.cuda.synchronize()
    dp.barrier()
    torch.cuda.synchronize()
    probe.enabled = True
    t0 = time.perf_counter()
    for i in range(args.steps):
        one_step(args.warmup + i)
    torch.cuda.synchronize()
    dp.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    probe.enabled = False
    dt = dp.all_reduce_max_scalar(dt, dev)
    log(f"timed region: {args.steps} steps in {dt:.3f} s -> {B * dp.world * args.steps / dt:.1f} pairs/s")
    # the reported loss is the same number on every rank: Stage 1 (global negatives) is already reduced inside the
    # step, Stage 2's local means are averaged here (outside the timed region)
    loss_t = last[0].detach().clone().reshape(1)
    if not stage1 and dp.world > 1:
        loss_t = dp.all_reduce_sum(loss_t) / dp.world
    loss_val = float(loss_t)
    st = opt.state()

    if dp.rank == 0:
        pairs = B * dp.world * args.steps
        # algorithmic FLOPs per pair: decoder 719.5 MFLOP/token fwd (full SxS attention counted)
        g = arch.gpt
        per_tok = 24 * g.hidden ** 2 * g.layers + 4 * S * g.hidden * g.layers + 2 * g.hidden * arch.dec_vocab
        seq_fwd = per_tok * S
        v = arch.vit
        vit_fwd = (24 * v.hidden ** 2 * v.layers + 4 * v.tokens * v.hidden * v.layers) * v.tokens \
            + 2 * v.patch_dim * v.hidden * (v.tokens - 1)
        n_fwd = 6 if args.reference_free else 8      # policy 2 x (fwd + 2 bwd) [+ reference 2 x fwd]
        flop_pair = n_fwd * seq_fwd + vit_fwd
        if stage1:  # text tower without LM head, fwd + 2 x bwd, one ViT forward (245.6 GFLOP/pair)
            flop_pair = 3 * (24 * g.hidden ** 2 * g.layers + 4 * S * g.hidden * g.layers) * S + vit_fwd
        if args.trainable_vision:
            flop_pair += 2 * vit_fwd     # the tower's backward
        # EXECUTED FLOPs per pair: the same formulas on the rows the kernels were actually launched on - the packed token
        # rows (incl. filler) for the trunk, l x l attention per sequence, and the scored rows only for the LM head
        # (forward x 2 policies; backward = logits recompute + data gradient + weight gradient)
        import numpy as _np
        ex_rows, ex_att, ex_lm = [], [], []
        for r in raw:
            keys = ("caption_mask",) if stage1 else ("preferred_mask", "rejected_mask")
            m = _np.concatenate([r[k].numpy() for k in keys])
            lens = _np.array([0 if not x.any() else int(_np.nonzero(x)[0].max()) + 1 for x in m])
            n = int(lens.sum())
            ex_rows.append(m.size if args.padded else (n + 63) // 64 * 64)
            ex_att.append(float(m.shape[0]) * S * S if args.padded else float((lens.astype(_np.float64) ** 2).sum()))
            ex_lm.append(int((m[:, 1:] != 0).sum()))
        rows_b, att_b, lm_b = (float(_np.mean(x)) for x in (ex_rows, ex_att, ex_lm))
        trunk_tok = 24 * g.hidden ** 2 * g.layers
        att_unit = 4 * g.hidden * g.layers
        n_trunk = 3 if stage1 else (3 if args.reference_free else 4)    # fwd + 2 x bwd (+ reference fwd)
        ex_step = n_trunk * (trunk_tok * rows_b + att_unit * att_b) + vit_fwd * B * (3 if args.trainable_vision else 1)
        if not stage1:
            ex_step += (4 if args.reference_free else 5) * 2 * g.hidden * arch.dec_vocab * lm_b
        exec_pair = ex_step / B
        frozen = "trainable" if args.trainable_vision else "frozen"
        short = {"openai/clip-vit-base-patch32": "CLIP-ViT-B/32", "openai/clip-vit-base-patch16": "CLIP-ViT-B/16",
                 "openai/clip-vit-large-patch14": "CLIP-ViT-L/14", "gpt2": "GPT-2", "gpt2-medium": "GPT-2-M",
                 "gpt2-large": "GPT-2-L", "gpt2-xl": "GPT-2-XL"}
        tag = (f"{short.get(args.vision_model, args.vision_model)}+{short.get(args.text_model, args.text_model)} "
               f"seq{S}" + (" trainable tower (secondary)" if args.trainable_vision else ""))
        res = {
            "metric": (f"Stage-1 NT-Xent image-caption pairs/sec, {tag} (secondary)" if stage1
                       else f"DPO preference-pairs/sec, {tag}"),
            "value": pairs / dt, "unit": "pairs/s", "n_gpus": dp.world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * dt / args.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "bf16", "data": "synthetic" if feeder is None else
            "synthetic, fed from HOST memory every step (pinned staging + async H2D + device-side index preparation, "
            "prefetch depth 2) - not the contract's HBM-resident number",
            "config": {"workload": (("Stage-1 NT-Xent step (tau 0.5, global negatives over the ranks), "
                                     f"{args.vision_model} ({frozen}) + {args.text_model} text tower, seq_len {S}, "
                                     "bf16 MFMA / f32 accumulate+master, AdamW+clip") if stage1 else
                                    ("Stage-2 DPO step, " + ("2-forward reference-free" if args.reference_free
                                                             else "4-forward policy/reference x chosen/rejected")
                                     + f", {args.vision_model} ({frozen}) + {args.text_model} decoder, seq_len {S}, "
                                     "beta 0.1, bf16 MFMA / f32 accumulate+master, AdamW+clip")),
                       "pairs_per_gpu": B, "global_pairs_per_step": B * dp.world, "seq_len": S,
                       "parallelism": f"dp{dp.world}",
                       "grad_allreduce": ("none (1 GPU)" if dp.world == 1 else
                                          ("bf16-compressed" if args.allreduce_bf16 else "f32") +
                                          " SUM over RCCL, per 4 decoder layers on a side stream during backward"),
                       "dropout": (f"train mode, p={args.dropout} at the reference's sites (fused, counter-based, "
                                   "replayed in backward); reference policy in eval mode") if args.dropout > 0
                       else "identity (p=0)"},
            "algorithmic_gflop_per_pair": flop_pair / 1e9,
            "executed_gflop_per_pair": exec_pair / 1e9,
            "rows": ("all B x S padded positions" if args.padded else
                     f"packed: {rows_b:.0f} of {len(raw[0]['image']) * (1 if stage1 else 2) * S} token rows per step"),
            "step_tflops_per_gpu": flop_pair * B / (dt / args.steps) / 1e12,
            "executed_tflops_per_gpu": exec_pair * B / (dt / args.steps) / 1e12,
            "loss": loss_val, "grad_norm": st["grad_norm"], "opt_steps": st["step"],
            # what the process group itself reports (N ranks over RCCL when the backend is nccl)
            "process_group": {"world_size": torch.distributed.get_world_size() if torch.distributed.is_initialized() else 1,
                              "backend": torch.distributed.get_backend() if torch.distributed.is_initialized() else "none"},
        }
        ps = probe.summary()
        if ps:
            default_cfg = (not stage1 and args.vision_model == "openai/clip-vit-base-patch32"
                           and args.text_model == "gpt2-medium" and S == 128 and not args.reference_free
                           and not args.trainable_vision)
            pmc = pmc_from_profile(B, "gemm256s_kernel<0, 1>") if default_cfg else None
            res["roofline"] = {"bound": "mfma", "achieved": ps["tflops"], "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s",
                               "frac": ps["tflops"] / PEAK_BF16_TFLOPS,
                               "traffic": pmc.get("hbm_bytes") if pmc else None,
                               "mfma_busy": pmc.get("mfma_busy") if pmc else None,
                               "traffic_note": ("NOT measured in this run: HBM-side bytes per launch (read = 2 x FETCH_SIZE on gfx950, + "
                                                "WRITE_SIZE) and matrix-pipe busy fraction (SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x "
                                                f"SQ_BUSY_CYCLES / 32)) copied from the committed rocprofv3 --pmc passes of this command, "
                                                f"{PMC_FILE}" + (f" (taken at commit {pmc.get('commit')})" if pmc and pmc.get("commit")
                                                                 else "")) if pmc else
                               f"null: no committed PMC passes for this workload ({PMC_FILE})",
                               "kernel": "gemm256s_kernel<0, 1> (every launch of the NN 256x256 LDS-DMA GEMM, phase-staggered schedule, in the timed steps)", "launches": ps["launches"],
                               "avg_launch_us": ps["avg_us"]}
        if ps and dp.world == 1:
            # measured ceiling beside the vendor peak: the library GEMM on this GPU, outside the timed region.
            # Yardstick only - nothing in the product path calls a BLAS library.
            try:
                n = 8192
                xa = (torch.rand(n, n, device=dev) * 2 - 1).bfloat16()
                xb = (torch.rand(n, n, device=dev) * 2 - 1).bfloat16()
                for _ in range(2):
                    torch.matmul(xa, xb)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                torch.cuda.synchronize()
                e0.record()
                for _ in range(10):
                    torch.matmul(xa, xb)
                e1.record()
                torch.cuda.synchronize()
                lib = 10 * 2.0 * n ** 3 / (e0.elapsed_time(e1) * 1e-3) / 1e12
                res["roofline"]["library_ceiling_tflops"] = lib
                res["roofline"]["frac_of_library_ceiling"] = ps["tflops"] / lib
                res["roofline"]["library_ceiling_note"] = "torch.matmul bf16 8192^3 (hipBLASLt), random operands, same GPU"
                del xa, xb
            except Exception as exc:  # noqa: BLE001 - a missing BLAS must not break the bench line
                res["roofline"]["library_ceiling_note"] = f"unavailable: {exc}"
        if dp.world == 1 and not args.no_cpu_baseline and not stage1:
            hip.gemm_probe = None
            log("cpu baseline (oracle on host cores) ...")
            res["cpu_baseline"] = cpu_baseline(model, arch, S, beta, args.cpu_pairs)
        print(json.dumps(res), flush=True)
    dp.barrier()
    if torch.distributed.is_initialized():
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
