"""Data-parallel engine on 2 CPU ranks (gloo): bucketed gradient all-reduce == single-process sum,
embedding all-gather ordering, global-negative NT-Xent == single-process loss on the concatenated batch."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import restatement as R
from pgca_amd.dist import DataParallel, bucket_plan


def test_bucket_plan_covers_exactly():
    assert bucket_plan(0, 8) == []
    assert bucket_plan(10, 4) == [(0, 4), (4, 8), (8, 10)]
    assert bucket_plan(8, 100) == [(0, 8)]
    for n, b in ((1000003, 4096), (64, 64), (65, 64)):
        plan = bucket_plan(n, b)
        assert plan[0][0] == 0 and plan[-1][1] == n and all(p[1] == q[0] for p, q in zip(plan, plan[1:]))


class FakeSeg:
    def __init__(self, grad):
        self.grad, self.numel = grad, grad.numel()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        dp = DataParallel(bucket_elems=1000)
        assert dp.world == world and dp.rank == rank
        # 1) bucketed all-reduce of flat gradient buffers (+ the overlapped range form)
        g = torch.Generator().manual_seed(100 + rank)
        segs = [FakeSeg(torch.randn(4097, generator=g)), FakeSeg(torch.randn(333, generator=g))]
        mine = [s.grad.clone() for s in segs]
        dp.all_reduce_grads(segs)
        dp.all_reduce_range(mine[0], 0, 2000)
        dp.all_reduce_range(mine[0], 2000, 4097)
        dp.join()
        # 2) embedding all-gather is rank-major
        x = torch.full((3, 4), float(rank))
        allx = dp.all_gather_rows(x)
        # 3) global-negative NT-Xent: each rank owns B rows of S and of S^t
        gg = torch.Generator().manual_seed(7)
        img = torch.nn.functional.normalize(torch.randn(world * 5, 16, generator=gg), dim=-1)
        txt = torch.nn.functional.normalize(torch.randn(world * 5, 16, generator=gg), dim=-1)
        lo, hi = dp.shard(world * 5)
        ia, ta = dp.all_gather_rows(img[lo:hi].contiguous()), dp.all_gather_rows(txt[lo:hi].contiguous())
        lab = torch.arange(lo, hi)
        part = (torch.nn.functional.cross_entropy(img[lo:hi] @ ta.t() / 0.5, lab, reduction="sum")
                + torch.nn.functional.cross_entropy(txt[lo:hi] @ ia.t() / 0.5, lab, reduction="sum")) / (2 * world * 5)
        total = dp.all_reduce_sum(part.clone().reshape(1))
        mx = dp.all_reduce_max_scalar(float(rank), "cpu")
        # numpy: pickled BY VALUE (a torch tensor travels as a file descriptor of the sender, which may have exited)
        q.put((rank, [s.grad.numpy() for s in segs], mine[0].numpy(), allx.numpy(), float(total),
               float(R.nt_xent(img, txt, 0.5)), mx))
    finally:
        dist.destroy_process_group()


def test_two_rank_gloo_collectives():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    ref = []
    for r in range(world):
        g = torch.Generator().manual_seed(100 + r)
        ref.append([torch.randn(4097, generator=g), torch.randn(333, generator=g)])
    want = [ref[0][0] + ref[1][0], ref[0][1] + ref[1][1]]
    for rank, grads, ranged, allx, total, single, mx in res:
        grads, ranged, allx = [torch.from_numpy(g) for g in grads], torch.from_numpy(ranged), torch.from_numpy(allx)
        assert torch.allclose(grads[0], want[0]) and torch.allclose(grads[1], want[1])
        assert torch.allclose(ranged, want[0])
        assert torch.equal(allx, torch.cat([torch.zeros(3, 4), torch.ones(3, 4)]))
        assert abs(total - single) <= 1e-5       # sharded global-negative loss == single-process loss
        assert mx == 1.0


def _worker_bf16(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        dp = DataParallel(bucket_elems=3000, compress_bf16=True)
        g = torch.Generator().manual_seed(500 + rank)
        # gradient-like: wide dynamic range across the buffer (1e-6 .. 1)
        x = torch.randn(10007, generator=g) * torch.logspace(-6, 0, 10007)
        seg = FakeSeg(x.clone())
        dp.all_reduce_grads([seg])
        ranged = x.clone()
        dp.all_reduce_range(ranged, 0, 10007)
        dp.join()
        q.put((rank, seg.grad.numpy(), ranged.numpy()))
    finally:
        dist.destroy_process_group()


def test_bf16_compressed_all_reduce_keeps_the_gradient_direction():
    """mi355x.allreduce_bf16: buckets are summed on the wire in bf16 (half the xGMI bytes).  Against the f32 sum the
    result must keep cosine >= 0.9999 and every element within bf16 rounding of the exact sum; both ranks identical."""
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_bf16, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    xs = []
    for r in range(world):
        g = torch.Generator().manual_seed(500 + r)
        xs.append(torch.randn(10007, generator=g) * torch.logspace(-6, 0, 10007))
    exact = (xs[0].double() + xs[1].double())
    res = [(r, torch.from_numpy(a), torch.from_numpy(b)) for r, a, b in res]
    for rank, grad, ranged in res:
        for got in (grad, ranged):
            c = float(got.double() @ exact / (got.double().norm() * exact.norm()))
            assert c >= 0.9999, c
            # each operand is rounded to bf16 (2^-9 relative) and so is the sum
            bound = (xs[0].abs() + xs[1].abs()).double() * 2.0 ** -8 + exact.abs() * 2.0 ** -8 + 1e-30
            assert bool(((got.double() - exact).abs() <= bound).all())
    assert torch.equal(res[0][1], res[1][1]) and torch.equal(res[0][2], res[1][2])
