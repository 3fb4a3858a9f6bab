#!/usr/bin/env python3
"""Attention forward / backward micro-benchmark (random bf16 data, ragged right padding as the synthetic workload).

    python tools/attn_bench.py [--seqs 256] [--heads 16] [--S 128] [--causal 1] [--drop 0.1]

Run once as is and once with PGCA_ATTN_TILED=1 to compare the single-tile and the key-tiled kernels at S <= 128.
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pgca_amd import hip  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seqs", type=int, default=256)
    ap.add_argument("--heads", type=int, default=16)
    ap.add_argument("--S", type=int, default=128)
    ap.add_argument("--causal", type=int, default=1)
    ap.add_argument("--drop", type=float, default=0.1)
    ap.add_argument("--iters", type=int, default=30)
    ap.add_argument("--packed", type=int, default=0, help="1: packed rows (cu offsets, the training layout), lengths 16..S")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    hip.load()
    B, S, heads = args.seqs, args.S, args.heads
    H = heads * 64
    g = torch.Generator().manual_seed(0)
    qkv = torch.randn(B * S, 3 * H, generator=g).to(dev).bfloat16()
    dout = torch.randn(B * S, H, generator=g).to(dev).bfloat16()
    lens = torch.randint(16, S + 1, (B,), generator=g)
    mask = (torch.arange(S)[None] < lens[:, None]).int().to(dev) if args.causal else None
    out = torch.empty(B * S, H, dtype=torch.bfloat16, device=dev)
    lse = torch.empty(B, heads, S, device=dev)
    dqkv = torch.empty(B * S, 3 * H, dtype=torch.bfloat16, device=dev)
    d = hip.drop_args(77, args.drop)
    cu = None
    if args.packed:
        cu = torch.zeros(B + 1, dtype=torch.int32)
        cu[1:] = torch.cumsum(lens, 0)
        cu = cu.to(dev)
        mask = None

    def timeit(fn):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(args.iters):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) * 1e3 / args.iters

    tf = timeit(lambda: hip.attention_fwd(qkv, mask, B, S, heads, bool(args.causal), out, lse, drop=d, cu=cu))
    tb = timeit(lambda: hip.attention_bwd(qkv, out, dout, lse, mask, B, S, heads, bool(args.causal), dqkv, drop=d, cu=cu))
    rows = int(lens.sum()) if args.packed else B * S
    fl = 4.0 * heads * 64 * (float((lens.double() ** 2).sum()) if args.packed else B * S * S)
    byt_f = rows * H * 2 * 4 + rows * heads * 4
    byt_b = rows * H * 2 * (3 + 1 + 1 + 3)
    print(f"packed={args.packed} tiled={os.environ.get('PGCA_ATTN_TILED', '0')} B={B} heads={heads} S={S} causal={args.causal} drop={args.drop}: "
          f"fwd {tf:7.1f} us ({fl / tf / 1e6:6.1f} TF/s, {byt_f / tf / 1e6:5.2f} TB/s)   "
          f"bwd {tb:7.1f} us ({2.5 * fl / tb / 1e6:6.1f} TF/s, {byt_b / tb / 1e6:5.2f} TB/s)")


if __name__ == "__main__":
    main()
