#!/usr/bin/env python3
"""Sweep of a dispatch knob (pgca_set_option) over the forward GEMM shapes of the bench step
(GPT-2-M decoder block at M = 32768 rows, fused epilogues as the engine issues them), rounds interleaved
in one process (cdna guide rule 24).

    python tools/gemm_sweep.py --option gemm_stagger --values 0,2,4,8,12 [--rows 32768] [--rounds 5]
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pgca_amd import hip  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--option", default="gemm_stagger")
    ap.add_argument("--values", default="0,2,4,8,12")
    ap.add_argument("--rows", type=int, default=32768)
    ap.add_argument("--hidden", type=int, default=1024)
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--library", action="store_true",
                    help="add a column: torch.matmul (hipBLASLt) on the same operands, PLAIN product without the epilogue")
    args = ap.parse_args()
    vals = [int(v) for v in args.values.split(",")]
    dev = torch.device("cuda:0")
    hip.load()
    M, H = args.rows, args.hidden
    g = torch.Generator().manual_seed(1)
    rb = lambda *s: (torch.rand(*s, generator=g) * 2 - 1).to(dev).bfloat16()  # noqa: E731
    x, x4 = rb(M, H), rb(M, 4 * H)
    wqkv, wo, wfc, wpr = rb(H, 3 * H), rb(H, H), rb(H, 4 * H), rb(4 * H, H)
    res = torch.randn(M, H, device=dev)
    out32 = torch.empty(M, H, device=dev)
    o3, o4, o4b = (torch.empty(M, 3 * H, dtype=torch.bfloat16, device=dev), torch.empty(M, 4 * H, dtype=torch.bfloat16, device=dev),
                   torch.empty(M, 4 * H, dtype=torch.bfloat16, device=dev))
    bias = {n: torch.zeros(n, device=dev) for n in (H, 3 * H, 4 * H)}
    drop = hip.drop_args(123, 0.1)
    shapes = {
        "qkv  (bias->bf16)": (lambda: hip.gemm(x, wqkv, M, 3 * H, H, hip.NN, bias=bias[3 * H], out_bf16=o3), 2.0 * M * 3 * H * H),
        "proj (res f32+drop)": (lambda: hip.gemm(x, wo, M, H, H, hip.NN, bias=bias[H], residual=res, out_f32=out32, drop=drop), 2.0 * M * H * H),
        "fc   (gelu, 2 outs)": (lambda: hip.gemm(x, wfc, M, 4 * H, H, hip.NN, epilogue=hip.EPI_GELU_NEW, bias=bias[4 * H], out_bf16=o4, aux_out=o4b), 2.0 * M * 4 * H * H),
        "fc   (gelu, 1 out)": (lambda: hip.gemm(x, wfc, M, 4 * H, H, hip.NN, epilogue=hip.EPI_GELU_NEW, bias=bias[4 * H], out_bf16=o4), 2.0 * M * 4 * H * H),
        "fc2  (res f32+drop)": (lambda: hip.gemm(x4, wpr, M, H, 4 * H, hip.NN, bias=bias[H], residual=res, out_f32=out32, drop=drop), 2.0 * M * 4 * H * H),
        "dgelu NT dgrad": (lambda: hip.gemm(x, wpr, M, 4 * H, H, hip.NT, epilogue=hip.EPI_DGELU_NEW, aux_in=o4b, out_bf16=o4), 2.0 * M * 4 * H * H),
        "plain NT dgrad K=4H": (lambda: hip.gemm(x4, wfc, M, H, 4 * H, hip.NT, out_bf16=o3[:, :H].contiguous()), 2.0 * M * 4 * H * H),
    }
    lib = {}
    if args.library:   # the library's plain GEMM (bf16 out) on the same shapes: what the fused kernels are held against
        wt = {"NT_fc": wpr.t().contiguous(), "NT_4h": wfc.t().contiguous()}
        lib = {
            "qkv  (bias->bf16)": lambda: torch.matmul(x, wqkv),
            "proj (res f32+drop)": lambda: torch.matmul(x, wo),
            "fc   (gelu, 2 outs)": lambda: torch.matmul(x, wfc),
            "fc   (gelu, 1 out)": lambda: torch.matmul(x, wfc),
            "fc2  (res f32+drop)": lambda: torch.matmul(x4, wpr),
            "dgelu NT dgrad": lambda: torch.matmul(x, wt["NT_fc"]),
            "plain NT dgrad K=4H": lambda: torch.matmul(x4, wt["NT_4h"]),
        }
    times = {(n, v): [] for n in shapes for v in vals}
    ltimes = {n: [] for n in lib}
    for rnd in range(args.rounds + 1):
        for v in vals:
            hip.set_option(args.option, v)
            for n, (fn, _) in shapes.items():
                fn()
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(args.iters):
                    fn()
                e1.record()
                torch.cuda.synchronize()
                if rnd:
                    times[n, v].append(e0.elapsed_time(e1) * 1e3 / args.iters)
    for n, fn in lib.items():
        for rnd in range(args.rounds + 1):
            fn()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(args.iters):
                fn()
            e1.record()
            torch.cuda.synchronize()
            if rnd:
                ltimes[n].append(e0.elapsed_time(e1) * 1e3 / args.iters)
    print(f"{args.option:>22s} " + " ".join(f"{v:>14d}" for v in vals) + ("   hipBLASLt plain" if lib else ""))
    for n, (_, fl) in shapes.items():
        cells = []
        for v in vals:
            t = sorted(times[n, v])[len(times[n, v]) // 2]
            cells.append(f"{t:7.1f}us {fl / t / 1e6:5.0f}T")
        if lib:
            t = sorted(ltimes[n])[len(ltimes[n]) // 2]
            cells.append(f"{t:7.1f}us {fl / t / 1e6:5.0f}T")
        print(f"{n:>22s} " + " ".join(cells), flush=True)


if __name__ == "__main__":
    main()
