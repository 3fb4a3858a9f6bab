#!/usr/bin/env python3
"""Micro-benchmark of the MFMA GEMM on the shapes of the GPT-2-M DPO step (random bf16 data).

    python tools/gemm_bench.py [--tile 128|256] [--iters 20] [--only NAME]

Prints TFLOP/s per shape; used with rocprofv3 --pmc to read LDS-conflict / MFMA-busy counters.
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pgca_amd import hip  # noqa: E402

SHAPES = [  # name, layout, M, N, K, epilogue
    ("qkv_fwd", hip.NN, 16384, 3072, 1024, "bias"),
    ("proj_fwd", hip.NN, 16384, 1024, 1024, "res"),
    ("fc_fwd", hip.NN, 16384, 4096, 1024, "gelu"),
    ("fc2_fwd", hip.NN, 16384, 1024, 4096, "res"),
    ("fc2_dgrad", hip.NT, 16384, 4096, 1024, "dgelu"),
    ("fc_dgrad", hip.NT, 16384, 1024, 4096, "none"),
    ("fc_wgrad", hip.TN, 1024, 4096, 16384, "acc"),
    ("fc2_wgrad", hip.TN, 4096, 1024, 16384, "acc"),
    ("qkv_wgrad", hip.TN, 1024, 3072, 16384, "acc"),
    ("lm_head", hip.NT, 9216, 50260, 1024, "rowstats"),
    # K-strided operands that fit the L2s / Infinity Cache: the TN main loop without HBM in the way
    ("tn_smallK", hip.TN, 4096, 4096, 2048, "acc"),
    ("nn_smallM", hip.NN, 4096, 4096, 2048, "bias"),
]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--tile", type=int, default=0)
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--only", default="")
    ap.add_argument("--ring", type=int, default=-1)
    ap.add_argument("--rows", type=int, default=16384)
    ap.add_argument("--timing", action="store_true", help="needs gemm_wide.o built with -DPGCA_GEMM_TIMING")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    hip.load()
    if args.tile:
        hip.set_option("gemm_tile", args.tile)
    if args.ring >= 0:
        hip.set_option("gemm_schedule", args.ring)
    for name, layout, M, N, K, epi in SHAPES:
        if args.only and args.only != name:
            continue
        if M == 16384:
            M = args.rows
        if K == 16384:
            K = args.rows
        g = torch.Generator(device="cpu").manual_seed(1)
        a_shape = (M, K) if layout != hip.TN else (K, M)
        b_shape = (N, K) if layout == hip.NT else (K, N)
        A = (torch.rand(a_shape, generator=g) * 2 - 1).to(dev).bfloat16()
        B = (torch.rand(b_shape, generator=g) * 2 - 1).to(dev).bfloat16()
        kw = {}
        if epi in ("bias", "gelu", "res"):
            kw["bias"] = torch.zeros(N, device=dev)
        if epi in ("bias", "none"):
            kw["out_bf16"] = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
        if epi == "gelu":
            kw.update(epilogue=hip.EPI_GELU_NEW, out_bf16=torch.empty(M, N, dtype=torch.bfloat16, device=dev),
                      aux_out=torch.empty(M, N, dtype=torch.bfloat16, device=dev))
        if epi == "dgelu":
            kw.update(epilogue=hip.EPI_DGELU_NEW, out_bf16=torch.empty(M, N, dtype=torch.bfloat16, device=dev),
                      aux_in=torch.zeros(M, N, dtype=torch.bfloat16, device=dev))
        if epi == "res":
            r = torch.zeros(M, N, device=dev)
            kw.update(residual=r, out_f32=torch.empty(M, N, device=dev))
        if epi == "acc":
            kw.update(out_f32=torch.zeros(M, N, device=dev), accumulate=True)
        if epi == "rowstats":
            nparts = 2 * ((N + 127) // 128)
            kw.update(epilogue=hip.EPI_ROWSTATS, targets=torch.zeros(M, dtype=torch.int64, device=dev),
                      stat_max=torch.empty(M, nparts, device=dev), stat_sum=torch.empty(M, nparts, device=dev),
                      stat_ld=nparts, target_val=torch.empty(M, device=dev))

        if args.timing and epi != "rowstats":
            tbuf = torch.zeros(((M + 255) // 256) * ((N + 255) // 256) * 8 * 8, device=dev)
            kw["stat_max"] = tbuf
            kw["stat_sum"] = torch.zeros_like(tbuf)

        def run():
            hip.gemm(A, B, M, N, K, layout, **kw)

        for _ in range(3):
            run()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(args.iters):
            run()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / args.iters
        if args.timing and "stat_max" in kw and epi != "rowstats":
            tb = kw["stat_max"].view(-1, 8).cpu()
            m = tb.mean(0)
            if args.ring == 4:
                steps = float(m[5])
                print(f"   per-step ticks: mma {m[0] / steps:.1f} vmwait {m[6] / steps:.1f} barrier {m[1] / steps:.1f} "
                      f"dma-issue {m[2] / steps:.1f} lgkm {m[3] / steps:.1f} total {m[4] / steps:.1f}")
            else:
                ph = kw["stat_sum"].view(-1, 8).cpu()
                if float(ph.abs().sum()) > 0:  # gemm_phase.o built with -DPGCA_GEMM_TIMING2
                    g0, g1 = ph.view(-1, 8, 8)[:, :4].mean((0, 1)), ph.view(-1, 8, 8)[:, 4:].mean((0, 1))
                    names = ["L0", "B1+w", "M0", "B2", "L1", "B1'+w", "M1", "B2'"]
                    print("   phase clocks per 32-deep tile  group0: " + " ".join(f"{n} {x:.0f}" for n, x in zip(names, g0)))
                    print("                                  group1: " + " ".join(f"{n} {x:.0f}" for n, x in zip(names, g1)))
                print(f"   per-workgroup clocks: prologue {m[0]:.0f} main loop {m[1]:.0f} "
                      f"({m[1] / max(float(m[3]), 1):.0f}/K-tile) epilogue {m[2]:.0f}")
                w0 = tb.view(-1, 8, 8)[:, 0]          # wave 0 of every workgroup (last launch)
                ok = w0[:, 7] > 0
                if bool(ok.any()):     # in-kernel shader clock: d s_memtime / d s_memrealtime x 100 MHz, median over workgroups
                    ghz = (w0[ok, 6].double() / w0[ok, 7].double() * 0.1).median()
                    print(f"   in-kernel clock (s_memtime / s_memrealtime x 100 MHz, median of {int(ok.sum())} workgroups): "
                          f"{float(ghz):.3f} GHz")
                st, en = w0[:, 4].double(), w0[:, 5].double()
                if float(en.max()) > 0:
                    t0 = float(st.min())
                    wrap = 2.0 ** 28
                    st, en = (st - t0) % wrap, (en - t0) % wrap
                    span = float(en.max())
                    busy = float((en - st).sum()) / 256.0
                    import numpy as np
                    q = np.percentile(st.numpy(), [0, 25, 50, 75, 100])
                    print(f"   launch span {span:.0f} clocks; sum of workgroup lifetimes / 256 CUs = {busy:.0f} "
                          f"({100 * busy / span:.0f} % of the span); start-time quartiles {q.round(-2)}; "
                          f"ideal MFMA clocks {2.0 * M * N * K / (256 * 4 * 512 * 2):.0f}")
        print(f"{name:10s} layout={layout} M={M:6d} N={N:6d} K={K:6d} {epi:8s} {us:9.1f} us  "
              f"{2.0 * M * N * K / us / 1e6:8.1f} TFLOP/s", flush=True)


if __name__ == "__main__":
    main()
