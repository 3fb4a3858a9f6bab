"""Yardstick only (not a product path): time torch.matmul (hipBLASLt / rocBLAS) on the GEMM shapes of the DPO step in the
three operand layouts the step uses, so the hand-written kernels are judged against a known-good library number measured
on the same box, not against a guess.

    python tools/blas_yardstick.py [--rows 73728]
"""
import argparse

import torch


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=73728, help="packed token rows of a 512-pair step")
    R = ap.parse_args().rows
    dev = torch.device("cuda:0")
    shapes = [  # name, layout, M, N, K
        ("qkv_fwd", "NN", R, 3072, 1024), ("proj_fwd", "NN", R, 1024, 1024), ("fc_fwd", "NN", R, 4096, 1024),
        ("fc2_fwd", "NN", R, 1024, 4096),
        ("qkv_dgrad", "NT", R, 1024, 3072), ("fc2_dgrad", "NT", R, 4096, 1024), ("fc_dgrad", "NT", R, 1024, 4096),
        ("qkv_wgrad", "TN", 1024, 3072, R), ("proj_wgrad", "TN", 1024, 1024, R), ("fc_wgrad", "TN", 1024, 4096, R),
        ("fc2_wgrad", "TN", 4096, 1024, R),
        ("square8k", "NN", 8192, 8192, 8192)]
    for name, lay, M, N, K in shapes:
        if lay == "NN":
            a = (torch.rand(M, K, device=dev) * 2 - 1).bfloat16()
            b = (torch.rand(K, N, device=dev) * 2 - 1).bfloat16()
        elif lay == "NT":                                   # B stored [N, K]
            a = (torch.rand(M, K, device=dev) * 2 - 1).bfloat16()
            b = (torch.rand(N, K, device=dev) * 2 - 1).bfloat16().t()
        else:                                               # A stored [K, M]
            a = (torch.rand(K, M, device=dev) * 2 - 1).bfloat16().t()
            b = (torch.rand(K, N, device=dev) * 2 - 1).bfloat16()
        for _ in range(3):
            torch.matmul(a, b)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            torch.matmul(a, b)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / 20
        print(f"{name:10s} {lay} M={M:6d} N={N:6d} K={K:6d}  {us:9.1f} us  {2.0 * M * N * K / us / 1e6:8.1f} TFLOP/s "
              f"(torch.matmul, bf16 out)", flush=True)


if __name__ == "__main__":
    main()
