#!/usr/bin/env python3
"""Throughput of the device-side image transforms: B decoded RGB images H x W -> 224 x 224 f32 through the validation
transform (pgca_image_preprocess) and the training transform with augmentations (pgca_image_train_transform; the host
time of drawing the parameters and building the per-image tap tables is reported separately).

    python tools/image_bench.py [--batch 256] [--height 375] [--width 500]
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pgca_amd.input import GpuImageProcessor  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--height", type=int, default=375)
    ap.add_argument("--width", type=int, default=500)
    ap.add_argument("--size", type=int, default=224)
    ap.add_argument("--iters", type=int, default=20)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    x = torch.randint(0, 256, (a.batch, a.height, a.width, 3), dtype=torch.uint8, device=dev)
    proc = GpuImageProcessor(a.size, device=dev)
    out = torch.empty(a.batch, 3, a.size, a.size, device=dev)
    for _ in range(3):
        proc.process_batch(x, out=out)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(a.iters):
        proc.process_batch(x, out=out)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / a.iters
    # algorithmic bytes: read the image, write + read the horizontal intermediate, write the f32 tensor
    b = a.batch * (a.height * a.width * 3 + 2 * a.height * a.size * 3 + 3 * a.size * a.size * 4)
    print(f"val   {a.batch} x {a.height}x{a.width} -> {a.size}: {ms * 1e3:.1f} us, {a.batch / ms * 1e3:.0f} images/s, "
          f"{b / ms / 1e9:.2f} TB/s algorithmic")

    import time
    from pgca_amd.input import draw_train_params
    g = torch.Generator().manual_seed(0)
    t0 = time.perf_counter()
    params = [draw_train_params(a.height, a.width, g) for _ in range(a.batch)]
    t_draw = time.perf_counter() - t0
    for _ in range(2):
        proc.process_train_batch(x, params, out=out)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.iters):
        proc.process_train_batch(x, params, out=out)
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / a.iters
    print(f"train {a.batch} x {a.height}x{a.width} -> {a.size}: {wall * 1e6:.1f} us wall per batch (host tables + upload + "
          f"3 kernels), {a.batch / wall:.0f} images/s; parameter draws {t_draw * 1e3:.1f} ms per batch on the host")


if __name__ == "__main__":
    main()
