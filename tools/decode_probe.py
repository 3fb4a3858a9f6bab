#!/usr/bin/env python3
"""Where a decode step's time goes: eager launches vs HIP-graph replay vs GPU time (events), per row count."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    from pgca_amd.model import PreferenceGuidedCaptioningModel
    dev = torch.device("cuda:0")
    model = PreferenceGuidedCaptioningModel("openai/clip-vit-base-patch32", "gpt2-medium", 512, freeze_vision_backbone=True,
                                            device=dev, seed=1)
    eng = model.caption_decoder.engine
    for R, skinny in ((1, 0), (1, 64), (2, 0), (2, 64), (4, 0), (4, 64), (8, 0), (8, 64), (16, 0), (16, 64), (32, 0),
                      (32, 64), (128, 0)):
        pv = torch.randn(R, 1024, device=dev)
        tok = torch.randint(0, 50257, (R,), device=dev)
        eng.trunk.SKINNY_ROWS = skinny
        eng.__dict__.pop("_graphs", None)
        print(f"--- rows {R}, {'skinny products' if skinny else 'tile GEMMs'}")
        for graphs in (True,):
            eng.use_graphs = graphs
            for rep in range(3):            # pass 0 sizes buffers / captures, passes 1-2 are timed
                eng.decode_begin(pv, 50)
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                t0 = time.perf_counter()
                e0.record()
                for t in range(48):
                    eng.decode_advance(tok)
                e1.record()
                t_issue = time.perf_counter() - t0
                torch.cuda.synchronize()
                t_wall = time.perf_counter() - t0
            print(f"R={R:4d} graphs={graphs!s:5} (still on: {eng.use_graphs}) per token: issue {1e3 * t_issue / 48:.3f} ms, "
                  f"wall {1e3 * t_wall / 48:.3f} ms, gpu {e0.elapsed_time(e1) / 48:.3f} ms", flush=True)


if __name__ == "__main__":
    main()
