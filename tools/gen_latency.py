#!/usr/bin/env python3
"""Caption-generation latency on one MI355X (reference target: p95 <= 150 ms per caption,
configs/default.yaml:92, measured the reference's way - evaluation/metrics.py:863-893: wall time of one
``generate_captions``-style call divided by the images in the batch).

    python tools/gen_latency.py [--runs 20] [--max-length 50] [--out profiles/r03_generation_latency.json]

Full-size ViT-B/32 + GPT-2-Medium decoder with seeded random weights (no tokenizer / checkpoints offline: the token
ids are produced, the string decode is not part of the number), images resident on the device.  Modes: the reference's
defaults (num_beams 4, do_sample, top_p 0.9, repetition penalty 1.1) and greedy; batch 1 and 32; K/V-cache decode
(default) and the cache-free recompute for comparison."""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def pct(v, q):
    v = sorted(v)
    return v[min(len(v) - 1, int(round(q * (len(v) - 1))))]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--runs", type=int, default=20)
    ap.add_argument("--max-length", type=int, default=50)
    ap.add_argument("--out", default=None)
    ap.add_argument("--no-cache-free", action="store_true")
    a = ap.parse_args()
    from pgca_amd.model import PreferenceGuidedCaptioningModel
    dev = torch.device("cuda:0")
    model = PreferenceGuidedCaptioningModel("openai/clip-vit-base-patch32", "gpt2-medium", 512, temperature=0.5,
                                            freeze_vision_backbone=True, device=dev, seed=42)
    model.eval()
    # random weights never emit [EOS]: every caption runs to max_length, the worst case for latency
    res = {"model": "CLIP-ViT-B/32 + GPT-2-M decoder, random init", "max_length": a.max_length, "runs": a.runs,
           "definition": "wall time of one batch call / images in the batch (reference evaluation/metrics.py:863-893)",
           "target_p95_ms": 150.0, "cases": []}
    modes = [("reference defaults: 4 beams, do_sample, top_p 0.9, repetition 1.1",
              dict(num_beams=4, do_sample=True, top_p=0.9, repetition_penalty=1.1)),
             ("greedy", dict(num_beams=1, do_sample=False, repetition_penalty=1.1))]
    for B in (1, 32):
        img = torch.randn(B, 3, 224, 224, device=dev)
        for label, kw in modes:
            for cache in ((True,) if a.no_cache_free else (True, False)):
                if not cache and (B == 32 and kw["num_beams"] > 1):
                    continue                      # 128 sequences x 50 recomputed prefixes: only worth one data point
                gen = torch.Generator(device=dev).manual_seed(0)
                lat = []
                for r in range(a.runs + 2):
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                    ids = model.generate_token_ids(img, max_length=a.max_length, generator=gen, use_cache=cache, **kw)
                    torch.cuda.synchronize()
                    if r >= 2:                    # two warm-up calls (buffer allocation)
                        lat.append(1e3 * (time.perf_counter() - t0))
                case = {"batch": B, "mode": label, "kv_cache": cache, "tokens": int(ids.shape[1]),
                        "call_ms_p50": pct(lat, 0.5), "call_ms_p95": pct(lat, 0.95),
                        "per_caption_ms_p50": pct(lat, 0.5) / B, "per_caption_ms_p95": pct(lat, 0.95) / B}
                res["cases"].append(case)
                print(json.dumps(case), flush=True)
    if a.out:
        with open(a.out, "w") as fh:
            json.dump(res, fh, indent=1)


if __name__ == "__main__":
    main()
