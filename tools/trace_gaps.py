#!/usr/bin/env python3
"""GPU idle time inside the timed steps, from a rocprofv3 --kernel-trace csv: the union of all kernel intervals (both
streams) against the span from the first to the last kernel of the last N steps' worth of dispatches.

    rocprofv3 --kernel-trace --output-format csv -d DIR -o t -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-probe
    python tools/trace_gaps.py DIR [--tail-frac 0.6]
"""
import argparse
import csv
import glob
import os


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("dir")
    ap.add_argument("--tail-frac", type=float, default=0.6, help="analyse the last fraction of the dispatches (timed steps)")
    a = ap.parse_args()
    f = glob.glob(os.path.join(a.dir, "**", "*kernel_trace.csv"), recursive=True)[0]
    rows = list(csv.DictReader(open(f)))
    iv = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows)
    iv = iv[int(len(iv) * (1 - a.tail_frac)):]
    span = iv[-1][1] - iv[0][0]
    busy, cur_s, cur_e = 0, iv[0][0], iv[0][1]
    gaps = []
    for s, e, name in iv[1:]:
        if s > cur_e:
            busy += cur_e - cur_s
            gaps.append((s - cur_e, name))
            cur_s, cur_e = s, e
        else:
            cur_e = max(cur_e, e)
    busy += cur_e - cur_s
    print(f"{len(iv)} dispatches over {span / 1e6:.1f} ms: busy {busy / 1e6:.1f} ms, idle {(span - busy) / 1e6:.1f} ms "
          f"({100 * (span - busy) / span:.1f} %), {len(gaps)} gaps, median {sorted(g for g, _ in gaps)[len(gaps) // 2] / 1e3:.1f} us")
    gaps.sort(reverse=True)
    agg = {}
    for g, n in gaps:
        k = n[:60]
        agg[k] = agg.get(k, 0) + g
    print("idle time by the kernel that FOLLOWS the gap (top 12):")
    for k, v in sorted(agg.items(), key=lambda kv: -kv[1])[:12]:
        print(f"  {v / 1e6:8.2f} ms  {k}")
    print("largest gaps:", [(round(g / 1e3), n[:30]) for g, n in gaps[:8]])


if __name__ == "__main__":
    main()
