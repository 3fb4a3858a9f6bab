#!/usr/bin/env python3
"""Per-kernel summary of the rocprofv3 PMC passes of ONE bench command (separate passes: the SQ counters, FETCH_SIZE,
WRITE_SIZE, TCC hit/miss - FETCH and WRITE cannot share a pass on gfx950):

    python tools/pmc_summary.py --sq DIR --fetch DIR --write DIR --tcc DIR --pairs 128 --out profiles/r02_pmc_summary.json

Units / corrections follow /opt/skills/guides/MI355X_MICROARCH.md: FETCH_SIZE and WRITE_SIZE are KiB; on gfx950
FETCH_SIZE tallies 128-B requests at 64 B, so wide streaming reads are doubled (calibrated in-run on AdamW: exactly
16 B read and 14 B written per parameter).  SQ_VALU_MFMA_BUSY_CYCLES counts 16 cycles per v_mfma_f32_16x16x32_bf16
(checked: FLOPs / 16384 x 16 reproduces the counter); SQ_BUSY_CYCLES is summed over the 32 shader engines, so
mfma_busy = MFMA_BUSY / (4 SIMDs x 256 CUs x SQ_BUSY / 32) is the fraction of matrix-pipe cycles in use while the
kernel runs, at whatever clock the chip held.
"""
import argparse
import collections
import csv
import glob
import json
import os


def counters(d):
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    n = collections.defaultdict(set)
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        with open(f, newline="") as fh:
            for r in csv.DictReader(fh):
                k = r["Kernel_Name"]
                acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
                n[k].add((r["Process_Id"], r["Dispatch_Id"]))
    return {k: ({c: v / len(n[k]) for c, v in cs.items()}, len(n[k])) for k, cs in acc.items()}


def durations(d):
    tot, cnt = collections.defaultdict(float), collections.Counter()
    for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
        with open(f, newline="") as fh:
            for r in csv.DictReader(fh):
                tot[r["Kernel_Name"]] += float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
                cnt[r["Kernel_Name"]] += 1
    return {k: (tot[k] / cnt[k] / 1e3, cnt[k]) for k in tot}


def main():
    ap = argparse.ArgumentParser()
    for a in ("sq", "fetch", "write", "tcc"):
        ap.add_argument("--" + a, required=True)
    ap.add_argument("--pairs", type=int, default=128)
    ap.add_argument("--commit", default="")
    ap.add_argument("--out", required=True)
    ap.add_argument("--top", type=int, default=14)
    a = ap.parse_args()
    sq, fe, wr, tc, du = counters(a.sq), counters(a.fetch), counters(a.write), counters(a.tcc), durations(a.sq)
    rows = []
    for k, (us, n) in sorted(du.items(), key=lambda kv: -kv[1][0] * kv[1][1])[:a.top]:
        c = sq.get(k, ({}, 0))[0]
        row = {"kernel": k, "launches": n, "avg_us_profiled": us}
        if c.get("SQ_BUSY_CYCLES"):
            clocks = c["SQ_BUSY_CYCLES"] / 32.0
            row["mfma_busy"] = c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (4 * 256 * clocks)
            row["clock_ghz_profiled"] = clocks / us / 1e3
            wc = max(c.get("SQ_WAVE_CYCLES", 0.0), 1.0)
            row["wave_time_waiting"] = c.get("SQ_WAIT_ANY", 0.0) / wc
            row["wave_time_lds_issue_stall"] = c.get("SQ_WAIT_INST_LDS", 0.0) / wc
            row["wave_time_issuing"] = c.get("SQ_ACTIVE_INST_ANY", 0.0) / wc
        if k in fe:
            row["hbm_read_bytes"] = 2.0 * fe[k][0].get("FETCH_SIZE", 0.0) * 1024.0
        if k in wr:
            row["hbm_write_bytes"] = wr[k][0].get("WRITE_SIZE", 0.0) * 1024.0
        if "hbm_read_bytes" in row and "hbm_write_bytes" in row:
            row["hbm_bytes"] = row["hbm_read_bytes"] + row["hbm_write_bytes"]
            row["hbm_tb_per_s_profiled"] = row["hbm_bytes"] / us / 1e6
        if k in tc:
            h, m = tc[k][0].get("TCC_HIT_sum", 0.0), tc[k][0].get("TCC_MISS_sum", 0.0)
            if h + m > 0:
                row["l2_hit_rate"] = h / (h + m)
        rows.append(row)
    out = {"command": "rocprofv3 --kernel-trace --pmc <set> -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-probe "
                      "(one pass per counter set)", "pairs_per_gpu": a.pairs, "commit": a.commit, "note": __doc__.split("\n\n")[1], "kernels": rows}
    with open(a.out, "w") as fh:
        json.dump(out, fh, indent=1)
    for r in rows:
        print(f'{r["kernel"][:58]:58s} n={r["launches"]:4d} {r["avg_us_profiled"]:8.1f}us mfma_busy={r.get("mfma_busy", 0):.2f} '
              f'read={r.get("hbm_read_bytes", 0) / 1e6:8.1f}MB write={r.get("hbm_write_bytes", 0) / 1e6:8.1f}MB '
              f'L2hit={r.get("l2_hit_rate", 0):.2f} wait={r.get("wave_time_waiting", 0):.2f}')


if __name__ == "__main__":
    main()
