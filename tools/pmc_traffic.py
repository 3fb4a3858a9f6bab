"""Per-kernel HBM traffic from two rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE cannot share a pass on gfx950).

    python tools/pmc_traffic.py <dir with FETCH_SIZE pass> <dir with WRITE_SIZE pass> [--out profiles/x.json]

Units and corrections follow /opt/skills/guides/MI355X_MICROARCH.md (HBM section): rocprofv3 reports both
counters in KiB; on gfx950 FETCH_SIZE tallies 128-B requests at 64 B, so wide streaming reads are doubled.
The AdamW kernel (exactly 16 B read + 14 B written per parameter, all 16-B-per-lane streams) is printed as the
in-run calibration of that correction.
"""
import argparse
import csv
import glob
import json
import os
from collections import defaultdict


def per_kernel(d, counter):
    acc = defaultdict(lambda: [0.0, 0])
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        per_dispatch = defaultdict(float)
        names = {}
        with open(f, newline="") as fh:
            for r in csv.DictReader(fh):
                if r["Counter_Name"] != counter:
                    continue
                key = (r["Process_Id"], r["Dispatch_Id"])
                per_dispatch[key] += float(r["Counter_Value"])      # one row per XCD/instance on some builds
                names[key] = (r["Kernel_Name"], int(r["Grid_Size"]))
        for key, v in per_dispatch.items():
            a = acc[names[key][0]]
            a[0] += v
            a[1] += 1
    return {k: (v[0] / v[1], v[1]) for k, v in acc.items()}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("fetch_dir")
    ap.add_argument("write_dir")
    ap.add_argument("--out", default=None)
    ap.add_argument("--match", default="gemm256s_kernel<0, 1>")
    a = ap.parse_args()
    fe, wr = per_kernel(a.fetch_dir, "FETCH_SIZE"), per_kernel(a.write_dir, "WRITE_SIZE")
    rows = []
    for k in sorted(set(fe) | set(wr)):
        if "pgca" not in k and "kernel" not in k:
            continue
        f, nf = fe.get(k, (0.0, 0))
        w, nw = wr.get(k, (0.0, 0))
        rows.append({"kernel": k, "launches": max(nf, nw), "fetch_kib_raw": f, "write_kib_raw": w,
                     "read_bytes": 2.0 * f * 1024.0, "write_bytes": w * 1024.0,
                     "hbm_bytes": 2.0 * f * 1024.0 + w * 1024.0})
    rows.sort(key=lambda r: -r["hbm_bytes"] * r["launches"])
    for r in rows[:24]:
        print(f'{r["kernel"][:70]:70s} n={r["launches"]:5d} read={r["read_bytes"] / 1e6:9.2f} MB '
              f'write={r["write_bytes"] / 1e6:9.2f} MB')
    sel = [r for r in rows if a.match in r["kernel"]]
    out = {"note": "average per launch; read = 2 x FETCH_SIZE KiB (gfx950 correction), write = WRITE_SIZE KiB",
           "dominant": sel[0] if sel else None, "kernels": rows}
    if a.out:
        with open(a.out, "w") as fh:
            json.dump(out, fh, indent=1)


if __name__ == "__main__":
    main()
