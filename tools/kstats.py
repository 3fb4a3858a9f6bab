#!/usr/bin/env python3
"""Per-kernel statistics from a rocprofv3 --kernel-trace run, from either output format:

    python tools/kstats.py gpurun_out/prof_dir [--csv out.csv] [--skip-launches N]

reads the rocpd SQLite database (rocprofv3's default output) or the *kernel_trace.csv files below the directory and
prints, per kernel: calls, total ms, average / min / max us, share of kernel time - the table rocprofv3 --stats writes,
in one format whatever the profiler version emitted.  --csv writes the same table (the file committed under profiles/).
"""
import argparse
import collections
import csv
import glob
import os
import re
import sqlite3


def from_db(path):
    db = sqlite3.connect(path)
    cur = db.cursor()
    tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
    out = []
    for t in tabs:
        if not t.startswith("rocpd_kernel_dispatch"):
            continue
        suffix = t[len("rocpd_kernel_dispatch"):]
        sym = "rocpd_info_kernel_symbol" + suffix
        q = (f'select s.kernel_name, d.start, d."end" from `{t}` d join `{sym}` s on d.kernel_id = s.id '
             f'order by d.start')
        out.extend(cur.execute(q).fetchall())
    return out


def from_csv(d):
    out = []
    for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
        with open(f, newline="") as fh:
            for r in csv.DictReader(fh):
                out.append((r["Kernel_Name"], int(r["Start_Timestamp"]), int(r["End_Timestamp"])))
    out.sort(key=lambda r: r[1])
    return out


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    return name.split("(")[0] if "<" not in name else re.sub(r"\(.*$", "", name)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("dir")
    ap.add_argument("--csv")
    ap.add_argument("--top", type=int, default=40)
    a = ap.parse_args()
    dbs = glob.glob(os.path.join(a.dir, "**", "*.db"), recursive=True)
    rows = []
    for p in dbs:
        rows.extend(from_db(p))
    if not rows:
        rows = from_csv(a.dir)
    agg = collections.defaultdict(list)
    for name, s, e in rows:
        agg[short(name)].append((e - s) / 1e3)
    total = sum(sum(v) for v in agg.values())
    table = sorted(((k, len(v), sum(v) / 1e3, sum(v) / len(v), min(v), max(v), 100.0 * sum(v) / total)
                    for k, v in agg.items()), key=lambda r: -r[2])
    print(f"{'kernel':70s} {'calls':>7s} {'total ms':>10s} {'avg us':>9s} {'min us':>9s} {'max us':>9s} {'%':>6s}")
    for r in table[:a.top]:
        print(f"{r[0][:70]:70s} {r[1]:7d} {r[2]:10.2f} {r[3]:9.1f} {r[4]:9.1f} {r[5]:9.1f} {r[6]:6.2f}")
    print(f"total kernel time {total / 1e3:.2f} ms over {len(rows)} dispatches")
    if a.csv:
        with open(a.csv, "w", newline="") as fh:
            w = csv.writer(fh)
            w.writerow(["Name", "Calls", "TotalDurationMs", "AverageUs", "MinUs", "MaxUs", "Percentage"])
            for r in table:
                w.writerow([r[0], r[1], f"{r[2]:.3f}", f"{r[3]:.2f}", f"{r[4]:.2f}", f"{r[5]:.2f}", f"{r[6]:.3f}"])


if __name__ == "__main__":
    main()
