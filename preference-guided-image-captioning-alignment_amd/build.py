"""Build ``libpgca_hip.so`` for gfx950 with hipcc (cross-compiles without a GPU).

    python -m pgca_amd.build            # incremental
    python -m pgca_amd.build --force

The library is built IN-TREE (next to ``csrc/``): it is git-ignored but travels to
the GPU box with the repository snapshot.
"""
from __future__ import annotations

import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor
from typing import List

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(CSRC, "_obj")
LIB = os.path.join(HERE, "libpgca_hip.so")
HEADER = os.path.join(os.path.dirname(HERE), "include", "pgca_hip.h")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function"]
# diagnostic builds only (e.g. PGCA_EXTRA_FLAGS="-DPGCA_GEMM_TIMING" for the s_memtime stamps of tools/gemm_bench.py)
FLAGS += os.environ.get("PGCA_EXTRA_FLAGS", "").split()


def hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if cand and (os.path.isabs(cand) and os.path.exists(cand) or not os.path.isabs(cand)):
            return cand
    raise RuntimeError("hipcc not found")


def sources() -> List[str]:
    return sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hip"))


def _stale(target: str, deps: List[str]) -> bool:
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = True, variant: str = "", extra_flags=()) -> str:
    """``variant`` (diagnostic builds: e.g. "timing" with ``-DPGCA_GEMM_TIMING``) goes to its own object directory and
    ``libpgca_hip_<variant>.so``; load it with ``PGCA_LIB=<path>`` (pgca_amd.hip).  The product library has no variant."""
    obj_dir, lib = OBJ, LIB
    flags = list(FLAGS) + list(extra_flags)
    if variant:
        obj_dir = os.path.join(CSRC, "_obj_" + variant)
        lib = os.path.join(HERE, f"libpgca_hip_{variant}.so")
    return _build(force, verbose, obj_dir, lib, flags)


def _build(force: bool, verbose: bool, OBJ: str, LIB: str, FLAGS) -> str:
    os.makedirs(OBJ, exist_ok=True)
    common = [os.path.join(CSRC, h) for h in os.listdir(CSRC) if h.endswith(".h")] + [HEADER]
    cc = hipcc()
    jobs = []
    objs = []
    for src in sources():
        obj = os.path.join(OBJ, os.path.basename(src)[:-4] + ".o")
        objs.append(obj)
        if force or _stale(obj, [src] + common):
            jobs.append((src, obj))

    def run(job):
        src, obj = job
        cmd = [cc] + FLAGS + ["-c", src, "-o", obj]
        p = subprocess.run(cmd, capture_output=True, text=True)
        if p.returncode != 0:
            raise RuntimeError(f"hipcc failed on {os.path.basename(src)}:\n{p.stdout}\n{p.stderr}")
        if verbose and p.stderr.strip():
            print(p.stderr, file=sys.stderr)
        return obj

    if jobs:
        if verbose:
            print(f"[pgca build] compiling {len(jobs)} file(s) for gfx950", file=sys.stderr)
        with ThreadPoolExecutor(max_workers=min(4, len(jobs))) as ex:
            list(ex.map(run, jobs))
    if force or jobs or _stale(LIB, objs):
        p = subprocess.run([cc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs,
                           capture_output=True, text=True)
        if p.returncode != 0:
            raise RuntimeError(f"link failed:\n{p.stdout}\n{p.stderr}")
    return LIB


if __name__ == "__main__":
    args = [a for a in sys.argv[1:] if a != "--force"]
    variant = ""
    extra = []
    for a in args:
        if a.startswith("--variant="):
            variant = a.split("=", 1)[1]
        elif a.startswith("-D"):
            extra.append(a)
    print(build(force="--force" in sys.argv, variant=variant, extra_flags=extra))
