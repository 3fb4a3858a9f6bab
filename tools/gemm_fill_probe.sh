#!/bin/bash
# How many VALU instructions the 256^2 main loop can carry per L phase / per MFMA gap before it slows down: rebuilds
# gemm_phase.hip with diagnostic fillers and times the qkv / fc2 shapes (run on the GPU box; restores the product build).
set -e
cd "$(dirname "$0")/.."
for v in "L=0 M=0" "L=16 M=0" "L=32 M=0" "L=64 M=0" "L=0 M=1" "L=0 M=2" "L=0 M=3" "L=0 M=4" "L=32 M=2"; do
  set -- $v
  l=${1#L=}; m=${2#M=}
  touch preference-guided-image-captioning-alignment_amd/csrc/gemm_phase.hip
  PGCA_EXTRA_FLAGS="-DPGCA_FILL_L=$l -DPGCA_FILL_M=$m" python -m pgca_amd.build > /dev/null 2>&1
  echo "== fillers per L phase $l, per MFMA gap $m"
  python tools/gemm_bench.py --rows 32768 --only qkv_fwd --iters 10 2>&1 | grep -v amdgpu.ids | tail -1
  python tools/gemm_bench.py --rows 32768 --only fc2_fwd --iters 10 2>&1 | grep -v amdgpu.ids | tail -1
done
touch preference-guided-image-captioning-alignment_amd/csrc/gemm_phase.hip
python -m pgca_amd.build > /dev/null 2>&1
