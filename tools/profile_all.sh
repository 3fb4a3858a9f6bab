#!/bin/bash
# One rocprofv3 kernel-trace pass and four PMC passes (separate, as MI355X_MICROARCH.md prescribes: SQ set, FETCH_SIZE,
# WRITE_SIZE, TCC hit/miss) of the default bench command, summarised into gpurun_out/<tag>_*.  Run ON THE GPU BOX:
#   bash tools/profile_all.sh r03 <commit>
set -e
TAG=${1:-r03}; COMMIT=${2:-unknown}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out/${TAG}_prof
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python3 $ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-probe"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o s -- $CMD > $OUT/stats.log 2>&1
echo "stats pass done"
rocprofv3 --kernel-trace --output-format csv --pmc SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY -d $OUT/sq -o p -- $CMD > $OUT/sq.log 2>&1
echo "sq pass done"
rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE -d $OUT/fetch -o p -- $CMD > $OUT/fetch.log 2>&1
echo "fetch pass done"
rocprofv3 --kernel-trace --output-format csv --pmc WRITE_SIZE -d $OUT/write -o p -- $CMD > $OUT/write.log 2>&1
echo "write pass done"
rocprofv3 --kernel-trace --output-format csv --pmc TCC_HIT_sum TCC_MISS_sum -d $OUT/tcc -o p -- $CMD > $OUT/tcc.log 2>&1
echo "tcc pass done"
cd $ROOT
python3 tools/pmc_summary.py --sq $OUT/sq --fetch $OUT/fetch --write $OUT/write --tcc $OUT/tcc --pairs 512 --commit $COMMIT \
    --out gpurun_out/${TAG}_pmc_summary.json | tee gpurun_out/${TAG}_pmc_summary.txt
python3 tools/kstats.py $OUT/stats --csv gpurun_out/${TAG}_bench_kernel_stats.csv | tee gpurun_out/${TAG}_kernel_stats.txt
# keep the merge-back small: the raw traces stay on the box
rm -rf $OUT/sq $OUT/fetch $OUT/write $OUT/tcc
find $OUT/stats -name "*.csv" ! -name "*kernel_stats*" -delete 2>/dev/null || true
