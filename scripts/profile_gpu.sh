#!/bin/bash
# Run on the GPU box (via gpurun): rocprofv3 kernel trace + PMC passes of the bench command.
# Outputs under gpurun_out/prof_<tag>/ ; summaries are copied into profiles/ by hand.
TAG=${1:-r01}
REPO=$(pwd)
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
CMD="python $REPO/bench.py --steps 10 --warmup 2 --no-cpu-baseline --large-obs 8000000"
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o trace -- $CMD > $OUT/trace_bench.log 2>&1
echo "trace rc=$?"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -o pmc -- $CMD > $OUT/pmc_fetch_bench.log 2>&1
echo "pmc fetch rc=$?"
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -o pmc -- $CMD > $OUT/pmc_write_bench.log 2>&1
echo "pmc write rc=$?"
cd $REPO
find $OUT -type f | head -50
du -sh $OUT
