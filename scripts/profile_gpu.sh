#!/bin/bash
# Run on the GPU box (via gpurun): rocprofv3 kernel trace + PMC passes of the bench command.
# Outputs under gpurun_out/prof_<tag>/ ; scripts/summarize_profile.py turns them into profiles/.
TAG=${1:-r01}
REPO=$(pwd)
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
# the default bench command, minus the CPU baseline (pure host work) ...
CMD="python $REPO/bench.py --steps 50 --warmup 5 --no-cpu-baseline"
cd /tmp
# 1) kernel trace + stats of the C2 workload only (no >L3 pass, so the per-kernel averages are clean)
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o trace -- $CMD --large-obs 0 > $OUT/trace_bench.log 2>&1
echo "trace rc=$?"
# 2) PMC passes (own runs, --kernel-trace only): HBM read / write bytes per dispatch, incl. the 8e6-obs pass
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -o pmc -- $CMD > $OUT/pmc_fetch_bench.log 2>&1
echo "pmc fetch rc=$?"
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -o pmc -- $CMD > $OUT/pmc_write_bench.log 2>&1
echo "pmc write rc=$?"
# 2b) instruction mix of the dominant kernel (own run, --kernel-trace only): FP64 issue is what bounds the streaming loop
# while the array is cache-resident
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_SALU SQ_INSTS_LDS --kernel-trace --output-format csv -d $OUT/pmc_valu -o pmc -- python $REPO/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-batched --large-obs 0 > $OUT/pmc_valu_bench.log 2>&1
echo "pmc valu rc=$?"
# 3) plain run of the same command for the bench line that goes with these profiles
$CMD > $OUT/bench_plain.log 2>&1
echo "plain rc=$?"
cd $REPO
grep -h '^{' $OUT/bench_plain.log | tail -1 | cut -c1-400
du -sh $OUT
