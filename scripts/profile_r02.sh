#!/bin/bash
# Run on the GPU box (via gpurun): rocprofv3 kernel trace + PMC passes (separate runs, --kernel-trace only beside --pmc)
# of scripts/r02_workloads.py and of the default bench command.  Outputs under gpurun_out/prof_r02/;
# scripts/summarize_r02.py turns them into profiles/r02_*.
REPO=$(pwd)
OUT=$REPO/gpurun_out/prof_r02
mkdir -p $OUT
export TMPDIR=/tmp
W="python $REPO/scripts/r02_workloads.py $R02_ARGS"
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/w_trace -o w -- $W > $OUT/w_trace.log 2>&1; echo "workloads trace rc=$?"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/w_fetch -o w -- $W > $OUT/w_fetch.log 2>&1; echo "workloads fetch rc=$?"
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/w_write -o w -- $W > $OUT/w_write.log 2>&1; echo "workloads write rc=$?"
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_SALU SQ_INSTS_LDS --kernel-trace --output-format csv -d $OUT/w_valu -o w -- $W --quick --max-obs 1000000 > $OUT/w_valu.log 2>&1; echo "workloads valu rc=$?"
# the default bench command (minus the CPU baseline: pure host work), kernel stats
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/b_trace -o b -- python $REPO/bench.py --steps 50 --warmup 5 --no-cpu-baseline > $OUT/b_trace.log 2>&1; echo "bench trace rc=$?"
python $REPO/bench.py --steps 50 --warmup 5 > $OUT/bench_plain.log 2>&1; echo "bench plain rc=$?"
cd $REPO
grep -h '^{' $OUT/bench_plain.log | tail -1 | cut -c1-600
# keep the merge small: the per-dispatch CSVs are what the summary needs
find $OUT -name "*.db" -delete 2>/dev/null
du -sh $OUT
