#!/bin/bash
# Run on the GPU box (via gpurun): rocprofv3 kernel trace + PMC passes (separate runs, --kernel-trace only beside --pmc) of
# scripts/r03_workloads.py and the kernel stats of the default bench command.  Outputs under gpurun_out/prof_r03/;
# scripts/summarize_r03.py turns them into profiles/r03_*.
REPO=$(pwd)
OUT=$REPO/gpurun_out/prof_r03
mkdir -p $OUT
export TMPDIR=/tmp
W="python $REPO/scripts/r03_workloads.py $R03_ARGS"
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/w_trace -o w -- $W > $OUT/w_trace.log 2>&1; echo "workloads trace rc=$?"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/w_fetch -o w -- $W > $OUT/w_fetch.log 2>&1; echo "workloads fetch rc=$?"
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/w_write -o w -- $W > $OUT/w_write.log 2>&1; echo "workloads write rc=$?"
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_SALU SQ_INSTS_LDS --kernel-trace --output-format csv -d $OUT/w_valu -o w -- $W --quick > $OUT/w_valu.log 2>&1; echo "workloads valu rc=$?"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/b_trace -o b -- python $REPO/bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-cold-start > $OUT/b_trace.log 2>&1; echo "bench trace rc=$?"
cd $REPO
find $OUT -name "*.db" -delete 2>/dev/null
du -sh $OUT
