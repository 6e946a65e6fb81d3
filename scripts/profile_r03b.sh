#!/bin/bash
# GPU box: rocprofv3 kernel trace / PMC passes (separate runs; --kernel-trace only beside --pmc) of scripts/r03_prof_probe.py targets and
# the kernel stats of the default bench command, every run under its own timeout.  Outputs: gpurun_out/prof_r03b/.
R=$(pwd); O=$R/gpurun_out/prof_r03b; mkdir -p $O
export TMPDIR=/tmp
cd /tmp
run() {  # name, timeout, rocprof args..., -- target args
  local name=$1 to=$2; shift 2
  local s=$(date +%s)
  timeout $to rocprofv3 "$@" > $O/$name.log 2>&1
  echo "$name rc=$? $(( $(date +%s) - s ))s"
}
P="python $R/scripts/r03_prof_probe.py"
VALU="SQ_INSTS_VALU SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_SALU SQ_INSTS_LDS"
for t in "resident 2048" "step 1000000" "eval 32000000"; do
  n=$(echo $t | tr " " "_")
  run ${n}_trace 240 --kernel-trace --stats --output-format csv -d $O/${n}_trace -o w -- $P $t
  run ${n}_fetch 240 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/${n}_fetch -o w -- $P $t
  run ${n}_write 240 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/${n}_write -o w -- $P $t
done
run resident_2048_valu 240 --pmc $VALU --kernel-trace --output-format csv -d $O/resident_2048_valu -o w -- $P resident 2048
run resident_8192_fetch 400 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/resident_8192_fetch -o w -- $P resident 8192
run bench_trace 600 --kernel-trace --stats --output-format csv -d $O/bench_trace -o b -- python $R/bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-cold-start
cd $R
find $O -name "*.db" -delete 2>/dev/null
du -sh $O
