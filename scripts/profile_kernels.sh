#!/bin/bash
# GPU box: rocprofv3 evidence for the cooperative solve (csrc/clc_coop.hpp, controller of csrc/clc_lmuni.hpp) and the resident
# batched kernel — kernel trace + stats, PMC passes (separate runs, never with other trace domains) of scripts/prof_probe.py, and the
# kernel stats of the default bench command.  Every run under its own timeout.
# Outputs: gpurun_out/prof_kernels/ (scripts/summarize_kernels.py turns them into profiles/<round>_*.{md,json,csv}).
R=$(pwd); O=$R/gpurun_out/prof_kernels; mkdir -p $O
export TMPDIR=/tmp
cd /tmp
run() {  # name, timeout, rocprof args..., -- target args
  local name=$1 to=$2; shift 2
  local s=$(date +%s)
  timeout $to rocprofv3 "$@" > $O/$name.log 2>&1
  echo "$name rc=$? $(( $(date +%s) - s ))s"
}
P="python $R/scripts/prof_probe.py"
VALU="SQ_INSTS_VALU SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_SALU SQ_INSTS_LDS"
if [ "$1" != "--bench-only" ]; then
CLC_PROBE_SOLVES=60 run coop_trace 120 --kernel-trace --stats --output-format csv -d $O/coop_trace -o w -- $P coop 1000000
run coop_fetch 120 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/coop_fetch -o w -- $P coop 1000000
run coop_write 120 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/coop_write -o w -- $P coop 1000000
run coop_valu 120 --pmc $VALU --kernel-trace --output-format csv -d $O/coop_valu -o w -- $P coop 1000000
run coop_valu4 120 --pmc $VALU --kernel-trace --output-format csv -d $O/coop_valu4 -o w -- $P coop 1000000 4
run coop_wait 120 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY --kernel-trace --output-format csv -d $O/coop_wait -o w -- $P coop 1000000
run coop_wait2 120 --pmc SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM --kernel-trace --output-format csv -d $O/coop_wait2 -o w -- $P coop 1000000
run res_trace 180 --kernel-trace --stats --output-format csv -d $O/res_trace -o w -- $P resident 8192
run res_fetch 180 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/res_fetch -o w -- $P resident 8192
run res_write 180 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/res_write -o w -- $P resident 8192
run res_valu 180 --pmc $VALU --kernel-trace --output-format csv -d $O/res_valu -o w -- $P resident 8192
fi
[ "$1" = "--counters-only" ] || run bench_trace 500 --kernel-trace --stats --output-format csv -d $O/bench_trace -o b -- python $R/bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-cold-start
cd $R
find $O -name "*.db" -delete 2>/dev/null
du -sh $O
