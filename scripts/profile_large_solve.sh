#!/bin/bash
# GPU box: rocprofv3 evidence for the default-path clc_solve of ONE problem beyond the chip (3.2e7 observations, 557 MB of rows: the step chain,
# csrc/clc_kernels.hpp step_kernel) — kernel trace + stats, then PMC passes (separate runs, never with other trace domains).
# Outputs: gpurun_out/prof_large/ (scripts/summarize_large_solve.py -> profiles/r06_large_solve.md, profiles/pmc_traffic_large_solve.json).
R=$(pwd); O=$R/gpurun_out/prof_large; mkdir -p $O
N=${1:-32000000}
export TMPDIR=/tmp
cd /tmp
run() {  # name, timeout, rocprof args..., -- target args
  local name=$1 to=$2; shift 2
  local s=$(date +%s)
  timeout $to rocprofv3 "$@" > $O/$name.log 2>&1
  echo "$name rc=$? $(( $(date +%s) - s ))s"
}
P="python $R/scripts/prof_probe.py"
run trace 300 --kernel-trace --stats --output-format csv -d $O/trace -o w -- $P large $N
run fetch 300 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/fetch -o w -- $P large $N
run write 300 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/write -o w -- $P large $N
run valu 300 --pmc SQ_INSTS_VALU SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY --kernel-trace --output-format csv -d $O/valu -o w -- $P large $N
run calib 300 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/calib -o w -- $P eval $N
cd $R
find $O -name "*.db" -delete 2>/dev/null
find $O -name "*agent_info*" -delete 2>/dev/null
du -sh $O; ls -R $O | head -40
