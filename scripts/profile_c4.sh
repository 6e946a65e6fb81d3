#!/bin/bash
# rocprofv3 kernel stats of one C4 shard (8 192 problems x 10^4 observations, what each of 8 GPUs holds) through clc_solve_batched
REPO=$(pwd); OUT=$REPO/gpurun_out/prof_c4; mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o c4 -- python $REPO/tests/tools/bench_batched.py 8192 20 500 > $OUT/bench.log 2>&1
echo "rc=$?"; grep -E "BEST|parity" $OUT/bench.log | cut -c1-200
