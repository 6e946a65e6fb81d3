#!/bin/bash
# GPU box: kernel trace + PMC passes of scripts/r03_occupancy.py --one for the three occupancy variants of eval_rows_kernel.
REPO=$(pwd)
OUT=$REPO/gpurun_out/prof_r03_occ
mkdir -p $OUT
export TMPDIR=/tmp
export CLC_LIBRARY=$REPO/camlasercalibratool_amd/csrc/libclc_hip_variants.so
cd /tmp
runp() { local n=$1 to=$2; shift 2; local s=$(date +%s); timeout $to "$@" > $OUT/$n.log 2>&1; echo "$n rc=$? $(( $(date +%s) - s ))s"; }
for v in default 512x4 768x4; do
  if [ $v = default ]; then unset CLC_EVAL_VARIANT; else export CLC_EVAL_VARIANT=$v; fi
  W="python $REPO/scripts/r03_occupancy.py --one"
  runp ${v}_trace 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${v}_trace -o w -- $W
  runp ${v}_pmc1 200 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU --kernel-trace --output-format csv -d $OUT/${v}_pmc1 -o w -- $W
  runp ${v}_pmc2 200 rocprofv3 --pmc SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d $OUT/${v}_pmc2 -o w -- $W
  runp ${v}_pmc3 200 rocprofv3 --pmc SQ_INST_CYCLES_VMEM SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/${v}_pmc3 -o w -- $W
done
cd $REPO
find $OUT -name "*.db" -delete 2>/dev/null
du -sh $OUT
