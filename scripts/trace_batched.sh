#!/bin/bash
REPO=$(pwd); OUT=$REPO/gpurun_out/trace_batched; mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
rocprofv3 --kernel-trace --memory-copy-trace --stats --output-format csv -d $OUT -o t -- python $REPO/tests/tools/bench_batched.py 2>&1 | grep -E "BEST" | cut -c1-200
cd $REPO
python - <<PY
import csv
rows=list(csv.DictReader(open("$OUT/t_kernel_trace.csv")))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
sel=[r for r in rows if 'clc::' in r['Kernel_Name'] or 'copy' in r['Kernel_Name'].lower()]
prev=None
for r in sel[-26:]:
    s=int(r['Start_Timestamp']); e=int(r['End_Timestamp'])
    gap=(s-prev)/1000 if prev else 0
    print(f"{r['Kernel_Name'][:50]:50s} dur {(e-s)/1000:8.2f} us gap {gap:8.2f} us grid {r['Grid_Size_X']:>7s} vgpr {r['VGPR_Count']}")
    prev=e
PY
