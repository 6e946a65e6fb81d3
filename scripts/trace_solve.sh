#!/bin/bash
# kernel trace of a few solves (GPU box)
REPO=$(pwd); OUT=$REPO/gpurun_out/trace_${1:-x}; mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
cat > /tmp/solve_few.py <<PY
import sys, numpy as np
sys.path.insert(0, "$REPO")
import camlasercalibratool_amd as clc
from camlasercalibratool_amd import simdata as sd
S = sd.sim_fixed_count(1000, 2000, 500, noise_sigma=0.01)
rec = clc.flatten_observations(S, False); x0 = sd.pose7_from_T(np.eye(4))
sv = clc.Solver(0); sv.upload(rec)
import os
sv.set_launch(0, int(os.environ.get('CLC_FLAGS','-1')))
for _ in range(6): r = sv.solve(x0)
print(r.summary.num_evaluations, r.summary.solve_ms)
PY
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o t -- python /tmp/solve_few.py 2>&1 | grep -v -E "^W2|^E2|amdgpu.ids" | tail -3
cd $REPO
python - <<PY
import csv
rows=list(csv.DictReader(open("$OUT/t_kernel_trace.csv")))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
sel=[r for r in rows if 'clc::' in r['Kernel_Name']]
prev=None
for r in sel[-34:]:
    s=int(r['Start_Timestamp']); e=int(r['End_Timestamp'])
    gap=(s-prev)/1000 if prev else 0
    print(f"{r['Kernel_Name'][:46]:46s} dur {(e-s)/1000:8.2f} us gap {gap:8.2f} us grid {r['Grid_Size_X']:>7s} vgpr {r['VGPR_Count']}")
    prev=e
PY
