#!/bin/bash
# GPU box, round 4: rocprofv3 kernel trace of the cooperative kernel's other instantiations — the 32-workgroup one-hop form (12 000 and
# 50 000 observations) and the 24-byte-slot form for points off the lidar plane (C2 shape with z).  Outputs: gpurun_out/prof_r04v/.
R=$(pwd); O=$R/gpurun_out/prof_r04v; mkdir -p $O
export TMPDIR=/tmp
cd /tmp
P="python $R/scripts/r03_prof_probe.py"
for t in "coop 12000" "coop 50000" "coop 1000000" "coopz 1000000"; do
  name=$(echo $t | tr ' ' '_')
  timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d $O/$name -o w -- $P $t > $O/$name.log 2>&1
  echo "$name rc=$?"
done
cd $R
find $O -name "*.db" -delete 2>/dev/null
