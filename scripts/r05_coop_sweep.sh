#!/bin/bash
# GPU box: C2 cooperative solve under different first-poll offsets (CLC_COOP_D1 / CLC_COOP_D2: shader cycles after barrier A at which the
# leaders / everybody first look at the boards; csrc/clc_coop.hpp) — one process per setting, scripts/r04_coop_only.py
# usage: scripts/r05_coop_sweep.sh "d1 list" "d2 list" [n_poses pts]
for d1 in ${1:-500 800 1100}; do for d2 in ${2:-2400 2800 3200 3500}; do
  echo -n "D1=$d1 D2=$d2 "; CLC_COOP_D1=$d1 CLC_COOP_D2=$d2 python scripts/r04_coop_only.py $3 $4 2>/dev/null | cut -c1-260
done; done
