#!/bin/bash
for T0 in 2 3 4; do for T1 in 1 2 3; do
  echo "== tile0 $T0 tile1 $T1"
  PC_AMD_UNEVEN_TILE0=$T0 PC_AMD_UNEVEN_TILE1=$T1 timeout -k 10 300 python scratch/bench_configs.py sphere planes lidar clusters c4 icp 2>&1 | grep -E "n=" | cut -c1-110 || exit 1
done; done
