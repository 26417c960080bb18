#!/bin/bash
# uneven-cloud timings for several "full tile" thresholds of the mixed first pass (0 = mixed pass off)
for t in ${THRESH:-0 16 32 48 64 96}; do
  echo "full tile >= $t"
  if [ "$t" = 0 ]; then export PC_AMD_NO_MIXED=1; else unset PC_AMD_NO_MIXED; export PC_AMD_FULL_TILE=$t; fi
  timeout -k 10 300 python -u scratch/bench_configs.py sphere planes lidar clusters c4 2>&1 | grep -v amdgpu.ids | cut -c1-118 || exit 1
  timeout -k 10 300 python -u scratch/c4_full.py ${C4N:-1e8} 2>&1 | grep "all queries:\|of 8\|brute" || exit 1
done
