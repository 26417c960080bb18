#!/bin/bash
# uneven shapes and config 4 under hint / tile knobs (development aid)
cd $GRAFT_REPO_ROOT
for v in "$@"; do
  echo "=== $v"
  env $v timeout -k 10 200 python scratch/bench_configs.py sphere planes lidar clusters 2>&1 | grep -E "n=" | cut -c1-100 || exit 1
  bash scratch/c4_sweep.sh "$v" || exit 1
done
