#!/bin/bash
# uneven shapes and config 4 (20 M points) against the number of index levels (development aid)
cd $GRAFT_REPO_ROOT
for lv in 8 6 5 4; do
  echo "== max levels $lv"
  PC_AMD_MAX_LEVELS=$lv timeout -k 10 300 python -u scratch/bench_configs.py 2>/dev/null | grep -E "knn " | grep -v uniform | cut -c1-100 || exit 1
  PC_AMD_MAX_LEVELS=$lv timeout -k 10 300 python bench.py --config c4 --points 20000000 --no-cpu --steps 2 2>/dev/null | python -c "import json,sys;d=json.loads(sys.stdin.read());print('c4 20M', d['ms_per_step'], d['step_parts_ms'])" || exit 1
  PC_AMD_MAX_LEVELS=$lv timeout -k 10 300 python bench.py --config c3 --no-cpu --frames 300 2>/dev/null | python -c "import json,sys;d=json.loads(sys.stdin.read());print('c3', d['latency_ms'])" || exit 1
done
