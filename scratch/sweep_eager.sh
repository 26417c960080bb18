#!/bin/bash
cd $GRAFT_REPO_ROOT
for e in 4 3 2; do
  echo "== eager levels $e"
  PC_AMD_EAGER_LEVELS=$e timeout -k 10 300 python -u scratch/bench_configs.py 2>/dev/null | grep -E "knn " | grep -v uniform | cut -c1-100 || exit 1
  PC_AMD_EAGER_LEVELS=$e timeout -k 10 300 python bench.py --config c4 --points 20000000 --no-cpu --steps 2 2>/dev/null | python -c "import json,sys;d=json.loads(sys.stdin.read());print('c4 20M', d['ms_per_step'], d['step_parts_ms'])" || exit 1
  PC_AMD_EAGER_LEVELS=$e timeout -k 10 300 python bench.py --config c3 --no-cpu --frames 300 2>/dev/null | python -c "import json,sys;d=json.loads(sys.stdin.read());print('c3', d['latency_ms'])" || exit 1
  PC_AMD_EAGER_LEVELS=$e timeout -k 10 300 python bench.py --config c5 --no-cpu 2>/dev/null | python -c "import json,sys;d=json.loads(sys.stdin.read());print('c5', d['ms_per_step'])" || exit 1
done
