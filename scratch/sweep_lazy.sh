#!/bin/bash
cd $GRAFT_REPO_ROOT
for d in 32 16 8 4; do
  echo "== lazy div $d"
  PC_AMD_LAZY_DIV=$d timeout -k 10 300 python -u scratch/bench_configs.py 2>/dev/null | grep -E "knn " | grep -v uniform | cut -c1-100 || exit 1
  PC_AMD_LAZY_DIV=$d timeout -k 10 300 python bench.py --config c4 --points 20000000 --no-cpu --steps 2 2>/dev/null | python -c "import json,sys;d=json.loads(sys.stdin.read());print('c4 20M', d['ms_per_step'], d['step_parts_ms'])" || exit 1
done
PC_AMD_DEBUG=1 timeout -k 10 300 python -u scratch/bench_configs.py 2>&1 | grep -E "level [0-9]:|open after|knn " | cut -c1-150 | tail -60
