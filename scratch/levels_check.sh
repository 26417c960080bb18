#!/bin/bash
# lazily made index levels: parity suite, then the uneven shapes and configs 3 / 4 (development aid)
cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -u -m pytest tests -m gpu -x -q --timeout 180 --timeout-method thread > gpurun_out/t11.log 2>&1; tail -3 gpurun_out/t11.log
timeout -k 10 300 python -u scratch/bench_configs.py 2>/dev/null | grep -E "knn " | cut -c1-100 || exit 1
timeout -k 10 300 python bench.py --config c4 --points 20000000 --no-cpu --steps 2 2>/dev/null | python -c "import json,sys;d=json.loads(sys.stdin.read());print('c4 20M', d['ms_per_step'], d['step_parts_ms'])" || exit 1
timeout -k 10 300 python bench.py --config c3 --no-cpu --frames 300 2>/dev/null | python -c "import json,sys;d=json.loads(sys.stdin.read());print('c3', d['latency_ms'])" || exit 1
