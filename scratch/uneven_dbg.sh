#!/bin/bash
# per-level log of the uneven shapes, knobs from the arguments (development aid)
cd $GRAFT_REPO_ROOT
for v in "$@"; do
  echo "=== $v"
  env $v PC_AMD_DEBUG=1 timeout -k 10 200 python scratch/bench_configs.py planes clusters 2>&1 | grep -E "hints|level |pass|open|n=" | awk '!seen[$0]++' | cut -c1-150 || exit 1
  env $v PC_AMD_DEBUG=1 timeout -k 10 300 python bench.py --config c4 --steps 1 --no-cpu 2>&1 | grep -E "level |pass|open|step_parts" | tail -28 | cut -c1-400
done
