#!/bin/bash
# headline kNN launch against the tile kernel's threshold seed (development aid)
cd $GRAFT_REPO_ROOT
for r in 1 2; do
for f in 1.001 1.1 1.2 1.35 1.6; do
  echo -n "seed $f: "
  PC_AMD_SEED_SCALE=$f timeout -k 10 100 python bench.py --no-cpu --no-pipeline --steps 20 2>/dev/null | python -c "import json,sys;d=json.loads(sys.stdin.read());print(d['roofline']['kernel_ms'],d['ms_per_step'])" || exit 1
done
done
