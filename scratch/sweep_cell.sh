#!/bin/bash
cd $GRAFT_REPO_ROOT
for r in 1 2; do
for f in 0.96 0.98 1.0 1.02 1.04; do
  echo -n "cell scale $f: "
  PC_AMD_CELL_SCALE=$f timeout -k 10 100 python bench.py --no-cpu --no-pipeline --steps 20 2>/dev/null | python -c "import json,sys;d=json.loads(sys.stdin.read());print(d['roofline']['kernel_ms'],d['ms_per_step'])" || exit 1
done
done
