#!/bin/bash
# cell-edge scale and tile shape sweep of the headline kNN launch (development aid)
cd $GRAFT_REPO_ROOT
for s in 0.92 0.96 1.0 1.04 1.08; do
  echo -n "scale $s: "
  PC_AMD_CELL_SCALE=$s timeout -k 10 100 python bench.py --no-cpu --no-pipeline --steps 10 2>/dev/null | python -c "import json,sys;d=json.loads(sys.stdin.read());print(d['roofline']['kernel_ms'],d['ms_per_step'])"
done
for t in 5,3,2 4,3,3 5,2,3 4,4,2 5,4,2 3,3,4; do
  echo -n "tile $t: "
  PC_AMD_TILE=$t timeout -k 10 100 python bench.py --no-cpu --no-pipeline --steps 10 2>/dev/null | python -c "import json,sys;d=json.loads(sys.stdin.read());print(d['roofline']['kernel_ms'],d['ms_per_step'])"
done
