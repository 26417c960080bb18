#!/bin/bash
cd $GRAFT_REPO_ROOT
for f in 0.97 0.975 0.98 0.985 0.99 0.995 1.0; do
  echo -n "cell scale $f: 10M k16 "
  PC_AMD_CELL_SCALE=$f timeout -k 10 100 python bench.py --no-cpu --no-pipeline --steps 20 2>/dev/null | python -c "import json,sys;d=json.loads(sys.stdin.read());print(round(d['roofline']['kernel_ms'],4),round(d['ms_per_step'],4),end=' ')" || exit 1
  for nk in "1000000 16" "4000000 16" "10000000 8" "10000000 32" "20000000 16"; do
    set -- $nk
    echo -n "| $1 k$2 "
    PC_AMD_CELL_SCALE=$f timeout -k 10 100 python bench.py --no-cpu --no-pipeline --steps 10 --points $1 --k $2 2>/dev/null | python -c "import json,sys;d=json.loads(sys.stdin.read());print(round(d['roofline']['kernel_ms'],4),end=' ')" || exit 1
  done
  echo
done
