#!/bin/bash
# A/B of two builds of the library on the headline kNN launch, alternating in one session (development aid)
cd $GRAFT_REPO_ROOT
for r in 1 2 3; do
  for v in A B; do
    if [ $v = A ]; then export PC_AMD_LIB=scratch/libpc_A.so; else unset PC_AMD_LIB; fi
    echo -n "$v: "
    timeout -k 10 100 python bench.py --no-cpu --no-pipeline --steps 20 2>/dev/null | python -c "import json,sys;d=json.loads(sys.stdin.read());print(d['roofline']['kernel_ms'],d['ms_per_step'])"
  done
done
