#!/bin/bash
# A/B of two builds on the uneven shapes (development aid): A = scratch/libpc_A.so, B = the library in the tree
cd $GRAFT_REPO_ROOT
for v in A B A B; do
  if [ $v = A ]; then export PC_AMD_LIB=$GRAFT_REPO_ROOT/scratch/libpc_A.so; else unset PC_AMD_LIB; fi
  echo "== $v"
  timeout -k 10 300 python -u scratch/bench_configs.py 2>/dev/null | grep -E "knn " | cut -c1-110 || exit 1
done
