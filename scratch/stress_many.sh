#!/bin/bash
# randomised parity sweeps with seeds the test-suite does not use (development aid): stress_many.sh seed...
cd $GRAFT_REPO_ROOT
for s in "$@"; do
  echo "== seed $s"
  timeout -k 10 400 python -u scratch/stress.py 60 $s 2>&1 | awk '{ if ($NF != "ok" || $(NF-2)+0 > 2000) print }' | grep -v amdgpu
done
