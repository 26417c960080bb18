#!/bin/bash
# A/B of two builds with per-kernel times (rocprofv3 kernel trace) on the headline launch (development aid):
# A = scratch/libpc_A.so, B = the library in the tree
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
for v in A B; do
  if [ $v = A ]; then export PC_AMD_LIB=$GRAFT_REPO_ROOT/scratch/libpc_A.so; else unset PC_AMD_LIB; fi
  rm -rf gpurun_out/kt_$v
  timeout -k 10 200 rocprofv3 --kernel-trace --stats -d gpurun_out/kt_$v -o kt --output-format csv -- python3 bench.py --no-cpu --no-pipeline --steps 20 > gpurun_out/kt_$v.json 2> gpurun_out/kt_$v.log || exit 1
  echo "== $v"
  grep -E "k_tile_scan|k_wave_query|k_scan<" gpurun_out/kt_$v/kt_kernel_stats.csv | cut -d, -f1-4 | cut -c1-150
done
bash scratch/ab.sh
