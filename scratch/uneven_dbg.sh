#!/bin/bash
# per-level log and kernel times of the uneven shapes (development aid)
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
PC_AMD_DEBUG=1 timeout -k 10 200 python scratch/bench_configs.py planes 2>&1 | grep -v amdgpu | tail -40
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d gpurun_out/uneven_kt -o kt --output-format csv -- python3 scratch/bench_configs.py planes clusters > gpurun_out/uneven_kt.log 2>&1
python3 - <<'PY'
import csv
rows=list(csv.DictReader(open('gpurun_out/uneven_kt/kt_kernel_stats.csv')))
for r in rows[:25]:
    print(r["Name"][:100].ljust(100), r["Calls"].rjust(5), str(int(float(r["AverageNs"]))).rjust(9), r["Percentage"])
PY
