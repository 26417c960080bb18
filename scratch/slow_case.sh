#!/bin/bash
# kernel stats of one stress case: slow_case.sh <seed> <case>
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
export SKIP=$(seq -s, 0 $(($2-1)))
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d gpurun_out/slow_kt -o kt --output-format csv -- python3 scratch/stress_one.py $(($2+1)) $1 $2 > gpurun_out/slow.log 2>&1 || exit 1
grep -E "pc_amd\]|^case" gpurun_out/slow.log | tail -30
python3 - <<'PY'
import csv
rows=list(csv.DictReader(open('gpurun_out/slow_kt/kt_kernel_stats.csv')))
for r in rows[:8]:
    print(r["Name"][:100].ljust(100), r["Calls"].rjust(5), "%10.1f us"%(float(r["AverageNs"])/1e3), r["Percentage"])
PY
