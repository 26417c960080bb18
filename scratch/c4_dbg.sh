#!/bin/bash
# kernel times and level log of the full-size config 4 (development aid)
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout -k 10 600 rocprofv3 --kernel-trace --stats -d gpurun_out/c4_kt -o kt --output-format csv -- python3 bench.py --config c4 --steps 1 --no-cpu > gpurun_out/c4_kt.json 2> gpurun_out/c4_kt.err
python3 - <<'PY'
import csv
rows=list(csv.DictReader(open('gpurun_out/c4_kt/kt_kernel_stats.csv')))
for r in rows[:22]:
    print(r["Name"][:100].ljust(100), r["Calls"].rjust(5), str(int(float(r["AverageNs"]))).rjust(10), r["Percentage"])
PY
PC_AMD_DEBUG=1 timeout -k 10 600 python3 bench.py --config c4 --steps 1 --no-cpu 2>&1 | grep "pc_amd" | tail -32
