#!/bin/bash
# kernel-trace statistics of a bench command: scratch/kt.sh <tag> <bench args...>
set -o pipefail
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
tag=$1; shift
O=gpurun_out/kt_$tag
rm -rf $O && mkdir -p $O
timeout -k 10 600 rocprofv3 --kernel-trace --stats -d $O -o kt --output-format csv -- python3 bench.py "$@" > $O/bench.json 2> $O/kt.log || { tail -5 $O/kt.log; exit 1; }
f=$(find $O -name "*kernel_stats.csv" | head -1)
cp $f $O/kernel_stats.csv
python3 - "$f" <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
for r in rows[:28]:
    print("%-90s calls %5s avg %10.1f us  %5s%%" % (r["Name"][:90], r["Calls"], float(r["AverageNs"])/1e3, r["Percentage"]))
PY
