#!/bin/bash
set -o pipefail
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/ub
rm -rf $O && mkdir -p $O
timeout -s KILL 90 rocprofv3 --pmc FETCH_SIZE -d $O/fetch -o p --output-format csv -- ./scratch/ubuild 10000000 16 11 > $O/fetch.log 2>&1 || exit 1
timeout -s KILL 90 rocprofv3 --pmc WRITE_SIZE -d $O/write -o p --output-format csv -- ./scratch/ubuild 10000000 16 11 > $O/write.log 2>&1 || exit 1
python3 - <<'PY'
import csv,glob,collections
for what in ("fetch","write"):
    f=glob.glob(f"gpurun_out/ub/{what}/**/*counter_collection.csv",recursive=True)
    acc=collections.defaultdict(list)
    for path in f:
        for row in csv.DictReader(open(path)):
            acc[row["Kernel_Name"][:60]].append(float(row["Counter_Value"]))
    for k,v in acc.items():
        print(what,k,"calls",len(v),"mean KiB",sum(v)/len(v),"last",v[-1])
PY
