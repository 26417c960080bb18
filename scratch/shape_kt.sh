#!/bin/bash
# kernel launches of one bench_configs shape (development aid): shape_kt.sh <shape>
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d gpurun_out/shape_kt -o kt --output-format csv -- python3 scratch/bench_configs.py $1 > gpurun_out/shape_kt.log 2>&1 || exit 1
python3 - <<'PY'
import csv
rows=list(csv.DictReader(open('gpurun_out/shape_kt/kt_kernel_trace.csv')))
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
# the last search: from the last k_cell_keys / first kernel after the last big gap
names=[r["Kernel_Name"] for r in rows]
# find last occurrence of a first-pass kernel
last=max(i for i,n in enumerate(names) if "k_open_range" in n or "k_batch_scan" in n and True)
# walk back to the start of that search: the previous hipMemset-like fill before the first scan kernel
start=last
while start>0 and (int(rows[start]["Start_Timestamp"])-int(rows[start-1]["End_Timestamp"]))<2_000_000: start-=1
t0=int(rows[start]["Start_Timestamp"])
tot=0
for r in rows[start:]:
    d=(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3
    tot+=d
    if d>30: print("%9.1f us  at %8.1f  %s grid %s" % (d,(int(r["Start_Timestamp"])-t0)/1e3,r["Kernel_Name"][:80],r["Grid_Size_X"]))
print("kernel time", tot, "span", (int(rows[-1]["End_Timestamp"])-t0)/1e3)
PY
