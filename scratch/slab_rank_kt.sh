#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
export PC_AMD_DEBUG=1
timeout -k 10 500 rocprofv3 --kernel-trace -d gpurun_out/slab_kt -o kt --output-format csv -- python3 scratch/slab_rank.py 3 8 > gpurun_out/slab_kt.out 2> gpurun_out/slab_kt.err || exit 1
grep "^[0-9]" gpurun_out/slab_kt.out
grep pc_amd gpurun_out/slab_kt.err | tail -14 | cut -c1-160
python3 - <<'PY'
import csv
rows=list(csv.DictReader(open('gpurun_out/slab_kt/kt_kernel_trace.csv')))
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
last=max(i for i,r in enumerate(rows) if 'k_select' in r["Kernel_Name"] or 'k_slab_select' in r["Kernel_Name"] or 'select_slab' in r["Kernel_Name"])
rows=rows[last:]
t0=int(rows[0]["Start_Timestamp"]); tot=0
for r in rows:
    d=(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e6; tot+=d
    if d>0.15: print("%8.2f +%6.2f ms %s grid %s"%((int(r["Start_Timestamp"])-t0)/1e6,d,r["Kernel_Name"].replace("(anonymous namespace)::","")[:70],r["Grid_Size_X"]))
print("kernel time", tot, "span", (int(rows[-1]["End_Timestamp"])-t0)/1e6, "launches", len(rows))
PY
