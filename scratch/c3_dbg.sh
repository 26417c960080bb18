#!/bin/bash
# kernel timeline and level log of one frame of config 3 (development aid)
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
export PC_AMD_DEBUG=1
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d gpurun_out/c3_kt -o kt --output-format csv -- python3 bench.py --config c3 --frames 4 --distinct-frames 2 --no-cpu > gpurun_out/c3_kt.json 2> gpurun_out/c3_kt.err || exit 1
grep "pc_amd" gpurun_out/c3_kt.err | tail -${LOGLINES:-30}
python3 - <<'PY'
import csv, json
rows=list(csv.DictReader(open('gpurun_out/c3_kt/kt_kernel_trace.csv')))
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
# last frame: from the last k_voxel_reduce's predecessor bbox
last=max(i for i,r in enumerate(rows) if 'k_bbox<float>' in r["Kernel_Name"])
rows=rows[last:]
t0=int(rows[0]["Start_Timestamp"]); prev=t0; busy=0
for r in rows:
    s,e=int(r["Start_Timestamp"]),int(r["End_Timestamp"])
    n=r["Kernel_Name"].replace("(anonymous namespace)::","").replace("void ","")
    print("%8.1f +%5.1f gap %5.1f  %s grid %s"%((s-t0)/1e3,(e-s)/1e3,(s-prev)/1e3,n[:70],r["Grid_Size_X"]))
    prev=e; busy+=e-s
print("frame span %.1f us, busy %.1f us, launches %d"%((prev-t0)/1e3,busy/1e3,len(rows)))
PY
