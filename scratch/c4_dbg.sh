#!/bin/bash
# kernel launches and level log of the full-size config 4, knobs from the environment (development aid)
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
export PC_AMD_DEBUG=1
timeout -k 10 600 rocprofv3 --kernel-trace --stats -d gpurun_out/c4_kt -o kt --output-format csv -- python3 bench.py --config c4 --steps 1 --no-cpu > gpurun_out/c4_kt.json 2> gpurun_out/c4_kt.err || exit 1
grep "pc_amd" gpurun_out/c4_kt.err | tail -24
python3 - <<'PY'
import csv, json
rows=list(csv.DictReader(open('gpurun_out/c4_kt/kt_kernel_trace.csv')))
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
rows=rows[len(rows)//2:]
for r in rows:
    n=r["Kernel_Name"]
    d=(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e6
    if d>1.0: print(n[28:75].ljust(50), "%8.2f ms"%d, "grid", r["Grid_Size_X"], r["Workgroup_Size_X"], "lds", r["LDS_Block_Size"])
d=json.load(open('gpurun_out/c4_kt.json')); print(d['step_parts_ms'])
PY
