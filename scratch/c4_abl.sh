#!/bin/bash
# first tile pass of config 4 under the ablation builds (timing only: the ablated kernels answer nothing)
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
for v in ${C4_ABL:-FULL NOFINISH NOSCAN NOQUERY}; do
  if [ $v = FULL ]; then unset PC_AMD_LIB; else export PC_AMD_LIB=$GRAFT_REPO_ROOT/ablbuild/libpc_abl_$v.so; fi
  rm -rf gpurun_out/c4_abl_$v
  timeout -k 10 400 rocprofv3 --kernel-trace -d gpurun_out/c4_abl_$v -o kt --output-format csv -- python3 bench.py --config c4 --steps 1 --no-cpu > gpurun_out/c4_abl_$v.json 2> gpurun_out/c4_abl_$v.err || { tail -3 gpurun_out/c4_abl_$v.err; exit 1; }
  python3 - $v <<'PY'
import csv, sys
v = sys.argv[1]
rows = list(csv.DictReader(open(f'gpurun_out/c4_abl_{v}/kt_kernel_trace.csv')))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
import os
kern = os.environ.get("C4_KERN", "k_tile_scan")
t = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6 for r in rows if kern in r["Kernel_Name"]]
print(v, kern, "launches (ms):", ["%.2f" % x for x in t])
PY
done
