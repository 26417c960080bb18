#!/bin/bash
# LDS bank conflicts of the tile kernel, attributed by ablation builds (ablbuild/*.so: staging only, + filter
# and insertion, + exact finish = the full kernel): one counter pass per build.
set -o pipefail
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/pmc_lds
for v in ${PMC_VARIANTS:-FULL NOQUERY NOSCAN NOFINISH}; do
  if [ $v = FULL ]; then unset PC_AMD_LIB; else export PC_AMD_LIB=$GRAFT_REPO_ROOT/ablbuild/libpc_abl_$v.so; fi
  rm -rf gpurun_out/pmc_lds/$v
  timeout -s KILL 120 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_SALU -d gpurun_out/pmc_lds/$v -o p --output-format csv -- python3 bench.py --steps 2 --warmup 1 --no-cpu --no-pipeline > gpurun_out/pmc_lds/$v.log 2>&1 || { tail -3 gpurun_out/pmc_lds/$v.log; exit 1; }
done
python3 - <<'PY'
import collections, csv, glob, json
out = {}
import os
for v in os.environ.get("PMC_VARIANTS", "FULL NOQUERY NOSCAN NOFINISH").split():
    f = glob.glob(f"gpurun_out/pmc_lds/{v}/**/*counter_collection.csv", recursive=True)[0]
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if "k_tile_scan<Rec32, 18, 0" in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    out[v] = {k: sum(x) / len(x) for k, x in agg.items()}
    print(v, {k: "%.4g" % x for k, x in out[v].items()})
json.dump(out, open("gpurun_out/pmc_lds/summary.json", "w"), indent=1)
PY
