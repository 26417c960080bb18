#!/bin/bash
# SQ counters of the kernels whose name contains $1, over `bench.py <rest of the arguments>` (two passes of 8)
set -o pipefail
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
piece=$1; shift
O=gpurun_out/pmc_kernel
rm -rf $O && mkdir -p $O
timeout -s KILL 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVES -d $O/a -o p --output-format csv -- python3 bench.py "$@" > $O/a.log 2>&1 || { tail -3 $O/a.log; exit 1; }
timeout -s KILL 600 rocprofv3 --pmc SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_SMEM -d $O/b -o p --output-format csv -- python3 bench.py "$@" > $O/b.log 2>&1 || { tail -3 $O/b.log; exit 1; }
python3 - "$O" "$piece" <<'PY'
import collections, csv, glob, sys
O, piece = sys.argv[1], sys.argv[2]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for d in ("a", "b"):
    for f in glob.glob(f"{O}/{d}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if piece in r["Kernel_Name"]:
                agg[r["Kernel_Name"][:70]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, cs in agg.items():
    print(k)
    m = {c: sum(v) / len(v) for c, v in cs.items()}
    for c, v in sorted(m.items()):
        print("   %-22s %16.0f" % (c, v), ("  per wave %10.1f" % (v / m["SQ_WAVES"])) if "SQ_WAVES" in m and m["SQ_WAVES"] else "")
PY
