#!/bin/bash
# HBM traffic counters (FETCH_SIZE, WRITE_SIZE: one pass each) of a bench configuration, per kernel.
#   scratch/pmc_config.sh <tag> <bench args...>      -> gpurun_out/pmc_<tag>/summary.json
set -o pipefail
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
tag=$1; shift
O=gpurun_out/pmc_$tag
rm -rf $O && mkdir -p $O
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -s KILL 900 rocprofv3 --pmc $c -d $O/$c -o p --output-format csv -- python3 bench.py "$@" > $O/$c.json 2> $O/$c.log || { tail -3 $O/$c.log; exit 1; }
done
python3 - "$O" "$@" <<'PY'
import collections, csv, glob, json, sys
O = sys.argv[1]
out = {"bench_args": sys.argv[2:], "kernels": {}}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    f = glob.glob(f"{O}/{c}/**/*counter_collection.csv", recursive=True)[0]
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        agg[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    for k, v in agg.items():
        e = out["kernels"].setdefault(k, {})
        e["calls"] = len(v)
        e[c + "_KiB_sum"] = sum(v)
json.dump(out, open(f"{O}/summary.json", "w"), indent=1)
tot_f = sum(e.get("FETCH_SIZE_KiB_sum", 0) for e in out["kernels"].values())
tot_w = sum(e.get("WRITE_SIZE_KiB_sum", 0) for e in out["kernels"].values())
print("all kernels: FETCH %.1f MiB (x2 on gfx950), WRITE %.1f MiB" % (tot_f / 1024, tot_w / 1024))
for k, e in sorted(out["kernels"].items(), key=lambda kv: -(2 * kv[1].get("FETCH_SIZE_KiB_sum", 0) + kv[1].get("WRITE_SIZE_KiB_sum", 0)))[:12]:
    print("%-80s calls %6d fetch %10.1f MiB write %10.1f MiB" % (k[:80], e["calls"], e.get("FETCH_SIZE_KiB_sum", 0) / 1024, e.get("WRITE_SIZE_KiB_sum", 0) / 1024))
PY
