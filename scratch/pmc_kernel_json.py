"""gpurun_out/pmc_kernel (two counter passes of scratch/pmc_kernel.sh) -> a JSON summary per kernel name.
   python3 scratch/pmc_kernel_json.py <piece of the kernel names> <output.json> <description of the command>"""
import collections, csv, glob, json, sys

piece, out, what = sys.argv[1], sys.argv[2], sys.argv[3]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for d in ("a", "b"):
    for f in glob.glob(f"gpurun_out/pmc_kernel/{d}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if piece in r["Kernel_Name"]:
                agg[r["Kernel_Name"][:70]][r["Counter_Name"]].append(float(r["Counter_Value"]))
res = {"source": what, "kernels": {}}
for k, cs in sorted(agg.items()):
    m = {c: sum(v) / len(v) for c, v in sorted(cs.items())}
    w = m.get("SQ_WAVES", 0.0)
    e = {"counters_per_launch_mean": {c: int(v) for c, v in m.items()}}
    if w:
        e["per_wave"] = {c: round(v / w, 1) for c, v in m.items() if c != "SQ_WAVES"}
        if "SQ_ACTIVE_INST_VALU" in m and m.get("SQ_WAVE_CYCLES"):
            e["valu_issue_share_of_wave_cycles"] = round(m["SQ_ACTIVE_INST_VALU"] / m["SQ_WAVE_CYCLES"], 3)
    res["kernels"][k] = e
json.dump(res, open(out, "w"), indent=1)
print(json.dumps({k: v.get("valu_issue_share_of_wave_cycles") for k, v in res["kernels"].items()}, indent=1))
