"""Copies the summaries of gpurun_out/prof and gpurun_out/pmc_sq* into profiles/ (round-1 names)."""
import collections, csv, glob, json, shutil
d = json.load(open("gpurun_out/prof/bench.json"))
shutil.copy("gpurun_out/prof/bench.json", "profiles/r01_bench.json")
shutil.copy("gpurun_out/prof/kt/kt_kernel_stats.csv", "profiles/r01_bench_kernel_stats.csv")
tr = {}
for dd in ("fetch", "write"):
    v = [float(r["Counter_Value"]) for r in csv.DictReader(open(f"gpurun_out/prof/{dd}/p_counter_collection.csv"))
         if "k_tile_scan" in r["Kernel_Name"]]
    tr[dd] = sum(v) / len(v)
t = json.load(open("profiles/r01_pmc_traffic.json"))
t["kernel"] = "k_tile_scan<Rec32,18,M_KNN> (10 M points, k=16; 4-wave workgroups, tile 4x3x3)"
t["FETCH_SIZE_KB"], t["WRITE_SIZE_KB"] = tr["fetch"], tr["write"]
t["traffic_bytes_per_launch"] = int((tr["fetch"] * 2 + tr["write"]) * 1024)
json.dump(t, open("profiles/r01_pmc_traffic.json", "w"), indent=1)
sq = {}
for dd in ("pmc_sq1", "pmc_sq2"):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(glob.glob(f"gpurun_out/{dd}/*counter_collection.csv")[0])):
        if "k_tile_scan" in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in agg.items():
        sq[k] = sum(v) / len(v)
json.dump({"source": "rocprofv3 --pmc (two passes of 8 SQ counters) over `python3 bench.py --steps 2 --warmup 1 --no-cpu --no-pipeline`, MI355X; per-launch means of k_tile_scan<Rec32,18,M_KNN> (10 M points, k=16)",
           "note": "wave-based SQ counters count quad-cycles; 1024 SIMDs",
           "counters": sq,
           "derived": {"valu_instructions_per_wave": sq["SQ_INSTS_VALU"] / sq["SQ_WAVES"],
                       "valu_issue_cycles_per_simd": sq["SQ_ACTIVE_INST_VALU"] * 4 / 1024,
                       "wave_cycles_waiting_frac": sq["SQ_WAIT_ANY"] / sq["SQ_WAVE_CYCLES"],
                       "wave_cycles_issue_stalled_frac": sq["SQ_WAIT_INST_ANY"] / sq["SQ_WAVE_CYCLES"],
                       "lds_bank_conflict_frac_of_lds_cycles": sq["SQ_LDS_BANK_CONFLICT"] / sq["SQ_LDS_IDX_ACTIVE"]}},
          open("profiles/r01_pmc_sq_tile_kernel.json", "w"), indent=1)
print(json.dumps({k: v for k, v in d.items() if k != "config"})[:1500])
print(tr, t["traffic_bytes_per_launch"])
print(json.load(open("profiles/r01_pmc_sq_tile_kernel.json"))["derived"])
for r in list(csv.DictReader(open("profiles/r01_bench_kernel_stats.csv")))[:6]:
    print(r["Name"][:70], r["Calls"], int(float(r["AverageNs"])), r["Percentage"])
