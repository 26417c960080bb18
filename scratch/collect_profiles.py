"""Copies the summaries of gpurun_out/prof (scratch/profile_round.sh) into profiles/ under the given prefix."""
import collections, csv, glob, json, shutil, sys
R = sys.argv[1] if len(sys.argv) > 1 else "r02"
P = "gpurun_out/prof"
d = json.load(open(f"{P}/bench.json"))
shutil.copy(f"{P}/bench.json", f"profiles/{R}_bench.json")
shutil.copy(f"{P}/kt/kt_kernel_stats.csv", f"profiles/{R}_bench_kernel_stats.csv")
for c in ("c2", "c3", "c4", "c5"):
    for src in (f"{P}/bench_{c}.json", f"gpurun_out/bench_{c}.json"):
        try:
            json.load(open(src))
            shutil.copy(src, f"profiles/{R}_bench_{c}.json")
            break
        except (OSError, ValueError):
            pass
KERNEL = "k_tile_scan<Rec32, 18, 0"
tr = {}
for dd in ("fetch", "write"):
    v = [float(r["Counter_Value"]) for r in csv.DictReader(open(f"{P}/{dd}/p_counter_collection.csv"))
         if KERNEL in r["Kernel_Name"]]
    tr[dd] = sum(v) / len(v)
t = {"source": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE (one pass each) over `python3 bench.py --steps 2 --warmup 1 "
               "--no-cpu --no-pipeline`, MI355X; per-launch means",
     "kernel": "k_tile_scan<Rec32,18,M_KNN> (10 M points, k=16; 4-wave workgroups, tile 4x4x2 cells of a 112^3 grid)",
     "points": 10_000_000, "k": 16,
     "note": "FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE counts 64-byte requests as 32 bytes, hence x2 "
             "(MI355X_MICROARCH.md, HBM / rocprofv3 section); WRITE_SIZE is exact",
     "FETCH_SIZE_KB": tr["fetch"], "WRITE_SIZE_KB": tr["write"],
     "traffic_bytes_per_launch": int((tr["fetch"] * 2 + tr["write"]) * 1024),
     "algorithmic_bytes_per_launch": 1_520_000_000}
t["traffic_over_algorithmic"] = t["traffic_bytes_per_launch"] / t["algorithmic_bytes_per_launch"]
json.dump(t, open(f"profiles/{R}_pmc_traffic.json", "w"), indent=1)
sq = {}
for dd in ("sq1", "sq2"):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(glob.glob(f"{P}/{dd}/*counter_collection.csv")[0])):
        if KERNEL in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in agg.items():
        sq[k] = sum(v) / len(v)
json.dump({"source": "rocprofv3 --pmc (two passes of 8 SQ counters) over `python3 bench.py --steps 2 --warmup 1 --no-cpu "
                     "--no-pipeline`, MI355X; per-launch means of k_tile_scan<Rec32,18,M_KNN> (10 M points, k=16)",
           "note": "wave-based SQ counters count quad-cycles; 1024 SIMDs",
           "counters": sq,
           "derived": {"valu_instructions_per_wave": sq["SQ_INSTS_VALU"] / sq["SQ_WAVES"],
                       "salu_instructions_per_wave": sq["SQ_INSTS_SALU"] / sq["SQ_WAVES"],
                       "lds_instructions_per_wave": sq["SQ_INSTS_LDS"] / sq["SQ_WAVES"],
                       "queries_per_wave": 10_000_000 / sq["SQ_WAVES"],
                       "valu_issue_cycles_per_simd": sq["SQ_ACTIVE_INST_VALU"] * 4 / 1024,
                       "wave_cycles_waiting_frac": sq["SQ_WAIT_ANY"] / sq["SQ_WAVE_CYCLES"],
                       "wave_cycles_issue_stalled_frac": sq["SQ_WAIT_INST_ANY"] / sq["SQ_WAVE_CYCLES"],
                       "lds_bank_conflict_frac_of_lds_cycles": sq["SQ_LDS_BANK_CONFLICT"] / sq["SQ_LDS_IDX_ACTIVE"]}},
          open(f"profiles/{R}_pmc_sq_tile_kernel.json", "w"), indent=1)
print(json.dumps({k: v for k, v in d.items() if k != "config"})[:1800])
print(tr, t["traffic_bytes_per_launch"], t["traffic_over_algorithmic"])
print(json.load(open(f"profiles/{R}_pmc_sq_tile_kernel.json"))["derived"])
for r in list(csv.DictReader(open(f"profiles/{R}_bench_kernel_stats.csv")))[:8]:
    print(r["Name"][:70], r["Calls"], int(float(r["AverageNs"])), r["Percentage"])
