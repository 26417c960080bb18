"""Copies the summaries of gpurun_out/r03 (scratch/profile_r03.sh) into profiles/ under the prefix r03 and derives
the per-step figures bench.py quotes: HBM traffic per launch / frame / step / iteration, launches per step, shares."""
import collections, csv, glob, json, os, shutil, sys

R = "r03"
P = "gpurun_out/r03"


def counter_csv(d):
    f = glob.glob(f"{P}/{d}/**/*counter_collection.csv", recursive=True)
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f[0])):
        agg[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    return agg


def counters_by_name(d):
    f = glob.glob(f"{P}/{d}/**/*counter_collection.csv", recursive=True)
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f[0])):
        agg[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return agg


def kernel_stats(d):
    f = glob.glob(f"{P}/{d}/**/*kernel_stats.csv", recursive=True)[0]
    return f, list(csv.DictReader(open(f)))


def find(agg, piece):
    for k in agg:
        if piece in k:
            return k
    raise KeyError(piece)


NOTE = ("FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE counts a 128-byte request as 64 bytes, hence x 2 "
        "(MI355X_MICROARCH.md, HBM / rocprofv3 section); WRITE_SIZE is exact for 16-byte-per-lane stores")

# ---- headline -------------------------------------------------------------------------------------------
d = json.load(open(f"{P}/bench.json"))
shutil.copy(f"{P}/bench.json", f"profiles/{R}_bench.json")
f, rows = kernel_stats("kt")
shutil.copy(f, f"profiles/{R}_bench_kernel_stats.csv")
fetch, write = counter_csv("FETCH_SIZE"), counter_csv("WRITE_SIZE")
for tag, piece, alg, name in (("", "k_tile_scan<Rec32, 18, 0", 1_520_000_000, "k_tile_scan<Rec32,18,M_KNN>"),
                              ("_pipeline", "k_tile_scan<Rec32, 18, 5", 1_690_000_000 + 240_000_000, "k_tile_scan<Rec32,18,M_ALL>")):
    kf, kw = find(fetch, piece), find(write, piece)
    fk, wk = sum(fetch[kf]) / len(fetch[kf]), sum(write[kw]) / len(write[kw])
    t = {"source": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE (one pass each) over `python3 bench.py --steps 2 --warmup 1 "
                   "--no-cpu`, MI355X; per-launch means",
         "kernel": name + " (10 M points, k = 16; 4-wave workgroups)", "points": 10_000_000, "k": 16, "note": NOTE,
         "FETCH_SIZE_KB": fk, "WRITE_SIZE_KB": wk, "traffic_bytes_per_launch": int((fk * 2 + wk) * 1024),
         "algorithmic_bytes_per_launch": alg}
    if tag:
        t["algorithmic_bytes_note"] = "169 B/point of the pipeline + 24 B/point: normals are written as fp64 (24 B), not the 12 B of the formula"
    t["traffic_over_algorithmic"] = t["traffic_bytes_per_launch"] / alg
    json.dump(t, open(f"profiles/{R}_pmc_traffic{tag}.json", "w"), indent=1)
    print(name, t["traffic_bytes_per_launch"], round(t["traffic_over_algorithmic"], 3))
build = {}
for piece in ("k_bucket_sort", "k_bucket_scatter", "k_bucket_hist", "k_bbox<float>", "k_bbox_final", "k_col_group_sums", "k_col_scan_groups",
              "k_col_apply", "k_bucket_starts"):
    try:
        kf, kw = find(fetch, piece), find(write, piece)
        build[piece] = {"fetch_bytes": int(sum(fetch[kf]) / len(fetch[kf]) * 2048), "write_bytes": int(sum(write[kw]) / len(write[kw]) * 1024)}
    except KeyError:
        pass
times = {r["Name"]: float(r["AverageNs"]) / 1e3 for r in rows}
for piece in build:
    build[piece]["avg_us"] = round(times[find(times, piece)], 1)
json.dump({"source": "the same passes; kernels of the index build (pc_build.hip), per launch", "note": NOTE,
           "algorithmic_bytes": 12 * 10_000_000 + 16 * 10_000_000 + 4 * 10_000_000 + 4 * 1_404_928,
           "kernels": build, "sum_us": round(sum(b["avg_us"] for b in build.values()), 1),
           "sum_traffic_bytes": sum(b["fetch_bytes"] + b["write_bytes"] for b in build.values())},
          open(f"profiles/{R}_pmc_traffic_build.json", "w"), indent=1)
sq = {}
for dd in ("sq1", "sq2"):
    agg = counters_by_name(dd)
    k = find(agg, "k_tile_scan<Rec32, 18, 0")
    for c, v in agg[k].items():
        sq[c] = sum(v) / len(v)
kern_us = times[find(times, "k_tile_scan<Rec32, 18, 0")]
json.dump({"source": "rocprofv3 --pmc (two passes of 8 SQ counters) over `python3 bench.py --steps 2 --warmup 1 --no-cpu "
                     "--no-pipeline`, MI355X; per-launch means of k_tile_scan<Rec32,18,M_KNN> (10 M points, k = 16)",
           "note": "wave-based SQ counters count quad-cycles; 1024 SIMDs; kernel_us from the kernel-trace run of the round",
           "counters": sq, "kernel_us": kern_us,
           "derived": {"valu_instructions_per_wave": sq["SQ_INSTS_VALU"] / sq["SQ_WAVES"],
                       "salu_instructions_per_wave": sq["SQ_INSTS_SALU"] / sq["SQ_WAVES"],
                       "lds_instructions_per_wave": sq["SQ_INSTS_LDS"] / sq["SQ_WAVES"],
                       "queries_per_wave": 10_000_000 / sq["SQ_WAVES"],
                       "valu_issue_cycles_per_simd": sq["SQ_ACTIVE_INST_VALU"] * 4 / 1024,
                       "valu_busy_frac_of_kernel": sq["SQ_ACTIVE_INST_VALU"] * 4 / 1024 / (kern_us * 1e-6 * 2.4e9),
                       "wave_cycles_waiting_frac": sq["SQ_WAIT_ANY"] / sq["SQ_WAVE_CYCLES"],
                       "wave_cycles_issue_stalled_frac": sq["SQ_WAIT_INST_ANY"] / sq["SQ_WAVE_CYCLES"],
                       "lds_bank_conflict_frac_of_lds_cycles": sq["SQ_LDS_BANK_CONFLICT"] / sq["SQ_LDS_IDX_ACTIVE"]}},
          open(f"profiles/{R}_pmc_sq_tile_kernel.json", "w"), indent=1)
print(json.load(open(f"profiles/{R}_pmc_sq_tile_kernel.json"))["derived"])

# ---- the other configurations ----------------------------------------------------------------------------
STEP_KERNEL = {"c3": "k_voxel_reduce", "c4": "k_voxel_reduce", "c5": "k_transform", "c2": None}
SEARCH = ("k_tile_scan", "k_batch_scan", "k_ring_scan", "k_wave_query", "k_scan<", "k_collect_flagged", "k_open_flags", "k_open_range",
          "k_tile_full", "k_tile_postpone", "k_flag_hist", "k_tile_nonempty", "k_tile_ids", "k_key_heads", "k_take", "k_compact",
          "k_fill_slots")
BINNING = ("k_cell_keys32<", "k_radix", "k_scan_", "k_gather_records")
for c in ("c2", "c3", "c4", "c5"):
    if not os.path.exists(f"{P}/bench_{c}.json"):
        continue
    try:
        json.load(open(f"{P}/bench_{c}.json"))
    except ValueError:
        continue
    shutil.copy(f"{P}/bench_{c}.json", f"profiles/{R}_bench_{c}.json")
    f, rows = kernel_stats(f"kt_{c}")
    shutil.copy(f, f"profiles/{R}_bench_{c}_kernel_stats.csv")
    fetch, write = counter_csv(f"FETCH_SIZE_{c}"), counter_csv(f"WRITE_SIZE_{c}")
    out = {"source": f"rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (one pass each) over `python3 bench.py --config {c} --no-cpu --steps 1`"
                     + (" --frames 300" if c == "c3" else "") + "; kernel times and launch counts from `rocprofv3 --kernel-trace --stats` of "
                     f"`python3 bench.py --config {c} --no-cpu`" + (" --frames 300" if c == "c3" else "") + ", MI355X", "note": NOTE}
    total_ns = sum(float(r["TotalDurationNs"]) for r in rows)
    calls = sum(int(r["Calls"]) for r in rows)
    if c == "c2":
        for tag, piece in (("knn", "k_tile_scan<Rec32, 18, 0"), ("pipeline", "k_tile_scan<Rec32, 18, 5")):
            kf, kw = find(fetch, piece), find(write, piece)
            out[f"traffic_bytes_per_launch_{tag}"] = int((sum(fetch[kf]) / len(fetch[kf]) * 2 + sum(write[kw]) / len(write[kw])) * 1024)
        out["traffic_bytes_per_launch"] = out["traffic_bytes_per_launch_knn"]
        out["points"], out["k"] = 1_000_000, 16
    else:
        sk = STEP_KERNEL[c]
        steps_kt = next(int(r["Calls"]) for r in rows if sk in r["Name"])
        steps_pmc = len(fetch[find(fetch, sk)])
        all_traffic = (sum(sum(v) for v in fetch.values()) * 2 + sum(sum(v) for v in write.values())) * 1024
        out["steps_in_kernel_trace"], out["steps_in_counter_passes"] = steps_kt, steps_pmc
        out["launches_per_step"] = calls / steps_kt
        out["kernel_time_us_per_step"] = total_ns / 1e3 / steps_kt
        out["traffic_bytes_per_step_all_kernels"] = int(all_traffic / steps_pmc)
        if c in ("c4", "c5"):
            st = (sum(sum(v) for k, v in fetch.items() if any(s in k for s in SEARCH + (BINNING if c == "c5" else ()))) * 2 +
                  sum(sum(v) for k, v in write.items() if any(s in k for s in SEARCH + (BINNING if c == "c5" else ())))) * 1024
            out["traffic_bytes_per_step_search"] = int(st / steps_pmc)
        if c == "c5":
            bt = sum(float(r["TotalDurationNs"]) for r in rows if any(s in r["Name"] for s in BINNING))
            st_ = sum(float(r["TotalDurationNs"]) for r in rows if any(s in r["Name"] for s in SEARCH + BINNING))
            out["binning_share_of_search_kernel_time"] = bt / st_
            out["search_kernel_time_us_per_iteration"] = st_ / 1e3 / steps_kt
        out["traffic_bytes_per_step"] = out.get("traffic_bytes_per_step_search", out["traffic_bytes_per_step_all_kernels"])
    top = sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:8]
    out["top_kernels"] = [{"name": r["Name"][:90], "calls": int(r["Calls"]), "avg_us": round(float(r["AverageNs"]) / 1e3, 1),
                           "share": float(r["Percentage"])} for r in top]
    json.dump(out, open(f"profiles/{R}_pmc_traffic_{c}.json", "w"), indent=1)
    print(c, {k: v for k, v in out.items() if k not in ("source", "note", "top_kernels")})
