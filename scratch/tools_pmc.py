"""Per-kernel averages of rocprofv3 --pmc counter CSVs (development aid)."""
import collections
import csv
import glob
import sys

for d in sys.argv[1:]:
    files = glob.glob(d + "/*counter_collection.csv") + glob.glob(d + "/*/*counter_collection.csv")
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    n = collections.Counter()
    for r in csv.DictReader(open(files[0])):
        k = r["Kernel_Name"][:64]
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        n[(k, r["Counter_Name"])] += 1
    for k, v in agg.items():
        if "k_tile_scan" in k or "k_wave_query" in k or "k_scan" in k:
            print(k, {a: f"{b / n[(k, a)]:.4g}" for a, b in v.items()})
