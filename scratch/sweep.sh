#!/bin/bash
for f in 1.0 1.03 1.06 1.1 1.15 1.2; do
  echo "== cell scale $f"
  PC_AMD_CELL_SCALE=$f timeout -k 10 120 python bench.py --steps 5 --warmup 2 --no-cpu --no-pipeline | python -c "
import sys,json
r=json.loads(sys.stdin.read()); print(round(r['value']/1e9,3),'Gq/s', round(r['ms_per_step'],3),'ms/step', round(r['roofline']['kernel_ms'],3),'knn ms')"
done
