#!/bin/bash
for f in ${SCALES:-0.9 0.95 1.0 1.05 1.1 1.2}; do
  PC_AMD_CELL_SCALE=$f timeout -k 10 120 python bench.py --steps 8 --warmup 2 --no-cpu --no-pipeline > gpurun_out/s.json 2>/dev/null || exit 1
  python -c "
import json
r=json.load(open('gpurun_out/s.json')); print('scale $f', round(r['value']/1e9,3),'Gq/s', round(r['ms_per_step'],3),'ms/step', round(r['roofline']['kernel_ms'],3),'knn ms')"
done
