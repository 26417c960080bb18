#!/bin/bash
for K in ${KS:-1 2 4 8 16 20 30 32}; do
  timeout -k 10 120 python bench.py --steps 6 --warmup 2 --no-cpu --no-pipeline --k $K > gpurun_out/s.json 2>/dev/null || exit 1
  python -c "
import json
r=json.load(open('gpurun_out/s.json')); print('k $K', round(r['ms_per_step'],3),'ms/step', round(r['roofline']['kernel_ms'],3),'knn ms')"
done
