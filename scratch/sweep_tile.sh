#!/bin/bash
for T in ${TILES:-5,3,2 4,4,2 4,2,4 2,4,4 3,3,3 4,3,3 5,3,3 4,4,3 3,3,4}; do
  PC_AMD_TILE=$T PC_AMD_DEBUG=1 timeout -k 10 120 python bench.py --steps 8 --warmup 2 --no-cpu --no-pipeline > gpurun_out/s.json 2> gpurun_out/s.err || exit 1
  python -c "
import json
r=json.load(open('gpurun_out/s.json')); print('tile $T', round(r['ms_per_step'],3),'ms/step', round(r['roofline']['kernel_ms'],3),'knn ms')"
  grep -m1 "tile pass" gpurun_out/s.err
done
