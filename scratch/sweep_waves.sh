#!/bin/bash
# headline kNN time and fused-pipeline time for 1..4 waves per tile workgroup
for W in ${WAVES:-1 2 3 4}; do
  PC_AMD_TILE_WAVES=$W PC_AMD_DEBUG=1 timeout -k 10 120 python bench.py --steps 10 --warmup 2 --no-cpu > gpurun_out/w$W.json 2> gpurun_out/w$W.err || exit 1
  grep -m1 "tile pass" gpurun_out/w$W.err
  python -c "
import json;d=json.load(open('gpurun_out/w$W.json'));print($W, d['ms_per_step'],d['value'],d['roofline']['kernel_ms'], d.get('pipeline'))"
done
