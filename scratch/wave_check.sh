#!/bin/bash
# parity suite, uneven shapes and headline after a change of the wave-per-query kernel (development aid)
cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -u -m pytest tests -m gpu -x -q --timeout 120 --timeout-method thread 2>&1 | tail -5 || exit 1
timeout -k 10 200 python scratch/bench_configs.py 2>&1 | grep -v amdgpu | tail -20 || exit 1
timeout -k 10 100 python bench.py --no-cpu --no-pipeline --steps 20 > gpurun_out/b_wave.json || exit 1
python -c "import json;d=json.load(open('gpurun_out/b_wave.json'));print(d['roofline']['kernel_ms'],d['ms_per_step'])"
