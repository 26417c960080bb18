#!/bin/bash
# parity suite, uneven shapes, then config 4 under the knobs of the coarser levels (development aid)
cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -u -m pytest tests -m gpu -x -q --timeout 120 --timeout-method thread 2>&1 | tail -5 || exit 1
timeout -k 10 200 python scratch/bench_configs.py 2>&1 | grep -v amdgpu | tail -20 || exit 1
bash scratch/c4_sweep.sh PC_X=0 PC_AMD_NO_HINTS=1 "PC_AMD_UNEVEN_TILE1=2 PC_AMD_UNEVEN_CAP=2048" "PC_AMD_UNEVEN_TILE1=2" PC_AMD_UNEVEN_CAP=2048
