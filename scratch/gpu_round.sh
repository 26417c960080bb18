#!/bin/bash
# GPU tests (one process), then the default bench line.
set -o pipefail
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread > gpurun_out/gputest.log 2>&1
rc=$?
tail -15 gpurun_out/gputest.log
[ $rc -ne 0 ] && exit $rc
timeout -k 10 300 python bench.py > gpurun_out/bench.json 2> gpurun_out/bench.err || { tail -5 gpurun_out/bench.err; exit 1; }
cut -c1-1500 gpurun_out/bench.json
