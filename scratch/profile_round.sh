#!/bin/bash
# Round profile: default bench line, rocprofv3 kernel stats of the same command, HBM traffic counters.
set -o pipefail
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/prof
rm -rf $O && mkdir -p $O
timeout -k 10 300 python3 bench.py > $O/bench.json 2> $O/bench.err || exit 1
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/kt -o kt --output-format csv -- python3 bench.py > $O/bench_profiled.json 2> $O/kt.log || exit 1
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE -d $O/fetch -o p --output-format csv -- python3 bench.py --steps 2 --warmup 1 --no-cpu --no-pipeline > $O/fetch.log 2>&1 || exit 1
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE -d $O/write -o p --output-format csv -- python3 bench.py --steps 2 --warmup 1 --no-cpu --no-pipeline > $O/write.log 2>&1 || exit 1
find $O -name "*.csv" | head -20
cut -c1-400 $O/bench.json
