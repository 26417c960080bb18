#!/bin/bash
# Round profile: default bench line, rocprofv3 kernel stats of the same command, HBM traffic counters, SQ counters.
set -o pipefail
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/prof
rm -rf $O && mkdir -p $O
timeout -k 10 300 python3 bench.py > $O/bench.json 2> $O/bench.err || exit 1
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/kt -o kt --output-format csv -- python3 bench.py > $O/bench_profiled.json 2> $O/kt.log || exit 1
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE -d $O/fetch -o p --output-format csv -- python3 bench.py --steps 2 --warmup 1 --no-cpu --no-pipeline > $O/fetch.log 2>&1 || exit 1
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE -d $O/write -o p --output-format csv -- python3 bench.py --steps 2 --warmup 1 --no-cpu --no-pipeline > $O/write.log 2>&1 || exit 1
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_INST_LDS -d $O/sq1 -o p --output-format csv -- python3 bench.py --steps 2 --warmup 1 --no-cpu --no-pipeline > $O/sq1.log 2>&1 || exit 1
timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_WAVES SQ_INSTS_VMEM_RD -d $O/sq2 -o p --output-format csv -- python3 bench.py --steps 2 --warmup 1 --no-cpu --no-pipeline > $O/sq2.log 2>&1 || exit 1
for c in c2 c3 c5; do timeout -k 10 400 python3 bench.py --config $c > $O/bench_$c.json 2> $O/bench_$c.err || exit 1; done
find $O -name "*.csv" | head -20
cut -c1-300 $O/bench.json
