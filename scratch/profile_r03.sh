#!/bin/bash
# The round's evidence in one GPU call: bench lines, rocprofv3 kernel statistics and HBM counters for the headline
# and for the configurations c2..c5, SQ counters of the tile kernel.  Everything lands under gpurun_out/r03/.
#   scratch/profile_r03.sh [quick|c4|only-<config>]   (quick: without config c4, which generates 100 M points four times; c4: only
#                                        config c4, added to what a quick run left -- the two fit a GPU call each)
set -o pipefail
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03
step() { echo "== $*" >&2; }
if [ "$1" != c4 ] && [ "${1#only-}" = "$1" ]; then
rm -rf $O && mkdir -p $O
step bench; timeout -k 10 400 python3 bench.py > $O/bench.json 2> $O/bench.err || { tail -5 $O/bench.err; exit 1; }
step kt; timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $O/kt -o kt --output-format csv -- python3 bench.py > $O/bench_profiled.json 2> $O/kt.log || exit 1
B="--steps 2 --warmup 1 --no-cpu"
for c in FETCH_SIZE WRITE_SIZE; do
  step $c; timeout -s KILL 200 rocprofv3 --pmc $c -d $O/$c -o p --output-format csv -- python3 bench.py $B > $O/$c.log 2>&1 || exit 1
done
step sq1; timeout -s KILL 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_INST_LDS -d $O/sq1 -o p --output-format csv -- python3 bench.py $B --no-pipeline > $O/sq1.log 2>&1 || exit 1
step sq2; timeout -s KILL 200 rocprofv3 --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_WAVES SQ_INSTS_VMEM_RD -d $O/sq2 -o p --output-format csv -- python3 bench.py $B --no-pipeline > $O/sq2.log 2>&1 || exit 1
fi
mkdir -p $O
CONFIGS="c2 c3 c5"; [ "$1" = quick ] || CONFIGS="c2 c3 c5 c4"; [ "$1" = c4 ] && CONFIGS=c4; [ "${1#only-}" != "$1" ] && CONFIGS=${1#only-}
for c in $CONFIGS; do
  X=""; [ $c = c3 ] && X="--frames 300"
  step "bench $c"; timeout -k 10 900 python3 bench.py --config $c > $O/bench_$c.json 2> $O/bench_$c.err || { tail -5 $O/bench_$c.err; exit 1; }
  step "kt $c"; timeout -k 10 900 rocprofv3 --kernel-trace --stats -d $O/kt_$c -o kt --output-format csv -- python3 bench.py --config $c --no-cpu $X > $O/kt_$c.json 2> $O/kt_$c.log || exit 1
  for m in FETCH_SIZE WRITE_SIZE; do
    step "$m $c"; timeout -s KILL 900 rocprofv3 --pmc $m -d $O/${m}_$c -o p --output-format csv -- python3 bench.py --config $c --no-cpu --steps 1 $X > $O/${m}_$c.json 2> $O/${m}_$c.log || exit 1
  done
done
find $O -name "*.csv" | wc -l
