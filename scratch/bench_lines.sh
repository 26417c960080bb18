#!/bin/bash
# the bench lines of all configurations, after the round's counter summaries are in profiles/ (they quote them)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03_lines
rm -rf $O && mkdir -p $O
timeout -k 10 400 python3 bench.py > $O/bench.json 2> $O/bench.err || { tail -3 $O/bench.err; exit 1; }
for c in c2 c3 c5 c4; do
  echo "== $c" >&2
  timeout -k 10 900 python3 bench.py --config $c > $O/bench_$c.json 2> $O/bench_$c.err || { tail -3 $O/bench_$c.err; exit 1; }
done
ls $O
