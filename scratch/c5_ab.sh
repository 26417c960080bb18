#!/bin/bash
# config 5 under the environment given (A/B aid): ms per iteration
cd $GRAFT_REPO_ROOT
timeout -k 10 300 python3 bench.py --config c5 --no-cpu > gpurun_out/c5_ab.json 2> gpurun_out/c5_ab.err || { tail -3 gpurun_out/c5_ab.err; exit 1; }
python3 -c "
import json; d=json.load(open('gpurun_out/c5_ab.json')); print('$1', 'ms/iteration %.3f' % d['ms_per_step'], 'Gq/s %.3f' % (d['value']/1e9))"
