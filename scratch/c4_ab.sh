#!/bin/bash
# config 4 under the environment given (A/B aid): step parts
cd $GRAFT_REPO_ROOT
timeout -k 10 400 python3 bench.py --config c4 --steps 2 --no-cpu > gpurun_out/c4_ab.json 2> gpurun_out/c4_ab.err || { tail -3 gpurun_out/c4_ab.err; exit 1; }
python3 -c "
import json; d=json.load(open('gpurun_out/c4_ab.json')); print('$1', {k: round(v,1) for k,v in d['step_parts_ms'].items()}, 'step %.1f' % d['ms_per_step'])"
