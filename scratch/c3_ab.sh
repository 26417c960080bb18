#!/bin/bash
# config 3 latency under the environment given (A/B aid): prints p50 / mean
cd $GRAFT_REPO_ROOT
timeout -k 10 300 python3 bench.py --config c3 --no-cpu > gpurun_out/c3_ab.json 2> gpurun_out/c3_ab.err || { tail -3 gpurun_out/c3_ab.err; exit 1; }
python3 -c "
import json; d=json.load(open('gpurun_out/c3_ab.json')); print('$1', d['latency_ms'], 'fps %.1f' % d['value'])"
