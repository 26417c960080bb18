#!/bin/bash
# usage: gpu_quick.sh "<pytest -k expression or file>"  -- selected GPU tests, then the default bench line without CPU legs
set -o pipefail
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 600 python -u -m pytest $1 -m gpu -x -q --timeout 300 --timeout-method thread > gpurun_out/gputest_quick.log 2>&1
rc=$?
tail -8 gpurun_out/gputest_quick.log
[ $rc -ne 0 ] && exit $rc
timeout -k 10 300 python bench.py --no-cpu > gpurun_out/bench_quick.json 2> gpurun_out/bench_quick.err || { tail -5 gpurun_out/bench_quick.err; exit 1; }
python - <<'PY'
import json
d=json.load(open('gpurun_out/bench_quick.json'))
print('value %.3f G/s  step %.3f ms  parts %s  frac %.4f  pipeline %.3f ms' % (d['value']/1e9, d['ms_per_step'], d['step_parts_ms'], d['roofline']['frac'], d.get('pipeline',{}).get('ms',0)))
PY
