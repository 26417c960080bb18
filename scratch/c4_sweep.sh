#!/bin/bash
# config 4 at full size under the pass-choice knobs (development aid)
cd $GRAFT_REPO_ROOT
run() {
  echo -n "$1: "
  env $1 timeout -k 10 300 python bench.py --config c4 --steps 1 --no-cpu 2>/dev/null | python -c "import json,sys;d=json.loads(sys.stdin.read());print({k:round(v,1) for k,v in d['step_parts_ms'].items()}, round(d['ms_per_step'],1))" || exit 1
}
for v in "$@"; do run "$v"; done
