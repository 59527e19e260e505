#!/bin/bash
# Scheduling sweep on the GPU box (from the repo root): one bench line per configuration of the worker / stream knobs.
set -u
run() { name=$1; shift; env "$@" python bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('$name', round(d['ms_per_step'],2), {k:round(v,1) for k,v in d['stages_ms'].items()})"; }
run default A=1
run s3 CKM_SIDE_STREAMS=3
run s2 CKM_SIDE_STREAMS=2
run s5 CKM_SIDE_STREAMS=5
run default2 A=1
