#!/bin/bash
# Scheduling sweep on the GPU box (from the repo root): one bench line per configuration of the worker / stream knobs.
#   tools/sweep_sched.sh "NAME VAR=VALUE ..." ...      (default: a sweep of the residue shares of the three length classes)
set -u
run() { name=$1; shift; env "$@" python bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('$name', round(d['ms_per_step'],2), {k:round(v,1) for k,v in d['stages_ms'].items()})"; }
if [ $# -eq 0 ]; then set -- "base A=1" "s18_66 CKM_SHARES=0.18,0.66" "s20_70 CKM_SHARES=0.20,0.70" "s14_58 CKM_SHARES=0.14,0.58" "s16_68 CKM_SHARES=0.16,0.68" "s20_62 CKM_SHARES=0.20,0.62" "s12_66 CKM_SHARES=0.12,0.66"; fi
for cfg in "$@"; do run $cfg; done
