#!/bin/bash
# One cfg2 bench line per knob variant.  usage (GPU box, repo root): bash tools/experiments/knobs.sh [outdir]
set -u
OUT=${1:-gpurun_out/knobs}; mkdir -p "$OUT"
export TMPDIR=/tmp CKM_BENCH_STEADY=0 CKM_BENCH_FROM_HOST=0
B="python bench.py --steps 6 --warmup 2 --no-cpu-baseline --lineage-bins 0"
run() { name=$1; shift; env "$@" timeout 200 $B > "$OUT/$name.json" 2> "$OUT/$name.err"; python - <<PY
import json
try:
    d = json.load(open("$OUT/$name.json")); s = d["stages_ms"]
    print("$name", round(d["ms_per_step"], 2), "ssv %.1f filters %.1f dom %.1f host %.1f" % (s["ssv"], s["filters"], s["domains"], s["host"]), d["rows"])
except Exception as e:
    print("$name failed", e)
PY
}
run base A=1
run chain_prio CKM_CHAIN_PRIO=1
run stream_prio CKM_STREAM_PRIO=1
run fused CKM_FUSED=1
run fused_tail6 CKM_FUSED_TAIL=6
run parts3 CKM_LONG_SHARE=0.25,0.5
run ens_joined CKM_ENS_JOINED=1
run env_separate CKM_ENV_INPLACE=0
run host_cascade_w3 CKM_CASCADE=host CKM_WORKERS=3
run threads16 CKM_HOST_THREADS=16
run base2 A=1
