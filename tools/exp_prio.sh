set -u
OUT=gpurun_out/r3b; mkdir -p $OUT
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_cascade.py tests/test_gpu_scan.py tests/test_gpu_fullsize.py -x -q 2>&1 | tail -2
export CKM_BENCH_STEADY=0 CKM_BENCH_FROM_HOST=0
B="python bench.py --steps 6 --warmup 2 --no-cpu-baseline --lineage-bins 0"
run() { name=$1; shift; env "$@" timeout 200 $B > $OUT/$name.json 2> $OUT/$name.err; python - <<PY
import json
try:
    d=json.load(open("$OUT/$name.json")); s=d["stages_ms"]; print("$name", round(d["ms_per_step"],2), "ssv %.1f filters %.1f dom %.1f host %.1f" % (s["ssv"], s["filters"], s["domains"], s["host"]), d["rows"])
except Exception as e: print("$name failed", e)
PY
}
run joined CKM_ENS_JOINED=1
run pergroup A=1
run joined2 CKM_ENS_JOINED=1
run pergroup2 A=1
