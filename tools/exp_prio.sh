set -u
OUT=gpurun_out/r3a; mkdir -p $OUT
export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_gpu_cascade.py -x -q 2>&1 | tail -2
export CKM_BENCH_STEADY=0 CKM_BENCH_FROM_HOST=0
B="python bench.py --steps 6 --warmup 2 --no-cpu-baseline --lineage-bins 0"
run() { name=$1; shift; env "$@" timeout 200 $B > $OUT/$name.json 2> $OUT/$name.err; python - <<PY
import json
try:
    d=json.load(open("$OUT/$name.json")); s=d["stages_ms"]; print("$name", round(d["ms_per_step"],2), "ssv %.1f filters %.1f dom %.1f host %.1f" % (s["ssv"], s["filters"], s["domains"], s["host"]), d["rows"])
except Exception as e: print("$name failed", e)
PY
}
run p2_25 CKM_LONG_SHARE=0.25
run p3_25_50 CKM_LONG_SHARE=0.25,0.5
run p3_25_40 CKM_LONG_SHARE=0.25,0.4
run p3_25_60 CKM_LONG_SHARE=0.25,0.6
run p3_15_55 CKM_LONG_SHARE=0.15,0.55
run p3_35_45 CKM_LONG_SHARE=0.35,0.45
run p4 CKM_LONG_SHARE=0.2,0.35,0.3
run p4b CKM_LONG_SHARE=0.25,0.4,0.25
