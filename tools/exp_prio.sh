set -u
export TMPDIR=/tmp
CKM_ENV_INPLACE=1 timeout 300 python -m pytest tests/test_gpu_scan.py tests/test_gpu_fullsize.py -x -q 2>&1 | tail -2
export CKM_BENCH_STEADY=0 CKM_BENCH_FROM_HOST=0
for v in 1 0; do CKM_ENV_INPLACE=$v timeout 100 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --lineage-bins 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('inplace $v', round(d['ms_per_step'],2), d['stages_ms'], d['rows'])"; done
