set -u
bash tools/collect_profiles.sh r02b 1000 > gpurun_out/collect_r02b.log 2>&1
tail -3 gpurun_out/collect_r02b.log
export CKM_BENCH_STEADY=0 CKM_BENCH_FROM_HOST=0
for t in 8 16 32; do CKM_HOST_THREADS=$t timeout 200 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --lineage-bins 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('threads $t', round(d['ms_per_step'],2), d['stages_ms'])"; done
