set -u
run() { name=$1; shift; env "$@" python bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('$name', round(d['ms_per_step'],2), {k:round(v,1) for k,v in d['stages_ms'].items()})"; }
run s8 A=1
run s4 CKM_SIDE_STREAMS=4
run s3 CKM_SIDE_STREAMS=3
run s2 CKM_SIDE_STREAMS=2
run s1 CKM_SIDE_STREAMS=1
run s3q12 CKM_SIDE_STREAMS=3 GPU_MAX_HW_QUEUES=12
run s2q8 CKM_SIDE_STREAMS=2 GPU_MAX_HW_QUEUES=8
run s3q24 CKM_SIDE_STREAMS=3 GPU_MAX_HW_QUEUES=24
