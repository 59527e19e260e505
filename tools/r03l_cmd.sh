set -u
export TMPDIR=/tmp
O=gpurun_out/r03l; mkdir -p $O
W=/tmp/ckm_r03l
run() { name=$1; shift; env "$@" python bench.py --steps 2 --warmup 2 --workdir $W --no-cfg2 --no-cpu-baseline --no-emulation > $O/bench_$name.json 2> $O/bench_$name.err
  python - <<P
import json
d=json.loads(open("$O/bench_$name.json").read().strip().split("\n")[-1])
print("$name", round(d["ms_per_step"]), {k: round(v,2) for k,v in d["parts_s_rank0"].items()}, d["searches_rank0"], d["workspace_rank0"]["high_water_bytes_max"]>>30)
P
}
run base A=1
run pairs500M CKM_FIND_PAIR_BUDGET=500000000
run pairs125M CKM_FIND_PAIR_BUDGET=125000000
run queues8 GPU_MAX_HW_QUEUES=8
run lanes3 CKM_FIND_PIPELINE=3
