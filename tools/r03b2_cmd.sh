set -u
export TMPDIR=/tmp
O=gpurun_out/r03b2; mkdir -p $O
for B in 125000000 62500000; do
  CKM_FIND_PAIR_BUDGET=$B python bench.py --config cfg3 --no-cpu-baseline --no-cfg2 --no-emulation --steps 1 --warmup 2 > $O/bench_$B.json 2> $O/bench_$B.err
  python - $B <<'P'
import json,sys
d=json.loads(open("gpurun_out/r03b2/bench_%s.json"%sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[1], d["ms_per_step"], d["parts_s_rank0"], "first", d["first_pass_s"], d["second_pass_s_same_bins"], "ws", d["workspace_rank0"], "searches", d["searches_rank0"], d["cascade_fallback_lanes_rank0"])
P
done
