set -u
export TMPDIR=/tmp
O=gpurun_out/r03k; mkdir -p $O
W=/tmp/ckm_r03k
for q in 16 32 64; do
  GPU_MAX_HW_QUEUES=$q python bench.py --steps 2 --warmup 2 --workdir $W --no-cfg2 --no-cpu-baseline --no-emulation > $O/bench_q$q.json 2> $O/bench_q$q.err
  python - <<P
import json
d=json.loads(open("$O/bench_q$q.json").read().strip().split("\n")[-1])
print("GPU_MAX_HW_QUEUES=$q", round(d["ms_per_step"]), d["parts_s_rank0"])
P
done
GPU_MAX_HW_QUEUES=32 python bench.py --config cfg2 --steps 5 --warmup 2 --no-cpu-baseline > $O/bench_cfg2_q32.json 2>/dev/null
python - <<P
import json
d=json.loads(open("$O/bench_cfg2_q32.json").read().strip().split("\n")[-1])
print("cfg2 q32", d["ms_per_step"], d["steady_state"]["ms_per_step"])
P
