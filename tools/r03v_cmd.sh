set -u
export TMPDIR=/tmp
O=gpurun_out/r03v; mkdir -p $O
python bench.py --steps 2 --warmup 1 --no-cpu-baseline > $O/bench_default.json 2> $O/bench_default.err
python - <<'P'
import json
d=json.loads(open("gpurun_out/r03v/bench_default.json").read().strip().splitlines()[-1])
print(d["ms_per_step"], d["parts_s_rank0"], d["gpu_host_split_s_rank0"]["ssv_kernels"], d["cfg2"]["ms_per_step"], d["cfg2"]["steady_state"]["ms_per_step"], d["emulated_rank0_of_8"]["wall_s"], d["first_pass_s"], d["first_pass_overhead_s"])
P
