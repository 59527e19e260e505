set -u
export TMPDIR=/tmp
ROOT=$PWD
O=$ROOT/gpurun_out/r03s; mkdir -p $O
python bench.py --steps 2 --warmup 1 --no-cpu-baseline > $O/bench_default.json 2> $O/bench_default.err
python - <<'P'
import json
d=json.loads(open("gpurun_out/r03s/bench_default.json").read().strip().splitlines()[-1])
print(d["ms_per_step"], d["parts_s_rank0"], d["gpu_host_split_s_rank0"]["ssv_kernels"], d["cfg2"]["ms_per_step"], d["cfg2"]["steady_state"]["ms_per_step"], d["emulated_rank0_of_8"]["wall_s"])
P
C3="--config cfg3 --no-cpu-baseline --no-cfg2 --no-emulation"
(cd /tmp && rocprofv3 --kernel-trace --output-format csv -d /tmp/tr1000 -o cfg3 -- python $ROOT/bench.py $C3 --steps 1 --warmup 1 > $O/bench_traced.json 2> $O/bench_traced.err)
f=$(find /tmp/tr1000 -name '*kernel_trace.csv' | head -1)
python tools/occupancy_timeline.py $f 50 last-step > $O/timeline_cfg3_1000bins.txt 2>&1
head -8 $O/timeline_cfg3_1000bins.txt | cut -c1-400; tail -4 $O/timeline_cfg3_1000bins.txt
