set -u
export TMPDIR=/tmp
ROOT=$PWD
O=$ROOT/gpurun_out/r03y; mkdir -p $O
python -m pytest tests -m gpu -x -q 2>&1 | tail -5 > $O/pytest_tail.txt; cat $O/pytest_tail.txt
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.txt 2>&1; tail -2 $O/smoke.txt
python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_shaped.json 2> $O/bench_driver_shaped.err
python - <<'P'
import json
d=json.loads(open("gpurun_out/r03y/bench_driver_shaped.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["steps"], d["parts_s_rank0"], d["device_state_timed_region"])
print(d["cfg2"]["ms_per_step"], d["cfg2"]["steady_state"]["ms_per_step"], d["cfg2"]["stages_ms"], d["cfg2"]["device_state_timed_region"])
print(d["emulated_rank0_of_8"]["wall_s"], d["first_pass_s"], d["first_pass_overhead_s"], d["cpu_baseline"])
print(d["step_utilisation"])
P
C3="--config cfg3 --no-cpu-baseline --no-cfg2 --no-emulation"
(cd /tmp && rocprofv3 --kernel-trace --output-format csv -d /tmp/tr1000 -o cfg3 -- python $ROOT/bench.py $C3 --steps 1 --warmup 1 > $O/bench_traced.json 2> $O/bench_traced.err)
f=$(find /tmp/tr1000 -name '*kernel_trace.csv' | head -1)
python tools/occupancy_timeline.py $f 50 last-step > $O/timeline_cfg3_1000bins.txt 2>&1
head -3 $O/timeline_cfg3_1000bins.txt | cut -c1-300
python bench.py --config cfg5 > $O/bench_cfg5.json 2> $O/bench_cfg5.err
CKM_BENCH_DEVICE=0 CKM_BENCH_DIST_BACKEND=gloo timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29656 bench.py --gpus 2 --steps 1 --warmup 2 --bins-total 128 > $O/bench_n2_gloo.json 2> $O/bench_n2_gloo.err
python - <<'P'
import json
d=json.loads(open("gpurun_out/r03y/bench_cfg5.json").read().strip().splitlines()[-1]); print("cfg5", d["ms_per_step"], d["cascade_fallback_lanes"])
d=json.loads(open("gpurun_out/r03y/bench_n2_gloo.json").read().strip().splitlines()[-1]); print("n2", d["ms_per_step"], d["n_gpus"])
P
