set -u
export TMPDIR=/tmp
O=gpurun_out/r03final; mkdir -p $O
python -m pytest tests -m gpu -x -q 2>&1 | tail -5 > $O/pytest_tail.txt; cat $O/pytest_tail.txt
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.txt 2>&1; tail -2 $O/smoke.txt
python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_shaped.json 2> $O/bench_driver_shaped.err
python - <<'P'
import json
d=json.loads(open("gpurun_out/r03final/bench_driver_shaped.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["steps"], d["parts_s_rank0"], d["device_state_timed_region"])
print(d["cfg2"]["ms_per_step"], d["cfg2"]["steady_state"]["ms_per_step"], d["emulated_rank0_of_8"]["wall_s"], d["first_pass_s"], d["first_pass_overhead_s"])
print(d["step_utilisation"]["cycles_per_inst_per_simd"], d["step_utilisation"]["valu_frac_of_measured_rate"], d["roofline"]["frac"], d["cpu_baseline"]["value"])
P
