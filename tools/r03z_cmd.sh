set -u
export TMPDIR=/tmp
O=gpurun_out/r03z; mkdir -p $O
python -m pytest tests/test_gpu_api.py tests/test_gpu_lineage.py tests/test_gpu_cascade.py -m gpu -x -q 2>&1 | tail -3 > $O/pytest_tail.txt; cat $O/pytest_tail.txt
python bench.py --steps 2 --warmup 2 --no-cpu-baseline --no-emulation > $O/bench.json 2> $O/bench.err
python - <<'P'
import json
d=json.loads(open("gpurun_out/r03z/bench.json").read().strip().splitlines()[-1])
print(d["ms_per_step"], d["parts_s_rank0"], d["first_pass_s"], d["second_pass_s_same_bins"], d["first_pass_overhead_s"], d["cfg2"]["ms_per_step"], d["cfg2"]["steady_state"]["ms_per_step"], d["cfg2"]["value_from_host"])
P
