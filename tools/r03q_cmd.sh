set -u
export TMPDIR=/tmp
O=gpurun_out/r03q; mkdir -p $O
CKM_TRACE=1 python bench.py --config cfg5 > $O/bench_cfg5.json 2> $O/bench_cfg5.err
grep -E "table |did not fit" $O/bench_cfg5.err | head -60 > $O/cfg5_fit.txt
python -m pytest tests/test_gpu_lineage.py tests/test_gpu_cascade.py -m gpu -x -q 2>&1 | tail -3 > $O/pytest_tail.txt
python - <<'P'
import json
d=json.loads(open("gpurun_out/r03q/bench_cfg5.json").read().strip().splitlines()[-1])
print(d["ms_per_step"], d["cascade_fallback_lanes"], d["workspace"], d["find_parts_s"])
P
cat $O/cfg5_fit.txt | head -30; cat $O/pytest_tail.txt
