set -u
export TMPDIR=/tmp
O=gpurun_out/r03u; mkdir -p $O
python -m pytest tests -m gpu -x -q 2>&1 | tail -5 > $O/pytest_tail.txt; cat $O/pytest_tail.txt
python bench.py --config cfg2 --no-cpu-baseline > $O/bench_cfg2.json 2> $O/bench_cfg2.err
python -c "
import json; d=json.loads(open('$O/bench_cfg2.json').read().strip().splitlines()[-1]); print(d['ms_per_step'], d['steady_state']['ms_per_step'])"
