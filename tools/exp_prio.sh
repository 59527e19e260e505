set -u
export TMPDIR=/tmp
mkdir -p gpurun_out/r02c
timeout 900 python -m pytest tests -x -q -m gpu 2>&1 | tail -3
timeout 400 python bench.py --steps 5 --warmup 2 > gpurun_out/r02c/bench_default.json 2> gpurun_out/r02c/bench_default.err
python -c "
import json; d=json.loads(open('gpurun_out/r02c/bench_default.json').read().strip().split('\n')[-1]); print(d['ms_per_step'], d['value'], d['steady_state']['ms_per_step'], d['lineage_wf_equiv'].get('bins_per_hour'), d['lineage_wf_equiv'].get('parts_s_rank0'), d['lineage_wf_equiv'].get('first_pass_s'))"
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
