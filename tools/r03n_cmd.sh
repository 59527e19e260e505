set -u
export TMPDIR=/tmp
O=gpurun_out/r03n; mkdir -p $O
python -m pytest tests -m gpu -q 2>&1 | tail -6 > $O/pytest_gpu.txt; tail -3 $O/pytest_gpu.txt
bash tools/collect_r03.sh r03n pmc3 pmc2 stats trace2 > $O/collect.log 2>&1
python tools/occupancy_timeline.py $O/trace2/bench_kernel_trace.csv 2 > $O/timeline_cfg2.txt 2>&1 || true
python tools/occupancy_timeline.py $O/stats3/cfg3_kernel_trace.csv 50 > $O/timeline_cfg3.txt 2>&1 || true
head -4 $O/timeline_cfg3.txt
tools/ubench/valu_rates > $O/valu_rates.txt 2>&1
