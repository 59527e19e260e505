set -u
export TMPDIR=/tmp
ROOT=$PWD
O=$ROOT/gpurun_out/r03r; mkdir -p $O
C3="--config cfg3 --no-cpu-baseline --no-cfg2 --no-emulation"
(cd /tmp && rocprofv3 --kernel-trace --output-format csv -d /tmp/tr1000 -o cfg3 -- python $ROOT/bench.py $C3 --steps 1 --warmup 1 > $O/bench_traced.json 2> $O/bench_traced.err)
f=$(find /tmp/tr1000 -name '*kernel_trace.csv' | head -1)
ls -la $f
python tools/occupancy_timeline.py $f 50 last-step > $O/timeline_cfg3_1000bins.txt 2>&1
head -8 $O/timeline_cfg3_1000bins.txt | cut -c1-600
python -m pytest tests/test_gpu_cascade.py -m gpu -x -q 2>&1 | tail -3 > $O/pytest_tail.txt; cat $O/pytest_tail.txt
