set -u
export TMPDIR=/tmp
O=gpurun_out/r03d; mkdir -p $O
python -m pytest tests -m gpu -q -x 2>&1 | tail -30 > $O/pytest_gpu.txt
tail -6 $O/pytest_gpu.txt
python bench.py --config cfg5 > $O/bench_cfg5.json 2> $O/bench_cfg5.err
tail -c 300 $O/bench_cfg5.err; head -c 600 $O/bench_cfg5.json; echo
python bench.py --steps 2 --warmup 2 > $O/bench_default.json 2> $O/bench_default.err
tail -c 300 $O/bench_default.err; head -c 300 $O/bench_default.json; echo
