set -u
export TMPDIR=/tmp
O=gpurun_out/r03h; mkdir -p $O
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; tail -2 $O/smoke.txt
python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench_default.err
tail -c 300 $O/bench_default.err; head -c 400 $O/bench_default.json; echo
python -m pytest tests/test_gpu_api.py tests/test_gpu_lineage.py tests/test_gpu_workers.py -m gpu -q 2>&1 | tail -4
