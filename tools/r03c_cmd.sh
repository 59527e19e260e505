set -u
export TMPDIR=/tmp
O=gpurun_out/r03c; mkdir -p $O
python -m pytest tests -m gpu -q 2>&1 | tail -25 > $O/pytest_gpu.txt
W=/tmp/ckm_r03c
python bench.py --steps 2 --warmup 2 --workdir $W --no-cfg2 --no-cpu-baseline > $O/bench_lanes3.json 2> $O/bench_lanes3.err
CKM_FIND_PIPELINE=2 python bench.py --steps 2 --warmup 2 --workdir $W --no-cfg2 --no-cpu-baseline --no-emulation > $O/bench_lanes2.json 2> $O/bench_lanes2.err
CKM_FIND_PIPELINE=4 python bench.py --steps 2 --warmup 2 --workdir $W --no-cfg2 --no-cpu-baseline --no-emulation > $O/bench_lanes4.json 2> $O/bench_lanes4.err
tail -5 $O/pytest_gpu.txt; for f in lanes3 lanes2 lanes4; do tail -c 300 $O/bench_$f.err; head -c 260 $O/bench_$f.json; echo; done
