set -u
export TMPDIR=/tmp
O=gpurun_out/r03a; mkdir -p $O
tools/ubench/valu_rates > $O/valu_rates.txt 2>&1
python -m pytest tests -m gpu -q 2>&1 | tail -15 > $O/pytest_gpu.txt
python bench.py --steps 2 --warmup 2 --host-profile $O/host_profile.txt > $O/bench_default.json 2> $O/bench_default.err
CKM_SSV=i16 python bench.py --config cfg2 --steps 5 --warmup 2 --no-cpu-baseline > $O/bench_cfg2_i16.json 2> $O/bench_cfg2_i16.err
tail -3 $O/pytest_gpu.txt; tail -c 600 $O/bench_default.err; head -c 400 $O/bench_default.json
