set -u
export TMPDIR=/tmp
O=gpurun_out/r03b; mkdir -p $O
python -m pytest tests -m gpu -q 2>&1 | tail -15 > $O/pytest_gpu.txt
python bench.py --steps 2 --warmup 2 --host-profile $O/host_profile.txt > $O/bench_default.json 2> $O/bench_default.err
(cd /tmp && CKM_WS_PER_MP=5 CKM_BENCH_STEADY=0 CKM_BENCH_FROM_HOST=0 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAVES SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$O/pmc_sq -o p -- python $GRAFT_REPO_ROOT/bench.py --config cfg2 --steps 1 --warmup 0 --no-cpu-baseline > $GRAFT_REPO_ROOT/$O/pmc_sq.json 2> $GRAFT_REPO_ROOT/$O/pmc_sq.err)
tail -3 $O/pytest_gpu.txt; tail -c 400 $O/bench_default.err; head -c 300 $O/bench_default.json
