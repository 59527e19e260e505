set -u
export TMPDIR=/tmp
O=gpurun_out/r03i; mkdir -p $O
python -m pytest tests -m gpu -q -x 2>&1 | tail -30 > $O/pytest_gpu.txt
tail -8 $O/pytest_gpu.txt
python bench.py --steps 2 --warmup 2 --no-cpu-baseline > $O/bench_default.json 2> $O/bench_default.err
tail -c 300 $O/bench_default.err; head -c 300 $O/bench_default.json; echo
(cd /tmp && CKM_BENCH_SKIP_WARM=1 CKM_WS_PER_MP=5 rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVES SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$O/pmc3_sq -o p -- python $GRAFT_REPO_ROOT/bench.py --config cfg3 --no-cpu-baseline --no-cfg2 --no-emulation --bins-total 48 --steps 1 --warmup 0 > $GRAFT_REPO_ROOT/$O/pmc3_sq.json 2> $GRAFT_REPO_ROOT/$O/pmc3_sq.err)
python tools/pmc_summary.py $O/pmc3_sq | awk '{print $1, $2, $3, $5}' | head -8
