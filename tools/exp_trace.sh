set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/r2y; mkdir -p $OUT
export TMPDIR=/tmp CKM_BENCH_STEADY=0 CKM_BENCH_FROM_HOST=0
(cd /tmp && timeout 300 rocprofv3 --kernel-trace -d $OUT/trace -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 2 --no-cpu-baseline --lineage-bins 0 > $OUT/trace.log 2>&1)
DB=$(find $OUT/trace -name "*.db" | head -1); echo db=$DB
python $GRAFT_REPO_ROOT/tools/timeline2.py $DB 20 -1 > $OUT/timeline.txt 2>&1
tail -22 $OUT/timeline.txt
rm -rf $OUT/trace
