set -u
export TMPDIR=/tmp
O=gpurun_out/r03f; mkdir -p $O
# the N = 2 launch the driver uses, on the one device of this box (test hooks: both ranks on device 0, collectives over gloo)
CKM_BENCH_DEVICE=0 CKM_BENCH_DIST_BACKEND=gloo timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29655 bench.py --gpus 2 --steps 1 --warmup 2 --bins-total 128 > $O/bench_n2_gloo.json 2> $O/bench_n2_gloo.err
tail -c 400 $O/bench_n2_gloo.err; head -c 400 $O/bench_n2_gloo.json; echo
bash tools/collect_r03.sh r03f trace2 pmc2 pmc3 > $O/collect.log 2>&1
python tools/occupancy_timeline.py $O/trace2/bench_kernel_trace.csv 2 > $O/timeline.txt 2>&1 || true
ls $O/trace2 | head; tail -12 $O/timeline.txt
