set -u
export TMPDIR=/tmp
O=gpurun_out/r03p; mkdir -p $O
bash tools/collect_r03.sh r03p pmc3 pmc2 stats > $O/collect.log 2>&1
python tools/occupancy_timeline.py $O/stats3/cfg3_kernel_trace.csv 50 > $O/timeline_cfg3.txt 2>&1 || true
python bench.py --config cfg5 > $O/bench_cfg5.json 2> $O/bench_cfg5.err
CKM_BENCH_DEVICE=0 CKM_BENCH_DIST_BACKEND=gloo timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29656 bench.py --gpus 2 --steps 1 --warmup 2 --bins-total 128 > $O/bench_n2_gloo.json 2> $O/bench_n2_gloo.err
head -c 300 $O/bench_cfg5.json; echo; head -c 200 $O/bench_n2_gloo.json; echo
