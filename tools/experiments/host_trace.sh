#!/bin/bash
# Host-side timestamps of the searches of a short cfg2 run.  usage (GPU box, repo root): bash tools/experiments/host_trace.sh
export TMPDIR=/tmp CKM_BENCH_STEADY=0 CKM_BENCH_FROM_HOST=0
CKM_TRACE=1 timeout 200 python bench.py --steps 2 --warmup 2 --no-cpu-baseline --lineage-bins 0 2>&1 >/dev/null | grep ckm-trace | tail -18
