#!/bin/bash
# Kernel timeline of one cfg2 search.  usage (GPU box, repo root): bash tools/experiments/trace.sh [outdir]
set -u
ROOT=$PWD
OUT=$ROOT/${1:-gpurun_out/trace}; mkdir -p "$OUT"
export TMPDIR=/tmp CKM_BENCH_STEADY=0 CKM_BENCH_FROM_HOST=0
(cd /tmp && timeout 300 rocprofv3 --kernel-trace -d "$OUT/trace" -o bench -- python "$ROOT/bench.py" --steps 2 --warmup 2 --no-cpu-baseline --lineage-bins 0 > "$OUT/trace.log" 2>&1)
DB=$(find "$OUT/trace" -name "*.db" | head -1)
python "$ROOT/tools/timeline2.py" "$DB" 20 -1 > "$OUT/timeline.txt" 2>&1
tail -22 "$OUT/timeline.txt"
rm -rf "$OUT/trace"
