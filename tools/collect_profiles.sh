#!/bin/bash
# Collects the artifacts kept under profiles/ (run on the GPU box from the repo root):
#   tools/collect_profiles.sh <tag>      e.g. r01c
# bench lines (default workers, and CKM_WORKERS=1), rocprofv3 kernel trace of the default configuration, and three
# --pmc passes (FETCH_SIZE | WRITE_SIZE | SQ_*) of ONE step with CKM_WORKERS=1 (counter collection serialises kernels).
set -u
TAG=${1:-r01x}
ROOT=$PWD
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
python bench.py > "$OUT/bench_default.json" 2> "$OUT/bench_default.err"
CKM_WORKERS=1 python bench.py --no-cpu-baseline > "$OUT/bench_w1.json" 2> "$OUT/bench_w1.err"
# ten times the bins (the shape of cfg3's phylo pass: 1000 bins x 43 models), to show the step scales with the input
python bench.py --bins 1000 --steps 2 --warmup 1 --no-cpu-baseline > "$OUT/bench_1000bins.json" 2> "$OUT/bench_1000bins.err"
(cd /tmp && rocprofv3 --kernel-trace -d "$OUT/trace" -o bench -- python "$ROOT/bench.py" --steps 3 --warmup 1 --no-cpu-baseline > "$OUT/trace.log" 2>&1)
for pass in "fetch FETCH_SIZE" "write WRITE_SIZE" "sq SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY"; do
  set -- $pass; name=$1; shift
  (cd /tmp && CKM_WORKERS=1 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d "$OUT/pmc_$name" -o p -- python "$ROOT/bench.py" --steps 1 --warmup 0 --no-cpu-baseline > "$OUT/pmc_$name.log" 2>&1)
done
ls -R "$OUT" | head -40
