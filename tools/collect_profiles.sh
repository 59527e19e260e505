#!/bin/bash
# Collects the artifacts kept under profiles/ (run on the GPU box from the repo root):
#   tools/collect_profiles.sh <tag> [cfg3 bins]      e.g. r02a 1000
# bench lines (default = device-driven cascade, one lane; CKM_CASCADE=host CKM_WORKERS=3 = the host-driven cascade of round 1),
# a 1000-bin cfg2 line, the cfg3 line, the rocprofv3 kernel trace of the default configuration, and three --pmc passes
# (FETCH_SIZE | WRITE_SIZE | SQ_*) of ONE search (counter collection serialises kernels; CKM_WS_PER_MP sizes the workspace so that
# the very first search already runs on the device-driven cascade).
set -u
TAG=${1:-r02x}
CFG3_BINS=${2:-0}
ROOT=$PWD
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
python bench.py --steps 5 --warmup 2 > "$OUT/bench_default.json" 2> "$OUT/bench_default.err"
CKM_CASCADE=host CKM_WORKERS=3 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --lineage-bins 0 > "$OUT/bench_hostcascade_w3.json" 2> "$OUT/bench_hostcascade_w3.err"
# ten times the bins (the shape of cfg3's phylo pass: 1000 bins x 43 models), to show the step scales with the input
python bench.py --bins 1000 --steps 2 --warmup 2 --no-cpu-baseline --lineage-bins 0 > "$OUT/bench_1000bins.json" 2> "$OUT/bench_1000bins.err"
(cd /tmp && CKM_BENCH_STEADY=0 CKM_BENCH_FROM_HOST=0 rocprofv3 --kernel-trace -d "$OUT/trace" -o bench -- python "$ROOT/bench.py" --steps 3 --warmup 2 --no-cpu-baseline --lineage-bins 0 > "$OUT/trace.log" 2>&1)
for pass in "fetch FETCH_SIZE" "write WRITE_SIZE" "sq SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY"; do
  set -- $pass; name=$1; shift
  (cd /tmp && CKM_WS_PER_MP=5 CKM_BENCH_STEADY=0 CKM_BENCH_FROM_HOST=0 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d "$OUT/pmc_$name" -o p -- python "$ROOT/bench.py" --steps 1 --warmup 0 --no-cpu-baseline --lineage-bins 0 > "$OUT/pmc_$name.log" 2>&1)
done
if [ "$CFG3_BINS" -gt 0 ]; then
  python bench.py --config cfg3 --bins-total "$CFG3_BINS" --steps 1 --warmup 1 > "$OUT/bench_cfg3.json" 2> "$OUT/bench_cfg3.err"
fi
ls "$OUT" | head -40
