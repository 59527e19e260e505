#!/bin/bash
# Round-3 profile collection (run on the GPU box from the repo root):  tools/collect_r03.sh <tag> [what...]
#   what: stats  = rocprofv3 --kernel-trace --stats of a cfg3 run (192 bins)
#         pmc3   = three --pmc passes (FETCH_SIZE | WRITE_SIZE | SQ_*) of ONE cfg3 step over a 48-bin sample (no warm pass)
#         pmc2   = the same three passes of ONE cfg2 search
#         trace2 = kernel trace of cfg2 with two steps in flight (for tools/timeline.py)
set -u
TAG=${1:-r03x}; shift
WHAT=${*:-stats pmc3}
ROOT=$PWD
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
C3="--config cfg3 --no-cpu-baseline --no-cfg2 --no-emulation --workdir /tmp/ckm_r03_work"     # (one synthetic world for all runs of this call)
for w in $WHAT; do
  case $w in
    stats)
      (cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats3" -o cfg3 -- python "$ROOT/bench.py" $C3 --bins-total 192 --steps 1 --warmup 1 > "$OUT/stats3.json" 2> "$OUT/stats3.err") ;;
    pmc3)
      for pass in "fetch FETCH_SIZE" "write WRITE_SIZE" "sq SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY"; do
        set -- $pass; name=$1; shift
        (cd /tmp && CKM_BENCH_SKIP_WARM=1 CKM_WS_PER_MP=5 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d "$OUT/pmc3_$name" -o p -- python "$ROOT/bench.py" $C3 --bins-total 48 --steps 1 --warmup 0 > "$OUT/pmc3_$name.json" 2> "$OUT/pmc3_$name.err")
      done ;;
    pmc2)
      for pass in "fetch FETCH_SIZE" "write WRITE_SIZE" "sq SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY"; do
        set -- $pass; name=$1; shift
        (cd /tmp && CKM_WS_PER_MP=5 CKM_BENCH_STEADY=0 CKM_BENCH_FROM_HOST=0 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d "$OUT/pmc_$name" -o p -- python "$ROOT/bench.py" --config cfg2 --steps 1 --warmup 0 --no-cpu-baseline > "$OUT/pmc_$name.json" 2> "$OUT/pmc_$name.err")
      done ;;
    trace2)
      (cd /tmp && CKM_BENCH_FROM_HOST=0 rocprofv3 --kernel-trace --output-format csv -d "$OUT/trace2" -o bench -- python "$ROOT/bench.py" --config cfg2 --steps 3 --warmup 2 --no-cpu-baseline > "$OUT/trace2.json" 2> "$OUT/trace2.err") ;;
  esac
done
ls -R "$OUT" | head -60
