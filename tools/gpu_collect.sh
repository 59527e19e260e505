#!/bin/bash
# ONE collection script for the GPU box (run from the repo root, e.g. through gpurun):
#     tools/gpu_collect.sh <tag> <what> [<what> ...]
# writes everything under gpurun_out/<tag>/ ; tools/make_profile_files.py <tag> reduces it to the files kept under profiles/.
#   tests            python -m pytest tests -m gpu -x -q                                   -> pytest_tail.txt
#   smoke            __graft_entry__.smoke()                                               -> smoke.txt
#   bench[:ARGS]     python bench.py ARGS (comma-separated, e.g. bench:--steps,2,--warmup,1) -> bench<k>.json / .err   (k counts the bench items of the call)
#   env:K=V          export K=V for the items that follow (env:K= unsets it)
#   stats3[:BINS]    rocprofv3 --kernel-trace --stats of a cfg3 run (default 192 bins: warm pass + one timed step)      -> stats3/
#   trace3[:BINS]    rocprofv3 --kernel-trace of cfg3 steps for tools/occupancy_timeline.py (default 1000 bins)         -> trace3/
#   pmc3[:BINS]      three separate --pmc passes (FETCH_SIZE | WRITE_SIZE | SQ_*) of ONE cfg3 step (default 48 bins)    -> pmc3_{fetch,write,sq}/
#   pmc2             the same three passes of ONE cfg2 search                                                           -> pmc_{fetch,write,sq}/
#   valu             tools/ubench/valu_rates (VALU issue rates of the SSV row body)                                     -> valu_rates.txt
#   genes_prof[:BINS] one ckm_genes_call over BINS (default 48) synthetic 2 Mb bins (tools/gene_profile.py): plain with CKM_TRACE=1, rocprofv3 --kernel-trace --stats,
#                    and a separate --pmc pass (SQ counters)                                                             -> genes_prof/{plain.txt,stats/,pmc/}, genes_prof.txt
#   genes[:THREADS]  the gene-calling tests + `bench.py --config genes` with CKM_TRACE=1 (phase times of a call on stderr)     -> pytest_genes.txt, genes.json / .err
# Counter passes never combine --pmc with anything but --kernel-trace (the pool's gpurun refuses other combinations).
set -u
TAG=${1:?tag}; shift
ROOT=$PWD
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
C3="--config cfg3 --no-cpu-baseline --no-cfg2 --no-emulation --no-verify --no-genes --hard-bins 0 --workdir /tmp/ckm_work"      # (one synthetic world for all runs of a call)
PMC_PASSES=("fetch FETCH_SIZE" "write WRITE_SIZE" "sq SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY")
nbench=0
for item in "$@"; do
  what=${item%%:*}; arg=""; [ "$item" != "$what" ] && arg=${item#*:}
  case $what in
    env)    k=${arg%%=*}; v=${arg#*=}; if [ -z "$v" ]; then unset "$k"; else export "$k=$v"; fi ;;
    tests)  python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > "$OUT/pytest_tail.txt"; cat "$OUT/pytest_tail.txt" ;;
    smoke)  python -c "import __graft_entry__ as g; g.smoke()" > "$OUT/smoke.txt" 2>&1; tail -2 "$OUT/smoke.txt" ;;
    bench)  nbench=$((nbench + 1)); python bench.py ${arg//,/ } > "$OUT/bench$nbench.json" 2> "$OUT/bench$nbench.err"; tail -c 600 "$OUT/bench$nbench.json" ;;
    stats3) (cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats3" -o cfg3 -- python "$ROOT/bench.py" $C3 --bins-total "${arg:-192}" --steps 1 --warmup 1 > "$OUT/stats3.json" 2> "$OUT/stats3.err") ;;
    trace3) (cd /tmp && rocprofv3 --kernel-trace --output-format csv -d "$OUT/trace3" -o cfg3 -- python "$ROOT/bench.py" $C3 --bins-total "${arg:-1000}" --steps 2 --warmup 1 > "$OUT/trace3.json" 2> "$OUT/trace3.err")
            python tools/occupancy_timeline.py "$OUT"/trace3/*_kernel_trace.csv 50 last-step > "$OUT/timeline3.txt" 2>&1; head -12 "$OUT/timeline3.txt" ;;
    pmc3)   for pass in "${PMC_PASSES[@]}"; do set -- $pass; name=$1; shift
              (cd /tmp && CKM_BENCH_SKIP_WARM=1 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d "$OUT/pmc3_$name" -o p -- python "$ROOT/bench.py" $C3 --bins-total "${arg:-48}" --steps 1 --warmup 0 > "$OUT/pmc3_$name.json" 2> "$OUT/pmc3_$name.err")
            done ;;
    pmc2)   for pass in "${PMC_PASSES[@]}"; do set -- $pass; name=$1; shift
              (cd /tmp && CKM_BENCH_STEADY=0 CKM_BENCH_FROM_HOST=0 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d "$OUT/pmc_$name" -o p -- python "$ROOT/bench.py" --config cfg2 --steps 1 --warmup 0 --no-cpu-baseline --no-verify > "$OUT/pmc_$name.json" 2> "$OUT/pmc_$name.err")
            done ;;
    genes)  python -m pytest tests/test_gpu_genes.py tests/test_gpu_orf.py -m gpu -x -q 2>&1 | tail -5 > "$OUT/pytest_genes.txt"; cat "$OUT/pytest_genes.txt"
            ( [ -n "$arg" ] && export CKM_GENE_THREADS=$arg; CKM_TRACE=1 python bench.py --config genes > "$OUT/genes.json" 2> "$OUT/genes.err" ); tail -c 700 "$OUT/genes.json" ;;
    genes_prof) G="$OUT/genes_prof"; mkdir -p "$G"
            CKM_TRACE=1 python tools/gene_profile.py "${arg:-48}" 11 2 > "$G/plain.txt" 2>&1
            (cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d "$G/stats" -o g -- python "$ROOT/tools/gene_profile.py" "${arg:-48}" 11 2 > "$G/stats.txt" 2>&1)
            (cd /tmp && rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY --kernel-trace --output-format csv -d "$G/pmc" -o g -- python "$ROOT/tools/gene_profile.py" "${arg:-48}" 11 1 > "$G/pmc.txt" 2>&1)
            rm -f "$G"/stats/*kernel_trace.csv "$G"/pmc/*kernel_trace.csv
            python tools/gene_profile_summary.py "$G" > "$OUT/genes_prof.txt" 2>&1; head -60 "$OUT/genes_prof.txt" ;;
    valu)   (cd tools/ubench && hipcc --offload-arch=gfx950 -O3 -o valu_rates valu_rates.hip 2>/dev/null; ./valu_rates) > "$OUT/valu_rates.txt" 2>&1; tail -20 "$OUT/valu_rates.txt" ;;
    *)      echo "unknown item: $item" ;;
  esac
done
du -sh "$OUT" | tail -1
