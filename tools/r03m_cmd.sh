set -u
export TMPDIR=/tmp
O=gpurun_out/r03m; mkdir -p $O
W=/tmp/ckm_r03m
run() { name=$1; shift; env "$@" python bench.py --steps 2 --warmup 2 --workdir $W --no-cfg2 --no-cpu-baseline --no-emulation > $O/bench_$name.json 2> $O/bench_$name.err
  python - <<P
import json
d=json.loads(open("$O/bench_$name.json").read().strip().split("\n")[-1])
print("$name", round(d["ms_per_step"]), {k: round(v,2) for k,v in d["parts_s_rank0"].items()})
P
}
run t16_32 A=1
run t40_80 CKM_SSV_THREADS=40,80
run t8_16 CKM_SSV_THREADS=8,16
run t0_24 CKM_SSV_THREADS=0,24
for t in "16,32" "40,80"; do CKM_SSV_THREADS=$t python bench.py --config cfg2 --steps 5 --warmup 2 --no-cpu-baseline > $O/cfg2_$t.json 2>/dev/null; python - <<P
import json
d=json.loads(open("$O/cfg2_$t.json").read().strip().split("\n")[-1])
print("cfg2 threads $t", d["ms_per_step"], d["steady_state"]["ms_per_step"])
P
done
