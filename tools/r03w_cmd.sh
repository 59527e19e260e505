set -u
export TMPDIR=/tmp
O=gpurun_out/r03w; mkdir -p $O
run2() { # name, env...
  name=$1; shift
  env "$@" python bench.py --config cfg2 --steps 5 --warmup 2 --no-cpu-baseline > $O/cfg2_$name.json 2> $O/cfg2_$name.err
  python - "$name" <<'P'
import json,sys
d=json.loads(open("gpurun_out/r03w/cfg2_%s.json"%sys.argv[1]).read().strip().splitlines()[-1])
print("cfg2 %-12s alone %.1f steady %.1f  stages %s"%(sys.argv[1], d["ms_per_step"], d["steady_state"]["ms_per_step"], {k:round(v,1) for k,v in d["stages_ms"].items()}))
P
}
run2 default X=1
run2 lateprio0 CKM_LATE_PRIO=0
run2 baton0 CKM_SSV_BATON=0
run2 bothoff CKM_LATE_PRIO=0 CKM_SSV_BATON=0
run2 latenormal CKM_LATE_NORMAL=1
C3="--config cfg3 --no-cpu-baseline --no-cfg2 --no-emulation"
CKM_LATE_NORMAL=1 python bench.py $C3 --steps 1 --warmup 1 > $O/cfg3_latenormal.json 2> $O/cfg3_latenormal.err
python -c "
import json; d=json.loads(open('$O/cfg3_latenormal.json').read().strip().splitlines()[-1]); print('cfg3 latenormal', d['ms_per_step'], d['parts_s_rank0'])"
