set -u
export TMPDIR=/tmp
O=gpurun_out/r03x; mkdir -p $O
python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-emulation > $O/bench.json 2> $O/bench.err
grep -E "ckm-trace" $O/bench.err | tail -400 > $O/trace_tail.txt
python - <<'P'
import json
d=json.loads(open("gpurun_out/r03x/bench.json").read().strip().splitlines()[-1])
print(d["device_state_timed_region"], d["cfg2"]["device_state_timed_region"]); print(d["ms_per_step"], d["cfg2"]["ms_per_step"], d["cfg2"]["steady_state"]["ms_per_step"], d["cfg2"]["stages_ms"])
P
