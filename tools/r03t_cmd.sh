set -u
export TMPDIR=/tmp
ROOT=$PWD
O=$ROOT/gpurun_out/r03t; mkdir -p $O
C3="--config cfg3 --no-cpu-baseline --no-cfg2 --no-emulation"
CKM_TRACE=2 python bench.py $C3 --steps 1 --warmup 1 > $O/bench.json 2> $O/bench.err
grep -E "find-trace|ckm-trace" $O/bench.err | grep -E "find-trace|ssv turn|chain queued|chain drained|results copied|cascade done|plan ready|tables ready" | tail -400 > $O/trace_tail.txt
wc -l $O/trace_tail.txt
python - <<'P'
import json
d=json.loads(open("gpurun_out/r03t/bench.json").read().strip().splitlines()[-1])
print(d["ms_per_step"], d["parts_s_rank0"], d["gpu_host_split_s_rank0"]["ssv_kernels"])
P
python tools/lane_trace.py $O/bench.err 3000 all | grep -E "seqs:|L[01] |chain drained|chain queued" | tail -70
