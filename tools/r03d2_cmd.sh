set -u
export TMPDIR=/tmp
O=gpurun_out/r03d2; mkdir -p $O
C3="--config cfg3 --no-cpu-baseline --no-cfg2 --no-emulation --bins-total 128 --steps 1 --warmup 1"
CKM_WS_VMM=1 timeout 45 python bench.py $C3 > $O/vmm1.json 2> $O/vmm1.err; echo "rc=$?"
timeout 45 python bench.py $C3 > $O/vmm0.json 2> $O/vmm0.err; echo "rc=$?"
python - <<'P'
import json
for n in ("vmm1","vmm0"):
    try:
        d=json.loads(open("gpurun_out/r03d2/%s.json"%n).read().strip().splitlines()[-1])
        print(n, "first_pass_s %.2f"%d["first_pass_s"], "step %.3f"%(d["ms_per_step"]/1e3), d["workspace_rank0"], d["cascade_fallback_lanes_rank0"], d["stage_pairs"]["envelopes"])
    except Exception as e:
        print(n, "failed", e)
P
tail -3 $O/vmm1.err
