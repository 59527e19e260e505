set -u
export TMPDIR=/tmp
O=gpurun_out/r03c2; mkdir -p $O
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 -o /tmp/vmm_probe tools/ubench/vmm_probe.hip 2> $O/vmm_build.err
timeout 90 /tmp/vmm_probe 16 1 > $O/vmm_probe.txt 2>&1; echo "rc=$?" >> $O/vmm_probe.txt; cat $O/vmm_probe.txt
timeout 150 python -m pytest tests -m gpu -x -q 2>&1 | tail -4 > $O/pytest_tail.txt; cat $O/pytest_tail.txt
