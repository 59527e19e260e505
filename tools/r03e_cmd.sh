set -u
export TMPDIR=/tmp
O=gpurun_out/r03e; mkdir -p $O
python -m pytest tests -m gpu -q 2>&1 | tail -30 > $O/pytest_gpu.txt
tail -6 $O/pytest_gpu.txt
python - > $O/orf_flags.txt 2>&1 <<'P'
import sys; sys.path.insert(0, ".")
from checkm_amd import _lib
ctx = _lib.Context(0)
for n in (1 << 28, 1 << 30, 1 << 31):
    ms = _lib.debug_orf_flags(ctx, n, 10)
    print("orf_flags_kernel %d bases: %.3f ms per launch, %.1f GB/s (2 B per base)" % (n, ms, 2.0 * n / ms / 1e6))
P
cat $O/orf_flags.txt
