set -u
export TMPDIR=/tmp
O=gpurun_out/r03g; mkdir -p $O
python -m pytest tests/test_gpu_lineage.py -m gpu -q -x 2>&1 | tail -30 > $O/pytest_lineage.txt
tail -12 $O/pytest_lineage.txt
python -m pytest tests -m gpu -q 2>&1 | tail -30 > $O/pytest_gpu.txt
tail -5 $O/pytest_gpu.txt
