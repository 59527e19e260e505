import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
# the tests drive one device in this process unless a test asks for workers itself (tests/test_gpu_workers.py): on a multi-GPU box
# MarkerGeneFinder.find would otherwise spawn a worker per visible device (checkm_amd/workers.py)
os.environ.setdefault("CKM_GPUS", "")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def gpu_ctx():
    from checkm_amd import _lib
    if _lib.device_count() < 1:
        pytest.fail("no HIP device visible but a gpu-marked test was selected")
    ctx = _lib.Context(0)
    yield ctx
    ctx.close()
