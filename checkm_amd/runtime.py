"""Process-wide device context: one ckm_ctx per (process, device), as the C ABI requires.
The device is CHECKM_AMD_DEVICE if set, else LOCAL_RANK when launched one-process-per-GPU, else 0."""
import atexit
import os

from checkm_amd import _lib

_ctx = None


def get_ctx():
    global _ctx
    if _ctx is None:
        dev = int(os.environ.get("CHECKM_AMD_DEVICE", os.environ.get("LOCAL_RANK", "0")))
        _ctx = _lib.Context(dev)          # raises CkmError(ENODEV) without a gfx950: there is no CPU path
        atexit.register(close)
    return _ctx


def close():
    global _ctx
    import sys
    mod = sys.modules.get("checkm_amd.markerGeneFinder")
    if mod is not None:
        mod.release_scan()              # hits, sequences and profile databases of this context
    if _ctx is not None:
        _ctx.close()
        _ctx = None
