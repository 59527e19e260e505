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


_ctx2 = None


def get_ctx2():
    """A second context on the same device: MarkerGeneFinder.find keeps two batches of bins in flight, one per context (a context
    runs one search at a time), so that the ingest, the host part and the tail of one batch run under the SSV phase of the other."""
    global _ctx2
    if _ctx2 is None:
        _ctx2 = _lib.Context(get_ctx().device)
    return _ctx2


def close():
    global _ctx
    import sys
    mod = sys.modules.get("checkm_amd.markerGeneFinder")
    if mod is not None:
        mod.release_scan(final=True)    # hits, sequences and profile databases of this context
    global _ctx2
    if _ctx2 is not None:
        _ctx2.close()
        _ctx2 = None
    if _ctx is not None:
        _ctx.close()
        _ctx = None
