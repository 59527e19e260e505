"""Process-wide device contexts: get_ctx() is the context everything single-shot uses; get_ctx_k(k) are the further ones find() alternates between.
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


_extra = []


def get_ctx_k(k):
    """Context number k on the process's device (k = 0: get_ctx()).  MarkerGeneFinder.find keeps several batches of bins in flight, one
    per context (a context runs one search at a time), so that the ingest, the host part and the chain tail of one batch run under the
    SSV phase of the others.  The contexts of a device share the resident profile databases (read-only device memory); each has its
    own streams, tables and float workspace, sized from a quarter of the memory that is free when it is created."""
    if k == 0:
        return get_ctx()
    while len(_extra) < k:
        _extra.append(_lib.Context(get_ctx().device))
    return _extra[k - 1]


def get_ctx2():
    return get_ctx_k(1)


def close():
    global _ctx
    import sys
    mod = sys.modules.get("checkm_amd.markerGeneFinder")
    if mod is not None:
        mod.release_scan(final=True)    # hits, sequences and profile databases of this context
    while _extra:
        _extra.pop().close()
    if _ctx is not None:
        _ctx.close()
        _ctx = None
