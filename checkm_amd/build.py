"""Builds libcheckm_hip.so in-tree with hipcc for gfx950 (cross-compiles without a GPU)."""
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")


def build(force=False, jobs=8, verbose=False):
    cmd = ["make", "-C", CSRC, "-j%d" % jobs]
    if not verbose:
        cmd.append("-s")
    if force:
        subprocess.check_call(["make", "-s", "-C", CSRC, "clean"])
    subprocess.check_call(cmd)
    path = os.path.join(CSRC, "libcheckm_hip.so")
    if not os.path.exists(path):
        raise RuntimeError("build did not produce %s" % path)
    return path
