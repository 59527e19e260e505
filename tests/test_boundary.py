"""The drop-in boundary: the shared object loads without a GPU and exports every symbol the header
declares; the product never touches the oracle; without a device the product fails loudly."""
import ctypes
import os
import re

import pytest

from checkm_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_functions():
    src = open(os.path.join(ROOT, "include", "checkm_hip.h")).read()
    src = re.sub(r'/\*.*?\*/', '', src, flags=re.S)
    return sorted(set(re.findall(r'\b(ckm_[a-z_0-9]+)\s*\(', src)))


def test_library_exports_every_declared_symbol():
    lib = ctypes.CDLL(_lib.LIB_PATH)
    names = header_functions()
    assert len(names) >= 20
    for n in names:
        assert hasattr(lib, n), "libcheckm_hip.so does not export %s" % n
    assert sorted(_lib.EXPORTS) == names
    assert lib.ckm_abi_version() == _lib.ABI_VERSION


def test_product_never_imports_the_oracle():
    bad = []
    for dirpath, _dirs, files in os.walk(os.path.join(ROOT, "checkm_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".cpp", ".h")):
                txt = open(os.path.join(dirpath, f)).read()
                if re.search(r'^\s*(from|import)\s+oracle\b', txt, flags=re.M) or 'p7oracle' in txt or 'reduce_oracle' in txt:
                    bad.append(os.path.join(dirpath, f))
                # nor the synthetic-input generators, nor the host emulation of the gene pipeline (tests/emu): test infrastructure both
                if f.endswith('.py') and re.search(r'^\s*(from|import)\s+(synthdata|tests)\b', txt, flags=re.M):
                    bad.append(os.path.join(dirpath, f))
    assert not bad, bad


def test_no_device_is_a_loud_error():
    """On the CPU-only build box ckm_ctx_create must fail with ENODEV (on a GPU box this test is vacuous)."""
    if _lib.device_count() > 0:
        pytest.skip("a device is visible")
    with pytest.raises(_lib.CkmError) as e:
        _lib.Context(0)
    assert e.value.code == -4
    assert "no CPU path" in str(e.value) or "HIP" in str(e.value)


def test_bad_arguments_return_codes():
    lib = _lib.load()
    n = ctypes.c_int32()
    assert lib.ckm_profiles_count(None, ctypes.byref(n)) == -1
    assert b"NULL" in lib.ckm_last_error()


@pytest.mark.skipif(not os.path.isdir("/root/reference/checkm"), reason="the reference package is only present in the build container")
def test_dropin_rebinds_the_reference_classes(tmp_path):
    """INTEGRATION.md option A, in a subprocess so the reference package never enters this test process."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import checkm_amd.dropin as d; d.install()\n"
            "import checkm.markerGeneFinder as a, checkm.resultsParser as b, checkm.markerSets as c, checkm.hmmerAligner as e, checkm.aminoAcidIdentity as f, checkm.hmmerModelParser as g\n"
            "mods = [a.MarkerGeneFinder.__module__, b.ResultsParser.__module__, b.ResultsManager.__module__, c.MarkerSet.__module__, c.MarkerSetParser.__module__,\n"
            "        e.HmmerAligner.__module__, f.AminoAcidIdentity.__module__]\n"
            "assert all(m.startswith('checkm_amd.') for m in mods), mods\n"
            "assert g.HmmModel.__module__ == 'checkm.hmmerModelParser'\n"
            "assert all(hasattr(e.HmmerAligner, n) for n in ('makeAlignmentTopHit', 'makeAlignmentToPhyloMarkers', 'makeAlignmentsOfMultipleHits'))\n"
            "import checkm_amd.markerGeneFinder as h\n"
            "import shutil\n"
            "gc = h.gene_caller()            # nucleotide bins: the device gene finder when no prodigal binary exists, else the mirror of ProdigalRunner\n"
            "assert gc == 'device' if shutil.which('prodigal') is None else gc.__module__ == 'checkm_amd.prodigal'\n"
            "import checkm.prodigal as pr\n"
            "assert pr.ProdigalRunner.__module__ == 'checkm_amd.prodigal'\n"
            "print('ok')\n")
    env = dict(os.environ, PYTHONPATH="/root/reference" + os.pathsep + root, CHECKM_DATA_PATH=str(tmp_path), PYTHONDONTWRITEBYTECODE="1")
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=120)
    assert out.returncode == 0 and out.stdout.strip().endswith("ok"), out.stderr[-1500:]


def test_entry_points_refuse_to_run_without_a_device(tmp_path):
    """No CPU path anywhere: bench.py stops with a message, MarkerGeneFinder.find logs an error and exits with code 1 (as the
    reference does when hmmsearch is missing, checkm/hmmer.py:131-137).  Only meaningful where there is no GPU."""
    import subprocess
    import sys
    import torch
    if torch.cuda.is_available():
        pytest.skip("a device is present")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--steps", "1"], capture_output=True, text=True, timeout=300)
    assert out.returncode != 0 and "needs a GPU" in (out.stderr + out.stdout)
    faa = tmp_path / "b.faa"
    faa.write_text(">g_1\nMKV*\n")
    code = ("import sys, logging\n"
            "logging.basicConfig(stream=sys.stderr)\n"
            "from checkm_amd.markerGeneFinder import MarkerGeneFinder\n"
            "MarkerGeneFinder(1).find([%r], %r, 'hmmer.analyze.txt', 'hmmer.analyze.ali.txt', 'none.hmm', False, False, True)\n" % (str(faa), str(tmp_path / "out")))
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300, env=dict(os.environ, PYTHONPATH=root))
    assert out.returncode == 1 and "No usable MI355X" in out.stderr
