"""Parity against a REAL hmmsearch, the moment one exists: skipped while HMMER is absent (it is absent from /root/reference, from
this image and from the GPU box), so the suite pins the scan half by itself the first time it runs on a machine that has HMMER
(reference call: checkm/hmmer.py:61-74, probe at :131-137).  The CPU half checks the diff tool on hand-made tables."""
import shutil

import pytest

from synthdata import synth
from tests import common
from tools import diff_vs_hmmsearch as dvh

ROW = "c000001_7            -            399 SYN000               PF90000.1     63   1.2e-21   70.3   0.1   1   1   3.4e-22   5.6e-21   68.9   0.1     1    63    20    84    18    90 0.97 # 1 # 2 # 1 # ID=1_7"


def _tbl(path, rows):
    with open(path, "w") as f:
        f.write("# header\n")
        for r in rows:
            f.write(r + "\n")
        f.write("#\n# [ok]\n")
    return str(path)


def test_diff_tool_classifies_rows(tmp_path):
    a = _tbl(tmp_path / "a.tbl", [ROW, ROW.replace("c000001_7 ", "c000001_8 ")])
    assert dvh.diff_tables(a, a)["identical"] == 2
    b = _tbl(tmp_path / "b.tbl", [ROW.replace(" 70.3 ", " 70.4 ").replace("1.2e-21", "1.3e-21"), ROW.replace("c000001_7 ", "c000001_9 ")])
    d = dvh.diff_tables(a, b)
    assert (d["last_digit"], d["only_hmmsearch"], d["only_ours"], d["identical"], d["coords"]) == (1, 1, 1, 0, 0)
    c = _tbl(tmp_path / "c.tbl", [ROW.replace("    20    84 ", "    21    84 ")])
    assert dvh.diff_tables(a, c)["coords"] == 1
    assert dvh.merge([d, d])["last_digit"] == 2


@pytest.mark.gpu
def test_rows_against_real_hmmsearch(gpu_ctx, tmp_path):
    if shutil.which("hmmsearch") is None:
        pytest.skip("no hmmsearch on PATH: scan-half parity stays unpinned (DESIGN.md section 2)")
    profs = common.mixed_profiles()
    hmm = common.hmm_file("mixed", profs)
    faa = str(tmp_path / "genes.faa")
    synth.write_fasta(faa, synth.make_bin(profs, 31337, n_orfs=600, dup_frac=0.3))
    d = dvh.run(hmm, faa)
    print(d)
    assert d["rows_hmmsearch"] > 0
    assert d["only_hmmsearch"] == 0 and d["only_ours"] == 0 and d["coords"] == 0, d          # hit-for-hit, coordinate-for-coordinate
    assert d["last_digit"] <= 0.02 * d["rows_hmmsearch"], d                                  # float summation order may flip a last digit (DESIGN D4/D5)
