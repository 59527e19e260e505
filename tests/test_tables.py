"""ckm_tables_read (the library's domtblout reader, no device needed) against the Python restatement of the reference's
HMMERParser / HmmerHitDOM (checkm/hmmer.py:184-200, 255-285)."""
import pytest

from checkm_amd import _lib
from checkm_amd.hmmer import read_domtblout

ROWS = [
    "c1_1                 -            312 PF00318.15           PF00318.15    211   1.3e-45  152.9   0.0   1   1   2.1e-47   1.5e-45  152.7   0.0     1   211    96   307    96   307 0.99 # 3 # 941 # 1 # ID=1_1;partial=00",
    "c1_2                 -            150 SYN0007              -              74   4.4e-12   40.7   1.2   1   2   0.00021     0.015   10.3   0.1     5    30     3    33     1    40 0.80 -",
    "c1_2                 -            150 SYN0007              -              74   4.4e-12   40.7   1.2   2   2   3.1e-11   2.2e-09   33.0   0.3     2    74    60   141    58   145 0.91 -",
    "scaffold_7_12        acc1         88 TIGR00001            TIGR00001      60         0  999.9  25.3   1   1         0         0  999.8  25.3     1    60     1    88     1    88 1.00 ribosomal   protein  L1",
]


def _write(path, lines):
    with open(path, "w") as f:
        f.write("\n".join(lines) + "\n")


def test_reader_matches_the_python_parser(tmp_path):
    p0, p1, p2 = str(tmp_path / "a.txt"), str(tmp_path / "b.txt"), str(tmp_path / "missing.txt")
    _write(p0, ["#  header", "# more"] + ROWS[:3] + ["#", "# [ok]"])
    _write(p1, [ROWS[3], "", ROWS[0]])                      # the empty line ends the table: the last row is never read
    t = _lib.Tables([p0, p1, p2])
    try:
        assert t.nbins == 3 and list(t.bin_row_off) == [0, 3, 4, 4]
        assert list(t.missing) == [False, False, True]
        want = [h.as_dict() for h in read_domtblout(p0)] + [h.as_dict() for h in read_domtblout(p1)]
        assert len(want) == 4
        for r, w in enumerate(want):
            assert t.hit(r) == w, (r, t.hit(r), w)
        assert t.hit(1)["query_accession"] == "SYN0007"      # '-' replaced by the query name
        assert t.hit(3)["target_description"] == "ribosomal protein L1"
        assert t.hit(3)["full_bias"] == 25.3                 # float64, not the float32 column
        assert t.assign_models(["PF00318.15", "SYN0007"]) == 1
        assert [int(x) for x in t.column("model")] == [0, 1, 1, 0xFFFFFFFF]
    finally:
        t.close()


def test_malformed_rows_are_errors(tmp_path):
    p = str(tmp_path / "bad.txt")
    _write(p, ["c1_1 - 312 PF00318.15 PF00318.15 211 1.3e-45"])
    with pytest.raises(_lib.CkmError) as e:
        _lib.Tables([p])
    assert e.value.code == -3 and "fewer than 23 columns" in str(e.value)
    _write(p, [ROWS[0].replace(" 312 ", " 3x2 ")])
    with pytest.raises(_lib.CkmError) as e:
        _lib.Tables([p])
    assert e.value.code == -3 and "not an integer" in str(e.value)
