"""User-supplied vectors from a REAL hmmsearch (tests/golden/hmmer/README.md): every (NAME.hmm, NAME.faa, NAME.domtblout) triple found
there is diffed against the oracle's rows (CPU) and against the table the MI355X scan writes (-m gpu).  Reference call being matched:
checkm/hmmer.py:61-74 with the options of checkm/markerGeneFinder.py:140-142.  No triple exists yet (no HMMER anywhere near this
repository): the tests skip themselves and the scan half stays "parity unpinned"."""
import glob
import json
import os

import pytest

from tools import diff_vs_hmmsearch as dvh

HERE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "hmmer")


def triples():
    out = []
    for hmm in sorted(glob.glob(os.path.join(HERE, "*.hmm"))):
        base = hmm[:-4]
        if os.path.exists(base + ".faa") and os.path.exists(base + ".domtblout"):
            opts = json.load(open(base + ".json")) if os.path.exists(base + ".json") else {}
            out.append((base, opts))
    return out


def read_fasta(path):
    recs, name, desc, seq = [], None, "", []
    for line in open(path):
        if line.startswith(">"):
            if name is not None:
                recs.append((name, desc, "".join(seq)))
            head = line[1:].rstrip("\n").split(None, 1)
            name, desc, seq = head[0], (head[1] if len(head) > 1 else ""), []
        else:
            seq.append(line.strip())
    if name is not None:
        recs.append((name, desc, "".join(seq)))
    return recs


def check(d, opts):
    assert d["rows_hmmsearch"] > 0, d
    assert d["only_hmmsearch"] == 0 and d["only_ours"] == 0 and d["coords"] == 0, d          # hit-for-hit, coordinate-for-coordinate
    assert d["last_digit"] <= opts.get("allow_last_digit_fraction", 0.02) * d["rows_hmmsearch"], d
    assert d.get("decision_relevant", 0) == 0, d          # (no row whose printed digits put it on the other side of a GA / TC / NC cutoff: checkm/resultsParser.py:340-377)


def test_loader_finds_complete_triples_only(tmp_path, monkeypatch):
    for name in ("a.hmm", "a.faa", "a.domtblout", "b.hmm", "b.faa"):
        (tmp_path / name).write_text("x")
    monkeypatch.setattr("tests.test_hmmer_golden.HERE", str(tmp_path))
    assert [os.path.basename(b) for b, _ in triples()] == ["a"]
    assert read_fasta.__doc__ is None and triples.__doc__ is None      # (helpers, not tests)


def test_oracle_against_hmmsearch_tables(tmp_path):
    found = triples()
    if not found:
        pytest.skip("no (hmm, faa, domtblout) triple from a real hmmsearch under tests/golden/hmmer: scan-half parity stays unpinned")
    from oracle import p7
    for base, opts in found:
        hs = p7.HmmSet(base + ".hmm")
        recs = read_fasta(base + ".faa")
        rows = hs.search(list(range(hs.n)), [p7.digitize(r[2]) for r in recs], [r[0] for r in recs])
        ours = str(tmp_path / (os.path.basename(base) + ".oracle.tbl"))
        with open(ours, "w") as f:
            f.write(hs.format_domtblout(rows, [r[0] for r in recs], [r[1] for r in recs]))
        check(dvh.diff_tables(base + ".domtblout", ours, base + ".hmm"), opts)
        hs.close()


@pytest.mark.gpu
def test_gpu_scan_against_hmmsearch_tables(gpu_ctx, tmp_path):
    found = triples()
    if not found:
        pytest.skip("no (hmm, faa, domtblout) triple from a real hmmsearch under tests/golden/hmmer: scan-half parity stays unpinned")
    from checkm_amd.markerGeneFinder import scan_files
    for base, opts in found:
        ours = str(tmp_path / (os.path.basename(base) + ".gpu.tbl"))
        scan_files(base + ".hmm", [base + ".faa"], [ours])
        check(dvh.diff_tables(base + ".domtblout", ours, base + ".hmm"), opts)
