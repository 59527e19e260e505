"""HMM files as CheckM's data bundle holds them (HMMER3/b and /f headers, DATE / NSEQ / EFFN / CKSUM / BM / SM / COM lines, DESC with spaces,
annotation columns behind every match line, records without COMPO, records without ACC and cutoffs of their own): the oracle's reader and
the mirror of CheckM's header parser (checkm/hmmerModelParser.py:46-83) on a fixture in that layout (tests/common.py:real_format_hmm_text).
The device reader is checked on the same fixture by tests/test_gpu_scan.py::test_real_format_hmm_file_searched."""
from synthdata import synth
from checkm_amd.hmmerModelParser import HmmModelParser
from oracle import p7
from tests import common


def _profiles():
    return synth.small_profiles(11, 12, 40, 300)[:8]


def test_oracle_reads_the_real_layout_like_the_plain_one():
    profs = _profiles()
    real, plain = common.real_format_hmm_file("real8", profs), common.hmm_file("plain8", profs)
    a, b = p7.HmmSet(real), p7.HmmSet(plain)
    assert a.n == b.n == len(profs) and [a.M(i) for i in range(a.n)] == [p.M for p in profs]
    assert [a.name(i) for i in range(a.n)] == [p.name for p in profs]
    assert a.acc(3) == "" and a.acc(7) == "" and a.acc(0) == profs[0].acc                  # records 3 and 7 carry no ACC line
    recs = synth.make_bin(profs, 5, n_orfs=40, dup_frac=0.3)
    dsq, names = [p7.digitize(r[2]) for r in recs], [r[0] for r in recs]
    ra, rb = a.search(list(range(a.n)), dsq, names), b.search(list(range(b.n)), dsq, names)
    assert len(ra) >= len(profs) and [common.row_key(r) for r in ra] == [common.row_key(r) for r in rb]     # (a missing COMPO line is recomputed from the model, annotation columns are skipped)
    a.close(); b.close()


def test_sticky_header_view_of_the_real_layout():
    profs = _profiles()
    models = HmmModelParser(common.real_format_hmm_file("real8", profs)).models()
    # records 3 and 7 have neither ACC nor cutoffs: CheckM's parser keeps the previous record's (checkm/hmmerModelParser.py:56 never
    # resets its key table), so they are filed under -- and replace -- the accession of records 2 and 6
    assert len(models) == 6 and profs[2].acc in models and models[profs[2].acc].name == profs[3].name
    assert models[profs[6].acc].name == profs[7].name and models[profs[6].acc].leng == profs[7].M
    for k in (0, 1, 4, 5):
        m = models[profs[k].acc]
        assert (m.name, m.leng) == (profs[k].name, profs[k].M)
    assert models[profs[2].acc].ga == models[profs[1].acc].ga or models[profs[2].acc].ga is not None     # cutoffs travel with the key table too
