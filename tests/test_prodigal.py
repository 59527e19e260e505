"""Gene-calling contract in front of the accelerated path: the ProdigalRunner / ProdigalGeneFeatureParser mirrors
against goldens produced by the reference's own classes (tools/gen_prodigal_golden.py: checkm/prodigal.py imported from
/root/reference, run against a stub `prodigal` whose output depends only on the translation table)."""
import json
import logging
import os
import random
import stat
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))


@pytest.fixture(scope="module")
def golden():
    return json.load(open(os.path.join(ROOT, "tests", "golden", "prodigal_cases.json")))["cases"]


def test_gff_parser_matches_the_reference(golden, tmp_path):
    from checkm_amd.prodigal import ProdigalGeneFeatureParser
    import gen_prodigal_golden as g
    for n, case in enumerate(golden):
        contigs = [tuple(c) for c in case["contigs"]]
        for table in ("4", "11"):
            f = tmp_path / ("g%d_%s.gff" % (n, table))
            f.write_text(case["spec"][table]["gff"])
            p = ProdigalGeneFeatureParser(str(f))
            got = g.parser_view(p, contigs, random.Random(n * 10 + int(table)))
            assert got == case["parser"][table], (n, table)


def test_runner_chooses_the_table_and_leaves_the_files_the_reference_does(golden, tmp_path, monkeypatch):
    from checkm_amd.prodigal import ProdigalRunner
    import gen_prodigal_golden as g
    logging.disable(logging.CRITICAL)
    stub_dir = tmp_path / "bin"
    stub_dir.mkdir()
    stub = stub_dir / "prodigal"
    stub.write_text(g.STUB)
    os.chmod(str(stub), os.stat(str(stub)).st_mode | stat.S_IEXEC)
    monkeypatch.setenv("PATH", str(stub_dir) + os.pathsep + os.environ["PATH"])
    rng = random.Random(1)
    for n, case in enumerate(golden):
        fasta = tmp_path / ("bin%d.fna" % n)
        with open(str(fasta), "w") as f:
            for cid, L in case["contigs"]:
                f.write(">%s some description\n" % cid)
                seq = "".join(rng.choice("ACGT") for _ in range(L))
                for i in range(0, L, 60):
                    f.write(seq[i:i + 60] + "\n")
        spec = tmp_path / ("spec%d.json" % n)
        spec.write_text(json.dumps(case["spec"]))
        monkeypatch.setenv("PRODIGAL_STUB_SPEC", str(spec))
        out = tmp_path / ("out%d" % n)
        out.mkdir()
        if "exit" in case:
            with pytest.raises(SystemExit) as e:
                ProdigalRunner(str(out)).run(str(fasta), case["bNucORFs"])
            assert e.value.code == case["exit"]
            continue
        r = ProdigalRunner(str(out))
        assert r.run(str(fasta), case["bNucORFs"]) == case["best"], n
        assert {f: open(os.path.join(str(out), f)).read() for f in sorted(os.listdir(str(out)))} == case["files"], n
        assert r.areORFsCalled(case["bNucORFs"])
    logging.disable(logging.NOTSET)
