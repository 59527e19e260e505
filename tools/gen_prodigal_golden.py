#!/usr/bin/env python
"""Goldens of the gene-calling contract produced by the REFERENCE's own classes (checkm/prodigal.py imported from
/root/reference): ProdigalGeneFeatureParser on synthetic GFF files, and ProdigalRunner.run against a stub `prodigal` executable
whose output depends only on the -g table (so the table choice, the retry path and the files left behind are the reference's).
usage: PYTHONPATH=/root/reference python tools/gen_prodigal_golden.py > tests/golden/prodigal_cases.json"""
import json
import os
import random
import stat
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
DATA = tempfile.mkdtemp(prefix="ckm_data_")          # the reference wants a data root at import time (checkm/checkmData.py:115-121)
os.makedirs(os.path.join(DATA, "pfam"))
open(os.path.join(DATA, "pfam", "Pfam-A.hmm.dat"), "w").close()
os.environ["CHECKM_DATA_PATH"] = DATA

STUB = r'''#!/usr/bin/env python3
import sys, json, os
a = sys.argv[1:]
if a == ['-h']:
    sys.exit(0)
opt = dict(zip(a[0::2], a[1::2])) if False else {}
i = 0
flags = {}
while i < len(a):
    if a[i] in ('-p', '-f', '-g', '-a', '-d', '-i'):
        flags[a[i]] = a[i + 1]; i += 2
    else:
        i += 1
spec = json.load(open(os.environ['PRODIGAL_STUB_SPEC']))
case = spec[flags['-g']]
if case.get('fail_single') and flags['-p'] == 'single':
    sys.exit(3)
if case.get('fail_always'):
    sys.exit(4)
sys.stdout.write(case['gff'])
open(flags['-a'], 'w').write(case['faa'] + ('# %s\n' % flags['-p']))
if '-d' in flags:
    open(flags['-d'], 'w').write(case['fna'])
'''


def make_gff(rng, contigs, density, table):
    out = ['##gff-version  3\n']
    for cid, L in contigs:
        out.append('# Sequence Data: seqnum=1;seqlen=%d;seqhdr="%s"\n' % (L, cid))
        out.append('# Model Data: version=Prodigal.v2.6.3;run_type=Single;model="Ab initio";gc_cont=50.00;transl_table=%d;uses_sd=1\n' % table)
        pos, k = 1, 0
        while pos < L - 100:
            glen = rng.randrange(90, 1500)
            gap = int(glen * (1.0 - density) / max(density, 0.05) * rng.uniform(0.3, 1.7)) - (rng.randrange(0, 40) if rng.random() < 0.2 else 0)
            end = min(L, pos + glen)
            k += 1
            out.append('%s\tProdigal_v2.6.3\tCDS\t%d\t%d\t%.1f\t%s\t0\tID=1_%d;partial=00;start_type=ATG;\n' % (cid, pos, end, rng.uniform(1, 99), rng.choice('+-'), k))
            pos = max(1, end + gap)
    if rng.random() < 0.3:
        out.insert(3, '"\n')
    return ''.join(out)


def parser_view(p, contigs, rng):
    view = {"translationTable": p.translationTable, "genes": {s: {g: list(v) for g, v in d.items()} for s, d in p.genes.items()},
            "lastCodingBase": dict(p.lastCodingBase), "coding": {}}
    for cid, L in contigs + [("absent", 10)]:
        qs = [(0, None)] + [(rng.randrange(0, L), rng.randrange(0, L + 50)) for _ in range(4)] + [(5, 2), (L + 10, None)]
        view["coding"][cid] = [[s, e, int(p.codingBases(cid, s, e))] for s, e in qs]
    return view


def main():
    from checkm.prodigal import ProdigalGeneFeatureParser, ProdigalRunner
    import logging
    logging.disable(logging.CRITICAL)
    rng = random.Random(20250926)
    cases = []
    tmp = tempfile.mkdtemp(prefix="ckm_prodigal_golden_")
    stub_dir = os.path.join(tmp, "bin")
    os.makedirs(stub_dir)
    stub = os.path.join(stub_dir, "prodigal")
    open(stub, "w").write(STUB)
    os.chmod(stub, os.stat(stub).st_mode | stat.S_IEXEC)
    os.environ["PATH"] = stub_dir + os.pathsep + os.environ["PATH"]
    # (density of table 4, density of table 11, total bases big?, failure mode)
    plans = [(0.90, 0.88, True, None), (0.90, 0.80, True, None), (0.76, 0.70, True, None), (0.69, 0.50, True, None), (0.755, 0.70, True, None),
             (0.97, 0.96, True, None), (0.99, 0.93, True, None), (0.85, 0.60, False, None), (0.90, 0.80, True, "fail_single"), (0.5, 0.9, True, None),
             (0.9, 0.9, True, "fail_always")]
    for n, (d4, d11, big, fail) in enumerate(plans):
        contigs = [("contig_%d" % k, rng.randrange(30000, 90000) if big else rng.randrange(5000, 20000)) for k in range(3)]
        fasta = os.path.join(tmp, "bin%d.fna" % n)
        with open(fasta, "w") as f:
            for cid, L in contigs:
                f.write(">%s some description\n" % cid)
                seq = "".join(rng.choice("ACGT") for _ in range(L))
                for i in range(0, L, 60):
                    f.write(seq[i:i + 60] + "\n")
        spec = {}
        for table, d in ((4, d4), (11, d11)):
            spec[str(table)] = {"gff": make_gff(rng, contigs, d, table), "faa": ">%s_1 # 1 # 90 # 1 # ID=1_1\nMKT*\n" % contigs[0][0],
                                "fna": ">%s_1 # 1 # 90 # 1 # ID=1_1\nATGAAAACCTAA\n" % contigs[0][0]}
            if fail:
                spec[str(table)][fail] = True
        spec_path = os.path.join(tmp, "spec%d.json" % n)
        json.dump(spec, open(spec_path, "w"))
        os.environ["PRODIGAL_STUB_SPEC"] = spec_path
        out = os.path.join(tmp, "out%d" % n)
        os.makedirs(out)
        case = {"contigs": contigs, "spec": spec, "bNucORFs": bool(n % 2 == 0)}
        try:
            best = ProdigalRunner(out).run(fasta, case["bNucORFs"])
            case["best"] = best
            case["files"] = {f: open(os.path.join(out, f)).read() for f in sorted(os.listdir(out))}
        except SystemExit as e:
            case["exit"] = int(e.code) if isinstance(e.code, int) else 1
        views = {}
        for table in ("4", "11"):
            g = os.path.join(tmp, "g%d_%s.gff" % (n, table))
            open(g, "w").write(spec[table]["gff"])
            views[table] = parser_view(ProdigalGeneFeatureParser(g), contigs, random.Random(n * 10 + int(table)))
        case["parser"] = views
        cases.append(case)
    json.dump({"note": "generated by tools/gen_prodigal_golden.py from the reference's checkm/prodigal.py", "cases": cases}, sys.stdout)
    sys.stdout.write("\n")


if __name__ == "__main__":
    main()
