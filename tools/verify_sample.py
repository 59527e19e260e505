#!/usr/bin/env python
"""Oracle check of a finished lineage_wf-shaped run AT THE SIZE IT RAN (bench.py --verify, tests/test_gpu_fullscale.py).

TEST INFRASTRUCTURE: imports oracle/ (the CPU restatements).  The product never imports this module.

For K sampled bins of an output directory that MarkerGeneFinder.find + ResultsParser wrote:
  scan half    the lines of bins/<binId>/<table> that belong to a SAMPLE of the bin's models -- at least `n_models`, one of every SSV
               launch class the bin's model list holds (8 lanes: 100 + ceil(M/16) for M <= 512; 16 lanes: ceil(M/32) above) and the rest
               at random -- are compared, as text, with what oracle.p7 (the scalar restatement of hmmsearch) writes for the same genes.
               Z is the bin's sequence count and domZ is per model, so the rows of a model do not depend on the other models searched.
  reduce half  the bin's row of the QA table is compared with oracle.reduce_oracle run on the WHOLE written table of the bin.
Returns {"bins", "models", "rows", "identical", ...}; `identical` is true only if every compared line and every QA row is equal.
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

SSV16_Q = [1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 18, 20, 22, 24, 26, 28, 30, 32, 36, 40, 48, 56, 64]
SSV8_Q = [2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 18, 20, 22, 24, 26, 28, 30, 32]


def launch_class(M):
    """The SSV launch class of a model of M nodes (host_profile.cpp: ssv_Q_for / the 8-lane classes; 65 = no SSV instance)."""
    if M <= 512:
        return 100 + next(q for q in SSV8_Q if q * 16 >= M)
    return next((q for q in SSV16_Q if q * 32 >= M), 65)


def sample_models(lengths, n_models, rng):
    """lengths: {key: M}.  One key of every launch class present, then random ones up to n_models.  Sorted keys."""
    by_cls = {}
    for k in sorted(lengths):
        by_cls.setdefault(launch_class(lengths[k]), []).append(k)
    pick = set()
    for cls in sorted(by_cls):
        v = by_cls[cls]
        pick.add(v[int(rng.integers(0, len(v)))])
    rest = [k for k in sorted(lengths) if k not in pick]
    rng.shuffle(rest)
    for k in rest[:max(0, n_models - len(pick))]:
        pick.add(k)
    return sorted(pick), len(by_cls)


def read_fasta(path):
    recs, name, desc, seq = [], None, "", []
    with open(path) as f:
        for line in f:
            if line.startswith(">"):
                if name is not None:
                    recs.append((name, desc, "".join(seq)))
                head = line[1:].rstrip("\n")
                name, _, desc = head.partition(" ")
                seq = []
            else:
                seq.append(line.strip())
    if name is not None:
        recs.append((name, desc, "".join(seq)))
    return recs


def table_lines(path):
    with open(path) as f:
        return [ln.rstrip("\n") for ln in f if ln.strip() and not ln.startswith("#")]


def _omodels(models):
    return {a: {"acc": a, "ga": list(m.ga) if m.ga else None, "tc": list(m.tc) if m.tc else None, "nc": list(m.nc) if m.nc else None, "leng": m.leng}
            for a, m in models.items()}


def verify(out_dir, table, hmm_path, bin_ids, files, models_by_bin, k_bins=3, n_models=40, seed=1, threads=None,
           marker_sets=None, qa_rows=None, pfam_text=None, with_rows=16):
    """out_dir/bins/<binId>/<table> against the oracles for `k_bins` sampled bins.
    models_by_bin: what find() returned ({binId: {acc: HmmModel}}).  marker_sets ({binId: BinMarkerSets}) + qa_rows ({binId: tab-separated
    row of printSummary format 1}) + pfam_text switch the reduce half on."""
    from concurrent.futures import ThreadPoolExecutor
    from oracle import p7
    from oracle import reduce_oracle as ro
    rng = np.random.default_rng(seed)
    threads = threads or min(32, os.cpu_count() or 1)
    hs = p7.HmmSet(hmm_path)
    index = {(hs.acc(i) or hs.name(i)): i for i in range(hs.n)}
    qname = {i: hs.name(i) for i in range(hs.n)}
    pick_bins = sorted(rng.choice(len(bin_ids), size=min(k_bins, len(bin_ids)), replace=False).tolist())
    out = {"bins": [], "models": 0, "rows": 0, "launch_classes": 0, "identical": True, "qa_rows_identical": None, "mismatches": []}
    classes_seen = set()
    jobs = []
    for b in pick_bins:
        binId = bin_ids[b]
        accs = [a for a in models_by_bin[binId] if a in index]
        lengths = {a: hs.M(index[a]) for a in accs}
        sample, _ncls = sample_models(lengths, n_models, rng)
        # ... and up to `with_rows` models that HAVE rows in the written table (a random model of a large database mostly has none, and
        # "no rows on either side" is the weakest kind of agreement)
        name_to_acc = {hs.name(index[a]): a for a in accs}
        hit_accs = sorted({name_to_acc[ln.split()[3]] for ln in table_lines(os.path.join(out_dir, "bins", binId, table)) if ln.split()[3] in name_to_acc} - set(sample))
        rng.shuffle(hit_accs)
        sample = sample + hit_accs[:with_rows]
        sample = sorted(sample, key=lambda a: index[a])          # rows come in HMM-file order (hmmsearch: one query after the other)
        classes_seen |= {launch_class(lengths[a]) for a in sample}
        recs = read_fasta(files[b])
        dsq = [p7.digitize(r[2]) for r in recs]
        jobs.append((b, binId, sample, recs, dsq))
    with ThreadPoolExecutor(max_workers=threads) as ex:
        futs = []
        for (b, binId, sample, recs, dsq) in jobs:
            names = [r[0] for r in recs]
            futs.append([ex.submit(hs.search, [index[a]], dsq, names) for a in sample])            # the C call releases the GIL
        for (b, binId, sample, recs, dsq), fl in zip(jobs, futs):
            rows = [r for f in fl for r in f.result()]
            want = [ln for ln in hs.format_domtblout(rows, [r[0] for r in recs], [r[1] for r in recs]).split("\n") if ln.strip() and not ln.startswith("#")]
            qn = {qname[index[a]] for a in sample}
            path = os.path.join(out_dir, "bins", binId, table)
            got_all = table_lines(path)
            got = [ln for ln in got_all if ln.split()[3] in qn]
            same = got == want
            if not same:
                out["identical"] = False
                for i in range(max(len(got), len(want))):
                    a = got[i] if i < len(got) else None
                    c = want[i] if i < len(want) else None
                    if a != c:
                        out["mismatches"].append({"bin": binId, "line": i, "got": a, "oracle": c})
                        break
            out["bins"].append({"bin": binId, "orfs": len(recs), "models_of_bin": len(models_by_bin[binId]), "models_checked": len(sample), "rows_checked": len(want),
                                "rows_of_bin": len(got_all), "identical": same})
            out["models"] += len(sample)
            out["rows"] += len(want)
    out["launch_classes"] = len(classes_seen)
    if marker_sets is not None and qa_rows is not None:
        ok = True
        for (b, binId, sample, recs, dsq) in jobs:
            with open(os.path.join(out_dir, "bins", binId, table)) as f:
                text = f.read()
            sel = marker_sets[binId].selectedMarkerSet()
            _mh, gc = ro.reduce_bin(text, _omodels(models_by_bin[binId]), pfam_text or "", [sorted(s) for s in sel.markerSet])
            f = qa_rows[binId].split("\t")
            same = [int(x) for x in f[5:11]] == gc[:6] and f[11] == "%0.2f" % gc[6] and f[12] == "%0.2f" % gc[7]
            if not same:
                ok = False
                out["mismatches"].append({"bin": binId, "qa_row": f[5:13], "reduce_oracle": gc})
        out["qa_rows_identical"] = ok
        out["identical"] = out["identical"] and ok
    hs.close()
    out["note"] = ("written %s lines of the sampled models == oracle/p7oracle.c text; QA rows == oracle/reduce_oracle.py on the whole written table; "
                   "the scan oracle is a restatement of hmmsearch (parity unpinned: no HMMER here)" % table)
    return out
