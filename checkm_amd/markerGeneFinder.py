"""MarkerGeneFinder: same signature and side effects as checkm/markerGeneFinder.py:41-96, but the
per-bin `hmmsearch` processes (markerGeneFinder.py:134-142) are ONE batched scan on the MI355X.

Contract kept: bins/<binId>/genes.faa and bins/<binId>/<tableOut> (domtblout text) exist afterwards
for every bin; the return value is {binId: {acc: HmmModel}} with picklable HmmModel objects.
The packed hits stay resident in this process so ResultsParser can skip the text round-trip.
"""
import gzip
import logging
import os
import shutil
import sys

from checkm_amd import _lib, runtime
from checkm_amd.common import binIdFromFilename, makeSurePathExists
from checkm_amd.defaultValues import DefaultValues
from checkm_amd.hmmerModelParser import models_dict
from checkm_amd.markerSets import MarkerSetParser

# (abs outDir, tableOut) -> dict(ctx, profiles, seqs, hits, bin_index {binId: b}, headers)
SCAN_CACHE = {}


def release_scan(outDir=None):
    """Free cached device objects (all, or those of one output directory)."""
    for key in list(SCAN_CACHE):
        if outDir is None or key[0] == os.path.abspath(outDir):
            ent = SCAN_CACHE.pop(key)
            ent["hits"].close(); ent["seqs"].close(); ent["profiles"].close()


def scan_files(hmm_file, fasta_files, table_files, E=0.1, domE=0.1, bin_models=None, keep=None):
    """Scan protein FASTA files (one bin each) against hmm_file; write one domtblout per bin."""
    ctx = runtime.get_ctx()
    profiles = _lib.Profiles(ctx, hmm_file)
    seqs = _lib.Seqs.from_fasta(ctx, list(fasta_files))
    hits = _lib.search(ctx, profiles, seqs, bin_models, E, domE)
    for b, path in enumerate(table_files):
        hits.write_domtblout(profiles, seqs, b, path)
    if keep is not None:
        keep.update(ctx=ctx, profiles=profiles, seqs=seqs, hits=hits)
    else:
        hits.close(); seqs.close(); profiles.close()


GENE_CALLER = None      # a ProdigalRunner-like class (outDir) -> .areORFsCalled(bNucORFs) / .run(binFile, bNucORFs) / .aaGeneFile; None = look for CheckM's


def gene_caller():
    """The class that calls genes for bins given as nucleotide FASTA: GENE_CALLER if set, else the reference's ProdigalRunner when
    CheckM is importable (the drop-in case), else None."""
    if GENE_CALLER is not None:
        return GENE_CALLER
    try:
        from checkm.prodigal import ProdigalRunner
        return ProdigalRunner
    except Exception:
        return None


class MarkerGeneFinder(object):
    """Identify marker genes within binned sequences (GPU scan of called genes)."""

    def __init__(self, threads):
        self.logger = logging.getLogger('timestamp')
        self.totalThreads = threads

    def _geneFiles(self, binFiles, outDir, bNucORFs, bCalledGenes):
        """bins/<binId>/genes.faa for every bin (checkm/markerGeneFinder.py:108-127).  Called genes (-g) are copied in; otherwise
        genes already present are reused and the rest are called by the reference's own ProdigalRunner (checkm/prodigal.py:54-153),
        `threads` bins at a time -- gene calling sits BEFORE the accelerated path and is not reimplemented here."""
        binIds, faa, todo = [], [], []
        for binFile in binFiles:
            binId = binIdFromFilename(binFile)
            binDir = os.path.join(outDir, 'bins', binId)
            makeSurePathExists(binDir)
            dst = os.path.join(binDir, DefaultValues.PRODIGAL_AA)
            if bCalledGenes:
                if binFile.endswith('.gz'):
                    with gzip.open(binFile, 'rt') as fin, open(dst, 'w') as fout:
                        shutil.copyfileobj(fin, fout)
                else:
                    shutil.copyfile(binFile, dst)
            else:
                todo.append((binFile, binDir, binId, dst))
            binIds.append(binId)
            faa.append(dst)
        if todo:
            runner = gene_caller()
            missing = [t for t in todo if runner is None and not os.path.exists(t[3])]
            if missing:
                self.logger.error("No called genes for bin %s and no gene caller available: run with -g/--genes, provide %s, or install prodigal "
                                  "next to CheckM (gene calling is outside the accelerated path)." % (missing[0][2], missing[0][3]))
                sys.exit(1)
            if runner is not None:
                def call(t):
                    binFile, binDir, _binId, _dst = t
                    prodigal = runner(binDir)
                    if not prodigal.areORFsCalled(bNucORFs):
                        prodigal.run(binFile, bNucORFs)
                    return prodigal.aaGeneFile
                from concurrent.futures import ThreadPoolExecutor
                with ThreadPoolExecutor(max_workers=max(1, int(self.totalThreads))) as pool:
                    called = list(pool.map(call, todo))
                for t, aa in zip(todo, called):
                    if os.path.abspath(aa) != os.path.abspath(t[3]):
                        shutil.copyfile(aa, t[3])
        return binIds, faa

    def find(self, binFiles, outDir, tableOut, hmmerOut, markerFile, bKeepAlignment, bNucORFs, bCalledGenes):
        try:
            runtime.get_ctx()
        except Exception as e:
            self.logger.error("No usable MI355X (gfx950) device for the marker-gene scan: %s" % e)
            sys.exit(1)
        self.logger.info("Identifying marker genes in %d bins on device %d:" % (len(binFiles), runtime.get_ctx().device))
        binIds, faa = self._geneFiles(binFiles, outDir, bNucORFs, bCalledGenes)
        parser = MarkerSetParser(self.totalThreads)
        db = parser.hmmDatabaseFor(markerFile)
        wanted = parser.markerAccessionsForBins(binIds, markerFile)
        keep = {}
        ctx = runtime.get_ctx()
        try:
            profiles = _lib.Profiles(ctx, db)
            acc_of = [h["acc"] if h["acc"] else h["name"] for h in profiles.headers]
            bin_models = None
            if any(w is not None for w in wanted.values()):
                bin_models = [[i for i, a in enumerate(acc_of) if wanted[b] is None or a in wanted[b]] for b in binIds]
            seqs = _lib.Seqs.from_fasta(ctx, faa)                                # read, digitized and packed by the library
            hits = _lib.search(ctx, profiles, seqs, bin_models, 0.1, 0.1)        # -E 0.1 --domE 0.1, markerGeneFinder.py:141
            for b, binId in enumerate(binIds):
                hits.write_domtblout(profiles, seqs, b, os.path.join(outDir, 'bins', binId, tableOut))
                if bKeepAlignment:
                    with open(os.path.join(outDir, 'bins', binId, hmmerOut), 'w') as f:
                        f.write("# alignments are not produced by the MI355X scan (--noali semantics)\n")
        except _lib.CkmError as e:
            self.logger.error('marker-gene scan failed: %s' % e)
            sys.exit(1)
        release_scan(outDir) if (os.path.abspath(outDir), tableOut) in SCAN_CACHE else None
        SCAN_CACHE[(os.path.abspath(outDir), tableOut)] = dict(ctx=ctx, profiles=profiles, seqs=seqs, hits=hits,
                                                                bin_index={b: i for i, b in enumerate(binIds)}, bin_models=bin_models)
        out = {}
        for b, binId in enumerate(binIds):
            sel = profiles.headers if bin_models is None else [profiles.headers[i] for i in bin_models[b]]
            out[binId] = models_dict(sel)
        self.logger.info("    Finished processing %d of %d (100.00%%) bins." % (len(binIds), len(binIds)))
        return out
