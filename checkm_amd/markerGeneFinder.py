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


class MarkerGeneFinder(object):
    """Identify marker genes within binned sequences (GPU scan of called genes)."""

    def __init__(self, threads):
        self.logger = logging.getLogger('timestamp')
        self.totalThreads = threads

    def find(self, binFiles, outDir, tableOut, hmmerOut, markerFile, bKeepAlignment, bNucORFs, bCalledGenes):
        try:
            runtime.get_ctx()
        except Exception as e:
            self.logger.error("No usable MI355X (gfx950) device for the marker-gene scan: %s" % e)
            sys.exit(1)
        self.logger.info("Identifying marker genes in %d bins on device %d:" % (len(binFiles), runtime.get_ctx().device))
        binIds, faa = [], []
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
            elif not os.path.exists(dst):
                # gene calling (prodigal, checkm/prodigal.py:54-153) sits BEFORE the accelerated path
                self.logger.error("No called genes for bin %s: run with -g/--genes or provide %s (gene calling is outside this path)." % (binId, dst))
                sys.exit(1)
            binIds.append(binId)
            faa.append(dst)
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
