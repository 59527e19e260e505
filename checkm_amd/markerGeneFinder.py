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
from checkm_amd.markerSets import MarkerSetParser, wanted_model

# (abs outDir, tableOut) -> dict(ctx, profiles, parts=[dict(seqs, hits, bins=[binId...])], where={binId: (part, b)},
#                                 owned=set(binId) scanned by THIS rank, world)
SCAN_CACHE = {}

PAIR_BUDGET = int(os.environ.get("CKM_FIND_PAIR_BUDGET", str(125 * 1000 * 1000)))   # (ORF, model) pairs per ckm_search call
RAMP = (0.125, 0.25, 0.5)            # shares of the budgets the first batches of a find() stop at (plan_batches)
RES_BUDGET = int(os.environ.get("CKM_FIND_RES_BUDGET", str(400 * 1000 * 1000)))      # residues (~ bytes of genes.faa) per call


# (abs path, mtime_ns, size) of an HMM file -> resident _lib.Profiles.  The reference's hmmsearch re-reads its model file for every
# bin (checkm/hmmer.py:70); here a database (checkm.hmm is ~200 MB of text) is parsed, configured and uploaded once per process.
PROFILE_CACHE = {}


def profiles_for(ctx, db):
    st = os.stat(db)
    # keyed by DEVICE, not context: the tables are read-only device memory, so the two contexts find() alternates between share one
    # resident copy (and one parse of the text)
    key = (os.path.abspath(db), st.st_mtime_ns, st.st_size, ctx.device)
    p = PROFILE_CACHE.get(key)
    if p is None or not p.h:
        for k in [k for k in PROFILE_CACHE if k[0] == key[0] and k[3] == key[3]]:      # the file changed: drop the stale copy unless a scan still uses it
            if not any(ent["profiles"] is PROFILE_CACHE[k] for ent in SCAN_CACHE.values()):
                PROFILE_CACHE.pop(k).close()
        p = _lib.Profiles(ctx, db)
        PROFILE_CACHE[key] = p
    return p


_RELEASERS = []


def _join_releasers():
    while _RELEASERS:
        _RELEASERS.pop().join()


def release_scan(outDir=None, final=False, background=False):
    """Free cached device objects (all, or those of one output directory).  Profile databases stay resident until everything is
    released (release_scan() without an argument).  Hit lists ResultsParser handed out lazily are filled in first (not at interpreter
    exit: final=True).  background=True frees the hits and sequences on a helper thread (hipFree waits for the device: 0.3 s for a
    thousand bins' scans before the block cache of round 4) -- the next release waits for it; find() does not."""
    _join_releasers()
    if SCAN_CACHE and not final:
        mod = sys.modules.get("checkm_amd.resultsParser")
        if mod is not None:
            mod.materialize_lazy_hits()
    pools, doomed = [], []
    for key in list(SCAN_CACHE):
        if outDir is None or key[0] == os.path.abspath(outDir):
            ent = SCAN_CACHE.pop(key)
            for part in ent["parts"]:
                doomed.append(part["hits"]); doomed.append(part["seqs"])
            if ent.get("pool") is not None and ent["pool"] not in pools:
                pools.append(ent["pool"])

    def free():
        for obj in doomed:
            obj.close()
    if background and doomed and outDir is not None and not final:
        import threading
        t = threading.Thread(target=free, daemon=True)
        t.start()
        _RELEASERS.append(t)
    else:
        free()
    for pool in pools:                            # the workers hold the resident scans of a multi-GPU find()
        try:
            if not final:
                pool.call("release", dict(outDir=outDir))
        except Exception:
            pass
    if outDir is None:
        for k in list(PROFILE_CACHE):
            PROFILE_CACHE.pop(k).close()


def plan_batches(sizes, nmodels, pair_budget=None, res_budget=None):
    """Cut the bins (given order) into contiguous batches of one ckm_search call each: the Forward/Backward workspace of a call grows
    with its (ORF, model) pairs, so a batch stops before PAIR_BUDGET pairs (ORFs estimated from the file size at ~320 bytes per
    record) or RES_BUDGET residues.  The first three batches stop at 1/8, 1/4 and 1/2 of the budgets: the device has nothing to do
    until the first batch's files are read (0.4 s at the start of a 1000-bin tree pass with full-size batches,
    profiles/r03y_timeline_cfg3_1000bins.txt), and lanes that start on batches of different size do not finish together.
    Returns a list of index lists."""
    pair_budget = PAIR_BUDGET if pair_budget is None else pair_budget
    res_budget = RES_BUDGET if res_budget is None else res_budget
    batches, cur, pairs, res = [], [], 0, 0
    for i, (sz, nm) in enumerate(zip(sizes, nmodels)):
        p = max(1, sz // 320) * max(1, nm)
        f = RAMP[len(batches)] if len(batches) < len(RAMP) else 1.0
        if cur and (pairs + p > pair_budget * f or res + sz > res_budget * f):
            batches.append(cur); cur, pairs, res = [], 0, 0
        cur.append(i); pairs += p; res += sz
    if cur:
        batches.append(cur)
    return batches


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
    """The class that calls genes for bins given as nucleotide FASTA: GENE_CALLER if set; 'device' (the library's own gene finder,
    checkm_amd/geneFinder.py: both translation tables of all bins in batched device calls) when CKM_GENE_CALLER=device or when no
    `prodigal` binary is on PATH; else checkm_amd.prodigal.ProdigalRunner (the mirror of checkm/prodigal.py:41-182 that runs the two
    translation tables of a bin side by side)."""
    if GENE_CALLER is not None:
        return GENE_CALLER
    want = os.environ.get("CKM_GENE_CALLER", "")
    if want == "device":
        return "device"
    if want != "prodigal" and shutil.which("prodigal") is None:
        # chosen implicitly: say so -- the device gene finder restates Prodigal 2.6.3's single-genome mode (parity unpinned: no prodigal was
        # available to pin it against), so completeness / contamination may differ from a run that calls genes with the prodigal binary
        logging.getLogger('timestamp').warning("No `prodigal` on PATH: genes are called by the library's own gene finder on the device (a restatement of "
                                               "Prodigal 2.6.3's single-genome mode, not validated against a prodigal binary; set CKM_GENE_CALLER=device to "
                                               "choose it explicitly, or install prodigal / pass called genes with -g).")
        return "device"
    if shutil.which("prodigal") is not None:
        from checkm_amd.prodigal import ProdigalRunner
        return ProdigalRunner
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
        binIds, faa, todo, copies = [], [], [], []
        for binFile in binFiles:
            binId = binIdFromFilename(binFile)
            binDir = os.path.join(outDir, 'bins', binId)
            makeSurePathExists(binDir)
            dst = os.path.join(binDir, DefaultValues.PRODIGAL_AA)
            if bCalledGenes:
                copies.append((binFile, dst))
            else:
                todo.append((binFile, binDir, binId, dst))
            binIds.append(binId)
            faa.append(dst)
        self._pending_copies, self._copy_pool = [], None
        read_from = list(faa)                    # the file each bin's scan reads
        if copies:
            # called genes (-g) are copied in, as the reference does per bin (markerGeneFinder.py:118-127) -- a thousand bins are a gigabyte
            # of protein text, and both passes of lineage_wf copy it.  Plain files are copied by host threads WHILE the scan reads the
            # source (same bytes); find() waits for the copies before it returns.  Compressed files are unpacked first, as before.
            from concurrent.futures import ThreadPoolExecutor

            def copy_in(job):
                src, dst = job
                if src.endswith('.gz'):
                    with gzip.open(src, 'rt') as fin, open(dst, 'w') as fout:
                        shutil.copyfileobj(fin, fout)
                else:
                    shutil.copyfile(src, dst)
            packed = [j for j in copies if j[0].endswith('.gz')]
            plain = [j for j in copies if not j[0].endswith('.gz')]
            if packed:
                with ThreadPoolExecutor(max_workers=min(8, len(packed))) as pool:
                    list(pool.map(copy_in, packed))
            if plain:
                self._copy_pool = ThreadPoolExecutor(max_workers=min(4, len(plain)))
                self._pending_copies = [self._copy_pool.submit(copy_in, j) for j in plain]
                src_of = {dst: src for src, dst in plain}
                read_from = [src_of.get(f, f) for f in faa]
        if todo:
            runner = gene_caller()
            missing = [t for t in todo if runner is None and not os.path.exists(t[3])]
            if missing:
                self.logger.error("No called genes for bin %s and no gene caller available: run with -g/--genes, provide %s, or install prodigal "
                                  "next to CheckM (gene calling is outside the accelerated path)." % (missing[0][2], missing[0][3]))
                sys.exit(1)
            if runner == "device":
                # every bin that still lacks its genes, both translation tables, in batched device calls (checkm/prodigal.py:72-133)
                from checkm_amd import geneFinder
                def called(t):      # ProdigalRunner.areORFsCalled (checkm/prodigal.py:155-164): genes.fna decides when nucleotide ORFs are asked for
                    f = os.path.join(t[1], DefaultValues.PRODIGAL_NT) if bNucORFs else t[3]
                    return os.path.exists(f) and os.stat(f).st_size != 0
                jobs = [(t[0], t[1]) for t in todo if not called(t)]
                # The gene finder is latency-bound (a workgroup per bin in its dynamic programs), the scan is VALU-bound: they share the
                # device well.  The bins' genes are called on a helper thread, sub-batch by sub-batch in bin order; the scan of a batch
                # waits for its bins' files (find(): scan()).  CKM_GENE_OVERLAP=0: call all genes first.
                import threading
                dst_of = {t[0]: t[3] for t in todo}
                self._gene_events = {dst_of[j[0]]: threading.Event() for j in jobs}
                self._gene_sizes = {dst_of[j[0]]: int(0.36 * (os.path.getsize(j[0]) if os.path.exists(j[0]) else 0) * (3.2 if j[0].endswith('.gz') else 1.0)) for j in jobs}
                self._gene_error = []
                events, errors = self._gene_events, self._gene_error

                def run_genes():
                    try:
                        geneFinder.call_bin_files(jobs, bNucORFs, logger=self.logger, on_bin_done=lambda f: events[dst_of[f]].set())
                    except BaseException as e:          # (reported by find() on the main thread)
                        errors.append(e)
                    finally:
                        for ev in events.values():
                            ev.set()
                if os.environ.get("CKM_GENE_OVERLAP", "1") == "0":
                    run_genes()
                    self._check_genes()
                else:
                    self._gene_thread = threading.Thread(target=run_genes, name="ckm-genes", daemon=True)
                    self._gene_thread.start()
            elif runner is not None:
                def call(t):
                    binFile, binDir, _binId, _dst = t
                    prodigal = runner(binDir)
                    if not prodigal.areORFsCalled(bNucORFs):
                        prodigal.run(binFile, bNucORFs)
                    return prodigal.aaGeneFile
                from concurrent.futures import ThreadPoolExecutor
                # (the mirror runs the two translation tables of a bin side by side: half as many bins at a time keeps `threads` processes busy)
                per_bin = 2 if getattr(runner, '__module__', '') == 'checkm_amd.prodigal' else 1
                with ThreadPoolExecutor(max_workers=max(1, int(self.totalThreads) // per_bin)) as pool:
                    called = list(pool.map(call, todo))
                for t, aa in zip(todo, called):
                    if os.path.abspath(aa) != os.path.abspath(t[3]):
                        shutil.copyfile(aa, t[3])
        return binIds, faa, read_from

    _gene_events, _gene_sizes, _gene_error, _gene_thread = {}, {}, [], None

    def _check_genes(self):
        """An error of the gene-calling helper ends the run the way the reference ends it (logger.error + sys.exit, checkm/prodigal.py:100-115)."""
        if self._gene_error:
            e = self._gene_error[0]
            self._fail(str(e) if isinstance(e, ValueError) else "gene calling failed: %r" % (e,))

    def _fail(self, msg):
        """logger.error + sys.exit(1), the reference's way out (checkm/hmmer.py:131-137) -- after telling the other ranks of a multi-GPU
        run, who would otherwise wait for this one in find()'s closing collective until it times out."""
        from checkm_amd import dist as cdist
        self.logger.error(msg)
        try:
            cdist.agree(False)
        except Exception:               # noqa: BLE001  (the process group itself may be what failed)
            pass
        sys.exit(1)

    def _wait_genes(self, path):
        ev = self._gene_events.get(path)
        if ev is not None:
            ev.wait()
            if self._gene_error:
                raise _lib.CkmError(-1, "gene calling failed")

    def find(self, binFiles, outDir, tableOut, hmmerOut, markerFile, bKeepAlignment, bNucORFs, bCalledGenes):
        """checkm/markerGeneFinder.py:45-96.  One process per GPU: under torchrun (WORLD_SIZE > 1) every rank calls find() with the
        same arguments, scans a size-balanced shard of the bins (checkm_amd/dist.py:shard_bins -- the reference's bin-level fan-out,
        markerGeneFinder.py:59-83) and writes the files of the bins it owns; the return value covers every bin on every rank.
        Within a rank the bins go through the device in batches: the FASTA files of batch k+1 are read, digitised and uploaded and the
        tables of batch k-1 are written by host threads while batch k is on the GPU."""
        from concurrent.futures import ThreadPoolExecutor
        from checkm_amd import dist as cdist
        from checkm_amd import workers
        # (a background release of an earlier scan -- release_scan(background=True) -- is NOT waited for here: freed blocks go to the
        #  library's block cache, no hipFree and no device-wide wait, so it may run underneath this scan's first batches; the next
        #  release, runtime.close() and bench.py's timed region wait for it.  Waiting here kept the device idle for 0.25-0.28 s at the
        #  start of every 1000-bin pass that follows another: profiles/r04x_timeline_cfg3.txt.)
        devs = workers.devices()
        if devs is not None:
            return self._find_with_workers(devs, binFiles, outDir, tableOut, hmmerOut, markerFile, bKeepAlignment, bNucORFs, bCalledGenes)
        try:
            ctx = runtime.get_ctx()
        except Exception as e:
            self.logger.error("No usable MI355X (gfx950) device for the marker-gene scan: %s" % e)
            sys.exit(1)
        rank, _local, world = cdist.env_rank()
        if world > 1:
            cdist.init_process_group()          # RCCL ('nccl') unless CKM_DIST_BACKEND says otherwise; no-op when the caller already did
        self.logger.info("Identifying marker genes in %d bins on device %d:" % (len(binFiles), ctx.device))
        from checkm_amd import parity
        self.logger.info(parity.log_line())
        parser = MarkerSetParser(self.totalThreads)
        db = parser.hmmDatabaseFor(markerFile)
        allIds = [binIdFromFilename(f) for f in binFiles]
        wanted = parser.markerAccessionsForBins(allIds, markerFile)
        try:
            profiles = profiles_for(ctx, db)
        except _lib.CkmError as e:
            self._fail('marker-gene scan failed: %s' % e)
        heads = profiles.headers
        models_of = model_lists(heads, allIds, wanted, self.logger, db)
        # ---- this rank's shard (weights = file size x models) ----
        mine = list(range(len(binFiles)))
        if world > 1:
            wts = []
            for f, b in zip(binFiles, allIds):
                try:
                    sz = os.path.getsize(f)
                except OSError:
                    sz = 0
                nm = len(models_of[b]) if models_of.get(b) is not None else profiles.n
                wts.append(sz * max(1, nm))
            mine = cdist.shard_bins(wts, world)[rank]
        myFiles = [binFiles[i] for i in mine]
        self._gene_events, self._gene_sizes, self._gene_error, self._gene_thread = {}, {}, [], None
        binIds, faa, read_from = self._geneFiles(myFiles, outDir, bNucORFs, bCalledGenes)
        # (from here to the end of the scan an exception of any kind must first wait for the gene-calling helper thread, which keeps issuing
        #  device calls: _scan_batches' finally)
        try:
            out, batches, parts, totals = self._scan_batches(ctx, profiles, db, heads, allIds, models_of, binIds, read_from, outDir, tableOut, hmmerOut, bKeepAlignment)
        except _lib.CkmError as e:
            self._check_genes()
            self._fail('marker-gene scan failed: %s' % e)
        self._check_genes()
        self._gene_events, self._gene_thread = {}, None
        where = {}
        for k, batch in enumerate(batches):
            for b, i in enumerate(batch):
                where[binIds[i]] = (k, b)
        key = (os.path.abspath(outDir), tableOut)
        release_key(key)
        SCAN_CACHE[key] = dict(ctx=ctx, profiles=profiles, parts=parts, where=where, owned=set(binIds), world=world,
                               all_bins=list(allIds), totals=totals)        # totals: stage counters summed over this rank's ckm_search calls
        if world > 1 and not cdist.agree(True):      # every rank's tables are on disk before anyone reads them -- or some rank has failed
            self.logger.error('marker-gene scan failed on another rank')
            sys.exit(1)
        if out is None:
            out = models_for_bins(heads, allIds, models_of)
        self.logger.info("    Finished processing %d of %d (100.00%%) bins." % (len(allIds), len(allIds)))
        return out

    def _scan_batches(self, ctx, profiles, db, heads, allIds, models_of, binIds, read_from, outDir, tableOut, hmmerOut, bKeepAlignment):
        """The scan of this rank's bins, batch by batch on two lanes; returns (find()'s return value if it was built beside the scan,
        batches, parts, totals).  Whatever ends it, the gene-calling helper thread has been joined and the pending copies are done."""
        from concurrent.futures import ThreadPoolExecutor
        try:
            return self._scan_batches_body(ctx, profiles, db, heads, allIds, models_of, binIds, read_from, outDir, tableOut, hmmerOut, bKeepAlignment)
        finally:
            if self._gene_thread is not None:
                self._gene_thread.join()
            for f in self._pending_copies:           # bins/<binId>/genes.faa is complete before find() returns
                f.result()
            if self._copy_pool is not None:
                self._copy_pool.shutdown()
            self._pending_copies, self._copy_pool = [], None

    def _scan_batches_body(self, ctx, profiles, db, heads, allIds, models_of, binIds, read_from, outDir, tableOut, hmmerOut, bKeepAlignment):
        from concurrent.futures import ThreadPoolExecutor
        # (a bin whose genes are still being called is planned with an estimate: proteins of a bacterial genome take about 0.36 of its bases)
        sizes = [self._gene_sizes[f] if f in self._gene_events else (os.path.getsize(f) if os.path.exists(f) else 0) for f in read_from]
        nmod = [len(models_of[b]) if models_of.get(b) is not None else profiles.n for b in binIds]
        batches = plan_batches(sizes, nmod)
        parts, where, totals = [None] * len(batches), {}, {}
        # Two batches in flight, one per context (a context runs one search at a time): while one batch's SSV launches fill the device,
        # the FASTA files of the other are read / digitised / uploaded, its tables written, and its chain tail and host part hidden.
        # CKM_FIND_PIPELINE=n for n lanes; measured on cfg3 (1000 bins, profiles/r03c_lanes.txt): 2 lanes 15.9 s, 3 lanes 17.2 s (find itself
        # 0.6 s faster, but releasing three contexts' scans costs 2 s), 4 lanes 16.5 s.
        lanes = [(ctx, profiles)]
        for k in range(1, min(len(batches), max(1, int(os.environ.get("CKM_FIND_PIPELINE", "2"))))):
            try:
                ck = runtime.get_ctx_k(k)
                lanes.append((ck, profiles_for(ck, db)))
            except _lib.CkmError:
                break
        import threading
        tot_lock = threading.Lock()
        # the float workspace of a batch is tens of GB the first time a context meets one (seconds of hipMalloc): start allocating it
        # now, beside the reading of the first batches' FASTA files (ckm_ctx_reserve; the search waits for it)
        fb_classes = (1, 2, 3, 4, 6, 8, 12, 16, 24, 32, 48, 64)        # rows of 64 floats per model node block (csrc/ckm_host.h: kFbQ)
        padded = [64 * next((q for q in fb_classes if 64 * q >= h["leng"]), 64) for h in heads]
        pos_cache = {}
        need = []                                  # per batch: (pairs, expected cells -- include/checkm_hip.h: ckm_ctx_reserve)
        cells_of = [(x + 64) * x for x in padded]
        for batch in batches:
            pairs = cells = 0
            for i in batch:
                m = models_of.get(binIds[i]) if models_of else None
                k = id(m)
                if k not in pos_cache:
                    pos_cache[k] = (len(m), sum(cells_of[x] for x in m)) if m is not None else (len(padded), sum(cells_of))
                pairs += max(1, sizes[i] // 320) * pos_cache[k][0]; cells += pos_cache[k][1]
            need.append((pairs, cells))
        for j, (c, _prof) in enumerate(lanes):
            mine = need[j:]                        # the largest batch this lane may meet: its first is batch j, after that whichever comes next
            if mine:
                try:
                    c.reserve(max(x[0] for x in mine), max(x[1] for x in mine))
                except _lib.CkmError:
                    pass

        # A lane takes the next batch when it is free (not every other one: with batches of unequal cost one lane used to run the last two
        # alone, profiles/r05F_emulated_rank_trace.txt).  The return value is built on this thread once every lane is inside its first
        # search: built at once, its 0.06 s of Python held the interpreter lock while the lanes tried to hand their first batches to the
        # library, and the device waited (60-70 ms at the start of each pass of that trace).
        import itertools
        next_batch = itertools.count()
        started = [threading.Event() for _ in lanes]

        def scan(k, lane):
            import time as _t
            c, prof = lanes[lane]
            batch = batches[k]
            for i in batch:
                self._wait_genes(read_from[i])                                    # (genes of this batch's bins still being called: find() overlaps the two)
            t0 = _t.perf_counter()
            seqs = _lib.Seqs.from_fasta(c, [read_from[i] for i in batch])         # read, digitized and packed by the library
            t1 = _t.perf_counter()
            bm = None if not models_of else [models_of[binIds[i]] if models_of[binIds[i]] is not None else list(range(profiles.n)) for i in batch]
            started[lane].set()
            hits = _lib.search(c, prof, seqs, bm, 0.1, 0.1)                       # -E 0.1 --domE 0.1, markerGeneFinder.py:141
            t2 = _t.perf_counter()
            st = c.stats()
            with tot_lock:
                totals["ingest_s"] = totals.get("ingest_s", 0.0) + (t1 - t0)
                totals["search_s"] = totals.get("search_s", 0.0) + (t2 - t1)
                for f in ("pairs_ssv", "pairs_msv_full", "pairs_bias", "pairs_vit", "pairs_vit_exact", "pairs_fwd", "pairs_dom", "envelopes", "regions_multi",
                          "cells_ssv", "residue_hmm", "ms_ssv", "ms_total", "ssv_launches", "cascade_fallback_lanes"):
                    totals[f] = totals.get(f, 0) + getattr(st, f)
                for f in ("ws_cap_bytes", "ws_used_bytes"):
                    totals[f] = max(totals.get(f, 0), getattr(st, f))
                totals["searches"] = totals.get("searches", 0) + 1
            part = dict(seqs=seqs, hits=hits, bins=[binIds[i] for i in batch], profiles=prof)
            if os.environ.get("CKM_TRACE") == "2":      # (same clock as the library's trace points)
                m0 = _t.monotonic() - _t.perf_counter()
                sys.stderr.write("find-trace lane %d batch %d ingest %.3f search %.3f .. %.3f\n" % (lane, k, 1e3 * (m0 + t0), 1e3 * (m0 + t1), 1e3 * (m0 + t2)))
            for b, i in enumerate(batch):
                hits.write_domtblout(prof, seqs, b, os.path.join(outDir, 'bins', binIds[i], tableOut))
                if bKeepAlignment:            # hmmsearch's -o text with the domain alignments (markerGeneFinder.py:138-142 drops --noali)
                    hits.write_alignments(c, prof, seqs, b, os.path.join(outDir, 'bins', binIds[i], hmmerOut))
            parts[k] = part
            if os.environ.get("CKM_TRACE") == "2":
                sys.stderr.write("find-trace lane %d batch %d written %.3f\n" % (lane, k, 1e3 * _t.monotonic()))
            with tot_lock:
                totals["write_s"] = totals.get("write_s", 0.0) + (_t.perf_counter() - t2)

        def lane_run(j):
            try:
                while True:
                    k = next(next_batch)
                    if k >= len(batches):
                        break
                    scan(k, j)
            finally:
                started[j].set()
        out = None
        if len(lanes) == 1:
            lane_run(0)
        else:
            with ThreadPoolExecutor(max_workers=len(lanes)) as ex:
                futs = [ex.submit(lane_run, j) for j in range(len(lanes))]
                for ev in started:
                    ev.wait(30.0)
                out = models_for_bins(heads, allIds, models_of)          # (the return value is built while the lanes scan: 0.06 s per pass of a 1000-bin run)
                for f in futs:
                    f.result()
        return out, batches, parts, totals

    def _find_with_workers(self, devs, binFiles, outDir, tableOut, hmmerOut, markerFile, bKeepAlignment, bNucORFs, bCalledGenes):
        """find() on a multi-GPU node: one worker process per device (checkm_amd/workers.py), spawned here as the reference's find()
        forks its own workers (checkm/markerGeneFinder.py:59-83); this process keeps everything outside the path to itself."""
        from checkm_amd import workers
        try:
            pool = workers.get_pool(devs)
            replies = pool.call("find", dict(binFiles=list(binFiles), outDir=outDir, tableOut=tableOut, hmmerOut=hmmerOut, markerFile=markerFile,
                                             bKeepAlignment=bKeepAlignment, bNucORFs=bNucORFs, bCalledGenes=bCalledGenes, threads=self.totalThreads))
        except workers.WorkerError as e:
            self.logger.error('marker-gene scan failed: %s' % e)
            sys.exit(1)
        self.logger.info("Identified marker genes in %d bins on %d devices." % (len(binFiles), len(devs)))
        heads = replies[0]["heads"]
        allIds = [binIdFromFilename(f) for f in binFiles]
        msp = MarkerSetParser(self.totalThreads)
        wanted = msp.markerAccessionsForBins(allIds, markerFile)
        models_of = model_lists(heads, allIds, wanted, None, msp.hmmDatabaseFor(markerFile))      # (the workers already warned about models left out)
        totals = {}
        for r in replies:
            for k, v in r["totals"].items():
                totals[k] = max(totals.get(k, 0), v) if k in ("ws_cap_bytes", "ws_used_bytes") else totals.get(k, 0) + v
        key = (os.path.abspath(outDir), tableOut)
        release_key(key)
        SCAN_CACHE[key] = dict(pool=pool, world=len(devs), owned=set(), parts=[], where={}, all_bins=list(allIds), totals=totals,
                               owners={b: r for r, rep in enumerate(replies) for b in rep["owned"]}, profiles=None, ctx=None)
        return models_for_bins(heads, allIds, models_of)


def model_lists(heads, allIds, wanted, logger=None, db=''):
    """{binId: indices into the profile database of the models the bin is scanned against} ({} when every bin takes every model); bins
    that ask for the same accession set share one list object."""
    models_of = {}
    if any(w is not None for w in wanted.values()):
        cache = {}
        for b in allIds:
            w = wanted[b]
            key = id(w)
            if key not in cache:
                cache[key] = None if w is None else [i for i, h in enumerate(heads) if wanted_model(h["name"], h["acc"], w)]
            models_of[b] = cache[key]
    too_long = [i for i, h in enumerate(heads) if not h.get("searchable", True)]
    if too_long:
        # the database holds models beyond the kernels' 4096 nodes: they stay out of every bin's scan, loudly (hmmsearch would search them)
        hit = sorted(set(too_long) & set(i for m in models_of.values() if m is not None for i in m)) if (models_of and all(m is not None for m in models_of.values())) else too_long
        if hit:
            if logger is not None:
                logger.warning("%d model(s) of %s are longer than the 4096 nodes the kernels are instantiated for and are NOT searched: %s" %
                               (len(hit), db, ", ".join("%s (LENG %d)" % (heads[i]["name"], heads[i]["leng"]) for i in hit[:8])))
            keep = [i for i in range(len(heads)) if heads[i].get("searchable", True)]
            drop, cache = set(hit), {}
            for b in allIds:
                m = models_of.get(b)
                if m is None:
                    models_of[b] = keep
                else:
                    if id(m) not in cache:
                        cache[id(m)] = [i for i in m if i not in drop]
                    models_of[b] = cache[id(m)]
    return models_of


def models_for_bins(heads, allIds, models_of):
    """find()'s return value, {binId: {acc: HmmModel}}: the (sticky) header view of each bin's model subset; bins with the same subset
    share one dict."""
    out = {}
    subset_cache = {}
    for b in allIds:
        m = models_of.get(b) if models_of else None
        if m is None:
            if None not in subset_cache:
                subset_cache[None] = models_dict(heads)
            out[b] = subset_cache[None]
        else:
            k2 = id(m)
            if k2 not in subset_cache:
                subset_cache[k2] = models_dict([heads[i] for i in m])
            out[b] = subset_cache[k2]
    return out


def release_key(key):
    """Drop one cached scan (hit lists handed out lazily are filled in first)."""
    if key not in SCAN_CACHE:
        return
    mod = sys.modules.get("checkm_amd.resultsParser")
    if mod is not None:
        mod.materialize_lazy_hits()       # hit lists handed out lazily still point into the scan that is replaced here
    old = SCAN_CACHE.pop(key)
    for part in old["parts"]:
        part["hits"].close(); part["seqs"].close()
