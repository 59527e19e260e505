"""Gene calling on the MI355X: what CheckM runs in front of the marker-gene scan when bins arrive as nucleotides.

The reference calls `prodigal -p single|meta -q -m -f gff -g <table> -a genes.faa -i <bin>` twice per bin, tables 11 and 4
(checkm/prodigal.py:74,86-93,131-133), and keeps table 4 when it raises the coding density enough.  Prodigal is a third-party C
program that is in neither /root/reference nor this image; what it does falls into (1) a byte scan that finds the start / stop NODES of
all six frames, (2) a per-genome training pass, (3) scoring and a dynamic program over the nodes, (4) gene records and translations.
This module is the Python face of all four on the device: `OrfNodes` (1, libcheckm_hip's ckm_orf_scan) and `call_bins` /
`call_bin_files` (1-4 for a batch of bins, ckm_genes_call: the nodes stay on the device from the codon flags to the gene records, the
host only takes the logarithms of the training tables), writing genes.faa / genes.gff in prodigal's layout (ckm_genes_write_bin) and
applying the reference's choice between the tables.  The pipeline is latency-bound (a workgroup per bin in the dynamic programs, a thread
per contig in the trace-back walks), so `call_bin_files` keeps several calls in flight: sub-batches of bins x both tables on CKM_GENE_LANES
host threads, each call on a stream of its own.  The single-genome mode only: `-p meta` (pre-trained models, used by CheckM below 100 kb) is
not built -- bins of 20-100 kb are trained on themselves here, smaller ones are refused.  The oracle is oracle/gene_full.c (parity
unpinned: a restatement of Prodigal 2.6.3 from memory).  There is no CPU implementation here: without a gfx950 device the library raises."""
from checkm_amd import _lib, runtime

ATG, GTG, TTG, STOP = 0, 1, 2, 3


def read_contigs_bytes(fastaFile):
    """[(id bytes, sequence bytes)] of a nucleotide FASTA file (plain or gzip), ids cut at the first whitespace as the reference's
    readFasta does (checkm/util/seqUtils.py:180-211).  Whole-file byte operations: a 2 Mb bin takes a millisecond or two."""
    import gzip
    opener = gzip.open if fastaFile.endswith('.gz') else open
    with opener(fastaFile, 'rb') as f:
        data = f.read()
    out = []
    # a record begins with '>' at the START OF A LINE (readFasta looks at line[0]); whatever precedes the first one is skipped
    if data.startswith(b'>'):
        start = 0
    else:
        start = data.find(b'\n>') + 1
        if start <= 0:
            return out
    for rec in data[start + 1:].split(b'\n>'):
        head, _nl, body = rec.partition(b'\n')
        hs = head.split(None, 1)
        out.append((hs[0] if hs else b'', body.translate(None, b'\n\r \t')))
    return out


def read_contigs(fastaFile):
    """[(id, sequence)] as str (tests, diagnostics); the batch path uses read_contigs_bytes."""
    return [(c.decode(), s.decode()) for c, s in read_contigs_bytes(fastaFile)]


class OrfNodes(object):
    """Start / stop nodes of a bin's contigs for one translation table: numpy columns contig, ndx, stop_val, type, strand_rev, edge."""

    def __init__(self, contigs, transTable=11, closedEnds=False):
        cols, self.stats = _lib.orf_nodes(runtime.get_ctx(), contigs, transTable, closedEnds)
        for k, v in cols.items():
            setattr(self, k, v)
        self.n = len(self.ndx)

    def orfs(self, minLength=90):
        """(contig, start, stop, strand) of every start node whose ORF is closed by a real stop and spans at least minLength bases:
        the candidates the gene finder's dynamic program chooses among."""
        sel = (self.type != STOP) & (self.edge == 0) & (abs(self.stop_val - self.ndx) + 3 >= minLength)
        return list(zip(self.contig[sel].tolist(), self.ndx[sel].tolist(), self.stop_val[sel].tolist(), (1 - 2 * self.strand_rev[sel].astype(int)).tolist()))


# ---- the gene finder's body on the device (round 4): genes.faa / genes.gff without an external binary ---------------------------------
SD_MOTIF = ["None", "GGA/GAG/AGG", "3Base/5BMM", "4Base/6BMM", "AGxAG", "AGxAG", "GGA/GAG/AGG", "GGxGG", "GGxGG", "AGxAG", "AGGAG(G)/GGAGG",
            "AGGA/GGAG/GAGG", "AGGA/GGAG/GAGG", "GGA/GAG/AGG", "GGxGG", "AGGA", "GGAG/GAGG", "AGxAGG/AGGxGG", "AGxAGG/AGGxGG", "AGxAGG/AGGxGG",
            "AGGAG/GGAGG", "AGGAG", "AGGAG", "GGAGG", "GGAGG", "AGGAGG", "AGGAGG", "AGGAGG"]
SD_SPACER = ["None", "3-4bp", "13-15bp", "13-15bp", "11-12bp", "3-4bp", "11-12bp", "11-12bp", "3-4bp", "5-10bp", "13-15bp", "3-4bp", "11-12bp", "5-10bp",
             "5-10bp", "5-10bp", "5-10bp", "11-12bp", "3-4bp", "5-10bp", "11-12bp", "3-4bp", "5-10bp", "3-4bp", "5-10bp", "11-12bp", "3-4bp", "5-10bp"]
START_TYPE = ["ATG", "GTG", "TTG", "Edge"]
MIN_SINGLE_GENOME = 20000          # prodigal refuses to train on less; CheckM itself switches to `-p meta` below 100 kb (checkm/prodigal.py:80-83)


def _motif_text(length, ndx):
    return "".join("ACGT"[(ndx >> (2 * i)) & 3] for i in range(length))


class BinGenes(object):
    """Genes of one bin for one translation table, as the device returned them, with prodigal's text outputs."""

    def __init__(self, contigs, table, cols, sel, trained, uses_sd, gc):
        self.contigs, self.table, self.trained, self.uses_sd, self.gc = contigs, table, bool(trained), int(uses_sd), float(gc)
        import numpy as np
        if isinstance(sel, slice):                     # (a bin's genes are one run of the batch's columns)
            self.cols = {f: cols[f][sel] for f in cols}
            self.n = sel.stop - sel.start
        else:
            sel = np.asarray(sel, dtype=np.int64)
            self.cols = {f: (cols[f][sel] if f != "proteins" else [cols[f][k] for k in sel]) for f in cols}
            self.n = len(sel)

    @property
    def rows(self):
        """One dict per gene (tests; the writers below use the columns)."""
        keys = list(self.cols)
        return [{f: (self.cols[f][k].item() if hasattr(self.cols[f][k], "item") else self.cols[f][k]) for f in keys} for k in range(self.n)]

    def coding_bases(self):
        """Bases covered by genes, overlaps counted once: what ProdigalGeneFeatureParser.codingBases sums over the contigs
        (checkm/prodigal.py:246-274)."""
        import numpy as np
        if self.n == 0:
            return 0
        c, s, e = self.cols["contig"].astype(np.int64), self.cols["begin"].astype(np.int64) - 1, self.cols["end"].astype(np.int64)
        order = np.lexsort((s, c))
        c, s, e = c[order], s[order], e[order]
        total = 0
        for ci in np.unique(c):
            m = c == ci
            ss, ee = s[m], e[m]
            prev = np.concatenate(([-1], np.maximum.accumulate(ee)[:-1]))
            total += int(np.maximum(0, ee - np.maximum(ss, prev)).sum())
        return total

    def write(self, aaFile, gffFile, ntFile=None):
        """genes.faa / genes.gff (/ genes.fna) in prodigal's layout: `>contig_n # begin # end # strand # attributes`, GFF3 CDS lines."""
        comp = bytes.maketrans(b"ACGTRYKMSWBDHVNacgtrykmswbdhvn", b"TGCAYRMKSWVHDBNtgcayrmkswvhdbn")      # (IUPAC codes complement too)
        L = {f: (v.tolist() if hasattr(v, "tolist") else v) for f, v in self.cols.items()}      # plain lists: one conversion per column, not one per field of every gene
        per = {}
        for k, c in enumerate(L["contig"]):
            per.setdefault(c, []).append(k)
        begin, end, strand_c, score, conf = L["begin"], L["end"], L["strand"], L["score"], L["conf"]
        cs, ss, rs, us, ts, gcc = L["cscore"], L["sscore"], L["rscore"], L["uscore"], L["tscore"], L["gc_cont"]
        rbs_bin, mot_len, mot_ndx, mot_sp, pl, pr, stt, prot = L["rbs_bin"], L["mot_len"], L["mot_ndx"], L["mot_spacer"], L["partial_left"], L["partial_right"], L["start_type"], L["proteins"]
        aa, gf, nt_out = [], ["##gff-version  3\n"], []
        for ci, (cid, seq) in enumerate(self.contigs):
            gf.append('# Sequence Data: seqnum=%d;seqlen=%d;seqhdr="%s"\n' % (ci + 1, len(seq), cid))
            gf.append('# Model Data: version=checkm_amd.device.gene_caller;run_type=Single;model="Ab initio";gc_cont=%.2f;transl_table=%d;uses_sd=%d\n'
                      % (100.0 * self.gc, self.table, self.uses_sd))
            for k, g in enumerate(per.get(ci, []), 1):
                if rbs_bin[g] >= 0:
                    motif, spacer = SD_MOTIF[rbs_bin[g]], SD_SPACER[rbs_bin[g]]
                elif mot_len[g] > 0:
                    motif, spacer = _motif_text(mot_len[g], mot_ndx[g]), "%dbp" % mot_sp[g]
                else:
                    motif, spacer = "None", "None"
                at = "ID=%d_%d;partial=%d%d;start_type=%s;rbs_motif=%s;rbs_spacer=%s;gc_cont=%.3f" % (
                    ci + 1, k, pl[g], pr[g], START_TYPE[stt[g]], motif, spacer, gcc[g])
                gf.append("%s\tcheckm_amd_device\tCDS\t%d\t%d\t%.1f\t%s\t0\t%s;conf=%.2f;score=%.2f;cscore=%.2f;sscore=%.2f;rscore=%.2f;uscore=%.2f;tscore=%.2f;\n"
                          % (cid, begin[g], end[g], score[g], "+" if strand_c[g] == 1 else "-", at, conf[g], score[g], cs[g], ss[g], rs[g], us[g], ts[g]))
                head = ">%s_%d # %d # %d # %d # %s\n" % (cid, k, begin[g], end[g], strand_c[g], at)
                p = prot[g]
                aa.append(head)
                aa.extend(p[i:i + 60] + "\n" for i in range(0, len(p), 60))
                if ntFile:
                    nt = seq[begin[g] - 1:end[g]]
                    if strand_c[g] != 1:
                        nt = nt.encode().translate(comp)[::-1].decode()
                    nt_out.append(head)
                    nt_out.extend(nt[i:i + 70] + "\n" for i in range(0, len(nt), 70))
        with open(aaFile, "w") as f:
            f.write("".join(aa))
        with open(gffFile, "w") as f:
            f.write("".join(gf))
        if ntFile:
            with open(ntFile, "w") as f:
                f.write("".join(nt_out))


def call_bins(bins, table, mask=True, ctx=None):
    """Genes of many bins for one translation table in ONE device call (ckm_genes_call): bins = [[(contig id, sequence), ...], ...].
    Returns a list of BinGenes (the columns copied out; diagnostics and tests -- call_bin_files keeps the results in the library)."""
    ctx = ctx if ctx is not None else runtime.get_ctx()
    cols, per_bin, stats = _lib.call_genes(ctx, [[s for _c, s in contigs] for contigs in bins], table, False, mask)
    import numpy as np
    at = np.searchsorted(cols["bin"], np.arange(len(bins) + 1))        # the records come in bin order
    out = [BinGenes(bins[b], table, cols, slice(int(at[b]), int(at[b + 1])), per_bin["trained"][b], per_bin["uses_sd"][b], per_bin["gc"][b]) for b in range(len(bins))]
    call_bins.last_stats = stats
    return out


def best_table(genes11, genes4, total_bases):
    """checkm/prodigal.py:117-133: table 4 only when its coding density beats table 11's by more than 0.05 and exceeds 0.7."""
    d11 = float(genes11.coding_bases()) / total_bases if total_bases else 0
    d4 = float(genes4.coding_bases()) / total_bases if total_bases else 0
    return (4 if (d4 - d11 > 0.05) and d4 > 0.7 else 11), {11: d11, 4: d4}


META_RANGE = 100000            # below this CheckM runs `prodigal -p meta` (checkm/prodigal.py:80-83)


def _lanes():
    import os
    # 12: with the files read and written by the library the pass is the device's, and past a dozen busy hardware queues (HIP runs with 16
    # here; more are worse) calls wait on each other (768 bins: 2.0 s with 8 - 16 lanes, 3.1 s with 24: profiles/r06z_gene_calls_in_flight.txt)
    return max(1, min(32, int(os.environ.get("CKM_GENE_LANES", "12"))))


def call_bin_files(jobs, bNucORFs=False, max_bases=None, logger=None, on_bin_done=None):
    """jobs = [(nucleotide FASTA of a bin, directory for genes.faa / genes.gff [/ genes.fna])].  Both translation tables per bin from the
    device, the reference's choice between them, prodigal's file layout.  Returns {binFile: (best table, {11: density, 4: density})}.
    The bins go through the device in sub-batches of <= max_bases (CKM_GENE_BATCH_MB, default 128 Mbase), several calls in flight.
    Raises ValueError when a bin is below the 20 kb the gene finder can train on (the pre-trained `-p meta` models CheckM would use
    below 100 kb are not built): BEFORE anything is written for files below 200 kB and for compressed files, whose bases are counted
    up front; a larger plain file is taken to hold enough and is counted when its sub-batch is read, so one that is mostly headers is
    refused then -- earlier sub-batches' files exist and on_bin_done has fired for them.  Raises after the other bins' files are written
    when a trained bin yields no genes (the reference treats empty prodigal output as a failure, checkm/prodigal.py:96-115).  on_bin_done(binFile) is called
    from a worker thread as soon as a bin's files are complete (MarkerGeneFinder.find scans the first bins while the last are called)."""
    import os
    import threading
    import time
    from concurrent.futures import ThreadPoolExecutor
    from checkm_amd.defaultValues import DefaultValues
    if max_bases is None:
        max_bases = int(os.environ.get("CKM_GENE_BATCH_MB", "128")) << 20
    lanes = _lanes()
    native_read = os.environ.get("CKM_GENE_NATIVE_READ", "1") != "0"
    out, lock = {}, threading.Lock()
    if not jobs:
        return out
    phases = {"read_s": 0.0, "device_calls_s": 0.0, "choose_and_write_s": 0.0, "lanes": lanes, "calls": 0, "wall_s": 0.0, "device_busy_s": 0.0}
    in_flight = [0, 0.0]                                    # device calls running now, and since when at least one is
    call_bin_files.last_phases = phases
    t_wall = time.perf_counter()
    # ---- sizes first, text later: a sub-batch's files are read when its turn comes (reading 1000 bins up front kept the device idle for 2.5 s).
    # Small and compressed files are read now -- their base counts decide refusals and warnings, which come BEFORE anything is written; a
    # plain file of 200 kB or more is taken to hold at least the 100 kb below which CheckM switches to `-p meta` (a file that size with fewer
    # than 20 kb of bases -- thousands of headers around a few bases each -- is still refused, when its sub-batch is read).
    t0 = time.perf_counter()
    fsize = [os.path.getsize(j[0]) if os.path.exists(j[0]) else 0 for j in jobs]
    early = [k for k in range(len(jobs)) if fsize[k] < 200000 or jobs[k][0].endswith('.gz')]
    contigs_of = [None] * len(jobs)
    if early:
        with ThreadPoolExecutor(max_workers=min(8, len(early))) as ex:
            for k, c in zip(early, ex.map(lambda k: read_contigs_bytes(jobs[k][0]), early)):
                contigs_of[k] = c
    totals = [sum(len(s) for _c, s in contigs_of[k]) if contigs_of[k] is not None else fsize[k] * 70 // 71 for k in range(len(jobs))]
    phases["read_s"] = time.perf_counter() - t0

    def refuse_small(ks):
        small = [(jobs[k][0], totals[k]) for k in ks if totals[k] < MIN_SINGLE_GENOME]
        if small:
            raise ValueError("bin %s holds %d bases%s: the device gene caller trains on the bin itself and needs %d (the pre-trained models of "
                             "`prodigal -p meta`, which CheckM uses below 100 kb, are not built); provide called genes (-g) or a prodigal binary"
                             % (small[0][0], small[0][1], " (and %d more such bins)" % (len(small) - 1) if len(small) > 1 else "", MIN_SINGLE_GENOME))

    def warn_meta(ks):
        if logger is not None:
            for k in ks:
                if totals[k] < META_RANGE:
                    logger.warning("Bin %s holds %d bases: its genes are called with a model trained on the bin itself (prodigal -p single); "
                                   "CheckM runs `prodigal -p meta` below %d bases (checkm/prodigal.py:80-83), so its gene set may differ."
                                   % (jobs[k][0], totals[k], META_RANGE))
    refuse_small(early)
    warn_meta(early)
    # ---- sub-batches ----
    batches, cur, size = [], [], 0
    for k in range(len(jobs)):
        if cur and size + totals[k] > max_bases:
            batches.append(cur); cur, size = [], 0
        cur.append(k); size += totals[k]
    if cur:
        batches.append(cur)
    ctx = runtime.get_ctx()
    empty, stats_last = [], {}
    ahead = threading.Semaphore(max(4, lanes))              # sub-batches read but not yet finished (bounds the text held in memory)

    failed = []                                             # the first error; once set, nothing new starts and the permits still come back

    def record(e):
        with lock:
            failed.append(e)

    def prepare(ks):
        ahead.acquire()
        try:
            if failed:
                raise RuntimeError("an earlier sub-batch failed")
            t1 = time.perf_counter()
            late = [k for k in ks if contigs_of[k] is None]
            if native_read and len(late) == len(ks) and all(os.path.isfile(jobs[k][0]) for k in ks):
                # every file of the sub-batch is a plain one not read yet: the library reads them, a file per host thread, straight into
                # the layout of the call (the Python path below holds the interpreter lock for four passes over the text: 5.7 ms per 2 Mb
                # bin, 0.37 s before the first call of a pass could start -- profiles/r06z_gene_pass_timeline_256bins.txt)
                batch = _lib.GeneBatch.from_files([jobs[k][0] for k in ks])
                for b, k in enumerate(ks):
                    totals[k] = batch.bases[b]
                refuse_small(late)
                warn_meta(late)
            else:
                for k in late:
                    contigs_of[k] = read_contigs_bytes(jobs[k][0])
                    totals[k] = sum(len(s) for _c, s in contigs_of[k])
                refuse_small(late)
                warn_meta(late)
                batch = _lib.GeneBatch([contigs_of[k] for k in ks])
                for k in ks:
                    contigs_of[k] = None                                    # (the batch holds the text now)
            with lock:
                phases["read_s"] += time.perf_counter() - t1
            return batch
        except BaseException:
            ahead.release()                                                 # (no call will run for this sub-batch)
            raise

    def finish(ks, calls):
        t1 = time.perf_counter()
        c11, c4 = calls[11], calls[4]
        u11, u4, n11, n4 = c11.coding_union(), c4.coding_union(), c11.genes_per_bin(), c4.genes_per_bin()

        def one(bk):
            b, k = bk
            binFile, binDir = jobs[k]
            total = totals[k]
            d11 = float(u11[b]) / total if total else 0
            d4 = float(u4[b]) / total if total else 0
            best = 4 if (d4 - d11 > 0.05) and d4 > 0.7 else 11
            call = c11 if best == 11 else c4
            call.write_bin(b, os.path.join(binDir, DefaultValues.PRODIGAL_AA), os.path.join(binDir, DefaultValues.PRODIGAL_GFF),
                           os.path.join(binDir, DefaultValues.PRODIGAL_NT) if bNucORFs else None)
            with lock:
                out[binFile] = (best, {11: d11, 4: d4})
                if (n11 if best == 11 else n4)[b] == 0:
                    empty.append(binFile)
            if on_bin_done is not None:
                on_bin_done(binFile)
        # the library writes a bin's files without the interpreter lock: a sub-batch's bins go to the writer threads side by side (one
        # after the other they were the tail of a pass -- 0.18 s for the 64 bins of the last sub-batch)
        for _ in writers.map(one, list(enumerate(ks))):
            pass
        with lock:
            phases["choose_and_write_s"] += time.perf_counter() - t1

    def run_table(ks, prep, calls, table):
        """One (sub-batch, table) call.  Whichever of a sub-batch's two calls ends last -- well or not -- picks the tables and writes the
        files if both exist, frees both calls and hands the sub-batch's permit back: an error never strands a permit or a device result."""
        try:
            batch = prep.result()
        except BaseException as e:                                          # (prepare gave the permit back itself)
            record(e)
            return
        call = None
        try:
            if not failed:
                t1 = time.perf_counter()
                with lock:
                    if in_flight[0] == 0:
                        in_flight[1] = t1
                    in_flight[0] += 1
                try:
                    call = _lib.GeneCall(ctx, batch, table, False, True)
                finally:
                    t2 = time.perf_counter()
                    with lock:
                        in_flight[0] -= 1
                        if in_flight[0] == 0:
                            phases["device_busy_s"] += t2 - in_flight[1]        # (the pass's time with at least one call in flight)
                        phases["device_calls_s"] += t2 - t1; phases["calls"] += 1
                with lock:
                    stats_last.update(call.stats)
        except BaseException as e:
            record(e)
        with lock:
            calls[table] = call
            last = len(calls) == 2
        if last:
            try:
                if calls[11] is not None and calls[4] is not None and not failed:
                    finish(ks, calls)
            except BaseException as e:
                record(e)
            finally:
                for c in calls.values():
                    if c is not None:
                        c.close()
                ahead.release()

    with ThreadPoolExecutor(max_workers=lanes) as pool, ThreadPoolExecutor(max_workers=4) as readers, ThreadPoolExecutor(max_workers=8) as writers:
        futs = []
        for ks in batches:
            prep = readers.submit(prepare, ks)
            calls = {}
            for table in (4, 11):                                           # (table 4 first: TGA is no stop there, its calls hold more nodes and take longer)
                futs.append(pool.submit(run_table, ks, prep, calls, table))
        for f in futs:
            f.result()                                                      # (run_table records errors instead of raising: every permit comes back)
    if failed:
        first = [e for e in failed if not (isinstance(e, RuntimeError) and str(e) == "an earlier sub-batch failed")]
        raise (first or failed)[0]
    call_bins.last_stats = stats_last
    phases["wall_s"] = time.perf_counter() - t_wall
    if empty:
        raise ValueError("the device gene caller found no genes in bin %s%s (the reference treats empty prodigal output as a failure, checkm/prodigal.py:96-115)"
                         % (empty[0], " and %d more" % (len(empty) - 1) if len(empty) > 1 else ""))
    return out
