#!/usr/bin/env python
"""bench.py -- the marker-gene hot path of CheckM on N MI355X, one JSON line (rank 0).

  --config cfg3 (default)  configs[2] (N = 1) / configs[3] (N > 1, strong scaling) of BASELINE.json, the configuration "bins/hour
                           (lineage_wf-equiv)" is quoted on: see below.  One step = one lineage_wf-equivalent pass over ALL --bins-total
                           bins.  A step lasts tens of seconds, so the number of timed steps is min(--steps, what fits --budget-seconds
                           judged from the warm pass): the line carries `steps` (run) and `steps_requested`.  At N = 1 the same line
                           carries `cfg2` (a short run of the configuration below), `cpu_baseline` (the CPU oracle on sampled bins with
                           their lineage's model subsets) and `emulated_ranks_of_8` (see --emulate-rank).
  --config cfg2            configs[1] of BASELINE.json: cpr_43-shaped 43-profile DB against 100 synthetic 2 Mb bins (~2k ORFs
                           each) PER GPU.  One step = one pass of the hot path (scan + reduce + the one gather of QA rows) over the
                           rank's bins, inputs resident in HBM.  `--scaling strong --bins-total N` shards N bins over the ranks.
                           The same line carries `lineage_wf_equiv`: a small lineage_wf-shaped run (cfg3 inputs) from FILES through
                           MarkerGeneFinder.find -> ResultsParser (--lineage-bins 0 skips it).
  --config cfg3            configs[2]/[3]: 2000-profile marker DB, bins of U[1500,6000] ORFs, per-bin model subsets from a Lineage
                           marker file (43 phylo + 300-1500 lineage models), through the product's own call sequence
                           (find(phylo.hmm) -> find(lineage.ms) -> analyseResults -> printSummary) from genes.faa files; under
                           torchrun the PRODUCT shards the bins over the ranks (strong scaling, cfg4).  --bins-total (default 1000).

  --emulate-rank R/W       cfg3 on ONE GPU as rank R of W would run it: the product shards the bins as under torchrun with W ranks, this
                           process scans rank R's shard and does every piece of all-bins host work a rank does; no process group, no
                           collective (the table holds the shard's rows).  The only configs[3] evidence obtainable on one GPU.

  python bench.py --gpus 1 --steps 3 --warmup 1
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...
"""
import argparse
import json
import os
import shutil
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")     # before torch initialises HIP: see checkm_amd/__init__.py

import numpy as np  # noqa: E402

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s
PROFILE_TAG = "r0"             # profiles/<tag>*_ssv_traffic.json / <tag>*_cfg3_ssv_traffic.json hold the PMC-pass figures of the SSV launches
NOMINAL_CYCLES_PER_INST = 2.0     # MI355X_MICROARCH.md: a wave64 VALU instruction issues over 2 cycles on a SIMD (the architectural figure; the packed 16-bit ops of the SSV row measure 4.2-4.6)
MEASURED_CYCLES_PER_INST = 4.25   # cycles per wave64 instruction per SIMD of the SSV row body (2 x v_pk_add_f16 clamp + v_pk_maximum3_f16 per two rows) run alone: tools/ubench/valu_rates.hip, profiles/r03_valu_rates.txt (6.35 cycles per register-row = 3 instructions per 2 rows)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--config", choices=["cfg2", "cfg3", "cfg5", "genes", "fasta", "emulate8"], default="cfg3")
    ap.add_argument("--budget-seconds", type=float, default=float(os.environ.get("CKM_BENCH_BUDGET_S", "240")),
                    help="cfg3: wall-clock budget of the timed region; the steps actually run are min(--steps, budget / estimated step)")
    ap.add_argument("--emulate-rank", default=None, help="cfg3: R/W -- run as rank R of W on this one GPU (no collectives)")
    ap.add_argument("--no-cfg2", action="store_true", help="cfg3: skip the nested cfg2 measurement")
    ap.add_argument("--no-emulation", action="store_true", help="cfg3: skip the nested rank-0-of-8 emulation")
    ap.add_argument("--host-profile", default=None, help="cfg3: write a cProfile of the timed steps (host side) to this file")
    ap.add_argument("--bins", type=int, default=100, help="cfg2: bins per GPU (weak scaling)")
    ap.add_argument("--orfs", type=int, default=2000, help="cfg2: ORFs per bin")
    ap.add_argument("--scaling", choices=["weak", "strong"], default="weak", help="cfg2: strong = --bins-total bins sharded over the ranks")
    ap.add_argument("--bins-total", type=int, default=1000, help="cfg2 strong / cfg3: bins of the whole job")
    ap.add_argument("--lineage-bins", type=int, default=0, help="cfg2: bins of a small lineage_wf-equivalent side measurement (0 = skip; cfg3 IS that measurement)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-genes", action="store_true", help="cfg3: skip the gene-calling side legs (gene_front_end, gene_calling, from_fasta)")
    ap.add_argument("--from-fasta-bins", type=int, default=512, help="cfg3 / genes: bins of the from_fasta leg (nucleotide bins -> genes -> tree pass -> analyze pass -> qa table); 0 skips it.  256 bins are mostly the ramp of the calls in flight (31 - 34 s per 1000), 512 read 28 s, 1000 read 27.4 s (profiles/r05C, r05w)")
    ap.add_argument("--hard-bins", type=int, default=128, help="cfg3: bins of the hard_workload side leg (the lineage pass over a HARDER synthetic world: per-bin composition skew, 5 %% low-complexity ORFs, 3-5 diverged paralogs per planted marker -- synthdata/synth_lineage.py: make_lineage_bin(hard=True)); 0 skips it")
    ap.add_argument("--verify", type=int, default=3, help="cfg3 / cfg5: bins of the last timed step whose written tables are diffed against the CPU oracle after the timed region (0 = off)")
    ap.add_argument("--no-verify", action="store_true")
    ap.add_argument("--pipeline", type=int, default=1,
                    help="cfg2: steps in flight at once (own context each); 1 = every step runs alone")
    ap.add_argument("--cpu-baseline-seconds", type=float, default=15.0)
    ap.add_argument("--cpu-baseline-threads", type=int, default=min(32, os.cpu_count() or 1))
    ap.add_argument("--workdir", default=None, help="where the synthetic files go (default: a fresh temp dir)")
    return ap.parse_args()


# ---------------------------------------------------------------------------------------------------------------------
# CPU legs (rank 0, N=1 only).  The oracle is the CHECKER: it is timed here as the baseline, never used by the product.
# ---------------------------------------------------------------------------------------------------------------------
def cpu_baseline(hmm_path, bins, budget_s, threads):
    """Scan half: the restated CPU oracle (kind 'port') on a bounded sample of the same workload: `threads` host threads, one bin each
    (the reference runs one hmmsearch process per bin), every thread searching all models against the first ORFs of its bin.
    Reduce half: oracle/reduce_oracle.py (the Python restatement of ResultsManager/PFAM/MarkerSet that tests pin against the
    reference's own classes; the classes themselves are absent on the GPU box) on the domtblout text of those rows."""
    from concurrent.futures import ThreadPoolExecutor
    from oracle import p7
    from oracle import reduce_oracle as ro
    hs = p7.HmmSet(hmm_path)
    simd_on = p7.set_simd(True)              # the integer filters striped on AVX2 (oracle/p7simd.c; same rows as the scalar loops)
    threads = max(1, min(threads, len(bins)))
    work = []
    for recs in bins[:threads]:
        work.append(([p7.digitize(r[2]) for r in recs], [r[0] for r in recs], [r[1] for r in recs]))
    dsq, names, _d = work[0]
    nseq = min(len(dsq), 200)
    t0 = time.perf_counter()
    hs.search([0], dsq[:nseq], names[:nseq])
    dt = max(time.perf_counter() - t0, 1e-3)
    cells_per_s = sum(len(d) for d in dsq[:nseq]) * hs.M(0) / dt
    models = list(range(hs.n))
    total_M = sum(hs.M(m) for m in models)
    nseq = int(min(min(len(w[0]) for w in work), max(50, budget_s * cells_per_s / (total_M * 300.0))))

    def one(w):
        return hs.search(models, w[0][:nseq], w[1][:nseq])                # the C call releases the GIL
    t0 = time.perf_counter()
    with ThreadPoolExecutor(max_workers=threads) as ex:
        rows = list(ex.map(one, work))
    dt = time.perf_counter() - t0
    residues = sum(sum(len(d) for d in w[0][:nseq]) for w in work)
    nrows = sum(len(r) for r in rows)
    # reduce half on the same rows
    omodels = {}
    for m in models:
        omodels[hs.acc(m)] = {"acc": hs.acc(m), "ga": None, "tc": [25.0, 25.0], "nc": None, "leng": hs.M(m)}
    texts = [hs.format_domtblout(r, w[1][:nseq], w[2][:nseq]) for r, w in zip(rows, work)]
    t0 = time.perf_counter()
    for t in texts:
        ro.reduce_bin(t, omodels, "", [sorted(omodels)])
    dt_red = time.perf_counter() - t0
    t0 = time.perf_counter()
    with ThreadPoolExecutor(max_workers=threads) as ex:
        pr = list(ex.map(lambda w: hs.msv_probe(models, w[0][:max(4, nseq // 4)]), work))
    gcups = sum(c for c, _k in pr) / max(time.perf_counter() - t0, 1e-6) / 1e9 / threads
    p7.set_simd(False)
    out = {"value": residues * len(models) / dt, "unit": "residue*HMM/s", "cores": threads, "kind": "port-simd" if simd_on else "port", "msv_gcups_per_core": gcups,
           "sample": "restated CPU oracle, %s, NOT HMMER (absent from the reference and this image); %d threads x (all %d models x first %d ORFs of "
                     "one bin each), %d residues, %d rows, %.1f s" % ("its integer filters striped on AVX2 (oracle/p7simd.c: byte MSV at %.1f GCUPS per core), the float stages scalar" % gcups
                                                                       if simd_on else "a SCALAR port (no AVX2 on this CPU)", threads, len(models), nseq, residues, nrows, dt),
           "reduce": {"kind": "port", "what": "oracle/reduce_oracle.py (Python restatement pinned against the reference's classes; 1 thread)",
                      "bins": len(texts), "rows": nrows, "seconds": dt_red, "bins_per_s": len(texts) / max(dt_red, 1e-9)}}
    hs.close()
    return out


def hmmer_leg(hmm_path, bins, threads, workdir):
    """BASELINE.md leg A: if a real `hmmsearch` is on PATH, time it on the same genes (one process per bin, --cpu 1, `threads` at a
    time, the reference's own fan-out: checkm/markerGeneFinder.py:59-83 -> checkm/hmmer.py:61-74) and diff its rows against ours
    (tools/diff_vs_hmmsearch.py).  Returns None when HMMER is absent (as in this image)."""
    exe = shutil.which("hmmsearch")
    if exe is None:
        return None
    from concurrent.futures import ThreadPoolExecutor
    from synthdata import synth
    sample = bins[:max(1, threads)]
    faa = []
    for b, recs in enumerate(sample):
        f = os.path.join(workdir, "hmmer_leg_%d.faa" % b)
        synth.write_fasta(f, recs)
        faa.append(f)

    def run(f):
        subprocess.check_call([exe, "--domtblout", f + ".tbl", "--noali", "--notextw", "-E", "0.1", "--domE", "0.1", "--cpu", "1", hmm_path, f],
                              stdout=subprocess.DEVNULL)
    t0 = time.perf_counter()
    with ThreadPoolExecutor(max_workers=max(1, threads)) as ex:
        list(ex.map(run, faa))
    dt = time.perf_counter() - t0
    residues = sum(sum(len(r[2]) for r in recs) for recs in sample)
    nmodels = sum(1 for line in open(hmm_path) if line.startswith("NAME"))
    ver = subprocess.run([exe, "-h"], stdout=subprocess.PIPE).stdout.decode(errors="replace").split("\n")[1:2]
    return {"value": residues * nmodels / dt, "unit": "residue*HMM/s", "cores": min(threads, len(sample)), "kind": "reference",
            "sample": "hmmsearch (%s) --cpu 1, %d processes at a time, %d bins x %d models, %.1f s" % (" ".join(ver).strip("# "), min(threads, len(sample)), len(sample), nmodels, dt),
            "tables": [f + ".tbl" for f in faa], "faa": faa}


# ---------------------------------------------------------------------------------------------------------------------
def ssv_roofline(st_like, ssv_ms_per_step, bins, orfs, extra_note="", clock_hz=2.4e9):
    """roofline of the dominant kernel ssv_kernel<Q>: algorithmic bytes = sum over (model, sequence) pairs of (L + 12) over the
    kernel's time measured with HIP events on the library's streams.  HBM traffic and the VALU instruction count come from separate
    rocprofv3 --pmc passes recorded in profiles/ (tools/gpu_collect.sh); they are quoted -- with their source -- only for the
    workload that was profiled."""
    alg_bytes = float(st_like["residue_hmm"]) + 12.0 * float(st_like["pairs_ssv"])
    ssv_s = max(ssv_ms_per_step, 1e-9) / 1e3
    achieved = alg_bytes / ssv_s / 1e9
    roof = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": None,
            "algorithmic_bytes": alg_bytes, "kernel": "ssv_kernel_h<Q>", "ms_per_step_kernel": ssv_ms_per_step,
            "note": "the kernel is VALU-issue-bound by design (SURVEY H3), not HBM-bound: see roofline_valu / step_utilisation" + extra_note}
    valu = None
    tf = None
    import glob
    found = sorted(f for f in glob.glob(os.path.join(ROOT, "profiles", PROFILE_TAG + "*_ssv_traffic.json")) if "cfg3" not in os.path.basename(f))   # the latest cfg2 pass of this round
    for cand in found[::-1] + [os.path.join(ROOT, "profiles", "r02b_ssv_traffic.json")]:
        if os.path.exists(cand):
            tf = cand
            break
    if tf is not None and bins == 100 and orfs == 2000:
        with open(tf) as f:
            pm = json.load(f)
        src = os.path.relpath(tf, ROOT)
        roof["traffic"] = pm["hbm_bytes_corrected"]
        roof["traffic_source"] = src + " (rocprofv3 --pmc FETCH_SIZE/WRITE_SIZE pass of this workload, recorded, not measured in this run)"
        cyc = ssv_s * clock_hz / (pm["valu_insts"] / 1024.0)
        valu = {"bound": "valu-issue", "clock_hz": clock_hz, "wave_insts_per_step": pm["valu_insts"], "source": src + " (rocprofv3 --pmc SQ_INSTS_VALU pass, recorded)",
                "cycles_per_inst_per_simd": cyc, "measured_rate_of_this_opcode_mix": MEASURED_CYCLES_PER_INST, "frac_of_measured_rate": min(1.0, MEASURED_CYCLES_PER_INST / cyc),
                "nominal_cycles_per_inst": NOMINAL_CYCLES_PER_INST, "frac_of_nominal_issue": min(1.0, NOMINAL_CYCLES_PER_INST / cyc),
                "note": "cycles per wave64 VALU instruction per SIMD over the SSV launches of this run (HIP events; they share the SIMDs with the chain kernels), "
                        "against the rate the row body of the kernel issues at when it runs alone (tools/ubench/valu_rates.hip, profiles/r03_valu_rates.txt) -- a "
                        "measured rate of this opcode mix, not an architectural peak (MI355X_MICROARCH.md quotes 2 cycles per wave64 op; f32 add/mul/fma measure "
                        "2.6-2.7, the packed 16-bit ops 4.4-4.6)"}
    if valu is not None and "all_kernels" in pm:
        valu["all_kernels_of_a_step"] = {"wave_insts": pm["all_kernels"]["valu_insts"], "hbm_bytes": pm["all_kernels"]["hbm_bytes_corrected"], "source": src}
    return roof, valu


def stage_pairs(st):
    g = (lambda k: int(st[k])) if isinstance(st, dict) else (lambda k: int(getattr(st, k)))
    return {"ssv": g("pairs_ssv"), "msv_full": g("pairs_msv_full"), "bias": g("pairs_bias"), "vit": g("pairs_vit"), "vit_exact": g("pairs_vit_exact"),
            "fwd": g("pairs_fwd"), "dom": g("pairs_dom"), "envelopes": g("envelopes"), "regions_multi": g("regions_multi")}


# ---------------------------------------------------------------------------------------------------------------------
# lineage_wf-equivalent run from files through the product classes
# ---------------------------------------------------------------------------------------------------------------------
def lineage_setup(workdir, nbins, rank, world, sync):
    """The synthetic lineage world + `nbins` genes.faa files (every rank writes a slice of them)."""
    from synthdata import synth, synth_lineage as sl
    from checkm_amd.defaultValues import DefaultValues
    data = os.path.join(workdir, "lineage_data")
    if rank == 0:
        sl.World(data)
    sync()
    w = sl.World(data, write=False)
    DefaultValues.set_data_root(data)
    binIds = ["bin_%04d" % b for b in range(nbins)]
    files = [os.path.join(workdir, "%s.faa" % b) for b in binIds]
    w.write_bin_files([(b, files[b]) for b in range(rank, nbins, world)], jobs=max(1, min(32, ((os.cpu_count() or 2) - 2) // world)))
    if rank == 0:
        w.write_marker_files(workdir, binIds)
    sync()
    return w, binIds, files, os.path.join(workdir, "lineage.ms")


def lineage_pass(w, binIds, files, lin, out, rank, called=True):
    """One lineage_wf-equivalent pass over the marker path: tree pass (phylo.hmm), analyze pass (lineage marker file), qa table.
    called=False: `files` are nucleotide bins, find() calls their genes (checkm/markerGeneFinder.py:113-127)."""
    from checkm_amd import markerGeneFinder as mgf
    from checkm_amd.defaultValues import DefaultValues
    from checkm_amd.markerSets import MarkerSetParser
    from checkm_amd.resultsParser import ResultsParser
    t0 = time.perf_counter()
    finder = mgf.MarkerGeneFinder(8)
    finder.find(files, out, DefaultValues.HMMER_TABLE_PHYLO_OUT, DefaultValues.HMMER_PHYLO_OUT, w.phylo_hmm, False, False, called)
    t1 = time.perf_counter()
    models = finder.find(files, out, DefaultValues.HMMER_TABLE_OUT, DefaultValues.HMMER_OUT, lin, False, False, called)
    t2 = time.perf_counter()
    tot = {}
    for tbl in (DefaultValues.HMMER_TABLE_PHYLO_OUT, DefaultValues.HMMER_TABLE_OUT):
        for k, v in mgf.SCAN_CACHE[(os.path.abspath(out), tbl)]["totals"].items():
            tot[k] = max(tot.get(k, 0), v) if k in ("ws_cap_bytes", "ws_used_bytes") else tot.get(k, 0) + v
    if rank == 0:
        os.makedirs(os.path.join(out, "storage"), exist_ok=True)
        with open(os.path.join(out, "storage", DefaultValues.BIN_STATS_OUT), "w") as f:
            for b in binIds:
                f.write("%s\t%s\n" % (b, repr({"GC": 0.5, "Genome size": 1})))
    from checkm_amd import dist as cdist
    cdist.barrier()
    msp = MarkerSetParser()
    sets = msp.getMarkerSets(out, binIds, lin)
    rp = ResultsParser(models)
    rp.analyseResults(out, DefaultValues.BIN_STATS_OUT, DefaultValues.HMMER_TABLE_OUT)
    rp.printSummary(1, None, sets, False, None, True, os.path.join(out, "qa_table.tsv") if rank == 0 else None, None)
    t3 = time.perf_counter()
    del rp, sets, models                 # (hit lists nobody looked at need not be filled in before the scan is released)
    t4 = time.perf_counter()
    mgf.release_scan(out, background=True)          # (the memory of the scans is returned by a helper thread while the next pass starts; the next release waits for it, find() does not)
    t5 = time.perf_counter()
    return {"tree_find_s": t1 - t0, "analyze_find_s": t2 - t1, "qa_s": t3 - t2, "total_s": t3 - t0, "drop_python_objects_s": t4 - t3, "release_call_s": t5 - t4}, tot


class Env(object):
    """What every configuration needs from the launch: rank / world, the device, barrier + sync, sums and maxima over the ranks."""

    def __init__(self, args):
        import torch
        import torch.distributed as dist
        from checkm_amd import dist as cdist
        self.rank = int(os.environ.get("RANK", "0"))
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        if not torch.cuda.is_available():
            raise SystemExit("bench.py needs a GPU: the hot path has no CPU implementation")
        # (test hooks: CKM_BENCH_DEVICE pins every rank to one device and CKM_BENCH_DIST_BACKEND=gloo keeps the collectives on the host, so
        #  the multi-rank control flow can be exercised on a one-GPU box; the driver's runs use neither)
        self.dev_index = int(os.environ.get("CKM_BENCH_DEVICE", local_rank))
        backend = os.environ.get("CKM_BENCH_DIST_BACKEND", "nccl")
        os.environ["CHECKM_AMD_DEVICE"] = str(self.dev_index)
        os.environ.setdefault("CKM_GPUS", "")           # this launch IS the GPU fan-out (one rank per GPU): find() must not spawn workers of its own
        os.environ.setdefault("CKM_DIST_BACKEND", backend)
        torch.cuda.set_device(self.dev_index)
        if self.world > 1:
            cdist.init_process_group(backend)
        self.dev = torch.device("cuda", self.dev_index) if backend == "nccl" else None
        self.torch, self.dist = torch, dist
        workdir = args.workdir
        if workdir is None:
            workdir = tempfile.mkdtemp(prefix="ckm_bench_") if self.rank == 0 else None
            if self.world > 1:
                box = [workdir]
                dist.broadcast_object_list(box, src=0)
                workdir = box[0]
        os.makedirs(workdir, exist_ok=True)
        self.workdir = workdir

    def sync(self):
        if self.world > 1:
            self.dist.barrier()
        self.torch.cuda.synchronize()

    def _reduce(self, x, op):
        if self.world == 1:
            return float(x)
        t = self.torch.tensor([float(x)], dtype=self.torch.float64, device=self.dev if self.dev is not None else "cpu")
        self.dist.all_reduce(t, op=op)
        return float(t.item())

    def all_sum(self, x):
        return self._reduce(x, self.dist.ReduceOp.SUM)

    def all_max(self, x):
        return self._reduce(x, self.dist.ReduceOp.MAX)

    def finish(self):
        if self.world > 1:
            self.dist.barrier()
            self.dist.destroy_process_group()


def main():
    args = parse()
    env = Env(args)
    if args.config == "cfg3":
        out = bench_cfg3(args, env)
    elif args.config == "genes":            # the gene-calling side legs of the cfg3 line alone (not a BASELINE.json configuration)
        out = {"gene_front_end": gene_front_end(), "gene_calling": gene_calling(env.workdir, cpu_bins=0 if args.no_cpu_baseline else 8)} if env.rank == 0 else None
    elif args.config == "fasta":            # the from_fasta leg of the cfg3 line alone (nucleotide bins -> genes -> tree pass -> analyze pass -> QA table)
        out = None
        if env.rank == 0:
            from synthdata import synth_lineage as sl
            from checkm_amd.defaultValues import DefaultValues
            data = os.path.join(env.workdir, "lineage_data")
            w = sl.World(data)
            DefaultValues.set_data_root(data)
            out = {"from_fasta": from_fasta(w, env.workdir, args.from_fasta_bins)}
    elif args.config == "emulate8":         # the emulated_ranks_of_8 leg of the cfg3 line alone (bench_cfg3 runs it as a child process)
        w, binIds, files, lin = lineage_setup(env.workdir, args.bins_total, env.rank, env.world, env.sync)
        out = emulate_8_ranks(w, binIds, files, lin, env.workdir, env.rank, env)
    elif args.config == "cfg5":
        out = bench_cfg5(args, env)
    else:
        out = bench_cfg2(args, env)
    if env.rank == 0 and out is not None:
        print(json.dumps(out))
    env.finish()


def bench_cfg2(args, env):
    """configs[1]: returns the line (rank 0) or None."""
    import torch
    from checkm_amd import _lib, dist as cdist
    from synthdata import synth
    from checkm_amd import qa as cqa
    rank, world, workdir, dev, local_rank = env.rank, env.world, env.workdir, env.dev, env.dev_index
    sync, all_sum, all_max = env.sync, env.all_sum, env.all_max
    out = None
    # ---- cfg2 inputs (synthetic, fixed seeds): profiles + this rank's bins, packed and resident in HBM ----
    profs = synth.cpr43_profiles()
    hmm_path = os.path.join(workdir, "cpr43_synth_rank%d.hmm" % rank)
    synth.write_hmm(hmm_path, profs)
    if args.scaling == "strong":
        # every cfg2 bin has the same shape, so the size-balanced shard (checkm_amd/dist.py:shard_bins) is a round-robin deal
        my_bins = cdist.shard_bins([1] * args.bins_total, world)[rank]
    else:
        my_bins = [rank * args.bins + b for b in range(args.bins)]
    nb = len(my_bins)
    t0 = time.perf_counter()
    bins = [synth.make_bin(profs, 1000 + b, n_orfs=args.orfs) for b in my_bins]
    t_gen = time.perf_counter() - t0
    ctx = _lib.Context(local_rank)
    prof = _lib.Profiles(ctx, hmm_path)
    t0 = time.perf_counter()
    seqs = _lib.Seqs(ctx, bins)
    t_pack = time.perf_counter() - t0
    plan = cqa.QAPlan.for_hmm_models(prof, [list(range(prof.n))] * nb)     # one marker set of all 43 accessions per bin
    max_rows = max(1, int(all_max(nb)))
    part_ms = {"search": 0.0, "reduce": 0.0, "gather": 0.0}

    def step(s=seqs):
        ta = time.perf_counter()
        hits = _lib.search(ctx, prof, s)
        tb = time.perf_counter()
        qa = plan.reduce(ctx, hits, s)
        tc = time.perf_counter()
        rows = cdist.pack_qa_rows(np.asarray(my_bins), qa.n_markers, qa.n_sets, qa.hist, qa.completeness, qa.contamination)
        table = cdist.gather_qa_rows(rows, max_rows, dev)
        td = time.perf_counter()
        part_ms["search"] += (tb - ta) * 1e3; part_ms["reduce"] += (tc - tb) * 1e3; part_ms["gather"] += (td - tc) * 1e3
        st = ctx.stats()
        n = hits.n
        hits.close(); qa.close()
        return st, n, table

    for _ in range(args.warmup):
        step()
    sync()
    for k in part_ms:
        part_ms[k] = 0.0
    ssv_ms = 0.0
    sampler = ClockSampler(env.dev_index).start() if rank == 0 else None
    if args.pipeline <= 1:
        t0 = time.perf_counter()
        for _ in range(args.steps):
            st, nrows, table = step()
            ssv_ms += st.ms_ssv
        sync()
        dt = time.perf_counter() - t0
    else:
        import threading
        from concurrent.futures import ThreadPoolExecutor
        lanes = [(ctx, prof, seqs, plan, threading.Lock())]
        for _ in range(args.pipeline - 1):
            c2 = _lib.Context(local_rank); p2 = _lib.Profiles(c2, hmm_path); s2 = _lib.Seqs(c2, bins)
            lanes.append((c2, p2, s2, cqa.QAPlan.for_hmm_models(p2, [list(range(p2.n))] * nb), threading.Lock()))

        def lane_step(i):
            c, p, s, pl, lock = lanes[i % len(lanes)]
            with lock:                                   # a context runs one search at a time
                hits = _lib.search(c, p, s)
                qa = pl.reduce(c, hits, s)
                rows = cdist.pack_qa_rows(np.asarray(my_bins), qa.n_markers, qa.n_sets, qa.hist, qa.completeness, qa.contamination)
                stl = c.stats(); n = hits.n
                hits.close(); qa.close()
            return rows, stl, n
        with ThreadPoolExecutor(max_workers=len(lanes)) as ex:
            list(ex.map(lane_step, range(len(lanes))))   # warm every lane (plans, buffers)
            sync()
            t0 = time.perf_counter()
            futs = [ex.submit(lane_step, i) for i in range(args.steps)]
            for f in futs:                               # gathers stay in step order on this thread (collectives must line up across ranks)
                rows, st, nrows = f.result()
                table = cdist.gather_qa_rows(rows, max_rows, dev)
                ssv_ms += st.ms_ssv
            sync()
            dt = time.perf_counter() - t0
    dt = all_max(dt)
    device_state = sampler.stop() if sampler is not None else None
    per_step = dt / args.steps
    total_residue_hmm = all_sum(float(st.residue_hmm))        # every rank reports what IT scanned
    total_bins = all_sum(nb)
    value = total_residue_hmm / per_step
    # the same step when the boundary hands over HOST buffers: digitise + pack + H2D of the rank's bins, then the step (SURVEY 8d:
    # "host<->device copies included"); measured once, outside the timed region
    dt_host = None
    if os.environ.get("CKM_BENCH_FROM_HOST", "1") != "0":         # (the counter passes of tools/gpu_collect.sh want exactly one search)
        saved = dict(part_ms)                   # (this extra step is not one of the timed ones: keep it out of step_parts_ms)
        t0 = time.perf_counter()
        seqs2 = _lib.Seqs(ctx, bins)
        st2, _n2, _t2 = step(seqs2)
        torch.cuda.synchronize()
        dt_host = all_max(time.perf_counter() - t0)
        seqs2.close()
        part_ms.update(saved)
    # steady state of a stream of batches: two steps in flight (own context each), so that the end effects of one step -- the chain of
    # its last model-length groups, the exact decisions and the row assembly on the host -- run under the SSV phase of the next.
    # MarkerGeneFinder.find works this way on its batches of bins; `value` above is every step ALONE.
    steady = None
    if args.pipeline <= 1 and world == 1 and os.environ.get("CKM_BENCH_STEADY", "1") != "0":
        import threading
        from concurrent.futures import ThreadPoolExecutor
        c2 = _lib.Context(local_rank); p2 = _lib.Profiles(c2, hmm_path); s2 = _lib.Seqs(c2, bins)
        pl2 = cqa.QAPlan.for_hmm_models(p2, [list(range(p2.n))] * nb)
        lanes2 = [(ctx, prof, seqs, plan), (c2, p2, s2, pl2)]

        def lane_steps(j, n):
            c, p_, s_, pl = lanes2[j]
            if j == 1 and n > 1:
                time.sleep(per_step / 2)             # identical steps would otherwise run in lockstep, tails and all
            for _ in range(n):
                hits = _lib.search(c, p_, s_)
                qa = pl.reduce(c, hits, s_)
                hits.close(); qa.close()
        nst = max(2, args.steps)
        with ThreadPoolExecutor(max_workers=2) as ex:
            list(ex.map(lambda j: lane_steps(j, 1), range(2)))
            list(ex.map(lambda j: lane_steps(j, 1), range(2)))      # (twice: the second context sizes its workspace on its first search)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            list(ex.map(lambda j: lane_steps(j, nst), range(2)))
            torch.cuda.synchronize()
            dts = time.perf_counter() - t0
        steady = {"steps_in_flight": 2, "steps": 2 * nst, "ms_per_step": dts / (2 * nst) * 1e3, "value": float(st.residue_hmm) * 2 * nst / dts,
                  "note": "two contexts, each running its steps back to back; no gather"}
        s2.close(); p2.close(); c2.close()
    # lineage_wf-equivalent side measurement (cfg3-shaped inputs, from files, product classes)
    lineage = None
    if args.lineage_bins > 0:
        try:
            # twice the bins: the first half is an untimed pass of the same shape (contexts, profile DBs, and the workspace the library
            # sizes from what the previous search needed -- growing it is a multi-GB hipMalloc, seconds the first time), the second half is timed
            nl = args.lineage_bins
            w, binIds, files, lin = lineage_setup(workdir, 2 * nl, rank, world, sync)
            warm_parts, _ = lineage_pass(w, binIds[:nl], files[:nl], lin, os.path.join(workdir, "lw_warm"), rank)
            sync()
            binIds, files = binIds[nl:], files[nl:]
            parts, tot = lineage_pass(w, binIds, files, lin, os.path.join(workdir, "lw_out"), rank)
            sync()
            wall = all_max(parts["total_s"])
            lineage = {"workload": "configs[2]-shaped: %d bins of U[1500,6000] ORFs, 2000-profile checkm.hmm, per-bin subsets from a Lineage marker file "
                                   "(43 phylo + 300-1500 lineage models), from genes.faa files through MarkerGeneFinder.find x2 -> ResultsParser.analyseResults "
                                   "-> printSummary" % args.lineage_bins,
                       "bins": args.lineage_bins, "seconds": wall, "bins_per_hour": args.lineage_bins / wall * 3600.0, "parts_s_rank0": parts,
                       "residue_hmm": all_sum(tot.get("residue_hmm", 0)), "residue_hmm_per_s": all_sum(tot.get("residue_hmm", 0)) / wall,
                       "first_pass_s": warm_parts["total_s"],
                       "note": "includes FASTA ingest, both scans, domtblout files, reduction and the QA table; timed on the second of two passes over different "
                               "bins of the same shape (first_pass_s = the first one, with context creation, profile upload and workspace allocation); "
                               "the full 1000-bin run is `bench.py --config cfg3`"}
        except SystemExit as e:            # the product's error path is logger.error + sys.exit
            lineage = {"error": "lineage_wf-equivalent run failed: exit %s" % (e.code,)}
    if rank == 0:
        clock_hz = sampled_clock_hz(device_state)
        roof, valu = ssv_roofline({"residue_hmm": st.residue_hmm, "pairs_ssv": st.pairs_ssv}, ssv_ms / args.steps, nb, args.orfs, clock_hz=clock_hz)
        roof["launches_per_step"] = int(st.ssv_launches)
        step_util = None
        if valu is not None and "all_kernels_of_a_step" in valu:
            ak = valu.pop("all_kernels_of_a_step")
            # the whole step against the device: every kernel's VALU instructions over the step's wall time (the SSV launches share the
            # SIMDs with the chains of the other groups, so the per-kernel fraction above understates how busy the device is)
            cyc = per_step * clock_hz / (ak["wave_insts"] / 1024.0)
            step_util = {"clock_hz": clock_hz, "valu_wave_insts_per_step": ak["wave_insts"], "cycles_per_inst_per_simd": cyc,
                         "valu_frac_of_measured_rate": min(1.0, MEASURED_CYCLES_PER_INST / cyc), "hbm_bytes_per_step": ak["hbm_bytes"],
                         "hbm_frac": ak["hbm_bytes"] / per_step / 1e9 / HBM_PEAK_GBS,
                         "source": ak["source"] + " (recorded PMC passes of this workload) over this run's ms_per_step"}
        out = {
            "metric": "residues*HMMs/s (marker-gene scan+reduce, cfg2: 43 profiles x 100 synthetic 2 Mb bins per GPU)",
            "value": value, "unit": "residue*HMM/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": per_step * 1e3, "steps_in_flight": args.pipeline, "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None,
            "device_state_timed_region": device_state,
            "dtype": "i16 (SSV/MSV bytes, Viterbi words) + f32 (Forward/Backward)", "data": "synthetic",
            "config": {"workload": "configs[1]: cpr_43-shaped 43 synthetic profiles (M 63..900, sum M %d) x %d bins x %d ORFs %s"
                                   % (sum(p.M for p in profs), nb if args.scaling == "weak" else args.bins_total, args.orfs,
                                      "per GPU" if args.scaling == "weak" else "in total, sharded over the ranks"),
                       "bins_per_gpu": nb, "bins_total": int(total_bins), "orfs_per_bin": args.orfs, "residues_rank0": seqs.total_residues,
                       "parallelism": "bins sharded over %d GPU(s); 1 all_gather of QA rows per step" % world},
            "value_from_host": (total_residue_hmm / dt_host) if dt_host else None,
            "value_from_host_note": "same step with digitise + pack + H2D of the bins inside the clock (host buffers at the boundary); `value` starts from HBM-resident inputs",
            "bins_per_hour_43models": total_bins / per_step * 3600.0,
            "steady_state": steady,
            "lineage_wf_equiv": lineage,
            "gcups_ssv": float(st.cells_ssv) / max(ssv_ms / args.steps / 1e3, 1e-12) / 1e9,
            "roofline": roof, "roofline_valu": valu, "step_utilisation": step_util,
            "stages_ms": {"ssv": st.ms_ssv, "filters": st.ms_filters, "fwdbwd": st.ms_fwdbwd, "domains": st.ms_domains, "host": st.ms_host, "search_total": st.ms_total},
            "step_parts_ms": {k: v / args.steps for k, v in part_ms.items()},
            "stage_pairs": stage_pairs(st),
            "rows": int(nrows), "qa_rows_gathered": int(len(table)), "setup_s": {"generate": t_gen, "pack_and_upload": t_pack},
        }
        if not args.no_cpu_baseline and world == 1:          # rank 0 at N=1 only: the other ranks would wait at the process-group teardown
            real = hmmer_leg(hmm_path, bins, args.cpu_baseline_threads, workdir)
            if real is not None:
                from tools import diff_vs_hmmsearch as dvh
                diffs = []
                for b, (tbl, faa) in enumerate(zip(real.pop("tables"), real.pop("faa"))):
                    mine = os.path.join(workdir, "hmmer_leg_%d.mine.tbl" % b)
                    from checkm_amd.markerGeneFinder import scan_files
                    scan_files(hmm_path, [faa], [mine])
                    diffs.append(dvh.diff_tables(tbl, mine, hmm_path))
                real["row_diff"] = dvh.merge(diffs)
                out["cpu_baseline"] = real
                out["cpu_baseline_port"] = cpu_baseline(hmm_path, bins, args.cpu_baseline_seconds, args.cpu_baseline_threads)
            else:
                out["cpu_baseline"] = cpu_baseline(hmm_path, bins, args.cpu_baseline_seconds, args.cpu_baseline_threads)
                out["cpu_baseline"]["hmmsearch_on_path"] = False
        else:
            out["cpu_baseline"] = None
    seqs.close(); prof.close(); ctx.close()
    return out


def cpu_baseline_cfg3(w, binIds, files, lin, budget_s, threads):
    """The CPU oracle on sampled cfg3 bins: `threads` host threads, one bin each (the reference runs one hmmsearch process per bin), every
    thread scanning the first ORFs of its bin against the model subset the bin's lineage asks for -- the analyze pass of lineage_wf, which is
    96 % of the path's residue x HMM work.  Round 6: the two integer filters run in their striped AVX2 form (oracle/p7simd.c, kind
    'port-simd': the byte MSV filter every pair goes through at the speed of a SIMD CPU implementation; same rows as the scalar loops --
    tests/test_oracle_integer_filters.py); the scalar port is timed beside it on a third of the budget.  NOT HMMER either way."""
    from concurrent.futures import ThreadPoolExecutor
    from checkm_amd.markerSets import MarkerSetParser, wanted_model
    from oracle import p7
    hs = p7.HmmSet(w.checkm_hmm)
    threads = max(1, min(threads, len(binIds)))
    step = max(1, len(binIds) // threads)
    sample = [k * step for k in range(threads)]
    wanted = MarkerSetParser().markerAccessionsForBins([binIds[k] for k in sample], lin)
    work = []
    for k in sample:
        acc = wanted[binIds[k]]
        models = [m for m in range(hs.n) if wanted_model(hs.name(m), hs.acc(m) or None, acc)]
        recs = w.bin_records(k)
        work.append((models, [p7.digitize(r[2]) for r in recs], [r[0] for r in recs]))

    def leg(simd, budget):
        on = p7.set_simd(simd)
        if simd and not on:
            return None
        models, dsq, names = work[0]
        n0 = min(len(dsq), 40)
        t0 = time.perf_counter()
        hs.search(models[:20], dsq[:n0], names[:n0])
        dt = max(time.perf_counter() - t0, 1e-3)
        cells_per_s = sum(len(d) for d in dsq[:n0]) * sum(hs.M(m) for m in models[:20]) / dt
        avg_M = sum(sum(hs.M(m) for m in wk[0]) for wk in work) / float(len(work))
        nseq = int(min(min(len(wk[1]) for wk in work), max(10, budget * cells_per_s / (avg_M * 330.0))))

        def one(wk):
            return hs.search(wk[0], wk[1][:nseq], wk[2][:nseq])                # the C call releases the GIL
        t0 = time.perf_counter()
        with ThreadPoolExecutor(max_workers=threads) as ex:
            rows = list(ex.map(one, work))
        dt = time.perf_counter() - t0
        # the MSV stage alone (every pair goes through it), all threads at once: cells per second and core
        nq = max(4, nseq // 4)

        def probe(wk):
            return hs.msv_probe(wk[0], wk[1][:nq])
        t0 = time.perf_counter()
        with ThreadPoolExecutor(max_workers=threads) as ex:
            pr = list(ex.map(probe, work))
        dtp = max(time.perf_counter() - t0, 1e-6)
        residue_hmm = sum(sum(len(d) for d in wk[1][:nseq]) * len(wk[0]) for wk in work)
        bins_equiv = sum(float(nseq) / len(wk[1]) for wk in work)              # fraction of a bin each thread got through
        return {"value": bins_equiv / dt * 3600.0, "unit": "bins/hour", "cores": threads, "residue_hmm_per_s": residue_hmm / dt,
                "msv_gcups_per_core": sum(c for c, _k in pr) / dtp / 1e9 / threads, "msv_byte_checksum": sum(k for _c, k in pr), "msv_probe_orfs": nq,
                "orfs_per_bin": nseq, "rows": sum(len(r) for r in rows), "seconds": dt}
    try:
        simd = leg(True, budget_s * 2.0 / 3.0)
        scalar = leg(False, budget_s / 3.0 if simd else budget_s)
    finally:
        p7.set_simd(False)
    lo_o, hi_o = min(len(wk[1]) for wk in work), max(len(wk[1]) for wk in work)
    lo_m, hi_m = min(len(wk[0]) for wk in work), max(len(wk[0]) for wk in work)
    what = ("%d threads x (one sampled cfg3 bin each: the first ORFs of its %d-%d against its lineage's %d-%d models, analyze pass only); bins/hour = the fraction of a bin "
            "each thread finished, summed, per hour on these %d cores" % (threads, lo_o, hi_o, lo_m, hi_m, threads))
    if simd:
        out = dict(simd)
        out["kind"] = "port-simd"
        out["scalar_port"] = scalar
        out["sample"] = ("restated CPU oracle with its two INTEGER filters striped on AVX2 (oracle/p7simd.c: byte MSV on 32 lanes at %.1f GCUPS per core here -- HMMER's SSE "
                         "filter is quoted at about 12 on 16 lanes -- and word Viterbi on 16; same rows as the scalar loops, which run %.2f GCUPS per core); the float stages "
                         "behind them (bias filter, Forward / Backward, domain definition: 2 %% of the pairs and fewer) stay scalar, where HMMER is 4-lane SSE -- so this is "
                         "a CPU anchor within a small factor of a SIMD hmmsearch, NOT HMMER, which is absent from the reference and this image.  %s; SIMD leg: first %d ORFs, "
                         "%d rows, %.1f s; scalar leg: first %d ORFs, %.1f s"
                         % (simd["msv_gcups_per_core"], scalar["msv_gcups_per_core"], what, simd["orfs_per_bin"], simd["rows"], simd["seconds"], scalar["orfs_per_bin"], scalar["seconds"]))
    else:
        out = dict(scalar)
        out["kind"] = "port"
        out["sample"] = ("restated CPU oracle: a SCALAR port (this CPU has no AVX2 for oracle/p7simd.c; %.2f GCUPS per thread in its byte filter) -- NOT HMMER; the GPU/CPU ratio of "
                         "this line is not a result.  %s; first %d ORFs, %d rows, %.1f s" % (scalar["msv_gcups_per_core"], what, scalar["orfs_per_bin"], scalar["rows"], scalar["seconds"]))
    out["hmmsearch_on_path"] = shutil.which("hmmsearch") is not None
    hs.close()
    return out


def verify_tables(out_dir, table, hmm_path, lin, binIds, files, k_bins, with_qa, workdir, seed=4):
    """--verify: after the timed region, K random bins of the LAST timed step's output directory are diffed against the CPU oracles
    (tools/verify_sample.py): the written domtblout lines of >= 40 sampled models per bin (every SSV launch class of the bin's model
    list represented) against oracle/p7oracle.c, and -- cfg3 -- the bin's row of the QA table against oracle/reduce_oracle.py.
    The per-bin model dictionaries come from an (untimed) find() over just those bins into a scratch directory: host logic only decides
    them, and that second set of tables must equal the first as well."""
    from checkm_amd import markerGeneFinder as mgf
    from checkm_amd.defaultValues import DefaultValues
    from checkm_amd.markerSets import MarkerSetParser
    from tools import verify_sample as vs
    rng = np.random.default_rng(seed)
    pick = sorted(rng.choice(len(binIds), size=min(k_bins, len(binIds)), replace=False).tolist())
    sub_ids, sub_files = [binIds[b] for b in pick], [files[b] for b in pick]
    t0 = time.perf_counter()
    scratch = os.path.join(workdir, "verify_scratch")
    models = mgf.MarkerGeneFinder(8).find(sub_files, scratch, table, DefaultValues.HMMER_OUT, lin, False, False, True)
    mgf.release_scan(scratch)
    again = all(open(os.path.join(scratch, "bins", b, table)).read() == open(os.path.join(out_dir, "bins", b, table)).read() for b in sub_ids)
    sets = qa_rows = pfam = None
    if with_qa:
        sets = MarkerSetParser().getMarkerSets(out_dir, sub_ids, lin)
        with open(os.path.join(out_dir, "qa_table.tsv")) as f:
            qa_rows = {ln.split("\t")[0]: ln.rstrip("\n") for ln in f.read().split("\n")[1:] if ln.strip()}
        pfam = open(DefaultValues.PFAM_CLAN_FILE).read()
    res = vs.verify(out_dir, table, hmm_path, sub_ids, sub_files, models, k_bins=len(sub_ids), n_models=40, seed=seed,
                    marker_sets=sets, qa_rows=qa_rows, pfam_text=pfam)
    res["same_tables_when_scanned_alone"] = bool(again)
    res["identical"] = bool(res["identical"] and again)
    res["seconds"] = time.perf_counter() - t0
    from checkm_amd import parity
    res["checked_against"] = "oracle/p7oracle.c + oracle/reduce_oracle.py"
    res.update(parity.statement())          # oracle_pinned: false, the declared deviations D1 / D2 / D4 / D5 -- what "identical" means here
    return res


def DefaultValues_HMMER_TABLE_OUT():
    from checkm_amd.defaultValues import DefaultValues
    return DefaultValues.HMMER_TABLE_OUT


def cfg3_counters(alg_bytes):
    """HBM traffic and VALU instruction count of the SSV launches, scaled from the rocprofv3 --pmc passes of a cfg3 SAMPLE
    (profiles/<tag>_cfg3_ssv_traffic.json: counters and algorithmic bytes of the sample; tools/gpu_collect.sh)."""
    import glob
    found = sorted(glob.glob(os.path.join(ROOT, "profiles", PROFILE_TAG + "*_cfg3_ssv_traffic.json")))
    if not found:
        return None
    with open(found[-1]) as f:
        pm = json.load(f)
    k = alg_bytes / float(pm["algorithmic_bytes"])
    return {"source": os.path.relpath(found[-1], ROOT), "sample": pm.get("config"), "scale": k,
            "hbm_bytes": pm["hbm_bytes_corrected"] * k, "valu_insts": pm["valu_insts"] * k,
            "all_valu_insts": pm["all_kernels"]["valu_insts"] * k, "all_hbm_bytes": pm["all_kernels"]["hbm_bytes_corrected"] * k,
            "traffic_over_algorithmic": pm["hbm_bytes_corrected"] / float(pm["algorithmic_bytes"])}


def bench_cfg5(args, env):
    """A one-GPU SLICE of configs[4] (10,000 synthetic 5 Mb MAGs x full Pfam/TIGRFAM marker set, 8 GPUs): a 10,000-profile database with
    a Pfam-like length distribution -- three of its models beyond the kernels' 2048 nodes -- against --bins (default 50) bins of 5000 ORFs,
    EVERY searchable model against every bin, through MarkerGeneFinder.find from files.  Not a default line: `bench.py --config cfg5`."""
    import numpy as np
    from checkm_amd import markerGeneFinder as mgf
    from synthdata import synth, synth_lineage as sl
    from checkm_amd.defaultValues import DefaultValues
    rank, world, workdir = env.rank, env.world, env.workdir
    if world != 1:
        raise SystemExit("cfg5 is a one-GPU slice")
    nbins = args.bins if args.bins != 100 else 50
    nmodels = 10000
    t0 = time.perf_counter()
    data = os.path.join(workdir, "cfg5_data")
    w = sl.World(data, n_models=nmodels)          # (seed 2000: the first 2000 profiles are the calibrated ones of cfg3, the rest carry the fitted STATS formula)
    DefaultValues.set_data_root(data)
    hmm = os.path.join(workdir, "pfam_like_%d.hmm" % nmodels)
    if not os.path.exists(hmm):
        shutil.copyfile(w.checkm_hmm, hmm)
        rng = np.random.default_rng(5)
        longs = [synth.random_profile(rng, M, "long%d" % M, "PF%05d.1" % (90000 + M)) for M in (2049, 3000, 4096, 4500)]
        for p in longs:
            p.stats = (-8.5 - 0.002 * p.M, 0.71, -9.5 - 0.002 * p.M, 0.71, -3.8, 0.71)
        synth.write_hmm(hmm, longs, mode="a")
    # --bins 1250 is one rank's share of configs[4] (10,000 MAGs over 8 GPUs).  Up to 160 distinct bins are generated (a pool of generator
    # processes); beyond that the names repeat them (hard links): the device's work per bin is the same, the box is not kept busy writing text
    distinct = min(nbins, 160)
    dfiles = [os.path.join(workdir, "mag_%04d.faa" % b) for b in range(distinct)]
    w.write_mag_files([(b, dfiles[b]) for b in range(distinct)])
    files = list(dfiles)
    for b in range(distinct, nbins):
        f = os.path.join(workdir, "mag_%04d.faa" % b)
        if not os.path.exists(f):
            try:
                os.link(dfiles[b % distinct], f)
            except OSError:
                shutil.copyfile(dfiles[b % distinct], f)
        files.append(f)
    t_setup = time.perf_counter() - t0
    finder = mgf.MarkerGeneFinder(8)
    warm = files[:min(nbins, 6)]
    t0 = time.perf_counter()
    finder.find(warm, os.path.join(workdir, "cfg5_warm"), DefaultValues.HMMER_TABLE_OUT, DefaultValues.HMMER_OUT, hmm, False, False, True)
    env.sync()
    first_s = time.perf_counter() - t0
    mgf.release_scan(os.path.join(workdir, "cfg5_warm"))
    t0 = time.perf_counter()
    out_dir = os.path.join(workdir, "cfg5_out")
    models = finder.find(files, out_dir, DefaultValues.HMMER_TABLE_OUT, DefaultValues.HMMER_OUT, hmm, False, False, True)
    env.sync()
    dt = time.perf_counter() - t0
    ent = mgf.SCAN_CACHE[(os.path.abspath(out_dir), DefaultValues.HMMER_TABLE_OUT)]
    tot = ent["totals"]
    heads = ent["profiles"].headers
    roof, _v = ssv_roofline(tot, tot.get("ms_ssv", 0.0), -1, -1, "; cfg5 slice: summed over the %d ckm_search calls" % tot.get("searches", 0))
    line = {"metric": "bins/hour + residues*HMMs/s, one-GPU slice of configs[4] (every searchable model of a 10,000-profile database against every bin)",
            "value": nbins / dt * 3600.0, "unit": "bins/hour", "n_gpus": 1, "steps": 1, "warmup": 1, "ms_per_step": dt * 1e3, "higher_is_better": True, "scaling": "strong",
            "vs_baseline": None, "dtype": "f16/i16 (SSV/MSV bytes held exactly, Viterbi words) + f32 (Forward/Backward)", "data": "synthetic",
            "config": {"workload": "configs[4], one rank's share when --bins 1250: %d bins of 5000 ORFs x %d profiles (lognormal lengths, median ~190; 3 of them 2049..4096 nodes: searched through the exact-MSV route; %d beyond 4096 and left out with a warning)"
                                   % (nbins, len(heads), sum(1 for h in heads if not h["searchable"])), "bins_total": nbins, "distinct_bins": distinct, "models_per_bin": len(next(iter(models.values())))},
            "residue_hmm_per_s": tot.get("residue_hmm", 0) / dt, "roofline": roof, "stage_pairs": stage_pairs(tot), "searches": int(tot.get("searches", 0)),
            "cascade_fallback_lanes": int(tot.get("cascade_fallback_lanes", 0)),
            "workspace": {"allocated_bytes_max": int(tot.get("ws_cap_bytes", 0)), "high_water_bytes_max": int(tot.get("ws_used_bytes", 0))},
            "hit_table": {"rows_total": int(sum(p["hits"].n for p in ent["parts"])), "rows_max_per_search": int(max(p["hits"].n for p in ent["parts"])),
                          "bins_max_per_search": int(max(len(p["bins"]) for p in ent["parts"]))},
            "find_parts_s": {k: tot.get(k, 0.0) for k in ("ingest_s", "search_s", "write_s")}, "first_pass_s": first_s, "first_pass_bins": len(warm),
            "setup_s": {"world_and_files": t_setup}, "cpu_baseline": None}
    mgf.release_scan()
    line["verify"] = None
    if args.verify > 0 and not args.no_verify:
        line["verify"] = verify_tables(out_dir, DefaultValues.HMMER_TABLE_OUT, hmm, hmm, ["mag_%04d" % b for b in range(nbins)], files, min(args.verify, 3), False, workdir)
    return line


def sampled_clock_hz(device_state):
    """Mean shader clock of the timed region (ClockSampler), or the 2.4 GHz boost clock of MI355X_MICROARCH.md when it was not sampled."""
    try:
        return float(device_state["sclk_mhz"]["mean"]) * 1e6
    except (TypeError, KeyError):
        return 2.4e9


class ClockSampler(object):
    """Shader clock / board power / temperature of this rank's GPU from the amdgpu hwmon files, sampled twice a second while the timed
    region runs (a saturated device may sit below its boost clock: the cycle counts of `roofline_valu` assume 2.4 GHz).  None when the
    files are not there."""

    def __init__(self, index=0):
        import glob
        self.files, self.samples, self._stop, self._thread = None, [], False, None
        # the hwmon directory of THIS device, found through its PCI address (the box may show other GPUs in sysfs that are not ours)
        try:
            import torch
            pr = torch.cuda.get_device_properties(index)
            addr = "%04x:%02x:%02x.0" % (pr.pci_domain_id, pr.pci_bus_id, pr.pci_device_id)
        except Exception:
            return
        hw = sorted(glob.glob(os.path.join("/sys/bus/pci/devices", addr, "hwmon", "hwmon*")))
        if hw:
            pick = lambda names: next((os.path.join(hw[0], n) for n in names if os.path.exists(os.path.join(hw[0], n))), None)
            self.files = {"sclk_mhz": (pick(["freq1_input"]), 1e-6), "power_w": (pick(["power1_average", "power1_input"]), 1e-6),
                          "temp_c": (pick(["temp2_input", "temp1_input"]), 1e-3)}
            self.addr = addr

    def _read(self):
        row = {}
        for k, (f, scale) in self.files.items():
            try:
                row[k] = float(open(f).read().strip()) * scale if f else None
            except (OSError, ValueError):
                row[k] = None
        return row

    def start(self):
        if self.files is None:
            return self
        import threading

        def loop():
            while not self._stop:
                self.samples.append(self._read())
                time.sleep(0.5)
        self._thread = threading.Thread(target=loop, daemon=True)
        self._thread.start()
        return self

    def stop(self):
        if self._thread is None:
            return None
        self._stop = True
        self._thread.join()
        out = {"samples": len(self.samples)}
        for k in ("sclk_mhz", "power_w", "temp_c"):
            v = [r[k] for r in self.samples if r.get(k) is not None]
            if v:
                out[k] = {"min": round(min(v), 1), "mean": round(sum(v) / len(v), 1), "max": round(max(v), 1)}
        out["note"] = "amdgpu hwmon of PCI device %s (freq1_input, power1_average, junction temperature), every 0.5 s over the timed region" % self.addr
        return out


def gene_front_end():
    """SURVEY 8f N1, first slice: the codon-flag kernel of the gene-calling front end (checkm_amd/csrc/kernels_orf.hip) is a pure streaming
    kernel -- 1 byte read + 1 byte written per base -- so it is the one kernel of this repository priced against the HBM roofline it is
    bound by; and the node extraction of one synthetic 4 Mb genome for both translation tables (what replaces prodigal's two node scans)."""
    import numpy as np
    from checkm_amd import _lib, runtime
    ctx = runtime.get_ctx()
    nbytes = 1 << 30                                  # 1 Gbase: four times the 256 MB last-level cache
    ms = _lib.debug_orf_flags(ctx, nbytes, 10)
    achieved = 2.0 * nbytes / (ms / 1e3) / 1e9
    rng = np.random.default_rng(5)
    genome = [np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, size=int(n))].tobytes() for n in rng.integers(20000, 200000, size=40)]
    t0 = time.perf_counter()
    n11, st11 = _lib.orf_nodes(ctx, genome, 11)
    n4, st4 = _lib.orf_nodes(ctx, genome, 4)
    wall = time.perf_counter() - t0
    return {"kernel": "orf_flags_kernel", "bases_per_launch": nbytes, "algorithmic_bytes_per_launch": 2 * nbytes, "ms_per_launch": ms,
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": None,
                         "note": "1 B read + 1 B written per base; 10 launches over 1 Gbase of device-generated nucleotides, HIP events on the launch stream"},
            "one_genome": {"bases": st11["bases"], "contigs": len(genome), "nodes_table11": int(len(n11["ndx"])), "nodes_table4": int(len(n4["ndx"])),
                           "flags_ms": st11["ms_flags"], "chains_ms": st11["ms_chain"], "wall_s_both_tables_incl_copies_and_sort": wall},
            "note": "the front end alone (start / stop nodes of six frames): the streaming kernel of the gene finder, priced against the HBM roofline; the whole gene finder is the gene_calling leg"}


def gene_calling(workdir, nbins=192, cpu_bins=8):
    """SURVEY 8f N1, the body: bins/hour of the device gene finder from NUCLEOTIDE FASTA files to genes.faa / genes.gff -- both
    translation tables of every bin (checkm/prodigal.py:72-133: two prodigal runs per bin and the coding-density rule), training, node
    scores, both dynamic programs, translations (checkm_amd/geneFinder.py -> ckm_genes_call), on synthetic 2 Mb bins of 20 contigs (24 distinct genomes under 192 names); the
    kernel times of the last call; and the same work by the CPU oracle (oracle/gene_full.c, one thread per bin) on a few of the bins."""
    from concurrent.futures import ThreadPoolExecutor
    from checkm_amd import geneFinder
    from synthdata import synth_genome as sg
    d = os.path.join(workdir, "gene_bins")
    os.makedirs(d, exist_ok=True)
    jobs, bases = [], 0
    t0 = time.perf_counter()
    uniq = {}
    for b in range(nbins):
        f = os.path.join(d, "gbin_%03d.fna" % b)
        u = b % 24                                                        # 24 distinct genomes, each under several names: the device's work per bin is the same
        if u not in uniq:
            uniq[u] = sg.make_genome(5000 + u, n_contigs=20, contig_len=(80000, 120000), gc=0.35 + 0.3 * (u % 11) / 10.0, sd_frac=0.6 if u % 3 else 0.0, table=4 if u % 16 == 7 else 11)
        g = uniq[u]
        bases += sum(len(s) for _c, s in g)
        if not os.path.exists(f):
            sg.write_fasta(f, g)
        od = os.path.join(d, "out_%03d" % b)
        os.makedirs(od, exist_ok=True)
        jobs.append((f, od))
    t_setup = time.perf_counter() - t0
    # untimed warm-up pass at full size: kernels loaded, device blocks of the calls' sizes in the library's block cache (a first pass of a process
    # spends 0.3-0.5 s per call in hipMalloc, serialised over the calls in flight; `first_pass_seconds` is that pass)
    t0 = time.perf_counter()
    geneFinder.call_bin_files(jobs)
    first_pass = time.perf_counter() - t0
    t0 = time.perf_counter()
    res = geneFinder.call_bin_files(jobs)
    dt = time.perf_counter() - t0
    st = dict(geneFinder.call_bins.last_stats)
    ngenes = 0
    for f, od in jobs:
        with open(os.path.join(od, "genes.faa")) as fh:
            ngenes += sum(1 for ln in fh if ln.startswith(">"))
    out = {"value": nbins / dt * 3600.0, "unit": "bins/hour", "bins": nbins, "bases": bases, "genes_written": ngenes, "seconds": dt,
           "bases_per_s": bases * 2 / dt, "tables_per_bin": 2, "table_4_chosen": sum(1 for v in res.values() if v[0] == 4),
           "last_call_kernel_ms": {k: st[k] for k in ("ms_dp_train", "ms_score", "ms_dp_find")}, "last_call_wall_ms": {"to_nodes": st["ms_nodes"], "total": st["ms_total"]},
           "setup_s": t_setup, "first_pass_seconds": first_pass, "python_phases_s": dict(geneFinder.call_bin_files.last_phases),
           "device_fraction_of_wall": None,
           "note": "from nucleotide FASTA files to genes.faa / genes.gff, tables 11 and 4 for every bin: sub-batches of <= 128 Mbase x both tables as calls in flight on "
                   "CKM_GENE_LANES host threads (each call on a stream of its own; nodes resident on the device from the codon flags to the gene records, the host takes "
                   "the logarithms of the training tables, reads the files and asks the library to write them).  The pipeline is latency-bound (a workgroup per bin in the "
                   "dynamic programs, nodes strictly in order), the codon-flag kernel is the HBM-bound one (gene_front_end.roofline); single-genome mode only (-p meta is "
                   "not built).  python_phases_s: device_calls_s is summed over the calls in flight, wall_s is the pass"}
    ph = out["python_phases_s"]
    if ph.get("wall_s"):
        # the share of the pass during which at least one device call was in flight (the files are read and written beside the calls, so
        # their summed thread times say nothing about the device; lines recorded before this figure existed subtracted them from the wall)
        if ph.get("device_busy_s"):
            out["device_fraction_of_wall"] = min(1.0, ph["device_busy_s"] / ph["wall_s"])
        else:
            out["device_fraction_of_wall"] = max(0.0, ph["wall_s"] - ph["read_s"] - ph["choose_and_write_s"] / max(1, ph["lanes"])) / ph["wall_s"]
    if cpu_bins <= 0:
        out["cpu_baseline"] = None
        return out
    try:
        from oracle import genes as og
        sample = [geneFinder.read_contigs(jobs[b][0]) for b in range(min(cpu_bins, nbins))]

        def one(contigs):
            for tt in (11, 4):
                og.find_genes([s for _c, s in contigs], tt)
        t0 = time.perf_counter()
        with ThreadPoolExecutor(max_workers=len(sample)) as ex:
            list(ex.map(one, sample))
        dtc = time.perf_counter() - t0
        out["cpu_baseline"] = {"value": len(sample) / dtc * 3600.0, "unit": "bins/hour", "cores": len(sample), "kind": "port",
                               "sample": "oracle/gene_full.c (a restatement of Prodigal 2.6.3's single-genome mode, NOT prodigal itself: none exists here), %d bins x both "
                                         "tables, one thread per bin, %.1f s" % (len(sample), dtc)}
    except Exception as e:            # (the oracle is the checker; the leg stands without it)
        out["cpu_baseline"] = {"error": str(e)}
    return out


def from_fasta(w, workdir, nbins):
    """lineage_wf's marker path from NUCLEOTIDE bins (what CheckM is normally given): find(phylo.hmm) calls the genes of every bin on the
    device (both translation tables, checkm/prodigal.py:72-133) while it scans the bins whose genes are ready, then find(lineage.ms),
    analyseResults, printSummary -- `nbins` synthetic genomes that carry the proteins of the first `nbins` bins of the lineage world."""
    from checkm_amd import geneFinder, markerGeneFinder as mgf
    d = os.path.join(workdir, "fasta_bins")
    os.makedirs(d, exist_ok=True)
    binIds = ["bin_%04d" % b for b in range(nbins)]
    files = [os.path.join(d, "%s.fna" % b) for b in binIds]
    t0 = time.perf_counter()
    w.write_nucleotide_bins([(b, files[b]) for b in range(nbins)])
    md = os.path.join(d, "markers")
    os.makedirs(md, exist_ok=True)
    lin, _tax = w.write_marker_files(md, binIds)
    bases = sum(os.path.getsize(f) for f in files) * 70 // 71
    t_setup = time.perf_counter() - t0
    mgf.release_scan()
    warm = os.path.join(d, "out_warm")                      # the gene finder's kernels loaded, its buffers in the block cache: a first pass over some of the bins
    shutil.rmtree(warm, ignore_errors=True)
    nw = nbins                           # (a whole pass: the block cache holds what the calls in flight of a steady pass need)
    lineage_pass(w, binIds[:nw], files[:nw], lin, warm, 0, called=False)
    mgf.release_scan()
    out = os.path.join(d, "out")
    shutil.rmtree(out, ignore_errors=True)
    t0 = time.perf_counter()
    parts, _tot = lineage_pass(w, binIds, files, lin, out, 0, called=False)
    mgf.release_scan()
    dt = time.perf_counter() - t0
    ngenes = comp = 0
    with open(os.path.join(out, "qa_table.tsv")) as fh:
        rows = [ln.rstrip("\n").split("\t") for ln in fh][1:]
    for r in rows:
        comp += float(r[-3])
    for b in binIds[:32]:
        with open(os.path.join(out, "bins", b, "genes.faa")) as fh:
            ngenes += sum(1 for ln in fh if ln.startswith(">"))
    return {"seconds": dt, "bins": nbins, "seconds_per_1000_bins": dt * 1000.0 / nbins, "bins_per_hour": nbins / dt * 3600.0, "bases": bases, "mbase_per_s": bases / dt / 1e6,
            "parts_s": parts, "gene_phases_s": dict(geneFinder.call_bin_files.last_phases), "setup_s": t_setup, "warm_bins": nw,
            "genes_per_bin_first_32": ngenes / 32.0, "mean_completeness": comp / max(1, len(rows)),
            "note": "nucleotide FASTA -> genes.faa / genes.gff (device gene finder, tables 11 and 4) -> hmmer.tree.txt -> hmmer.analyze.txt -> QA table, through "
                    "MarkerGeneFinder.find(bCalledGenes=False) x 2 + ResultsParser; the tree pass's scan runs beside the gene calling, the analyze pass reuses the called genes "
                    "(checkm/markerGeneFinder.py:113-117); genomes of ~0.9 coding density carrying the lineage world's proteins"}


def hard_workload(w, workdir, nbins, base_pairs, base_bins, verify_bins):
    """The lineage pass over `nbins` bins of a HARDER world (make_lineage_bin(hard=True)): the plain world's sequences are iid draws from
    one background with one planted copy per marker, so its survivor rates (F1 1.9 %, Forward 0.12 %) are the generator's; real
    proteomes -- composition of their own, low-complexity stretches, paralog families around every marker -- send more pairs into the
    expensive stages.  Untimed side leg (a warm pass, then one measured pass); survivors per bin are set against the plain world's."""
    from checkm_amd import markerGeneFinder as mgf
    d = os.path.join(workdir, "hard_bins")
    os.makedirs(d, exist_ok=True)
    binIds = ["hbin_%04d" % b for b in range(nbins)]
    files = [os.path.join(d, "%s.faa" % b) for b in binIds]
    t0 = time.perf_counter()
    w.write_bin_files([(b, files[b]) for b in range(nbins)], hard=True)
    md = os.path.join(d, "markers")
    os.makedirs(md, exist_ok=True)
    lin, _tax = w.write_marker_files(md, binIds)
    t_setup = time.perf_counter() - t0
    mgf.release_scan()
    out = os.path.join(d, "out")
    # the plain world's first `nbins` bins by the same procedure, for the ratio at equal size (a pass of this size is mostly ramp and tail)
    pids = ["bin_%04d" % b for b in range(nbins)]
    pfiles = [os.path.join(workdir, "%s.faa" % b) for b in pids]
    mp = os.path.join(d, "markers_plain")
    os.makedirs(mp, exist_ok=True)
    plin, _t = w.write_marker_files(mp, pids)
    lineage_pass(w, pids, pfiles, plin, os.path.join(d, "out_plain"), 0)
    mgf._join_releasers()
    t0 = time.perf_counter()
    lineage_pass(w, pids, pfiles, plin, os.path.join(d, "out_plain"), 0)
    mgf._join_releasers()
    dt_plain = time.perf_counter() - t0
    # two warm passes: the first meets tables sized for the plain world (a table that overflows sends its lane through the host-driven
    # cascade once and is larger from then on -- the fallbacks of every pass are on the line)
    fb = []
    for _ in range(2):
        _p, t_w = lineage_pass(w, binIds, files, lin, out, 0)
        fb.append(int(t_w.get("cascade_fallback_lanes", 0)))
        mgf._join_releasers()
    t0 = time.perf_counter()
    parts, tot = lineage_pass(w, binIds, files, lin, out, 0)
    mgf._join_releasers()
    dt = time.perf_counter() - t0
    fb.append(int(tot.get("cascade_fallback_lanes", 0)))
    sp = stage_pairs(tot)
    ratio = {k: (sp[k] / float(nbins)) / (base_pairs[k] / float(base_bins)) for k in sp if base_pairs.get(k)}
    res = {"bins": nbins, "seconds": dt, "seconds_per_1000_bins": dt * 1000.0 / nbins, "bins_per_hour": nbins / dt * 3600.0, "parts_s": parts, "stage_pairs": sp,
           "plain_world_same_bins_seconds": dt_plain, "slowdown_vs_plain_world_same_bins": dt / dt_plain,
           "per_bin_vs_plain_world": ratio, "ssv_kernels_s": tot.get("ms_ssv", 0.0) / 1e3, "cascade_fallback_lanes": fb[-1], "cascade_fallback_lanes_by_pass": fb, "setup_s": t_setup,
           "note": "per-bin composition ~ Dirichlet(40 x Swiss-Prot), 5 % low-complexity ORFs, per planted marker 3-5 paralogs (35-75 % of a sampled domain's residues kept) and "
                   "2-4 block-scrambled copies; per_bin_vs_plain_world = this leg's pairs per bin at each stage over the headline workload's; the slowdown is against the plain "
                   "world's first bins in a pass of the same size; cascade_fallback_lanes_by_pass = two warm passes, then the measured one"}
    if verify_bins > 0:
        res["verify"] = verify_tables(out, DefaultValues_HMMER_TABLE_OUT(), w.checkm_hmm, lin, binIds, files, verify_bins, True, d)
    mgf.release_scan()
    return res


def emulate_8_ranks(w, binIds, files, lin, workdir, rank, env):
    """EVERY rank of 8 on this GPU, one after the other: what configs[3] costs each rank (its LPT shard on the device + the host work a rank
    does), so that the projection is the SLOWEST rank's wall, not rank 0's."""
    from checkm_amd import markerGeneFinder as mgf
    walls, parts8, ssv8, searches8 = [], [], [], []
    try:
        os.environ["CKM_EMULATE_RANK"] = "0/8"
        lineage_pass(w, binIds, files, lin, os.path.join(workdir, "cfg3_emu"), rank)       # (warm: tables and workspace at a rank's size)
        for r in range(8):
            os.environ["CKM_EMULATE_RANK"] = "%d/8" % r
            mgf._join_releasers()
            env.sync()
            t0 = time.perf_counter()
            eparts, etot = lineage_pass(w, binIds, files, lin, os.path.join(workdir, "cfg3_emu"), rank)
            mgf._join_releasers()
            env.sync()
            walls.append(time.perf_counter() - t0); parts8.append(eparts); ssv8.append(etot.get("ms_ssv", 0.0) / 1e3); searches8.append(int(etot.get("searches", 0)))
    finally:
        del os.environ["CKM_EMULATE_RANK"]
    mx, mn, mean = max(walls), min(walls), sum(walls) / len(walls)
    return {"per_rank_wall_s": walls, "max_wall_s": mx, "min_wall_s": mn, "mean_wall_s": mean, "imbalance_max_over_mean": mx / mean,
            "per_rank_ssv_kernels_s": ssv8, "per_rank_searches": searches8, "slowest_rank": int(walls.index(mx)), "parts_s_slowest_rank": parts8[walls.index(mx)]}


def child_leg(argv, timeout_s=900):
    """A side leg in a process of its own (the same device; this process keeps what it holds): a rank of configs[3] IS a fresh process, and
    so is a user's cfg2-sized run -- inside this long-lived one the legs that ran before cost the later ones 6-10 % of host-side speed
    (profiles/r06_not_adopted.txt).  Returns the child's JSON line, or {"error": ...}."""
    import subprocess
    try:
        r = subprocess.run([sys.executable, os.path.abspath(__file__)] + argv, capture_output=True, text=True, timeout=timeout_s)
        lines = [ln for ln in r.stdout.strip().split("\n") if ln.startswith("{")]
        if r.returncode != 0 or not lines:
            return {"error": "child leg %s: rc %d: %s" % (" ".join(argv[:2]), r.returncode, r.stderr[-400:])}
        return json.loads(lines[-1])
    except Exception as e:          # noqa: BLE001  (a side leg must not take the line down)
        return {"error": "child leg %s: %r" % (" ".join(argv[:2]), e)}


def bench_cfg3(args, env):
    """configs[2] (N = 1) / configs[3] (N > 1, strong scaling): the lineage_wf marker path over --bins-total bins from files; the
    product shards the bins over the ranks."""
    rank, world, workdir = env.rank, env.world, env.workdir
    nbins = args.bins_total
    emu = None
    if args.emulate_rank:
        r, wz = (int(x) for x in args.emulate_rank.split("/"))
        if world != 1:
            raise SystemExit("--emulate-rank runs on one GPU")
        emu = (r, wz)
    t0 = time.perf_counter()
    w, binIds, files, lin = lineage_setup(workdir, nbins, rank, world, env.sync)
    t_setup = time.perf_counter() - t0
    if emu:
        os.environ["CKM_EMULATE_RANK"] = "%d/%d" % emu            # checkm_amd/dist.py: the product shards as rank r of w, no process group
    share = world if not emu else emu[1]
    # the first pass of the process, on a slice of the bins: contexts, the 2000-profile database, device tables and workspace at working size
    warm = min(nbins, 128 * share)
    t0 = time.perf_counter()
    if os.environ.get("CKM_BENCH_SKIP_WARM") != "1":           # (counter passes of tools/gpu_collect.sh want exactly one step's launches)
        lineage_pass(w, binIds[:warm], files[:warm], lin, os.path.join(workdir, "cfg3_warm"), rank)
    env.sync()
    first_pass_s = time.perf_counter() - t0
    t0 = time.perf_counter()
    for k in range(max(0, args.warmup - 1) if warm == nbins else min(1, max(0, args.warmup - 1))):
        lineage_pass(w, binIds[:warm], files[:warm], lin, os.path.join(workdir, "cfg3_warm"), rank)
    env.sync()
    second_pass_s = time.perf_counter() - t0
    est = (second_pass_s if second_pass_s > 0 else first_pass_s) * nbins / float(warm)
    # warmup steps proper: whole steps (all the bins), untimed -- the first one at full size still grows tables and the workspace
    # (first_full_step_s on the line); at most two, however many were asked for, so that the run stays within minutes
    full_warm = 0 if (warm == nbins or os.environ.get("CKM_BENCH_SKIP_WARM") == "1") else min(args.warmup, 2)
    full_warm_walls = []
    for k in range(full_warm):
        ts = time.perf_counter()
        lineage_pass(w, binIds, files, lin, os.path.join(workdir, "cfg3_out"), rank)
        env.sync()
        full_warm_walls.append(time.perf_counter() - ts)
    if full_warm_walls:
        est = full_warm_walls[-1]
    est = env.all_max(est)                  # (every rank must run the same number of steps: each one holds a collective)
    steps = int(max(1, min(args.steps, args.budget_seconds // max(est, 1e-3))))
    prof = None
    if args.host_profile and rank == 0:
        import cProfile
        prof = cProfile.Profile()
    env.sync()
    sampler = ClockSampler(env.dev_index).start() if rank == 0 else None
    t0 = time.perf_counter()
    if prof is not None:
        prof.enable()
    step_walls = []
    for k in range(steps):
        ts = time.perf_counter()
        parts, tot = lineage_pass(w, binIds, files, lin, os.path.join(workdir, "cfg3_out"), rank)
        step_walls.append(time.perf_counter() - ts)
    from checkm_amd import markerGeneFinder as _mgf
    tj = time.perf_counter()
    _mgf._join_releasers()               # the background release of the last step's scans belongs to the timed region (every earlier one is waited for by the release that follows it)
    last_release_s = time.perf_counter() - tj
    if prof is not None:
        prof.disable()
    env.sync()
    dt = env.all_max(time.perf_counter() - t0)
    device_state = sampler.stop() if sampler is not None else None
    if prof is not None:
        import io
        import pstats
        buf = io.StringIO()
        ps = pstats.Stats(prof, stream=buf)
        ps.sort_stats("cumulative").print_stats(60)
        ps.sort_stats("tottime").print_stats(40)
        with open(args.host_profile, "w") as f:
            f.write("# cProfile of the timed region of `bench.py --config cfg3` (%d step(s) over %d bins), main thread only\n" % (steps, nbins) + buf.getvalue())
    per_step = dt / steps
    residue_hmm = env.all_sum(tot.get("residue_hmm", 0))
    ssv_ms = env.all_max(tot.get("ms_ssv", 0.0))
    if rank != 0:
        return None
    roof, _valu = ssv_roofline(tot, tot.get("ms_ssv", 0.0), -1, -1, "; cfg3: summed over the %d ckm_search calls of rank 0's batches in one step" % tot.get("searches", 0))
    roof["launches_per_step"] = int(tot.get("ssv_launches", 0))
    valu = step_util = None
    clock_hz = sampled_clock_hz(device_state)
    cnt = cfg3_counters(roof["algorithmic_bytes"])
    if cnt is not None:
        roof["traffic"] = cnt["hbm_bytes"]
        roof["traffic_source"] = "%s: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of a cfg3 sample (%s), scaled by algorithmic bytes x %.1f" % (cnt["source"], cnt["sample"], cnt["scale"])
        cyc = (tot.get("ms_ssv", 0.0) / 1e3) * clock_hz / (cnt["valu_insts"] / 1024.0)
        valu = {"bound": "valu-issue", "clock_hz": clock_hz, "wave_insts_per_step": cnt["valu_insts"], "source": cnt["source"] + " (--pmc SQ_INSTS_VALU pass of the sample, scaled)",
                "cycles_per_inst_per_simd": cyc, "measured_rate_of_this_opcode_mix": MEASURED_CYCLES_PER_INST, "frac_of_measured_rate": min(1.0, MEASURED_CYCLES_PER_INST / cyc),
                "nominal_cycles_per_inst": NOMINAL_CYCLES_PER_INST, "frac_of_nominal_issue": min(1.0, NOMINAL_CYCLES_PER_INST / cyc),
                "note": "time = HIP events over the SSV launches of every search of this run, summed (one SSV phase at a time per device since the contexts share a baton); "
                        "the launches share the SIMDs with the chain kernels of the groups ahead of them and of the other context's search -- step_utilisation prices "
                        "the whole step instead; the rate is what "
                        "tools/ubench/valu_rates.hip measures for the row body of the kernel alone (profiles/r03_valu_rates.txt), not an architectural peak"}
    if cnt is not None:
        # the whole step against the device: every kernel's VALU instructions (scaled from the sample's PMC passes) over the step's wall time
        cyc = per_step * clock_hz / (cnt["all_valu_insts"] / 1024.0)
        step_util = {"clock_hz": clock_hz, "clock_note": "mean shader clock sampled over the timed region (device_state_timed_region); 2.4 GHz when the hwmon files are absent",
                     "valu_wave_insts_per_step": cnt["all_valu_insts"], "cycles_per_inst_per_simd": cyc, "valu_frac_of_measured_rate": min(1.0, MEASURED_CYCLES_PER_INST / cyc),
                     "valu_frac_of_nominal_issue": min(1.0, NOMINAL_CYCLES_PER_INST / cyc),
                     "hbm_bytes_per_step": cnt["all_hbm_bytes"], "hbm_frac": cnt["all_hbm_bytes"] / per_step / 1e9 / HBM_PEAK_GBS,
                     "source": cnt["source"] + " (all kernels of the recorded counter passes, scaled by algorithmic bytes) over this run's ms_per_step"}
    out = {"metric": "bins/hour (lineage_wf-equiv marker path: tree pass + analyze pass + qa, from genes.faa files) + residues*HMMs/s",
           "value": nbins / per_step * 3600.0, "unit": "bins/hour", "n_gpus": world, "steps": steps, "steps_requested": args.steps, "warmup": args.warmup,
           "ms_per_step": per_step * 1e3, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
           "dtype": "f16/i16 (SSV/MSV bytes held exactly, Viterbi words) + f32 (Forward/Backward)", "data": "synthetic",
           "config": {"workload": "configs[%d]: full lineage_wf marker DB (2000 synthetic profiles = checkm.hmm, + 43 = phylo.hmm) x %d bins of U[1500,6000] ORFs; tree pass "
                                  "(43 models, every bin) + analyze pass (per-bin model subsets from a Lineage marker file: marker genes of the bin's lineage chain + clan "
                                  "expansion, 300-1500 models) + qa table, through MarkerGeneFinder.find x2 -> ResultsParser.analyseResults -> printSummary (main.py:969-982)"
                                  % (2 if world == 1 else 3, nbins),
                      "bins_total": nbins, "parallelism": "bins sharded over %d GPU(s) by MarkerGeneFinder.find (file size x models); 1 all_gather of QA rows" % world,
                      "steps_note": "one step = all %d bins; %d step(s) fit the %.0f s budget of the timed region (estimated %.1f s per step from the warm pass)" % (nbins, steps, args.budget_seconds, est)},
           "residue_hmm_per_s": residue_hmm / per_step, "residue_hmm_per_step": residue_hmm,
           "summary": None,          # (filled at the end, placed early: the side legs' headline figures survive a truncated tail of this line)
           "first_pass_s": first_pass_s, "first_pass_bins": warm, "second_pass_s_same_bins": second_pass_s if second_pass_s > 0 else None,
           "first_pass_overhead_s": (first_pass_s - second_pass_s) if second_pass_s > 0 else None,
           "parts_s_rank0": parts, "step_walls_s_rank0": step_walls, "last_release_wait_s": last_release_s, "warmup_full_steps_s_rank0": full_warm_walls, "roofline": roof, "roofline_valu": valu, "step_utilisation": step_util, "stage_pairs": stage_pairs(tot), "ssv_ms_max_rank": ssv_ms,
           "gpu_host_split_s_rank0": {"ssv_kernels": tot.get("ms_ssv", 0.0) / 1e3, "search_calls_sum": tot.get("ms_total", 0.0) / 1e3,
                                      "ingest": tot.get("ingest_s", 0.0), "search": tot.get("search_s", 0.0), "write": tot.get("write_s", 0.0),
                                      "tree_find": parts["tree_find_s"], "analyze_find": parts["analyze_find_s"], "qa": parts["qa_s"],
                                      "note": "ingest/search/write are summed over the two scan lanes (they overlap in time); tree_find + analyze_find + qa = the step"},
           "searches_rank0": int(tot.get("searches", 0)), "cascade_fallback_lanes_rank0": int(tot.get("cascade_fallback_lanes", 0)),
           "workspace_rank0": {"allocated_bytes_max": int(tot.get("ws_cap_bytes", 0)), "high_water_bytes_max": int(tot.get("ws_used_bytes", 0)),
                               "note": "per context (find() keeps two on the device, each with a workspace of its own): the largest workspace a context held and the most a "
                                       "single search asked of it, over the searches of the last timed step.  Rounds 3-4 printed the tree pass's and the analyze pass's maxima "
                                       "ADDED UP under these names (131.7 / 63.2 GB in round 4's early runs were 65.8 + 65.8 and 15.2 + 48.0)"},
           "device_state_timed_region": device_state, "setup_s": {"world_and_files": t_setup}}
    if emu:
        out["emulated_rank"] = "%d/%d" % emu
        out["metric"] += " -- EMULATION of rank %d of %d on one GPU (shard of the bins, all-bins host work, no collective)" % emu
        return out
    if world == 1 and args.verify > 0 and not args.no_verify:
        out["verify"] = verify_tables(os.path.join(workdir, "cfg3_out"), DefaultValues_HMMER_TABLE_OUT(), w.checkm_hmm, lin, binIds, files, args.verify, True, workdir)
    else:
        out["verify"] = None
    if world == 1:
        if not args.no_emulation:
            from checkm_amd import markerGeneFinder as mgf
            mgf.release_scan()
            em = child_leg(["--config", "emulate8", "--bins-total", str(nbins), "--workdir", workdir])
            if "error" not in em:
                em["projected_bins_per_hour_8gpu"] = nbins / em["max_wall_s"] * 3600.0
                em["projected_speedup_over_1gpu"] = per_step / em["max_wall_s"]
                em["note"] = ("this ONE GPU as rank r of 8 for r = 0..7 in turn, in a process of their own (a rank is one): LPT shard of the %d bins (dist.shard_bins: file size x models), "
                              "the host work a rank does, no collective (the one all_gather of QA rows) and no contention for the shared output directory -- a projection from "
                              "the slowest emulated rank, not a measurement of configs[3]; no N > 1 run has ever happened on hardware" % nbins)
            out["emulated_ranks_of_8"] = em
        if not args.no_genes:
            out["gene_front_end"] = gene_front_end()
            out["gene_calling"] = gene_calling(workdir)
            if args.from_fasta_bins > 0:
                out["from_fasta"] = from_fasta(w, workdir, min(args.from_fasta_bins, nbins))
        if args.hard_bins > 0:
            out["hard_workload"] = hard_workload(w, workdir, args.hard_bins, out["stage_pairs"], nbins, 0 if args.no_verify else min(2, args.verify))
        if not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline_cfg3(w, binIds, files, lin, args.cpu_baseline_seconds, args.cpu_baseline_threads)
        else:
            out["cpu_baseline"] = None
        if not args.no_cfg2:
            from checkm_amd import markerGeneFinder as mgf
            mgf.release_scan()
            c2 = child_leg(["--config", "cfg2", "--steps", "5", "--warmup", "2", "--no-cpu-baseline", "--no-verify", "--bins", "100", "--orfs", "2000"])
            out["cfg2"] = c2 if "error" in c2 else {k: c2[k] for k in ("metric", "value", "unit", "ms_per_step", "steps", "warmup", "config", "steady_state", "value_from_host", "gcups_ssv", "roofline",
                                                                       "roofline_valu", "step_utilisation", "stages_ms", "step_parts_ms", "stage_pairs", "rows", "device_state_timed_region") if k in c2}
    else:
        out["cpu_baseline"] = None
    g, ff, em, ver = out.get("gene_calling") or {}, out.get("from_fasta") or {}, out.get("emulated_ranks_of_8") or {}, out.get("verify") or {}
    hw = out.get("hard_workload") or {}
    out["summary"] = {"hard_workload_seconds_per_1000_bins": hw.get("seconds_per_1000_bins"), "hard_workload_pairs_per_bin_vs_plain": hw.get("per_bin_vs_plain_world"),
                      "gene_calling_bins_per_hour": g.get("value"), "gene_calling_device_fraction_of_wall": g.get("device_fraction_of_wall"),
                      "from_fasta_seconds_per_1000_bins": ff.get("seconds_per_1000_bins"), "from_fasta_bins": ff.get("bins"),
                      "emulated_8_ranks_max_wall_s": em.get("max_wall_s"), "emulated_8_ranks_projected_speedup_over_1gpu": em.get("projected_speedup_over_1gpu"),
                      "verify_identical": ver.get("identical"), "verify_qa_rows_identical": ver.get("qa_rows_identical"),
                      "n_gpus_this_run": world,
                      "note": ("the 8-rank figures are one GPU emulating each rank in turn, without the collective; before this round's end no N > 1 run existed"
                               if world == 1 else "this line IS an N > 1 run: %d ranks, bins sharded by the product, one all_gather of QA rows per step; the one-GPU legs "
                                                  "(verify, emulation, gene calling, from_fasta, cfg2, cpu_baseline) are not run at N > 1" % world)}
    return out


if __name__ == "__main__":
    main()
