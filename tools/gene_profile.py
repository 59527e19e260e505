"""One gene-calling call (ckm_genes_call, one translation table) over synthetic 2 Mb bins: the workload of the rocprofv3 passes of the
gene finder (tools/gpu_collect.sh genes_prof).  Usage: python tools/gene_profile.py [bins=48] [table=11] [reps=2]"""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from checkm_amd import _lib, runtime      # noqa: E402
from synthdata import synth_genome as sg  # noqa: E402

nbins = int(sys.argv[1]) if len(sys.argv) > 1 else 48
table = int(sys.argv[2]) if len(sys.argv) > 2 else 11
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 2
uniq = {}
bins = []
for b in range(nbins):
    u = b % 12
    if u not in uniq:
        uniq[u] = [s for _c, s in sg.make_genome(5000 + u, n_contigs=20, contig_len=(80000, 120000), gc=0.35 + 0.3 * (u % 11) / 10.0, sd_frac=0.6 if u % 3 else 0.0)]
    bins.append(uniq[u])
ctx = runtime.get_ctx()
for r in range(reps):
    t0 = time.perf_counter()
    cols, per_bin, stats = _lib.call_genes(ctx, bins, table)
    print("rep %d: %.3f s, %d genes, dp_train %.1f ms, dp_find %.1f ms, score %.1f ms, total %.1f ms" % (r, time.perf_counter() - t0, len(cols["begin"]), stats["ms_dp_train"], stats["ms_dp_find"], stats["ms_score"], stats["ms_total"]))
