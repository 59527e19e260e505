"""Two passes of the device gene finder over N synthetic 2 Mb nucleotide bins from files (checkm_amd/geneFinder.call_bin_files: both
translation tables of every bin, sub-batches as calls in flight), half a second apart so that a rocprofv3 kernel trace of the run splits
at the gap (tools/gene_pass_timeline.py).  Usage: python tools/gene_pass.py [bins=256] [workdir=/tmp/ckm_gene_pass]"""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from checkm_amd import geneFinder          # noqa: E402
from synthdata import synth_genome as sg   # noqa: E402

nbins = int(sys.argv[1]) if len(sys.argv) > 1 else 256
d = sys.argv[2] if len(sys.argv) > 2 else "/tmp/ckm_gene_pass"
os.makedirs(d, exist_ok=True)
jobs, uniq = [], {}
for b in range(nbins):
    f = os.path.join(d, "gbin_%03d.fna" % b)
    u = b % 24
    if not os.path.exists(f):
        if u not in uniq:
            uniq[u] = sg.make_genome(5000 + u, n_contigs=20, contig_len=(80000, 120000), gc=0.35 + 0.3 * (u % 11) / 10.0, sd_frac=0.6 if u % 3 else 0.0, table=4 if u % 16 == 7 else 11)
        sg.write_fasta(f, uniq[u])
    od = os.path.join(d, "out_%03d" % b)
    os.makedirs(od, exist_ok=True)
    jobs.append((f, od))
for p in range(2):
    t0 = time.perf_counter()
    geneFinder.call_bin_files(jobs)
    dt = time.perf_counter() - t0
    print("pass %d: %.3f s, %.0f bins/hour, phases %s" % (p, dt, nbins / dt * 3600.0, dict(geneFinder.call_bin_files.last_phases)), flush=True)
    time.sleep(0.5)
