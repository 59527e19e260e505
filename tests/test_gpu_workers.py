"""Multi-GPU without a launcher: MarkerGeneFinder.find spawns one worker process per device itself (checkm_amd/workers.py), as the
reference's find() forks its own bin workers (checkm/markerGeneFinder.py:59-83).  Here CKM_GPUS=0,0 puts two workers on the one device
of the test box (collectives over gloo, as two ranks cannot share a device under RCCL); the single process that calls
find -> analyseResults -> printSummary -> cacheResults, unmodified, must produce the same table, the same per-bin files and the same
storage/ caches as a run without workers."""
import os
import subprocess
import sys

import pytest

from synthdata import synth
from tests import common

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(work, marker, tag, gpus, fmts="1,2,5"):
    env = dict(os.environ)
    env.update(CHECKM_AMD_DEVICE="0", CKM_GPUS=gpus, CKM_DIST_BACKEND="gloo", PYTHONPATH=ROOT, HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "workers_find.py"), work, marker, tag, fmts], env=env,
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=900)
    assert r.returncode == 0, r.stdout.decode(errors="replace")[-4000:]


def test_find_with_two_workers_equals_the_single_process_run(gpu_ctx, tmp_path):
    profs = synth.small_profiles(11, 12, 40, 300)
    hmm = common.hmm_file("s11", profs)
    work = str(tmp_path)
    os.makedirs(os.path.join(work, "data", "pfam"))
    with open(os.path.join(work, "data", "pfam", "Pfam-A.hmm.dat"), "w") as f:
        for i, p in enumerate(p for p in profs if p.acc.startswith("PF")):
            f.write("# STOCKHOLM 1.0\n#=GF ID   fam%d\n#=GF AC   %s\n%s//\n" % (i, p.acc, "#=GF CL   CL0001\n" if i < 2 else ""))
    for b in range(7):
        synth.write_fasta(os.path.join(work, "bin_%d.faa" % b), synth.make_bin(profs, 700 + b, n_orfs=60 + 45 * b, dup_frac=0.5))
    _run(work, hmm, "one", "")
    _run(work, hmm, "two", "0,0")
    assert open(os.path.join(work, "mode_one.txt")).read().split()[1] == "0"
    mode = open(os.path.join(work, "mode_two.txt")).read().split("\n")
    assert mode[0] == "workers 2"
    owners = dict(kv.split(":") for kv in mode[1].split()[1:])
    assert len(owners) == 7 and set(owners.values()) == {"0", "1"}          # both workers own bins; every bin has one owner
    for fmt in (1, 2, 5):
        one = open(os.path.join(work, "table_one_fmt%d.tsv" % fmt)).read()
        assert one == open(os.path.join(work, "table_two_fmt%d.tsv" % fmt)).read() and len(one.strip().split("\n")) >= 8
    for b in range(7):
        for name in ("hmmer.analyze.txt", "genes.faa"):
            a = open(os.path.join(work, "out_one", "bins", "bin_%d" % b, name)).read()
            assert a == open(os.path.join(work, "out_two", "bins", "bin_%d" % b, name)).read()
    for name in ("bin_stats_ext.tsv", "marker_gene_stats.tsv"):
        a = open(os.path.join(work, "out_one", "storage", name)).read()
        assert a == open(os.path.join(work, "out_two", "storage", name)).read() and len(a) > 100


def test_find_with_eight_workers_equals_the_single_process_run(gpu_ctx, tmp_path):
    """The 8-GPU shape on the one device of the test box: CKM_GPUS=0,0,0,0,0,0,0,0 -- eight worker processes, LPT shards, the tables of the
    bins written by their owners into ONE output directory, the QA rows exchanged by one all_gather (gloo here) -- through the unmodified
    find -> analyseResults -> printSummary -> cacheResults, byte-equal to the run without workers."""
    profs = synth.small_profiles(11, 12, 40, 300)
    hmm = common.hmm_file("s11", profs)
    work = str(tmp_path)
    os.makedirs(os.path.join(work, "data", "pfam"))
    with open(os.path.join(work, "data", "pfam", "Pfam-A.hmm.dat"), "w") as f:
        for i, p in enumerate(p for p in profs if p.acc.startswith("PF")):
            f.write("# STOCKHOLM 1.0\n#=GF ID   fam%d\n#=GF AC   %s\n%s//\n" % (i, p.acc, "#=GF CL   CL0001\n" if i < 2 else ""))
    nb = 19
    for b in range(nb):
        synth.write_fasta(os.path.join(work, "bin_%02d.faa" % b), synth.make_bin(profs, 900 + b, n_orfs=40 + 17 * b, dup_frac=0.5))
    _run(work, hmm, "one", "", fmts="1,2")
    _run(work, hmm, "eight", "0,0,0,0,0,0,0,0", fmts="1,2")
    mode = open(os.path.join(work, "mode_eight.txt")).read().split("\n")
    assert mode[0] == "workers 8"
    owners = dict(kv.split(":") for kv in mode[1].split()[1:])
    assert len(owners) == nb and set(owners.values()) == {str(r) for r in range(8)}        # every worker owns bins; every bin has one owner
    # ... and the bins a worker owns are its shard of dist.shard_bins (longest first, file size x models), the partition SURVEY 8(e) names
    from checkm_amd import dist as cdist
    files = sorted(f for f in os.listdir(work) if f.endswith(".faa"))
    shards = cdist.shard_bins([os.path.getsize(os.path.join(work, f)) * len(profs) for f in files], 8)
    assert owners == {files[i][:-len(".faa")]: str(r) for r, sh in enumerate(shards) for i in sh}
    for fmt in (1, 2):
        one = open(os.path.join(work, "table_one_fmt%d.tsv" % fmt)).read()
        assert one == open(os.path.join(work, "table_eight_fmt%d.tsv" % fmt)).read() and len(one.strip().split("\n")) == nb + 1
    for b in range(nb):
        a = open(os.path.join(work, "out_one", "bins", "bin_%02d" % b, "hmmer.analyze.txt")).read()
        assert a == open(os.path.join(work, "out_eight", "bins", "bin_%02d" % b, "hmmer.analyze.txt")).read()
    for name in ("bin_stats_ext.tsv", "marker_gene_stats.tsv"):
        a = open(os.path.join(work, "out_one", "storage", name)).read()
        assert a == open(os.path.join(work, "out_eight", "storage", name)).read() and len(a) > 100
