"""One process per GPU through the PRODUCT path: two ranks (pinned to the one device of the test box, collectives over gloo) run
MarkerGeneFinder.find on their shard of the bins and print ONE gathered QA table; it must equal the table of a single-rank run.
Reference shape: the bin-level fan-out of checkm/markerGeneFinder.py:59-83."""
import os
import subprocess
import sys

import pytest

from checkm_amd import synth
from tests import common

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _launch(world, work, marker, fmt, port):
    env = dict(os.environ)
    env.update(CHECKM_AMD_DEVICE="0", CKM_DIST_BACKEND="gloo", PYTHONPATH=ROOT, HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "tests", "multirank_find.py"), work, marker, str(fmt)]
    r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=600)
    assert r.returncode == 0, r.stdout.decode(errors="replace")[-4000:]


def test_two_ranks_give_the_single_rank_table(gpu_ctx, tmp_path):
    profs = synth.small_profiles(11, 12, 40, 300)
    hmm = common.hmm_file("s11", profs)
    work = str(tmp_path)
    os.makedirs(os.path.join(work, "data", "pfam"))
    with open(os.path.join(work, "data", "pfam", "Pfam-A.hmm.dat"), "w") as f:
        for i, p in enumerate(p for p in profs if p.acc.startswith("PF")):
            f.write("# STOCKHOLM 1.0\n#=GF ID   fam%d\n#=GF AC   %s\n%s//\n" % (i, p.acc, "#=GF CL   CL0001\n" if i < 2 else ""))
    for b in range(7):                                  # uneven sizes: the shards are balanced by weight, not by count
        synth.write_fasta(os.path.join(work, "bin_%d.faa" % b), synth.make_bin(profs, 900 + b, n_orfs=60 + 45 * b, dup_frac=0.5))
    for fmt in (1, 5):
        _launch(1, work, hmm, fmt, 29611)
        _launch(2, work, hmm, fmt, 29612)
        one = open(os.path.join(work, "table_world1_fmt%d.tsv" % fmt)).read()
        two = open(os.path.join(work, "table_world2_fmt%d.tsv" % fmt)).read()
        assert one == two and len(one.strip().split("\n")) >= 8
    owned = [open(os.path.join(work, "owned_world2_rank%d.txt" % r)).read().split() for r in (0, 1)]
    assert owned[0] and owned[1] and not set(owned[0]) & set(owned[1]) and len(owned[0]) + len(owned[1]) == 7
    for b in range(7):                                  # every table was written by its owner, byte-identical to the single-rank run
        a = open(os.path.join(work, "out_world1", "bins", "bin_%d" % b, "hmmer.analyze.txt")).read()
        assert a == open(os.path.join(work, "out_world2", "bins", "bin_%d" % b, "hmmer.analyze.txt")).read()
