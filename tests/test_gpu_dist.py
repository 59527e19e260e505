"""One process per GPU through the PRODUCT path: two ranks (pinned to the one device of the test box, collectives over gloo) run
MarkerGeneFinder.find on their shard of the bins and print ONE gathered QA table; it must equal the table of a single-rank run.
Reference shape: the bin-level fan-out of checkm/markerGeneFinder.py:59-83."""
import os
import subprocess
import sys

import pytest

from synthdata import synth
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


_RCCL_SNIPPET = r"""
import os, sys
import numpy as np
import torch
import torch.distributed as dist
from checkm_amd import dist as cdist
os.environ["CHECKM_AMD_DEVICE"] = "0"
cdist_backend = "nccl"                       # RCCL on ROCm
torch.cuda.set_device(0)
dist.init_process_group(backend=cdist_backend, rank=0, world_size=1)
assert dist.get_backend() == "nccl"
dev = cdist.collective_device()
assert dev is not None and dev.type == "cuda"
n = 1000
rng = np.random.default_rng(3)
rows = cdist.pack_qa_rows(np.arange(n), rng.integers(50, 1500, n), rng.integers(10, 400, n), rng.integers(0, 900, (n, 6)), rng.random(n) * 100.0, rng.random(n) * 7.0)
out = cdist.gather_qa_rows(rows, n + 5, dev, even_alone=True)           # float64 rows through RCCL's all_gather on the device
assert out.shape == rows.shape and out.dtype == np.float64
assert (out.view(np.uint64) == rows.view(np.uint64)).all()              # bit patterns: completeness / contamination travel as doubles
t = torch.ones(1 << 20, dtype=torch.float64, device=dev)
dist.all_reduce(t)
dist.barrier()
torch.cuda.synchronize()
maps = open("/proc/self/maps").read()
lib = sorted({ln.split("/")[-1].strip() for ln in maps.split("\n") if "rccl" in ln.lower()})
print("RCCL_OK", lib, torch.cuda.nccl.version())
dist.destroy_process_group()
"""


def test_qa_rows_travel_through_rccl_on_the_device(gpu_ctx):
    """The one collective of the multi-GPU path -- all_gather of float64 QA rows (checkm_amd/dist.py: gather_qa_rows; the reference's fan-in is
    the writer process of checkm/markerGeneFinder.py:59-83) -- executed by RCCL itself on the device, world size 1: what a one-GPU box can
    prove about the N > 1 path (the library loads, a communicator initialises, device buffers of QA rows go through all_gather bit for
    bit).  No N > 1 run exists: the scaling curve is the driver's to measure."""
    env = dict(os.environ)
    env.update(PYTHONPATH=ROOT, MASTER_ADDR="127.0.0.1", MASTER_PORT="29631", RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("CKM_DIST_BACKEND", None)
    r = subprocess.run([sys.executable, "-c", _RCCL_SNIPPET], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=600)
    out = r.stdout.decode(errors="replace")
    assert r.returncode == 0 and "RCCL_OK" in out, out[-3000:]
    assert "rccl" in out.split("RCCL_OK", 1)[1].lower(), out[-500:]
