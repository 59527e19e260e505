"""The N>1 path on CPU: bin sharding and the single all_gather of QA rows, world_size 2 over gloo."""
import os
import socket
import subprocess
import sys

import numpy as np

from checkm_amd import dist as cdist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_shard_bins_balances_and_covers():
    rng = np.random.default_rng(0)
    w = list(rng.integers(1, 1000, size=37))
    for n in (1, 2, 4, 8):
        shards = cdist.shard_bins(w, n)
        assert sorted(b for s in shards for b in s) == list(range(37))
        loads = [sum(w[b] for b in s) for s in shards]
        assert max(loads) - min(loads) <= max(w)


WORKER = r'''
import os, sys
sys.path.insert(0, %r)
import numpy as np
from checkm_amd import dist as cdist
d = cdist.init_process_group("gloo")
rank = d.get_rank()
shards = cdist.shard_bins([5, 1, 9, 3, 7], d.get_world_size())
mine = shards[rank]
rows = cdist.pack_qa_rows(np.array(mine), [10] * len(mine), [4] * len(mine), np.tile(np.arange(6), (len(mine), 1)),
                          [50.0 + b for b in mine], [1.0 * b for b in mine])
table = cdist.gather_qa_rows(rows, 5)
assert table.shape == (5, cdist.QA_WIDTH), table.shape
assert list(table[:, 0]) == [0, 1, 2, 3, 4]
assert list(table[:, 9]) == [50.0, 51.0, 52.0, 53.0, 54.0]
assert (table[:, 3:9] == np.arange(6)).all()
d.barrier(); d.destroy_process_group()
print("rank", rank, "ok")
'''


def test_gather_qa_rows_world_size_2_gloo(tmp_path):
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    script = tmp_path / "worker.py"
    script.write_text(WORKER % ROOT)
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    for p in procs:
        out, _ = p.communicate(timeout=240)
        assert p.returncode == 0, out.decode()
        assert b"ok" in out


AGREE_WORKER = r'''
import os, sys
sys.path.insert(0, %r)
from checkm_amd import dist as cdist
d = cdist.init_process_group("gloo")
rank = d.get_rank()
assert cdist.agree(True) is True                      # everyone well
got = cdist.agree(rank != 1)                          # rank 1 reports a failure: every rank learns it in the same collective
assert got is False, got
d.destroy_process_group()
print("rank", rank, "ok")
'''


def test_a_failed_rank_is_heard_by_its_peers(tmp_path):
    """find()'s closing collective (checkm_amd/markerGeneFinder.py: cdist.agree) -- a rank that fails says so before it exits, so the
    others end with an error instead of waiting for it."""
    assert cdist.agree(True) is True and cdist.agree(False) is False          # no process group: the caller's own state
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    script = tmp_path / "agree.py"
    script.write_text(AGREE_WORKER % ROOT)
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    for p in procs:
        out, _ = p.communicate(timeout=240)
        assert p.returncode == 0, out.decode()
        assert b"ok" in out
