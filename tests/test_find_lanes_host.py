"""Host logic of MarkerGeneFinder.find with the library calls replaced by stand-ins: every batch is scanned exactly once whichever lane
takes it, the tables land in the right bins, the return value covers every bin, and an error inside a lane ends the run the way the
reference ends it (logger.error + sys.exit(1), checkm/hmmer.py:71-74) instead of hanging the other lane or the calling thread.  The
device path itself is tests/test_gpu_api.py / test_gpu_lineage.py."""
import os
import threading
import time

import pytest

from checkm_amd import _lib, markerGeneFinder as mgf, runtime, workers
from synthdata import synth


class _Stats(object):
    def __getattr__(self, name):
        return 0


class _Ctx(object):
    def __init__(self, k):
        self.k, self.device, self.h = k, 0, 1
        self.reserved = []

    def reserve(self, pairs, cells):
        self.reserved.append((pairs, cells))

    def stats(self):
        return _Stats()


class _Profiles(object):
    def __init__(self, heads):
        self.headers, self.n, self.h = heads, len(heads), 1

    def close(self):
        pass


class _Seqs(object):
    made = []

    def __init__(self, ctx, files):
        self.ctx, self.files = ctx, list(files)
        _Seqs.made.append(self)

    def close(self):
        pass


class _Hits(object):
    def __init__(self, ctx, seqs):
        self.ctx, self.seqs = ctx, seqs

    def write_domtblout(self, prof, seqs, b, path):
        with open(path, "w") as f:
            f.write("# table of %s scanned on lane %d\n" % (os.path.basename(seqs.files[b]), self.ctx.k))

    def close(self):
        pass


@pytest.fixture
def fake_library(monkeypatch, tmp_path):
    profs = synth.small_profiles(31, 5, 30, 60)
    hmm = str(tmp_path / "m.hmm")
    for k, pr in enumerate(profs):
        pr.stats = (-8.0 - k, 0.71, -9.0 - k, 0.71, -4.0 - k, 0.70)
    synth.write_hmm(hmm, profs)
    heads = [dict(name=p.name, acc=p.acc, leng=p.M) for p in profs]
    ctxs = {}
    state = dict(searches=[], lock=threading.Lock(), fail_on=None, delay={})

    def get_ctx_k(k):
        return ctxs.setdefault(k, _Ctx(k))
    monkeypatch.setattr(runtime, "get_ctx", lambda: get_ctx_k(0))
    monkeypatch.setattr(runtime, "get_ctx_k", get_ctx_k)
    monkeypatch.setattr(workers, "devices", lambda: None)
    monkeypatch.setattr(mgf, "profiles_for", lambda ctx, db: _Profiles(heads))
    monkeypatch.setattr(_lib.Seqs, "from_fasta", staticmethod(lambda c, files: _Seqs(c, files)))

    def search(c, prof, seqs, bm, E, domE):
        with state["lock"]:
            n = len(state["searches"])
            state["searches"].append((c.k, tuple(seqs.files)))
        if state["fail_on"] is not None and n == state["fail_on"]:
            raise _lib.CkmError(-3, "the device fell over in search %d" % n)
        time.sleep(state["delay"].get(n, 0.01))
        return _Hits(c, seqs)
    monkeypatch.setattr(_lib, "search", search)
    monkeypatch.setattr(mgf, "PAIR_BUDGET", 40 * 5 * 3)          # ~3 bins of 40 ORFs x 5 models per full batch
    _Seqs.made = []
    yield hmm, state, ctxs
    mgf.SCAN_CACHE.clear()


def _bins(tmp_path, n):
    files = []
    for b in range(n):
        f = tmp_path / ("bin_%02d.faa" % b)
        f.write_text("".join(">g%d_%d\n%s\n" % (b, i, "MKVLAAGIVGLRST" * 22) for i in range(40)))     # ~13 kB: 40 'ORFs' by the 320-bytes rule
        files.append(str(f))
    return files


@pytest.mark.timeout(120)
def test_every_batch_once_whichever_lane_takes_it(fake_library, tmp_path):
    hmm, state, ctxs = fake_library
    files = _bins(tmp_path, 17)
    state["delay"] = {1: 0.4}                                   # the second search is slow: its lane must not be handed every other batch regardless
    out = str(tmp_path / "out")
    models = mgf.MarkerGeneFinder(2).find(files, out, "hmmer.analyze.txt", "hmmer.analyze.ali.txt", hmm, False, False, True)
    assert sorted(models) == ["bin_%02d" % b for b in range(17)] and all(len(m) == 5 for m in models.values())
    scanned = [f for _lane, fs in state["searches"] for f in fs]
    assert sorted(os.path.basename(f) for f in scanned) == sorted("bin_%02d.faa" % b for b in range(17))     # each bin in exactly one search
    assert len(state["searches"]) >= 5
    by_lane = {}
    for lane, fs in state["searches"]:
        by_lane[lane] = by_lane.get(lane, 0) + 1
    assert set(by_lane) == {0, 1} and by_lane[0] >= by_lane[1] + 2                # the lane that was not stuck took the batches that came up meanwhile
    for b in range(17):
        d = os.path.join(out, "bins", "bin_%02d" % b)
        assert open(os.path.join(d, "hmmer.analyze.txt")).read().startswith("# table of bin_%02d.faa" % b)
        assert os.path.getsize(os.path.join(d, "genes.faa")) == os.path.getsize(files[b])                      # called genes are copied in before find() returns
    assert all(c.reserved for c in ctxs.values())


@pytest.mark.timeout(120)
def test_an_error_inside_a_lane_ends_the_run(fake_library, tmp_path, caplog):
    hmm, state, _ctxs = fake_library
    files = _bins(tmp_path, 17)
    state["fail_on"] = 3
    with pytest.raises(SystemExit) as e:
        mgf.MarkerGeneFinder(2).find(files, str(tmp_path / "out"), "hmmer.analyze.txt", "hmmer.analyze.ali.txt", hmm, False, False, True)
    assert e.value.code == 1
    assert any("marker-gene scan failed" in r.getMessage() and "fell over" in r.getMessage() for r in caplog.records)


@pytest.mark.timeout(120)
def test_an_error_in_the_first_search_does_not_strand_the_caller(fake_library, tmp_path):
    """The calling thread waits until every lane is inside its first search before it builds the return value: a lane that dies before
    (or in) its first search must still release it."""
    hmm, state, _ctxs = fake_library
    files = _bins(tmp_path, 9)
    state["fail_on"] = 0
    t0 = time.time()
    with pytest.raises(SystemExit):
        mgf.MarkerGeneFinder(2).find(files, str(tmp_path / "out"), "hmmer.analyze.txt", "hmmer.analyze.ali.txt", hmm, False, False, True)
    assert time.time() - t0 < 20
