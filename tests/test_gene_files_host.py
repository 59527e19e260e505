"""Host logic of geneFinder.call_bin_files with the device calls replaced by stand-ins: sub-batching, the choice between the tables, the
refusals and warnings, and -- above all -- that an error in any call or while a sub-batch is read ends the pass with that error instead
of stranding the permits that bound the text held in memory (a dead device fails EVERY call: the pass must still end) or a device result.
The device path itself is tests/test_gpu_genes.py."""
import logging
import os
import threading
import time

import numpy as np
import pytest

from checkm_amd import _lib, geneFinder, runtime


class _FakeCall(object):
    live = 0
    made = 0
    lock = threading.Lock()
    fail_when = None                 # callable(batch, table) -> exception or None

    def __init__(self, ctx, batch, table, closed, mask):
        err = _FakeCall.fail_when(batch, table) if _FakeCall.fail_when else None
        if err is not None:
            raise err
        self.batch, self.table, self.h = batch, table, 1
        self.stats = {"table": table}
        time.sleep(0.01)
        with _FakeCall.lock:
            _FakeCall.live += 1
            _FakeCall.made += 1

    def coding_union(self):          # table 4 covers 0.9 of bins whose first contig starts with 'T' (the stand-in for a recoded genome), else 0.5; table 11 0.6
        frac = [0.9 if (self.table == 4 and self.batch.text[int(self.batch.off[self.batch.bin_first[b]])] == ord('T')) else (0.5 if self.table == 4 else 0.6)
                for b in range(self.batch.nbins)]
        return np.asarray([int(f * n) for f, n in zip(frac, self.batch.bases)], dtype=np.uint64)

    def genes_per_bin(self):
        return np.asarray([0 if self.batch.text[int(self.batch.off[self.batch.bin_first[b]])] == ord('G') else 7 for b in range(self.batch.nbins)], dtype=np.uint32)

    def write_bin(self, b, faa, gff, fna):
        for p in (faa, gff, fna):
            if p:
                with open(p, 'w') as f:
                    f.write("table %d bin %d\n" % (self.table, b))

    def close(self):
        if self.h:
            self.h = 0
            with _FakeCall.lock:
                _FakeCall.live -= 1


@pytest.fixture
def fake_device(monkeypatch):
    monkeypatch.setattr(_lib, "GeneCall", _FakeCall)
    monkeypatch.setattr(runtime, "get_ctx", lambda: object())
    monkeypatch.setenv("CKM_GENE_LANES", "2")            # -> four permits: eleven failing sub-batches would strand them all
    _FakeCall.live = _FakeCall.made = 0
    _FakeCall.fail_when = None
    yield _FakeCall
    _FakeCall.fail_when = None


def _bins(tmp_path, specs):
    """specs: [(first base, total bases)] -> jobs"""
    jobs = []
    for k, (first, n) in enumerate(specs):
        d = tmp_path / ("bin%03d" % k)
        d.mkdir()
        f = tmp_path / ("bin%03d.fna" % k)
        seq = first + "A" * (n - 1)
        with open(f, "w") as out:
            out.write(">c1 some description\n")
            for i in range(0, len(seq), 70):
                out.write(seq[i:i + 70] + "\n")
        jobs.append((str(f), str(d)))
    return jobs


@pytest.mark.timeout(120)
def test_sub_batches_tables_and_files(tmp_path, fake_device):
    jobs = _bins(tmp_path, [("A", 300000), ("T", 250000), ("A", 30000), ("A", 400000), ("T", 210000)])
    done = []
    log = logging.getLogger("test_gene_files_host")
    records = []
    h = logging.Handler()
    h.emit = records.append
    log.addHandler(h)
    try:
        res = geneFinder.call_bin_files(jobs, bNucORFs=True, max_bases=600000, logger=log, on_bin_done=done.append)
    finally:
        log.removeHandler(h)
    assert sorted(done) == sorted(j[0] for j in jobs)
    assert [res[j[0]][0] for j in jobs] == [11, 4, 11, 11, 4]
    assert abs(res[jobs[1][0]][1][4] - 0.9) < 1e-3 and abs(res[jobs[0][0]][1][11] - 0.6) < 1e-3
    for (f, d), t in zip(jobs, (11, 4, 11, 11, 4)):
        for name in ("genes.faa", "genes.gff", "genes.fna"):
            assert open(os.path.join(d, name)).read().startswith("table %d " % t)
    assert fake_device.live == 0 and fake_device.made == 6
    ph = geneFinder.call_bin_files.last_phases
    assert ph["calls"] == 6                                                   # 3 sub-batches (<= 600 kb each) x 2 tables
    assert 0.0 < ph["device_busy_s"] <= ph["wall_s"] and ph["device_busy_s"] <= ph["device_calls_s"] + 1e-9    # some call in flight <= the calls' summed times
    warned = [r.getMessage() for r in records if r.levelno == logging.WARNING]
    assert len(warned) == 1 and "bin002.fna" in warned[0] and "-p meta" in warned[0]          # the 30 kb bin


@pytest.mark.timeout(120)
def test_small_bin_is_refused_before_anything_is_written(tmp_path, fake_device):
    jobs = _bins(tmp_path, [("A", 300000), ("A", 5000), ("A", 300000)])
    with pytest.raises(ValueError) as e:
        geneFinder.call_bin_files(jobs)
    assert "bin001.fna" in str(e.value) and "-p meta" in str(e.value)
    assert fake_device.made == 0 and not any(os.listdir(d) for _f, d in jobs)


@pytest.mark.timeout(120)
def test_every_call_failing_still_ends_the_pass(tmp_path, fake_device):
    """A device that fails every call: eleven sub-batches against four permits.  The first error comes back; nothing hangs."""
    jobs = _bins(tmp_path, [("A", 250000)] * 11)
    fake_device.fail_when = lambda batch, table: _lib.CkmError(-3, "the device fell over")
    with pytest.raises(_lib.CkmError):
        geneFinder.call_bin_files(jobs, max_bases=260000)
    assert fake_device.live == 0


@pytest.mark.timeout(120)
def test_one_failing_call_frees_its_sibling(tmp_path, fake_device):
    jobs = _bins(tmp_path, [("A", 250000)] * 6)
    seen = []

    def fail(batch, table):
        with _FakeCall.lock:
            seen.append(table)
            return _lib.CkmError(-3, "one call failed") if (table == 11 and seen.count(11) == 2) else None
    fake_device.fail_when = fail
    with pytest.raises(_lib.CkmError) as e:
        geneFinder.call_bin_files(jobs, max_bases=260000)
    assert "one call failed" in str(e.value)
    assert fake_device.live == 0                                              # the table-4 result of the failed sub-batch was freed too


@pytest.mark.timeout(120)
def test_refusal_found_when_a_late_sub_batch_is_read(tmp_path, fake_device):
    """A plain file of >= 200 kB is only read when its sub-batch's turn comes; if it then holds < 20 kb of bases (headers around a few
    bases each) the pass ends with the refusal, every result freed."""
    jobs = _bins(tmp_path, [("A", 250000)] * 5)
    odd = tmp_path / "odd.fna"
    with open(odd, "w") as f:
        for k in range(9000):
            f.write(">contig_%05d padded header text to make the file long\nAC\n" % k)
    (tmp_path / "odd").mkdir()
    jobs.insert(3, (str(odd), str(tmp_path / "odd")))
    assert os.path.getsize(odd) >= 200000
    with pytest.raises(ValueError) as e:
        geneFinder.call_bin_files(jobs, max_bases=260000)
    assert "odd.fna" in str(e.value)
    assert fake_device.live == 0


@pytest.mark.timeout(120)
def test_bin_without_genes_is_an_error_after_the_others_are_written(tmp_path, fake_device):
    jobs = _bins(tmp_path, [("A", 250000), ("G", 250000), ("A", 250000)])
    with pytest.raises(ValueError) as e:
        geneFinder.call_bin_files(jobs)
    assert "bin001.fna" in str(e.value) and "no genes" in str(e.value)
    assert all(os.path.exists(os.path.join(d, "genes.faa")) for _f, d in jobs) and fake_device.live == 0


def test_library_reader_equals_the_python_reader(tmp_path):
    """ckm_nuc_batch_read (the sub-batches' files read by the library's host threads) against geneFinder.read_contigs_bytes, the statement
    of CheckM's record rules (checkm/util/seqUtils.py:180-211): hand-made edge files and random ones over the bytes that matter."""
    cases = [b">c1 desc here\nACGT\nAC GT\r\n\n>c2\tx\nTT>TT\n>\n>  spaced id  more\nAAAA", b"junk before\nmore junk\n>x1\nACGTN\n", b"no records at all\n", b">", b"",
             b">only header", b"\n>lead\nAC\n\n\n>z\n", b">h1\r\nACGT\r\nTTTT\r\n>h2 d\r\nGG\r\n", b"x>notarecord\n>r\nAA\x0bBB\x0cCC\n", b"\n", b"\n>", b">\n>\n>"]
    rng = np.random.default_rng(3)
    alphabet = np.frombuffer(b">>\n\n\n\r \tACGTNacgt-*\x0b", dtype=np.uint8)
    for _ in range(200):
        cases.append(rng.choice(alphabet, int(rng.integers(0, 200))).tobytes())
    paths = []
    for k, data in enumerate(cases):
        p = tmp_path / ("f%03d.fna" % k)
        p.write_bytes(data)
        paths.append(str(p))
    nb = _lib.GeneBatch.from_files(paths)
    want = [geneFinder.read_contigs_bytes(p) for p in paths]
    assert nb.nbins == len(paths)
    for k in range(len(paths)):
        assert nb.contigs(k) == want[k], (k, cases[k])
        assert nb.bases[k] == sum(len(s) for _c, s in want[k])
    pb = _lib.GeneBatch(want)                                                # the same layout as the Python path builds
    assert pb.text == nb.text.tobytes() and pb.off.tolist() == nb.off.tolist() and pb.bin_first.tolist() == nb.bin_first.tolist()
    assert [pb.ids[c] for c in range(pb.ncontigs)] == [c for k in range(len(paths)) for c, _s in nb.contigs(k)]
    nb.close()
    with pytest.raises(_lib.CkmError):
        _lib.GeneBatch.from_files([str(tmp_path / "missing.fna")])
    assert _lib.GeneBatch.from_files([]).nbins == 0
