"""AddressSanitizer + UBSan over the host-only parsers of the library: the HMMER3/f reader with the profile configuration
(host_profile.cpp; the reference reads the same files at checkm/hmmerModelParser.py:54-83) and the domtblout reader
(ckm_tables.cpp; checkm/hmmer.py:184-200).  Damaged inputs must be accepted or rejected -- never crash.  No device needed."""
import json
import os
import shutil
import subprocess

import pytest

from synthdata import synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "checkm_amd", "csrc")


@pytest.fixture(scope="module")
def harness(tmp_path_factory):
    gxx = shutil.which("g++")
    if not gxx:
        pytest.skip("g++ not available")
    d = tmp_path_factory.mktemp("sanitize")
    exe = str(d / "fuzz_host")
    cmd = [gxx, "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", "-fno-omit-frame-pointer", "-pthread",
           "-I", CSRC, os.path.join(ROOT, "tests", "native", "fuzz_host.cpp"), os.path.join(CSRC, "host_profile.cpp"), os.path.join(CSRC, "ckm_tables.cpp"), os.path.join(CSRC, "fasta_ingest.cpp"), "-o", exe]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-3000:]
    return exe, d


def _run(exe, mode, path, work, rounds, seed):
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1")
    out = subprocess.run([exe, mode, path, str(work), str(rounds), str(seed)], capture_output=True, text=True, timeout=900, env=env)
    assert out.returncode == 0, (out.stdout[-500:], out.stderr[-4000:])
    return json.loads(out.stdout.strip().split("\n")[-1])


def test_hmm_reader_survives_damaged_files(harness):
    exe, d = harness
    profs = synth.small_profiles(5, 3, mlo=20, mhi=70)
    for k, pr in enumerate(profs):
        pr.stats = (-8.0 - k, 0.71, -9.0 - k, 0.71, -4.0 - k, 0.70)      # any calibration will do: only the reader is under test
    path = str(d / "valid.hmm")
    synth.write_hmm(path, profs)
    got = _run(exe, "hmm", path, d, 400, 11)
    assert got["accepted"] >= 1 and got["rejected"] >= 100      # the undamaged file parses; most damage is noticed and refused


def test_domtblout_reader_survives_damaged_tables(harness):
    exe, d = harness
    case = json.load(open(os.path.join(ROOT, "tests", "golden", "reduce_cases.json")))["cases"][0]
    path = str(d / "valid.domtblout")
    with open(path, "w") as f:
        f.write(case["domtblout"])
    got = _run(exe, "dom", path, d, 600, 12)
    assert got["accepted"] >= 1 and got["rejected"] >= 50


def test_parallel_merge_of_a_batch_equals_the_serial_loop(harness, tmp_path):
    """The files of a batch are laid end to end and ordered by length a bin per thread (fasta_ingest.cpp: merge_fasta_bins, build_seq_order);
    on 1 and on 7 threads the columns must equal the serial loop they replaced, under ASan/UBSan -- an empty file, records without residues
    and an unreadable file among them."""
    exe, d = harness
    profs = synth.small_profiles(21, 6, 40, 120)
    recs = synth.make_bin(profs, 99, n_orfs=400, dup_frac=0.3)
    path = str(tmp_path / "valid.faa")
    synth.write_fasta(path, recs)
    got = _run(exe, "merge", path, tmp_path, 23, 5)
    assert got["mode"] == "merge" and got["files"] == 23 and got["nseq"] > 3000 and got["empty_records"] >= 2


def test_parallel_merge_under_thread_sanitizer(tmp_path):
    """The same merge under TSan: the threads write disjoint stretches of shared columns."""
    gxx = shutil.which("g++")
    if not gxx:
        pytest.skip("g++ not available")
    exe = str(tmp_path / "fuzz_tsan")
    cmd = [gxx, "-std=c++17", "-O1", "-g", "-fsanitize=thread", "-pthread", "-I", CSRC, os.path.join(ROOT, "tests", "native", "fuzz_host.cpp"),
           os.path.join(CSRC, "host_profile.cpp"), os.path.join(CSRC, "ckm_tables.cpp"), os.path.join(CSRC, "fasta_ingest.cpp"), "-o", exe]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    if out.returncode != 0 and "tsan" in out.stderr.lower():
        pytest.skip("ThreadSanitizer runtime not available")
    assert out.returncode == 0, out.stderr[-3000:]
    profs = synth.small_profiles(22, 6, 40, 120)
    path = str(tmp_path / "valid.faa")
    synth.write_fasta(path, synth.make_bin(profs, 98, n_orfs=300, dup_frac=0.3))
    run = subprocess.run([exe, "merge", path, str(tmp_path), "17", "9"], capture_output=True, text=True, timeout=600)
    if run.returncode != 0 and "unexpected memory mapping" in run.stderr:
        pytest.skip("ThreadSanitizer cannot run in this container")
    assert run.returncode == 0 and "ThreadSanitizer" not in run.stderr, run.stderr[-3000:]
    assert json.loads(run.stdout.strip().split("\n")[-1])["files"] == 17


def test_hmm_reader_refuses_fields_that_are_not_probabilities(harness):
    """hmmsearch would atof() these and score with the result; a damaged number is refused here, with file:line in the message."""
    exe, d = harness
    profs = synth.small_profiles(6, 1, mlo=20, mhi=30)
    profs[0].stats = (-8.0, 0.71, -9.0, 0.71, -4.0, 0.70)
    good = str(d / "one.hmm")
    synth.write_hmm(good, profs)
    lines = open(good).read().split("\n")
    body = next(i for i, ln in enumerate(lines) if ln.split()[:1] == ["1"])       # match emissions of node 1
    for bad_token in ("abc", "-0.25", "1.2.3", "inf"):
        toks = lines[body].split()
        toks[3] = bad_token
        bad = str(d / "bad.hmm")
        with open(bad, "w") as f:
            f.write("\n".join(lines[:body] + ["  " + "  ".join(toks)] + lines[body + 1:]))
        out = subprocess.run([exe, "hmm", bad, str(d), "0", "1"], capture_output=True, text=True, timeout=120)
        assert out.returncode == 1 and "is not a probability field" in out.stderr and ":%d:" % (body + 1) in out.stderr, (bad_token, out.stderr)


def test_host_pool_under_thread_sanitizer(tmp_path):
    """The per-worker thread pool (host_pool.h): thousands of back-to-back jobs on two pools, under TSan."""
    gxx = shutil.which("g++")
    if not gxx:
        pytest.skip("g++ not available")
    exe = str(tmp_path / "stress_pool")
    out = subprocess.run([gxx, "-std=c++17", "-O1", "-g", "-fsanitize=thread", "-pthread", "-I", CSRC, os.path.join(ROOT, "tests", "native", "stress_pool.cpp"), "-o", exe],
                         capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-3000:]
    run = subprocess.run([exe], capture_output=True, text=True, timeout=900, env=dict(os.environ, TSAN_OPTIONS="halt_on_error=1"))
    assert run.returncode == 0 and run.stdout.strip().endswith("ok"), (run.stdout[-300:], run.stderr[-4000:])


def test_fasta_reader_matches_a_python_parse_and_survives_damage(harness):
    """fasta_ingest.cpp: names / descriptions / digitised residues of a genes.faa file against a plain Python parse by the rules
    CheckM applies to the same file (checkm/util/seqUtils.py:180-211), then damaged copies, three threads at a time."""
    exe, d = harness
    profs = synth.small_profiles(7, 2, mlo=20, mhi=40)
    recs = synth.make_bin(profs, 3, n_orfs=60)
    path = str(d / "genes.faa")
    synth.write_fasta(path, recs)
    with open(path, "a") as f:                       # blank lines, blanks around residues, lower case, an unknown symbol, CRLF, an empty record
        f.write("\n>odd_1 # 5 # 40 # 1 # ID=9_1  two  blanks\r\n  mkvl AC\t\r\n\r\n?xZ*\n>empty_2\n>last_3\nMM")
    alphabet = "ACDEFGHIKLMNPQRSTVWY-BJZOUX*~"
    code = {c: i for i, c in enumerate(alphabet)}
    code.update({c.lower(): i for i, c in enumerate(alphabet)})
    names, descs, seqs = [], [], []
    for line in open(path, "rb").read().decode("latin-1").split("\n"):
        line = line.rstrip("\r")
        if line.startswith(">"):
            parts = line[1:].split(None, 1)
            names.append(parts[0] if parts else "")
            descs.append(parts[1].strip() if len(parts) > 1 else "")
            seqs.append([])
        elif names and line.strip():
            seqs[-1].extend(code.get(c, 26) for c in line.strip())
    h = 1469598103934665603
    for n, ds, sq in zip(names, descs, seqs):
        for blob in (n.encode("latin-1"), b"|", ds.encode("latin-1"), b"|", bytes(sq), b"\n"):
            for c in blob:
                h = ((h ^ c) * 1099511628211) & 0xFFFFFFFFFFFFFFFF
    out = subprocess.run([exe, "fasta", path, str(d), "500", "13"], capture_output=True, text=True, timeout=900,
                         env=dict(os.environ, ASAN_OPTIONS="detect_leaks=1", UBSAN_OPTIONS="print_stacktrace=1"))
    assert out.returncode == 0, (out.stdout[-500:], out.stderr[-4000:])
    lines = out.stdout.strip().split("\n")
    digest, counts = json.loads(lines[0]), json.loads(lines[-1])
    assert digest["nseq"] == len(names) and digest["total_res"] == sum(len(x) for x in seqs) and digest["maxL"] == max(len(x) for x in seqs)
    assert digest["bytes"] == sum((len(x) + 15) // 16 * 16 for x in seqs)
    assert digest["fnv"] == "%016x" % h
    assert counts["accepted"] >= 400
