"""The trace tools that produce profiles/*timeline* and profiles/*lane_trace* run on small hand-made inputs (no GPU, no rocprofv3)."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args):
    out = subprocess.run([sys.executable] + args, capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr[-2000:]
    return out.stdout


def test_occupancy_timeline_last_step_window(tmp_path):
    # two "steps", each: an SSV launch 0..80 ms, a chain kernel 70..100 ms, a hole, then the QA kernel; times in ns
    rows = [("Kind", "Kernel_Name", "Start_Timestamp", "End_Timestamp")]
    ms = 1000000
    for step, base in enumerate((0, 1000 * ms)):
        rows.append(("KERNEL_DISPATCH", "void ckm::ssv_kernel_h<18>(ckm::SsvBlockWork const*)", base + 10 * ms, base + 90 * ms))
        rows.append(("KERNEL_DISPATCH", "void ckm::fwd_kernel<12>(ckm::WorkQueue)", base + 80 * ms, base + 110 * ms))
        rows.append(("KERNEL_DISPATCH", "ckm::count_sets_kernel(int)", base + 150 * ms, base + 151 * ms))
    f = tmp_path / "t_kernel_trace.csv"
    f.write_text("\n".join(",".join('"%s"' % c for c in r) for r in rows) + "\n")
    out = _run([os.path.join(ROOT, "tools", "occupancy_timeline.py"), str(f), "10", "last-step"])
    first = [l for l in out.splitlines() if l.startswith("window")][0]
    # the window is the second step: from the first step's QA kernel (151 ms) to the second's (1151 ms)
    assert first.startswith("window 1000.0 ms: SSV launch(es) running 80.0 ms (8.0 %), only chain/copy kernels 21.0 ms")
    assert "nothing on the device 899.0 ms" in first
    assert "longest stretches with nothing on the device" in out and "859 @ 0" in out


def test_lane_trace_merges_the_two_clocks(tmp_path):
    f = tmp_path / "err.txt"
    f.write_text("\n".join([
        "ckm-trace 0x55aa00012340     1000.000 plan ready",
        "some other line",
        "ckm-trace 0x55aa00012340     1004.000 chain queued",
        "find-trace lane 0 batch 3 ingest 900.000 search 990.000 .. 1600.000",
        "ckm-trace 0x55aa00012340     1500.500 chain drained",
        "ckm-trace 0x55aa00012340     1501.000 results copied",
        "find-trace lane 0 batch 3 written 1650.000",
    ]) + "\n")
    out = _run([os.path.join(ROOT, "tools", "lane_trace.py"), str(f), "10000"]).splitlines()
    labels = [l.split(None, 2)[2] for l in out]
    assert labels == ["b3 ingest start", "b3 search start", "plan ready", "chain queued", "chain drained", "b3 search end", "b3 written"]
    assert out[0].split()[1] == "L0" and out[2].split()[1] == "12340"
    every = _run([os.path.join(ROOT, "tools", "lane_trace.py"), str(f), "10000", "all"])
    assert "results copied" in every
