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


def test_collection_script_parses():
    out = subprocess.run(["bash", "-n", os.path.join(ROOT, "tools", "gpu_collect.sh")], capture_output=True, text=True)
    assert out.returncode == 0, out.stderr


def test_profile_files_from_counter_passes(tmp_path):
    """tools/make_profile_files.py on a hand-made collection: the SSV figures bench.py scales (FETCH doubled + WRITE, VALU instructions of the
    ssv kernels only) and the step totals, which leave out the gene-calling legs' kernels when a collection ran them in the same process."""
    import json
    src, dst = tmp_path / "out", tmp_path / "prof"
    tag = src / "t1"
    dst.mkdir()
    head = '"Correlation_Id","Dispatch_Id","Agent_Id","Queue_Id","Process_Id","Thread_Id","Grid_Size","Kernel_Id","Kernel_Name","Workgroup_Size","LDS_Block_Size","Scratch_Size","VGPR_Count","Accum_VGPR_Count","SGPR_Count","Counter_Name","Counter_Value","Start_Timestamp","End_Timestamp"\n'
    kernels = [("void ckm::ssv_kernel_h<4>(ckm::X)", 100.0), ("void ckm::ssv_kernel_h8<2>(ckm::X)", 50.0), ("void ckm::fwd_kernel<4>(ckm::X)", 30.0),
               ("ckm::gene_dp_kernel(ckm::GeneNodesDev, unsigned int const*)", 1000.0), ("ckm::orf_flags_kernel(unsigned char const*)", 500.0)]

    def write(pass_name, counters):
        d = tag / ("pmc3_" + pass_name)
        d.mkdir(parents=True)
        with open(d / "p_counter_collection.csv", "w") as f:
            f.write(head)
            for k, (name, v) in enumerate(kernels):
                for c, scale in counters:
                    f.write('%d,%d,"Agent 2",1,1,1,64,1,"%s",64,0,0,8,0,16,"%s",%f,%d,%d\n' % (k, k, name, c, v * scale, 1000 * k, 1000 * k + 500000))
        (tag / ("pmc3_%s.json" % pass_name)).write_text(json.dumps({"config": {"workload": "w", "bins_total": 48}, "ms_per_step": 1.0,
                                                                     "roofline": {"algorithmic_bytes": 1e6, "ms_per_step_kernel": 1.0}}) + "\n")
    write("fetch", [("FETCH_SIZE", 1.0)])
    write("write", [("WRITE_SIZE", 0.5)])
    write("sq", [("SQ_INSTS_VALU", 10.0), ("SQ_INSTS_LDS", 2.0)])
    env = dict(os.environ, CKM_PROFILE_SRC=str(src), CKM_PROFILE_DST=str(dst))
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "make_profile_files.py"), "t1"], capture_output=True, text=True, env=env, timeout=120)
    assert out.returncode == 0, out.stderr[-2000:]
    d = json.load(open(dst / "t1_cfg3_ssv_traffic.json"))
    assert d["FETCH_SIZE_KB"] == 150.0 and d["WRITE_SIZE_KB"] == 75.0                      # the two SSV kernel families only
    assert d["hbm_bytes_corrected"] == 2 * 150.0 * 1024 + 75.0 * 1024                      # FETCH doubled (MI355X_MICROARCH.md), WRITE as reported
    assert d["valu_insts"] == 1500.0 and d["lds_insts"] == 300.0 and d["algorithmic_bytes"] == 1e6
    assert d["all_kernels"]["valu_insts"] == 1800.0 and d["all_kernels"]["FETCH_SIZE_KB"] == 180.0       # + fwd; gene_* / orf_* left out
    text = open(dst / "t1_cfg3_pmc_summary.txt").read()
    assert "ssv_kernel" in text and "gene_dp_kernel" in text                                   # (the per-kernel table itself lists everything)


def test_gene_pass_timeline_and_phase_means(tmp_path):
    """tools/gene_pass_timeline.py (exposed time per kernel family over the last gap-free run of a kernel trace) and tools/gene_phase_means.py
    (mean phase durations of ckm_genes_call from CKM_TRACE lines) on hand-made inputs."""
    ms = 1000000
    rows = [("Kind", "Kernel_Name", "Start_Timestamp", "End_Timestamp"),
            ("KERNEL_DISPATCH", "ckm::gene::chain_kernel(ckm::gene::ChainArgs)", 0, 5 * ms),                       # an earlier pass: cut off by the 0.3 s gap
            ("KERNEL_DISPATCH", "void ckm::gene::gene_dp_kernel<0, 1024>(ckm::gene::Nodes)", 1000 * ms, 1100 * ms),
            ("KERNEL_DISPATCH", "ckm::gene::chain_kernel(ckm::gene::ChainArgs)", 1050 * ms, 1060 * ms),
            ("KERNEL_DISPATCH", "void ckm::gene::g_map_kernel<ckm::gene::gene_pipeline(x)::{lambda(unsigned long)#17}>(unsigned long, {lambda(unsigned long)#17})", 1100 * ms, 1130 * ms)]
    f = tmp_path / "g_kernel_trace.csv"
    f.write_text("\n".join(",".join('"%s"' % c for c in r) for r in rows) + "\n")
    out = _run([os.path.join(ROOT, "tools", "gene_pass_timeline.py"), str(f)])
    assert out.splitlines()[0].startswith("window 0.130 s, 3 launches; a kernel running 100.0 % of it")
    exposed = {l[:40].strip(): l.split()[-1] for l in out.splitlines()[4:] if l.strip()}
    assert exposed == {"gene_dp_kernel<0, 1024>": "90.0", "map #17": "30.0", "chain_kernel": "0.0"}
    t = tmp_path / "err.txt"
    t.write_text("\n".join(["ckm-trace genes call 0 table 11      10.0 ms  text, planes, flags", "ckm-trace genes call 0 table 11      30.0 ms  nodes in working order",
                            "ckm-trace genes call 1 table 4       20.0 ms  text, planes, flags", "ckm-trace genes call 1 table 4       70.0 ms  nodes in working order", "noise"]) + "\n")
    out = _run([os.path.join(ROOT, "tools", "gene_phase_means.py"), str(t)]).splitlines()
    assert out[0].split("mean")[1].split()[0] == "15.0" and out[1].split("mean")[1].split()[0] == "35.0" and out[-1] == "sum of means 50.0 ms"
    assert _run([os.path.join(ROOT, "tools", "gene_phase_means.py"), str(t), "1"]).splitlines()[-1] == "sum of means 70.0 ms"
