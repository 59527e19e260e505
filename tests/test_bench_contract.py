"""The bench line's contract, checked without a GPU: the flags the driver passes parse, and the line recorded by the round's driver-shaped
run (profiles/r06A_bench_driver_shaped_line.json = `python bench.py --gpus 1 --steps 20 --warmup 5` on one MI355X) carries every field the contract
names, with figures that agree with each other."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LINE = os.path.join(ROOT, "profiles", "r06A_bench_driver_shaped_line.json")


def test_driver_flags_parse():
    for args in (["--gpus", "1", "--steps", "20", "--warmup", "5"], ["--config", "genes", "--no-cpu-baseline"], ["--config", "cfg5"],
                 ["--emulate-rank", "3/8", "--no-genes", "--no-verify"]):
        out = subprocess.run([sys.executable, "-c", "import sys; sys.argv = ['bench.py'] + %r; import bench; a = bench.parse(); print(a.gpus, a.steps, a.warmup, a.config)" % (args,)],
                             capture_output=True, text=True, cwd=ROOT, timeout=120)
        assert out.returncode == 0, out.stderr[-1500:]


def test_recorded_line_meets_the_contract():
    d = json.load(open(LINE))
    base = json.load(open(os.path.join(ROOT, "BASELINE.json")))
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config",
              "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == d["steps_requested"] == 20 and d["warmup"] == 5
    assert d["unit"] == "bins/hour" and d["higher_is_better"] is True and d["vs_baseline"] is None and d["data"].startswith("synthetic")
    assert "workload" in d["config"] and "model" not in d["config"]
    assert d["config"]["workload"].startswith("configs[2]") and d["config"]["bins_total"] == 1000
    assert d["unit"] in base["metric"]
    # value and ms_per_step are the same measurement; the listed steps average to it
    assert abs(d["value"] - 1000 / (d["ms_per_step"] / 1e3) * 3600) / d["value"] < 1e-6
    walls = d["step_walls_s_rank0"]
    assert len(walls) == d["steps"] and abs(sum(walls) / len(walls) * 1e3 - d["ms_per_step"]) / d["ms_per_step"] < 0.01
    r = d["roofline"]
    assert r["bound"] in ("hbm", "mfma") and r["unit"] in ("GB/s", "TFLOP/s") and r["peak"] == 8000.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12
    assert abs(r["achieved"] - r["algorithmic_bytes"] / (r["ms_per_step_kernel"] / 1e3) / 1e9) / r["achieved"] < 1e-6
    assert r["traffic"] is None or r["traffic"] > 0
    assert r["ms_per_step_kernel"] < d["ms_per_step"]                        # the dominant kernel's time lies inside the step
    c = d["cpu_baseline"]
    assert c["kind"] in ("port", "port-simd", "reference") and c["cores"] >= 1 and c["value"] > 0 and c["unit"] == d["unit"] and "scalar" in c["sample"].lower()
    if c["kind"] == "port-simd":          # (round 6: the integer filters striped on AVX2, the scalar port timed beside it on the same sampled bins)
        assert 1.0 < c["msv_gcups_per_core"] < 60.0 and c["scalar_port"]["value"] < c["value"] and "NOT HMMER" in c["sample"]
    v = d["verify"]
    assert v["identical"] is True and v["qa_rows_identical"] is True and v["same_tables_when_scanned_alone"] is True and v["mismatches"] == []
    assert v["oracle_pinned"] is False and sorted(v["known_deviations"]) == ["D1", "D2", "D4", "D5"]      # what "identical" is measured against
    h = d["hard_workload"]
    assert h["verify"]["identical"] is True and h["cascade_fallback_lanes"] == 0 and abs(h["seconds_per_1000_bins"] - h["seconds"] / h["bins"] * 1000) < 1e-6
    assert h["per_bin_vs_plain_world"]["regions_multi"] > 2 and h["slowdown_vs_plain_world_same_bins"] > 1.0
    assert d["cascade_fallback_lanes_rank0"] == 0
    e = d["emulated_ranks_of_8"]
    assert len(e["per_rank_wall_s"]) == 8 and e["max_wall_s"] == max(e["per_rank_wall_s"])
    sm = d["summary"]                                                         # (the figures a reader needs first, early in the line)
    assert sm["verify_identical"] is True and abs(sm["emulated_8_ranks_max_wall_s"] - e["max_wall_s"]) < 1e-9
    g = d["gene_calling"]
    assert g["unit"] == "bins/hour" and abs(sm["gene_calling_bins_per_hour"] - g["value"]) < 1e-6 and g["tables_per_bin"] == 2
    assert abs(g["value"] - g["bins"] / g["seconds"] * 3600) / g["value"] < 1e-6 and 0 < g["device_fraction_of_wall"] <= 1
    ff = d["from_fasta"]
    assert abs(ff["seconds_per_1000_bins"] - ff["seconds"] / ff["bins"] * 1000) < 1e-6 and ff["parts_s"]["total_s"] <= ff["seconds"]
    w = d["workspace_rank0"]
    assert w["high_water_bytes_max"] <= w["allocated_bytes_max"] <= 1.3 * w["high_water_bytes_max"]
