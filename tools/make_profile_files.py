#!/usr/bin/env python
"""Turns what tools/gpu_collect.sh left under gpurun_out/<tag>/ into the files committed under profiles/<tag>_*.
usage: python tools/make_profile_files.py <tag> [<tag> ...]
  bench<k>.json            -> <tag>_bench<k>_line.json (the JSON line, indented)
  pytest_tail.txt, smoke.txt, timeline3.txt, valu_rates.txt -> copied
  stats3/ (+ stats3.json)  -> <tag>_cfg3_kernel_stats.txt   per kernel family / template instance: calls, total, average duration
  pmc3_{fetch,write,sq}/   -> <tag>_cfg3_pmc_summary.txt + <tag>_cfg3_ssv_traffic.json (what bench.py scales roofline.traffic / roofline_valu from)
  pmc_{fetch,write,sq}/    -> <tag>_pmc_summary.txt + <tag>_ssv_traffic.json (cfg2)"""
import collections
import csv
import json
import os
import re
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HDR = ("# FETCH_SIZE / WRITE_SIZE are in KB as reported; MI355X_MICROARCH.md (HBM section): FETCH_SIZE reads 1/2 of the bytes of a wide\n"
       "# coalesced streaming read on gfx950 -> corrected HBM read bytes = 2 * FETCH_SIZE * 1024.  WRITE_SIZE is taken as reported.\n"
       "# SQ_*_CYCLES / SQ_WAIT_* / SQ_ACTIVE_* are in quad-cycle units.  ssv_kernel = ssv_kernel_h<Q> + ssv_kernel_h8<Q8> (packed-half rows).\n")


def last_json(path):
    return json.loads(open(path).read().strip().split("\n")[-1])


NOT_THE_STEP = ("gene_", "orf_")      # kernels of bench.py's gene-calling side legs, when a collection ran them in the same process: not part of a cfg3 step


def counter_total(src, d, counter, only=None):
    f = os.path.join(src, d, "p_counter_collection.csv")
    return sum(float(r["Counter_Value"]) for r in csv.DictReader(open(f)) if r["Counter_Name"] == counter and (only is None or only in r["Kernel_Name"])
               and not any("ckm::" + x in r["Kernel_Name"] for x in NOT_THE_STEP))


def pmc_files(src, dst, prefix, what, line, cmd):
    dirs = [os.path.join(src, prefix + n) for n in ("fetch", "write", "sq")]
    if not all(os.path.exists(os.path.join(d, "p_counter_collection.csv")) for d in dirs):
        return
    pm = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "pmc_summary.py")] + dirs, capture_output=True, text=True).stdout
    open(dst(what + "pmc_summary.txt"), "w").write("# rocprofv3 --pmc passes (FETCH_SIZE | WRITE_SIZE | SQ_*, separate runs, each with --kernel-trace only), MI355X; ONE step each of:\n# " + cmd + "\n" + HDR + pm)
    f, w = counter_total(src, prefix + "fetch", "FETCH_SIZE", "ssv_kernel"), counter_total(src, prefix + "write", "WRITE_SIZE", "ssv_kernel")
    valu = counter_total(src, prefix + "sq", "SQ_INSTS_VALU", "ssv_kernel")
    lds = counter_total(src, prefix + "sq", "SQ_INSTS_LDS", "ssv_kernel")
    ms = [float(l.split()[2]) for l in pm.split("\n") if l.startswith("ssv_kernel ")][0]
    fa, wa = counter_total(src, prefix + "fetch", "FETCH_SIZE"), counter_total(src, prefix + "write", "WRITE_SIZE")
    va = counter_total(src, prefix + "sq", "SQ_INSTS_VALU")
    json.dump({"config": line["config"]["workload"] if "cfg3" in what else "cfg2: 43 profiles x 100 bins x 2000 ORFs, 1 GPU, one search",
               "kernel": "ssv_kernel_h<Q> / ssv_kernel_h8<Q8> (all launches of one step; kernels serialised by counter collection)",
               "algorithmic_bytes": line["roofline"]["algorithmic_bytes"], "FETCH_SIZE_KB": f, "WRITE_SIZE_KB": w, "hbm_bytes_corrected": 2 * f * 1024 + w * 1024,
               "note": "FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 reports half of wide coalesced reads); algorithmic bytes = sum over (model, sequence) pairs of (L + 12)",
               "valu_insts": valu, "lds_insts": lds, "ssv_ms_under_pmc": ms,
               "all_kernels": {"valu_insts": va, "FETCH_SIZE_KB": fa, "WRITE_SIZE_KB": wa, "hbm_bytes_corrected": 2 * fa * 1024 + wa * 1024,
                               "note": "every kernel of the step (SSV + the chains + ensembles + copies), same passes; the gene_* / orf_* kernels of bench.py's gene-calling side legs, which ran in the same process, are left out"}},
              open(dst(what + "ssv_traffic.json"), "w"), indent=1)


def one(tag):
    # (CKM_PROFILE_SRC / CKM_PROFILE_DST: other directories than gpurun_out/ and profiles/ -- tests/test_tools.py)
    src = os.path.join(os.environ.get("CKM_PROFILE_SRC", os.path.join(ROOT, "gpurun_out")), tag)
    out_dir = os.environ.get("CKM_PROFILE_DST", os.path.join(ROOT, "profiles"))
    dst = lambda name: os.path.join(out_dir, "%s_%s" % (tag, name))
    for name in sorted(os.listdir(src)):
        m = re.match(r"bench(\d+)\.json$", name)
        if m and os.path.getsize(os.path.join(src, name)) > 0:
            json.dump(last_json(os.path.join(src, name)), open(dst("bench%s_line.json" % m.group(1)), "w"), indent=1)
    for name, out in (("pytest_tail.txt", "pytest_gpu_tail.txt"), ("smoke.txt", "smoke.txt"), ("timeline3.txt", "timeline_cfg3.txt"), ("valu_rates.txt", "valu_rates.txt")):
        if os.path.exists(os.path.join(src, name)):
            shutil.copyfile(os.path.join(src, name), dst(out))
    ks = os.path.join(src, "stats3", "cfg3_kernel_stats.csv")
    if os.path.exists(ks):
        fam, inst = collections.defaultdict(lambda: [0, 0.0]), []
        for r in csv.DictReader(open(ks)):
            m = re.search(r'ckm::([a-z0-9_]+kernel(?:_h8|_h)?)(<[^>]*>)?', r["Name"])
            k = m.group(1) if m else r["Name"][:40]
            fam[k][0] += int(r["Calls"]); fam[k][1] += float(r["TotalDurationNs"])
            inst.append(((m.group(1) + (m.group(2) or "")) if m else r["Name"][:40], int(r["Calls"]), float(r["TotalDurationNs"]), float(r["AverageNs"])))
        tot = sum(v[1] for v in fam.values())
        line = last_json(os.path.join(src, "stats3.json"))
        with open(dst("cfg3_kernel_stats.txt"), "w") as f:
            f.write("# rocprofv3 --kernel-trace --stats -- python bench.py --config cfg3 --bins-total %d --steps 1 --warmup 1 --no-cpu-baseline --no-cfg2 --no-emulation --no-verify (MI355X)\n"
                    "# = the warm pass over 128 bins + ONE timed step (bench line of this run: %.0f ms per step, SSV launches %.0f ms by HIP events).\n"
                    "# Kernels of the two scan lanes and of the chain streams overlap in time: durations SUM to more than the wall time.\n"
                    % (line["config"]["bins_total"], line["ms_per_step"], line["roofline"]["ms_per_step_kernel"]))
            f.write("%-28s %8s %12s %12s %6s\n" % ("kernel family", "calls", "total_ms", "avg_us", "pct"))
            for k, v in sorted(fam.items(), key=lambda kv: -kv[1][1]):
                f.write("%-28s %8d %12.2f %12.1f %6.1f\n" % (k, v[0], v[1] / 1e6, v[1] / v[0] / 1e3, 100 * v[1] / tot))
            f.write("\n# per template instance (top 40)\n%-34s %8s %12s %12s\n" % ("kernel", "calls", "total_ms", "avg_us"))
            for n, c, t, a in sorted(inst, key=lambda x: -x[2])[:40]:
                f.write("%-34s %8d %12.2f %12.1f\n" % (n, c, t / 1e6, a / 1e3))
    if os.path.exists(os.path.join(src, "pmc3_sq.json")):
        line3 = last_json(os.path.join(src, "pmc3_sq.json"))
        nb3 = int(line3["config"].get("bins_total", 48))
        pmc_files(src, dst, "pmc3_", "cfg3_", line3,
                  "CKM_BENCH_SKIP_WARM=1 python bench.py --config cfg3 --bins-total %d --steps 1 --warmup 0 --no-cpu-baseline --no-cfg2 --no-emulation --no-verify   (%d of the 1000 bins of configs[2], default batching)" % (nb3, nb3))
    if os.path.exists(os.path.join(src, "pmc_sq.json")):
        pmc_files(src, dst, "pmc_", "", last_json(os.path.join(src, "pmc_sq.json")),
                  "CKM_BENCH_STEADY=0 CKM_BENCH_FROM_HOST=0 python bench.py --config cfg2 --steps 1 --warmup 0 --no-cpu-baseline --no-verify")


if __name__ == "__main__":
    for t in sys.argv[1:]:
        one(t)
