#!/usr/bin/env python
"""Turns what tools/collect_profiles.sh left under gpurun_out/<tag>/ into the files committed under profiles/<tag>_*.
usage: python tools/make_profile_files.py <tag>"""
import csv
import json
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def last_json(path):
    return json.loads(open(path).read().strip().split("\n")[-1])


def main():
    tag = sys.argv[1]
    src = os.path.join(ROOT, "gpurun_out", tag)
    dst = lambda name: os.path.join(ROOT, "profiles", "%s_%s" % (tag, name))
    json.dump(last_json(os.path.join(src, "bench_default.json")), open(dst("bench_cfg2_line.json"), "w"), indent=1)
    for name, out in (("bench_w1.json", "bench_cfg2_workers1_line.json"), ("bench_hostcascade_w3.json", "bench_cfg2_hostcascade_w3_line.json"),
                      ("bench_cfg3.json", "bench_cfg3_line.json")):
        if os.path.exists(os.path.join(src, name)) and os.path.getsize(os.path.join(src, name)) > 0:
            json.dump(last_json(os.path.join(src, name)), open(dst(out), "w"), indent=1)
    if os.path.exists(os.path.join(src, "bench_1000bins.json")):
        json.dump(last_json(os.path.join(src, "bench_1000bins.json")), open(dst("bench_1000bins_line.json"), "w"), indent=1)
    ks = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "rocprof_summary.py"), os.path.join(src, "trace", "bench_results.db"), "4"],
                        capture_output=True, text=True).stdout
    open(dst("bench_cfg2_kernel_stats.txt"), "w").write(
        "# rocprofv3 --kernel-trace -- python bench.py --steps 3 --warmup 2 --no-cpu-baseline --lineage-bins 0   (MI355X, default: device-driven cascade, one lane)\n"
        "# summarised by tools/rocprof_summary.py.  NOTE: the chains of the model-length groups run on up to 14 streams underneath the SSV launches, so\n"
        "# durations of concurrent kernels overlap in time and their SUM (ms_per_step) exceeds the wall time of a step; avg_us is the duration of one\n"
        "# launch while it shares the device.  The serialised figures are in %s_pmc_summary.txt (counter collection runs one kernel at a time).\n" % tag + ks)
    pm = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "pmc_summary.py")] + [os.path.join(src, d) for d in ("pmc_fetch", "pmc_write", "pmc_sq")],
                        capture_output=True, text=True).stdout
    hdr = ("# rocprofv3 --pmc passes, MI355X; ONE search each of: CKM_WS_PER_MP=5 CKM_BENCH_STEADY=0 CKM_BENCH_FROM_HOST=0 python bench.py --steps 1 --warmup 0 --no-cpu-baseline --lineage-bins 0\n"
           "# (cfg2: 43 profiles x 100 bins x 2000 ORFs; separate passes: --pmc FETCH_SIZE | --pmc WRITE_SIZE | --pmc SQ_*, each with --kernel-trace only;\n"
           "#  collected by tools/collect_profiles.sh, summarised by tools/pmc_summary.py)\n"
           "# FETCH_SIZE / WRITE_SIZE are in KB as reported; MI355X_MICROARCH.md (HBM section): FETCH_SIZE reads 1/2 of the bytes of a wide\n"
           "# coalesced streaming read on gfx950 -> corrected HBM read bytes = 2 * FETCH_SIZE * 1024.  WRITE_SIZE is taken as reported.\n"
           "# SQ_*_CYCLES / SQ_WAIT_* / SQ_ACTIVE_* are in quad-cycle units.\n")
    line = [l for l in pm.split("\n") if l.startswith("ssv_kernel ")][0]
    valu = float(re.search(r"SQ_INSTS_VALU=(\d+)", line).group(1)); ms = float(line.split()[2])
    tail = ("\n# ssv_kernel: VALU wave-instructions per SIMD = %.0f / 1024 = %.3e ; kernel time %.2f ms (serialised, under the counters) -> %.2f cycles per VALU\n"
            "# instruction per SIMD at 2.4 GHz; the architectural issue peak is 1 wave64 instruction per 4 cycles per SIMD (64 lanes over a 16-lane SIMD).\n"
            % (valu, valu / 1024, ms, ms * 1e-3 * 2.4e9 / (valu / 1024)))
    open(dst("pmc_summary.txt"), "w").write(hdr + pm + tail)

    def total(d, counter):
        return sum(float(r["Counter_Value"]) for r in csv.DictReader(open(os.path.join(src, d, "p_counter_collection.csv")))
                   if "ssv_kernel" in r["Kernel_Name"] and r["Counter_Name"] == counter)
    f, w = total("pmc_fetch", "FETCH_SIZE"), total("pmc_write", "WRITE_SIZE")

    def total_all(d, counter):
        return sum(float(r["Counter_Value"]) for r in csv.DictReader(open(os.path.join(src, d, "p_counter_collection.csv"))) if r["Counter_Name"] == counter)
    valu_all = total_all("pmc_sq", "SQ_INSTS_VALU")
    f_all, w_all = total_all("pmc_fetch", "FETCH_SIZE"), total_all("pmc_write", "WRITE_SIZE")
    json.dump({"config": "cfg2: 43 profiles x 100 bins x 2000 ORFs, 1 GPU, one search (kernels serialised by counter collection)",
               "kernel": "ssv_kernel<Q> (all launches of one step)", "FETCH_SIZE_KB": f, "WRITE_SIZE_KB": w, "hbm_bytes_corrected": 2 * f * 1024 + w * 1024,
               "note": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate passes; FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 reports half of wide coalesced reads)",
               "valu_insts": valu, "ssv_ms_under_pmc": ms,
               "all_kernels": {"valu_insts": valu_all, "FETCH_SIZE_KB": f_all, "WRITE_SIZE_KB": w_all, "hbm_bytes_corrected": 2 * f_all * 1024 + w_all * 1024,
                               "note": "every kernel of the search (SSV + the chains + ensembles + copies), same passes"}},
              open(dst("ssv_traffic.json"), "w"), indent=1)
    print(open(dst("ssv_traffic.json")).read())


if __name__ == "__main__":
    main()
