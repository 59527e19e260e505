"""Reduce the rocprofv3 passes of tools/gene_profile.py (one ckm_genes_call over 48 bins of 2 Mb) to a text summary:
   python tools/gene_profile_summary.py <dir with stats/ and pmc/> > profiles/<tag>_gene_dp_pmc.txt"""
import collections
import csv
import glob
import os
import re
import sys

d = sys.argv[1]


def short(n):
    m = re.search(r"ckm::(?:gene::)?([a-z0-9_]+kernel)(<[^>]*>)?", n)
    if m and not n.startswith("void ckm::gene::g_map"):
        return m.group(1) + (m.group(2) or "")
    if "g_map_waves_kernel" in n:
        return "g_map_waves_kernel<lambda> (a walk per wavefront)"
    if "g_map_kernel" in n:
        return "g_map_kernel<lambda> (thread per index)"
    return n[:48]


def lam(n):
    """the lambdas of gene_pipe.h one by one: `map #17` = the 17th lambda(size_t) of gene_pipeline, `map name#1` = the first lambda of a helper"""
    if "g_map" not in n:
        return None
    tags = re.findall(r"\{lambda\(([^)]*)\)#(\d+)\}", n)
    kind = "waves" if "g_map_waves" in n else "map"
    if not tags:
        return kind + " ?"
    outer = [t for t in tags if "Nodes" in t[0]]
    inner = [t for t in tags if "Nodes" not in t[0]]
    if outer:
        return "%s helper(%s)#%s . #%s" % (kind, "Nodes, size_t" + (", int" if "int" in outer[0][0] else ""), outer[0][1], inner[0][1] if inner else "?")
    return "%s #%s" % (kind, inner[0][1])


print("# one ckm_genes_call (table 11) over 48 synthetic bins of 2 Mb, tools/gene_profile.py; rocprofv3 --kernel-trace --stats and a separate --pmc pass (SQ counters, quad-cycle units for *_CYCLES / WAIT / ACTIVE)")
ks = glob.glob(os.path.join(d, "stats", "*kernel_stats.csv"))
if ks:
    fam = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(ks[0])):
        k = short(r["Name"])
        fam[k][0] += int(r["Calls"]); fam[k][1] += float(r["TotalDurationNs"])
    tot = sum(v[1] for v in fam.values())
    print("\n%-52s %8s %12s %12s %6s" % ("kernel (two calls: a warm one and a timed one)", "calls", "total_ms", "avg_us", "pct"))
    for k, v in sorted(fam.items(), key=lambda kv: -kv[1][1])[:16]:
        print("%-52s %8d %12.2f %12.1f %6.1f" % (k, v[0], v[1] / 1e6, v[1] / v[0] / 1e3, 100 * v[1] / tot))
pc = glob.glob(os.path.join(d, "pmc", "*counter_collection.csv"))
if pc:
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    waves = collections.defaultdict(float)
    for r in csv.DictReader(open(pc[0])):
        k = short(r["Kernel_Name"])
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
    print("\n%-52s %14s %14s %14s %14s %14s %14s" % ("kernel (one call)", "INSTS_VALU", "INSTS_SALU", "INSTS_LDS", "WAVE_CYCLES", "ACTIVE_ANY", "WAIT_ANY"))
    for k, v in sorted(agg.items(), key=lambda kv: -kv[1].get("SQ_WAVE_CYCLES", 0))[:10]:
        print("%-52s %14.4g %14.4g %14.4g %14.4g %14.4g %14.4g" % (k, v.get("SQ_INSTS_VALU", 0), v.get("SQ_INSTS_SALU", 0), v.get("SQ_INSTS_LDS", 0), v.get("SQ_WAVE_CYCLES", 0), v.get("SQ_ACTIVE_INST_ANY", 0), v.get("SQ_WAIT_ANY", 0)))
    lagg = collections.defaultdict(lambda: collections.defaultdict(float))
    for r in csv.DictReader(open(pc[0])):
        k = lam(r["Kernel_Name"])
        if k:
            lagg[k][r["Counter_Name"]] += float(r["Counter_Value"]); lagg[k]["launches"] += 1.0 / max(1, len(set(agg[short(r["Kernel_Name"])].keys())))
    if lagg:
        print("\n%-44s %14s %14s %14s %14s" % ("thread-per-index kernels, lambda by lambda", "INSTS_VALU", "WAVE_CYCLES", "ACTIVE_ANY", "WAIT_ANY"))
        for k, v in sorted(lagg.items(), key=lambda kv: -kv[1].get("SQ_WAVE_CYCLES", 0))[:14]:
            print("%-44s %14.4g %14.4g %14.4g %14.4g" % (k, v.get("SQ_INSTS_VALU", 0), v.get("SQ_WAVE_CYCLES", 0), v.get("SQ_ACTIVE_INST_ANY", 0), v.get("SQ_WAIT_ANY", 0)))
    for k, v in agg.items():
        if k.startswith("gene_dp_kernel") and v.get("SQ_WAVE_CYCLES"):
            print("\n%s: waiting (s_waitcnt / barrier) %.0f %% of the wavefronts' cycles, issuing %.0f %%" % (k, 100 * v["SQ_WAIT_ANY"] / v["SQ_WAVE_CYCLES"], 100 * v["SQ_ACTIVE_INST_ANY"] / v["SQ_WAVE_CYCLES"]))
for f in ("plain.txt",):
    p = os.path.join(d, f)
    if os.path.exists(p):
        print("\n# phase times of the call (CKM_TRACE=1) and its kernel times by HIP events")
        print("".join(ln for ln in open(p) if ln.startswith("ckm-trace genes") or ln.startswith("rep "))[-2500:])
