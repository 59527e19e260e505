"""Mean duration of each phase of ckm_genes_call over the calls of a pass, from the CKM_TRACE=1 lines on stderr:
   python tools/gene_phase_means.py <stderr file> [first call id to count]"""
import collections
import re
import sys

first = int(sys.argv[2]) if len(sys.argv) > 2 else 0
ph = collections.defaultdict(list)
last, order = {}, []
for ln in open(sys.argv[1]):
    m = re.match(r"ckm-trace genes call (\d+) table (\d+)\s+([\d.]+) ms  (.*)", ln)
    if not m:
        continue
    c, t, lab = int(m.group(1)), float(m.group(3)), m.group(4).strip()
    if c < first:
        continue
    ph[lab].append(t - last.get(c, 0.0))
    last[c] = t
    if lab not in order:
        order.append(lab)
tot = 0.0
for lab in order:
    v = ph[lab]
    print("%-40s n=%3d mean %8.1f ms  max %8.1f" % (lab, len(v), sum(v) / len(v), max(v)))
    tot += sum(v) / len(v)
print("sum of means %.1f ms" % tot)
