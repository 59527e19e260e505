"""Shader clock while ONE gene-calling call runs (the dynamic program is latency-bound and keeps few wavefronts busy: does the device
clock down?).  Usage: python tools/gene_clock_probe.py [bins=48]"""
import glob
import os
import sys
import threading
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch                                           # noqa: E402
from checkm_amd import _lib, runtime                  # noqa: E402
from synthdata import synth_genome as sg              # noqa: E402

nbins = int(sys.argv[1]) if len(sys.argv) > 1 else 48
pr = torch.cuda.get_device_properties(0)
addr = "%04x:%02x:%02x.0" % (pr.pci_domain_id, pr.pci_bus_id, pr.pci_device_id)
hw = sorted(glob.glob(os.path.join("/sys/bus/pci/devices", addr, "hwmon", "hwmon*")))
f_clk = os.path.join(hw[0], "freq1_input") if hw else None
f_pow = os.path.join(hw[0], "power1_average") if hw else None
uniq = [[s for _c, s in sg.make_genome(5000 + u, n_contigs=20, contig_len=(80000, 120000), gc=0.35 + 0.03 * u, sd_frac=0.6)] for u in range(6)]
bins = [uniq[b % 6] for b in range(nbins)]
ctx = runtime.get_ctx()
_lib.call_genes(ctx, bins[:4], 11)
samples, stop = [], [False]


def loop():
    while not stop[0]:
        try:
            samples.append((time.perf_counter(), float(open(f_clk).read()) * 1e-6, float(open(f_pow).read()) * 1e-6 if f_pow and os.path.exists(f_pow) else -1))
        except Exception:
            pass
        time.sleep(0.02)


th = threading.Thread(target=loop, daemon=True); th.start()
time.sleep(0.3)
for r in range(3):
    t0 = time.perf_counter()
    cols, per_bin, st = _lib.call_genes(ctx, bins, 11)
    t1 = time.perf_counter()
    inside = [s for s in samples if t0 <= s[0] <= t1]
    print("rep %d: %.3f s, dp_train %.0f ms; sclk MHz during the call: min %.0f mean %.0f max %.0f; power W mean %.0f (%d samples)" % (
        r, t1 - t0, st["ms_dp_train"], min(s[1] for s in inside), sum(s[1] for s in inside) / len(inside), max(s[1] for s in inside), sum(s[2] for s in inside) / len(inside), len(inside)))
stop[0] = True
idle = [s for s in samples[:10]]
print("before the calls: sclk MHz", [round(s[1]) for s in idle][:6])
