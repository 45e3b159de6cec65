"""Phase overlap inside a CU for csrc/stem_x3.hip (variant built with -DSUO_SX_PROF; launch 8 of tools/bench_stem.py dumps the stamps):
   SUO_HIP_LIB=suo_slam_amd/variants/libsuo_hip_sxprof.so SUO_SX_PROF_OUT=/tmp/sx.bin python tools/bench_stem.py 256 && python tools/stem_phases.py /tmp/sx.bin"""
import sys

import numpy as np

a = np.fromfile(sys.argv[1], dtype=np.int64).reshape(-1, 6)
t = a[:, :5]
hw = (a[:, 5] >> 32) & 0xFFFFFFFF
xcc = a[:, 5] & 0xF
cu = (hw >> 8) & 0xF
sh = (hw >> 12) & 0x1
se = (hw >> 13) & 0x7
key = ((xcc * 8 + se) * 2 + sh) * 16 + cu
t0 = t[:, 0].min()
d = np.diff(t, axis=1)
print("workgroups", len(a), " distinct CUs", len(np.unique(key)), " kernel span (ticks)", t[:, 4].max() - t0)
print("phase ticks (mean / p10 / p90): stage %d / %d / %d   products %d / %d / %d   patch %d / %d / %d   stores %d / %d / %d   total %d" % (
    d[:, 0].mean(), np.percentile(d[:, 0], 10), np.percentile(d[:, 0], 90), d[:, 1].mean(), np.percentile(d[:, 1], 10), np.percentile(d[:, 1], 90),
    d[:, 2].mean(), np.percentile(d[:, 2], 10), np.percentile(d[:, 2], 90), d[:, 3].mean(), np.percentile(d[:, 3], 10), np.percentile(d[:, 3], 90), (t[:, 4] - t[:, 0]).mean()))
# per CU: at a sample of instants, how many resident workgroups are in each phase
res = np.zeros((6, 6), dtype=np.int64)       # [n in stage][n in products]
conc = []
for k in np.unique(key)[:64]:
    w = t[key == k]
    lo, hi = w[:, 0].min(), w[:, 4].max()
    for x in np.linspace(lo + (hi - lo) * 0.1, lo + (hi - lo) * 0.9, 200):
        live = (w[:, 0] <= x) & (x < w[:, 4])
        ns = int(((w[:, 0] <= x) & (x < w[:, 1])).sum())
        nm = int(((w[:, 1] <= x) & (x < w[:, 2])).sum())
        conc.append(live.sum())
        res[min(ns, 5), min(nm, 5)] += 1
print("resident workgroups per CU (mean):", np.mean(conc))
print("share of instants by (workgroups staging, workgroups in the product loop):")
tot = res.sum()
for i in range(6):
    print("  staging=%d: " % i + "  ".join("%5.1f%%" % (100.0 * res[i, j] / tot) for j in range(6)))
