"""Phase totals of the sliding-window inner-product kernel k_sbw (diagnostic build): per workgroup, cycles summed over its cells."""
import sys
import numpy as np
a = np.fromfile(sys.argv[1], dtype=np.int64).reshape(-1, 8)
a = a[a[:, 5] > 0]
names = ["ring loads + radial accumulation", "barrier (previous tile consumed)", "LDS write + barrier", "vertical contraction + stores"]
tot = a[:, 5].astype(float)
print("workgroups %d   cells per workgroup: median %d   total cycles: median %.0f mean %.0f" % (len(a), np.median(a[:, 4]), np.median(tot), tot.mean()))
for i, n in enumerate(names):
    print("  %-40s mean %8.0f cycles per workgroup (%4.1f %%), %6.0f per cell" % (n, a[:, i].mean(), 100 * a[:, i].mean() / tot.mean(), (a[:, i] / np.maximum(a[:, 4], 1)).mean()))
