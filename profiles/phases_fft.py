"""Phase stamps of the node-space inverse FFT kernel (diagnostic build; see profiles/phases.py):
   SX_FFT_PHASES_OUT=... ; stamps per workgroup: 0 entry, 1 coefficients of the first group arrived and combined, 2 first slot
   staged + transformed, 3 after its LDS barrier, 4 its copy-out issued, 5 second slot transformed, 6 its copy-out issued, 7 end."""
import sys
import numpy as np

a = np.fromfile(sys.argv[1], dtype=np.int64).reshape(-1, 8)
a = a[(a > 0).all(axis=1)]
names = ["entry -> first coefficients combined (setup + load latency)", "stage + 4 FFT passes (slot 0)", "LDS barrier", "copy-out issue (slot 0)",
         "slot 1: combine/stage/FFT", "slot 1: barrier + copy-out issue", "remaining slots"]
d = np.diff(a, axis=1)
tot = a[:, 7] - a[:, 0]
print("workgroups %d   total cycles per workgroup: median %.0f  mean %.0f  p10 %.0f  p90 %.0f" % (len(a), np.median(tot), tot.mean(), np.percentile(tot, 10), np.percentile(tot, 90)))
for i, n in enumerate(names):
    print("  %-62s median %7.0f  mean %7.0f  (%4.1f %%)" % (n, np.median(d[:, i]), d[:, i].mean(), 100 * d[:, i].mean() / tot.mean()))
