"""Phase stamps of the native-ring inverse DFT kernel (diagnostic build; see profiles/phases.py):  SX_DFT_PHASES_OUT=...
   per workgroup, as seen by wave 0: [0] total cycles, [1] staging of the coefficient sets (loads .. barrier), [2] matrix-core
   loops, [3] result stores, [4] row tiles wave 0 did, [5] ring length, [6] MFMAs wave 0 issued, [7] total in 100 MHz real-time ticks."""
import sys
import numpy as np

a = np.fromfile(sys.argv[1], dtype=np.int64).reshape(-1, 8)
a = a[a[:, 0] > 0]
print("workgroups %d" % len(a))
for lo, hi in ((4, 256), (260, 512), (516, 768), (772, 1024)):
    b = a[(a[:, 5] >= lo) & (a[:, 5] <= hi)]
    if not len(b):
        continue
    tot = b[:, 0].mean()
    print("ring length %4d..%4d: %5d workgroups, cycles/workgroup mean %8.0f | staging %4.1f %%  matrix-core loops %4.1f %%  stores %4.1f %%  other (barrier waits, set-up) %4.1f %% | row tiles (wave 0) %.1f"
          % (lo, hi, len(b), tot, 100 * b[:, 1].mean() / tot, 100 * b[:, 2].mean() / tot, 100 * b[:, 3].mean() / tot,
             100 * (b[:, 0] - b[:, 1] - b[:, 2] - b[:, 3]).mean() / tot, b[:, 4].mean()))
    c = b[b[:, 6] > 0]
    print("      wave 0: %.1f cycles of its loop time per MFMA it issued; shader clock %.2f GHz (cycle counter / 100 MHz real-time counter)"
          % (c[:, 2].sum() / c[:, 6].sum(), c[:, 0].sum() / (c[:, 7].sum() * 10.0) ))
