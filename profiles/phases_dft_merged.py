"""Phase stamps of the merged-pass native-ring inverse DFT kernel (diagnostic build): SX_DFT_PHASES_OUT=...
   per workgroup, as seen by wave 0: [0] total cycles, [1] staging of the passes' coefficient sets (loads .. barrier), [2] its units
   (matrix-core loops + result stores), [3] waits at the barrier in front of a pass (the other waves' units), [4] units of wave 0, [5] ring length."""
import sys
import numpy as np

a = np.fromfile(sys.argv[1], dtype=np.int64).reshape(-1, 8)
a = a[a[:, 0] > 0]
loops = a[:, 4] >> 8
a[:, 4] &= 255
print("workgroups %d, sum of workgroup cycles %.3e" % (len(a), a[:, 0].sum()))
if a[:, 6].sum() > 0:      # 100 MHz real-time ticks per workgroup: shader clock while the kernel runs, and the span of the launch
    print("shader clock %.2f GHz (cycles / real time over all workgroups); first start .. last end %.3f ms; sum of workgroup times / 256 CUs %.3f ms"
          % (a[:, 0].sum() / (a[:, 6].sum() * 10.0), (a[:, 7] + a[:, 6]).max() * 1e-5 - a[:, 7].min() * 1e-5, a[:, 6].sum() * 1e-5 / 256))
for lo, hi in ((4, 256), (260, 512), (516, 768), (772, 1024)):
    b = a[(a[:, 5] >= lo) & (a[:, 5] <= hi)]
    if not len(b):
        continue
    tot = b[:, 0].mean()
    print("ring length %4d..%4d: %5d workgroups, cycles/workgroup mean %8.0f (%4.1f %% of all workgroup cycles) | staging %4.1f %%  units of wave 0 %4.1f %%  "
          "barrier in front of a pass %4.1f %%  other (twiddle table, set-up, last barrier) %4.1f %% | units (wave 0) %.1f"
          % (lo, hi, len(b), tot, 100 * b[:, 0].sum() / a[:, 0].sum(), 100 * b[:, 1].mean() / tot, 100 * b[:, 2].mean() / tot, 100 * b[:, 3].mean() / tot,
             100 * (b[:, 0] - b[:, 1] - b[:, 2] - b[:, 3]).mean() / tot, b[:, 4].mean())
          + ("" if loops.sum() == 0 else "  | matrix-core loops %4.1f %% of the units' time" % (100.0 * loops[(a[:, 5] >= lo) & (a[:, 5] <= hi)].sum() / max(b[:, 2].sum(), 1))))
