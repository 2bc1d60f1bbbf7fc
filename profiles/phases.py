"""Summarise the in-kernel phase stamps of the diagnostic build (make -C scythe.jl_amd/csrc phases):
   SCYTHE_HIP_LIB=profiles/libscythe_hip_phases.so SX_PHASES_OUT=gpurun_out/phases.bin python bench.py --steps 5 --warmup 2 --no-cpu-baseline
   python profiles/phases.py gpurun_out/phases.bin
Stamps per workgroup of k_phys_hrbl_cell (s_memtime, shader cycles): 0 entry, 1 first-phase values in LDS (first 20 node loads
arrived), 2 before the MFMA phase, 3 after it, 4 results visible, 5/6/7 ring 0/1/2 finished (stores issued)."""
import sys
import numpy as np

a = np.fromfile(sys.argv[1], dtype=np.int64).reshape(-1, 8)
a = a[a[:, 0] > 0]
names = ["entry -> X in LDS (wait for 20 node loads + X)", "-> before MFMA (barrier, drag, barrier)", "MFMA phase", "-> results visible",
         "ring 0 (wait for rest of loads, tendencies, stores)", "ring 1", "ring 2"]
d = np.diff(a, axis=1)
tot = a[:, 7] - a[:, 0]
print("workgroups %d   total cycles per workgroup: median %.0f  mean %.0f  p10 %.0f  p90 %.0f" % (len(a), np.median(tot), tot.mean(), np.percentile(tot, 10), np.percentile(tot, 90)))
for i, n in enumerate(names):
    print("  %-55s median %7.0f  mean %7.0f  (%4.1f %%)" % (n, np.median(d[:, i]), d[:, i].mean(), 100 * d[:, i].mean() / tot.mean()))
span = a[:, 7].max() - a[:, 0].min()
print("kernel span %.0f cycles; sum of workgroup times / span = %.1f workgroups in flight on average" % (span, tot.sum() / span))
