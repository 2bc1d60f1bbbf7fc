import os, sys, time; sys.path.insert(0, os.getcwd())
import numpy as np
from oracle import oracle_c as OC
from tests import cases
print("cpu_count", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)), "omp default", OC.lib().orc_num_threads(), flush=True)
try:
    print("cpu.max", open("/sys/fs/cgroup/cpu.max").read().strip())
except Exception as e:
    print("no cpu.max", e)
case = cases.rlz_hrbl(num_cells=40, zDim=64)
case["ts"] = 0.2
for n in (0, 16, 32, 8):
    if n: OC.lib().orc_set_num_threads(n)
    t0 = time.time(); m = cases.OracleModel(case); t1 = time.time(); m.step(); t2 = time.time()
    print("threads", n or "default", "init %.1fs step %.1fs" % (t1 - t0, t2 - t1), flush=True)
