import sys, numpy as np
sys.path.insert(0, ".")
from tests import cases
for maker, kw in [(cases.kat_r, {}), (cases.rz_advection, {}), (cases.rl_slab, {"num_cells": 6}), (cases.rlz_hrbl, {"num_cells": 8, "zDim": 32, "ring_L": 32})]:
    case = maker(**kw)
    ref = cases.OracleModel(case); hip = cases.HipModel(case, storage="f32")
    for n in range(3):
        ref.step(); hip.step()
    a, b = hip.physical(), ref.physical()
    print(case["name"])
    for d in range(a.shape[2]):
        print("  slot", d, ["%.1e" % (np.abs(a[:, v, d] - b[:, v, d]).max() / max(np.abs(b[:, v, d]).max(), 1e-300)) for v in range(a.shape[1])])
