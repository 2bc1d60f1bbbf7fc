"""Generates tests/golden/config5_sampled.npz: config 5 (RLZ 1023 x 512 x 128, 67 M points, 6 variables)
stepped TWICE by the C oracle at FULL size, reduced to a sample that fits a repository:

  * `A_cols`       the spectral state (A coefficients) of `ncols` (z-mode, wavenumber block) columns, all 344 radial nodes,
                   all variables, and the per-variable max |A| over the WHOLE array (the scale of the 1e-10 bar);
  * `pt_idx/pt_val` the oracle's evaluation of its own coefficients (all 7 derivative slots) at sampled points of the cells
                   0 (innermost), 84 (last truncated), 85 (first full-spectrum) and 340 (outermost);
  * `e_orc`        for the sampled rings, the fp64 oracle's derivative-slot error against the EXTENDED-precision evaluation
                   of its own coefficients (tests/cases.py::check_full's criterion: the HIP path must be no less accurate).

This is the oracle (a restatement; Julia and Springsteel are absent here), run once - about 10 minutes and 40 GB on 8 cores:
    python tests/golden/make_config5_fixture.py
The GPU test that reads it: tests/test_gpu_configs.py::test_config5_full_size_against_the_sampled_oracle_fixture."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

CELLS = (0, 84, 85, 340)
RINGS = (0, 253, 254, 256, 1022)      # innermost; the last truncated ring (kmax 254); the first two full-spectrum ones; outermost
NCOLS, NPTS, STEPS, SEED = 48, 192, 2, 20261004


def main():
    from oracle import oracle_c as OC
    from tests import cases
    case = cases.config5_case()
    t0 = time.time()
    orc = cases.OracleModel(case)
    g = orc.g
    for s in range(STEPS):
        orc.step()
        print("step %d done, %.0f s" % (s + 1, time.time() - t0), flush=True)
    A = np.asarray(orc.A)                                            # [S_patch, V], node fastest
    rng = np.random.default_rng(SEED)
    cols = np.stack([rng.integers(0, g.b_zDim, NCOLS), rng.integers(0, g.K2, NCOLS)], axis=1)
    cols[0], cols[1], cols[2] = (0, 0), (0, 1), (g.b_zDim - 1, g.K2 - 1)
    A_cols = np.zeros((NCOLS, g.b_rDim, g.V))
    for i, (zm, blk) in enumerate(cols):
        for v in range(g.V):
            A_cols[i, :, v] = A[:, v].reshape(g.b_zDim, g.K2, g.b_rDim)[zm, blk]
    A_scale = np.abs(A).max(axis=0)
    pt_idx, pt_val, pt_scale = [], [], np.zeros((g.V, g.D))
    ring_phys = {}
    for cell in CELLS:
        ref = OC.TileOracle(g, cell, 1).inverse(A)                  # [3 rings x 512 x 128, V, D]
        base = int(g.ringstart[3 * cell]) * g.zDim
        pick = np.sort(rng.choice(ref.shape[0], NPTS, replace=False))
        pt_idx.append(base + pick)
        pt_val.append(ref[pick])
        pt_scale = np.maximum(pt_scale, np.abs(ref).max(axis=0))
        for r in (3 * cell, 3 * cell + 1, 3 * cell + 2):
            if r in RINGS:
                n = int(g.L[r]) * g.zDim
                o = (int(g.ringstart[r]) - int(g.ringstart[3 * cell])) * g.zDim
                ring_phys[r] = ref[o:o + n]
        print("cell %d done, %.0f s" % (cell, time.time() - t0), flush=True)
    e_orc = cases.slot_errors_vs_extended_rings(g, ring_phys, A, list(RINGS))
    print("oracle vs extended precision on rings %s:" % (RINGS,), e_orc, flush=True)
    out = os.path.join(ROOT, "tests", "golden", "config5_sampled.npz")
    np.savez_compressed(out, cols=cols, A_cols=A_cols, A_scale=A_scale, pt_idx=np.concatenate(pt_idx), pt_val=np.concatenate(pt_val),
                        pt_scale=pt_scale, rings=np.array(RINGS), e_orc=e_orc, cells=np.array(CELLS), steps=STEPS,
                        ts=case["ts"], shape=np.array([341, 512, 128]))
    print("wrote %s (%.0f KB) in %.0f s" % (out, os.path.getsize(out) / 1024, time.time() - t0))


if __name__ == "__main__":
    main()
