"""-m gpu: the HIP path (through the C ABI) against the CPU oracle on identical seeded inputs."""
import numpy as np
import pytest

from tests import cases

pytestmark = pytest.mark.gpu
TOL = 1e-10     # north_star: fields within 1e-10 relative of the CPU reference


def _run(case, nsteps, num_tiles=1, oracle_tiles=None, exchange="a2a"):
    hip = cases.HipModel(case, num_tiles=num_tiles, exchange=exchange)
    orc = cases.OracleModel(case, tiles=oracle_tiles)
    if hip.A is not None:
        e0 = cases.rel_err(hip.A, orc.A)
        assert e0 < TOL, "initial A coefficients differ: %g" % e0
    for _ in range(nsteps):
        hip.step()
        orc.step()
    a, b = hip.physical(), orc.physical()
    assert np.isfinite(a).all()
    return cases.rel_err_per_var(a, b)


@pytest.mark.parametrize("bcl,bcr", [("R0", "R0"), ("R1T0", "R1T1"), ("R1T2", "R2T10"), ("R2T20", "R3"), ("R3", "R1T0")])
def test_r_grid_boundary_conditions(bcl, bcr):
    assert _run(cases.r_bcs(bcl, bcr), 20) < TOL


def test_r_grid_periodic_kat_config_short():
    assert _run(cases.kat_r(), 50) < TOL


def test_rz_advection():
    assert _run(cases.rz_advection(), 10) < TOL


def test_rz_semiimplicit():
    assert _run(cases.rz_semiimplicit(), 6) < TOL


@pytest.mark.parametrize("ring_L", [None, 16])
def test_rl_advection(ring_L):
    assert _run(cases.rl_advection(ring_L=ring_L), 10) < TOL


@pytest.mark.parametrize("twoway", [False, True])
def test_rl_slab(twoway):
    assert _run(cases.rl_slab(twoway=twoway), 5) < TOL


@pytest.mark.parametrize("ring_L", [None, 32])
def test_rlz_hrbl(ring_L):
    assert _run(cases.rlz_hrbl(ring_L=ring_L), 4) < TOL


@pytest.mark.parametrize("ring_L,zDim,cells", [(256, 20, 4), (64, 16, 5), (128, 9, 4)])
def test_rlz_hrbl_fft_rings(ring_L, zDim, cells):
    """Power-of-two uniform rings take the Stockham FFT kernels (incl. partial z-chunks and odd log2 L)."""
    assert _run(cases.rlz_hrbl(num_cells=cells, zDim=zDim, ring_L=ring_L), 3) < TOL


@pytest.mark.parametrize("zDim", [64, 32])
def test_rlz_hrbl_mfma_column_operators(zDim):
    """zDim 64 / 32 take the f64-MFMA column-operator kernel (16 columns per workgroup, ragged last block) and, on
    uniform rings, the node-space ("radial last") inverse: rings 1..6 ring-wise, rings 7..9 from node transforms."""
    assert _run(cases.rlz_hrbl(num_cells=3, zDim=zDim, ring_L=16), 3) < TOL


def test_node_space_inverse_equals_ring_wise_inverse(monkeypatch):
    """Same model with the node-space path switched off (SX_NODE_MODE=0): fields agree to rounding."""
    case = cases.rlz_hrbl(num_cells=8, zDim=32, ring_L=32)
    a = cases.HipModel(case)
    monkeypatch.setenv("SX_NODE_MODE", "0")
    b = cases.HipModel(case)
    for _ in range(4):
        a.step()
        b.step()
    assert cases.rel_err_per_var(a.physical(), b.physical()) < 1e-11


def test_rl_slab_fft_rings():
    assert _run(cases.rl_slab(ring_L=64), 4) < TOL


def test_rlz_advection():
    assert _run(cases.rlz_advection(), 6) < TOL


@pytest.mark.parametrize("maker,kw,ntiles", [(cases.rlz_hrbl, {"num_cells": 9, "zDim": 32, "ring_L": 16}, 3),
                                             (cases.kat_r, {}, 2), (cases.kat_r, {}, 3), (cases.rl_slab, {"num_cells": 9}, 2),
                                             (cases.rl_slab, {"num_cells": 10}, 3), (cases.rlz_hrbl, {"num_cells": 7}, 2),
                                             (cases.rz_semiimplicit, {"num_cells": 9}, 3)])
@pytest.mark.parametrize("exchange", ["gather", "a2a"])
def test_tiles_on_one_gpu_match_single_patch_oracle(maker, kw, ntiles, exchange):
    """Several tile handles on one GPU against the one-patch oracle: "gather" = halo + gather of owned rows + redundant
    solve (the reference's protocol), "a2a" = transposed solve (pack / all-to-all / solve / all-to-all / unpack)."""
    case = maker(**kw)
    assert _run(case, 4, num_tiles=ntiles, exchange=exchange) < TOL
