"""-m gpu: the HIP path (through the C ABI) against the CPU oracle on identical seeded inputs."""
import numpy as np
import pytest

from tests import cases

pytestmark = pytest.mark.gpu
TOL = 1e-10     # north_star: fields within 1e-10 relative of the CPU reference


def _run(case, nsteps, num_tiles=1, oracle_tiles=None, exchange="a2a", impl="torch"):
    hip = cases.HipModel(case, num_tiles=num_tiles, exchange=exchange, impl=impl)
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


def _values_tight_lambda_derivatives_amplified(case, nsteps, rings, title):
    """For patches with many rings (kmax^2 = 8,100 .. 65,025) the d/dlambda and d2/dlambda2 slots of two correct fp64 runs
    differ by k, k^2 times the last-bit differences of their states - more than 1e-10 of the slot's scale.  State (A
    coefficients), values, d/dr and d2/dr2 are held to 1e-10; every slot is measured against the extended-precision
    evaluation of the run's own coefficients (HIP no less accurate than the oracle) and against the spread between the
    two independently written fp64 oracles (tests/cases.py::check_full)."""
    ref, hip, alt = cases.OracleModel(case), cases.HipModel(case), cases.OracleModel(case, numpy_twin=True)
    for _ in range(nsteps):
        ref.step()
        hip.step()
        alt.step()
    a, b = hip.physical(), ref.physical()
    assert cases.rel_err_per_var(a[:, :, :3], b[:, :, :3]) < TOL
    cases.check_full(hip, ref, rings, title, orc_alt=alt)


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


@pytest.mark.parametrize("ring_L,zDim,cells", [(256, 20, 4), (64, 16, 5), (128, 9, 4), (512, 12, 3)])
def test_rlz_hrbl_fft_rings(ring_L, zDim, cells):
    """Power-of-two uniform rings take the Stockham FFT kernels (incl. partial z-chunks, odd log2 L, and the 512-point
    transform that spans two waves)."""
    assert _run(cases.rlz_hrbl(num_cells=cells, zDim=zDim, ring_L=ring_L), 3) < TOL


@pytest.mark.parametrize("zDim", [64, 32, 128])
def test_rlz_hrbl_mfma_column_operators(zDim):
    """zDim 64 / 32 / 128 take the f64-MFMA column-operator kernel (16 or 8 columns per workgroup, ragged last block)
    and, for 64 / 32 on uniform rings, the node-space ("radial last") inverse."""
    assert _run(cases.rlz_hrbl(num_cells=3, zDim=zDim, ring_L=16), 3) < TOL


def test_rlz_hrbl_node_space_at_128_levels():
    """zDim 128 (config 5's column length): cell-wise kernel with 2 azimuths x 128 levels per workgroup, results of the
    column operators written over their inputs in LDS."""
    case = cases.rlz_hrbl(num_cells=6, zDim=128, ring_L=16)
    case["ts"] = 0.05
    assert _run(case, 3) < TOL


@pytest.mark.parametrize("zDim,ring_L", [(32, 32), (128, 16)])
def test_node_space_inverse_equals_ring_wise_inverse(monkeypatch, zDim, ring_L):
    """Same model with the node-space path switched off (SX_NODE_MODE=0): fields agree to rounding."""
    case = cases.rlz_hrbl(num_cells=8, zDim=zDim, ring_L=ring_L)
    if zDim >= 128:
        case["ts"] = 0.05
    a = cases.HipModel(case)
    monkeypatch.setenv("SX_NODE_MODE", "0")
    b = cases.HipModel(case)
    for _ in range(4):
        a.step()
        b.step()
    assert cases.rel_err_per_var(a.physical(), b.physical()) < 1e-11


def test_vertical_inverse_fused_into_the_node_fft_gives_the_same_fields(monkeypatch):
    """SX_FUSE_ZINV=1 (off by default: measured slower): the node-space units form their coefficient slabs on the
    matrix cores inside the inverse FFT kernel instead of reading k_zinv's output - same fields as the oracle."""
    monkeypatch.setenv("SX_FUSE_ZINV", "1")
    for kw in ({"num_cells": 8, "zDim": 32, "ring_L": 32}, {"num_cells": 6, "zDim": 64, "ring_L": 256}):
        assert _run(cases.rlz_hrbl(**kw), 3) < TOL


@pytest.mark.parametrize("kw", [{"num_cells": 8, "zDim": 32, "ring_L": 32}, {"num_cells": 6, "zDim": 64, "ring_L": 256}])
def test_deferred_diagnostic_variable_is_bit_identical_where_it_can_be_observed(monkeypatch, kw):
    """SX_DEFER_DIAG=1 (opt-in): the diagnostic w of the HRBL set is written before it is read (src/shallowWaterModels.jl:69, 430),
    so its spline coefficients are consumed by output only; sx_advance then sends the five prognostic variables through the
    forward transform and the solve and w's follow when something reads A or B.  Every observable - physical (all slots, w
    included), the A coefficients, the restart blob - is BIT-identical to the run that transforms all six every step."""
    case = cases.rlz_hrbl(**kw)
    ref = cases.HipModel(case)
    monkeypatch.setenv("SX_DEFER_DIAG", "1")
    dfr = cases.HipModel(case)
    for _ in range(5):
        ref.step()
        dfr.step()
    assert np.array_equal(dfr.A, ref.A)
    assert np.array_equal(dfr.physical(), ref.physical())
    for _ in range(3):
        ref.step()
        dfr.step()
    assert np.array_equal(dfr.run.tiles[0].get_state(), ref.run.tiles[0].get_state())
    assert np.array_equal(dfr.physical(), ref.physical())


def test_rlz_hrbl_native_rings_on_the_matrix_core_dft():
    """Native ragged rings with >= 8 levels take the f64-MFMA truncated-DFT kernels (sx_dft.hip): 90 rings of 8..364
    points, four launch classes, partial level chunk (zDim 20)."""
    _values_tight_lambda_derivatives_amplified(cases.rlz_hrbl(num_cells=30, zDim=20), 3, [0, 1, 44, 45, 88, 89],
                                               "RLZ HRBL, 30 cells of native rings x 20 levels (kmax 90), 3 steps")


def test_rl_slab_fft_rings():
    assert _run(cases.rl_slab(ring_L=64), 4) < TOL


def test_rl_slab_512_point_rings_all_wavenumbers():
    """90 cells x 512-point rings: kmax grows to 255, so every bin of the two-wave 512-point FFT carries signal."""
    _values_tight_lambda_derivatives_amplified(cases.rl_slab(num_cells=90, ring_L=512), 3, [0, 1, 134, 135, 268, 269],
                                               "RL slab, 90 cells x 512-point rings (kmax 255), 3 steps")


def test_rlz_advection():
    assert _run(cases.rlz_advection(), 6) < TOL


@pytest.mark.parametrize("maker,kw,steps", [
    (cases.rlz_advection, {"num_cells": 3, "zDim": 4}, 4),                      # the smallest legal grid: 3 cells, 4 Chebyshev levels
    (cases.rlz_advection, {"num_cells": 3, "zDim": 4, "ring_L": 16}, 4),        # ... on the shortest power-of-two ring table
    (cases.rz_advection, {"num_cells": 3, "zDim": 5}, 4),
    (cases.rl_advection, {"num_cells": 3}, 4),
    (cases.rlz_hrbl, {"num_cells": 3, "zDim": 16, "ring_L": 16}, 2),           # one 16-level chunk, every ring on the ring-wise path (kDim 7 < rings 9)
    (cases.rlz_advection, {"num_cells": 4, "zDim": 200}, 2),                    # 200 levels: the non-matrix-core column kernels
    (cases.r_bcs, {"bcl": "R3", "bcr": "R3", "num_cells": 7}, 4),               # 10 nodes, 6 of them fixed: four free coefficients, the fewest a spline class takes
])
def test_edge_shapes(maker, kw, steps):
    """Smallest and largest shapes the ABI accepts: 3 cells (calcTileSizes' minimum), 4 levels (sx_create's minimum), one
    16-level chunk, 200 levels, a spline with four free coefficients.  At 200 levels the Chebyshev d2/dz2 slot of two correct
    fp64 evaluations differs by N^4 eps = 3.5e-7 of its scale: the values are held to 1e-10 there, the slots to 1e-6."""
    case = maker(**kw)
    if kw.get("zDim", 0) < 100:
        assert _run(case, steps) < TOL
        return
    hip, orc = cases.HipModel(case), cases.OracleModel(case)
    for _ in range(steps):
        hip.step()
        orc.step()
    a, b = hip.physical(), orc.physical()
    assert cases.rel_err_per_var(a[:, :, :1], b[:, :, :1]) < TOL
    assert cases.rel_err_per_var(a, b) < 1e-6


def test_too_few_cells_for_the_boundary_conditions_is_refused():
    import scythe_jl_amd as S
    with pytest.raises(S.ScytheHipError, match="too few cells"):
        cases.HipModel(cases.r_bcs(bcl="R3", bcr="R3", num_cells=4))


@pytest.mark.parametrize("maker,kw,ntiles", [(cases.rlz_hrbl, {"num_cells": 9, "zDim": 32, "ring_L": 16}, 3),
                                             (cases.kat_r, {}, 2), (cases.kat_r, {}, 3), (cases.rl_slab, {"num_cells": 9}, 2),
                                             (cases.rl_slab, {"num_cells": 10}, 3), (cases.rlz_hrbl, {"num_cells": 7}, 2),
                                             (cases.rz_semiimplicit, {"num_cells": 9}, 3)])
@pytest.mark.parametrize("exchange", ["gather", "a2a"])
@pytest.mark.parametrize("impl", ["torch", "lib"])
def test_tiles_on_one_gpu_match_single_patch_oracle(maker, kw, ntiles, exchange, impl):
    """Several tile handles on one GPU against the one-patch oracle: "gather" = halo + gather of owned rows + redundant
    solve (the reference's protocol), "a2a" = transposed solve (pack / all-to-all / solve / all-to-all / unpack).
    impl "lib": sx_exchange's own buffers and offset tables, driven through the library's loopback transport
    (sx_exchange_local) - what ncclSend / ncclRecv move between ranks is copied between the handles instead."""
    case = maker(**kw)
    assert _run(case, 4, num_tiles=ntiles, exchange=exchange, impl=impl) < TOL


@pytest.mark.parametrize("maker,kw,ntiles", [(cases.rlz_hrbl, {"num_cells": 24, "zDim": 32, "ring_L": 16}, 3),
                                             (cases.rlz_hrbl, {"num_cells": 20, "zDim": 10}, 2),            # native ragged rings
                                             (cases.kat_r, {}, 2), (cases.kat_r, {}, 3), (cases.kat_r, {}, 8), (cases.kat_r, {}, 10),   # PERIODIC: wrap-around rows; 10 tiles: the 160-row reduced operator
                                             (cases.r_bcs, {"bcl": "R1T0", "bcr": "R1T1", "num_cells": 128}, 16),
                                             (cases.rl_slab, {"num_cells": 20}, 2), (cases.rl_slab, {"num_cells": 30}, 3), (cases.rl_slab, {"num_cells": 31, "ring_L": 16}, 4),
                                             (cases.rl_slab, {"num_cells": 80, "ring_L": 16}, 2),            # 40 unknowns per tile: memory-resident local solve
                                             (cases.rz_semiimplicit, {"num_cells": 21}, 3),
                                             (cases.r_bcs, {"bcl": "R0", "bcr": "R0", "num_cells": 40}, 4),
                                             (cases.r_bcs, {"bcl": "R1T0", "bcr": "R1T1", "num_cells": 40}, 4),
                                             (cases.r_bcs, {"bcl": "R1T2", "bcr": "R2T10", "num_cells": 40}, 3),
                                             (cases.r_bcs, {"bcl": "R2T20", "bcr": "R3", "num_cells": 40}, 4),
                                             (cases.r_bcs, {"bcl": "R3", "bcr": "R1T0", "num_cells": 72}, 8)])
@pytest.mark.parametrize("impl", ["torch", "lib"])
def test_interface_only_solve_on_tiles_matches_single_patch_oracle(maker, kw, ntiles, impl):
    """The interface-only ("partitioned") patch solve - every tile solves its own rows, 10 rows per tile and column go
    through a reduced system (sx_iface.hip; SURVEY.md 8(e)(i), replacing the redundant whole-patch solve of
    src/semiimplicit.jl:285) - against the ONE-patch oracle: every radial boundary-condition class incl. PERIODIC, 2 to 8
    tiles, even and uneven, through the Python-side all-to-all stand-in and through sx_exchange's own buffers (loopback)."""
    case = maker(**kw)
    assert _run(case, 4, num_tiles=ntiles, exchange="iface", impl=impl) < TOL


@pytest.mark.parametrize("maker,kw,ntiles,exchange", [(cases.rlz_hrbl, {"num_cells": 24, "zDim": 32, "ring_L": 16}, 3, "iface"),
                                                      (cases.rlz_hrbl, {"num_cells": 9, "zDim": 10}, 2, "a2a"),          # native rings
                                                      (cases.rz_semiimplicit, {"num_cells": 21}, 3, "iface"),
                                                      (cases.rl_slab, {"num_cells": 20}, 2, "iface")])
def test_patch_spectral_assembled_from_tiles_is_the_one_patch_array(maker, kw, ntiles, exchange):
    """ModelRun.patch_spectral() (what spectral_out_<t>.csv is written from, src/semiimplicit.jl:288-293) in the transposed and
    interface-only modes, where no tile holds the whole patch: the array assembled from the tiles' owned rows must be the
    one-tile run's patchSpectral, and fed back through set_patch_spectral_a + tileTransform! it must reproduce the tiles'
    own physical fields.  Grids with more than one (z-mode, wavenumber) block per variable - on the 1-D grid of the
    notebook case the node axis cannot be confused with anything."""
    import scythe_jl_amd as S
    case = maker(**kw)
    multi, one = cases.HipModel(case, num_tiles=ntiles, exchange=exchange), cases.HipModel(case)
    for _ in range(3):
        multi.step()
        one.step()
    a = multi.run.patch_spectral()
    b = one.run.tiles[0].patchSpectral
    assert a.shape == b.shape
    assert cases.rel_err(a, b) < TOL
    phys = multi.physical()
    g1 = one.run.tiles[0]
    g1.set_patch_spectral_a(a)
    g1.tileTransform_()
    assert cases.rel_err_per_var(g1.physical, phys) < TOL


@pytest.mark.parametrize("maker,kw", [(cases.r_bcs, {"bcl": "R0", "bcr": "R0", "num_cells": 40}), (cases.r_bcs, {"bcl": "R1T0", "bcr": "R1T1", "num_cells": 171}),
                                      (cases.r_bcs, {"bcl": "R1T2", "bcr": "R2T10", "num_cells": 33}), (cases.r_bcs, {"bcl": "R2T20", "bcr": "R3", "num_cells": 64}),
                                      (cases.r_bcs, {"bcl": "R3", "bcr": "R1T0", "num_cells": 7}), (cases.r_bcs, {"bcl": "R1T1", "bcr": "R0", "num_cells": 400}),
                                      (cases.kat_r, {}), (cases.kat_r, {"num_cells": 7}), (cases.kat_r, {"num_cells": 101}),
                                      (cases.rz_semiimplicit, {"num_cells": 21, "zDim": 16}), (cases.rl_slab, {"num_cells": 20}),
                                      (cases.rl_slab, {"num_cells": 12, "ring_L": 64}), (cases.rlz_hrbl, {"num_cells": 9, "zDim": 10})])
def test_parallel_cyclic_reduction_solve_equals_the_serial_cholesky_solve(monkeypatch, maker, kw):
    """splineTransform! two ways on the device, on random B coefficients: the LDS-staged parallel cyclic reduction (k_solve_pcr,
    csrc/sx_pcr.hip: what launches with few right-hand sides take) against the lane-per-column banded Cholesky recurrence (k_solve),
    for every radial boundary-condition class incl. PERIODIC, the k = 0 / k >= 1 class split of RL / RLZ grids and the single column
    per group of RZ grids - and both against the oracle's dense definition of the solve."""
    import scythe_jl_amd as S
    case = maker(**kw)
    gp, mp = cases.hip_params(case)
    monkeypatch.setenv("SX_SOLVE_PCR", "1")
    g1 = S.Grid(gp, mp)
    monkeypatch.setenv("SX_SOLVE_PCR", "0")
    g0 = S.Grid(gp, mp)
    rng = np.random.default_rng(5)
    shared = rng.standard_normal((int(g1.dims.s_patch), g1.V))
    out = []
    for g in (g1, g0):
        g.set_patch_spectral_b(shared)
        g.splineTransform_()
        out.append(g.patchSpectral)
    assert np.abs(out[0]).max() > 0
    assert cases.rel_err(out[0], out[1]) < 1e-13
    ref = cases.oracle_grid(case).spline_transform(np.asfortranarray(shared))      # the numpy oracle's dense definition (Spline1D.SA)
    assert cases.rel_err(out[0], ref) < 1e-12
    g1.close()
    g0.close()


@pytest.mark.parametrize("num_cells,zDim,b_zDim", [(5, 9, None), (21, 33, 20), (40, 64, None), (16, 128, 128), (7, 17, 17), (33, 250, 100)])
def test_rz_fused_matrix_core_transforms_equal_the_general_kernels(monkeypatch, num_cells, zDim, b_zDim):
    """RZ grids: tileTransform! and spectralTransform! through the fused radius-on-the-matrix-cores kernels (csrc/sx_rz.hip,
    the default) against the general vertical + radial kernels (SX_RZ_FUSED=0) on random coefficients / fields: every derivative
    slot and every B coefficient, level and mode counts that are not multiples of the 16 x 16 x 4 tile included."""
    import scythe_jl_amd as S
    case = cases.rz_advection(num_cells=num_cells, zDim=zDim)
    if b_zDim:
        case["grid"]["b_zDim"] = b_zDim
    gp, mp = cases.hip_params(case)
    g1 = S.Grid(gp, mp)
    monkeypatch.setenv("SX_RZ_FUSED", "0")
    g0 = S.Grid(gp, mp)
    rng = np.random.default_rng(9)
    a = rng.standard_normal((int(g1.dims.s_patch), g1.V))
    vals = rng.standard_normal((g1.N, g1.V))
    res = []
    for g in (g1, g0):
        g.set_patch_spectral_a(a)
        g.tileTransform_()
        ph = g.physical
        g.set_physical_values(vals)
        g.spectralTransform_()
        res.append((ph, g.spectral))
    assert np.abs(res[0][0]).max() > 0 and np.abs(res[0][1]).max() > 0
    for d in range(res[0][0].shape[2]):
        assert cases.rel_err(res[0][0][:, :, d], res[1][0][:, :, d]) < 1e-12, d
    assert cases.rel_err(res[0][1], res[1][1]) < 1e-13
    g1.close()
    g0.close()


@pytest.mark.parametrize("num_cells,zDim", [(8, 12), (10, 33), (6, 64), (9, 128), (5, 200)])
def test_semi_implicit_adjustment_on_the_matrix_cores_equals_the_scalar_kernel(monkeypatch, num_cells, zDim):
    """semiimplicit_adjustment (src/semiimplicit.jl:521-597): the four column operators as f64-MFMA products of 16 columns
    (k_semi_mfma, csrc/sx_rz.hip, the default) against the scalar kernel (SX_SEMI_MFMA=0) - Euler, AB2 and AB3 steps, level counts
    that are not multiples of the tile - and against the oracle."""
    case = cases.rz_semiimplicit(num_cells=num_cells, zDim=zDim)
    a = cases.HipModel(case)
    monkeypatch.setenv("SX_SEMI_MFMA", "0")
    b = cases.HipModel(case)
    orc = cases.OracleModel(case)
    for _ in range(5):
        a.step()
        b.step()
        orc.step()
    fa, fb = a.run.tiles[0].var_np1, b.run.tiles[0].var_np1
    assert np.isfinite(fa).all()
    for v in range(fa.shape[1]):
        # (two summation orders of operators whose norm grows as zDim^4)
        assert np.abs(fa[:, v] - fb[:, v]).max() <= 1e-12 * max(1.0, (zDim / 64.0) ** 4) * max(np.abs(fb[:, v]).max(), 1e-300), v
    pa, po = a.physical(), orc.physical()
    # the model fields (beyond 128 levels the column operators' O(zDim^4) norms put two correct fp64 runs further apart than 1e-10)
    assert cases.rel_err_per_var(pa[:, :, :1], po[:, :, :1]) < TOL * max(1.0, (zDim / 128.0) ** 4)
    if zDim <= 33:          # (beyond that the d2/dz2 slot of two fp64 runs differs by N^4 eps: tests/test_gpu_configs.py treats config 3)
        assert cases.rel_err_per_var(pa, po) < TOL


@pytest.mark.parametrize("num_cells,twoway", [(4, False), (9, True), (33, False), (100, False), (130, True)])
def test_rl_quarter_wave_dft_kernels_equal_the_half_ring_kernels(monkeypatch, num_cells, twoway):
    """RL grids on native ragged rings: the quarter-wave matrix-core DFT kernels over one work list (round 4, the default) against
    the half-ring kernels in two ring classes (SX_DFT_RLQ=0), tileTransform! and spectralTransform! on random data - incl. rings whose
    wavenumbers pass through the LDS in several chunks (kmax > 96) and patches beyond kmax 319."""
    import scythe_jl_amd as S
    case = cases.rl_slab(num_cells=num_cells, twoway=twoway)
    gp, mp = cases.hip_params(case)
    g1 = S.Grid(gp, mp)
    monkeypatch.setenv("SX_DFT_RLQ", "0")
    g0 = S.Grid(gp, mp)
    rng = np.random.default_rng(13)
    a = rng.standard_normal((int(g1.dims.s_patch), g1.V))
    vals = rng.standard_normal((g1.N, g1.V))
    res = []
    for g in (g1, g0):
        g.set_patch_spectral_a(a)
        g.tileTransform_()
        ph = g.physical
        g.set_physical_values(vals)
        g.spectralTransform_()
        res.append((ph, g.spectral))
    for d in range(res[0][0].shape[2]):
        assert cases.rel_err(res[0][0][:, :, d], res[1][0][:, :, d]) < 1e-12, d
    assert cases.rel_err(res[0][1], res[1][1]) < 1e-12
    g1.close()
    g0.close()


@pytest.mark.parametrize("maker,kw,env", [(cases.kat_r, {}, {}), (cases.r_bcs, {"bcl": "R1T0", "bcr": "R1T1"}, {}), (cases.rz_semiimplicit, {"num_cells": 9, "zDim": 16}, {}),
                                           (cases.rz_advection, {}, {}), (cases.rl_slab, {"num_cells": 12}, {}), (cases.rl_slab, {"num_cells": 8, "ring_L": 64}, {}),
                                           (cases.rlz_hrbl, {"num_cells": 6, "zDim": 10}, {}), (cases.rlz_hrbl, {"num_cells": 8, "zDim": 32, "ring_L": 32}, {}),
                                           (cases.rlz_hrbl, {"num_cells": 8, "zDim": 32, "ring_L": 32}, {"SX_OVERLAP": "1"}),
                                           (cases.rlz_hrbl, {"num_cells": 8, "zDim": 32, "ring_L": 32}, {"SX_DEFER_DIAG": "1"})])
def test_step_replayed_from_a_hip_graph_is_bit_identical(monkeypatch, maker, kw, env):
    """sx_step with SX_GRAPH=1: from the third step on the launches of a step are captured once per rotation of the tendency
    history and replayed as one hipGraph launch.  Same kernels, same arguments: the fields after 14 steps (Euler, AB2, the three
    captures, then replays of each) are bit-identical to plain launches - also with the second stream of SX_OVERLAP=1 inside the
    capture and with the deferred diagnostic variable, whose coefficients must still follow on demand."""
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    case = maker(**kw)
    plain = cases.HipModel(case)
    monkeypatch.setenv("SX_GRAPH", "1")
    graph = cases.HipModel(case)
    for _ in range(14):
        plain.step()
        graph.step()
    fa, fb = plain.run.tiles[0].var_np1, graph.run.tiles[0].var_np1
    assert np.isfinite(fa).all()
    assert np.array_equal(fa, fb)
    assert np.array_equal(plain.run.tiles[0].patchSpectral, graph.run.tiles[0].patchSpectral)
    assert np.array_equal(plain.physical(), graph.physical())
    # and it keeps stepping correctly after something else used the handle
    for _ in range(4):
        plain.step()
        graph.step()
    assert np.array_equal(plain.run.tiles[0].var_np1, graph.run.tiles[0].var_np1)


def test_step_graph_on_a_user_stream_and_restart(monkeypatch, tmp_path):
    """The capture on a non-default stream (the caller's torch stream), and a run restarted from a checkpoint at t >= 3: its first two
    steps are plain launches (lazily created state must exist before a capture), then it captures - bit-identical to the plain run."""
    import torch
    case = cases.rlz_hrbl(num_cells=8, zDim=32, ring_L=32)
    plain = cases.HipModel(case)
    for _ in range(9):
        plain.step()
    ck = str(tmp_path / "ck.npz")
    plain.run.save_checkpoint(ck)
    for _ in range(9):
        plain.step()
    monkeypatch.setenv("SX_GRAPH", "1")
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        g = cases.HipModel(case)
        g.run.load_checkpoint(ck)
        for _ in range(9):
            g.step()
        st.synchronize()
        assert np.array_equal(plain.run.tiles[0].var_np1, g.run.tiles[0].var_np1)


@pytest.mark.parametrize("num_cells,zDim", [(6, 10), (23, 16), (44, 20), (67, 16), (86, 8), (100, 8)])
def test_merged_pass_native_inverse_dft_equals_the_one_set_per_pass_kernel(monkeypatch, num_cells, zDim):
    """RLZ grids on native ragged rings: the merged-pass inverse DFT kernel (round 4, the default: two coefficient sets and up to
    four planes per pass, plane sets compiled into the loop, the last round's row tiles split by planes over the idle waves)
    against the one-set-per-pass kernel (SX_DFT_MERGE=0): every derivative slot of tileTransform! (the full slot mask) on random
    coefficients, and the fields after 3 steps of the boundary-layer set (the equation set's slot masks: 3 planes for h, ug, vg,
    6 for ub, vb) - ring counts with every remainder of row tiles modulo the 8 waves, and patches too large for two sets in LDS."""
    import scythe_jl_amd as S
    case = cases.rlz_hrbl(num_cells=num_cells, zDim=zDim)
    case["ts"] = 0.2
    gp, mp = cases.hip_params(case)
    g1 = S.Grid(gp, mp)
    monkeypatch.setenv("SX_DFT_MERGE", "0")
    g0 = S.Grid(gp, mp)
    monkeypatch.delenv("SX_DFT_MERGE")
    rng = np.random.default_rng(17)
    a = rng.standard_normal((int(g1.dims.s_patch), g1.V))
    out = []
    for g in (g1, g0):
        g.set_patch_spectral_a(a)
        g.tileTransform_()
        out.append(g.physical)
    for d in range(out[0].shape[2]):
        assert cases.rel_err(out[0][:, :, d], out[1][:, :, d]) < 1e-12, d
    g1.close()
    g0.close()
    m1 = cases.HipModel(case)
    monkeypatch.setenv("SX_DFT_MERGE", "0")
    m0 = cases.HipModel(case)
    for _ in range(3):
        m1.step()
        m0.step()
    f1, f0 = m1.run.tiles[0].var_np1, m0.run.tiles[0].var_np1
    assert np.isfinite(f1).all()
    for v in range(f1.shape[1]):
        assert np.abs(f1[:, v] - f0[:, v]).max() <= 1e-11 * max(np.abs(f0[:, v]).max(), 1e-300), v


@pytest.mark.parametrize("planes", ["2", "2 whole workgroup", "0"])
@pytest.mark.parametrize("num_cells,zDim", [(3, 8), (4, 8), (6, 10), (23, 16), (44, 20), (67, 16), (86, 8), (100, 8)])
def test_eighth_wave_native_inverse_dft_equals_the_one_set_per_pass_kernel(monkeypatch, num_cells, zDim, planes):
    """The eighth-wave units of the merged kernel (the default: even wavenumbers folded once more about the middle of the quarter
    ring, rows l and L/4 - l in one unit of two planes) and its quarter-wave units (SX_DFT_EIGHTH=0) against the one-set-per-pass
    quarter-wave kernel: every slot of tileTransform! on random coefficients and 3 steps of the boundary-layer set.  Ring lengths
    4 .. 404: L/4 odd and even (with and without a self-mirrored row), one to four eighth-ring row tiles, the last round split by
    planes.  The eighth-wave kernel runs as two 256-thread workgroups per CU with one coefficient set and half a twiddle table each where
    those fit 80 KB of LDS (up to 85 cells; the default) and as one 512-thread workgroup with two sets ("whole workgroup": SX_DFT_HALFWG=0,
    and every patch beyond 85 cells)."""
    import scythe_jl_amd as S
    case = cases.rlz_hrbl(num_cells=num_cells, zDim=zDim)
    case["ts"] = 0.2
    gp, mp = cases.hip_params(case)
    if "whole" in planes:
        monkeypatch.setenv("SX_DFT_HALFWG", "0")
    planes = planes.split()[0]
    monkeypatch.setenv("SX_DFT_EIGHTH", str(planes))
    g1 = S.Grid(gp, mp)
    monkeypatch.delenv("SX_DFT_EIGHTH")
    monkeypatch.setenv("SX_DFT_MERGE", "0")
    g0 = S.Grid(gp, mp)
    monkeypatch.delenv("SX_DFT_MERGE")
    rng = np.random.default_rng(23)
    a = rng.standard_normal((int(g1.dims.s_patch), g1.V))
    out = []
    for g in (g1, g0):
        g.set_patch_spectral_a(a)
        g.tileTransform_()
        out.append(g.physical)
    for d in range(out[0].shape[2]):
        assert cases.rel_err(out[0][:, :, d], out[1][:, :, d]) < 1e-12, d
    g1.close()
    g0.close()
    if num_cells < 3:
        return
    monkeypatch.setenv("SX_DFT_EIGHTH", str(planes))
    m1 = cases.HipModel(case)
    monkeypatch.delenv("SX_DFT_EIGHTH")
    monkeypatch.setenv("SX_DFT_MERGE", "0")
    m0 = cases.HipModel(case)
    for _ in range(3):
        m1.step()
        m0.step()
    f1, f0 = m1.run.tiles[0].var_np1, m0.run.tiles[0].var_np1
    assert np.isfinite(f1).all()
    for v in range(f1.shape[1]):
        assert np.abs(f1[:, v] - f0[:, v]).max() <= 1e-11 * max(np.abs(f0[:, v]).max(), 1e-300), v


@pytest.mark.parametrize("num_cells,zDim", [(23, 16), (67, 16)])
def test_eighth_wave_units_with_fp32_stored_derivative_planes(monkeypatch, num_cells, zDim):
    """storage = "f32" on native ragged rings: the eighth-wave units write the derivative planes as fp32 (their float
    instantiation) - same values as the quarter-wave one-set-per-pass kernel's fp32 planes to rounding of the fp32 store, and
    within the declared fp32 bars of the all-fp64 planes."""
    import scythe_jl_amd as S
    case = cases.rlz_hrbl(num_cells=num_cells, zDim=zDim)
    gp32, mp32 = cases.hip_params(case, storage="f32")
    gp64, mp64 = cases.hip_params(case)
    g8 = S.Grid(gp32, mp32)
    monkeypatch.setenv("SX_DFT_MERGE", "0")
    g0 = S.Grid(gp32, mp32)
    monkeypatch.delenv("SX_DFT_MERGE")
    g64 = S.Grid(gp64, mp64)
    rng = np.random.default_rng(29)
    a = rng.standard_normal((int(g8.dims.s_patch), g8.V))
    out = []
    for g in (g8, g0, g64):
        g.set_patch_spectral_a(a)
        g.tileTransform_()
        out.append(g.physical)
        g.close()
    for d in range(out[0].shape[2]):
        assert cases.rel_err(out[0][:, :, d], out[1][:, :, d]) < (1e-12 if d == 0 else 2e-7), d      # value slot fp64, derivative slots one fp32 ulp
        assert cases.rel_err(out[0][:, :, d], out[2][:, :, d]) < (1e-12 if d == 0 else 5e-7), d


def test_interface_only_solve_refuses_tiles_that_are_too_small():
    import scythe_jl_amd as S
    case = cases.rl_slab(num_cells=9)
    with pytest.raises(S.ScytheHipError, match="fewer than 6 free spline coefficients"):
        cases.HipModel(case, num_tiles=3, exchange="iface", impl="lib")


def test_check_nan_sees_a_nan_on_the_node_space_path():
    """checkCFL: a NaN planted in an outer ring (whose `physical` planes the node-space path never writes during
    sx_advance) must be reported after the next step; a clean run reports nothing."""
    import scythe_jl_amd as S
    case = cases.rlz_hrbl(num_cells=8, zDim=32, ring_L=32)
    m = cases.HipModel(case)
    m.step()
    g = m.run.tiles[0]
    assert not g.check_nan()
    pts = S.getGridpoints(g)
    vals = case["ic"](pts.reshape(len(pts), -1))
    vals[-5, 1] = np.nan
    m.run.set_initial_conditions([vals])
    m.run.t = 0
    m.step()
    assert g.check_nan()


# fp32-storage mode: the derivative slots of `physical` and of the node-space transforms are
# stored as fp32; the value slot, all arithmetic, every spectral array, the solve and the time-stepping state stay fp64.
# Declared tolerance against the fp64 oracle after 3 steps: values 1e-6 of the variable's scale (fp32 rounding reaches
# them only through ts * tendency; the largest case is HRBL at zDim 64, where the Chebyshev d/dz operator, norm
# O(zDim^2), acts on the fp32-rounded vertical flux: 3.6e-7; all other cases stay below 5e-9), derivative slots 5e-5 of
# the slot's scale (one fp32 rounding, 6e-8, plus the k^2-amplified echo of the value error in d2/dlambda2).
F32_TOL_VAL, F32_TOL_DER = 1e-6, 5e-5


@pytest.mark.parametrize("maker,kw", [(cases.kat_r, {}),
                                       (cases.rz_advection, {}),
                                       (cases.rl_slab, {"num_cells": 6}),
                                       (cases.rl_slab, {"num_cells": 6, "ring_L": 32}),
                                       (cases.rlz_hrbl, {"num_cells": 4, "zDim": 12}),
                                       (cases.rlz_hrbl, {"num_cells": 8, "zDim": 32, "ring_L": 32}),
                                       (cases.rlz_hrbl, {"num_cells": 6, "zDim": 64, "ring_L": 16}),
                                       (cases.rlz_hrbl, {"num_cells": 4, "zDim": 128, "ring_L": 512})])   # config 5, cut down
def test_fp32_storage_mode_within_declared_tolerance(maker, kw):
    case = maker(**kw)
    if kw.get("zDim", 0) >= 128:
        case["ts"] = 0.05      # 128 Chebyshev levels (0.3 m end spacing): the explicit vertical mixing needs a short step
    ref = cases.OracleModel(case)
    hip = cases.HipModel(case, storage="f32")
    for _ in range(3):
        ref.step()
        hip.step()
    a, b = hip.physical(), ref.physical()
    err_val = cases.rel_err_per_var(a[:, :, :1], b[:, :, :1])
    err_all = cases.rel_err_per_var(a, b)
    assert err_val < F32_TOL_VAL, err_val
    assert 1e-9 < err_all < F32_TOL_DER, err_all      # the lower bound proves the fp32 path really ran


@pytest.mark.parametrize("kw,within", [({"num_cells": 8, "zDim": 32, "ring_L": 32}, True), ({"num_cells": 6, "zDim": 64, "ring_L": 16}, False)])
def test_fp32_spectral_intermediates_measured_against_the_declared_tolerance(kw, within):
    """storage "f32x" (sx_grid_desc.storage_f32 = 2; BASELINE.json configs[4] "fp32 mixed-precision transforms", SURVEY.md
    8(d) item 5 "fp32 storage for physical AND transform intermediates"): besides the derivative planes also the transform
    intermediates - the vertically inverted coefficients and the ring spectra - are stored as fp32, sums accumulated in
    fp64.  Measured against the fp32 mode's declared bars (values 1e-6, derivative slots 5e-5 vs the fp64 oracle, 3 steps):
    at 32 levels it holds them (8e-8 / 4e-5); at 64 levels it does NOT (6e-5 / 2e-3, MI355X) - both intermediates are in
    PHYSICAL space along the Chebyshev column, an fp32 rounding there is white noise in z at 6e-8, and the column's second
    derivative (vertical mixing of the HRBL set) amplifies it by O(zDim^4) on its way back into the state.  That is why
    config 5 (128 levels) runs with storage "f32" - derivative planes only, which feed ts * tendency and nothing else - and
    why this mode is kept as a measurement, not a recommendation."""
    case = cases.rlz_hrbl(**kw)
    ref = cases.OracleModel(case)
    hip = cases.HipModel(case, storage="f32x")
    for _ in range(3):
        ref.step()
        hip.step()
    a, b = hip.physical(), ref.physical()
    err_val = cases.rel_err_per_var(a[:, :, :1], b[:, :, :1])
    err_all = cases.rel_err_per_var(a, b)
    print("\nf32x %s: values %.2e, all slots %.2e" % (kw, err_val, err_all))
    assert np.isfinite(a).all() and err_val > 1e-10            # fp32 intermediates really were in the state path
    if within:
        assert err_val < F32_TOL_VAL and err_all < F32_TOL_DER, (err_val, err_all)
    else:
        assert err_val > F32_TOL_VAL, err_val                  # the measured reason for not using it at 64+ levels
        assert err_val < 1e-3                                  # ... an accuracy loss, not a blow-up


def test_fp32_spectral_intermediates_are_refused_where_the_kernels_do_not_exist():
    import scythe_jl_amd as S
    with pytest.raises(S.ScytheHipError, match="storage_f32 = 2"):
        cases.HipModel(cases.rlz_hrbl(num_cells=4, zDim=12), storage="f32x")          # native rings, 12 levels


def test_fp32_storage_512_point_rings_every_wavenumber_against_the_oracle():
    """Config 5's transform shape where the oracle still steps in seconds: 90 cells x 512-point rings (kmax reaches 255, so
    every bin of the two-wave 512-point FFT carries signal) x 16 levels, fp32-stored derivative planes, 3 steps against the
    fp64 oracle at the declared fp32-mode tolerance."""
    case = cases.rlz_hrbl(num_cells=90, zDim=16, ring_L=512)
    ref = cases.OracleModel(case)
    hip = cases.HipModel(case, storage="f32")
    for _ in range(3):
        ref.step()
        hip.step()
    a, b = hip.physical(), ref.physical()
    err_val = cases.rel_err_per_var(a[:, :, :1], b[:, :, :1])
    err_all = cases.per_slot_errors(a, b)
    print("\nfp32-stored derivative planes, 270 rings x 512 x 16, 3 steps: values %.2e, slots %s" % (err_val, " ".join("%.1e" % e for e in err_all)))
    assert err_val < F32_TOL_VAL, err_val
    assert 1e-9 < err_all.max() < F32_TOL_DER, err_all


@pytest.mark.parametrize("maker,kw,ntiles", [(cases.rlz_hrbl, {"num_cells": 8, "zDim": 32, "ring_L": 32}, 1),
                                             (cases.rz_semiimplicit, {}, 1),
                                             (cases.rl_slab, {"num_cells": 9}, 3)])
def test_checkpoint_restart_continues_bit_identically(tmp_path, maker, kw, ntiles):
    """save_checkpoint after step 5 -> a fresh run that loads it continues exactly like the uninterrupted one."""
    case = maker(**kw)
    a = cases.HipModel(case, num_tiles=ntiles)
    for _ in range(5):
        a.step()
    ck = str(tmp_path / "ck.npz")
    a.run.save_checkpoint(ck)
    for _ in range(4):
        a.step()
    b = cases.HipModel(case, num_tiles=ntiles)
    b.run.load_checkpoint(ck)
    assert b.run.t == 5
    for _ in range(4):
        b.step()
    assert np.array_equal(a.physical(), b.physical())


def test_max_abs_diagnostic_matches_host_reduction():
    m = cases.HipModel(cases.rlz_hrbl(num_cells=5, zDim=10))
    m.step()
    g = m.run.tiles[0]
    assert np.array_equal(g.max_abs(), np.abs(g.var_np1).max(axis=0))


@pytest.mark.parametrize("semi", [True, False])
def test_rz_euler_test_moist_equation_set(semi):
    """Euler_test: moist thermodynamics in the pressure-gradient force, reference-state advection, AI2* adjustment.
    Checked against the numpy oracle (the C port does not carry this equation set)."""
    case = cases.rz_euler(semiimplicit=semi)
    ref = cases.OracleModel(case, numpy_twin=True)
    hip = cases.HipModel(case)
    for _ in range(4):
        ref.step()
        hip.step()
    assert cases.rel_err_per_var(hip.physical(), ref.physical()) < TOL


def test_euler_test_with_reference_state_built_from_a_sounding_file(tmp_path):
    """createModelTile's path: ModelParameters.ref_state_file -> interpolate_reference_file -> ReferenceState on the device
    (Pxi_bar from the reference state), then the same run in the oracle fed with the same profiles."""
    import scythe_jl_amd as S
    case = cases.rz_euler(num_cells=6, zDim=16)
    f = tmp_path / "sounding.txt"
    f.write_text("1000.0 300.0 10.0\n" + "".join("%g %g %g\n" % (a, 300.0 + 4.0e-3 * a, 10.0 * np.exp(-a / 2.0e3))
                                                 for a in np.linspace(250.0, 10500.0, 42)))
    g = dict(case["grid"])
    gp = S.GridParameters(**g)
    mp = S.ModelParameters(ts=case["ts"], equation_set="Euler_test", grid_params=gp, physical_params={"K": 10.0},
                           options={"semiimplicit": True}, ref_state_file=str(f))
    run = S.ModelRun(mp)
    rs = mp.ref_state
    assert rs is not None and 300.0 ** 2 < rs.Pxi_bar < 360.0 ** 2
    pts = S.getGridpoints(run.tiles[0])
    run.set_initial_conditions([case["ic"](pts.reshape(len(pts), -1))])
    case["par"] = dict(K=10.0, Pxi_bar=rs.Pxi_bar, ref_state=dict(sbar=rs.sbar, xibar=rs.xibar, mubar=rs.mubar))
    ref = cases.OracleModel(case, numpy_twin=True)
    for _ in range(3):
        run.step()
        ref.step()
    assert not run.tiles[0].check_nan()
    assert cases.rel_err_per_var(run.physical(), ref.physical()) < TOL
    run.close()


@pytest.mark.parametrize("maker,kw,ntiles", [(cases.rl_slab, {"num_cells": 10}, 3), (cases.rlz_hrbl, {"num_cells": 7}, 2), (cases.kat_r, {}, 2)])
def test_index_maps_reproduce_the_shared_sum(maker, kw, ntiles):
    """sx_index_maps (calcPatchMap / calcHaloMap, src/semiimplicit.jl:79-86) used exactly as the Julia glue of INTEGRATION.md
    uses them - sharedSpectral[patchOwned] .= tileSpectral[tileOwned]; the next tile adds tileSpectral[tileHalo] at the
    previous tile's patchHalo (:320-329) - must give the patch's B coefficients: the spline solve of that host-assembled
    array equals the oracle's A."""
    import scythe_jl_amd as S
    case = maker(**kw)
    orc = cases.OracleModel(case)
    gp, mp = cases.hip_params(case)
    run = S.ModelRun(mp, num_tiles=ntiles, device="cuda", exchange="gather")
    shared = None
    halo_prev = None
    for g in run.tiles:
        pts = S.getGridpoints(g)
        g.set_physical_values(case["ic"](pts.reshape(len(pts), -1)))
        g.spectralTransform_()
        tb = g.spectral                                    # [s_tile, V], reference tile layout
        po, to, ph, th = g.index_maps()
        if shared is None:
            shared = np.zeros((int(g.dims.s_patch), g.V), order="F")
        shared[po - 1, :] = tb[to - 1, :]                  # :323
        if halo_prev is not None:
            idx, vals = halo_prev
            shared[idx - 1, :] += vals                     # :329
        halo_prev = (ph, tb[th - 1, :]) if len(ph) else None
    assert halo_prev is None                               # the last tile owns all its rows
    run.close()
    one = S.ModelRun(mp, num_tiles=1, device="cuda")
    g1 = one.tiles[0]
    g1.set_patch_spectral_b(shared)
    g1.splineTransform_()
    assert cases.rel_err(g1.patchSpectral, orc.A) < TOL
    one.close()


@pytest.mark.gpu
@pytest.mark.parametrize("num_cells,zDim", [(100, 4), (130, 4), (130, 6)])
def test_native_rings_beyond_the_scalar_kernels_with_fewer_than_eight_levels(num_cells, zDim):
    """RLZ grids with fewer than 8 levels take the scalar DFT kernels (most of the MFMA N dimension would be empty) - as far as
    those reach (511 points).  Longer native rings (100 cells: 1,204 points, single-pass matrix-core kernels; 130 cells: 1,564
    points, kmax 389, the chunked ones) run on the matrix cores with a partial level chunk; they used to be refused."""
    case = cases.rlz_advection(num_cells=num_cells, zDim=zDim)
    case["grid"]["xmax"] = 2.5 * num_cells / 4.0
    assert _run(case, 2) < TOL


@pytest.mark.gpu
def test_rings_outside_every_transform_path_fail_loudly():
    """Ring lengths that no azimuthal kernel can take must be refused with a clear message - not fail in a launch with
    'invalid argument', and never compute something else.  (Native patches run up to 426 cells = rings of 5,112 points; the
    limit left is a UNIFORM ring table that is neither a power of two <= 512 nor a multiple of 4 and too long for the scalar
    kernel's LDS staging.)"""
    import scythe_jl_amd as S
    gp = S.GridParameters(geometry="RLZ", xmin=0.0, xmax=3.0e5, num_cells=12, vars={"u": 1}, zmin=0.0, zmax=1.0e3, zDim=16,
                          ring_uniform_L=1026)
    g = S.createGrid(gp)
    try:
        g.set_physical_values(np.zeros((g.N, 1)))
        with pytest.raises(S.ScytheHipError, match="outside every transform path"):
            g.spectralTransform_()
    finally:
        g.close()


@pytest.mark.gpu
def test_two_handles_driven_from_two_host_threads():
    """include/scythe_hip.h: calls on ONE handle are serial, different handles may be driven concurrently (the reference keeps one
    mtile per worker process).  Two models stepped from two Python threads (ctypes drops the GIL inside the library) must give
    bit for bit what each gives alone; the error text is per thread."""
    import threading
    import scythe_jl_amd as S

    def solo(case, steps):
        m = cases.HipModel(case)
        for _ in range(steps):
            m.step()
        out = m.physical().copy()
        m.run.close()
        return out
    ca, cb = cases.rlz_hrbl(num_cells=9, zDim=32, ring_L=32), cases.rl_slab(num_cells=20)
    ra, rb = solo(ca, 6), solo(cb, 40)
    assert np.isfinite(ra).all() and np.isfinite(rb).all()
    got, errs = {}, []

    def work(name, case, steps):
        try:
            got[name] = solo(case, steps)
            # a failing call on this thread must not disturb the other thread's handle or message
            with pytest.raises(S.ScytheHipError, match="too few cells"):
                cases.HipModel(cases.r_bcs(bcl="R3", bcr="R3", num_cells=4))
        except BaseException as e:       # noqa: BLE001 - reported by the main thread
            errs.append((name, e))
    ts = [threading.Thread(target=work, args=("a", ca, 6)), threading.Thread(target=work, args=("b", cb, 40))]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    assert not errs, errs
    assert np.array_equal(got["a"], ra) and np.array_equal(got["b"], rb)


@pytest.mark.gpu
def test_create_destroy_cycles_return_the_device_memory():
    """sx_destroy (and the release of the exchange state) give back everything sx_create, the timers and the exchange buffers took:
    40 create / step / close cycles - one tile, three tiles with each exchange protocol through the library's loopback transport
    - leave the free device memory where it was (hipMemGetInfo; one allocation granule of slack) and the host's resident set too."""
    import torch

    def cycle(i):
        kind = i % 4
        case = cases.rlz_hrbl(num_cells=27, zDim=32, ring_L=64)
        if kind == 0:
            m = cases.HipModel(case)
        else:
            m = cases.HipModel(case, num_tiles=3, exchange=["a2a", "gather", "iface"][kind - 1], impl="lib")
        m.run.tiles[0].enable_timers(True)
        for _ in range(3):
            m.step()
        m.physical()
        m.run.close()
    for i in range(4):                     # first use of every path: code objects, the library's lazily created state
        cycle(i)
    import gc
    import psutil
    torch.cuda.synchronize()
    gc.collect()
    free0, rss0 = torch.cuda.mem_get_info()[0], psutil.Process().memory_info().rss
    for i in range(40):
        cycle(i)
    torch.cuda.synchronize()
    gc.collect()
    free1, rss1 = torch.cuda.mem_get_info()[0], psutil.Process().memory_info().rss
    assert free0 - free1 <= (2 << 20), "device memory lost over 40 cycles: %.1f MB" % ((free0 - free1) / 2 ** 20)
    # host side: operator tables, work lists, timer events (a cycle builds ~15 MB of them; 40 leaked cycles would be 600 MB)
    assert rss1 - rss0 < (100 << 20), "host memory grew by %.0f MB over 40 cycles" % ((rss1 - rss0) / 2 ** 20)
    print("\n40 cycles: device memory %+.2f MB, host RSS %+.1f MB" % ((free1 - free0) / 2 ** 20, (rss1 - rss0) / 2 ** 20))
