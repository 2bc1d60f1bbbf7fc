"""CPU-only: the host-side mirror of the reference interface (no compute)."""
import pytest
import numpy as np

import scythe_jl_amd as S
from scythe_jl_amd.model import bc_name, grid_desc


def test_grid_parameters_derived_fields_match_notebook_instance():
    """GridParameters("R", -50.0, 50.0, 100, 300, 103, 2.0, ..., 1, 103, 0, 300, 0) printed in
    notebooks/LinearAdvection_example.ipynb cell 1."""
    gp = S.GridParameters(geometry="R", xmin=-50.0, xmax=50.0, num_cells=100,
                          BCL={"u": S.CubicBSpline.PERIODIC}, BCR={"u": S.CubicBSpline.PERIODIC}, vars={"u": 1})
    assert (gp.rDim, gp.b_rDim, gp.l_q) == (300, 103, 2.0)
    assert (gp.spectralIndexL, gp.spectralIndexR, gp.patchOffsetL, gp.patchOffsetR, gp.tile_num) == (1, 103, 0, 300, 0)
    assert gp.BCL["u"] == {"PERIODIC": 0} and S.CubicBSpline.R0 == {"R0": 0}
    g64 = S.GridParameters(geometry="RLZ", xmin=0, xmax=1, num_cells=4, zDim=64)
    assert g64.b_zDim == 43
    assert S.GridParameters(geometry="RZ", xmin=0, xmax=1, num_cells=4, zDim=128).b_zDim == 86
    t = S.GridParameters(geometry="R", xmin=0, xmax=1, num_cells=10, spectralIndexL=35)
    assert t.patchOffsetL == 102 and t.spectralIndexR == 47


def test_bc_tags():
    assert bc_name(S.CubicBSpline.R1T0) == "R1T0" and bc_name(S.CubicBSpline.R1T1) == "R1T1"
    assert bc_name(S.CubicBSpline.R2T20) == "R2T20" and bc_name("R3") == "R3"
    gp = S.GridParameters(geometry="RL", xmin=0.0, xmax=3.0e5, num_cells=100,
                          BCL={"h": S.CubicBSpline.R1T1, "u": S.CubicBSpline.R1T0}, BCR={"u": S.CubicBSpline.R1T1},
                          vars={"h": 1, "u": 2})
    d, keep = grid_desc(gp, 10, 20, 3)
    assert list(keep["bcl"]) == [2, 1] and list(keep["bcr"]) == [0, 2] and list(keep["bcl0"]) == [2, 1]
    assert (d.tile_cell0, d.tile_num_cells, d.tile_num, d.nvars) == (10, 20, 3, 2)


def test_model_parameters_defaults():
    mp = S.ModelParameters(grid_params=S.GridParameters())
    assert mp.equation_set == "LinearAdvection1D" and mp.options["semiimplicit"] is False


def test_patch_layout_row_ownership():
    gp = S.GridParameters(geometry="R", xmin=0.0, xmax=10.0, num_cells=10, vars={"u": 1})
    lay = S.PatchLayout(gp, 3, n_cols=7)
    assert lay.ncells == [4, 3, 3] and lay.cell0 == [0, 4, 7] and lay.max_rows == 7
    assert [lay.owned_rows(t) for t in range(3)] == [4, 3, 6]
    off = lay.row_offsets()
    # every patch row is owned exactly once; rows of tile t live in slot t of the gather buffer
    assert len(set(off.tolist())) == 13
    assert list(off[:4] // 7) == [0, 1, 2, 3] and list(off[4:7] // 7) == [7, 8, 9] and list(off[7:] // 7) == [14, 15, 16, 17, 18, 19]


def test_csv_reader_matches_notebook_format(tmp_path):
    """read_physical_grid picks variables by column name from the CSV the notebook writes (r,u / r,l,h,u,v,...)."""
    from scythe_jl_amd.io import read_physical_grid

    class FakeTile:
        def __init__(self, n):
            self.N = n

    class FakeRun:
        num_tiles = 2
        tiles = [FakeTile(4), FakeTile(2)]
        tile_ids = [0, 1]

        class layout:
            tile_sizes = np.array([[0, 0], [0, 0], [0, 0], [0, 0], [4, 2]], dtype=float)

    gp = S.GridParameters(geometry="RL", xmin=0.0, xmax=1.0, num_cells=6, vars={"h": 1, "u": 2})
    path = tmp_path / "ic.csv"
    data = np.arange(24, dtype=float).reshape(6, 4)
    np.savetxt(path, data, delimiter=",", header="r,l,u,h", comments="")      # columns in a different order than vars
    vals = read_physical_grid(str(path), gp, FakeRun)
    assert [v.shape for v in vals] == [(4, 2), (2, 2)]
    assert np.array_equal(vals[0][:, 0], data[:4, 3]) and np.array_equal(vals[1][:, 1], data[4:, 2])


# ----------------------------------------------------------------------------- Chebyshev column ops / reference state
def test_chebyshev_column_ops_match_the_oracle_operators():
    """sx_cheb_column_ops (the library's extended-precision assembly) against the oracle's Cheb class: filtered values,
    first / second derivative and integral from the bottom, with truncation and with vertical BCs."""
    from oracle import oracle_np as O
    for n, bd, bcb, bct in [(12, 12, "R0", "R0"), (20, 0, "R0", "R0"), (16, 11, "R1T0", "R1T1"), (33, 0, "R1T1", "R0")]:
        col = S.Chebyshev1D(0.0, 2.5e3, n, bd, bcb, bct)
        ch = O.Cheb(0.0, 2.5e3, n, bdim=bd or None, bcb=bcb, bct=bct)
        assert np.allclose(col.z, ch.z, rtol=0, atol=1e-9)
        scale = lambda m: np.abs(m).max()
        for mine, ref in [(col._rec, ch.M[0] @ ch.CBm), (col._dz, ch.M[1] @ ch.CBm), (col._dzz, ch.M[2] @ ch.CBm),
                          (col._int, ch.T @ ch.Ic @ ch.CAm @ ch.CBm)]:
            assert np.abs(mine - ref).max() <= 1e-12 * scale(ref)


def _dry_isentropic(gp, theta0=300.0, psfc=1000.0):
    alt = np.linspace(0.0, gp.zmax, 41)
    return S.reference_state.reference_state_from_sounding(gp, psfc, alt, np.full(41, theta0), np.zeros(41))


def test_reference_state_of_a_dry_isentropic_atmosphere():
    """theta = 300 K, no vapour: Exner pressure is linear in z, so p(z), T(z) are known in closed form.
    interpolate_reference_file integrates ln p level by level with the density of the level below and then re-integrates
    once with the Chebyshev column (src/reference_state.jl:70-111) - a first-order scheme with one correction - so the
    result approaches the closed form as the column is refined, while these hold to rounding at any resolution: the
    potential temperature recovered from (sbar, xibar) is 300 K, the surface pressure is the sounding's, mubar =
    bhyp(0) = 0, the derivative columns are the column operators applied to the values, Pxi_bar = mean(gamma Rd T)."""
    T = S.thermodynamics
    err = []
    for nz in (24, 48):
        gp = S.GridParameters(geometry="RZ", xmin=0.0, xmax=1.0e4, num_cells=6, zmin=0.0, zmax=1.0e4, zDim=nz, b_zDim=nz,
                              vars={"s": 1, "xi": 2, "mu": 3, "u": 4, "w": 5})
        rs = _dry_isentropic(gp)
        col = S.Chebyshev1D(0.0, 1.0e4, nz, nz)
        exner = 1.0 - T.gravity * col.z / (T.Cpd * 300.0)
        q_v, rho_d, Tk, p = T.thermodynamic_tuple(rs.sbar[:, 0], rs.xibar[:, 0], rs.mubar[:, 0])
        assert np.abs(Tk * (T.p_0 / p) ** (T.Rd / T.Cpd) - 300.0).max() < 1e-9
        assert abs(p[0] - 1000.0) < 1e-9
        assert np.abs(rs.mubar).max() < 1e-20 and np.abs(q_v).max() < 1e-12      # bhyp(0) = 0 up to rounding of q0 - q0^2 / q0
        col.uMish[:] = rs.xibar[:, 0]
        assert np.abs(col.CIxtransform() - rs.xibar[:, 1]).max() < 1e-12 * np.abs(rs.xibar[:, 1]).max()
        assert np.abs(col.CIxxtransform() - rs.xibar[:, 2]).max() < 1e-9 * np.abs(rs.xibar[:, 2]).max()
        assert abs(rs.Pxi_bar / ((T.Cpd / T.Cvd) * T.Rd * Tk).mean() - 1.0) < 1e-12
        assert rs.packed().shape == (3, 3, nz)
        err.append(np.abs(p / (1000.0 * exner ** (T.Cpd / T.Rd)) - 1.0).max())
    assert err[0] < 0.02 and err[1] < 0.6 * err[0]          # 1.2 % at 24 levels, shrinking with resolution


def test_reference_state_file_readers(tmp_path):
    """interpolate_reference_file parses `psfc theta qv` + `alt theta qv` lines and refuses levels above the sounding
    (src/reference_state.jl:17-68); exact_reference_state checks the level heights (src/reference_state.jl:170-180)."""
    gp = S.GridParameters(geometry="RZ", xmin=0.0, xmax=1.0e4, num_cells=6, zmin=0.0, zmax=8.0e3, zDim=10, b_zDim=10,
                          vars={"s": 1, "xi": 2, "mu": 3, "u": 4, "w": 5})
    f = tmp_path / "sounding.txt"
    f.write_text("1000.0 300.0 14.0\n" + "".join("%g %g %g\n" % (a, 300.0 + 3.0e-3 * a, 14.0 * np.exp(-a / 2.5e3))
                                                 for a in np.linspace(500.0, 9000.0, 18)))
    mp = S.ModelParameters(equation_set="Euler_test", grid_params=gp, ref_state_file=str(f))
    rs = S.reference_state.interpolate_reference_file(mp)
    assert rs.sbar.shape == (10, 3) and np.all(np.isfinite(rs.sbar)) and rs.mubar[0, 0] > rs.mubar[-1, 0] > 0.0
    assert 300.0 ** 2 < rs.Pxi_bar < 360.0 ** 2
    f.write_text("1000.0 300.0 14.0\n500.0 301.0 12.0\n")            # sounding ends below the model top
    with pytest.raises(ValueError):
        S.reference_state.interpolate_reference_file(mp)
    z = S.Chebyshev1D(0.0, 8.0e3, 10, 10).z
    g = tmp_path / "exact.txt"
    g.write_text("".join("%r %r %r %r 0.0\n" % (float(zz), float(a), float(b), float(c))
                         for zz, a, b, c in zip(z, rs.sbar[:, 0], rs.xibar[:, 0], rs.mubar[:, 0])))
    mp2 = S.ModelParameters(equation_set="Euler_test", grid_params=gp, ref_state_file=str(g), options={"exact_reference_state": True})
    ex = S.reference_state.exact_reference_state(mp2)
    assert np.abs(ex.sbar[:, 0] - rs.sbar[:, 0]).max() < 1e-9 * np.abs(rs.sbar[:, 0]).max()
    g.write_text("".join("%r 1.0 1.0 1.0 0.0\n" % float(zz + 1.0) for zz in z))
    with pytest.raises(ValueError):
        S.reference_state.exact_reference_state(mp2)


def test_cost_balanced_tile_split_for_uniform_rings():
    """PatchLayout(split="cost"): contiguous tiles of at least 3 cells that cover the patch; tiles holding the inner
    (ring-wise path) cells get fewer cells; native rings and split="reference" keep calcTileSizes' partition."""
    gp = S.GridParameters(geometry="RLZ", xmin=0.0, xmax=3.0e5, num_cells=171, zmin=0.0, zmax=2.0e3, zDim=64,
                          vars={"h": 1, "u": 2, "v": 3, "ub": 4, "vb": 5, "wb": 6}, ring_uniform_L=256)
    for n in (2, 4, 8):
        ref = S.PatchLayout(gp, n)
        lay = S.PatchLayout(gp, n, split="cost")
        assert sum(lay.ncells) == 171 and min(lay.ncells) >= 3
        assert lay.cell0 == [sum(lay.ncells[:t]) for t in range(n)]
        assert lay.ncells[0] < ref.ncells[0] and lay.ncells[-1] > ref.ncells[-1]
        w = [sum(1.6 if 3 * c < 127 else 1.0 for c in range(c0, c0 + m)) for c0, m in zip(lay.cell0, lay.ncells)]
        assert max(w) - min(w) <= 2 * 1.6 + 1e-9                    # balanced to within a cell or two
        assert int(lay.tile_sizes[4].sum()) == 513 * 256 * 64
    native = S.GridParameters(geometry="RL", xmin=0.0, xmax=3.0e5, num_cells=30, vars={"h": 1})
    assert S.PatchLayout(native, 3, split="cost").ncells == S.PatchLayout(native, 3).ncells
    with pytest.raises(ValueError):
        S.PatchLayout(gp, 2, split="nope")


def test_output_time_tag_is_julias_string_of_round_t_2():
    """write_output names its files string(round(t; digits=2)) (src/io.jl:5): 3 * 0.1 -> "0.3", not 0.30000000000000004."""
    from scythe_jl_amd.io import julia_float_string as j, output_time_tag as tag
    for x, want in {0.0: "0.0", 100.0: "100.0", 0.3: "0.3", 1e6: "1.0e6", 1234567.0: "1.234567e6", 123456.0: "123456.0",
                    1e-5: "1.0e-5", 0.0001: "0.0001", 10800.0: "10800.0", 2.5e-7: "2.5e-7", -3.25: "-3.25", 1e22: "1.0e22"}.items():
        assert j(x) == want, (x, j(x), want)
    assert tag(3 * 0.1) == "0.3" and tag(100.0) == "100.0" and tag(0.0) == "0.0" and tag(86400.004) == "86400.0"
    assert tag(2000 * 0.05) == "100.0"          # the notebook's final file: physical_out_100.0.csv


def test_partitioned_interface_only_patch_solve_equals_the_patch_solve():
    """Numpy statement of the SPIKE-type B -> A solve for radial tiles (oracle/partitioned_np.py): every tile
    solves its own rows, 6 interface unknowns per tile boundary are exchanged - against the one-patch Cholesky solve of the
    reference's matrix (src/semiimplicit.jl:285 via Springsteel SAtransform), all radial boundary-condition classes."""
    from oracle import oracle_np as O
    from oracle.partitioned_np import PartitionedBandedSolve
    import scythe_jl_amd as S
    rng = np.random.default_rng(5)
    for bcl, bcr in (("R0", "R0"), ("R1T0", "R1T1"), ("R1T1", "R0"), ("R2T10", "R1T0"), ("R3", "R0"), ("PERIODIC", "PERIODIC")):
        sp = O.Spline1D(0.0, 3.0e5, 171, bcl=bcl, bcr=bcr)
        n = sp.PQ.shape[0]
        b = sp.G @ rng.standard_normal((sp.bdim, 7))
        ref = np.linalg.solve(sp.PQ, b)
        for N in (2, 4, 8):
            lay = S.PatchLayout(S.GridParameters(geometry="R", xmin=0.0, xmax=3.0e5, num_cells=171, vars={"u": 1}), N)
            bounds = [0] + [min(n - 1, c) for c in lay.cell0[1:]] + [n]
            ps = PartitionedBandedSolve(sp.PQ, bounds)
            got = ps.solve(b)
            assert np.abs(got - ref).max() <= 1e-12 * np.abs(ref).max(), (bcl, bcr, N, np.abs(got - ref).max() / np.abs(ref).max())
            wrap = 6 if bcl == "PERIODIC" else 0
            assert len(ps.I) == 6 * (N - 1) + wrap, (bcl, N, len(ps.I))
            assert max(ps.sends) <= 6 and max(ps.needs) <= 6 + wrap


def test_staged_interface_solve_equals_the_patch_solve():
    """The staged form the device runs (oracle/partitioned_np.InterfaceSolve: tile rows in, 10 rows per tile through the reduced
    system, tile rows out; halo rows and PERIODIC wrap rows travel as foreign rows) against the one-patch solve on the summed
    B rows (src/semiimplicit.jl:320-329, 285), every radial boundary-condition class, even and uneven tiles."""
    from oracle import oracle_np as O
    from oracle.partitioned_np import InterfaceSolve
    rng = np.random.default_rng(11)
    for bcl, bcr in (("R0", "R0"), ("R1T0", "R1T1"), ("R1T2", "R2T10"), ("R2T20", "R3"), ("R3", "R1T0"), ("PERIODIC", "PERIODIC")):
        for cells in ((20, 20), (9, 7, 8, 16), (12, 6, 6, 6, 6, 6, 6, 12)):
            nc = sum(cells)
            sp = O.Spline1D(0.0, 3.0e5, nc, bcl=bcl, bcr=bcr)
            cell0 = [sum(cells[:t]) for t in range(len(cells))]
            Bt = [rng.standard_normal((n + 3, 5)) for n in cells]            # every tile's own rows incl. its 3 halo rows
            shared = np.zeros((nc + 3, 5))
            for c0, b in zip(cell0, Bt):
                shared[c0:c0 + len(b)] += b
            ref = sp.SA(shared)
            ps = InterfaceSolve(sp, cell0, list(cells))
            loc = [ps.local(t, Bt[t]) for t in range(len(cells))]
            out = ps.reduce(np.stack([s for _, s in loc]))
            for t, (y, _) in enumerate(loc):
                got = ps.apply(t, y, out[t])
                want = ref[cell0[t]:cell0[t] + cells[t] + 3]
                assert np.abs(got - want).max() <= 1e-11 * np.abs(ref).max(), (bcl, bcr, cells, t)


def test_exchange_auto_picks_the_interface_only_solve_where_tiles_allow_it():
    """ModelRun(exchange="auto") / integrate_model: "iface" when every tile has >= 9 cells (6 free spline coefficients under any
    boundary condition), else the transposed solve - decided from calcTileSizes alone, i.e. identically on every rank."""
    import scythe_jl_amd as S
    gp = lambda nc: S.GridParameters(geometry="R", xmin=0.0, xmax=1.0, num_cells=nc, vars={"u": 1})
    pick = lambda nc, n: "iface" if n > 1 and min(S.PatchLayout(gp(nc), n).ncells) >= 9 else "a2a"
    assert pick(100, 2) == "iface" and pick(171, 8) == "iface" and pick(24, 4) == "a2a" and pick(100, 1) == "a2a"
    # native rings balance gridpoints: the outer tiles are the short ones
    lay = S.PatchLayout(S.GridParameters(geometry="RL", xmin=0.0, xmax=1.0, num_cells=171, vars={"u": 1}), 8)
    assert sum(lay.ncells) == 171 and min(lay.ncells) >= 9
