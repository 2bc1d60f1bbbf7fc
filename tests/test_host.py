"""CPU-only: the host-side mirror of the reference interface (no compute)."""
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
