"""-m gpu: the reference's notebook known-answer (notebooks/LinearAdvection_example.ipynb) reproduced END TO END by the
HIP path: integrate_model() reading the initial-condition CSV, 2000 steps on the GPU (1 and 2 tiles), writing
physical_out_<t>.csv exactly as the notebook reads them back, compared with the printed values."""
import json
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

KAT = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "linear_advection_kat.json")))
IDX = KAT["index_0based"]


def _model(tmp_path, S):
    gp = S.GridParameters(geometry="R", xmin=-50.0, xmax=50.0, num_cells=100,
                          BCL={"u": S.CubicBSpline.PERIODIC}, BCR={"u": S.CubicBSpline.PERIODIC}, vars={"u": 1})
    return S.ModelParameters(ts=0.05, integration_time=100.0, output_interval=50.0, equation_set="LinearAdvection1D",
                             initial_conditions=str(tmp_path / "1d_linear_advection_test_ics.csv"),
                             output_dir=str(tmp_path / "linear_advection_test"), grid_params=gp,
                             physical_params={"c_0": 1.0, "K": 0.0})


@pytest.mark.parametrize("num_tiles", [1, 2])
def test_notebook_known_answer_through_integrate_model(tmp_path, num_tiles):
    import scythe_jl_amd as S
    model = _model(tmp_path, S)
    # notebook cells 2-5: grid, Gaussian initial condition, CSV
    grid = S.createGrid(model.grid_params)
    x = S.getGridpoints(grid)
    assert np.max(np.abs(x[IDX] - np.array(KAT["gridpoints"]))) < 1e-13
    grid.close()
    u0 = np.exp(-(x / 20.0) ** 2)
    np.savetxt(model.initial_conditions, np.stack([x, u0], axis=1), delimiter=",", header="r,u", comments="", fmt="%.17g")
    # notebook cell 6
    assert S.integrate_model(model, num_tiles=num_tiles) is True
    # notebook cells 7 and 9
    read = lambda t: np.loadtxt(os.path.join(model.output_dir, "physical_out_%s.csv" % t), delimiter=",", skiprows=1)
    initial, mid, final = read(0.0), read(50.0), read(100.0)
    assert initial.shape == (300, 4) and mid.shape == (300, 4)          # r, u, u_r, u_rr
    with open(os.path.join(model.output_dir, "physical_out_100.0.csv")) as f:
        assert f.readline().strip() == "r,u,u_r,u_rr"
    # b_rDim = num_cells + 3 spline coefficients; with two tiles (transposed solve) the file is assembled from the tiles' owned rows
    spec = np.loadtxt(os.path.join(model.output_dir, "spectral_out_100.0.csv"), delimiter=",", skiprows=1)
    assert spec.shape == (103, 2)
    g1 = S.createGrid(model.grid_params)
    g1.set_patch_spectral_a(spec[:, 1:])
    g1.tileTransform_()
    assert np.max(np.abs(g1.physical[:, 0, 0] - final[:, 1])) < 1e-12    # the written coefficients ARE the written field
    g1.close()
    rel = np.max(np.abs(final[IDX, 1] / np.array(KAT["final_u"]) - 1.0))
    assert rel < 1e-11, rel
    l2 = np.sqrt(np.sum((initial[:, 1] - final[:, 1]) ** 2))
    assert abs(l2 / KAT["l2_norm"] - 1.0) < 1e-9


@pytest.mark.parametrize("geometry,ring_L,tiles", [("RZ", None, 1), ("RL", 8, 1), ("RL", None, 1), ("RLZ", 8, 1), ("RLZ", None, 1),
                                                   ("RLZ", 8, 2), ("RZ", None, 3)])
def test_notebook_known_answer_on_rz_rl_rlz_grids(geometry, ring_L, tiles):
    """The reference's only fixture carried through the azimuthal and vertical transform paths of the HIP library: the notebook's
    equation posed on an RZ / RL / RLZ grid (tests/cases.py::kat_in_geometry; uniform rings = FFT kernels, native ragged rings up
    to 1,204 points = DFT kernels, 1-3 tiles), 2000 steps, the printed values at every ring point and level."""
    from tests import cases
    m = cases.HipModel(cases.kat_in_geometry(geometry, ring_L=ring_L), num_tiles=tiles, exchange="a2a")
    dev = cases.kat_deviation(m, KAT)
    m.run.close()
    print("\n%s ring_L=%s tiles=%d: max relative deviation from the notebook's values %.2e" % (geometry, ring_L, tiles, dev))
    assert dev < 1e-11
