"""The oracle against the reference's only known-answer data (notebooks/LinearAdvection_example.ipynb)."""
import json
import os

import numpy as np
import pytest

from oracle import oracle_np as O
from tests import cases

KAT = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "linear_advection_kat.json")))
IDX = KAT["index_0based"]


def test_gridpoints_match_notebook():
    g = cases.oracle_grid(cases.kat_r())
    x = g.gridpoints()
    assert len(x) == 300
    assert np.max(np.abs(x[IDX] - np.array(KAT["gridpoints"]))) < 1e-13


@pytest.mark.parametrize("tiles", [None, [(0, 50), (50, 50)], [(0, 33), (33, 30), (63, 37)]])
def test_c_oracle_reproduces_notebook_final_state(tiles):
    """2000 steps of periodic advection; the notebook itself ran with 2 workers (= 2 tiles)."""
    m = cases.OracleModel(cases.kat_r(), tiles=tiles)
    init = m.physical()[:, 0, 0]
    for _ in range(KAT["model"]["steps"]):
        m.step()
    fin = m.physical()[:, 0, 0]
    rel = np.max(np.abs(fin[IDX] / np.array(KAT["final_u"]) - 1.0))
    assert rel < 1e-11, rel
    l2 = np.sqrt(np.sum((init - fin) ** 2))
    assert abs(l2 / KAT["l2_norm"] - 1.0) < 1e-9


def test_numpy_definition_matches_c_oracle_on_kat_prefix():
    a = cases.OracleModel(cases.kat_r())
    b = cases.OracleModel(cases.kat_r(), numpy_twin=True)
    for _ in range(60):
        a.step()
        b.step()
    assert cases.rel_err(a.physical(), b.physical()) < 1e-11


def test_textbook_quadrature_weights_do_not_reproduce_the_notebook(monkeypatch):
    """Guards the non-textbook 8:5:8 weight ratio: 5:8:5 is off by > 1e-4 after 200 steps."""
    ref = cases.OracleModel(cases.kat_r(), numpy_twin=True)
    monkeypatch.setattr(O, "QUAD_W", np.array([5.0, 8.0, 5.0]) / 18.0)
    alt = cases.OracleModel(cases.kat_r(), numpy_twin=True)
    for _ in range(200):
        ref.step()
        alt.step()
    assert cases.rel_err(alt.physical()[:, :, :1], ref.physical()[:, :, :1]) > 1e-5


@pytest.mark.parametrize("geometry", ["RZ", "RL", "RLZ"])
def test_notebook_known_answer_on_rz_rl_rlz_grids(geometry):
    """The reference's fixture through the Fourier and Chebyshev paths: an axisymmetric, z-independent field advected by a unit
    radial wind on an RZ / RL / RLZ grid obeys the notebook's equation at every (lambda, z) (tests/cases.py::kat_in_geometry)."""
    m = cases.OracleModel(cases.kat_in_geometry(geometry))
    assert cases.kat_deviation(m, KAT) < 1e-11
