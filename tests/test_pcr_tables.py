"""CPU-only: the parallel-cyclic-reduction tables of the B -> A spline solve (csrc/sx_setup.cpp::build_pcr_tables, applied on the
device by csrc/sx_pcr.hip::k_solve_pcr) against the banded Cholesky statement of the same solve (what k_solve applies) and
against the oracle's dense definition, for every radial boundary-condition class incl. PERIODIC - through the host helper
sx_spline_solve_check, which applies the tables level by level in double exactly as the kernel does."""
import ctypes as C

import numpy as np
import pytest

from oracle import oracle_np as O

BCS = ["R0", "R1T0", "R1T1", "R1T2", "R2T10", "R2T20", "R3"]


def _solve(nc, bcl, bcr, b, xmin=0.0, xmax=None, l_q=2.0):
    import scythe_jl_amd as S
    from scythe_jl_amd import _lib as L
    lib = S.load()
    xmax = float(nc) if xmax is None else xmax
    nb = nc + 3
    a1, a2 = np.zeros(nb), np.zeros(nb)
    lev = C.c_int32(0)
    L.check(lib.sx_spline_solve_check(nc, xmin, xmax, l_q, L.BC[bcl], L.BC[bcr], np.ascontiguousarray(b).ctypes.data_as(L.P_D),
                                      a1.ctypes.data_as(L.P_D), a2.ctypes.data_as(L.P_D), C.byref(lev)))
    return a1, a2, lev.value


@pytest.mark.parametrize("nc", [3, 4, 5, 7, 8, 9, 16, 21, 42, 100, 171, 341, 1000])
def test_pcr_tables_equal_the_cholesky_solve_for_every_boundary_condition_pair(nc):
    rng = np.random.default_rng(nc)
    worst = 0.0
    for bcl in BCS:
        for bcr in BCS:
            nfree = nc + 3 - {"R0": 0, "R3": 3}.get(bcl, 1 if bcl.startswith("R1") else 2) - {"R0": 0, "R3": 3}.get(bcr, 1 if bcr.startswith("R1") else 2)
            if nfree < 4:
                continue
            b = rng.standard_normal(nc + 3)
            a1, a2, lev = _solve(nc, bcl, bcr, b, xmax=3.0e5)
            assert lev <= 7, (nc, lev)          # the couplings decay doubly exponentially: never more than 7 levels
            e = np.abs(a1 - a2).max() / np.abs(a2).max()
            worst = max(worst, e)
            assert e < 2e-13, (nc, bcl, bcr, e)
    print("nc = %d: PCR vs Cholesky, worst over the 49 boundary-condition pairs %.1e" % (nc, worst))


@pytest.mark.parametrize("nc", [7, 8, 9, 10, 11, 33, 100, 101, 171, 500])
def test_pcr_tables_periodic(nc):
    rng = np.random.default_rng(100 + nc)
    b = rng.standard_normal(nc + 3)
    a1, a2, lev = _solve(nc, "PERIODIC", "PERIODIC", b, xmin=-50.0, xmax=50.0)
    assert np.abs(a1 - a2).max() < 2e-13 * np.abs(a2).max()
    # the periodic images: a_{-1} = a_{n-1}, a_n = a_0, a_{n+1} = a_1
    assert a1[0] == a1[nc] and a1[nc + 1] == a1[1] and a1[nc + 2] == a1[2]


@pytest.mark.parametrize("bcl,bcr,nc", [("R1T0", "R1T1", 24), ("R0", "R0", 10), ("R2T20", "R3", 30), ("PERIODIC", "PERIODIC", 100), ("R1T1", "R0", 171)])
def test_pcr_tables_against_the_oracles_dense_definition(bcl, bcr, nc):
    """a = Gamma^T (Gamma (P + eps_q Q) Gamma^T)^-1 Gamma b with the numpy oracle's dense operator (oracle_np.Spline1D.SA)."""
    sp = O.Spline1D(0.0, 12.0, nc, bcl=bcl, bcr=bcr)
    rng = np.random.default_rng(7)
    b = rng.standard_normal(nc + 3)
    a1, a2, _ = _solve(nc, bcl, bcr, b, xmax=12.0)
    ref = sp.SA(b)
    assert np.abs(a1 - ref).max() < 1e-12 * np.abs(ref).max()
    assert np.abs(a2 - ref).max() < 1e-12 * np.abs(ref).max()
