"""CPU-only checks of the oracle itself: the loop-based C port against the dense numpy definition, and the
mathematical properties the external (Springsteel) part must satisfy since no reference fixture pins it."""
import numpy as np
import pytest

pytestmark = pytest.mark.filterwarnings("ignore:divide by zero")

from oracle import oracle_np as O
from tests import cases


@pytest.mark.parametrize("maker,kw,tiles,steps", [
    (cases.r_bcs, dict(bcl="R1T0", bcr="R2T10"), None, 5),
    (cases.rz_advection, {}, [(0, 4), (4, 6)], 4),
    (cases.rz_semiimplicit, {}, [(0, 3), (3, 5)], 5),
    (cases.rl_advection, {}, None, 4),
    (cases.rl_advection, dict(ring_L=16), [(0, 3), (3, 5)], 4),
    (cases.rl_slab, dict(twoway=True), [(0, 3), (3, 5)], 4),
    (cases.rlz_hrbl, {}, [(0, 2), (2, 3)], 3),
    (cases.rlz_hrbl, dict(ring_L=16), None, 3),
    (cases.rlz_advection, {}, None, 3),
])
def test_c_port_equals_numpy_definition(maker, kw, tiles, steps):
    case = maker(**kw)
    a = cases.OracleModel(case, tiles=tiles)
    b = cases.OracleModel(case, tiles=tiles, numpy_twin=True)
    assert cases.rel_err(a.A, b.A) < 1e-12
    for _ in range(steps):
        a.step()
        b.step()
    assert cases.rel_err_per_var(a.physical(), b.physical()) < 5e-10


def test_tiles_equal_single_patch():
    case = cases.rl_slab(num_cells=9)
    a = cases.OracleModel(case)
    b = cases.OracleModel(case, tiles=[(0, 3), (3, 3), (6, 3)])
    for _ in range(3):
        a.step()
        b.step()
    assert cases.rel_err_per_var(b.physical(), a.physical()) < 1e-11


@pytest.mark.parametrize("bcl,bcr", [("R1T0", "R0"), ("R1T1", "R1T0"), ("R1T2", "R1T1"), ("R2T10", "R2T20"), ("R3", "R3")])
def test_radial_boundary_conditions_are_satisfied(bcl, bcr):
    """The spline reconstructed from A coefficients obeys the stated condition at both ends to rounding."""
    s = O.Spline1D(0.0, 7.0, 14, 2.0, bcl, bcr)
    rng = np.random.default_rng(3)
    a = s.SA(s.SB(rng.standard_normal(len(s.mish))))
    ends = np.array([s.xmin, s.xmax])
    val, d1, d2 = (s.basis(ends, d) @ a for d in range(3))
    scale = np.abs(a).max()
    for side, bc in ((0, bcl), (1, bcr)):
        if bc in ("R1T0", "R2T10", "R2T20", "R3"):
            assert abs(val[side]) < 1e-12 * scale
        if bc in ("R1T1", "R2T10", "R3"):
            assert abs(d1[side]) < 1e-12 * scale / s.DX
        if bc in ("R1T2", "R2T20", "R3"):
            assert abs(d2[side]) < 1e-12 * scale / s.DX ** 2


def test_spline_roundtrip_and_rescaling_invariance():
    """SA(SB(u)) reproduces an in-space function; physical answers are invariant under x -> lambda x."""
    s1 = O.Spline1D(0.0, 10.0, 20, 2.0)
    s2 = O.Spline1D(0.0, 1000.0, 20, 2.0)
    f = lambda t: np.exp(-((t - 0.5) / 0.2) ** 2)
    a1 = s1.SA(s1.SB(f(s1.mish / 10.0)))
    a2 = s2.SA(s2.SB(f(s2.mish / 1000.0)))
    assert np.max(np.abs(a1 - a2)) < 1e-12
    d1 = s1.basis(s1.mish, 1) @ a1
    d2 = s2.basis(s2.mish, 1) @ a2
    assert np.max(np.abs(d1 - 100.0 * d2)) < 1e-10
    # the filter penalises the third derivative only, so a quadratic is reproduced exactly
    p = lambda x: 1.0 + 0.3 * x - 0.02 * x ** 2
    a = s1.SA(s1.SB(p(s1.mish)))
    assert np.max(np.abs(s1.basis(s1.mish, 0) @ a - p(s1.mish))) < 1e-11


def test_fourier_ring_operators():
    rg = O.Ring(20, 4, 0.37)
    lam = rg.lam
    u = 0.3 + np.cos(2 * lam) - 0.5 * np.sin(3 * lam) + 0.25 * np.cos(4 * lam + 0.2)
    c = rg.FB @ u
    assert np.max(np.abs(rg.FI[0] @ c - u)) < 1e-13
    du = -2 * np.sin(2 * lam) - 1.5 * np.cos(3 * lam) - np.sin(4 * lam + 0.2)
    ddu = -4 * np.cos(2 * lam) + 4.5 * np.sin(3 * lam) - 4 * np.cos(4 * lam + 0.2)
    assert np.max(np.abs(rg.FI[1] @ c - du)) < 1e-12
    assert np.max(np.abs(rg.FI[2] @ c - ddu)) < 1e-12
    # all rings share the lambda = 0 phase reference: the same field gives the same coefficients on a shifted ring
    rg2 = O.Ring(28, 4, 1.1)
    u2 = 0.3 + np.cos(2 * rg2.lam) - 0.5 * np.sin(3 * rg2.lam) + 0.25 * np.cos(4 * rg2.lam + 0.2)
    assert np.max(np.abs(rg2.FB @ u2 - c)) < 1e-13


def test_chebyshev_operators():
    ch = O.Cheb(0.0, 5.0, 24, 24)
    z = ch.z
    assert z[0] == 0.0 and abs(z[-1] - 5.0) < 1e-14            # index 0 is the bottom
    f, fz, fzz = np.exp(0.3 * z) * np.sin(z), None, None
    fz = np.exp(0.3 * z) * (0.3 * np.sin(z) + np.cos(z))
    fzz = np.exp(0.3 * z) * ((0.09 - 1) * np.sin(z) + 0.6 * np.cos(z))
    b = ch.CBm @ f
    assert np.max(np.abs(ch.M[0] @ b - f)) < 1e-12             # CI(CA(CB u)) = u when nothing is truncated
    assert np.max(np.abs(ch.M[1] @ b - fz)) < 1e-9
    assert np.max(np.abs(ch.M[2] @ b - fzz)) < 1e-7
    integ = ch.Mint @ b                                         # integral from the bottom
    exact = (np.exp(0.3 * z) * (0.3 * np.sin(z) - np.cos(z)) + 1.0) / (0.09 + 1.0)
    assert abs(integ[0]) < 1e-13 and np.max(np.abs(integ - exact)) < 1e-11
    # collocation matrices agree with the coefficient recurrences (dct_1st/2nd_derivative vs CIx/CIxx)
    a = np.linalg.solve(ch.dct_matrix(), f)
    assert np.max(np.abs(ch.dct_1st_derivative() @ a - ch.M[1] @ b)) < 1e-9
    assert np.max(np.abs(ch.dct_2nd_derivative() @ a - ch.M[2] @ b)) < 1e-7
    assert O.default_bzdim(64) == 43 and O.default_bzdim(128) == 86


@pytest.mark.parametrize("bcb,bct", [("R1T0", "R0"), ("R1T1", "R1T0"), ("R0", "R1T1")])
def test_vertical_boundary_conditions_are_satisfied(bcb, bct):
    ch = O.Cheb(0.0, 3.0, 16, 11, bcb, bct)
    b = np.random.default_rng(5).standard_normal(11)
    val, dz = ch.M[0] @ b, ch.M[1] @ b
    for idx, bc in ((0, bcb), (-1, bct)):
        if bc == "R1T0":
            assert abs(val[idx]) < 1e-12
        if bc == "R1T1":
            assert abs(dz[idx]) < 1e-11


def test_helmholtz_solution_satisfies_dirichlet_rows():
    """rows 1-2 of H impose w = 0 at bottom and top (src/semiimplicit.jl:777-779)."""
    ch = O.Cheb(0.0, 1.0e4, 16, 16)
    H = O.helmholtz_matrix(ch, 1.2e5, 2.5)
    g = np.zeros(16)
    g[2:] = np.random.default_rng(7).standard_normal(14)
    w = ch.T @ np.linalg.solve(H, g)
    assert abs(w[0]) < 1e-9 * np.abs(w).max() and abs(w[-1]) < 1e-9 * np.abs(w).max()


# ----------------------------------------------------------------------------- pins to the named third-party algorithms
# Springsteel's ring / column transforms are FFTW plans (R2HC / HC2R for the Fourier rings, REDFT00 for the Chebyshev
# columns).  numpy.fft.rfft / irfft and scipy.fft.dct(type=1) implement the same published definitions
# (unnormalised sums), so the oracle's dense operators are pinned to them here: normalisation placement, sign of the
# imaginary part, phase reference and bottom-first ordering included.
@pytest.mark.parametrize("L,kmax,off", [(8, 1, 0.0), (20, 4, 0.37), (256, 127, 0.0), (364, 90, 0.5 * (2 * np.pi / 364) * 89), (31, 15, 1.3)])
def test_ring_operators_are_fftw_r2hc_hc2r(L, kmax, off):
    rg = O.Ring(L, kmax, off)
    rng = np.random.default_rng(L)
    u = rng.standard_normal(L)
    k = np.arange(kmax + 1)
    # forward: FB = (R2HC / L) rotated to the common lambda = 0 reference:  b_k = e^{-i k off} rfft(u)[k] / L
    spec = np.exp(-1j * k * off) * np.fft.rfft(u)[: kmax + 1] / L
    c = rg.FB @ u
    assert abs(c[0] - spec[0].real) < 1e-15 * L
    assert np.max(np.abs(c[1::2] - spec[1:].real)) < 1e-14 and np.max(np.abs(c[2::2] - spec[1:].imag)) < 1e-14
    # inverse: FI = HC2R of the rotated-back half-complex spectrum, truncated at kmax:  u = L irfft(e^{+i k off} a)
    a = rng.standard_normal(1 + 2 * kmax)
    full = np.zeros(L // 2 + 1, dtype=complex)
    full[0] = a[0]
    full[1: kmax + 1] = (a[1::2] + 1j * a[2::2]) * np.exp(1j * k[1:] * off)
    if L % 2 == 0 and kmax == L // 2:
        pytest.skip("Nyquist bin is never kept (kmax < L / 2)")
    for ld, fac in ((0, np.ones(L // 2 + 1)), (1, 1j * np.arange(L // 2 + 1)), (2, -np.arange(L // 2 + 1) ** 2.0)):
        ref = L * np.fft.irfft(fac * full, n=L)
        scale = max(np.abs(ref).max(), 1.0)
        assert np.max(np.abs(rg.FI[ld] @ a - ref)) < 1e-13 * scale


@pytest.mark.parametrize("N,bdim", [(9, 9), (16, 11), (64, 43), (128, 128)])
def test_chebyshev_operators_are_fftw_redft00(N, bdim):
    from scipy.fft import dct
    ch = O.Cheb(0.0, 7.0, N, bdim)
    rng = np.random.default_rng(N)
    u = rng.standard_normal(N)                                     # index 0 = bottom (z = zmin)
    # CB = REDFT00 / (2 (N - 1)), first b_zDim coefficients
    assert np.max(np.abs(ch.CBm @ u - dct(u, type=1)[:bdim] / (2 * (N - 1)))) < 1e-14
    # CI = REDFT00 of the coefficient column (dct_matrix)
    a = rng.standard_normal(N)
    assert np.max(np.abs(ch.T @ a - dct(a, type=1))) < 1e-12
    # round trip with nothing truncated: REDFT00(REDFT00(u)) = 2 (N - 1) u
    if bdim == N:
        assert np.max(np.abs(ch.T @ (ch.CBm @ u) - u)) < 1e-13
    # bottom-first: a field that grows with height has a NEGATIVE first Chebyshev coefficient (x = +1 is the bottom)
    assert (ch.CBm @ ch.z)[1] < 0.0


# ----------------------------------------------------------------------------- semi-implicit solve: the reference's arithmetic
@pytest.mark.parametrize("zDim", [16, 64])
def test_helmholtz_lu_path_is_the_reference_arithmetic_and_agrees_with_the_extended_one(zDim):
    """HelmholtzLU follows src/semiimplicit.jl:768-781, 586-595 literally (Float64 h_a, getrf, getrs, CItransform!,
    CIxtransform); the extended-precision inverse is the exact-arithmetic arbiter.  After 5 steps of the RZ model the
    fields of the two agree to 1e-12: the LU's rounding noise does not reach the model state."""
    from scipy.linalg import lu_factor, lu_solve
    ch = O.Cheb(0.0, 1.0e3, zDim, zDim, "R1T0", "R1T0")
    tau, pxi = 2.5, 1.2e5
    H = O.helmholtz_matrix_f64(ch, pxi, tau)
    c = tau * tau * pxi
    assert np.array_equal(H[0], c * ch.dct_matrix()[0]) and np.array_equal(H[1], c * ch.dct_matrix()[-1])
    assert np.array_equal(H[2:], (c * ch.dct_2nd_derivative() - ch.dct_matrix())[1:-1])
    g = np.zeros((3, zDim))
    g[:, 2:] = np.random.default_rng(1).standard_normal((3, zDim - 2))
    w, wz = O.HelmholtzLU(ch, pxi, tau).solve(g)
    a = lu_solve(lu_factor(H), g.T)
    assert np.array_equal(w, (ch.T @ a).T)
    W, X = O.semi_matrices(ch, pxi, tau)
    assert np.max(np.abs(w - g @ W.T)) < 1e-11 * np.abs(w).max()
    case = cases.config3_rz(num_cells=6, zDim=zDim)
    m_ext = cases.OracleModel(case, numpy_twin=True)
    m_lu = cases.OracleModel(case, helmholtz="lu")
    for _ in range(5):
        m_ext.step()
        m_lu.step()
    pe, pl = m_ext.physical(), m_lu.physical()
    for v in range(pe.shape[1]):
        assert np.abs(pe[:, v, 0] - pl[:, v, 0]).max() < 1e-12 * np.abs(pe[:, v, 0]).max()


# ----------------------------------------------------------------------------- two operator constructions
def test_c_oracle_builds_its_own_operators_and_they_match_the_numpy_definition():
    """scythe_oracle_ops.c constructs basis tables, quadrature weights, the boundary-condition projection + Cholesky factor,
    the Chebyshev column operators and the Helmholtz operator from the definitions, independently of oracle_np.py.  Operator
    by operator the two constructions agree to rounding - so test_c_port_equals_numpy_definition (above; OPS = "c" is the
    default) compares two constructions as well as two ways of applying them."""
    from oracle import oracle_c as OC
    assert OC.OPS == "c"
    g = cases.oracle_grid(cases.rlz_hrbl(num_cells=9, zDim=20))
    rel = lambda a, b: np.abs(a - b).max() / np.abs(b).max()
    for v in ("h", "ub"):
        c, n = OC.c_cheb(g, v), g.cheb(v)
        assert np.array_equal(c["z"], n.z)
        for a, b in ((c["M"][0], n.M[0]), (c["M"][1], n.M[1]), (c["M"][2], n.M[2]), (c["CB"], n.CBm), (c["Vint"], n.Vint),
                     (c["Vdz"], n.Vdz), (c["Vrec"], n.Vrec), (c["T"], n.dct_matrix()), (c["D1"], n.dct_1st_derivative()),
                     (c["D2"], n.dct_2nd_derivative())):
            assert rel(a, b) < 1e-14
    for bcl, bcr in (("R0", "R0"), ("R1T0", "R1T1"), ("R1T1", "R0"), ("R2T10", "R3"), ("R1T2", "R2T20"), ("PERIODIC", "PERIODIC")):
        nfree, per, rl, rr, gl, gr, Lb, La = OC.c_spline_class(g, bcl, bcr)
        s = g.spline(bcl, bcr)
        Lc, n, nb = s.cho, s.G.shape[0], g.b_rDim
        assert nfree == n and per == (bcl == "PERIODIC")
        band = np.zeros((nb, 4))
        for i in range(n):
            for q in range(4):
                if i - q >= 0:
                    band[i, 3 - q] = Lc[i, i - q]
        assert rel(Lb, band) < 1e-13
        if per:
            assert rel(La[:, :n], Lc[n - 3:, :]) < 1e-13
        else:
            assert (rl, rr) == (O.BC_RANK[bcl], O.BC_RANK[bcr])
            for i in range(rl):
                assert (gl[i, 0], gl[i, 1]) == (s.G[0, i], s.G[1, i])
            for i in range(rr):
                assert (gr[i, 0], gr[i, 1]) == (s.G[n - 1, nb - 1 - i], s.G[n - 2, nb - 1 - i])
    # basis values and weights
    buf, w3 = np.zeros(4), np.zeros(3)
    spl = g.spline("R0", "R0")
    for c in (0, 4, 8):
        for mu in range(3):
            x = O.mish_points(g.xmin, g.DX, c, 1)[mu:mu + 1]
            for d in range(3):
                OC.lib().orc_ops_phi(g.xmin, g.DX, c, mu, d, OC._pd(buf), None)
                ref = spl.basis(x, d)[0, c:c + 4]
                assert np.abs(buf - ref).max() <= 1e-15 * max(np.abs(ref).max(), 1e-300)
    OC.lib().orc_ops_wq(g.DX, OC._pd(w3))
    assert np.abs(w3 - g.DX * O.QUAD_W).max() < 1e-15 * g.DX
    # Helmholtz operator (semi-implicit solve)
    gz = cases.oracle_grid(cases.rz_semiimplicit(zDim=24))
    W, X = np.zeros((24, 24)), np.zeros((24, 24))
    assert OC.lib().orc_ops_helmholtz(gz.zmin, gz.zmax, 24, 1.2e5, 2.5, OC._pd(W), OC._pd(X)) == 0
    Wn, Xn = O.semi_matrices(gz.cheb("w"), 1.2e5, 2.5)
    assert rel(W, Wn) < 1e-13 and rel(X, Xn) < 1e-13


@pytest.mark.parametrize("geometry,ring_L", [("RLZ", 32), ("RL", None), ("RZ", None)])
def test_oracle_reproduces_the_closed_form_derivatives_of_an_analytic_field(geometry, ring_L):
    """The oracle against mathematics (tests/test_gpu_analytic.py does the same with the HIP path): forward transform, spline
    solve and inverse transform of u = F(r) A(lambda) G(z); every derivative slot within the cubic spline's truncation error of
    the closed form, falling at the spline's rate (16 / 8 / 4 per halving of DX) - no convention error survives that."""
    from tests import test_gpu_analytic as T

    def errors(nc):
        nv = {"RLZ": {"h": 1, "u": 2, "v": 3}, "RL": {"h": 1, "u": 2, "v": 3}, "RZ": {"h": 1, "u": 2, "v": 3, "w": 4}}[geometry]
        grid = dict(geometry=geometry, xmin=0.0, xmax=16.0, num_cells=nc, vars=nv)
        if "Z" in geometry:
            grid.update(zmin=0.0, zmax=T.ZMAX, zDim=24)
        if "L" in geometry:
            grid.update(ring_L=ring_L)
        keep = {}

        def ic(p):
            r = p[:, 0]
            e = T.field(r, p[:, 1] if "L" in geometry else 0 * r, p[:, -1] if "Z" in geometry else 0 * r + 1.0)
            keep["exact"] = e
            v = np.zeros((len(r), len(nv)))
            v[:, 0] = e[:, 0]
            return v
        eq = {"RLZ": "LinearAdvectionRLZ", "RL": "LinearAdvectionRL", "RZ": "LinearAdvectionRZ"}[geometry]
        ph = cases.OracleModel(dict(name="analytic", grid=grid, eq=eq, ts=0.01, par=dict(K=0.0), ic=ic)).physical()
        slots = {"RLZ": [0, 1, 2, 3, 4, 5, 6], "RL": [0, 1, 2, 3, 4], "RZ": [0, 1, 2, 5, 6]}[geometry]
        ex = keep["exact"]
        return slots, [np.abs(ph[:, 0, d] - ex[:, s]).max() / np.abs(ex[:, s]).max() for d, s in enumerate(slots)]

    slots, e64 = errors(64)
    _, e128 = errors(128)
    for s, a, b in zip(slots, e64, e128):
        assert a < T.BOUND[s], (T.NAMES[s], a)
        assert a / b > {1: 5.0, 2: 3.0}.get(s, 10.0), (T.NAMES[s], a, b)


def test_oracle_semiimplicit_standing_acoustic_wave_is_second_order_to_the_closed_form():
    """tests/test_gpu_analytic.py's standing-wave check on the oracle (C port and the numpy twin with the reference's LU)."""
    from tests import test_gpu_analytic as T
    import functools
    for cls in (cases.OracleModel, functools.partial(cases.OracleModel, numpy_twin=True, helmholtz="lu")):
        a = T.standing_wave_errors(cls, 0.5, 40)
        b = T.standing_wave_errors(cls, 0.25, 80)
        assert a[0] < 6e-3 and a[1] < 4e-3
        assert 3.5 < a[0] / b[0] < 4.5 and 3.5 < a[1] / b[1] < 4.5


def test_oracle_advected_and_diffused_sine_matches_the_closed_form():
    from tests import test_gpu_analytic as T
    a, amp = T.decaying_wave_error(cases.OracleModel, 100, 0.05, 800)
    b, _ = T.decaying_wave_error(cases.OracleModel, 200, 0.025, 1600)
    assert 0.4 < amp < 0.6 and a < 3e-4 and b < 7e-5 and a / b > 3.0


def test_oracle_bessel_mode_decays_at_the_rate_of_the_polar_laplacian():
    from tests import test_gpu_analytic as T
    err, change = T.bessel_mode_error(cases.OracleModel, "RL", 64)
    assert change > 0.03 and err < 1e-4
