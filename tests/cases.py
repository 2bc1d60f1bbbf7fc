"""Shared test cases: one description drives the CPU oracle (oracle/) and the HIP path (scythe.jl_amd)."""
import numpy as np

from oracle import oracle_np as O
from oracle import oracle_c as OC

VARS6 = {"h": 1, "u": 2, "v": 3, "ub": 4, "vb": 5, "wb": 6}
BCL6 = {"h": "R1T1", "u": "R1T0", "v": "R1T0", "ub": "R1T0", "vb": "R1T0", "wb": "R1T1"}   # models/cha_bell2024
BCR6 = {"h": "R0", "u": "R1T1", "v": "R0", "ub": "R1T1", "vb": "R0", "wb": "R0"}
SW_PAR = dict(g=9.81, K=5000.0, Cd=2.4e-3, Hfree=2000.0, Hb=1000.0, f=5.0e-5, S1=1.0e-4, Kh=5000.0, Um=2.0, Vm=-1.0)


def _vortex(r):
    Rmax, V0 = 5.0e4, 50.0 / 5.0e4
    return np.where(r < Rmax, V0 * r, Rmax * Rmax * V0 / np.maximum(r, 1.0))


def kat_r(num_cells=100):
    """models/LinearAdvection1D.jl with the notebook's initial condition."""
    return dict(name="kat_r", grid=dict(geometry="R", xmin=-50.0, xmax=50.0, num_cells=num_cells, vars={"u": 1},
                                        BCL={"u": "PERIODIC"}, BCR={"u": "PERIODIC"}),
                eq="LinearAdvection1D", ts=0.05, par=dict(c_0=1.0, K=0.0),
                ic=lambda p: np.exp(-(p[:, 0] / 20.0) ** 2)[:, None])


def r_bcs(bcl="R1T0", bcr="R1T1", num_cells=24):
    return dict(name="r_%s_%s" % (bcl, bcr), grid=dict(geometry="R", xmin=0.0, xmax=12.0, num_cells=num_cells, vars={"u": 1},
                                                      BCL={"u": bcl}, BCR={"u": bcr}),
                eq="LinearAdvection1D", ts=0.01, par=dict(c_0=0.5, K=0.02),
                ic=lambda p: (np.sin(np.pi * p[:, 0] / 12.0) ** 2 * np.exp(-((p[:, 0] - 5.0) / 2.0) ** 2))[:, None])


def rz_advection(num_cells=10, zDim=14):
    def ic(p):
        r, z = p[:, 0], p[:, 1]
        b = np.exp(-((r - 5.0e3) / 2.0e3) ** 2 - ((z - 4.0e3) / 2.5e3) ** 2)
        return np.stack([b, 2.0 + 0.0 * r, 0.0 * r, 0.5 * np.sin(np.pi * z / 1.0e4)], axis=1)
    return dict(name="rz_adv", grid=dict(geometry="RZ", xmin=0.0, xmax=1.0e4, num_cells=num_cells, zmin=0.0, zmax=1.0e4,
                                         zDim=zDim, vars={"h": 1, "u": 2, "v": 3, "w": 4}, BCL={"h": "R1T1"},
                                         BCB={"w": "R1T0"}, BCT={"w": "R1T0"}),
                eq="LinearAdvectionRZ", ts=1.0, par=dict(K=20.0), ic=ic)


def rz_semiimplicit(num_cells=8, zDim=12):
    def ic(p):
        r, z = p[:, 0], p[:, 1]
        b = np.exp(-((r - 5.0e3) / 2.0e3) ** 2 - ((z - 5.0e3) / 2.0e3) ** 2)
        s = np.sin(np.pi * z / 1.0e4)
        return np.stack([b, 1.0e-3 * b, 0.5 * b, s * b, 0.5 * s * b], axis=1)
    return dict(name="rz_semi", grid=dict(geometry="RZ", xmin=0.0, xmax=1.0e4, num_cells=num_cells, zmin=0.0, zmax=1.0e4,
                                          zDim=zDim, b_zDim=zDim, vars={"s": 1, "xi": 2, "mu": 3, "u": 4, "w": 5},
                                          BCB={"w": "R1T0"}, BCT={"w": "R1T0"}),
                eq="LinearAcousticRZ", ts=2.0, par=dict(K=10.0, Pxi_bar=1.2e5), ic=ic, semiimplicit=True)


def rz_euler(num_cells=8, zDim=12, semiimplicit=True):
    """Euler_test (src/testModels.jl:100-215) about a stably stratified, slightly moist reference state.  The reference
    profiles are analytic; their derivative columns come from the ORACLE's Chebyshev operators (the same filtered
    CB -> CA -> CI / CIx / CIxx the reference applies, src/reference_state.jl:140-160) and are handed to both sides."""
    zmax = 1.0e4
    ch = O.Cheb(0.0, zmax, zDim, bdim=zDim)
    z = ch.z
    prof = dict(sbar=20.0 + 0.012 * z, xibar=-z / 8.5e3, mubar=0.5 * ((4e-3 * np.exp(-z / 2.5e3) + 1e-7) - 1e-14 / (4e-3 * np.exp(-z / 2.5e3) + 1e-7)))
    ref = {}
    for k, v in prof.items():
        b = ch.CBm @ v
        ref[k] = np.stack([ch.M[0] @ b, ch.M[1] @ b, ch.M[2] @ b], axis=1)

    def ic(p):
        r, zz = p[:, 0], p[:, 1]
        b = np.exp(-((r - 1.0e4) / 4.0e3) ** 2 - ((zz - 4.0e3) / 2.0e3) ** 2)
        s_ = np.sin(np.pi * zz / zmax)
        return np.stack([0.5 * b, 1.0e-4 * b, 2.0e-4 * b, 0.5 * s_ * b, 0.2 * s_ * b], axis=1)
    par = dict(K=10.0, Pxi_bar=1.0e5, ref_state=ref)
    return dict(name="rz_euler", grid=dict(geometry="RZ", xmin=0.0, xmax=2.0e4, num_cells=num_cells, zmin=0.0, zmax=zmax,
                                           zDim=zDim, b_zDim=zDim, vars={"s": 1, "xi": 2, "mu": 3, "u": 4, "w": 5},
                                           BCB={"w": "R1T0"}, BCT={"w": "R1T0"}),
                eq="Euler_test", ts=1.0, par=par, ic=ic, semiimplicit=semiimplicit)


def rl_advection(num_cells=8, ring_L=None):
    def ic(p):
        r, l = p[:, 0], p[:, 1]
        e = np.exp(-(r / 4.0) ** 2)
        return np.stack([e * (1 + (r / 4) ** 2 * np.cos(2 * l) + (r / 4) * np.sin(l)), 0.3 + 0 * r, 0.2 * r], axis=1)
    return dict(name="rl_adv", grid=dict(geometry="RL", xmin=0.0, xmax=10.0, num_cells=num_cells, vars={"h": 1, "u": 2, "v": 3},
                                         ring_L=ring_L),
                eq="LinearAdvectionRL", ts=0.01, par=dict(K=0.003), ic=ic)


def rl_slab(num_cells=8, twoway=False, ring_L=None):
    """models/cha_bell2024/*.jl boundary conditions and parameters on a small patch."""
    def ic(p):
        r, l = p[:, 0], p[:, 1]
        vb = _vortex(r)
        h = 100.0 * np.exp(-(r / 1.0e5) ** 2) * (1 + 0.1 * np.cos(2 * l))
        u = 0.5 * np.sin(l) * r / 3.0e5
        return np.stack([h, u, vb * (1 + 0.05 * np.cos(l)), 0.8 * u, 0.7 * vb, 0.0 * r], axis=1)
    return dict(name="rl_slab", grid=dict(geometry="RL", xmin=0.0, xmax=3.0e5, num_cells=num_cells, vars=VARS6, BCL=BCL6,
                                          BCR=BCR6, ring_L=ring_L),
                eq="Twoway_ShallowWater_Slab" if twoway else "Oneway_ShallowWater_Slab", ts=3.0, par=dict(SW_PAR), ic=ic)


def rlz_hrbl(num_cells=5, zDim=10, ring_L=None):
    def ic(p):
        r, l, z = p.T
        vb = _vortex(r)
        dec = 1.0 - np.exp(-(z + 50.0) / 300.0)
        h = 100.0 * np.exp(-(r / 1.0e5) ** 2) * (1 + 0.1 * np.cos(2 * l))
        u = 0.5 * np.sin(l) * r / 3.0e5
        return np.stack([h, u, vb * (1 + 0.05 * np.cos(l)), (u - 2.0 * r / 3.0e5) * dec, 0.7 * vb * dec, 0.0 * r], axis=1)
    return dict(name="rlz_hrbl", grid=dict(geometry="RLZ", xmin=0.0, xmax=3.0e5, num_cells=num_cells, vars=VARS6, BCL=BCL6,
                                           BCR=BCR6, zmin=0.0, zmax=2000.0, zDim=zDim, ring_L=ring_L),
                eq="Oneway_ShallowWater_HeightResolvedBL", ts=3.0, par=dict(SW_PAR), ic=ic)


def rlz_advection(num_cells=4, zDim=9, ring_L=None):
    def ic(p):
        r, l, z = p.T
        e = np.exp(-(r / 4.0) ** 2) * np.cos(0.4 * z)
        return np.stack([e * (1 + (r / 4) * np.sin(l)), 0.3 + 0 * r, 0.2 * r], axis=1)
    return dict(name="rlz_adv", grid=dict(geometry="RLZ", xmin=0.0, xmax=10.0, num_cells=num_cells, vars={"h": 1, "u": 2, "v": 3},
                                          zmin=0.0, zmax=3.0, zDim=zDim, ring_L=ring_L, BCB={"h": "R1T1"}),
                eq="LinearAdvectionRLZ", ts=0.01, par=dict(K=0.003), ic=ic)


def kat_in_geometry(geometry, ring_L=8, zDim=4):
    """The notebook's known answer (models/LinearAdvection1D.jl: u_t = -c_0 u_r, PERIODIC, 100 cells on [-50, 50], ts 0.05)
    posed on an RZ / RL / RLZ grid: LinearAdvectionRZ / RL / RLZ with K = 0, a unit radial wind and a field that does not
    depend on lambda or z integrate the same equation at every (lambda, z), so the notebook's printed values must come out at
    every ring point and level - the reference's only fixture, carried through the azimuthal and vertical transform paths."""
    nv = {"h": 1, "u": 2, "v": 3, "w": 4} if geometry == "RZ" else {"h": 1, "u": 2, "v": 3}
    grid = dict(geometry=geometry, xmin=-50.0, xmax=50.0, num_cells=100, vars=nv, BCL={k: "PERIODIC" for k in nv},
                BCR={k: "PERIODIC" for k in nv})
    if "Z" in geometry:
        grid.update(zmin=0.0, zmax=1.0, zDim=zDim)
    if "L" in geometry:
        grid.update(ring_L=ring_L)

    def ic(p):
        v = np.zeros((len(p), len(nv)))
        v[:, 0], v[:, 1] = np.exp(-(p[:, 0] / 20.0) ** 2), 1.0
        return v
    return dict(name="kat_" + geometry, grid=grid, eq="LinearAdvection" + geometry, ts=0.05, par=dict(K=0.0), ic=ic)


def kat_deviation(model, kat, steps=2000):
    """max relative deviation of the field from the notebook's printed values, over every ring point and level at those radii"""
    for _ in range(steps):
        model.step()
    ph = model.physical()
    r = model.pts[:, 0] if hasattr(model, "pts") else None
    if r is None:
        import scythe_jl_amd as S
        pts = np.concatenate([S.getGridpoints(g).reshape(g.N, -1) for g in model.run.tiles], axis=0)
        r = pts[:, 0]
    worst = 0.0
    for x, u in zip(kat["gridpoints"], kat["final_u"]):
        sel = np.abs(r - x) < 1e-9
        assert sel.any()
        worst = max(worst, np.abs(ph[sel, 0, 0] / u - 1.0).max())
    return worst


# ----------------------------------------------------------------------------- builders
def oracle_grid(case):
    g = dict(case["grid"])
    return O.Grid(g.pop("geometry"), g.pop("xmin"), g.pop("xmax"), g.pop("num_cells"), g.pop("vars"), **g)


def even_tiles(nc, n):
    sizes = [nc // n + (1 if t < nc % n else 0) for t in range(n)]
    out, c0 = [], 0
    for s in sizes:
        out.append((c0, s))
        c0 += s
    return out


class OracleModel:
    """C-oracle model (optionally split into tiles) started from the case's initial condition."""

    def __init__(self, case, tiles=None, numpy_twin=False, helmholtz="extended"):
        """helmholtz="lu" (numpy twin only): the reference's Float64 LU arithmetic for the semi-implicit column solve."""
        self.g = oracle_grid(case)
        if numpy_twin or helmholtz != "extended":
            self.m = O.Model(self.g, case["eq"], case["ts"], case["par"], tiles=tiles,
                             semiimplicit=case.get("semiimplicit", False), pxi_bar=case["par"].get("Pxi_bar", 0.0),
                             helmholtz=helmholtz)
        else:
            self.m = OC.ModelOracle(self.g, case["eq"], case["ts"], case["par"], tiles=tiles,
                                    semiimplicit=case.get("semiimplicit", False))
        pts = self.g.gridpoints()
        self.pts = pts.reshape(len(pts), -1)
        self.m.set_initial(case["ic"](self.pts))

    def step(self):
        self.m.step()

    def physical(self):
        return self.m.physical()

    @property
    def A(self):
        return self.m.A


def hip_params(case, storage="f64"):
    import scythe_jl_amd as S
    g = dict(case["grid"])
    ring_L = g.pop("ring_L", None)
    gp = S.GridParameters(ring_uniform_L=ring_L or 0, storage=storage, **g)
    par = dict(case["par"])
    rs = par.pop("ref_state", None)
    mp = S.ModelParameters(ts=case["ts"], equation_set=case["eq"], grid_params=gp, physical_params=par,
                           options={"semiimplicit": case.get("semiimplicit", False)})
    if rs is not None:
        mp.ref_state = S.ReferenceState(rs["sbar"], rs["xibar"], rs["mubar"], np.zeros_like(rs["sbar"]), par["Pxi_bar"])
    return gp, mp


class HipModel:
    """The product path: tiles are libscythe_hip handles, exchange on device buffers."""

    def __init__(self, case, num_tiles=1, device="cuda", exchange="a2a", storage="f64", impl="torch"):
        """impl="lib" (num_tiles > 1): the exchange runs inside libscythe_hip.so through its loopback transport (the RCCL
        path's buffers and offset tables, copies instead of sends); "torch": the Python-side stand-in (driver.Local*Exchange)."""
        import scythe_jl_amd as S
        self.gp, self.mp = hip_params(case, storage)
        self.run = S.ModelRun(self.mp, num_tiles=num_tiles, device=device, exchange=exchange, impl=impl)
        vals = []
        for g in self.run.tiles:
            pts = S.getGridpoints(g)
            vals.append(case["ic"](pts.reshape(len(pts), -1)))
        self.run.set_initial_conditions(vals)

    def step(self):
        self.run.step()

    def physical(self):
        return self.run.physical()

    @property
    def A(self):
        if self.run.exchange_kind in ("a2a", "iface"):
            return None          # no tile holds the whole patch in the transposed / interface-only solve
        return self.run.tiles[0].patchSpectral


def rel_err(a, b):
    """max over derivative slots of max|a-b| / max|b| (per-slot scale)."""
    a, b = np.asarray(a), np.asarray(b)
    if a.ndim == 3:
        return max(np.abs(a[:, :, d] - b[:, :, d]).max() / max(np.abs(b[:, :, d]).max(), 1e-300) for d in range(a.shape[2]))
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-300)


def rel_err_per_var(a, b):
    """max over (var, slot) of max|a-b| / scale[var, slot].

    scale = max|b[:, var, slot]|, floored at gain[slot] * max|b[:, var, 0]| where gain[slot] is the largest
    derivative-to-value ratio any variable shows in that slot: a derivative slot of a field that does not vary in
    that direction holds only rounding noise, and that noise is proportional to the field's own magnitude times the
    norm of the derivative operator."""
    worst = 0.0
    vmax = [np.abs(b[:, v, 0]).max() for v in range(a.shape[1])]
    for d in range(a.shape[2]):
        gain = max((np.abs(b[:, v, d]).max() / vmax[v]) for v in range(a.shape[1]) if vmax[v] > 0)
        for v in range(a.shape[1]):
            sc = max(np.abs(b[:, v, d]).max(), gain * vmax[v])
            diff = np.abs(a[:, v, d] - b[:, v, d]).max()
            if sc > 0:
                worst = max(worst, diff / sc)
            elif diff > 0:
                return np.inf
    return worst


# ----------------------------------------------------------------------------- BASELINE.json configs at full size
from bench_configs import config2_literal     # noqa: E402,F401  (one definition for bench.py and the tests)


def config5_case():
    """SURVEY.md 8(d) config 5 at its full size: 341 cells -> 1023 rings x 512 azimuthal points x 128 levels, HRBL set
    (bench.py --workload rlz_1023x512x128)."""
    case = rlz_hrbl(num_cells=341, zDim=128, ring_L=512)
    case["ts"] = 0.02                    # bench.TS_OF: 0.3 m end spacing of the 128-level Chebyshev column
    return case


from bench_configs import config3_rz          # noqa: E402,F401


# ----------------------------------------------------------------------------- measured accuracy of derivative slots
def ring_points(g, rings):
    """Index array of the points (z fastest) of the listed patch rings inside a one-tile physical array."""
    return np.concatenate([np.arange(g.ringstart[r] * g.zDim, (g.ringstart[r] + g.L[r]) * g.zDim) for r in rings])


def slot_errors_vs_extended(g, phys, A, rings):
    """For each derivative slot: max over variables of  max|phys - truth| / scale  on the sampled rings, where `truth` is
    the inverse transform of the SAME Float64 coefficients A evaluated in extended precision (oracle_np.inverse_xp) and
    scale is rel_err_per_var's (slot scale floored by the operator gain times the variable's magnitude).
    This measures the rounding error of one implementation's tileTransform!; it does not depend on any other one."""
    pts = {r: ring_points(g, [r]) for r in rings}
    return slot_errors_vs_extended_rings(g, {r: phys[pts[r]] for r in rings}, A, rings)


def slot_errors_vs_extended_rings(g, ring_phys, A, rings):
    """slot_errors_vs_extended with the evaluated fields given ring by ring ({ring: [L * zDim, V, D]}) - for grids whose
    whole `physical` array is never formed on the host (config 5)."""
    truth = O.inverse_xp(g, A, rings)
    t = np.concatenate([truth[r] for r in rings], axis=0)
    a = np.concatenate([ring_phys[r] for r in rings], axis=0)
    t64 = np.asarray(t, dtype=np.float64)
    vmax = [max(np.abs(t64[:, v, 0]).max(), 1e-300) for v in range(a.shape[1])]
    out = np.zeros(a.shape[2])
    for d in range(a.shape[2]):
        gain = max(np.abs(t64[:, v, d]).max() / vmax[v] for v in range(a.shape[1]))
        for v in range(a.shape[1]):
            sc = max(np.abs(t64[:, v, d]).max(), gain * vmax[v], 1e-300)
            out[d] = max(out[d], float(np.abs(a[:, v, d].astype(O.XP) - t[:, v, d]).max()) / sc)
    return out


def report_slots(title, g, rows):
    print("\n%s - max error per derivative slot, relative to the slot's scale" % title)
    print("  %-44s" % "" + "".join("%10s" % s for s in g.slots))
    for name, e in rows:
        print("  %-44s" % name + "".join("%10.1e" % x for x in e))


def per_slot_errors(a, b):
    """rel_err_per_var slot by slot: out[d] = max over variables of max|a - b| / scale[var, d], with rel_err_per_var's scale
    (the slot's own magnitude, floored by the operator gain times the variable's magnitude: a derivative slot of a field
    that does not vary in that direction holds rounding noise only)."""
    vmax = [np.abs(b[:, v, 0]).max() for v in range(a.shape[1])]
    out = np.zeros(a.shape[2])
    for d in range(a.shape[2]):
        gain = max((np.abs(b[:, v, d]).max() / vmax[v]) for v in range(a.shape[1]) if vmax[v] > 0)
        for v in range(a.shape[1]):
            sc = max(np.abs(b[:, v, d]).max(), gain * vmax[v])
            diff = np.abs(a[:, v, d] - b[:, v, d]).max()
            if sc > 0:
                out[d] = max(out[d], diff / sc)
            elif diff > 0:
                out[d] = np.inf
    return out


def check_full(hip, orc, rings, title, orc_alt=None):
    """Parity at full size, in separately measured parts:
    (1) STATE: the A coefficients and the value slot of HIP and oracle agree to 1e-10 after the steps.
    (2) TRANSFORM ACCURACY (tests/py::slot_errors_vs_extended): each side's derivative slots against the
        extended-precision evaluation of ITS OWN coefficients on a sample of rings (innermost, middle, outermost = largest
        kmax).  Demanded: the HIP path is no less accurate than the fp64 oracle, err(HIP) <= 2 err(oracle) (+ 2e-15 for
        slots the oracle happens to hit exactly).
    (3) WHOLE GRID, slot by slot, HIP vs oracle.  The derivative operators amplify the last-bit differences of two fp64
        states by k, k^2 (azimuth) or N^2, N^4 (Chebyshev), so two CORRECT runs differ in those slots by far more than
        1e-10 of the slot's scale.  The noise floor is measured, not assumed: `orc_alt` is the same oracle with the patch
        split into two tiles (another summation order of the same arithmetic); HIP must be within 10 x that floor."""
    a, b = hip.physical(), orc.physical()
    assert np.isfinite(a).all()
    eA = rel_err(hip.A, orc.A)
    vals = max(np.abs(a[:, v, 0] - b[:, v, 0]).max() / np.abs(b[:, v, 0]).max()
               for v in range(b.shape[1]) if np.abs(b[:, v, 0]).max() > 0)
    g = orc.g
    e_hip = slot_errors_vs_extended(g, a, np.asarray(hip.A), rings)
    e_orc = slot_errors_vs_extended(g, b, np.asarray(orc.A), rings)
    d = per_slot_errors(a, b)
    rows = [("HIP vs extended precision (own A)", e_hip), ("fp64 oracle vs extended precision (own A)", e_orc),
            ("HIP vs fp64 oracle (whole grid)", d)]
    floor = None
    if orc_alt is not None:
        floor = per_slot_errors(orc_alt.physical(), b)
        rows.append(("oracle, 2 tiles vs 1 tile (fp64 noise floor)", floor))
    report_slots(title + " (A coefficients %.1e, values %.1e)" % (eA, vals), g, rows)
    assert eA < 1e-10 and vals < 1e-10, (eA, vals)
    assert (e_hip <= 2.0 * e_orc + 2e-15).all(), (e_hip, e_orc)
    if floor is not None:
        assert (d <= 10.0 * floor + 1e-12).all(), (d, floor)
    return e_hip, e_orc, d


