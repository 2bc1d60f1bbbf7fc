"""BASELINE.json configs 2 and 3 as concrete synthetic inputs: ONE definition for bench.py's `--workload`
legs and for the parity tests (tests/cases.py re-exports them).  Plain dictionaries: grid keywords of GridParameters, equation
set, time step, physical parameters, initial condition as a function of the gridpoints.  Nothing here touches oracle/."""
import numpy as np

VARS6 = {"h": 1, "u": 2, "v": 3, "ub": 4, "vb": 5, "wb": 6}
BCL6 = {"h": "R1T1", "u": "R1T0", "v": "R1T0", "ub": "R1T0", "vb": "R1T0", "wb": "R1T1"}   # models/cha_bell2024/*.jl
BCR6 = {"h": "R0", "u": "R1T1", "v": "R0", "ub": "R1T1", "vb": "R0", "wb": "R0"}


def config1_r(num_cells=100):
    """configs[0]: models/LinearAdvection1D.jl (R grid, 100 cells on [-50, 50], PERIODIC, ts 0.05) with the notebook's initial condition."""
    return dict(name="config1", grid=dict(geometry="R", xmin=-50.0, xmax=50.0, num_cells=num_cells, vars={"u": 1},
                                          BCL={"u": "PERIODIC"}, BCR={"u": "PERIODIC"}),
                eq="LinearAdvection1D", ts=0.05, par=dict(c_0=1.0, K=0.0), ic=lambda p: np.exp(-(p[:, 0] / 20.0) ** 2)[:, None])


def config2_literal(twoway=False):
    """configs[1]: models/cha_bell2024/Oneway_ShallowWater_Slab.jl verbatim (100 cells, native ragged rings,
    N = 181,800) with the notebook's Rankine vortex + wave-2 perturbation (Cha_Bell_WCD2024_initialization.ipynb)."""
    def ic(p):
        r, l = p[:, 0], p[:, 1]
        Rmax, Vmax, eps, f = 50000.0, 50.0, 5000.0, 5.0e-5
        V0 = Vmax / Rmax
        zeta = 2.0 * V0
        vbar = np.where(r < Rmax, V0 * r, Rmax * Rmax * V0 / r)
        # gradient-balanced height dh/dr = (f v + v^2 / r) / g in closed form (pointwise, so tiles can call it separately)
        g = 9.81
        h_in = (f * V0 + V0 * V0) * r * r / (2.0 * g)
        h_rmax = (f * V0 + V0 * V0) * Rmax * Rmax / (2.0 * g)
        rs = np.maximum(r, Rmax)
        h_out = h_rmax + (f * Rmax ** 2 * V0 * np.log(rs / Rmax) + 0.5 * Rmax ** 4 * V0 ** 2 * (1.0 / Rmax ** 2 - 1.0 / rs ** 2)) / g
        h = np.where(r < Rmax, h_in, h_out)
        inner = r < Rmax
        vp = np.where(inner, 0.5 * zeta * r * (eps * np.cos(2 * l) / Rmax),
                      0.5 * zeta * (Rmax ** 2 / r) * (-eps * np.cos(2 * l) * Rmax / r ** 2))
        up = np.where(inner, 0.5 * zeta * r * (eps * np.sin(2 * l) / Rmax),
                      0.5 * zeta * (Rmax ** 2 / r) * (eps * np.sin(2 * l) * Rmax / r ** 2))
        return np.stack([h, up, vbar + vp, 0.8 * up, 0.8 * (vbar + vp), 0.0 * r], axis=1)
    par = dict(g=9.81, K=5000.0, Cd=2.4e-3, Hfree=2000.0, Hb=1000.0, f=5.0e-5, S1=1.0e-4)
    return dict(name="config2", grid=dict(geometry="RL", xmin=0.0, xmax=3.0e5, num_cells=100, vars=VARS6, BCL=BCL6, BCR=BCR6),
                eq="Twoway_ShallowWater_Slab" if twoway else "Oneway_ShallowWater_Slab", ts=3.0, par=par, ic=ic)


def config3_rz(num_cells=171, zDim=128):
    """configs[2]: RZ 513 x 128 with Chebyshev vertical (b_zDim = zDim) and the semi-implicit adjustment
    (src/semiimplicit.jl:521-597; LinearAcousticRZ = Euler_test's variable layout and implicit terms, src/testModels.jl:188-205,
    with a linearised pressure-gradient force).

    SURVEY.md 8(d) asks for a step with c = (1.25 ts)^2 Pxi_bar (2 / Lz)^2 = O(1..100), i.e. a vertical acoustic CFL far
    above 1 that the AI2* adjustment has to absorb: ts = 2 s on a 1 km deep column gives c = 3.0 (vertical CFL ~ 4,600 at
    the 0.15 m end spacing of 128 Chebyshev levels).  Everything that stays explicit must then be stable at ts = 2 s:
    horizontal acoustic waves (sqrt(Pxi_bar) = 346 m/s: DX = 5.8 km, ts c pi / DX = 0.37 < 0.72 for AB3), vertical
    advection and diffusion against the Chebyshev operators' spectral radii (|w| <= 0.02 m/s, K = 1e-3 m^2/s).  The run is
    finite for thousands of steps; an unstable choice (the earlier xmax = 10 km, ts = 0.1 s) blows up after 40."""
    def ic(p):
        r, z = p[:, 0], p[:, 1]
        b = np.exp(-((r - 5.0e5) / 2.0e5) ** 2 - ((z - 500.0) / 200.0) ** 2)
        s = np.sin(np.pi * z / 1.0e3)
        return np.stack([b, 1.0e-3 * b, 0.5 * b, 0.05 * s * b, 0.02 * s * b], axis=1)
    return dict(name="config3", grid=dict(geometry="RZ", xmin=0.0, xmax=1.0e6, num_cells=num_cells, zmin=0.0, zmax=1.0e3,
                                          zDim=zDim, b_zDim=zDim, vars={"s": 1, "xi": 2, "mu": 3, "u": 4, "w": 5},
                                          BCB={"w": "R1T0"}, BCT={"w": "R1T0"}),
                eq="LinearAcousticRZ", ts=2.0, par=dict(K=1.0e-3, Pxi_bar=1.2e5), ic=ic, semiimplicit=True)


def model_parameters(S, case, storage="f64"):
    """GridParameters / ModelParameters of the host mirror (scythe.jl_amd) for one of the dictionaries above."""
    g = dict(case["grid"])
    ring_L = g.pop("ring_L", None)
    gp = S.GridParameters(ring_uniform_L=ring_L or 0, storage=storage, **g)
    par = dict(case["par"])
    par.pop("ref_state", None)
    return S.ModelParameters(ts=case["ts"], equation_set=case["eq"], grid_params=gp, physical_params=par,
                             options={"semiimplicit": case.get("semiimplicit", False)})
