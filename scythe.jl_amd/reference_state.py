"""Host mirror of src/reference_state.jl: the horizontally uniform ReferenceState (sbar, xibar, mubar profiles with their
first and second vertical derivatives, and the mean squared sound speed Pxi_bar) that Euler_test and the semi-implicit
adjustment are linearised about.  One-time set-up work on the host; the Chebyshev column operations come from the library
(sx_cheb_column_ops), so the derivatives are the same discrete operators the device applies."""
import ctypes as C
from dataclasses import dataclass

import numpy as np

from . import _lib as L
from . import thermodynamics as T


class Chebyshev1D:
    """One Chebyshev column (Springsteel Chebyshev1D): CBtransform! -> CAtransform! -> CItransform! / CIxtransform /
    CIxxtransform / CIInttransform as dense collocation operators on the column's values (index 0 = bottom)."""

    def __init__(self, zmin, zmax, zDim, bDim=0, BCB="R0", BCT="R0"):
        n = int(zDim)
        self.z = np.zeros(n)
        self._rec, self._dz, self._dzz, self._int = (np.zeros((n, n)) for _ in range(4))
        p = lambda a: a.ctypes.data_as(L.P_D)
        L.check(L.load().sx_cheb_column_ops(C.c_double(zmin), C.c_double(zmax), n, int(bDim or 0), L.BC[BCB], L.BC[BCT],
                                            p(self.z), p(self._rec), p(self._dz), p(self._dzz), p(self._int)))
        self.uMish = np.zeros(n)

    def CItransform(self):
        return self._rec @ self.uMish

    def CIxtransform(self):
        return self._dz @ self.uMish

    def CIxxtransform(self):
        return self._dzz @ self.uMish

    def CIInttransform(self, C0=0.0):
        return C0 + self._int @ self.uMish


@dataclass
class ReferenceState:                              # src/reference_state.jl:4-10
    sbar: np.ndarray                               # [zDim, 3]: value, d/dz, d2/dz2
    xibar: np.ndarray
    mubar: np.ndarray
    mu_lbar: np.ndarray
    Pxi_bar: float

    def packed(self):
        """[3][3][zDim] block for sx_model_desc.ref_state."""
        return np.ascontiguousarray(np.stack([self.sbar.T, self.xibar.T, self.mubar.T]), dtype=np.float64)


def _column(gp):
    return Chebyshev1D(gp.zmin, gp.zmax, gp.zDim, gp.b_zDim, "R0", "R0")          # :95-102, :143-150


def transform_reference_state(gp, ref):            # src/reference_state.jl:140-160
    """Filtered values and vertical derivatives without BCs, in place on ref [zDim, 3]."""
    col = _column(gp)
    col.uMish[:] = ref[:, 0]
    ref[:, 0] = col.CItransform()
    ref[:, 1] = col.CIxtransform()
    ref[:, 2] = col.CIxxtransform()
    return ref


def _finish(gp, sbar, xibar, mubar, mu_lbar):
    for r in (sbar, xibar, mubar):
        transform_reference_state(gp, r)
    Pxi = T.P_xi_from_s(sbar[:, 0], xibar[:, 0], mubar[:, 0])                       # :126-132
    rho_bar = T.dry_density(xibar[:, 0])
    q_bar = T.ahyp(mubar[:, 0])
    return ReferenceState(sbar, xibar, mubar, mu_lbar, float(np.mean(Pxi / (rho_bar * (1.0 + q_bar)))))


def reference_state_from_sounding(gp, sfc_pressure, alt, theta_in, q_v_in):
    """interpolate_reference_file after the file has been parsed (src/reference_state.jl:43-134): sounding levels
    alt [m] (alt[0] = 0), potential temperature [K], vapour mixing ratio [g/kg]; surface pressure in hPa."""
    z = _column(gp).z
    alt, theta_in, q_v_in = (np.asarray(a, dtype=float) for a in (alt, theta_in, q_v_in))
    n = len(z)
    theta, q_v = np.zeros(n), np.zeros(n)
    theta[0], q_v[0] = theta_in[0], q_v_in[0]      # assumes the first level in both cases is the surface
    for i in range(1, n):
        found = False
        for j in range(1, len(alt)):
            if alt[j - 1] < z[i] < alt[j]:
                f = (z[i] - alt[j - 1]) / (alt[j] - alt[j - 1])
                theta[i] = theta_in[j - 1] + f * (theta_in[j] - theta_in[j - 1])
                q_v[i] = q_v_in[j - 1] + f * (q_v_in[j] - q_v_in[j - 1])
                found = True
            elif alt[j] == z[i]:
                theta[i], q_v[i] = theta_in[j], q_v_in[j]
                found = True
        if not found:
            raise ValueError("Can't find an interpolating level for reference state (level %d)" % (i + 1))   # DomainError :66
    q_v = q_v * 1.0e-3
    # first-guess hydrostatic integration, level by level (:70-92)
    Tk, p, rho_t = np.zeros(n), np.zeros(n), np.zeros(n)
    p[0] = sfc_pressure
    for i in range(n):
        if i > 0:
            p[i] = np.exp(np.log(p[i - 1]) + dlnpdz * (z[i] - z[i - 1]))
        Tk[i] = theta[i] / (T.p_0 / p[i]) ** (T.Rd / T.Cpd)
        e = T.vapor_pressure(p[i], q_v[i])
        rho_d = 100.0 * (p[i] - e) / (Tk[i] * T.Rd)
        rho_t[i] = rho_d * (1.0 + q_v[i])
        dlnpdz = -T.gravity * rho_t[i] / (p[i] * 100.0)
    # re-integrate with the Chebyshev column to adjust T (:94-111)
    col = _column(gp)
    col.uMish[:] = -T.gravity * rho_t
    p_new = col.CIInttransform(sfc_pressure * 100.0) / 100.0
    Tk = theta / (T.p_0 / p_new) ** (T.Rd / T.Cpd)
    e = T.vapor_pressure(p_new, q_v)
    rho_d = 100.0 * (p_new - e) / (Tk * T.Rd)
    sbar, xibar, mubar, mu_lbar = (np.zeros((n, 3)) for _ in range(4))
    sbar[:, 0] = T.entropy(Tk, rho_d, q_v)
    xibar[:, 0] = T.log_dry_density(rho_d)
    mubar[:, 0] = T.bhyp(q_v)
    return _finish(gp, sbar, xibar, mubar, mu_lbar)


def interpolate_reference_file(model, z=None):     # src/reference_state.jl:17-135
    """Sounding file: first line `sfc_pressure[hPa] theta[K] q_v[g/kg]`, then `altitude[m] theta q_v` per level."""
    with open(model.ref_state_file) as f:
        rows = [ln.split() for ln in f if ln.strip()]
    sfc = float(rows[0][0])
    alt = [0.0] + [float(r[0]) for r in rows[1:]]
    theta = [float(rows[0][1])] + [float(r[1]) for r in rows[1:]]
    q_v = [float(rows[0][2])] + [float(r[2]) for r in rows[1:]]
    return reference_state_from_sounding(model.grid_params, sfc, alt, theta, q_v)


def exact_reference_state(model, z=None):          # src/reference_state.jl:162-199
    """File already in hydrostatic balance: `z sbar xibar mubar mu_lbar` per model level."""
    gp = model.grid_params
    zl = _column(gp).z
    sbar, xibar, mubar, mu_lbar = (np.zeros((len(zl), 3)) for _ in range(4))
    with open(model.ref_state_file) as f:
        for i in range(len(zl)):
            parts = f.readline().split()
            if abs(float(parts[0]) - zl[i]) > 1e-9 * max(1.0, abs(zl[i])):
                raise ValueError("Model level does not match reference level (level %d)" % (i + 1))       # DomainError :178
            sbar[i, 0], xibar[i, 0], mubar[i, 0], mu_lbar[i, 0] = (float(x) for x in parts[1:5])
    transform_reference_state(gp, mu_lbar)
    return _finish(gp, sbar, xibar, mubar, mu_lbar)
