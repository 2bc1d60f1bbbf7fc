"""Host mirror of src/thermodynamics.jl (vectorised numpy): what the reference-state construction and output
diagnostics need.  The per-step thermodynamics of the equation sets runs on the device (csrc/sx_kernels.hip, namespace
thermo); these functions are set-up / post-processing helpers with the reference's names and argument order."""
import numpy as np

# Constants from Emanuel (1994)  (src/thermodynamics.jl:1-17)
Rd = 287.04
Rv = 461.50
Eps = Rd / Rv
Cvd = 716.96
Cvv = 1410.0
Cpd = Cvd + Rd
Cpv = Cvv + Rv
Cl = 4186.0
Ci = 2106.0
gravity = 9.81
L_v0 = 2.501e6
T_0 = 273.16
p_0 = 1000.0
q0 = 1.0e-7


def sat_pressure_liquid(Tk):                      # :19-23
    Tc = np.asarray(Tk, dtype=float) - 273.15
    return 6.112 * np.exp(17.67 * Tc / (Tc + 243.5))


rho_d0 = 100.0 * p_0 / (T_0 * Rd)                 # :31-32
rho_v0 = 100.0 * float(sat_pressure_liquid(T_0)) / (T_0 * Rv)


def L_v(Tk):                                      # :41-44
    return L_v0 + ((Cpv - Cl) * (np.asarray(Tk, dtype=float) - T_0))


def entropy(Tk, rho_d, q_v):                      # :46-56
    Tk, rho_d, q_v = (np.asarray(x, dtype=float) for x in (Tk, rho_d, q_v))
    with np.errstate(divide="ignore", invalid="ignore"):
        qfactor = np.where(q_v != 0.0, q_v * (Rv * np.log(q_v * rho_d / rho_v0) - (L_v(T_0) / T_0)), 0.0)
    Cfactor = Cvd + (q_v * Cvv)
    return (Cfactor * np.log(Tk / T_0)) - (Rd * np.log(rho_d / rho_d0)) - qfactor


def temperature(s, rho_d, q_v):                   # :67-80
    s, rho_d, q_v = (np.asarray(x, dtype=float) for x in (s, rho_d, q_v))
    Cfactor = Cvd + (q_v * Cvv)
    with np.errstate(divide="ignore", invalid="ignore"):
        qfactor = np.where(q_v != 0.0, (rho_d * q_v / rho_v0) ** ((q_v * Rv) / Cfactor), 1.0)
    rhofactor = (rho_d / rho_d0) ** (Rd / Cfactor)
    Tfactor = np.exp((s - (q_v * L_v(T_0) / T_0)) / Cfactor)
    return T_0 * Tfactor * rhofactor * qfactor


def pressure(s, rho_d, q_v):                      # :82-88   (hPa)
    Tk = temperature(s, rho_d, q_v)
    return (0.01 * Rd * Tk * rho_d) + (0.01 * Rv * Tk * rho_d * q_v)


def vapor_pressure(p, q_v):                       # :90-95
    return (p * q_v) / (Eps + q_v)


def mixing_ratio(p, e):                           # :97-100
    return (Eps * e) / (p - e)


def bhyp(q_v):                                    # :184-188
    q_v = np.asarray(q_v, dtype=float)
    return 0.5 * ((q_v + q0) - (q0 * q0 / (q_v + q0)))


def ahyp(mu):                                     # :190-198
    mu = np.asarray(mu, dtype=float)
    return np.where(mu < 0.0, 0.0, np.sqrt(mu * mu + q0 * q0) + mu - q0)


def dmudq(mu, q_v):                               # :200-203
    return ((q_v + q0) - mu) / (q_v + q0)


def dry_density(xi):                              # :205-208
    return rho_d0 * np.exp(xi)


def log_dry_density(rho_d):                       # :210-213
    return np.log(np.asarray(rho_d, dtype=float) / rho_d0)


def P_s(Tk, rho_d, q_v):                          # :215-219
    return Tk * ((rho_d * Rd) + (q_v * rho_d * Rv)) / (Cvd + (q_v * Cvv))


def P_xi(Tk, rho_d, q_v):                         # :221-224
    return (Rd + (q_v * rho_d * Rv)) * ((rho_d * Tk) + P_s(Tk, rho_d, q_v))


def thermodynamic_tuple(s, xi, mu):               # :260-269
    q_v = ahyp(mu)
    rho_d = dry_density(xi)
    Tk = temperature(s, rho_d, q_v)
    p = (0.01 * Rd * Tk * rho_d) + (0.01 * Rv * Tk * rho_d * q_v)
    return q_v, rho_d, Tk, p


def P_xi_from_s(s, xi, mu):                       # :226-230
    q_v, rho_d, Tk, _ = thermodynamic_tuple(s, xi, mu)
    return P_xi(Tk, rho_d, q_v)
