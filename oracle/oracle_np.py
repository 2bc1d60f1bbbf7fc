"""
TEST INFRASTRUCTURE — NOT PRODUCT CODE.

numpy "definition" oracle for the Scythe.jl spectral-transform time-stepping hot path.
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.

It restates, as dense linear operators (O(N^2) on purpose - it is a definition, not an
implementation), the algorithm that the reference obtains from the external, un-vendored
package Springsteel.jl (Project.toml:20, unpinned) at these call sites:

  spectralTransform!  (forward, physical -> B coefficients)   src/semiimplicit.jl:135,734
  splineTransform!    (B -> A banded SPD solve with patch BCs) src/semiimplicit.jl:237,285
  tileTransform!      (A -> physical value + derivatives)      src/semiimplicit.jl:241,305
  Chebyshev column ops CB/CA/CI/CIx/CIxx/CIInt                src/semiimplicit.jl:569-596,
                                                              src/shallowWaterModels.jl:423-504
and the in-tree step structure (src/semiimplicit.jl:301-332, 672-698, 521-597, 768-781) plus
equation sets (src/testModels.jl:1-98, src/shallowWaterModels.jl:1-233, 346-511).

Parity status:
  * R-grid cubic-B-spline + AB3 pipeline: PINNED by the reference's notebook known-answer
    (notebooks/LinearAdvection_example.ipynb cells 2, 7, 9) -> tests/golden/linear_advection_kat.json.
  * Fourier / Chebyshev / RL / RZ / RLZ layouts / semi-implicit solve: PARITY UNPINNED (no reference
    fixture exists, Springsteel source and Julia are absent); these follow SURVEY.md 8(c)'s spec.
"""
import numpy as np

# ----------------------------------------------------------------------------- constants
SQRT35 = np.sqrt(3.0 / 5.0)
GAUSS_OFF = np.array([-SQRT35 / 2.0, 0.0, SQRT35 / 2.0])   # mish offsets inside a unit cell, centred
QUAD_W = np.array([8.0, 5.0, 8.0]) / 21.0                  # ratio 8:5:8 pinned by the notebook KAT
MUBAR = 3
XP = np.longdouble            # x87 80-bit extended precision on this platform
PI_X = 4 * np.arctan(XP(1))

BC_RANK = {"R0": 0, "R1T0": 1, "R1T1": 1, "R1T2": 1, "R2T10": 2, "R2T20": 2, "R3": 3}
R1_COEF = {"R1T0": (-4.0, -1.0), "R1T1": (0.0, 1.0), "R1T2": (2.0, -1.0)}


# ----------------------------------------------------------------------------- cubic B-spline
def bspline(delta, d):
    """d-th derivative (w.r.t. delta) of the cardinal cubic B-spline, vectorised."""
    delta = np.asarray(delta)
    if delta.dtype != XP:
        delta = delta.astype(np.float64)
    z = np.abs(delta)
    s = np.where(delta > 0, 1.0, -1.0)
    p = 2.0 - z
    q = np.maximum(1.0 - z, 0.0)
    if d == 0:
        out = p ** 3 / 6.0 - 4.0 * q ** 3 / 6.0
    elif d == 1:
        out = -s * (p * p / 2.0 - 2.0 * q * q)
    elif d == 2:
        out = p - 4.0 * q
    else:
        out = s * np.where(z < 1.0, 3.0, -1.0)
    return np.where(z < 2.0, out, 0.0)


def gamma_matrix(nc, bcl, bcr):
    """BC projection Gamma (nfree x bDim): full coefficients a = Gamma^T a_free."""
    bdim = nc + 3
    if bcl == "PERIODIC" or bcr == "PERIODIC":
        assert bcl == bcr == "PERIODIC"
        G = np.zeros((nc, bdim))
        for m in range(-1, nc + 2):
            G[m % nc, m + 1] = 1.0
        return G
    rl, rr = BC_RANK[bcl], BC_RANK[bcr]
    nfree = bdim - rl - rr
    G = np.zeros((nfree, bdim))
    for j in range(nfree):
        G[j, rl + j] = 1.0
    # left boundary: dependent coefficients expressed through the first free ones
    if bcl in R1_COEF:
        al, be = R1_COEF[bcl]
        G[0, 0] += al
        G[1, 0] += be
    elif bcl == "R2T10":
        G[0, 0] += 1.0
        G[0, 1] += -0.5
    elif bcl == "R2T20":
        G[0, 0] += -1.0
    # right boundary, mirrored
    if bcr in R1_COEF:
        al, be = R1_COEF[bcr]
        G[nfree - 1, bdim - 1] += al
        G[nfree - 2, bdim - 1] += be
    elif bcr == "R2T10":
        G[nfree - 1, bdim - 1] += 1.0
        G[nfree - 1, bdim - 2] += -0.5
    elif bcr == "R2T20":
        G[nfree - 1, bdim - 1] += -1.0
    return G


class Spline1D:
    """Cubic B-spline on [xmin, xmax] with nc cells; nodes m = -1..nc+1 (array index m+1)."""

    def __init__(self, xmin, xmax, nc, l_q=2.0, bcl="R0", bcr="R0"):
        self.xmin, self.xmax, self.nc = float(xmin), float(xmax), int(nc)
        self.DX = (self.xmax - self.xmin) / self.nc
        self.bdim = self.nc + 3
        self.l_q, self.bcl, self.bcr = l_q, bcl, bcr
        self.mish = mish_points(self.xmin, self.DX, 0, self.nc)
        self.W = np.tile(self.DX * QUAD_W, self.nc)
        ph0 = self.basis(self.mish, 0)
        ph3 = self.basis(self.mish, 3)
        eps_q = (l_q * self.DX / (2.0 * np.pi)) ** 6
        self.P = ph0.T @ (self.W[:, None] * ph0) + eps_q * (ph3.T @ (self.W[:, None] * ph3))
        self.G = gamma_matrix(self.nc, bcl, bcr)
        self.PQ = self.G @ self.P @ self.G.T
        self.cho = np.linalg.cholesky(self.PQ)

    def basis(self, x, d):
        """Phi_d[i, m+1] = d^d/dx^d phi_m(x_i)."""
        x = np.asarray(x, dtype=np.float64)
        xm = self.xmin + (np.arange(self.bdim) - 1) * self.DX
        delta = (x[:, None] - xm[None, :]) / self.DX
        return bspline(delta, d) / self.DX ** d

    def SB(self, u):           # physical at own mish -> B
        return self.basis(self.mish, 0).T @ (self.W * u)

    def SA(self, b):           # B -> A (with BCs)
        y = np.linalg.solve(self.cho, self.G @ b)
        return self.G.T @ np.linalg.solve(self.cho.T, y)


def mish_points(xmin, DX, cell0, ncells):
    c = np.arange(cell0, cell0 + ncells)
    return (xmin + DX * (c[:, None] + 0.5 + GAUSS_OFF[None, :])).reshape(-1)


# ----------------------------------------------------------------------------- Fourier ring
class Ring:
    def __init__(self, L, kmax, offset):
        self.L, self.kmax, self.offset = int(L), int(kmax), float(offset)
        self.lam = self.offset + 2.0 * np.pi * np.arange(self.L) / self.L
        # angles in extended precision, rounded once: in Float64 the product k * lambda alone loses k ulp (1e-13 at k = 300)
        lam_x = XP(self.offset) + 2 * PI_X * np.arange(self.L, dtype=XP) / XP(self.L)
        nb = 1 + 2 * self.kmax
        self.FB = np.zeros((nb, self.L))
        self.FI = [np.zeros((self.L, nb)) for _ in range(3)]
        self.FB[0, :] = 1.0 / self.L
        self.FI[0][:, 0] = 1.0
        for k in range(1, self.kmax + 1):
            c, s = np.cos(k * lam_x).astype(np.float64), np.sin(k * lam_x).astype(np.float64)
            self.FB[2 * k - 1] = c / self.L
            self.FB[2 * k] = -s / self.L
            self.FI[0][:, 2 * k - 1] = 2.0 * c
            self.FI[0][:, 2 * k] = -2.0 * s
            self.FI[1][:, 2 * k - 1] = -2.0 * k * s
            self.FI[1][:, 2 * k] = -2.0 * k * c
            self.FI[2][:, 2 * k - 1] = -2.0 * k * k * c
            self.FI[2][:, 2 * k] = 2.0 * k * k * s


# ----------------------------------------------------------------------------- Chebyshev column


class Cheb:
    """Chebyshev-Gauss-Lobatto column, index 0 = bottom (z = zmin, x = +1).

    The operators are assembled in extended precision and rounded to double once: the second-derivative collocation
    matrix has entries O(N^4 / Lz^2), so factors rounded to double before multiplication would put cond * eps
    (1e-9 at N = 128) of avoidable noise into every product."""

    def __init__(self, zmin, zmax, N, bdim=None, bcb="R0", bct="R0"):
        self.zmin, self.zmax, self.N = float(zmin), float(zmax), int(N)
        self.bdim = int(bdim) if bdim else default_bzdim(N)
        N = self.N
        n = np.arange(N)
        self.x = np.cos(n * np.pi / (N - 1))
        self.z = self.x * (-0.5 * (self.zmax - self.zmin)) + 0.5 * (self.zmin + self.zmax)
        w = np.full(N, 2.0, dtype=XP)
        w[0] = w[-1] = 1.0
        nk = np.outer(n, n).astype(XP)
        Tx = w[None, :] * np.cos(nk * PI_X / XP(N - 1))          # dct_matrix: a -> values
        CBx = (Tx / XP(2 * (N - 1)))[: self.bdim, :]             # values -> b (truncated)
        Lz = XP(self.zmax) - XP(self.zmin)
        # coefficient-space derivative (a -> ax), DCT-I normalisation u = a0 + 2 sum a_k T_k + a_{N-1} T_{N-1}
        Dc = np.zeros((N, N), dtype=XP)
        for j in range(N):
            a = np.zeros(N, dtype=XP)
            a[j] = 1.0
            ax = np.zeros(N + 2, dtype=XP)
            for k in range(N - 1, 0, -1):
                ck = a[k] if k == N - 1 else 2 * a[k]
                ax[k - 1] = ax[k + 1] + k * ck
            Dc[:, j] = ax[:N]
        Dcx = Dc * (XP(-2) / Lz)
        # coefficient-space indefinite integral (zero at the bottom)
        Ic = np.zeros((N, N), dtype=XP)
        for j in range(N):
            a = np.zeros(N, dtype=XP)
            a[j] = 1.0
            ai = np.zeros(N, dtype=XP)
            for k in range(1, N - 1):
                up = a[k + 1] if k + 1 < N - 1 else a[k + 1] / 2
                ai[k] = (a[k - 1] - up) / XP(2 * k)
            ai[N - 1] = a[N - 2] / XP(N - 1)
            ai *= (XP(-0.5) * Lz)
            ai[0] = -(2 * ai[1:N - 1].sum() + ai[N - 1])
            Ic[:, j] = ai
        TDx = Tx @ Dcx
        TDDx = TDx @ Dcx
        # BC projection in coefficient space (orthogonal projection onto the constraint null space)
        rows = []
        for bc, row in ((bcb, 0), (bct, N - 1)):
            if bc == "R0":
                continue
            if bc == "R1T0":
                rows.append(Tx[row, :])
            elif bc == "R1T1":
                rows.append(TDx[row, :])
            elif bc == "R1T2":
                rows.append(TDDx[row, :])
            else:
                raise ValueError("unsupported vertical BC " + bc)
        pad = np.zeros((N, self.bdim), dtype=XP)
        pad[: self.bdim, : self.bdim] = np.eye(self.bdim, dtype=XP)
        proj = np.eye(N, dtype=XP)
        if rows:
            Cm = np.array(rows, dtype=XP)
            G = Cm @ Cm.T
            if len(rows) == 1:
                Gi = 1 / G
            else:
                det = G[0, 0] * G[1, 1] - G[0, 1] * G[1, 0]
                Gi = np.array([[G[1, 1], -G[0, 1]], [-G[1, 0], G[0, 0]]], dtype=XP) / det
            proj = proj - Cm.T @ Gi @ Cm
        CAx = proj @ pad                                # b -> a
        f64 = lambda m: np.asarray(m, dtype=np.float64)
        self._x = dict(T=Tx, Dc=Dcx, TD=TDx, TDD=TDDx, CA=CAx, CB=CBx)   # extended-precision factors (Helmholtz assembly, inverse_xp)
        self.T, self.Dc, self.CBm, self.CAm, self.Ic = f64(Tx), f64(Dcx), f64(CBx), f64(CAx), f64(Ic)
        self.M = [f64(Tx @ CAx), f64(TDx @ CAx), f64(TDDx @ CAx)]       # b -> values, d/dz, d2/dz2
        self.Mint = f64(Tx @ Ic @ CAx)                                     # b -> integral from the bottom
        # full column operators values -> values (CB, CA, then CI / CIx / CIInt), products taken before rounding
        self.Vrec, self.Vdz, self.Vint = f64(Tx @ CAx @ CBx), f64(TDx @ CAx @ CBx), f64(Tx @ Ic @ CAx @ CBx)

    # dense collocation matrices used by calc_Helmholtz_semiimplicit_matrix (src/semiimplicit.jl:772-775)
    def dct_matrix(self):
        return self.T

    def dct_1st_derivative(self):
        return np.asarray(self._x["TD"], dtype=np.float64)

    def dct_2nd_derivative(self):
        return np.asarray(self._x["TDD"], dtype=np.float64)


def default_bzdim(zdim):
    return int(min(zdim, np.floor((2 * zdim - 1) / 3) + 1))


# ----------------------------------------------------------------------------- grid
DERIV_SLOTS = {"R": ["u", "r", "rr"], "RZ": ["u", "r", "rr", "z", "zz"],
               "RL": ["u", "r", "rr", "l", "ll"], "RLZ": ["u", "r", "rr", "l", "ll", "z", "zz"]}


class Grid:
    """Patch description + tile-level transforms. Tiles are (cell0, ncells) sub-ranges of the patch."""

    def __init__(self, geometry, xmin, xmax, num_cells, vars, BCL=None, BCR=None, l_q=2.0,
                 zmin=0.0, zmax=0.0, zDim=0, b_zDim=None, BCB=None, BCT=None,
                 ring_L=None, BCL_k0=None):
        self.geometry = geometry
        self.xmin, self.xmax, self.nc = float(xmin), float(xmax), int(num_cells)
        self.DX = (self.xmax - self.xmin) / self.nc
        self.vars = dict(vars)                                   # name -> 1-based index
        self.names = [n for n, _ in sorted(self.vars.items(), key=lambda kv: kv[1])]
        self.V = len(self.names)
        self.l_q = l_q
        dflt = lambda d, v: (d or {}).get(v, "R0")
        self.BCL = {v: dflt(BCL, v) for v in self.names}
        self.BCR = {v: dflt(BCR, v) for v in self.names}
        self.BCL_k0 = {v: (BCL_k0 or {}).get(v, self.BCL[v]) for v in self.names}
        self.BCB = {v: dflt(BCB, v) for v in self.names}
        self.BCT = {v: dflt(BCT, v) for v in self.names}
        self.has_l = "L" in geometry
        self.has_z = "Z" in geometry
        self.rDim = MUBAR * self.nc
        self.b_rDim = self.nc + 3
        self.zDim = int(zDim) if self.has_z else 1
        self.b_zDim = (int(b_zDim) if b_zDim else default_bzdim(self.zDim)) if self.has_z else 1
        self.zmin, self.zmax = zmin, zmax
        self.slots = DERIV_SLOTS[geometry]
        self.D = len(self.slots)
        # ring table (patch-level, ring index ri = 1..rDim)
        if self.has_l:
            ri = np.arange(1, self.rDim + 1)
            if ring_L is None:                     # native Springsteel layout
                self.L = 4 + 4 * ri
                self.kmax = ri.copy()
                self.off = 0.5 * (2.0 * np.pi / self.L) * (ri - 1)
            else:                                  # uniform ring table (perf shape)
                self.L = np.full(self.rDim, int(ring_L))
                self.kmax = np.minimum(ri, int(ring_L) // 2 - 1)
                self.off = np.zeros(self.rDim)
        else:
            self.L = np.ones(self.rDim, dtype=int)
            self.kmax = np.zeros(self.rDim, dtype=int)
            self.off = np.zeros(self.rDim)
        self.ringstart = np.concatenate([[0], np.cumsum(self.L)])
        self.kDim = int(self.kmax.max())
        self.K2 = 1 + 2 * self.kDim
        self._rings = {}
        self._spl = {}
        self._cheb = {}

    # -- helpers
    @property
    def rings(self):
        return _LazyRings(self)

    def spline(self, bcl, bcr):
        key = (bcl, bcr)
        if key not in self._spl:
            self._spl[key] = Spline1D(self.xmin, self.xmax, self.nc, self.l_q, bcl, bcr)
        return self._spl[key]

    def cheb(self, v):
        key = (self.BCB[v], self.BCT[v])
        if key not in self._cheb:
            self._cheb[key] = Cheb(self.zmin, self.zmax, self.zDim, self.b_zDim, *key)
        return self._cheb[key]

    def var_spline(self, v, blk):
        bcl = self.BCL_k0[v] if blk == 0 else self.BCL[v]
        return self.spline(bcl, self.BCR[v])

    def tile_rings(self, cell0, ncells):
        return range(MUBAR * cell0, MUBAR * (cell0 + ncells))

    def tile_npoints(self, cell0, ncells):
        rr = self.tile_rings(cell0, ncells)
        return int((self.ringstart[rr[-1] + 1] - self.ringstart[rr[0]]) * self.zDim)

    def tile_K2(self, cell0, ncells):
        rr = self.tile_rings(cell0, ncells)
        return 1 + 2 * int(self.kmax[rr[0]:rr[-1] + 1].max())

    def S_patch(self):
        return self.b_zDim * self.K2 * self.b_rDim

    def gridpoints(self, cell0=0, ncells=None):
        ncells = self.nc if ncells is None else ncells
        r = mish_points(self.xmin, self.DX, cell0, ncells)
        zc = self.cheb(self.names[0]).z if self.has_z else np.zeros(1)
        nz = self.zDim
        blocks = []            # one block per ring: point order (l, z), z fastest - per element the same operations as a scalar loop
        for i, ring in enumerate(self.tile_rings(cell0, ncells)):
            L = int(self.L[ring])
            cols = [np.full(L * nz, r[i])]
            if self.has_l:
                lam = self.off[ring] + 2.0 * np.pi * np.arange(L) / L
                cols.append(np.repeat(lam, nz))
            if self.has_z:
                cols.append(np.tile(zc[:nz], L))
            blocks.append(np.stack(cols, axis=1))
        out = np.concatenate(blocks, axis=0)
        return out[:, 0] if out.shape[1] == 1 else out

    # -- forward: tile physical values [N_t, V] -> tile B coefficients, reference tile layout
    #    spectral[(zm*K2_t + blk)*(ncells+3) + j, v]
    def forward(self, values, cell0=0, ncells=None):
        ncells = self.nc if ncells is None else ncells
        rr = list(self.tile_rings(cell0, ncells))
        K2t = self.tile_K2(cell0, ncells)
        nbt = ncells + 3
        tile_spl = Spline1D(self.xmin + cell0 * self.DX, self.xmin + (cell0 + ncells) * self.DX, ncells)
        ph0 = tile_spl.basis(tile_spl.mish, 0)                 # [3*ncells, nbt]
        # quadrature weights use the patch DX (identical up to rounding)
        W = np.tile(self.DX * QUAD_W, ncells)
        out = np.zeros((self.b_zDim * K2t * nbt, self.V))
        base = self.ringstart[rr[0]]
        for vi, v in enumerate(self.names):
            ch = self.cheb(v) if self.has_z else None
            F = np.zeros((len(rr), self.b_zDim, K2t))           # ring, zmode, block
            for i, ring in enumerate(rr):
                L = self.L[ring]
                p0 = (self.ringstart[ring] - base) * self.zDim
                u = values[p0:p0 + L * self.zDim, vi].reshape(L, self.zDim)   # [lambda, z]
                bz = (ch.CBm @ u.T) if self.has_z else u.T                    # [zm, lambda]
                c = bz @ self.rings[ring].FB.T                                # [zm, blocks of ring]
                F[i, :, :c.shape[1]] = c
            for zm in range(self.b_zDim):
                for blk in range(K2t):
                    b = ph0.T @ (W * F[:, zm, blk])
                    o = (zm * K2t + blk) * nbt
                    out[o:o + nbt, vi] = b
        return out

    # -- the shared-array sum protocol of src/semiimplicit.jl:320-329, 279-282
    def add_tile_to_shared(self, shared, btile, cell0, ncells, last):
        K2t = self.tile_K2(cell0, ncells)
        nbt = ncells + 3
        for zm in range(self.b_zDim):
            for blk in range(K2t):
                o = (zm * K2t + blk) * nbt
                p = (zm * self.K2 + blk) * self.b_rDim + cell0
                shared[p:p + nbt, :] += btile[o:o + nbt, :]
        return shared

    # -- B -> A for the whole patch (reference layout [S_patch, V])
    def spline_transform(self, shared):
        A = np.zeros_like(shared)
        for vi, v in enumerate(self.names):
            for zm in range(self.b_zDim):
                for blk in range(self.K2):
                    p = (zm * self.K2 + blk) * self.b_rDim
                    A[p:p + self.b_rDim, vi] = self.var_spline(v, blk).SA(shared[p:p + self.b_rDim, vi])
        return A

    # -- inverse: patch A -> tile physical [N_t, V, D]
    def inverse(self, A, cell0=0, ncells=None):
        ncells = self.nc if ncells is None else ncells
        rr = list(self.tile_rings(cell0, ncells))
        Nt = self.tile_npoints(cell0, ncells)
        phys = np.zeros((Nt, self.V, self.D))
        r = mish_points(self.xmin, self.DX, cell0, ncells)
        spl = self.spline("R0", "R0")
        PH = [spl.basis(r, d) for d in range(3)]               # [rings, b_rDim]
        base = self.ringstart[rr[0]]
        sl = {s: i for i, s in enumerate(self.slots)}
        for vi, v in enumerate(self.names):
            ch = self.cheb(v) if self.has_z else None
            Av = A[:, vi].reshape(self.b_zDim, self.K2, self.b_rDim)
            for i, ring in enumerate(rr):
                rg = self.rings[ring]
                nb = 1 + 2 * rg.kmax
                L = rg.L
                p0 = (self.ringstart[ring] - base) * self.zDim
                sel = slice(p0, p0 + L * self.zDim)
                for d, name in enumerate(["u", "r", "rr"]):
                    coef = Av[:, :nb, :] @ PH[d][i]                          # [zm, blocks]
                    lam_sets = [(0, name)] if d > 0 else [(0, "u"), (1, "l"), (2, "ll")]
                    for ld, sname in lam_sets:
                        if sname not in sl:
                            continue
                        f = coef @ rg.FI[ld].T                               # [zm, lambda]
                        if self.has_z:
                            phys[sel, vi, sl[sname]] = (ch.M[0] @ f).T.reshape(-1)
                            if sname == "u":
                                phys[sel, vi, sl["z"]] = (ch.M[1] @ f).T.reshape(-1)
                                phys[sel, vi, sl["zz"]] = (ch.M[2] @ f).T.reshape(-1)
                        else:
                            phys[sel, vi, sl[sname]] = f.T.reshape(-1)
        return phys


def inverse_xp(grid, A, rings, cell0=0):
    """tileTransform! for the listed patch rings in EXTENDED precision (numpy longdouble) from Float64 A coefficients and
    Float64 gridpoints: the arbiter for "which fp64 evaluation of a derivative slot is closer to the exact one".
    Returns {ring: phys[L * zDim, V, D]} (longdouble).  Dense and slow on purpose; pass a sample of rings."""
    g = grid
    sl = {s_: i for i, s_ in enumerate(g.slots)}
    out = {}
    xm = XP(g.xmin) + (np.arange(g.b_rDim, dtype=XP) - 1) * XP(g.DX)
    for ring in rings:
        c = ring // MUBAR
        r64 = mish_points(g.xmin, g.DX, c, 1)[ring % MUBAR]                   # the Float64 gridpoint, as handed to the user
        n0 = c                                                                # the 4 spline nodes that overlap this ring's cell: c .. c + 3
        delta = ((XP(r64) - xm[n0:n0 + 4]) / XP(g.DX))[None, :]
        PH = [bspline(delta, d)[0] / XP(g.DX) ** d for d in range(3)]        # [4]
        L, km = int(g.L[ring]), int(g.kmax[ring])
        nb = 1 + 2 * km
        lam = XP(g.off[ring]) + 2 * PI_X * np.arange(L, dtype=XP) / XP(L)
        FI = [np.zeros((L, nb), dtype=XP) for _ in range(3)]
        FI[0][:, 0] = 1
        for k in range(1, km + 1):
            cs, sn = np.cos(k * lam), np.sin(k * lam)
            FI[0][:, 2 * k - 1], FI[0][:, 2 * k] = 2 * cs, -2 * sn
            FI[1][:, 2 * k - 1], FI[1][:, 2 * k] = -2 * k * sn, -2 * k * cs
            FI[2][:, 2 * k - 1], FI[2][:, 2 * k] = -2 * k * k * cs, 2 * k * k * sn
        phys = np.zeros((L * g.zDim, g.V, g.D), dtype=XP)
        for vi, v in enumerate(g.names):
            Av = A[:, vi].reshape(g.b_zDim, g.K2, -1)[:, :nb, n0:n0 + 4].astype(XP)      # only what this ring reads
            if g.has_z:
                x = g.cheb(v)._x
                Mx = [x["T"] @ x["CA"], x["TD"] @ x["CA"], x["TDD"] @ x["CA"]]
            for d, name in enumerate(["u", "r", "rr"]):
                coef = Av @ PH[d]                                             # [zm, blocks]
                for ld, sname in ([(0, name)] if d > 0 else [(0, "u"), (1, "l"), (2, "ll")]):
                    if sname not in sl:
                        continue
                    f = coef @ FI[ld].T                                       # [zm, lambda]
                    if g.has_z:
                        phys[:, vi, sl[sname]] = (Mx[0] @ f).T.reshape(-1)
                        if sname == "u":
                            phys[:, vi, sl["z"]] = (Mx[1] @ f).T.reshape(-1)
                            phys[:, vi, sl["zz"]] = (Mx[2] @ f).T.reshape(-1)
                    else:
                        phys[:, vi, sl[sname]] = f.T.reshape(-1)
        out[ring] = phys
    return out


class _LazyRings:
    """Dense ring operators are built on first use (the C oracle never needs them)."""

    def __init__(self, grid):
        self.g = grid

    def __getitem__(self, r):
        g = self.g
        if r not in g._rings:
            g._rings[r] = Ring(g.L[r], g.kmax[r], g.off[r])
        return g._rings[r]


# ----------------------------------------------------------------------------- time stepping
def explicit_timestep(t, ts, u, e_n, e_nm1, e_nm2):
    """Euler / AB2 / AB3 (src/semiimplicit.jl:672-698). Returns (u_np1, e_nm1', e_nm2')."""
    if t == 1:
        return u + ts * e_n, e_n.copy(), e_nm2
    if t == 2:
        return u + (0.5 * ts) * ((3.0 * e_n) - e_nm1), e_n.copy(), e_nm1.copy()
    return u + ((ts / 12.0) * ((23.0 * e_n) - (16.0 * e_nm1) + (5.0 * e_nm2))), e_n.copy(), e_nm1.copy()


def helmholtz_matrix(ch, pxi_bar, tau, extended=False):
    """calc_Helmholtz_semiimplicit_matrix (src/semiimplicit.jl:768-781), assembled in extended precision (rounded to
    double unless extended=True).  The literal Float64 assembly is helmholtz_matrix_f64."""
    c = XP(tau) * XP(tau) * XP(pxi_bar)
    dct, dct2 = ch._x["T"], ch._x["TDD"]
    h = c * dct2 - dct
    H = np.vstack([c * dct[0:1, :], c * dct[-1:, :], h[1:-1, :]])
    return H if extended else np.asarray(H, dtype=np.float64)


def helmholtz_matrix_f64(ch, pxi_bar, tau):
    """calc_Helmholtz_semiimplicit_matrix statement by statement in Float64 (src/semiimplicit.jl:768-781):
        dct  = Chebyshev.dct_matrix(nz);  dct2 = Chebyshev.dct_2nd_derivative(nz, column_length)          (:772-775)
        h    = (ts_term .* ts_term .* Pxi_bar) .* dct2 .- dct                                               (:776)
        bc1, bc2 = (ts_term .* ts_term .* Pxi_bar) .* dct[1, :], ... .* dct[nz, :]                          (:777-778)
        h_a  = [bc1'; bc2'; h[2:nz-1, :]]                                                                   (:779)
    with the collocation matrices as Float64 arrays (what the Springsteel functions return)."""
    dct, dct2 = ch.dct_matrix(), ch.dct_2nd_derivative()
    c = (tau * tau) * pxi_bar
    h = c * dct2 - dct
    return np.vstack([(c * dct[0, :])[None, :], (c * dct[-1, :])[None, :], h[1:-1, :]])


class HelmholtzLU:
    """`factorize(h_a)` + `h_a \\ g` as the reference executes them (src/semiimplicit.jl:780, 586-589): Julia's
    factorize of a dense general matrix is LAPACK getrf (partial pivoting), `\\` on the factorisation is getrs -
    scipy.linalg.lu_factor / lu_solve call the same two routines.  The solution a is the Chebyshev coefficient column
    of w; CItransform! / CIxtransform (:592, 595) turn it into values and d/dz values."""

    def __init__(self, ch, pxi_bar, tau):
        from scipy.linalg import lu_factor
        self.ch = ch
        self.H = helmholtz_matrix_f64(ch, pxi_bar, tau)
        self.lu = lu_factor(self.H)

    def solve(self, rhs_cols):
        """rhs_cols [ncol, nz] -> (w values [ncol, nz], dw/dz values [ncol, nz])."""
        from scipy.linalg import lu_solve
        a = lu_solve(self.lu, np.ascontiguousarray(rhs_cols.T))          # [nz, ncol]
        return (self.ch.T @ a).T, (self.ch.T @ (self.ch.Dc @ a)).T


def inverse_extended(H):
    """Inverse of a (badly conditioned, cond ~ N^4) Helmholtz matrix by Gauss-Jordan elimination with partial pivoting
    in extended precision (numpy longdouble = x87 80-bit here). The reference solves with a double-precision LU
    (src/semiimplicit.jl:586-589); at zDim = 128 that alone carries ~1e-8 relative rounding error, so the oracle
    evaluates the same operator more accurately instead of adding its own copy of that noise."""
    n = H.shape[0]
    A = np.concatenate([H.astype(np.longdouble), np.eye(n, dtype=np.longdouble)], axis=1)
    for c in range(n):
        p = c + int(np.argmax(np.abs(A[c:, c])))
        if p != c:
            A[[c, p]] = A[[p, c]]
        A[c] = A[c] / A[c, c]
        f = A[:, c].copy()
        f[c] = 0.0
        A -= np.outer(f, A[c])
    return A[:, n:]


def semi_matrices(ch, pxi_bar, tau):
    """g -> w values (T H^-1) and g -> dw/dz values (T Dc H^-1) as double matrices."""
    Hinv = inverse_extended(helmholtz_matrix(ch, pxi_bar, tau, extended=True))
    return np.asarray(ch._x["T"] @ Hinv, dtype=np.float64), np.asarray(ch._x["TD"] @ Hinv, dtype=np.float64)


# ----------------------------------------------------------------------------- moist thermodynamics (Euler_test)
# Restated from src/thermodynamics.jl (constants :2-17, :31-32): only what Euler_test (src/testModels.jl:100-215) calls.
TH = dict(Rd=287.04, Rv=461.50, Cvd=716.96, Cvv=1410.0, Cl=4186.0, gravity=9.81, L_v0=2.501e6, T_0=273.16, p_0=1000.0, q0=1.0e-7)
TH["Cpv"] = TH["Cvv"] + TH["Rv"]
TH["rho_d0"] = 100.0 * TH["p_0"] / (TH["T_0"] * TH["Rd"])
TH["rho_v0"] = 100.0 * (6.112 * np.exp(17.67 * (TH["T_0"] - 273.15) / ((TH["T_0"] - 273.15) + 243.5))) / (TH["T_0"] * TH["Rv"])


def th_ahyp(mu):                                       # src/thermodynamics.jl:190-198
    q0 = TH["q0"]
    return np.where(mu < 0.0, 0.0, np.sqrt(mu * mu + q0 * q0) + mu - q0)


def th_dmudq(mu, q_v):                                 # :200-203
    return ((q_v + TH["q0"]) - mu) / (q_v + TH["q0"])


def th_dry_density(xi):                                # :205-208
    return TH["rho_d0"] * np.exp(xi)


def th_temperature(s, rho_d, q_v):                     # :67-80  (L_v(T_0) = L_v0, :41-44)
    Cf = TH["Cvd"] + (q_v * TH["Cvv"])
    with np.errstate(divide="ignore", invalid="ignore"):
        qf = np.where(q_v != 0.0, (rho_d * q_v / TH["rho_v0"]) ** ((q_v * TH["Rv"]) / Cf), 1.0)
    rf = (rho_d / TH["rho_d0"]) ** (TH["Rd"] / Cf)
    Tf = np.exp((s - (q_v * TH["L_v0"] / TH["T_0"])) / Cf)
    return TH["T_0"] * Tf * rf * qf


def th_P_s(Tk, rho_d, q_v):                            # :215-219
    return Tk * ((rho_d * TH["Rd"]) + (q_v * rho_d * TH["Rv"])) / (TH["Cvd"] + (q_v * TH["Cvv"]))


def th_P_xi(Tk, rho_d, q_v):                           # :221-224
    return (TH["Rd"] + (q_v * rho_d * TH["Rv"])) * ((rho_d * Tk) + th_P_s(Tk, rho_d, q_v))


def th_P_qv(Tk, rho_d, q_v):                           # :232-242
    with np.errstate(divide="ignore", invalid="ignore"):
        qf = TH["Rv"] * (1 + np.log(q_v * rho_d / TH["rho_v0"])) - (TH["Cvv"] * np.log(Tk / TH["T_0"])) - TH["L_v0"] / TH["T_0"]
    return np.where(q_v != 0.0, (rho_d * TH["Rv"] * Tk) + qf * th_P_s(Tk, rho_d, q_v), 0.0)


def th_pressure_gradient(Tk, rho_d, q_v, s_x, xi_x, qv_x):     # :250-258
    return (th_P_s(Tk, rho_d, q_v) * s_x) + (th_P_xi(Tk, rho_d, q_v) * xi_x) + (th_P_qv(Tk, rho_d, q_v) * qv_x)


def tendency(grid, eq, par, phys, pts, col_ops=None):
    """Pointwise tendencies of the in-scope equation sets. phys [N,V,D]; returns (expdot [N,V], impdot [N,V] or None, phys)
    (phys is returned because the shallow-water sets overwrite the diagnostic w in slot 1)."""
    N = phys.shape[0]
    E = np.zeros((N, grid.V))
    I = None
    P = lambda v, s: phys[:, v - 1, grid.slots.index(s)]
    if eq == "LinearAdvection1D":                       # src/testModels.jl:1-20
        E[:, 0] = -(par["c_0"] * P(1, "r")) + (par["K"] * P(1, "rr"))
    elif eq == "LinearAdvectionRZ":                     # src/testModels.jl:22-45
        r = pts[:, 0]
        E[:, 0] = (-P(2, "u") * P(1, "r")) + (-P(4, "u") * P(1, "z")) + \
                  (par["K"] * ((P(1, "r") / r) + P(1, "rr") + P(1, "zz")))
    elif eq in ("LinearAdvectionRL", "LinearAdvectionRLZ"):   # src/testModels.jl:47-98
        r = pts[:, 0]
        E[:, 0] = (-P(2, "u") * P(1, "r")) - (P(3, "u") * (P(1, "l") / r))
        if par["K"] > 0.0 or eq == "LinearAdvectionRLZ":
            E[:, 0] += par["K"] * ((P(1, "r") / r) + P(1, "rr") + (P(1, "ll") / (r * r)))
    elif eq in ("Oneway_ShallowWater_Slab", "Twoway_ShallowWater_Slab"):   # src/shallowWaterModels.jl:1-233
        r = pts[:, 0]
        g, K, Cd, Hfree, Hb, f = (par[k] for k in ("g", "K", "Cd", "Hfree", "Hb", "f"))
        h, hr, hl = P(1, "u"), P(1, "r"), P(1, "l")
        ug, ugr, ugl = P(2, "u"), P(2, "r"), P(2, "l")
        vg, vgr, vgl = P(3, "u"), P(3, "r"), P(3, "l")
        ub, ubr, ubrr, ubl, ubll = P(4, "u"), P(4, "r"), P(4, "rr"), P(4, "l"), P(4, "ll")
        vb, vbr, vbrr, vbl, vbll = P(5, "u"), P(5, "r"), P(5, "rr"), P(5, "l"), P(5, "ll")
        U = 0.78 * np.sqrt((ub * ub) + (vb * vb))
        w = -Hb * ((ub / r) + ubr + (vbl / r))
        phys[:, 5, 0] = w
        w_ = 0.5 * np.abs(w) - w
        E[:, 0] = ((-vg * hl / r) + (-ug * hr)) + (-(Hfree + h) * ((ug / r) + ugr + (vgl / r)))
        if eq.startswith("Twoway"):
            E[:, 0] += -(Hfree + h) * w * par["S1"]
        E[:, 1] = ((-vg * ugl / r) + (-ug * ugr)) + (-g * hr) + (vg * (f + (vg / r)))
        E[:, 2] = ((-vg * vgl / r) + (-ug * vgr)) + (-g * (hl / r)) + (-ug * (f + (vg / r)))
        E[:, 3] = ((-vb * ubl / r) + (-ub * ubr)) + (-g * hr) + (vb * (f + (vb / r))) + (-(Cd * U * ub / Hb)) \
            + (w_ * (ug - ub) / Hb) \
            + (K * ((ubr / r) + ubrr - (ub / (r * r)) + (ubll / (r * r)) - (2.0 * vbl / (r * r))))
        E[:, 4] = ((-vb * vbl / r) + (-ub * vbr)) + (-g * (hl / r)) + (-ub * (f + (vb / r))) + (-(Cd * U * vb / Hb)) \
            + (w_ * (vg - vb) / Hb) \
            + (K * ((vbr / r) + vbrr - (vb / (r * r)) + (vbll / (r * r)) + (2.0 * ubl / (r * r))))
    elif eq == "Oneway_ShallowWater_HeightResolvedBL":   # src/shallowWaterModels.jl:346-511
        nz = grid.zDim
        ch = grid.cheb("h")                              # "h" carries no vertical BC (ref :422-425)
        Mint = ch.Vint
        Mdz = ch.Vdz
        colv = lambda a: a.reshape(-1, nz)
        r, lam, z = colv(pts[:, 0]), colv(pts[:, 1]), colv(pts[:, 2])
        g, Kh, Cd0, Hfree, f, Um, Vm = (par[k] for k in ("g", "Kh", "Cd", "Hfree", "f", "Um", "Vm"))
        Q = lambda v, s: colv(P(v, s))
        h, hr, hl = Q(1, "u"), Q(1, "r"), Q(1, "l")
        ug, ugr, ugl = Q(2, "u"), Q(2, "r"), Q(2, "l")
        vg, vgr, vgl = Q(3, "u"), Q(3, "r"), Q(3, "l")
        ub, ubr, ubrr, ubl, ubll, ubz = (Q(4, s) for s in ("u", "r", "rr", "l", "ll", "z"))
        vb, vbr, vbrr, vbl, vbll, vbz = (Q(5, s) for s in ("u", "r", "rr", "l", "ll", "z"))
        S = np.sqrt((ubz * ubz) + (vbz * vbz))
        l = 1.0 / ((1.0 / (0.4 * z)) + (1.0 / 80.0))
        Kv = (l ** 2) * S
        wb = (-((ub / r) + ubr + (vbl / r))) @ Mint.T
        phys[:, 5, 0] = wb.reshape(-1)
        sfcu = (Um * np.cos(lam[:, 0])) + (Vm * np.sin(lam[:, 0]))
        sfcv = (Vm * np.cos(lam[:, 0])) - (Um * np.sin(lam[:, 0]))
        u10, v10 = ub[:, 1] + sfcu, vb[:, 1] + sfcv
        U10 = np.sqrt(u10 ** 2 + v10 ** 2)
        Cd = np.where(U10 < 5.2, 1.0e-3, np.where(U10 < 33.6, 4.4e-4 * U10 ** 0.5, Cd0))
        fu = Kv * ubz
        fu[:, 0] = Cd * U10 * u10
        fv = Kv * vbz
        fv[:, 0] = Cd * U10 * v10
        VDu, VDv = fu @ Mdz.T, fv @ Mdz.T
        E4 = ((-vb * ubl / r) + (-ub * ubr) + (-wb * ubz)) + (-g * hr) + (vb * (f + (vb / r))) + VDu \
            + (Kh * ((ubr / r) + ubrr - (ub / (r * r)) + (ubll / (r * r)) - (2.0 * vbl / (r * r))))
        E5 = ((-vb * vbl / r) + (-ub * vbr) + (-wb * vbz)) + (-g * (hl / r)) + (-ub * (f + (vb / r))) + VDv \
            + (Kh * ((vbr / r) + vbrr - (vb / (r * r)) + (vbll / (r * r)) + (2.0 * ubl / (r * r))))
        E[:, 0] = (((-vg * hl / r) + (-ug * hr)) + (-(Hfree + h) * ((ug / r) + ugr + (vgl / r)))).reshape(-1)
        E[:, 1] = (((-vg * ugl / r) + (-ug * ugr)) + (-g * hr) + (vg * (f + (vg / r)))).reshape(-1)
        E[:, 2] = (((-vg * vgl / r) + (-ug * vgr)) + (-g * (hl / r)) + (-ug * (f + (vg / r)))).reshape(-1)
        E[:, 3] = E4.reshape(-1)
        E[:, 4] = E5.reshape(-1)
    elif eq == "LinearAcousticRZ":
        # Synthetic RZ vehicle for semiimplicit_adjustment (config 3). NOT a reference equation set: it keeps
        # Euler_test's variable layout (s, xi, mu, u, w) and its two implicit terms (src/testModels.jl:188-189,
        # 204-205) but linearises the pressure-gradient force so no moist thermodynamics is needed.
        K, pxi = par["K"], par["Pxi_bar"]
        u, w = P(4, "u"), P(5, "u")
        adv = lambda v: (-u * P(v, "r")) + (-w * P(v, "z"))
        dif = lambda v: K * (P(v, "rr") + P(v, "zz"))
        I = np.zeros((N, grid.V))
        E[:, 0] = adv(1) + dif(1)
        E[:, 1] = adv(2) - P(4, "r") - P(5, "z")
        I[:, 1] = -P(5, "z")
        E[:, 2] = adv(3) + dif(3)
        E[:, 3] = adv(4) + (-(pxi * P(2, "r"))) + dif(4)
        E[:, 4] = adv(5) + (-(pxi * P(2, "z"))) + dif(5)
        I[:, 4] = -(pxi * P(2, "z"))
    elif eq == "Euler_test":                            # src/testModels.jl:100-215
        # par["ref_state"] = dict(sbar, xibar, mubar: [zDim, 3] value / d/dz / d2/dz2 per level; ReferenceState,
        # src/reference_state.jl:4-10); Pxi_bar in par.  The reference indexes the [zDim] profiles with the column's points
        # (z fastest), i.e. level k of every column sees row k.
        K, pxi, rs = par["K"], par["Pxi_bar"], par["ref_state"]
        nz = grid.zDim
        lev = np.arange(N) % nz
        sbar_z, xibar, xibar_z = rs["sbar"][lev, 1], rs["xibar"][lev, 0], rs["xibar"][lev, 1]
        sbar, mubar, mubar_z = rs["sbar"][lev, 0], rs["mubar"][lev, 0], rs["mubar"][lev, 1]
        u, w = P(4, "u"), P(5, "u")
        q_v = th_ahyp(P(3, "u") + mubar)
        rho_d = th_dry_density(P(2, "u") + xibar)
        Tk = th_temperature(P(1, "u") + sbar, rho_d, q_v)
        rho_t = rho_d * (1.0 + q_v)
        dm = th_dmudq(P(3, "u") + mubar, q_v)
        qvp_x, qvp_z = P(3, "r") / dm, P(3, "z") / dm
        rhobar = th_dry_density(xibar) * (1.0 + th_ahyp(mubar))
        rho_p = rho_t - rhobar
        dif = lambda v: K * (P(v, "rr") + P(v, "zz"))
        I = np.zeros((N, grid.V))
        E[:, 0] = ((-u * P(1, "r")) + (-w * (P(1, "z") + sbar_z))) + dif(1)
        E[:, 1] = ((-u * P(2, "r")) + (-w * (P(2, "z") + xibar_z))) - P(4, "r") - P(5, "z")
        I[:, 1] = -P(5, "z")
        E[:, 2] = ((-u * P(3, "r")) + (-w * (P(3, "z") + mubar_z))) + dif(3)
        E[:, 3] = ((-u * P(4, "r")) + (-w * P(4, "z"))) + \
                  (-(th_pressure_gradient(Tk, rho_d, q_v, P(1, "r"), P(2, "r"), qvp_x) / rho_t)) + dif(4)
        E[:, 4] = ((-u * P(5, "r")) + (-w * P(5, "z"))) + \
                  (-(TH["gravity"] * rho_p / rho_t) - (th_pressure_gradient(Tk, rho_d, q_v, P(1, "z"), P(2, "z"), qvp_z) / rho_t)) + dif(5)
        I[:, 4] = -(pxi * P(2, "z"))
    else:
        raise ValueError("equation set not in scope: " + eq)
    return E, I, phys


class Model:
    """One patch split into radial tiles, stepped with the reference's per-step protocol
    (src/semiimplicit.jl:258-332). Pure numpy; small cases only."""

    def __init__(self, grid, equation_set, ts, params, tiles=None, semiimplicit=False, pxi_bar=0.0, helmholtz="extended"):
        """helmholtz = "extended": the Helmholtz operator inverted once in extended precision ("truth" arbiter: what the
        column solve would return in exact arithmetic, to ~1e-15);  "lu": the reference's own arithmetic, a Float64
        matrix factorised by LAPACK getrf and solved per column by getrs (HelmholtzLU)."""
        self.g, self.eq, self.ts, self.par = grid, equation_set, float(ts), dict(params)
        self.tiles = tiles or [(0, grid.nc)]
        self.semi, self.pxi = semiimplicit, pxi_bar
        self.helmholtz = helmholtz
        self._lu = {}
        self.hist = [dict(e1=None, e2=None, i1=None, i2=None) for _ in self.tiles]
        self.A = None
        self.t = 0

    def set_initial(self, values_patch):
        """values_patch [N_patch, V]: initialize_model + first splineTransform! (:134-136, :233-237)."""
        shared = np.zeros((self.g.S_patch(), self.g.V))
        self._sum_tiles(shared, lambda c0, n: self._tile_slice(values_patch, c0, n))
        self.A = self.g.spline_transform(shared)

    def _tile_slice(self, arr, c0, n):
        g = self.g
        p0 = g.ringstart[MUBAR * c0] * g.zDim
        return arr[p0:p0 + g.tile_npoints(c0, n)]

    def _sum_tiles(self, shared, fn):
        for i, (c0, n) in enumerate(self.tiles):
            b = self.g.forward(fn(c0, n), c0, n)
            self.g.add_tile_to_shared(shared, b, c0, n, i == len(self.tiles) - 1)

    def physical(self):
        return np.concatenate([self.g.inverse(self.A, c0, n) for c0, n in self.tiles], axis=0)

    def step(self):
        self.t += 1
        t, g = self.t, self.g
        shared = np.zeros((g.S_patch(), g.V))
        for i, (c0, n) in enumerate(self.tiles):
            phys = g.inverse(self.A, c0, n)                                  # tileTransform!
            pts = g.gridpoints(c0, n)
            pts = pts.reshape(len(pts), -1)
            E, I, phys = tendency(g, self.eq, self.par, phys, pts)
            hs = self.hist[i]
            if t == 1:
                hs["e1"] = np.zeros_like(E)
                hs["e2"] = np.zeros_like(E)
            unp1, hs["e1"], hs["e2"] = explicit_timestep(t, self.ts, phys[:, :, 0], E, hs["e1"], hs["e2"])
            if self.semi:
                unp1 = self._semiimplicit(i, t, unp1, I)
            b = g.forward(unp1, c0, n)                                       # calcTendency
            g.add_tile_to_shared(shared, b, c0, n, i == len(self.tiles) - 1)
        self.A = g.spline_transform(shared)                                  # splineTransform!

    def _semiimplicit(self, i, t, unp1, impdot):
        """semiimplicit_adjustment (src/semiimplicit.jl:521-597), all columns of tile i at once."""
        g, ts, hs = self.g, self.ts, self.hist[i]
        nz = g.zDim
        wi, xi = g.vars["w"] - 1, g.vars["xi"] - 1
        I_n = impdot
        if t == 1:
            hs["i1"] = np.zeros_like(I_n)
            hs["i2"] = np.zeros_like(I_n)
        I1, I2 = hs["i1"], hs["i2"]
        out = unp1.copy()
        star = {}
        for v in (wi, xi):
            x = unp1[:, v]
            if t == 1:
                tau = 0.5 * ts
                x = x - (ts * I_n[:, v]) + (ts * 0.5 * I_n[:, v])
            elif t == 2:
                tau = 1.25 * ts
                x = x - (0.5 * ts) * ((3.0 * I_n[:, v]) - I1[:, v]) - (ts * I_n[:, v]) + (ts * 0.75 * I1[:, v])
            else:
                tau = 1.25 * ts
                x = x - ((ts / 12.0) * ((23.0 * I_n[:, v]) - (16.0 * I1[:, v]) + (5.0 * I2[:, v]))) \
                    - (ts * I_n[:, v]) + (ts * 0.75 * I1[:, v])
            star[v] = x.reshape(-1, nz)
        hs["i2"], hs["i1"] = I1.copy(), I_n.copy()
        chx, chw = g.cheb(g.names[xi]), g.cheb(g.names[wi])
        xi_star = star[xi] @ chx.Vrec.T
        xi_star_z = tau * self.pxi * (star[xi] @ chx.Vdz.T)
        gg = xi_star_z - star[wi]
        rhs = np.zeros_like(gg)
        rhs[:, 2:] = gg[:, 1:nz - 1]
        if self.helmholtz == "lu":
            if tau not in self._lu:
                self._lu[tau] = HelmholtzLU(chw, self.pxi, tau)
            w_val, w_z = self._lu[tau].solve(rhs)
        else:
            W, X = semi_matrices(chw, self.pxi, tau)
            w_val, w_z = rhs @ W.T, rhs @ X.T
        out[:, wi] = w_val.reshape(-1)
        out[:, xi] = (xi_star - tau * w_z).reshape(-1)
        return out
