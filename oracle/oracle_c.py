"""
TEST INFRASTRUCTURE — NOT PRODUCT CODE.

ctypes driver for oracle/liboracle.so (scythe_oracle.c + scythe_oracle_ops.c).  The C side builds its OWN operators
(basis tables, boundary-condition projection + Cholesky factors, Chebyshev matrices, Helmholtz operator:
scythe_oracle_ops.c, OPS = "c") and applies them with loops, so moderate sizes finish in seconds and the same code is the
"port" CPU baseline of bench.py; with OPS = "numpy" the operators are taken from the dense definitions in oracle_np.py
instead (the cross-check of the two constructions is tests/test_oracle_consistency.py).
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.
"""
import ctypes as C
import os
import subprocess
import numpy as np
from . import oracle_np as O

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None
OPS = os.environ.get("ORACLE_OPS", "c")       # "c": scythe_oracle_ops.c builds the operators; "numpy": oracle_np.py's are handed in
BC_CODE = {"R0": 0, "R1T0": 1, "R1T1": 2, "R1T2": 3, "R2T10": 4, "R2T20": 5, "R3": 6, "PERIODIC": 7}

EQ_IDS = {"LinearAdvection1D": 0, "LinearAdvectionRZ": 1, "LinearAdvectionRL": 2, "LinearAdvectionRLZ": 3,
          "Oneway_ShallowWater_Slab": 4, "Twoway_ShallowWater_Slab": 5,
          "Oneway_ShallowWater_HeightResolvedBL": 6, "LinearAcousticRZ": 7}
PAR_ORDER = ["g", "K", "Cd", "Hfree", "Hb", "f", "S1", "c_0", "Kh", "Um", "Vm", "Pxi_bar"]


def build(force=False):
    so = os.path.join(_HERE, "liboracle.so")
    src = os.path.join(_HERE, "scythe_oracle.c")
    src2 = os.path.join(_HERE, "scythe_oracle_ops.c")
    if force or not os.path.exists(so) or os.path.getmtime(so) < max(os.path.getmtime(src), os.path.getmtime(src2)):
        subprocess.check_call(["make", "-C", _HERE, "-s", "-B"])
    return so


def usable_cpus():
    """CPUs this process may actually use: the scheduler affinity, cut down to the cgroup's CPU quota.  The GPU boxes show all
    256 host CPUs to a container that owns 16 of them through a quota; OpenMP's default of one thread per VISIBLE CPU made a
    step of the C port 8 x slower there (4.8 s against 0.6 s with 16 threads, 40-cell RLZ patch)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, -(-int(quota) // int(period))))
    except (OSError, ValueError):
        pass
    return n


def lib():
    global _LIB
    if _LIB is None:
        so = os.path.join(_HERE, "liboracle.so")
        if not os.path.exists(so):
            build()
        _LIB = C.CDLL(so)
        _LIB.orc_num_threads.restype = C.c_int
        if "OMP_NUM_THREADS" not in os.environ:
            _LIB.orc_set_num_threads(usable_cpus())
        _LIB.orc_ops_phi.argtypes = [C.c_double, C.c_double, C.c_int, C.c_int, C.c_int, P_D, P_D]
        _LIB.orc_ops_wq.argtypes = [C.c_double, P_D]
        _LIB.orc_ops_spline_class.argtypes = [C.c_double, C.c_double, C.c_int, C.c_double, C.c_int, C.c_int, P_I, P_I, P_I, P_I, P_D, P_D, P_D, P_D]
        _LIB.orc_ops_cheb.argtypes = [C.c_double, C.c_double, C.c_int, C.c_int, C.c_int, C.c_int] + [P_D] * 9
        _LIB.orc_ops_helmholtz.argtypes = [C.c_double, C.c_double, C.c_int, C.c_double, C.c_double, P_D, P_D]
    return _LIB


P_I = C.POINTER(C.c_int)
P_D = C.POINTER(C.c_double)
P_L = C.POINTER(C.c_long)


class OrcGrid(C.Structure):
    _fields_ = [("has_l", C.c_int), ("has_z", C.c_int), ("V", C.c_int), ("D", C.c_int), ("b_rDim", C.c_int),
                ("cell0", C.c_int), ("ncells", C.c_int), ("nrings", C.c_int), ("zDim", C.c_int), ("b_zDim", C.c_int),
                ("K2", C.c_int), ("K2t", C.c_int), ("N", C.c_long),
                ("L", P_I), ("kmax", P_I), ("off", P_D), ("pstart", P_L), ("phi", P_D), ("m0", P_I), ("wq", P_D),
                ("Mz", P_D), ("CBz", P_D), ("slot", C.c_int * 7),
                ("nclass", C.c_int), ("cls", P_I), ("nfree", P_I), ("periodic", P_I), ("rl", P_I), ("rr", P_I),
                ("gl", P_D), ("gr", P_D), ("Lband", P_D), ("Larrow", P_D)]


class OrcStep(C.Structure):
    _fields_ = [("eq", C.c_int), ("t", C.c_int), ("semiimplicit", C.c_int), ("ts", C.c_double), ("par", P_D),
                ("r", P_D), ("lam", P_D), ("z", P_D), ("Mint", P_D), ("Mdz", P_D), ("Mrec", P_D),
                ("Wmat", P_D), ("Xmat", P_D), ("w_index", C.c_int), ("xi_index", C.c_int), ("tau", C.c_double)]


def _pd(a):
    return a.ctypes.data_as(P_D)


def _pi(a):
    return a.ctypes.data_as(P_I)


def c_cheb(grid, v):
    """Chebyshev column operators of variable v built by scythe_oracle_ops.c: dict(z, M [3][N][Zb], CB [Zb][N], Vint, Vdz, Vrec)."""
    g = grid
    N, Zb = g.zDim, g.b_zDim
    key = ("c_cheb", g.BCB[v], g.BCT[v])
    if key not in g._cheb:
        o = dict(z=np.zeros(N), M=np.zeros((3, N, Zb)), CB=np.zeros((Zb, N)), Vint=np.zeros((N, N)), Vdz=np.zeros((N, N)),
                 Vrec=np.zeros((N, N)), T=np.zeros((N, N)), D1=np.zeros((N, N)), D2=np.zeros((N, N)))
        rc = lib().orc_ops_cheb(g.zmin, g.zmax, N, Zb, BC_CODE[g.BCB[v]], BC_CODE[g.BCT[v]],
                                *[_pd(o[k]) for k in ("z", "M", "CB", "Vint", "Vdz", "Vrec", "T", "D1", "D2")])
        if rc:
            raise ValueError("orc_ops_cheb failed (unsupported vertical BC?)")
        g._cheb[key] = o
    return g._cheb[key]


def c_spline_class(grid, bcl, bcr):
    g = grid
    nb = g.b_rDim
    nfree, per, rl, rr = (C.c_int() for _ in range(4))
    gl, gr, Lband, Larrow = np.zeros((3, 2)), np.zeros((3, 2)), np.zeros((nb, 4)), np.zeros((3, nb))
    rc = lib().orc_ops_spline_class(g.xmin, g.xmax, g.nc, g.l_q, BC_CODE[bcl], BC_CODE[bcr], C.byref(nfree), C.byref(per),
                                    C.byref(rl), C.byref(rr), _pd(gl), _pd(gr), _pd(Lband), _pd(Larrow))
    if rc:
        raise ValueError("orc_ops_spline_class failed: %d" % rc)
    return nfree.value, per.value, rl.value, rr.value, gl, gr, Lband, Larrow


class TileOracle:
    """C-oracle view of one radial tile (cell0, ncells) of a numpy-oracle Grid."""

    def __init__(self, grid: O.Grid, cell0=0, ncells=None):
        g = grid
        self.g = g
        ncells = g.nc if ncells is None else ncells
        self.cell0, self.ncells = cell0, ncells
        rr = list(g.tile_rings(cell0, ncells))
        self.N = g.tile_npoints(cell0, ncells)
        k = self._keep = {}
        k["L"] = np.ascontiguousarray(g.L[rr], dtype=np.int32)
        k["kmax"] = np.ascontiguousarray(g.kmax[rr], dtype=np.int32)
        k["off"] = np.ascontiguousarray(g.off[rr], dtype=np.float64)
        k["pstart"] = np.ascontiguousarray(g.ringstart[rr] - g.ringstart[rr[0]], dtype=np.int64)
        phi = np.zeros((3, len(rr), 4))
        m0 = np.zeros(len(rr), dtype=np.int32)
        if OPS == "c":
            buf = np.zeros(4)
            for i in range(len(rr)):
                c = cell0 + i // 3
                m0[i] = c
                for d in range(3):
                    lib().orc_ops_phi(g.xmin, g.DX, c, i % 3, d, _pd(buf), None)
                    phi[d, i, :] = buf
            wq3 = np.zeros(3)
            lib().orc_ops_wq(g.DX, _pd(wq3))
            k["wq"] = np.tile(wq3, ncells)
        else:
            r = O.mish_points(g.xmin, g.DX, cell0, ncells)
            spl = g.spline("R0", "R0")
            for i in range(len(rr)):
                c = cell0 + i // 3
                m0[i] = c
                for d in range(3):
                    phi[d, i, :] = spl.basis(r[i:i + 1], d)[0, c:c + 4]
            k["wq"] = np.tile(g.DX * O.QUAD_W, ncells)
        k["phi"], k["m0"] = phi, m0
        if g.has_z:
            Mz = np.zeros((g.V, 3, g.zDim, g.b_zDim))
            for vi, v in enumerate(g.names):
                for d in range(3):
                    Mz[vi, d] = c_cheb(g, v)["M"][d] if OPS == "c" else g.cheb(v).M[d]
            k["Mz"] = Mz
            k["CBz"] = np.ascontiguousarray(c_cheb(g, g.names[0])["CB"] if OPS == "c" else g.cheb(g.names[0]).CBm)
        else:
            k["Mz"] = np.ones((g.V, 3, 1, 1))
            k["CBz"] = np.ones((1, 1))
        # spline solve classes
        keys, cls = [], np.zeros((g.V, 2), dtype=np.int32)
        for vi, v in enumerate(g.names):
            for q, bcl in enumerate((g.BCL_k0[v], g.BCL[v])):
                key = (bcl, g.BCR[v])
                if key not in keys:
                    keys.append(key)
                cls[vi, q] = keys.index(key)
        nc_ = len(keys)
        nb = g.b_rDim
        nfree = np.zeros(nc_, dtype=np.int32)
        per = np.zeros(nc_, dtype=np.int32)
        rl = np.zeros(nc_, dtype=np.int32)
        rr_ = np.zeros(nc_, dtype=np.int32)
        gl = np.zeros((nc_, 3, 2))
        gr = np.zeros((nc_, 3, 2))
        Lband = np.zeros((nc_, nb, 4))
        Larrow = np.zeros((nc_, 3, nb))
        for ci, (bcl, bcr) in enumerate(keys):
            if OPS == "c":
                nfree[ci], per[ci], rl[ci], rr_[ci], gl[ci], gr[ci], Lband[ci], Larrow[ci] = c_spline_class(g, bcl, bcr)
                continue
            s = g.spline(bcl, bcr)
            n = s.G.shape[0]
            nfree[ci] = n
            Lc = s.cho
            if bcl == "PERIODIC":
                per[ci] = 1
                Larrow[ci, :, :n] = Lc[n - 3:, :]
            else:
                rl[ci], rr_[ci] = O.BC_RANK[bcl], O.BC_RANK[bcr]
                for i in range(rl[ci]):
                    gl[ci, i, 0], gl[ci, i, 1] = s.G[0, i], s.G[1, i]
                for i in range(rr_[ci]):
                    gr[ci, i, 0], gr[ci, i, 1] = s.G[n - 1, nb - 1 - i], s.G[n - 2, nb - 1 - i]
            for i in range(n):
                for q in range(4):
                    if i - q >= 0:
                        Lband[ci, i, 3 - q] = Lc[i, i - q]
        k.update(cls=cls, nfree=nfree, per=per, rl=rl, rr=rr_, gl=gl, gr=gr, Lband=Lband, Larrow=Larrow)
        og = OrcGrid()
        og.has_l, og.has_z, og.V, og.D, og.b_rDim = int(g.has_l), int(g.has_z), g.V, g.D, g.b_rDim
        og.cell0, og.ncells, og.nrings, og.zDim, og.b_zDim = cell0, ncells, len(rr), g.zDim, g.b_zDim
        og.K2, og.K2t, og.N = g.K2, g.tile_K2(cell0, ncells), self.N
        og.L, og.kmax, og.off = _pi(k["L"]), _pi(k["kmax"]), _pd(k["off"])
        og.pstart = k["pstart"].ctypes.data_as(P_L)
        og.phi, og.m0, og.wq, og.Mz, og.CBz = _pd(phi), _pi(m0), _pd(k["wq"]), _pd(k["Mz"]), _pd(k["CBz"])
        names = ["u", "r", "rr", "l", "ll", "z", "zz"]
        for i, nme in enumerate(names):
            og.slot[i] = g.slots.index(nme) if nme in g.slots else -1
        og.nclass, og.cls, og.nfree, og.periodic = nc_, _pi(cls), _pi(nfree), _pi(per)
        og.rl, og.rr, og.gl, og.gr, og.Lband, og.Larrow = _pi(rl), _pi(rr_), _pd(gl), _pd(gr), _pd(Lband), _pd(Larrow)
        self.og = og
        self.S_tile = g.b_zDim * og.K2t * (ncells + 3)
        pts = g.gridpoints(cell0, ncells)
        self.pts = pts.reshape(len(pts), -1)

    # reference layouts: values [N, V] / spectral [S, V] / physical [N, V, D]  (Fortran order == Julia)
    def forward(self, values):
        vals = np.asfortranarray(values, dtype=np.float64)
        out = np.zeros((self.S_tile, self.g.V), order="F")
        lib().orc_forward(C.byref(self.og), _pd(vals), _pd(out))
        return out

    def add_to_shared(self, btile, shared):
        assert shared.flags.f_contiguous and btile.flags.f_contiguous
        lib().orc_add_tile(C.byref(self.og), _pd(btile), _pd(shared))

    def spline_solve(self, shared):
        sh = np.asfortranarray(shared, dtype=np.float64)
        A = np.zeros_like(sh, order="F")
        lib().orc_spline_solve(C.byref(self.og), _pd(sh), _pd(A))
        return A

    def inverse(self, A, out=None):
        A = np.asfortranarray(A, dtype=np.float64)
        phys = out if out is not None else np.zeros((self.N, self.g.V, self.g.D), order="F")
        lib().orc_inverse(C.byref(self.og), _pd(A), _pd(phys))
        return phys


class ModelOracle:
    """Patch split into radial tiles, stepped with the reference's per-step protocol, C loops."""

    def __init__(self, grid, equation_set, ts, params, tiles=None, semiimplicit=False):
        self.g, self.eq, self.ts = grid, equation_set, float(ts)
        self.par = np.array([float(params.get(k, 0.0)) for k in PAR_ORDER])
        self.tiles = [TileOracle(grid, c0, n) for c0, n in (tiles or [(0, grid.nc)])]
        self.semi = bool(semiimplicit)
        g = grid
        self.state = []
        for tl in self.tiles:
            z = lambda: np.zeros((tl.N, g.V), order="F")
            self.state.append(dict(phys=np.zeros((tl.N, g.V, g.D), order="F"), E=z(), e1=z(), e2=z(),
                                   I=z(), i1=z(), i2=z(), np1=z()))
        self.A = None
        self.t = 0
        self._mats = {}
        nz = g.zDim
        self.Mint = self.Mdz = self.Mrec = np.zeros((1, 1))
        if g.has_z:
            hv = "h" if "h" in g.vars else g.names[0]
            if OPS == "c":
                ch = c_cheb(g, hv)
                self.Mint, self.Mdz = np.ascontiguousarray(ch["Vint"]), np.ascontiguousarray(ch["Vdz"])
            else:
                ch = g.cheb(hv)
                self.Mint, self.Mdz = np.ascontiguousarray(ch.Vint), np.ascontiguousarray(ch.Vdz)
        if self.semi:
            if OPS == "c":
                chx = c_cheb(g, "xi")
                self.Mdz, self.Mrec = np.ascontiguousarray(chx["Vdz"]), np.ascontiguousarray(chx["Vrec"])
            else:
                chx = g.cheb("xi")
                self.Mdz, self.Mrec = np.ascontiguousarray(chx.Vdz), np.ascontiguousarray(chx.Vrec)

    def _semi_mats(self, tau):
        if tau not in self._mats:
            g = self.g
            pxi = self.par[PAR_ORDER.index("Pxi_bar")]
            if OPS == "c":
                W, X = np.zeros((g.zDim, g.zDim)), np.zeros((g.zDim, g.zDim))
                if lib().orc_ops_helmholtz(g.zmin, g.zmax, g.zDim, pxi, tau, _pd(W), _pd(X)):
                    raise ValueError("orc_ops_helmholtz: singular matrix")
            else:
                W, X = O.semi_matrices(g.cheb("w"), pxi, tau)
            self._mats[tau] = (np.ascontiguousarray(W), np.ascontiguousarray(X))
        return self._mats[tau]

    def set_initial(self, values_patch):
        g = self.g
        shared = np.zeros((g.S_patch(), g.V), order="F")
        p0 = 0
        for tl in self.tiles:
            tl.add_to_shared(tl.forward(values_patch[p0:p0 + tl.N]), shared)
            p0 += tl.N
        self.A = self.tiles[0].spline_solve(shared)

    def physical(self):
        return np.concatenate([tl.inverse(self.A) for tl in self.tiles], axis=0)

    def step(self):
        self.t += 1
        g = self.g
        shared = np.zeros((g.S_patch(), g.V), order="F")
        for tl, s in zip(self.tiles, self.state):
            tl.inverse(self.A, out=s["phys"])
            st = OrcStep()
            st.eq, st.t, st.semiimplicit, st.ts = EQ_IDS[self.eq], self.t, int(self.semi), self.ts
            st.par = _pd(self.par)
            cols = [np.ascontiguousarray(tl.pts[:, i]) for i in range(tl.pts.shape[1])]
            st.r = _pd(cols[0])
            if g.has_l:
                st.lam = _pd(cols[1])
            if g.has_z:
                st.z = _pd(cols[-1])
            st.Mint, st.Mdz, st.Mrec = _pd(self.Mint), _pd(self.Mdz), _pd(self.Mrec)
            if self.semi:
                tau = (0.5 if self.t == 1 else 1.25) * self.ts
                W, X = self._semi_mats(tau)
                st.Wmat, st.Xmat, st.tau = _pd(W), _pd(X), tau
                st.w_index, st.xi_index = g.vars["w"] - 1, g.vars["xi"] - 1
            lib().orc_tendency_step(C.byref(tl.og), C.byref(st), _pd(s["phys"]), _pd(s["E"]), _pd(s["e1"]),
                                    _pd(s["e2"]), _pd(s["I"]), _pd(s["i1"]), _pd(s["i2"]), _pd(s["np1"]))
            tl.add_to_shared(tl.forward(s["np1"]), shared)
        self.A = self.tiles[0].spline_solve(shared)
