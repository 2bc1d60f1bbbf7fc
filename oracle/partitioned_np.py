"""TEST INFRASTRUCTURE (oracle/): dense-numpy restatement of the interface-only ("SPIKE"-type) form of the B -> A patch
solve for radial tiles, the checker of the device path in scythe.jl_amd/csrc/sx_iface.hip.  Nothing in the product imports
this.  SURVEY.md 8(e)(i); the reference solves the whole patch redundantly on every rank after gathering all of B
(src/semiimplicit.jl:272-285).  Here every tile solves ITS OWN rows and only the few unknowns that couple tiles are exchanged:

    M = D + R        D = the diagonal blocks of the tiles,  R = what couples rows of one tile to unknowns of another
    a = D^-1 b - D^-1 R a                    R has non-zero columns only at the interface unknowns I
    y = D^-1 b           (local, a chain of n / N rows instead of n)
    S = D^-1 R[:, I]     ("spikes": they depend on the matrix only - one set per boundary-condition class, built at set-up)
    (1 + S[I, :]) a_I = y_I                  (|I| = 6 (N - 1) unknowns per column for the half-bandwidth-3 spline matrix)
    a = y - S a_I        (local: 6 multiply-adds per row and column)

Per step a tile contributes its 6 interface values of y per column and needs the a_I of its own two interfaces: with the
reduced systems split by column over the ranks that is two all-to-alls of 6 rows instead of two of n / N + 3 rows.
"""
import numpy as np


class PartitionedBandedSolve:
    def __init__(self, M, bounds):
        """M: [n, n] (banded, possibly with periodic corner blocks); bounds: partition starts, bounds[0] = 0, bounds[-1] = n."""
        M = np.asarray(M, dtype=np.float64)
        n = M.shape[0]
        assert bounds[0] == 0 and bounds[-1] == n and all(b1 > b0 for b0, b1 in zip(bounds[:-1], bounds[1:]))
        self.n, self.bounds = n, list(bounds)
        R = M.copy()
        self.blocks = []
        for s, e in zip(bounds[:-1], bounds[1:]):
            self.blocks.append(np.linalg.cholesky(M[s:e, s:e]))
            R[s:e, s:e] = 0.0
        self.I = np.flatnonzero(np.abs(R).sum(axis=0) > 0.0)               # interface unknowns
        self.S = self._local(R[:, self.I])                                   # spikes, [n, |I|]
        self.T = np.eye(len(self.I)) + self.S[self.I, :]                     # reduced system
        self.Tinv = np.linalg.inv(self.T)
        # what partition g sends (its own interface rows) and needs (the interface unknowns its spikes touch)
        self.sends = [int(((self.I >= s) & (self.I < e)).sum()) for s, e in zip(bounds[:-1], bounds[1:])]
        self.needs = [int((np.abs(self.S[s:e, :]).sum(axis=0) > 0.0).sum()) for s, e in zip(bounds[:-1], bounds[1:])]

    def _local(self, rhs):
        out = np.empty_like(rhs, dtype=np.float64)
        for (s, e), c in zip(zip(self.bounds[:-1], self.bounds[1:]), self.blocks):
            out[s:e] = np.linalg.solve(c.T, np.linalg.solve(c, rhs[s:e]))
        return out

    def solve(self, b):
        y = self._local(np.asarray(b, dtype=np.float64))
        a_I = self.Tinv @ y[self.I]
        return y - self.S @ a_I


class InterfaceSolve:
    """The staged form the device path runs (sx_iface.hip), restated with dense numpy linear algebra on the oracle's own
    spline matrix: tile rows in, tile rows out, 10 rows per tile through the reduced system.

        local(t, B_t)   -> y'_t, send_t [10, cols]     send = 6 edge values of y' + the <= 4 rows of the tile that fold onto
                                                        unknowns of another tile (3 halo rows, PERIODIC wrap rows)
        reduce(inp)     -> out [N, 10, cols]            6 right-hand-side corrections + <= 4 foreign coefficients per tile
        apply(t, y', r) -> A rows of tile t [ncells_t + 3, cols]

    sp: oracle_np.Spline1D (sp.PQ = Gamma P Gamma^T, sp.G = Gamma [free unknowns x patch rows]); tiles as (cell0, ncells)."""

    E, X = 6, 4

    def __init__(self, sp, cell0, ncells):
        self.sp, self.N = sp, len(cell0)
        self.cell0, self.ncells = list(cell0), list(ncells)
        M, G = sp.PQ, sp.G
        nf, nb = G.shape
        # primary row of unknown u: the row it is stored in (rl + u, or u + 1 for PERIODIC)
        periodic = sp.bcl == "PERIODIC"
        sh = 1 if periodic else (nb - nf) - _rank(sp.bcr)
        assert all(G[u, u + sh] == 1.0 for u in range(nf))
        self.sh = sh
        own = [n + (3 if t == self.N - 1 else 0) for t, n in enumerate(self.ncells)]
        self.u0 = [max(0, c - sh) for c in self.cell0]
        self.u1 = [min(nf, c + o - sh) for c, o in zip(self.cell0, own)]
        assert self.u0[0] == 0 and self.u1[-1] == nf and all(a == b for a, b in zip(self.u1[:-1], self.u0[1:]))
        assert all(b - a >= self.E for a, b in zip(self.u0, self.u1)), "a tile owns fewer than 6 free coefficients"
        self.tile_of = np.concatenate([np.full(b - a, t) for t, (a, b) in enumerate(zip(self.u0, self.u1))])
        self.I = np.concatenate([np.r_[a:a + 3, b - 3:b] for a, b in zip(self.u0, self.u1)])
        same = self.tile_of[:, None] == self.tile_of[None, :]
        self.D = np.where(same, M, 0.0)
        self.R = M - self.D
        assert not np.any(np.delete(self.R, self.I, axis=1)) and not np.any(np.delete(self.R, self.I, axis=0))
        # rows of tile t with support outside the tile's unknowns
        self.xrows = []
        for t in range(self.N):
            rows = range(self.cell0[t], self.cell0[t] + self.ncells[t] + 3)
            inside = lambda r: np.any(G[self.u0[t]:self.u1[t], r])
            outside = lambda r: np.any(G[:self.u0[t], r]) or np.any(G[self.u1[t]:, r])
            assert not any(inside(r) and outside(r) for r in rows)
            xr = [r for r in rows if outside(r)]
            assert len(xr) <= self.X
            self.xrows.append(xr)
        self.G = G

    def _blk(self, t):
        return slice(self.u0[t], self.u1[t])

    def local(self, t, Bt):
        """Bt: the tile's B rows [ncells_t + 3, cols]."""
        rows = np.arange(self.cell0[t], self.cell0[t] + self.ncells[t] + 3)
        bl = self.G[self._blk(t)][:, rows] @ Bt                       # rows that fold onto this tile's own unknowns
        y = np.linalg.solve(self.D[self._blk(t), self._blk(t)], bl)
        send = np.zeros((self.E + self.X, Bt.shape[1]))
        send[:3], send[3:6] = y[:3], y[-3:]
        for s, r in enumerate(self.xrows[t]):
            send[self.E + s] = Bt[r - self.cell0[t]]
        return y, send

    def reduce(self, inp):
        """inp [N, 10, cols] -> out [N, 10, cols]."""
        N, nf = self.N, self.G.shape[0]
        cols = inp.shape[2]
        yI = inp[:, :self.E].reshape(N * self.E, cols)
        Ef = np.zeros((nf, cols))                                     # E f: the foreign rows folded onto their unknowns
        for t in range(N):
            for s, r in enumerate(self.xrows[t]):
                Ef += self.G[:, [r]] * inp[t, self.E + s][None, :]
        I = self.I
        y_I = yI + np.linalg.solve(self.D, Ef)[I]
        T = np.eye(len(I)) + np.linalg.solve(self.D, self.R[:, I])[I]
        a_I = np.linalg.solve(T, y_I)
        c = Ef[I] - self.R[np.ix_(I, I)] @ a_I
        a_full = np.zeros((nf, cols))
        a_full[I] = a_I
        out = np.zeros_like(inp)
        out[:, :self.E] = c.reshape(N, self.E, cols)
        for t in range(N):
            for s, r in enumerate(self.xrows[t]):
                out[t, self.E + s] = self.G[:, r] @ a_full
        return out

    def apply(self, t, y, recv):
        nt = self.u1[t] - self.u0[t]
        rhs = np.zeros((nt, y.shape[1]))
        rhs[:3], rhs[-3:] = recv[:3], recv[3:6]
        a = y + np.linalg.solve(self.D[self._blk(t), self._blk(t)], rhs)
        rows = np.arange(self.cell0[t], self.cell0[t] + self.ncells[t] + 3)
        A = self.G[self._blk(t)][:, rows].T @ a                       # own unknowns and the rows that depend on them
        for s, r in enumerate(self.xrows[t]):
            A[r - self.cell0[t]] = recv[self.E + s]
        return A


def _rank(bc):
    return {"R0": 0, "R1T0": 1, "R1T1": 1, "R1T2": 1, "R2T10": 2, "R2T20": 2, "R3": 3, "PERIODIC": 0}[bc]
