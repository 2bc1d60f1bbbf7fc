"""Interface-only ("SPIKE"-type) form of the B -> A patch solve for radial tiles - a numerical PROTOTYPE on the host
(numpy), not yet a device path.  SURVEY.md 8(e)(i); the reference solves the whole patch redundantly on every rank after
gathering all of B (src/semiimplicit.jl:272-285), this package's default transposes the system with two all-to-alls
(DESIGN.md 5).  Here every tile solves ITS OWN rows and only the few unknowns that couple tiles are exchanged:

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
