// Host-side operator construction for libscythe_hip.so.
//
// Everything here restates arithmetic the reference obtains from the external Springsteel.jl package
// (Project.toml:20) - cubic B-spline basis, quadrature, P + eps_q Q assembly, boundary-condition projection,
// Cholesky factorisation, Chebyshev collocation / derivative / integral operators - following SURVEY.md 8(c).
// The quadrature-weight ratio 8:5:8 is the one the reference's notebook known-answer pins.
#include "sx_internal.hpp"
#include <algorithm>
#include <cmath>
#include <cstring>

namespace sx {

static const double SQRT35 = std::sqrt(3.0 / 5.0);
static const double GAUSS_OFF[MUBAR] = {-SQRT35 / 2.0, 0.0, SQRT35 / 2.0};
static const double QUAD_W[MUBAR] = {8.0 / 21.0, 5.0 / 21.0, 8.0 / 21.0};

// d-th derivative of the cardinal cubic B-spline with respect to its argument
static double bspl(double delta, int d) {
    double z = std::fabs(delta);
    if (z >= 2.0) return 0.0;
    double s = delta > 0 ? 1.0 : -1.0;
    double p = 2.0 - z, q = z < 1.0 ? 1.0 - z : 0.0;
    switch (d) {
        case 0: return p * p * p / 6.0 - 4.0 * q * q * q / 6.0;
        case 1: return -s * (p * p / 2.0 - 2.0 * q * q);
        case 2: return p - 4.0 * q;
        default: return s * (z < 1.0 ? 3.0 : -1.0);
    }
}

// phi[d][mu][j]: d-th x-derivative of basis function of node (cell - 1 + j) at mish point mu of that cell
void basis_tables(double DX, double phi[4][MUBAR][4]) {
    for (int d = 0; d < 4; d++) {
        double sc = 1.0 / std::pow(DX, d);
        for (int mu = 0; mu < MUBAR; mu++)
            for (int j = 0; j < 4; j++) phi[d][mu][j] = bspl(1.5 + GAUSS_OFF[mu] - j, d) * sc;
    }
}

void quad_weights(double DX, double w[MUBAR]) {
    for (int mu = 0; mu < MUBAR; mu++) w[mu] = DX * QUAD_W[mu];
}

int bc_rank(int bc) {
    switch (bc) {
        case SX_BC_R0: return 0;
        case SX_BC_R1T0: case SX_BC_R1T1: case SX_BC_R1T2: return 1;
        case SX_BC_R2T10: case SX_BC_R2T20: return 2;
        case SX_BC_R3: return 3;
        default: return -1;
    }
}

// rows of dependent boundary coefficients: a_dep[i] = g[i][0] * free_first + g[i][1] * free_second
static bool boundary_rows(int bc, double g[3][2]) {
    std::memset(g, 0, sizeof(double) * 6);
    switch (bc) {
        case SX_BC_R0: case SX_BC_R3: return true;
        case SX_BC_R1T0: g[0][0] = -4.0; g[0][1] = -1.0; return true;
        case SX_BC_R1T1: g[0][0] = 0.0; g[0][1] = 1.0; return true;
        case SX_BC_R1T2: g[0][0] = 2.0; g[0][1] = -1.0; return true;
        case SX_BC_R2T10: g[0][0] = 1.0; g[1][0] = -0.5; return true;
        case SX_BC_R2T20: g[0][0] = -1.0; g[1][0] = 0.0; return true;
        default: return false;
    }
}

bool build_spline_class(int nc, double DX, double l_q, int bcl, int bcr, SplineClass &out, std::string &err) {
    const int nb = nc + 3;
    out.bcl = bcl;
    out.bcr = bcr;
    out.periodic = (bcl == SX_BC_PERIODIC || bcr == SX_BC_PERIODIC);
    if (out.periodic && bcl != bcr) { err = "PERIODIC must be set on both sides"; return false; }
    if (!out.periodic) {
        out.rl = bc_rank(bcl);
        out.rr = bc_rank(bcr);
        if (out.rl < 0 || out.rr < 0 || !boundary_rows(bcl, out.gl) || !boundary_rows(bcr, out.gr)) {
            err = "unknown radial boundary condition code";
            return false;
        }
        out.nfree = nb - out.rl - out.rr;
    } else {
        out.rl = out.rr = 0;
        out.nfree = nc;
    }
    const int n = out.nfree;
    if (n < (out.periodic ? 7 : 4)) { err = "too few cells for the requested boundary conditions"; return false; }

    // P + eps_q Q, 7-diagonal: Pb[mi][mj - mi + 3]
    double phi[4][MUBAR][4], w[MUBAR];
    basis_tables(DX, phi);
    quad_weights(DX, w);
    const double eps_q = std::pow(l_q * DX / (2.0 * M_PI), 6);
    std::vector<double> Pb((size_t)nb * 7, 0.0);
    for (int c = 0; c < nc; c++)
        for (int mu = 0; mu < MUBAR; mu++)
            for (int j = 0; j < 4; j++)
                for (int k = 0; k < 4; k++)
                    Pb[(size_t)(c + j) * 7 + (k - j + 3)] +=
                        w[mu] * (phi[0][mu][j] * phi[0][mu][k] + eps_q * phi[3][mu][j] * phi[3][mu][k]);
    auto P = [&](int i, int j) -> double {
        int dj = j - i + 3;
        return (dj < 0 || dj > 6) ? 0.0 : Pb[(size_t)i * 7 + dj];
    };
    // Gamma as sparse rows: free j -> list of (full index, weight)
    std::vector<std::vector<std::pair<int, double>>> G(n);
    if (out.periodic) {
        for (int m = 0; m < nb; m++) G[(m - 1 + n) % n].push_back({m, 1.0});
    } else {
        for (int j = 0; j < n; j++) G[j].push_back({out.rl + j, 1.0});
        for (int i = 0; i < out.rl; i++) {
            if (out.gl[i][0] != 0.0) G[0].push_back({i, out.gl[i][0]});
            if (out.gl[i][1] != 0.0) G[1].push_back({i, out.gl[i][1]});
        }
        for (int i = 0; i < out.rr; i++) {
            if (out.gr[i][0] != 0.0) G[n - 1].push_back({nb - 1 - i, out.gr[i][0]});
            if (out.gr[i][1] != 0.0) G[n - 2].push_back({nb - 1 - i, out.gr[i][1]});
        }
    }
    // PQ = Gamma P Gamma^T (dense, symmetric)
    std::vector<double> A((size_t)n * n, 0.0);
    for (int a = 0; a < n; a++)
        for (int b = 0; b <= a; b++) {
            double s = 0.0;
            for (auto &ia : G[a])
                for (auto &jb : G[b]) s += ia.second * P(ia.first, jb.first) * jb.second;
            A[(size_t)a * n + b] = s;
            A[(size_t)b * n + a] = s;
        }
    out.Mdense = A;
    out.Gam = G;
    // dense Cholesky A = L L^T (lower)
    for (int j = 0; j < n; j++) {
        double d = A[(size_t)j * n + j];
        for (int k = 0; k < j; k++) d -= A[(size_t)j * n + k] * A[(size_t)j * n + k];
        if (!(d > 0.0)) { err = "spline matrix is not positive definite"; return false; }
        d = std::sqrt(d);
        A[(size_t)j * n + j] = d;
        for (int i = j + 1; i < n; i++) {
            // skip structural zeros cheaply: only band rows and (periodic) the last three rows are non-zero
            if (i - j > 3 && !(out.periodic && i >= n - 3)) { A[(size_t)i * n + j] = 0.0; continue; }
            double s = A[(size_t)i * n + j];
            for (int k = 0; k < j; k++) s -= A[(size_t)i * n + k] * A[(size_t)j * n + k];
            A[(size_t)i * n + j] = s / d;
        }
    }
    out.Lband.assign((size_t)nb * 4, 0.0);
    out.Larrow.assign((size_t)3 * nb, 0.0);
    for (int i = 0; i < n; i++)
        for (int q = 0; q < 4; q++)
            if (i - q >= 0) out.Lband[(size_t)i * 4 + (3 - q)] = A[(size_t)i * n + (i - q)];
    if (out.periodic)
        for (int r = 0; r < 3; r++)
            for (int k = 0; k <= n - 3 + r; k++) out.Larrow[(size_t)r * nb + k] = A[(size_t)(n - 3 + r) * n + k];
    return true;
}

// ---------------------------------------------------------------------------------------------- parallel cyclic reduction tables
typedef long double pxr;
struct B3 { pxr a[9]; };
static B3 b3_zero() { B3 z; for (int i = 0; i < 9; i++) z.a[i] = 0.0L; return z; }
static B3 b3_mul(const B3 &x, const B3 &y) {
    B3 r = b3_zero();
    for (int i = 0; i < 3; i++)
        for (int k = 0; k < 3; k++)
            for (int j = 0; j < 3; j++) r.a[i * 3 + j] += x.a[i * 3 + k] * y.a[k * 3 + j];
    return r;
}
static B3 b3_sub(const B3 &x, const B3 &y) { B3 r; for (int i = 0; i < 9; i++) r.a[i] = x.a[i] - y.a[i]; return r; }
static bool b3_inv(const B3 &m, B3 &out) {
    const pxr *a = m.a;
    const pxr c00 = a[4] * a[8] - a[5] * a[7], c01 = a[5] * a[6] - a[3] * a[8], c02 = a[3] * a[7] - a[4] * a[6];
    const pxr det = a[0] * c00 + a[1] * c01 + a[2] * c02;
    if (det == 0.0L) return false;
    const pxr id = 1.0L / det;
    out.a[0] = c00 * id; out.a[1] = (a[2] * a[7] - a[1] * a[8]) * id; out.a[2] = (a[1] * a[5] - a[2] * a[4]) * id;
    out.a[3] = c01 * id; out.a[4] = (a[0] * a[8] - a[2] * a[6]) * id; out.a[5] = (a[2] * a[3] - a[0] * a[5]) * id;
    out.a[6] = c02 * id; out.a[7] = (a[1] * a[6] - a[0] * a[7]) * id; out.a[8] = (a[0] * a[4] - a[1] * a[3]) * id;
    return true;
}
static pxr b3_norm(const B3 &m) { pxr s = 0.0L; for (int i = 0; i < 9; i++) s = std::max(s, fabsl(m.a[i])); return s; }

bool build_pcr_tables(const SplineClass &sc, int nb, PcrTables &t, std::string &err) {
    const int n = sc.nfree;
    t.n = n; t.periodic = sc.periodic;
    t.nblk = (n + 2) / 3;
    const int np = 3 * t.nblk;
    if ((int)sc.Mdense.size() != n * n) { err = "build_pcr_tables: the class has no dense matrix"; return false; }
    auto Mc = [&](int i, int j) -> pxr { return (pxr)sc.Mdense[(size_t)i * n + j]; };
    // band part (PERIODIC: the corner blocks leave through the rank-6 correction below); padding unknowns are identity rows
    auto Mb = [&](int i, int j) -> pxr {
        if (i >= n || j >= n) return i == j ? 1.0L : 0.0L;
        return std::abs(i - j) <= 3 ? Mc(i, j) : 0.0L;
    };
    std::vector<B3> Lb(t.nblk), Db(t.nblk), Ub(t.nblk);
    for (int i = 0; i < t.nblk; i++) {
        Lb[i] = Db[i] = Ub[i] = b3_zero();
        for (int p = 0; p < 3; p++)
            for (int q = 0; q < 3; q++) {
                Db[i].a[p * 3 + q] = Mb(3 * i + p, 3 * i + q);
                if (i > 0) Lb[i].a[p * 3 + q] = Mb(3 * i + p, 3 * (i - 1) + q);
                if (i + 1 < t.nblk) Ub[i].a[p * 3 + q] = Mb(3 * i + p, 3 * (i + 1) + q);
            }
    }
    t.coef.clear();
    t.levels = 0;
    for (int s = 1; s < 2 * t.nblk; s <<= 1) {
        pxr off = 0.0L, dia = 0.0L;
        for (int i = 0; i < t.nblk; i++) { off = std::max(off, std::max(b3_norm(Lb[i]), b3_norm(Ub[i]))); dia = std::max(dia, b3_norm(Db[i])); }
        // block diagonal already - exactly (s >= nblk) or far below anything a double-precision right-hand side can see
        if (off == 0.0L || off < dia * 1e-34L) break;
        std::vector<B3> Di(t.nblk), Ln(t.nblk), Dn(t.nblk), Un(t.nblk);
        for (int i = 0; i < t.nblk; i++)
            if (!b3_inv(Db[i], Di[i])) { err = "build_pcr_tables: singular diagonal block"; return false; }
        const size_t base = t.coef.size();
        t.coef.resize(base + (size_t)t.nblk * 18, 0.0);
        for (int i = 0; i < t.nblk; i++) {
            B3 al = b3_zero(), ga = b3_zero();
            Dn[i] = Db[i]; Ln[i] = b3_zero(); Un[i] = b3_zero();
            if (i - s >= 0) {
                al = b3_mul(Lb[i], Di[i - s]);
                Dn[i] = b3_sub(Dn[i], b3_mul(al, Ub[i - s]));
                Ln[i] = b3_sub(b3_zero(), b3_mul(al, Lb[i - s]));
            }
            if (i + s < t.nblk) {
                ga = b3_mul(Ub[i], Di[i + s]);
                Dn[i] = b3_sub(Dn[i], b3_mul(ga, Lb[i + s]));
                Un[i] = b3_sub(b3_zero(), b3_mul(ga, Ub[i + s]));
            }
            for (int q = 0; q < 9; q++) {
                t.coef[base + (size_t)i * 18 + q] = (double)al.a[q];
                t.coef[base + (size_t)i * 18 + 9 + q] = (double)ga.a[q];
            }
        }
        Lb.swap(Ln); Db.swap(Dn); Ub.swap(Un);
        t.levels++;
    }
    t.dinv.assign((size_t)t.nblk * 9, 0.0);
    for (int i = 0; i < t.nblk; i++) {
        B3 di;
        if (!b3_inv(Db[i], di)) { err = "build_pcr_tables: singular reduced block"; return false; }
        for (int q = 0; q < 9; q++) t.dinv[(size_t)i * 9 + q] = (double)di.a[q];
    }
    // Gamma b and Gamma^T x as gather tables
    t.gin_row.assign((size_t)np * 4, -1); t.gin_w.assign((size_t)np * 4, 0.0);
    t.gout_j.assign((size_t)nb * 2, -1); t.gout_w.assign((size_t)nb * 2, 0.0);
    std::vector<int> cnt_out(nb, 0);
    for (int j = 0; j < n; j++) {
        if (sc.Gam[j].size() > 4) { err = "build_pcr_tables: more than 4 patch rows per free unknown"; return false; }
        int q = 0;
        for (auto &e : sc.Gam[j]) {
            t.gin_row[(size_t)j * 4 + q] = e.first; t.gin_w[(size_t)j * 4 + q] = e.second; q++;
            if (cnt_out[e.first] >= 2) { err = "build_pcr_tables: more than 2 free unknowns per patch row"; return false; }
            t.gout_j[(size_t)e.first * 2 + cnt_out[e.first]] = j; t.gout_w[(size_t)e.first * 2 + cnt_out[e.first]] = e.second;
            cnt_out[e.first]++;
        }
    }
    t.G.clear();
    if (sc.periodic) {
        // M_c = M_b + E C E^T over the edge unknowns e = {0, 1, 2, n-3, n-2, n-1};  x = y - W (I + C E^T W)^-1 C E^T y,  W = M_b^-1 E
        if (n < 7) { err = "build_pcr_tables: periodic class with fewer than 7 unknowns"; return false; }
        const int e[6] = {0, 1, 2, n - 3, n - 2, n - 1};
        // W by banded Gaussian elimination (no pivoting: M_b is symmetric positive definite) in extended precision
        std::vector<pxr> band((size_t)n * 7), W((size_t)n * 6, 0.0L);
        for (int i = 0; i < n; i++)
            for (int d = -3; d <= 3; d++) band[(size_t)i * 7 + d + 3] = (i + d >= 0 && i + d < n) ? Mb(i, i + d) : 0.0L;
        for (int q = 0; q < 6; q++) W[(size_t)e[q] * 6 + q] = 1.0L;
        for (int k = 0; k < n; k++) {
            const pxr piv = band[(size_t)k * 7 + 3];
            if (!(piv > 0.0L)) { err = "build_pcr_tables: band part of the periodic matrix is not positive definite"; return false; }
            for (int i = k + 1; i <= std::min(n - 1, k + 3); i++) {
                const pxr f = band[(size_t)i * 7 + (k - i) + 3] / piv;
                if (f == 0.0L) continue;
                for (int j = k; j <= std::min(n - 1, k + 3); j++) band[(size_t)i * 7 + (j - i) + 3] -= f * band[(size_t)k * 7 + (j - k) + 3];
                for (int q = 0; q < 6; q++) W[(size_t)i * 6 + q] -= f * W[(size_t)k * 6 + q];
            }
        }
        for (int k = n - 1; k >= 0; k--)
            for (int q = 0; q < 6; q++) {
                pxr s = W[(size_t)k * 6 + q];
                for (int j = k + 1; j <= std::min(n - 1, k + 3); j++) s -= band[(size_t)k * 7 + (j - k) + 3] * W[(size_t)j * 6 + q];
                W[(size_t)k * 6 + q] = s / band[(size_t)k * 7 + 3];
            }
        pxr C[6][6], S[6][6], Si[6][6];
        for (int p = 0; p < 6; p++)
            for (int q = 0; q < 6; q++) C[p][q] = std::abs(e[p] - e[q]) > 3 ? Mc(e[p], e[q]) : 0.0L;
        for (int p = 0; p < 6; p++)
            for (int q = 0; q < 6; q++) {
                pxr s = p == q ? 1.0L : 0.0L;
                for (int k = 0; k < 6; k++) s += C[p][k] * W[(size_t)e[k] * 6 + q];
                S[p][q] = s;
                Si[p][q] = p == q ? 1.0L : 0.0L;
            }
        for (int c = 0; c < 6; c++) {            // Gauss-Jordan with partial pivoting, 6 x 6
            int piv = c;
            for (int r = c + 1; r < 6; r++) if (fabsl(S[r][c]) > fabsl(S[piv][c])) piv = r;
            if (S[piv][c] == 0.0L) { err = "build_pcr_tables: singular periodic correction"; return false; }
            for (int j = 0; j < 6; j++) { std::swap(S[piv][j], S[c][j]); std::swap(Si[piv][j], Si[c][j]); }
            const pxr d = 1.0L / S[c][c];
            for (int j = 0; j < 6; j++) { S[c][j] *= d; Si[c][j] *= d; }
            for (int r = 0; r < 6; r++) {
                if (r == c) continue;
                const pxr f = S[r][c];
                for (int j = 0; j < 6; j++) { S[r][j] -= f * S[c][j]; Si[r][j] -= f * Si[c][j]; }
            }
        }
        pxr SC[6][6];
        for (int p = 0; p < 6; p++)
            for (int q = 0; q < 6; q++) { pxr s = 0.0L; for (int k = 0; k < 6; k++) s += Si[p][k] * C[k][q]; SC[p][q] = s; }
        t.G.assign((size_t)np * 6, 0.0);
        for (int i = 0; i < n; i++)
            for (int q = 0; q < 6; q++) { pxr s = 0.0L; for (int k = 0; k < 6; k++) s += W[(size_t)i * 6 + k] * SC[k][q]; t.G[(size_t)i * 6 + q] = (double)s; }
    }
    return true;
}

void pcr_apply_host(const PcrTables &t, int nb, const double *b, double *a) {
    const int np = 3 * t.nblk;
    std::vector<double> r(np, 0.0), rn(np, 0.0);
    for (int j = 0; j < t.n; j++) {
        double s = 0.0;
        for (int q = 0; q < 4; q++) if (t.gin_row[(size_t)j * 4 + q] >= 0) s += t.gin_w[(size_t)j * 4 + q] * b[t.gin_row[(size_t)j * 4 + q]];
        r[j] = s;
    }
    for (int l = 0; l < t.levels; l++) {
        const int s = 1 << l;
        for (int i = 0; i < t.nblk; i++) {
            const double *c = t.coef.data() + ((size_t)l * t.nblk + i) * 18;
            for (int p = 0; p < 3; p++) {
                double v = r[3 * i + p];
                for (int q = 0; q < 3; q++) {
                    if (i - s >= 0) v -= c[p * 3 + q] * r[3 * (i - s) + q];
                    if (i + s < t.nblk) v -= c[9 + p * 3 + q] * r[3 * (i + s) + q];
                }
                rn[3 * i + p] = v;
            }
        }
        r.swap(rn);
    }
    std::vector<double> x(np, 0.0);
    for (int i = 0; i < t.nblk; i++)
        for (int p = 0; p < 3; p++) {
            double v = 0.0;
            for (int q = 0; q < 3; q++) v += t.dinv[(size_t)i * 9 + p * 3 + q] * r[3 * i + q];
            x[3 * i + p] = v;
        }
    if (t.periodic) {
        const int e[6] = {0, 1, 2, t.n - 3, t.n - 2, t.n - 1};
        double ye[6];
        for (int q = 0; q < 6; q++) ye[q] = x[e[q]];
        for (int i = 0; i < t.n; i++) {
            double v = x[i];
            for (int q = 0; q < 6; q++) v -= t.G[(size_t)i * 6 + q] * ye[q];
            x[i] = v;
        }
    }
    for (int m = 0; m < nb; m++) {
        double s = 0.0;
        for (int q = 0; q < 2; q++) if (t.gout_j[(size_t)m * 2 + q] >= 0) s += t.gout_w[(size_t)m * 2 + q] * x[t.gout_j[(size_t)m * 2 + q]];
        a[m] = s;
    }
}

void cholesky_apply_host(const SplineClass &sc, int nb, const double *b, double *a) {
    // dense statement of what k_solve computes: rhs = Gamma b, L L^T x = rhs with the class's factor rows, a = Gamma^T x
    const int n = sc.nfree;
    std::vector<long double> rhs(n, 0.0L), y(n), x(n);
    for (int j = 0; j < n; j++)
        for (auto &e : sc.Gam[j]) rhs[j] += (long double)e.second * b[e.first];
    auto Lf = [&](int i, int j) -> long double {          // factor entry (i >= j)
        if (i - j <= 3) return sc.Lband[(size_t)i * 4 + (3 - (i - j))];
        if (sc.periodic && i >= n - 3) return sc.Larrow[(size_t)(i - (n - 3)) * nb + j];
        return 0.0L;
    };
    for (int i = 0; i < n; i++) {
        long double s = rhs[i];
        for (int j = (sc.periodic && i >= n - 3) ? 0 : std::max(0, i - 3); j < i; j++) s -= Lf(i, j) * y[j];
        y[i] = s / Lf(i, i);
    }
    for (int i = n - 1; i >= 0; i--) {
        long double s = y[i];
        for (int j = i + 1; j < n; j++) {
            const long double l = Lf(j, i);
            if (l != 0.0L) s -= l * x[j];
        }
        x[i] = s / Lf(i, i);
    }
    for (int m = 0; m < nb; m++) a[m] = 0.0;
    for (int j = 0; j < n; j++)
        for (auto &e : sc.Gam[j]) a[e.first] += (double)((long double)e.second * x[j]);
}

// ---------------------------------------------------------------------------------------------- Chebyshev
// The collocation operators are assembled in extended precision and rounded to double once at the end: the
// second-derivative matrix has entries O(N^4 / Lz^2), and products of double-rounded factors would carry a relative
// error of cond * eps into every step (visible as ~1e-9 field differences at zDim = 128).
typedef long double xreal;
typedef std::vector<xreal> xmat;

static void matmul(const xmat &A, const xmat &B, xmat &Cm, int n, int k, int m) {
    Cm.assign((size_t)n * m, 0.0L);
    for (int i = 0; i < n; i++)
        for (int p = 0; p < k; p++) {
            xreal a = A[(size_t)i * k + p];
            if (a == 0.0L) continue;
            for (int j = 0; j < m; j++) Cm[(size_t)i * m + j] += a * B[(size_t)p * m + j];
        }
}

static std::vector<double> to_double(const xmat &A) { return std::vector<double>(A.begin(), A.end()); }

struct ChebX {       // extended-precision factors shared by build_cheb_ops and build_helmholtz
    xmat T, Dc, TD, TDD;
};

static void cheb_factors(double zmin, double zmax, int N, ChebX &x) {
    const xreal PI = 4.0L * atanl(1.0L);
    const xreal Lz = (xreal)zmax - (xreal)zmin;
    x.T.assign((size_t)N * N, 0.0L);
    for (int n = 0; n < N; n++)
        for (int k = 0; k < N; k++)
            x.T[(size_t)n * N + k] = ((k == 0 || k == N - 1) ? 1.0L : 2.0L) * cosl((xreal)((long)n * k) * PI / (xreal)(N - 1));
    // coefficient-space derivative: ax_{k-1} = ax_{k+1} + k c_k with c_k = 2 a_k (interior), a_{N-1} (last)
    x.Dc.assign((size_t)N * N, 0.0L);
    std::vector<xreal> ax(N + 2);
    for (int j = 0; j < N; j++) {
        std::fill(ax.begin(), ax.end(), 0.0L);
        for (int k = N - 1; k >= 1; k--) {
            xreal ck = (k == j) ? ((k == N - 1) ? 1.0L : 2.0L) : 0.0L;
            ax[k - 1] = ax[k + 1] + k * ck;
        }
        for (int i = 0; i < N; i++) x.Dc[(size_t)i * N + j] = ax[i] * (-2.0L / Lz);
    }
    matmul(x.T, x.Dc, x.TD, N, N, N);
    matmul(x.TD, x.Dc, x.TDD, N, N, N);
}

bool build_cheb_ops(double zmin, double zmax, int nz, int Zb, int bcb, int bct, ChebOps &o, std::string &err) {
    if (nz < 4) { err = "zDim must be >= 4"; return false; }
    o.bcb = bcb;
    o.bct = bct;
    const int N = nz;
    const xreal PI = 4.0L * atanl(1.0L);
    const xreal Lz = (xreal)zmax - (xreal)zmin;
    o.z.resize(N);
    for (int n = 0; n < N; n++) o.z[n] = std::cos(n * M_PI / (N - 1)) * (-0.5 * (zmax - zmin)) + 0.5 * (zmin + zmax);
    (void)PI;
    ChebX x;
    cheb_factors(zmin, zmax, N, x);
    xmat CB((size_t)Zb * N);
    for (int k = 0; k < Zb; k++)
        for (int n = 0; n < N; n++) CB[(size_t)k * N + n] = x.T[(size_t)k * N + n] / (2.0L * (N - 1));
    // coefficient-space integral, zero at the bottom (x = +1 where every T_k = 1)
    xmat Ic((size_t)N * N, 0.0L);
    std::vector<xreal> ai(N);
    for (int j = 0; j < N; j++) {
        std::fill(ai.begin(), ai.end(), 0.0L);
        auto a = [&](int k) { return k == j ? 1.0L : 0.0L; };
        for (int k = 1; k < N - 1; k++) {
            xreal up = (k + 1 < N - 1) ? a(k + 1) : 0.5L * a(k + 1);
            ai[k] = (a(k - 1) - up) / (2.0L * k);
        }
        ai[N - 1] = a(N - 2) / (xreal)(N - 1);
        xreal s = 0.0L;
        for (int k = 1; k < N; k++) {
            ai[k] *= (-0.5L * Lz);
            s += (k == N - 1 ? 1.0L : 2.0L) * ai[k];
        }
        ai[0] = -s;
        for (int i = 0; i < N; i++) Ic[(size_t)i * N + j] = ai[i];
    }
    // BC projection (orthogonal projection onto the null space of the constraint rows)
    std::vector<std::vector<xreal>> rows;
    const int bcs[2] = {bcb, bct}, rix[2] = {0, N - 1};
    for (int s = 0; s < 2; s++) {
        const xmat *src = nullptr;
        switch (bcs[s]) {
            case SX_BC_R0: continue;
            case SX_BC_R1T0: src = &x.T; break;
            case SX_BC_R1T1: src = &x.TD; break;
            case SX_BC_R1T2: src = &x.TDD; break;
            default: err = "unsupported vertical boundary condition"; return false;
        }
        rows.emplace_back(src->begin() + (size_t)rix[s] * N, src->begin() + (size_t)(rix[s] + 1) * N);
    }
    xmat proj((size_t)N * N, 0.0L);
    for (int i = 0; i < N; i++) proj[(size_t)i * N + i] = 1.0L;
    if (!rows.empty()) {
        const int m = (int)rows.size();
        xreal G[2][2] = {{0, 0}, {0, 0}}, Gi[2][2] = {{0, 0}, {0, 0}};
        for (int a = 0; a < m; a++)
            for (int b = 0; b < m; b++)
                for (int k = 0; k < N; k++) G[a][b] += rows[a][k] * rows[b][k];
        if (m == 1) {
            Gi[0][0] = 1.0L / G[0][0];
        } else {
            xreal det = G[0][0] * G[1][1] - G[0][1] * G[1][0];
            Gi[0][0] = G[1][1] / det; Gi[0][1] = -G[0][1] / det; Gi[1][0] = -G[1][0] / det; Gi[1][1] = G[0][0] / det;
        }
        for (int i = 0; i < N; i++)
            for (int j = 0; j < N; j++) {
                xreal s = 0.0L;
                for (int a = 0; a < m; a++)
                    for (int b = 0; b < m; b++) s += rows[a][i] * Gi[a][b] * rows[b][j];
                proj[(size_t)i * N + j] -= s;
            }
    }
    xmat CA((size_t)N * Zb, 0.0L);
    for (int i = 0; i < N; i++)
        for (int k = 0; k < Zb; k++) CA[(size_t)i * Zb + k] = proj[(size_t)i * N + k];
    xmat M0, M1, M2, TI, TICA, Mint, Mdz, Mrec, Mdzz;
    matmul(x.T, CA, M0, N, N, Zb);
    matmul(x.TD, CA, M1, N, N, Zb);
    matmul(x.TDD, CA, M2, N, N, Zb);
    matmul(x.T, Ic, TI, N, N, N);
    matmul(TI, CA, TICA, N, N, Zb);
    matmul(TICA, CB, Mint, N, Zb, N);
    matmul(M1, CB, Mdz, N, Zb, N);
    matmul(M0, CB, Mrec, N, Zb, N);
    matmul(M2, CB, Mdzz, N, Zb, N);
    o.T = to_double(x.T); o.Dc = to_double(x.Dc); o.CB = to_double(CB); o.CA = to_double(CA);
    o.M[0] = to_double(M0); o.M[1] = to_double(M1); o.M[2] = to_double(M2);
    o.Mint = to_double(Mint); o.Mdz = to_double(Mdz); o.Mrec = to_double(Mrec); o.Mdzz = to_double(Mdzz);
    o.zmin = zmin; o.zmax = zmax;
    return true;
}

bool build_helmholtz(const ChebOps &w, double pxi_bar, double tau, std::vector<double> &Wmat, std::vector<double> &Xmat,
                     std::string &err) {
    const int N = (int)w.z.size();
    const xreal c = (xreal)tau * (xreal)tau * (xreal)pxi_bar;
    ChebX x;
    cheb_factors(w.zmin, w.zmax, N, x);
    // H = [c T[0,:]; c T[N-1,:]; (c TDD - T)[1..N-2, :]]   (src/semiimplicit.jl:776-779)
    xmat H((size_t)N * N), Inv((size_t)N * N, 0.0L);
    for (int j = 0; j < N; j++) {
        H[j] = c * x.T[j];
        H[(size_t)N + j] = c * x.T[(size_t)(N - 1) * N + j];
        for (int i = 1; i < N - 1; i++) H[(size_t)(i + 1) * N + j] = c * x.TDD[(size_t)i * N + j] - x.T[(size_t)i * N + j];
    }
    for (int i = 0; i < N; i++) Inv[(size_t)i * N + i] = 1.0L;
    // Gauss-Jordan with partial pivoting in extended precision (the reference factorises in double, :780)
    for (int col = 0; col < N; col++) {
        int piv = col;
        for (int r = col + 1; r < N; r++)
            if (fabsl(H[(size_t)r * N + col]) > fabsl(H[(size_t)piv * N + col])) piv = r;
        if (H[(size_t)piv * N + col] == 0.0L) { err = "singular Helmholtz matrix"; return false; }
        if (piv != col)
            for (int j = 0; j < N; j++) {
                std::swap(H[(size_t)piv * N + j], H[(size_t)col * N + j]);
                std::swap(Inv[(size_t)piv * N + j], Inv[(size_t)col * N + j]);
            }
        xreal d = 1.0L / H[(size_t)col * N + col];
        for (int j = 0; j < N; j++) { H[(size_t)col * N + j] *= d; Inv[(size_t)col * N + j] *= d; }
        for (int r = 0; r < N; r++) {
            if (r == col) continue;
            xreal f = H[(size_t)r * N + col];
            if (f == 0.0L) continue;
            for (int j = 0; j < N; j++) {
                H[(size_t)r * N + j] -= f * H[(size_t)col * N + j];
                Inv[(size_t)r * N + j] -= f * Inv[(size_t)col * N + j];
            }
        }
    }
    xmat W, X;
    matmul(x.T, Inv, W, N, N, N);
    matmul(x.TD, Inv, X, N, N, N);
    Wmat = to_double(W);
    Xmat = to_double(X);
    return true;
}

void ring_table(int has_l, int uniform_L, int ri, int &L, int &kmax, double &off) {
    if (!has_l) { L = 1; kmax = 0; off = 0.0; return; }
    if (uniform_L > 0) {
        L = uniform_L;
        kmax = ri < uniform_L / 2 - 1 ? ri : uniform_L / 2 - 1;
        off = 0.0;
    } else {
        L = 4 + 4 * ri;
        kmax = ri;
        off = 0.5 * (2.0 * M_PI / L) * (ri - 1);
    }
}

}  // namespace sx
