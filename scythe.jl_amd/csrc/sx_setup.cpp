// Host-side operator construction for libscythe_hip.so.
//
// Everything here restates arithmetic the reference obtains from the external Springsteel.jl package
// (Project.toml:20) - cubic B-spline basis, quadrature, P + eps_q Q assembly, boundary-condition projection,
// Cholesky factorisation, Chebyshev collocation / derivative / integral operators - following SURVEY.md 8(c).
// The quadrature-weight ratio 8:5:8 is the one the reference's notebook known-answer pins.
#include "sx_internal.hpp"
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
