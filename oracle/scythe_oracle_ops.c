/*
 * TEST INFRASTRUCTURE — NOT PRODUCT CODE.
 *
 * Operator construction of the C oracle, written from the definitions independently of
 * oracle/oracle_np.py: cubic-B-spline basis tables, quadrature weights, the boundary-condition projection and Cholesky
 * factor of Gamma (P + eps_q Q) Gamma^T, the Chebyshev column operators (extended precision, rounded once) and the
 * Helmholtz operator of calc_Helmholtz_semiimplicit_matrix (src/semiimplicit.jl:768-781).  With these the C port no longer
 * borrows its operators from the numpy definition, so "numpy definition vs C port" compares two constructions as well as
 * two ways of applying them (tests/test_oracle_consistency.py).
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library.
 */
#ifndef _GNU_SOURCE
#define _GNU_SOURCE
#endif
#include <math.h>
#include <stdlib.h>
#include <string.h>

typedef long double xr;

enum { BC_R0 = 0, BC_R1T0, BC_R1T1, BC_R1T2, BC_R2T10, BC_R2T20, BC_R3, BC_PERIODIC };

/* cardinal cubic B-spline and its derivatives with respect to delta */
static double bsp(double delta, int d) {
    double z = fabs(delta), s = delta > 0 ? 1.0 : -1.0;
    if (z >= 2.0) return 0.0;
    double p = 2.0 - z, q = z < 1.0 ? 1.0 - z : 0.0;
    switch (d) {
        case 0: return p * p * p / 6.0 - 4.0 * q * q * q / 6.0;
        case 1: return -s * (p * p / 2.0 - 2.0 * q * q);
        case 2: return p - 4.0 * q;
        default: return s * (z < 1.0 ? 3.0 : -1.0);
    }
}

/* mish point mu of patch cell c, and the 4 basis functions (nodes c-1 .. c+2, array index c .. c+3) that do not vanish there */
void orc_ops_phi(double xmin, double DX, int cell, int mu, int d, double *out4, double *x_out) {
    const double off = (mu == 0 ? -1.0 : mu == 2 ? 1.0 : 0.0) * (sqrt(3.0 / 5.0) / 2.0);
    const double x = xmin + DX * (cell + 0.5 + off);
    for (int j = 0; j < 4; j++) {
        const double xm = xmin + (cell - 1 + j) * DX;
        out4[j] = bsp((x - xm) / DX, d) / pow(DX, d);
    }
    if (x_out) *x_out = x;
}

void orc_ops_wq(double DX, double *w3) { w3[0] = DX * 8.0 / 21.0; w3[1] = DX * 5.0 / 21.0; w3[2] = DX * 8.0 / 21.0; }

static int bc_rank(int bc) { return bc == BC_R0 ? 0 : bc <= BC_R1T2 ? 1 : bc <= BC_R2T20 ? 2 : bc == BC_R3 ? 3 : -1; }

/* Gamma [nfree][nb]: full coefficients a = Gamma^T a_free */
static int gamma_mat(int nc, int bcl, int bcr, double *G, int *nfree_out) {
    const int nb = nc + 3;
    if (bcl == BC_PERIODIC || bcr == BC_PERIODIC) {
        if (bcl != bcr) return 1;
        memset(G, 0, sizeof(double) * nc * nb);
        for (int m = -1; m <= nc + 1; m++) G[(((m % nc) + nc) % nc) * nb + m + 1] = 1.0;
        *nfree_out = nc;
        return 0;
    }
    const int rl = bc_rank(bcl), rr = bc_rank(bcr);
    if (rl < 0 || rr < 0) return 1;
    const int n = nb - rl - rr;
    memset(G, 0, sizeof(double) * n * nb);
    for (int j = 0; j < n; j++) G[j * nb + rl + j] = 1.0;
    /* rank 1: a_{-1} = alpha a_0 + beta a_1;  R2T10: a_{-1} = a_1, a_0 = -a_1 / 2;  R2T20: a_0 = 0, a_{-1} = -a_1;  R3: first three zero */
    if (bcl == BC_R1T0) { G[0 * nb + 0] += -4.0; G[1 * nb + 0] += -1.0; }
    else if (bcl == BC_R1T1) { G[1 * nb + 0] += 1.0; }
    else if (bcl == BC_R1T2) { G[0 * nb + 0] += 2.0; G[1 * nb + 0] += -1.0; }
    else if (bcl == BC_R2T10) { G[0 * nb + 0] += 1.0; G[0 * nb + 1] += -0.5; }
    else if (bcl == BC_R2T20) { G[0 * nb + 0] += -1.0; }
    if (bcr == BC_R1T0) { G[(n - 1) * nb + nb - 1] += -4.0; G[(n - 2) * nb + nb - 1] += -1.0; }
    else if (bcr == BC_R1T1) { G[(n - 2) * nb + nb - 1] += 1.0; }
    else if (bcr == BC_R1T2) { G[(n - 1) * nb + nb - 1] += 2.0; G[(n - 2) * nb + nb - 1] += -1.0; }
    else if (bcr == BC_R2T10) { G[(n - 1) * nb + nb - 1] += 1.0; G[(n - 1) * nb + nb - 2] += -0.5; }
    else if (bcr == BC_R2T20) { G[(n - 1) * nb + nb - 1] += -1.0; }
    *nfree_out = n;
    return 0;
}

/* One boundary-condition class of the B -> A solve: returns 0 on success. Lband [nb][4] = L[i][i-3..i], Larrow [3][nb] = last
 * three rows of L (periodic), gl / gr [3][2] = dependent boundary coefficients in terms of the first / last two free ones. */
int orc_ops_spline_class(double xmin, double xmax, int nc, double l_q, int bcl, int bcr, int *nfree, int *periodic, int *rl_out,
                         int *rr_out, double *gl, double *gr, double *Lband, double *Larrow) {
    const int nb = nc + 3;
    const double DX = (xmax - xmin) / nc;
    const double twopi = 2.0 * M_PI;
    const double eps_q = pow(l_q * DX / twopi, 6.0);
    double w3[3];
    orc_ops_wq(DX, w3);
    double *P = (double *)calloc((size_t)nb * nb, sizeof(double));
    double *G = (double *)calloc((size_t)nb * nb, sizeof(double));
    for (int c = 0; c < nc; c++)
        for (int mu = 0; mu < 3; mu++) {
            double p0[4], p3[4];
            orc_ops_phi(xmin, DX, c, mu, 0, p0, NULL);
            orc_ops_phi(xmin, DX, c, mu, 3, p3, NULL);
            for (int a = 0; a < 4; a++)
                for (int b = 0; b < 4; b++) P[(c + a) * nb + c + b] += w3[mu] * (p0[a] * p0[b] + eps_q * p3[a] * p3[b]);
        }
    int n = 0;
    if (gamma_mat(nc, bcl, bcr, G, &n)) { free(P); free(G); return 1; }
    /* PQ = G P G^T */
    double *T = (double *)calloc((size_t)n * nb, sizeof(double));
    double *PQ = (double *)calloc((size_t)n * n, sizeof(double));
    for (int i = 0; i < n; i++)
        for (int k = 0; k < nb; k++) {
            const double g = G[i * nb + k];
            if (g == 0.0) continue;
            for (int j = 0; j < nb; j++) T[i * nb + j] += g * P[k * nb + j];
        }
    for (int i = 0; i < n; i++)
        for (int j = 0; j < n; j++) {
            double s = 0.0;
            for (int k = 0; k < nb; k++) s += T[i * nb + k] * G[j * nb + k];
            PQ[i * n + j] = s;
        }
    /* Cholesky, lower, in place */
    for (int j = 0; j < n; j++) {
        double d = PQ[j * n + j];
        for (int k = 0; k < j; k++) d -= PQ[j * n + k] * PQ[j * n + k];
        if (!(d > 0.0)) { free(P); free(G); free(T); free(PQ); return 2; }
        d = sqrt(d);
        PQ[j * n + j] = d;
        for (int i = j + 1; i < n; i++) {
            double s = PQ[i * n + j];
            for (int k = 0; k < j; k++) s -= PQ[i * n + k] * PQ[j * n + k];
            PQ[i * n + j] = s / d;
        }
    }
    const int per = (bcl == BC_PERIODIC);
    *nfree = n; *periodic = per;
    *rl_out = per ? 0 : bc_rank(bcl); *rr_out = per ? 0 : bc_rank(bcr);
    memset(gl, 0, sizeof(double) * 6); memset(gr, 0, sizeof(double) * 6);
    memset(Lband, 0, sizeof(double) * nb * 4); memset(Larrow, 0, sizeof(double) * 3 * nb);
    if (per) {
        for (int q = 0; q < 3; q++)
            for (int j = 0; j <= n - 3 + q; j++) Larrow[q * nb + j] = PQ[(n - 3 + q) * n + j];      /* lower triangle only */
    } else {
        for (int i = 0; i < *rl_out; i++) { gl[i * 2] = G[0 * nb + i]; gl[i * 2 + 1] = G[1 * nb + i]; }
        for (int i = 0; i < *rr_out; i++) { gr[i * 2] = G[(n - 1) * nb + nb - 1 - i]; gr[i * 2 + 1] = G[(n - 2) * nb + nb - 1 - i]; }
    }
    for (int i = 0; i < n; i++)
        for (int q = 0; q < 4; q++)
            if (i - q >= 0) Lband[i * 4 + 3 - q] = PQ[i * n + i - q];
    free(P); free(G); free(T); free(PQ);
    return 0;
}

/* ------------------------------------------------------------------ Chebyshev column (Gauss-Lobatto, index 0 = bottom) */
typedef struct { int N; xr *T, *Dc, *Ic, *TD, *TDD; } chx;

static xr *xalloc(size_t n) { return (xr *)calloc(n, sizeof(xr)); }
static void xmm(const xr *A, const xr *B, xr *C, int n, int m, int p) {   /* C[n][p] = A[n][m] B[m][p] */
    for (int i = 0; i < n; i++)
        for (int j = 0; j < p; j++) {
            xr s = 0;
            for (int k = 0; k < m; k++) s += A[i * m + k] * B[k * p + j];
            C[i * p + j] = s;
        }
}

static void chx_build(chx *c, xr zmin, xr zmax, int N) {
    const xr PI = 4.0L * atanl(1.0L), Lz = zmax - zmin;
    c->N = N;
    c->T = xalloc((size_t)N * N); c->Dc = xalloc((size_t)N * N); c->Ic = xalloc((size_t)N * N);
    c->TD = xalloc((size_t)N * N); c->TDD = xalloc((size_t)N * N);
    for (int n = 0; n < N; n++)
        for (int k = 0; k < N; k++) c->T[n * N + k] = ((k == 0 || k == N - 1) ? 1.0L : 2.0L) * cosl((xr)n * k * PI / (N - 1));
    /* u = a_0 + 2 sum a_k T_k + a_{N-1} T_{N-1}: derivative and integral in coefficient space, column j = image of e_j */
    for (int j = 0; j < N; j++) {
        xr *ax = xalloc(N + 2);
        for (int k = N - 1; k >= 1; k--) {
            const xr ck = (k == j) ? ((k == N - 1) ? 1.0L : 2.0L) : 0.0L;
            ax[k - 1] = ax[k + 1] + k * ck;
        }
        for (int k = 0; k < N; k++) c->Dc[k * N + j] = ax[k] * (-2.0L / Lz);
        free(ax);
        xr *ai = xalloc(N);
        for (int k = 1; k < N - 1; k++) {
            const xr lo = (k - 1 == j) ? 1.0L : 0.0L;
            const xr up = (k + 1 == j) ? ((k + 1 < N - 1) ? 1.0L : 0.5L) : 0.0L;
            ai[k] = (lo - up) / (2.0L * k);
        }
        ai[N - 1] = (N - 2 == j) ? 1.0L / (N - 1) : 0.0L;
        xr s0 = 0;
        for (int k = 0; k < N; k++) ai[k] *= -0.5L * Lz;
        for (int k = 1; k < N - 1; k++) s0 += 2.0L * ai[k];
        ai[0] = -(s0 + ai[N - 1]);
        for (int k = 0; k < N; k++) c->Ic[k * N + j] = ai[k];
        free(ai);
    }
    xmm(c->T, c->Dc, c->TD, N, N, N);
    xmm(c->TD, c->Dc, c->TDD, N, N, N);
}
static void chx_free(chx *c) { free(c->T); free(c->Dc); free(c->Ic); free(c->TD); free(c->TDD); }

/* b -> a: zero padding to N followed by the orthogonal projection onto the null space of the boundary conditions */
static int chx_ca(const chx *c, int Zb, int bcb, int bct, xr *CA /* [N][Zb] */) {
    const int N = c->N;
    xr rows[2][512];
    int nr = 0;
    const int bcs[2] = {bcb, bct}, at[2] = {0, N - 1};
    if (N > 512) return 1;
    for (int e = 0; e < 2; e++) {
        if (bcs[e] == BC_R0) continue;
        const xr *src = bcs[e] == BC_R1T0 ? c->T : bcs[e] == BC_R1T1 ? c->TD : bcs[e] == BC_R1T2 ? c->TDD : NULL;
        if (!src) return 1;
        for (int k = 0; k < N; k++) rows[nr][k] = src[at[e] * N + k];
        nr++;
    }
    xr *proj = xalloc((size_t)N * N);
    for (int i = 0; i < N; i++) proj[i * N + i] = 1.0L;
    if (nr) {
        xr g[2][2] = {{0, 0}, {0, 0}}, gi[2][2];
        for (int a = 0; a < nr; a++)
            for (int b = 0; b < nr; b++)
                for (int k = 0; k < N; k++) g[a][b] += rows[a][k] * rows[b][k];
        if (nr == 1) gi[0][0] = 1.0L / g[0][0];
        else {
            const xr det = g[0][0] * g[1][1] - g[0][1] * g[1][0];
            gi[0][0] = g[1][1] / det; gi[0][1] = -g[0][1] / det; gi[1][0] = -g[1][0] / det; gi[1][1] = g[0][0] / det;
        }
        for (int i = 0; i < N; i++)
            for (int j = 0; j < N; j++) {
                xr s = 0;
                for (int a = 0; a < nr; a++)
                    for (int b = 0; b < nr; b++) s += rows[a][i] * gi[a][b] * rows[b][j];
                proj[i * N + j] -= s;
            }
    }
    for (int i = 0; i < N; i++)
        for (int j = 0; j < Zb; j++) CA[i * Zb + j] = proj[i * N + j];
    free(proj);
    return 0;
}

static void round_out(const xr *src, double *dst, size_t n) { for (size_t i = 0; i < n; i++) dst[i] = (double)src[i]; }

/* z [N]; M [3][N][Zb] (b -> value, d/dz, d2/dz2); CB [Zb][N]; Vint, Vdz, Vrec [N][N] (values -> CB, CA, CIInt / CIx / CI);
 * T, D1, D2 [N][N] (dct_matrix, dct_1st_derivative, dct_2nd_derivative as Float64).  Any output may be NULL. */
int orc_ops_cheb(double zmin, double zmax, int N, int Zb, int bcb, int bct, double *z, double *M, double *CB, double *Vint,
                 double *Vdz, double *Vrec, double *T, double *D1, double *D2) {
    chx c;
    chx_build(&c, zmin, zmax, N);
    xr *CA = xalloc((size_t)N * Zb), *CBx = xalloc((size_t)Zb * N), *t1 = xalloc((size_t)N * N), *t2 = xalloc((size_t)N * N);
    if (chx_ca(&c, Zb, bcb, bct, CA)) { chx_free(&c); free(CA); free(CBx); free(t1); free(t2); return 1; }
    for (int k = 0; k < Zb; k++)
        for (int n = 0; n < N; n++) CBx[k * N + n] = c.T[k * N + n] / (2.0L * (N - 1));
    if (z) for (int n = 0; n < N; n++) z[n] = cos(n * M_PI / (N - 1)) * (-0.5 * (zmax - zmin)) + 0.5 * (zmin + zmax);
    if (CB) round_out(CBx, CB, (size_t)Zb * N);
    const xr *ops[3] = {c.T, c.TD, c.TDD};
    for (int d = 0; d < 3; d++) {
        xmm(ops[d], CA, t1, N, N, Zb);
        if (M) round_out(t1, M + (size_t)d * N * Zb, (size_t)N * Zb);
    }
    xr *cacb = xalloc((size_t)N * N);
    xmm(CA, CBx, cacb, N, Zb, N);
    if (Vrec) { xmm(c.T, cacb, t1, N, N, N); round_out(t1, Vrec, (size_t)N * N); }
    if (Vdz) { xmm(c.TD, cacb, t1, N, N, N); round_out(t1, Vdz, (size_t)N * N); }
    if (Vint) { xmm(c.T, c.Ic, t2, N, N, N); xmm(t2, cacb, t1, N, N, N); round_out(t1, Vint, (size_t)N * N); }
    if (T) round_out(c.T, T, (size_t)N * N);
    if (D1) round_out(c.TD, D1, (size_t)N * N);
    if (D2) round_out(c.TDD, D2, (size_t)N * N);
    chx_free(&c); free(CA); free(CBx); free(t1); free(t2); free(cacb);
    return 0;
}

/* W = T H^-1 and X = T Dc H^-1 [N][N] with H of calc_Helmholtz_semiimplicit_matrix (src/semiimplicit.jl:768-781) assembled and
 * inverted (Gauss-Jordan, partial pivoting) in extended precision: the exact-arithmetic arbiter, see oracle_np.HelmholtzLU for
 * the reference's own Float64 LU arithmetic. */
int orc_ops_helmholtz(double zmin, double zmax, int N, double pxi_bar, double tau, double *W, double *X) {
    chx c;
    chx_build(&c, zmin, zmax, N);
    const xr cc = (xr)tau * (xr)tau * (xr)pxi_bar;
    const int W2 = 2 * N;
    xr *A = xalloc((size_t)N * W2);
    for (int j = 0; j < N; j++) {
        A[0 * W2 + j] = cc * c.T[0 * N + j];
        A[1 * W2 + j] = cc * c.T[(N - 1) * N + j];
        for (int i = 1; i < N - 1; i++) A[(i + 1) * W2 + j] = cc * c.TDD[i * N + j] - c.T[i * N + j];
    }
    for (int i = 0; i < N; i++) A[i * W2 + N + i] = 1.0L;
    for (int col = 0; col < N; col++) {
        int p = col;
        for (int i = col + 1; i < N; i++) if (fabsl(A[i * W2 + col]) > fabsl(A[p * W2 + col])) p = i;
        if (A[p * W2 + col] == 0.0L) { chx_free(&c); free(A); return 1; }
        if (p != col) for (int j = 0; j < W2; j++) { xr t = A[col * W2 + j]; A[col * W2 + j] = A[p * W2 + j]; A[p * W2 + j] = t; }
        const xr d = A[col * W2 + col];
        for (int j = 0; j < W2; j++) A[col * W2 + j] /= d;
        for (int i = 0; i < N; i++) {
            if (i == col) continue;
            const xr f = A[i * W2 + col];
            if (f == 0.0L) continue;
            for (int j = 0; j < W2; j++) A[i * W2 + j] -= f * A[col * W2 + j];
        }
    }
    xr *Hi = xalloc((size_t)N * N), *t1 = xalloc((size_t)N * N);
    for (int i = 0; i < N; i++)
        for (int j = 0; j < N; j++) Hi[i * N + j] = A[i * W2 + N + j];
    xmm(c.T, Hi, t1, N, N, N);
    round_out(t1, W, (size_t)N * N);
    xmm(c.TD, Hi, t1, N, N, N);
    round_out(t1, X, (size_t)N * N);
    chx_free(&c); free(A); free(Hi); free(t1);
    return 0;
}
