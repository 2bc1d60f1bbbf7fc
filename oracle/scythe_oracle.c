/*
 * TEST INFRASTRUCTURE — NOT PRODUCT CODE.
 *
 * Plain-C CPU restatement ("port") of the Scythe.jl spectral-transform time-stepping hot path.
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library.
 *
 * It applies, with ordinary loops (OpenMP over rings / spectral blocks / columns - the axes the reference
 * threads over, src/semiimplicit.jl:309), operators given as plain arrays (basis tables, Cholesky factors, Chebyshev
 * matrices: built by scythe_oracle_ops.c, or - for the cross-check - taken from the dense definitions of
 * oracle/oracle_np.py), in the reference's own array layouts: physical[point, var, deriv] and spectral[index, var],
 * column-major as in Julia.
 *
 *   orc_forward        spectralTransform!(tile)   src/semiimplicit.jl:734   (Springsteel, external)
 *   orc_spline_solve   splineTransform!           src/semiimplicit.jl:285   (Springsteel, external)
 *   orc_inverse        tileTransform!             src/semiimplicit.jl:305   (Springsteel, external)
 *   orc_tendency_step  equation set + explicit_timestep + semiimplicit_adjustment
 *                      src/testModels.jl:1-98, src/shallowWaterModels.jl:1-233,346-511,
 *                      src/semiimplicit.jl:672-698, 521-597
 *
 * Parity status: R-grid B-spline/AB3 path pinned by the notebook KAT (tests/golden); everything else
 * "parity unpinned" (see oracle/oracle_np.py header).
 */
#ifndef _GNU_SOURCE
#define _GNU_SOURCE   /* M_PIl */
#endif
#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

typedef struct {
    int has_l, has_z;
    int V, D;
    int b_rDim;            /* patch nodes */
    int cell0, ncells;     /* tile */
    int nrings;            /* 3*ncells */
    int zDim, b_zDim;
    int K2;                /* patch blocks 1+2*kDim */
    int K2t;               /* tile blocks */
    long N;                /* tile points (incl. z) */
    const int *L;          /* [nrings] */
    const int *kmax;       /* [nrings] */
    const double *off;     /* [nrings] */
    const long *pstart;    /* [nrings] first horizontal point of ring, tile-relative */
    const double *phi;     /* [3][nrings][4] */
    const int *m0;         /* [nrings] patch node array index of first non-zero basis function */
    const double *wq;      /* [nrings] */
    const double *Mz;      /* [V][3][zDim][b_zDim] */
    const double *CBz;     /* [b_zDim][zDim] */
    int slot[7];           /* u r rr l ll z zz -> slot index or -1 */
    /* spline solve classes */
    int nclass;
    const int *cls;        /* [V][2]: class for k = 0, k >= 1 */
    const int *nfree;      /* [nclass] */
    const int *periodic;   /* [nclass] */
    const int *rl;         /* [nclass] left rank */
    const int *rr;         /* [nclass] right rank */
    const double *gl;      /* [nclass][3][2] dependent-left coefficient rows */
    const double *gr;      /* [nclass][3][2] */
    const double *Lband;   /* [nclass][b_rDim][4]   L[i][i-3..i] */
    const double *Larrow;  /* [nclass][3][b_rDim]   rows nfree-3..nfree-1 (periodic only) */
} orc_grid;

void orc_set_num_threads(int n) {
#ifdef _OPENMP
    if (n > 0) omp_set_num_threads(n);
#else
    (void)n;
#endif
}

int orc_num_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

/* ------------------------------------------------------------------ small complex FFT (power of two) */
static void fft_pow2(double *re, double *im, int n, int sign, const double *tc, const double *ts) {
    /* in-place iterative radix-2; tc/ts = cos/sin(2 pi j / n), j < n/2 */
    int j = 0;
    for (int i = 1; i < n; i++) {
        int bit = n >> 1;
        for (; j & bit; bit >>= 1) j ^= bit;
        j ^= bit;
        if (i < j) {
            double t = re[i]; re[i] = re[j]; re[j] = t;
            t = im[i]; im[i] = im[j]; im[j] = t;
        }
    }
    for (int len = 2; len <= n; len <<= 1) {
        int step = n / len;
        for (int i = 0; i < n; i += len) {
            for (int k = 0; k < len / 2; k++) {
                double wr = tc[k * step], wi = sign * ts[k * step];
                int a = i + k, b = i + k + len / 2;
                double xr = re[b] * wr - im[b] * wi, xi = re[b] * wi + im[b] * wr;
                re[b] = re[a] - xr; im[b] = im[a] - xi;
                re[a] += xr; im[a] += xi;
            }
        }
    }
}

static int is_pow2(int n) { return n >= 4 && (n & (n - 1)) == 0; }

typedef struct { int L; double *c, *s; } trig_t;   /* cos/sin(2 pi j / L), j < L */

static void trig_init(trig_t *t, int L) {
    t->L = L;
    t->c = (double *)malloc(sizeof(double) * L);
    t->s = (double *)malloc(sizeof(double) * L);
    for (int j = 0; j < L; j++) {
        t->c[j] = (double)cosl(2.0L * M_PIl * j / L);
        t->s[j] = (double)sinl(2.0L * M_PIl * j / L);
    }
}
static void trig_free(trig_t *t) { free(t->c); free(t->s); }

/* forward ring transform of nline real lines x[line][l] (stride L) -> c[line][blk] (stride cs), blocks 0..2*kmax */
static void ring_forward(const trig_t *tg, int L, int kmax, double off, const double *x, int nline, double *c, int cs,
                         double *wr, double *wi) {
    double inv = 1.0 / L;
    if (is_pow2(L)) {
        for (int ln = 0; ln < nline; ln += 2) {
            const double *x0 = x + (long)ln * L;
            const double *x1 = (ln + 1 < nline) ? x + (long)(ln + 1) * L : NULL;
            for (int l = 0; l < L; l++) { wr[l] = x0[l]; wi[l] = x1 ? x1[l] : 0.0; }
            fft_pow2(wr, wi, L, -1, tg->c, tg->s);
            for (int k = 0; k <= kmax; k++) {
                int nk = (L - k) % L;
                /* X0 = (Z_k + conj Z_-k)/2 ; X1 = (Z_k - conj Z_-k)/(2i) */
                double ar = 0.5 * (wr[k] + wr[nk]), ai = 0.5 * (wi[k] - wi[nk]);
                double br = 0.5 * (wi[k] + wi[nk]), bi = -0.5 * (wr[k] - wr[nk]);
                double pr = (double)cosl((long double)k * (long double)off), pi = -(double)sinl((long double)k * (long double)off);   /* k * off formed in extended precision: the Float64 product alone loses k ulp */
                double *c0 = c + (long)ln * cs, *c1 = c + (long)(ln + 1) * cs;
                if (k == 0) {
                    c0[0] = ar * inv;
                    if (x1) c1[0] = br * inv;
                } else {
                    c0[2 * k - 1] = (ar * pr - ai * pi) * inv;
                    c0[2 * k] = (ar * pi + ai * pr) * inv;
                    if (x1) {
                        c1[2 * k - 1] = (br * pr - bi * pi) * inv;
                        c1[2 * k] = (br * pi + bi * pr) * inv;
                    }
                }
            }
        }
    } else {
        for (int ln = 0; ln < nline; ln++) {
            const double *xl = x + (long)ln * L;
            double *cl = c + (long)ln * cs;
            for (int k = 0; k <= kmax; k++) {
                double sr = 0.0, si = 0.0;
                for (int l = 0; l < L; l++) {
                    int j = (int)(((long)l * k) % L);
                    sr += xl[l] * tg->c[j];
                    si -= xl[l] * tg->s[j];
                }
                double pr = (double)cosl((long double)k * (long double)off), pi = -(double)sinl((long double)k * (long double)off);   /* k * off formed in extended precision: the Float64 product alone loses k ulp */
                if (k == 0) cl[0] = sr * inv;
                else {
                    cl[2 * k - 1] = (sr * pr - si * pi) * inv;
                    cl[2 * k] = (sr * pi + si * pr) * inv;
                }
            }
        }
    }
}

/* inverse ring transform with lambda-derivative order ld: c[line][blk] -> x[line][l] */
static void ring_inverse(const trig_t *tg, int L, int kmax, double off, const double *c, int cs, int nline, int ld,
                         double *x, double *wr, double *wi) {
    if (is_pow2(L)) {
        for (int ln = 0; ln < nline; ln += 2) {
            int two = (ln + 1 < nline);
            for (int l = 0; l < L; l++) { wr[l] = 0.0; wi[l] = 0.0; }
            for (int q = 0; q < 1 + two; q++) {
                const double *cl = c + (long)(ln + q) * cs;
                for (int k = 0; k <= kmax; k++) {
                    double cr = (k == 0) ? cl[0] : cl[2 * k - 1], ci = (k == 0) ? 0.0 : cl[2 * k];
                    if (ld == 1) { double t = cr; cr = -k * ci; ci = k * t; }
                    else if (ld == 2) { cr *= -(double)k * k; ci *= -(double)k * k; }
                    /* undo the phase reference: multiply by e^{+ik off} */
                    double pr = (double)cosl((long double)k * (long double)off), pi = (double)sinl((long double)k * (long double)off);
                    double zr = cr * pr - ci * pi, zi = cr * pi + ci * pr;
                    /* Hermitian extension; sequence q goes into real (q=0) or imaginary (q=1) part */
                    int nk = (L - k) % L;
                    if (q == 0) {
                        wr[k] += zr; wi[k] += zi;
                        if (k > 0) { wr[nk] += zr; wi[nk] -= zi; }
                    } else {
                        wr[k] -= zi; wi[k] += zr;
                        if (k > 0) { wr[nk] += zi; wi[nk] += zr; }
                    }
                }
            }
            fft_pow2(wr, wi, L, +1, tg->c, tg->s);
            for (int l = 0; l < L; l++) {
                x[(long)ln * L + l] = wr[l];
                if (two) x[(long)(ln + 1) * L + l] = wi[l];
            }
        }
    } else {
        for (int ln = 0; ln < nline; ln++) {
            const double *cl = c + (long)ln * cs;
            double *xl = x + (long)ln * L;
            for (int l = 0; l < L; l++) xl[l] = (ld == 0) ? cl[0] : 0.0;
            for (int k = 1; k <= kmax; k++) {
                double cr = cl[2 * k - 1], ci = cl[2 * k];
                if (ld == 1) { double t = cr; cr = -k * ci; ci = k * t; }
                else if (ld == 2) { cr *= -(double)k * k; ci *= -(double)k * k; }
                double pr = (double)cosl((long double)k * (long double)off), pi = (double)sinl((long double)k * (long double)off);
                double zr = 2.0 * (cr * pr - ci * pi), zi = 2.0 * (cr * pi + ci * pr);
                for (int l = 0; l < L; l++) {
                    int j = (int)(((long)l * k) % L);
                    xl[l] += zr * tg->c[j] - zi * tg->s[j];
                }
            }
        }
    }
}

/* ------------------------------------------------------------------ forward transform */
void orc_forward(const orc_grid *g, const double *values, double *btile) {
    const int nbt = g->ncells + 3, Zb = g->b_zDim, nz = g->zDim, K2t = g->K2t;
    const long S_t = (long)Zb * K2t * nbt;
    double *F = (double *)calloc((size_t)g->nrings * Zb * K2t, sizeof(double));
    for (int v = 0; v < g->V; v++) {
        const double *val = values + (long)v * g->N;
        memset(F, 0, sizeof(double) * (size_t)g->nrings * Zb * K2t);
#pragma omp parallel
        {
            int Lmax = 1;
            for (int i = 0; i < g->nrings; i++) if (g->L[i] > Lmax) Lmax = g->L[i];
            double *bz = (double *)malloc(sizeof(double) * (size_t)Zb * Lmax);
            double *wr = (double *)malloc(sizeof(double) * Lmax), *wi = (double *)malloc(sizeof(double) * Lmax);
            trig_t tg; tg.L = 0; tg.c = tg.s = NULL;
#pragma omp for schedule(dynamic, 1)
            for (int i = 0; i < g->nrings; i++) {
                int L = g->L[i];
                const double *u = val + g->pstart[i] * nz;          /* [l][z] */
                if (g->has_z) {
                    for (int zm = 0; zm < Zb; zm++)
                        for (int l = 0; l < L; l++) {
                            double s = 0.0;
                            const double *cb = g->CBz + (long)zm * nz, *ul = u + (long)l * nz;
                            for (int z = 0; z < nz; z++) s += cb[z] * ul[z];
                            bz[(long)zm * L + l] = s;
                        }
                } else {
                    for (int l = 0; l < L; l++) bz[l] = u[l];
                }
                double *Fi = F + (long)i * Zb * K2t;
                if (g->has_l) {
                    if (tg.L != L) { if (tg.c) trig_free(&tg); trig_init(&tg, L); }
                    ring_forward(&tg, L, g->kmax[i], g->off[i], bz, Zb, Fi, K2t, wr, wi);
                } else {
                    for (int zm = 0; zm < Zb; zm++) Fi[(long)zm * K2t] = bz[zm];
                }
            }
            if (tg.c) trig_free(&tg);
            free(bz); free(wr); free(wi);
        }
        double *bv = btile + (long)v * S_t;
#pragma omp parallel for collapse(2) schedule(static)
        for (int zm = 0; zm < Zb; zm++)
            for (int blk = 0; blk < K2t; blk++) {
                double *b = bv + ((long)zm * K2t + blk) * nbt;
                for (int j = 0; j < nbt; j++) b[j] = 0.0;
                for (int i = 0; i < g->nrings; i++) {
                    double f = g->wq[i] * F[((long)i * Zb + zm) * K2t + blk];
                    const double *ph = g->phi + (long)i * 4;        /* d = 0 */
                    int j0 = g->m0[i] - g->cell0;
                    for (int jj = 0; jj < 4; jj++) b[j0 + jj] += ph[jj] * f;
                }
            }
    }
    free(F);
}

/* shared[patch] += tile (src/semiimplicit.jl:320-329) */
void orc_add_tile(const orc_grid *g, const double *btile, double *shared) {
    const int nbt = g->ncells + 3, Zb = g->b_zDim;
    const long S_t = (long)Zb * g->K2t * nbt, S_p = (long)Zb * g->K2 * g->b_rDim;
    for (int v = 0; v < g->V; v++)
        for (int zm = 0; zm < Zb; zm++)
            for (int blk = 0; blk < g->K2t; blk++) {
                const double *b = btile + (long)v * S_t + ((long)zm * g->K2t + blk) * nbt;
                double *s = shared + (long)v * S_p + ((long)zm * g->K2 + blk) * g->b_rDim + g->cell0;
                for (int j = 0; j < nbt; j++) s[j] += b[j];
            }
}

/* ------------------------------------------------------------------ B -> A */
static void solve_one(const orc_grid *g, int c, const double *b, double *a, double *y) {
    const int n = g->nfree[c], nb = g->b_rDim;
    const double *Lb = g->Lband + (long)c * nb * 4;
    const double *La = g->Larrow + (long)c * 3 * nb;
    const double *gl = g->gl + (long)c * 6, *gr = g->gr + (long)c * 6;
    const int rl = g->rl[c], rr = g->rr[c], per = g->periodic[c];
    /* y = Gamma b */
    if (per) {
        for (int j = 0; j < n; j++) y[j] = 0.0;
        for (int m = 0; m < nb; m++) y[(m - 1 + n) % n] += b[m];
    } else {
        for (int j = 0; j < n; j++) y[j] = b[rl + j];
        for (int i = 0; i < rl; i++) { y[0] += gl[i * 2] * b[i]; y[1] += gl[i * 2 + 1] * b[i]; }
        for (int i = 0; i < rr; i++) { y[n - 1] += gr[i * 2] * b[nb - 1 - i]; y[n - 2] += gr[i * 2 + 1] * b[nb - 1 - i]; }
    }
    /* forward substitution L y' = y */
    int nband = per ? n - 3 : n;
    for (int i = 0; i < nband; i++) {
        double s = y[i];
        for (int q = 1; q <= 3; q++) if (i - q >= 0) s -= Lb[i * 4 + (3 - q)] * y[i - q];
        y[i] = s / Lb[i * 4 + 3];
    }
    if (per) {
        for (int r = 0; r < 3; r++) {
            int i = n - 3 + r;
            double s = y[i];
            const double *row = La + (long)r * nb;
            for (int k = 0; k < i; k++) s -= row[k] * y[k];
            y[i] = s / row[i];
        }
        /* back substitution L^T x = y */
        for (int r = 2; r >= 0; r--) {
            int i = n - 3 + r;
            double s = y[i];
            for (int r2 = r + 1; r2 < 3; r2++) s -= La[(long)r2 * nb + i] * y[n - 3 + r2];
            y[i] = s / La[(long)r * nb + i];
        }
        for (int i = n - 4; i >= 0; i--) {
            double s = y[i];
            for (int q = 1; q <= 3; q++) if (i + q < n - 3) s -= Lb[(i + q) * 4 + (3 - q)] * y[i + q];
            for (int r = 0; r < 3; r++) s -= La[(long)r * nb + i] * y[n - 3 + r];
            y[i] = s / Lb[i * 4 + 3];
        }
        for (int m = 0; m < nb; m++) a[m] = y[(m - 1 + n) % n];
    } else {
        for (int i = n - 1; i >= 0; i--) {
            double s = y[i];
            for (int q = 1; q <= 3; q++) if (i + q < n) s -= Lb[(i + q) * 4 + (3 - q)] * y[i + q];
            y[i] = s / Lb[i * 4 + 3];
        }
        for (int j = 0; j < n; j++) a[rl + j] = y[j];
        for (int i = 0; i < rl; i++) a[i] = gl[i * 2] * y[0] + gl[i * 2 + 1] * y[1];
        for (int i = 0; i < rr; i++) a[nb - 1 - i] = gr[i * 2] * y[n - 1] + gr[i * 2 + 1] * y[n - 2];
    }
}

void orc_spline_solve(const orc_grid *g, const double *shared, double *A) {
    const int nb = g->b_rDim;
    const long ncol = (long)g->b_zDim * g->K2;
#pragma omp parallel
    {
        double *y = (double *)malloc(sizeof(double) * (nb + 4));
#pragma omp for collapse(2) schedule(static)
        for (int v = 0; v < g->V; v++)
            for (long col = 0; col < ncol; col++) {
                int blk = (int)(col % g->K2);
                int c = g->cls[v * 2 + (blk == 0 ? 0 : 1)];
                long o = ((long)v * ncol + col) * nb;
                solve_one(g, c, shared + o, A + o, y);
            }
        free(y);
    }
}

/* ------------------------------------------------------------------ inverse transform */
void orc_inverse(const orc_grid *g, const double *A, double *phys) {
    const int Zb = g->b_zDim, nz = g->zDim, K2 = g->K2, nb = g->b_rDim;
    const long S_p = (long)Zb * K2 * nb, N = g->N;
    for (int v = 0; v < g->V; v++) {
        const double *Av = A + (long)v * S_p;
#pragma omp parallel
        {
            int Lmax = 1;
            for (int i = 0; i < g->nrings; i++) if (g->L[i] > Lmax) Lmax = g->L[i];
            double *R = (double *)malloc(sizeof(double) * (size_t)3 * Zb * K2);
            double *f = (double *)malloc(sizeof(double) * (size_t)Zb * Lmax);
            double *wr = (double *)malloc(sizeof(double) * Lmax), *wi = (double *)malloc(sizeof(double) * Lmax);
            trig_t tg; tg.L = 0; tg.c = tg.s = NULL;
#pragma omp for schedule(dynamic, 1)
            for (int i = 0; i < g->nrings; i++) {
                int L = g->L[i], nblk = 1 + 2 * g->kmax[i];
                if (!g->has_l) nblk = 1;
                /* radial evaluation: R[d][zm][blk] */
                for (int d = 0; d < 3; d++) {
                    const double *ph = g->phi + ((long)d * g->nrings + i) * 4;
                    for (int zm = 0; zm < Zb; zm++)
                        for (int blk = 0; blk < nblk; blk++) {
                            const double *a = Av + ((long)zm * K2 + blk) * nb + g->m0[i];
                            R[((long)d * Zb + zm) * K2 + blk] = ph[0] * a[0] + ph[1] * a[1] + ph[2] * a[2] + ph[3] * a[3];
                        }
                }
                if (g->has_l && tg.L != L) { if (tg.c) trig_free(&tg); trig_init(&tg, L); }
                /* (radial d, lambda ld) -> slot */
                static const int comb[5][3] = {{0, 0, 0}, {1, 0, 1}, {2, 0, 2}, {0, 1, 3}, {0, 2, 4}};
                for (int q = 0; q < 5; q++) {
                    int d = comb[q][0], ld = comb[q][1], sl = g->slot[comb[q][2]];
                    if (sl < 0) continue;
                    if (g->has_l) ring_inverse(&tg, L, g->kmax[i], g->off[i], R + (long)d * Zb * K2, K2, Zb, ld, f, wr, wi);
                    else for (int zm = 0; zm < Zb; zm++) f[zm] = R[((long)d * Zb + zm) * K2];
                    /* vertical */
                    int nzs = (q == 0 && g->has_z) ? 3 : 1;
                    for (int zs = 0; zs < nzs; zs++) {
                        int slot = (zs == 0) ? sl : g->slot[4 + zs];
                        double *out = phys + ((long)slot * g->V + v) * N + g->pstart[i] * nz;
                        if (g->has_z) {
                            const double *M = g->Mz + (((long)v * 3 + zs) * nz) * Zb;
                            for (int l = 0; l < L; l++)
                                for (int z = 0; z < nz; z++) {
                                    double s = 0.0;
                                    for (int zm = 0; zm < Zb; zm++) s += M[(long)z * Zb + zm] * f[(long)zm * L + l];
                                    out[(long)l * nz + z] = s;
                                }
                        } else {
                            for (int l = 0; l < L; l++) out[l] = f[l];
                        }
                    }
                }
            }
            if (tg.c) trig_free(&tg);
            free(R); free(f); free(wr); free(wi);
        }
    }
}

/* ------------------------------------------------------------------ equation sets + time stepping */
enum { EQ_LINADV_1D = 0, EQ_LINADV_RZ = 1, EQ_LINADV_RL = 2, EQ_LINADV_RLZ = 3, EQ_ONEWAY_SLAB = 4,
       EQ_TWOWAY_SLAB = 5, EQ_ONEWAY_HRBL = 6, EQ_LINACOUSTIC_RZ = 7 };
/* params: [g, K, Cd, Hfree, Hb, f, S1, c_0, Kh, Um, Vm, Pxi_bar] */
enum { P_G = 0, P_K, P_CD, P_HFREE, P_HB, P_F, P_S1, P_C0, P_KH, P_UM, P_VM, P_PXI };

typedef struct {
    int eq, t, semiimplicit;
    double ts;
    const double *par;
    const double *r, *lam, *z;      /* gridpoint coordinates [N] (lam/z may be NULL) */
    const double *Mint;             /* [zDim][zDim]  column -> integral from bottom   (HRBL) */
    const double *Mdz;              /* [zDim][zDim]  column -> truncated d/dz          (HRBL, semi-implicit xi) */
    const double *Mrec;             /* [zDim][zDim]  column -> truncated reconstruction (semi-implicit xi) */
    const double *Wmat, *Xmat;      /* [zDim][zDim]  g -> w, g -> dw/dz for the current tau (semi-implicit) */
    int w_index, xi_index;          /* 0-based */
    double tau;
} orc_step;

#define PH(v, s) (phys + ((long)g->slot[s] * g->V + (v)) * N)
enum { S_U = 0, S_R, S_RR, S_L, S_LL, S_Z, S_ZZ };

static void matvec(const double *M, int n, const double *x, double *y) {
    for (int i = 0; i < n; i++) {
        double s = 0.0;
        for (int j = 0; j < n; j++) s += M[(long)i * n + j] * x[j];
        y[i] = s;
    }
}

void orc_tendency_step(const orc_grid *g, const orc_step *st, double *phys, double *expdot, double *e1, double *e2,
                       double *impdot, double *i1, double *i2, double *var_np1) {
    const long N = g->N;
    const double *par = st->par, *r = st->r;
    const int V = g->V, nz = g->zDim;
    double *E[8];
    for (int v = 0; v < V; v++) E[v] = expdot + (long)v * N;
    switch (st->eq) {
    case EQ_LINADV_1D: {
        const double *ur = PH(0, S_R), *urr = PH(0, S_RR);
#pragma omp parallel for
        for (long p = 0; p < N; p++) E[0][p] = -(par[P_C0] * ur[p]) + (par[P_K] * urr[p]);
    } break;
    case EQ_LINADV_RZ: {
        const double *hr = PH(0, S_R), *hrr = PH(0, S_RR), *hz = PH(0, S_Z), *hzz = PH(0, S_ZZ), *u = PH(1, S_U), *w = PH(3, S_U);
#pragma omp parallel for
        for (long p = 0; p < N; p++)
            E[0][p] = (-u[p] * hr[p]) + (-w[p] * hz[p]) + (par[P_K] * ((hr[p] / r[p]) + hrr[p] + hzz[p]));
    } break;
    case EQ_LINADV_RL:
    case EQ_LINADV_RLZ: {
        const double *hr = PH(0, S_R), *hrr = PH(0, S_RR), *hl = PH(0, S_L), *hll = PH(0, S_LL), *u = PH(1, S_U), *vv = PH(2, S_U);
        int diff = (st->eq == EQ_LINADV_RLZ) || par[P_K] > 0.0;
#pragma omp parallel for
        for (long p = 0; p < N; p++) {
            double e = (-u[p] * hr[p]) - (vv[p] * (hl[p] / r[p]));
            if (diff) e += par[P_K] * ((hr[p] / r[p]) + hrr[p] + (hll[p] / (r[p] * r[p])));
            E[0][p] = e;
        }
    } break;
    case EQ_ONEWAY_SLAB:
    case EQ_TWOWAY_SLAB: {
        const double G = par[P_G], K = par[P_K], Cd = par[P_CD], Hfree = par[P_HFREE], Hb = par[P_HB], f = par[P_F], S1 = par[P_S1];
        const double *h = PH(0, S_U), *hr = PH(0, S_R), *hl = PH(0, S_L);
        const double *ug = PH(1, S_U), *ugr = PH(1, S_R), *ugl = PH(1, S_L);
        const double *vg = PH(2, S_U), *vgr = PH(2, S_R), *vgl = PH(2, S_L);
        const double *ub = PH(3, S_U), *ubr = PH(3, S_R), *ubrr = PH(3, S_RR), *ubl = PH(3, S_L), *ubll = PH(3, S_LL);
        const double *vb = PH(4, S_U), *vbr = PH(4, S_R), *vbrr = PH(4, S_RR), *vbl = PH(4, S_L), *vbll = PH(4, S_LL);
        double *wv = phys + ((long)g->slot[S_U] * V + 5) * N;
        int two = (st->eq == EQ_TWOWAY_SLAB);
#pragma omp parallel for
        for (long p = 0; p < N; p++) {
            double rp = r[p];
            double U = 0.78 * sqrt((ub[p] * ub[p]) + (vb[p] * vb[p]));
            double w = -Hb * ((ub[p] / rp) + ubr[p] + (vbl[p] / rp));
            wv[p] = w;
            double w_ = 0.5 * fabs(w) - w;
            E[5][p] = 0.0;
            double e0 = ((-vg[p] * hl[p] / rp) + (-ug[p] * hr[p])) + (-(Hfree + h[p]) * ((ug[p] / rp) + ugr[p] + (vgl[p] / rp)));
            if (two) e0 += -(Hfree + h[p]) * w * S1;
            E[0][p] = e0;
            E[1][p] = ((-vg[p] * ugl[p] / rp) + (-ug[p] * ugr[p])) + (-G * hr[p]) + (vg[p] * (f + (vg[p] / rp)));
            E[2][p] = ((-vg[p] * vgl[p] / rp) + (-ug[p] * vgr[p])) + (-G * (hl[p] / rp)) + (-ug[p] * (f + (vg[p] / rp)));
            E[3][p] = ((-vb[p] * ubl[p] / rp) + (-ub[p] * ubr[p])) + (-G * hr[p]) + (vb[p] * (f + (vb[p] / rp))) + (-(Cd * U * ub[p] / Hb))
                + (w_ * (ug[p] - ub[p]) / Hb)
                + (K * ((ubr[p] / rp) + ubrr[p] - (ub[p] / (rp * rp)) + (ubll[p] / (rp * rp)) - (2.0 * vbl[p] / (rp * rp))));
            E[4][p] = ((-vb[p] * vbl[p] / rp) + (-ub[p] * vbr[p])) + (-G * (hl[p] / rp)) + (-ub[p] * (f + (vb[p] / rp))) + (-(Cd * U * vb[p] / Hb))
                + (w_ * (vg[p] - vb[p]) / Hb)
                + (K * ((vbr[p] / rp) + vbrr[p] - (vb[p] / (rp * rp)) + (vbll[p] / (rp * rp)) + (2.0 * ubl[p] / (rp * rp))));
        }
    } break;
    case EQ_ONEWAY_HRBL: {
        const double G = par[P_G], Kh = par[P_KH], Cd0 = par[P_CD], Hfree = par[P_HFREE], f = par[P_F], Um = par[P_UM], Vm = par[P_VM];
        const double *h = PH(0, S_U), *hr = PH(0, S_R), *hl = PH(0, S_L);
        const double *ug = PH(1, S_U), *ugr = PH(1, S_R), *ugl = PH(1, S_L);
        const double *vg = PH(2, S_U), *vgr = PH(2, S_R), *vgl = PH(2, S_L);
        const double *ub = PH(3, S_U), *ubr = PH(3, S_R), *ubrr = PH(3, S_RR), *ubl = PH(3, S_L), *ubll = PH(3, S_LL), *ubz = PH(3, S_Z);
        const double *vb = PH(4, S_U), *vbr = PH(4, S_R), *vbrr = PH(4, S_RR), *vbl = PH(4, S_L), *vbll = PH(4, S_LL), *vbz = PH(4, S_Z);
        double *wv = phys + ((long)g->slot[S_U] * V + 5) * N;
        long ncol = N / nz;
#pragma omp parallel
        {
            double *div = (double *)malloc(sizeof(double) * nz * 6);
            double *wb = div + nz, *fu = div + 2 * nz, *fv = div + 3 * nz, *vdu = div + 4 * nz, *vdv = div + 5 * nz;
#pragma omp for schedule(static)
            for (long c = 0; c < ncol; c++) {
                long p0 = c * nz;
                for (int k = 0; k < nz; k++) {
                    long p = p0 + k;
                    div[k] = -((ub[p] / r[p]) + ubr[p] + (vbl[p] / r[p]));
                    double S = sqrt((ubz[p] * ubz[p]) + (vbz[p] * vbz[p]));
                    double l = 1.0 / ((1.0 / (0.4 * st->z[p])) + (1.0 / 80.0));
                    double Kv = (l * l) * S;
                    fu[k] = Kv * ubz[p];
                    fv[k] = Kv * vbz[p];
                }
                matvec(st->Mint, nz, div, wb);
                double sfcu = (Um * cos(st->lam[p0])) + (Vm * sin(st->lam[p0]));
                double sfcv = (Vm * cos(st->lam[p0])) - (Um * sin(st->lam[p0]));
                double u10 = ub[p0 + 1] + sfcu, v10 = vb[p0 + 1] + sfcv;
                double U10 = sqrt(u10 * u10 + v10 * v10);
                double Cd = Cd0;
                if (U10 < 5.2) Cd = 1.0e-3;
                else if (U10 < 33.6) Cd = 4.4e-4 * pow(U10, 0.5);
                fu[0] = Cd * U10 * u10;
                fv[0] = Cd * U10 * v10;
                matvec(st->Mdz, nz, fu, vdu);
                matvec(st->Mdz, nz, fv, vdv);
                for (int k = 0; k < nz; k++) {
                    long p = p0 + k;
                    double rp = r[p];
                    wv[p] = wb[k];
                    E[5][p] = 0.0;
                    E[0][p] = ((-vg[p] * hl[p] / rp) + (-ug[p] * hr[p])) + (-(Hfree + h[p]) * ((ug[p] / rp) + ugr[p] + (vgl[p] / rp)));
                    E[1][p] = ((-vg[p] * ugl[p] / rp) + (-ug[p] * ugr[p])) + (-G * hr[p]) + (vg[p] * (f + (vg[p] / rp)));
                    E[2][p] = ((-vg[p] * vgl[p] / rp) + (-ug[p] * vgr[p])) + (-G * (hl[p] / rp)) + (-ug[p] * (f + (vg[p] / rp)));
                    E[3][p] = ((-vb[p] * ubl[p] / rp) + (-ub[p] * ubr[p]) + (-wb[k] * ubz[p])) + (-G * hr[p]) + (vb[p] * (f + (vb[p] / rp))) + vdu[k]
                        + (Kh * ((ubr[p] / rp) + ubrr[p] - (ub[p] / (rp * rp)) + (ubll[p] / (rp * rp)) - (2.0 * vbl[p] / (rp * rp))));
                    E[4][p] = ((-vb[p] * vbl[p] / rp) + (-ub[p] * vbr[p]) + (-wb[k] * vbz[p])) + (-G * (hl[p] / rp)) + (-ub[p] * (f + (vb[p] / rp))) + vdv[k]
                        + (Kh * ((vbr[p] / rp) + vbrr[p] - (vb[p] / (rp * rp)) + (vbll[p] / (rp * rp)) + (2.0 * ubl[p] / (rp * rp))));
                }
            }
            free(div);
        }
    } break;
    case EQ_LINACOUSTIC_RZ: {
        const double K = par[P_K], pxi = par[P_PXI];
        const double *u = PH(3, S_U), *w = PH(4, S_U);
        double *I[5];
        for (int v = 0; v < 5; v++) I[v] = impdot + (long)v * N;
#pragma omp parallel for
        for (long p = 0; p < N; p++) {
#define ADVT(v) ((-u[p] * PH(v, S_R)[p]) + (-w[p] * PH(v, S_Z)[p]))
#define DIFT(v) (K * (PH(v, S_RR)[p] + PH(v, S_ZZ)[p]))
            E[0][p] = ADVT(0) + DIFT(0);
            E[1][p] = ADVT(1) - PH(3, S_R)[p] - PH(4, S_Z)[p];
            E[2][p] = ADVT(2) + DIFT(2);
            E[3][p] = ADVT(3) + (-(pxi * PH(1, S_R)[p])) + DIFT(3);
            E[4][p] = ADVT(4) + (-(pxi * PH(1, S_Z)[p])) + DIFT(4);
            I[0][p] = 0.0; I[2][p] = 0.0; I[3][p] = 0.0;
            I[1][p] = -PH(4, S_Z)[p];
            I[4][p] = -(pxi * PH(1, S_Z)[p]);
        }
    } break;
    default: break;
    }
    /* explicit_timestep (src/semiimplicit.jl:672-698) */
    const double ts = st->ts;
    const int t = st->t;
    for (int v = 0; v < V; v++) {
        const double *u = phys + ((long)g->slot[S_U] * V + v) * N;
        double *en = expdot + (long)v * N, *a1 = e1 + (long)v * N, *a2 = e2 + (long)v * N, *un = var_np1 + (long)v * N;
#pragma omp parallel for
        for (long p = 0; p < N; p++) {
            if (t == 1) {
                un[p] = u[p] + (ts * en[p]);
                a1[p] = en[p];
            } else if (t == 2) {
                un[p] = u[p] + (0.5 * ts) * ((3.0 * en[p]) - a1[p]);
                a2[p] = a1[p];
                a1[p] = en[p];
            } else {
                un[p] = u[p] + ((ts / 12.0) * ((23.0 * en[p]) - (16.0 * a1[p]) + (5.0 * a2[p])));
                a2[p] = a1[p];
                a1[p] = en[p];
            }
        }
    }
    if (!st->semiimplicit) return;
    /* semiimplicit_adjustment (src/semiimplicit.jl:521-597) */
    const int wi = st->w_index, xi = st->xi_index;
    const double tau = st->tau, pxi = par[P_PXI];
    long ncol = N / nz;
#pragma omp parallel
    {
        double *buf = (double *)malloc(sizeof(double) * nz * 8);
        double *ws = buf, *xs = buf + nz, *xrec = buf + 2 * nz, *xz = buf + 3 * nz, *gv = buf + 4 * nz, *wn = buf + 5 * nz, *wz = buf + 6 * nz;
#pragma omp for schedule(static)
        for (long c = 0; c < ncol; c++) {
            for (int k = 0; k < nz; k++) {
                long p = c * nz + k;
                double out[2];
                int vv[2] = {wi, xi};
                for (int q = 0; q < 2; q++) {
                    long o = (long)vv[q] * N + p;
                    double x = var_np1[o], In = impdot[o], I1 = i1[o], I2 = i2[o];
                    if (t == 1) x = x - (ts * In) + (ts * 0.5 * In);
                    else if (t == 2) x = x - (0.5 * ts) * ((3.0 * In) - I1) - (ts * In) + (ts * 0.75 * I1);
                    else x = x - ((ts / 12.0) * ((23.0 * In) - (16.0 * I1) + (5.0 * I2))) - (ts * In) + (ts * 0.75 * I1);
                    out[q] = x;
                    i2[o] = I1;
                    i1[o] = In;
                }
                ws[k] = out[0];
                xs[k] = out[1];
            }
            matvec(st->Mrec, nz, xs, xrec);
            matvec(st->Mdz, nz, xs, xz);
            gv[0] = 0.0; gv[1] = 0.0;
            for (int k = 1; k < nz - 1; k++) gv[k + 1] = (tau * pxi * xz[k]) - ws[k];
            matvec(st->Wmat, nz, gv, wn);
            matvec(st->Xmat, nz, gv, wz);
            for (int k = 0; k < nz; k++) {
                long p = c * nz + k;
                var_np1[(long)wi * N + p] = wn[k];
                var_np1[(long)xi * N + p] = xrec[k] - (tau * wz[k]);
            }
        }
        free(buf);
    }
}
