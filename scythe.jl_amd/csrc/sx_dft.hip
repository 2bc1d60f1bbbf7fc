// Azimuthal transforms for ring lengths that are not powers of two - Springsteel's native ragged rings
// (ring ri has 4 + 4 ri points and keeps wavenumbers 0..ri, i.e. only about L/4 of them) - as dense truncated DFTs on the
// f64 matrix cores.  The spectrum is short (2 kmax + 1 ~ L/2 real coefficients), so the DFT of one ring is a genuine
// [L x (2 kmax + 1)] x [(2 kmax + 1) x columns] product with columns = vertical levels x derivative planes; an FFT of
// these lengths (4 * anything, including primes) would need Bluestein with three padded transforms per line.
// v_mfma_f64_16x16x4_f64 runs at the vector fp64 rate on MI355X, what it buys here is that one generated twiddle tile
// serves 16 levels (and up to three derivative planes) instead of one fused multiply-add.
//
//   inverse  x[l][z] = sum_j T[l][j] C[j][z]      T[l][0] = 1, T[l][2k-1] = cos(2 pi k l / L), T[l][2k] = -sin(...)
//            C = radial evaluation of Az (4 nodes), phase reference, factor 2 (same staging as k_rl_inverse)
//            d/dlambda and d2/dlambda2 use the same T with C' = i k C and C'' = -k^2 C formed while loading B
//   forward  F[z][j] = (1/L) sum_l X[z][l] T'[l][j]   T'[l][2k] = cos, T'[l][2k+1] = -sin  (blk indexing of Fl)
//
// Rings with fewer than 8 levels (RL grids) stay on the scalar kernels of sx_kernels.hip: the MFMA N dimension is the
// vertical level.
#include "sx_internal.hpp"
#include <algorithm>
#include <cstdlib>

namespace sx {

#define HIPCHK3(x)                                                                                  \
    do {                                                                                            \
        hipError_t e_ = (x);                                                                        \
        if (e_ != hipSuccess) set_error(std::string(#x) + ": " + hipGetErrorString(e_));            \
    } while (0)

typedef double dft_d4 __attribute__((ext_vector_type(4)));

constexpr int DZC = 16;       // levels per workgroup = MFMA N
constexpr int CST = 17;       // row stride (doubles) of the LDS tiles: 4 consecutive rows land in different banks

// ------------------------------------------------------------------------------------------------ inverse
template <class ST>
__global__ void __launch_bounds__(512)
k_rl_inverse_dft(const double *__restrict__ Az, Planes<ST> phys, const double *__restrict__ phi, const int *__restrict__ Lr,
                 const int *__restrict__ kmaxr, const int64_t *__restrict__ pstart, const int64_t *__restrict__ twoff,
                 const double2 *__restrict__ tw, const int64_t *__restrict__ phoff, const double2 *__restrict__ ph, int V,
                 int nz, int nsz, int K2, int nrings, int64_t N, int64_t azrow, int s_u, int s_r, int s_rr, int s_l, int s_ll,
                 int s_z, int s_zz, const int *__restrict__ slotmask, int ring0, int lcap) {
    extern __shared__ double sm[];
    const int ring = ring0 + blockIdx.z, v = blockIdx.y, z0 = blockIdx.x * DZC;
    const int mask = slotmask[v];
    const int zc = min(DZC, nz - z0);
    const int L = Lr[ring], km = kmaxr[ring];
    const int J4 = (2 * km + 1 + 3) & ~3;                       // coefficient rows, padded to the MFMA K step
    double2 *twl = reinterpret_cast<double2 *>(sm);             // [L]   (cos, sin)(2 pi m / L)
    double *C = sm + 2 * (size_t)lcap;                          // [J4][CST]
    const int j0 = ring / MUBAR;
    const double2 *phr = ph + phoff[ring];
    const int64_t p0 = pstart[ring];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, nw = blockDim.x >> 6;
    for (int m = tid; m < L; m += blockDim.x) twl[m] = tw[twoff[ring] + m];

    for (int q = 0; q < 5; q++) {
        // coefficient set q: (sz, d) = (0,0) (0,1) (0,2) (1,0) (2,0)
        const int sz = q < 3 ? 0 : q - 2, d = q < 3 ? q : 0;
        if (sz >= nsz) break;
        const int slot0 = (q == 0) ? s_u : (q == 1) ? s_r : (q == 2) ? s_rr : (q == 3) ? s_z : s_zz;
        const bool need0 = (mask >> slot0) & 1;
        const bool needl = (q == 0) && ((mask >> s_l) & 1), needll = (q == 0) && ((mask >> s_ll) & 1);
        if (!need0 && !needl && !needll) continue;
        const double *pf = phi + ((int64_t)d * nrings + ring) * 4;
        const double f0 = pf[0], f1 = pf[1], f2 = pf[2], f3 = pf[3];
        __syncthreads();                                        // the previous set has been consumed (and twl is complete)
        for (int e = tid; e < DZC * (km + 1); e += blockDim.x) {
            const int k = e % (km + 1), zz = e / (km + 1);
            double cr = 0.0, ci = 0.0;
            if (zz < zc) {
                const double *a = Az + (int64_t)j0 * azrow + (((int64_t)v * nsz + sz) * nz + (z0 + zz)) * K2;
                if (k == 0) {
                    cr = f0 * a[0] + f1 * a[azrow] + f2 * a[2 * azrow] + f3 * a[3 * azrow];
                } else {
                    const int b = 2 * k;
                    cr = f0 * a[b] + f1 * a[azrow + b] + f2 * a[2 * azrow + b] + f3 * a[3 * azrow + b];
                    ci = f0 * a[b + 1] + f1 * a[azrow + b + 1] + f2 * a[2 * azrow + b + 1] + f3 * a[3 * azrow + b + 1];
                    const double2 w = phr[k];                   // e^{+i k off}
                    const double tr = cr * w.x - ci * w.y;
                    ci = 2.0 * (cr * w.y + ci * w.x);
                    cr = 2.0 * tr;
                }
            }
            if (k == 0) C[zz] = cr;
            else { C[(2 * k - 1) * CST + zz] = cr; C[(2 * k) * CST + zz] = ci; }
        }
        for (int e = tid; e < (J4 - (2 * km + 1)) * DZC; e += blockDim.x)           // zero the padding rows
            C[(2 * km + 1 + e / DZC) * CST + (e % DZC)] = 0.0;
        __syncthreads();

        const int i = lane & 15, kk = lane >> 4;
        const bool is_cos = kk & 1;
        // two row tiles per pass: the B operands (and their derivative factors) are read once for both, and the two
        // accumulator chains are independent, so the matrix pipe is not left waiting on a single dependent chain
        for (int mt = wave; mt * 16 < L; mt += 2 * nw) {
            const int mtb = mt + nw;
            const bool two = mtb * 16 < L;
            const int l0 = min(mt * 16 + i, L - 1), l1 = min(mtb * 16 + i, L - 1);     // padded rows repeat the last point
            // this lane's A column: j = 4 js + kk  ->  wavenumber k = (j + 1) / 2; its angle index m = (k l) mod L
            int k = (kk + 1) >> 1;
            int m0 = (int)(((int64_t)k * l0) % L), m1 = (int)(((int64_t)k * l1) % L);
            int s0 = 2 * l0, s1 = 2 * l1;
            if (s0 >= L) s0 -= L;
            if (s1 >= L) s1 -= L;
            dft_d4 au0 = {0.0, 0.0, 0.0, 0.0}, al0 = au0, all0 = au0, au1 = au0, al1 = au0, all1 = au0;
            for (int js = 0; js < J4 / 4; js++) {
                const double2 t0 = twl[m0], t1 = twl[m1];
                double a0 = is_cos ? t0.x : -t0.y, a1 = is_cos ? t1.x : -t1.y;
                if (js == 0 && kk == 0) { a0 = 1.0; a1 = 1.0; }
                const int j = 4 * js + kk;
                const double bu = C[j * CST + i];
                if (need0) {
                    au0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, bu, au0, 0, 0, 0);
                    if (two) au1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, bu, au1, 0, 0, 0);
                }
                if (needl) {
                    // i k (cr + i ci) = -k ci + i k cr: the cos row takes -k * (its sin partner), the sin row k * (its cos partner)
                    const double bp = (j == 0 || j > 2 * km) ? 0.0 : (is_cos ? -(double)k * C[(j + 1) * CST + i] : (double)k * C[(j - 1) * CST + i]);
                    al0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, bp, al0, 0, 0, 0);
                    if (two) al1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, bp, al1, 0, 0, 0);
                }
                if (needll) {
                    const double bq = -((double)k * k) * bu;
                    all0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, bq, all0, 0, 0, 0);
                    if (two) all1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, bq, all1, 0, 0, 0);
                }
                k += 2;
                m0 += s0;
                if (m0 >= L) m0 -= L;
                m1 += s1;
                if (m1 >= L) m1 -= L;
            }
            // D tile: lane holds column n = lane & 15 (level), rows (lane >> 4) + 4 r (ring points)
            if (i < zc) {
#pragma unroll
                for (int half = 0; half < 2; half++) {
                    if (half == 1 && !two) break;
                    const dft_d4 &au = half ? au1 : au0, &al = half ? al1 : al0, &all_ = half ? all1 : all0;
#pragma unroll
                    for (int r = 0; r < 4; r++) {
                        const int lo = (half ? mtb : mt) * 16 + kk + 4 * r;
                        if (lo >= L) continue;
                        const int64_t pt = (p0 + lo) * nz + z0 + i;
                        if (need0) {
                            if (slot0 == 0) phys.val[(int64_t)v * N + pt] = au[r];
                            else phys.der[((int64_t)(slot0 - 1) * V + v) * N + pt] = (ST)au[r];
                        }
                        if (needl) phys.der[((int64_t)(s_l - 1) * V + v) * N + pt] = (ST)al[r];
                        if (needll) phys.der[((int64_t)(s_ll - 1) * V + v) * N + pt] = (ST)all_[r];
                    }
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------ inverse, RL grids
// Without a vertical dimension the MFMA columns are the requested (variable, derivative plane) pairs of the ring (19 for the
// shallow-water slab sets): all coefficient sets - value, d/dr, d2/dr2 by radial evaluation, d/dlambda, d2/dlambda2 by
// i k / -k^2 - are formed once while staging, so one generated twiddle tile serves every plane.
struct PlaneCols {
    int n;                    // columns in use (<= 16: one MFMA column tile per pass)
    int v[16], slot[16];      // column -> variable, derivative slot
    int colof[8][5];          // variable, kind (u, r, rr, l, ll) -> column or -1
};
struct PlaneGroups { int ng; PlaneCols g[3]; };    // 16 columns per group; blockIdx.z picks the group
constexpr int CSTP = 17;      // row stride of the 16-column coefficient tile

template <class ST>
__global__ void __launch_bounds__(512)
k_rl_inverse_dft_planes(const double *__restrict__ A, Planes<ST> phys, const double *__restrict__ phi, const int *__restrict__ Lr,
                        const int *__restrict__ kmaxr, const int64_t *__restrict__ pstart, const int64_t *__restrict__ twoff,
                        const double2 *__restrict__ tw, const int64_t *__restrict__ phoff, const double2 *__restrict__ ph, int V,
                        int K2, int nrings, int64_t N, int64_t arow, PlaneGroups pgs, int ring0, int lcap) {
    extern __shared__ double sm[];
    const PlaneCols &pc = pgs.g[blockIdx.z];
    // one ring has too little work per plane to fill the chip with a workgroup per ring (300 rings at config 2): the ring's
    // row tiles are split over gridDim.y workgroups, each staging the (small) coefficient tile for itself
    const int ring = ring0 + blockIdx.x;
    const int part = blockIdx.y, nparts = gridDim.y;
    const int L = Lr[ring], km = kmaxr[ring];
    if (part * (int)(blockDim.x >> 6) * 16 >= L) return;        // nothing for this part (uniform for the workgroup)
    const int J4 = (2 * km + 1 + 3) & ~3;
    double2 *twl = reinterpret_cast<double2 *>(sm);
    double *C = sm + 2 * (size_t)lcap;                          // [J4][CSTP]
    const int j0 = ring / MUBAR;
    const double2 *phr = ph + phoff[ring];
    const int64_t p0 = pstart[ring];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, nw = blockDim.x >> 6;
    for (int m = tid; m < L; m += blockDim.x) twl[m] = tw[twoff[ring] + m];
    for (int e = tid; e < J4 * CSTP; e += blockDim.x) C[e] = 0.0;
    __syncthreads();
    const double *pf = phi + (int64_t)ring * 4;
    for (int e = tid; e < V * (km + 1); e += blockDim.x) {
        const int k = e % (km + 1), v = e / (km + 1);
        const double *a = A + (int64_t)j0 * arow + (int64_t)v * K2 + 2 * k;
        double cr[3], ci[3];
#pragma unroll
        for (int d = 0; d < 3; d++) {
            const double *f = pf + (int64_t)d * nrings * 4;
            cr[d] = f[0] * a[0] + f[1] * a[arow] + f[2] * a[2 * arow] + f[3] * a[3 * arow];
            ci[d] = (k == 0) ? 0.0 : f[0] * a[1] + f[1] * a[arow + 1] + f[2] * a[2 * arow + 1] + f[3] * a[3 * arow + 1];
            if (k > 0) {
                const double2 w = phr[k];
                const double tr = cr[d] * w.x - ci[d] * w.y;
                ci[d] = 2.0 * (cr[d] * w.y + ci[d] * w.x);
                cr[d] = 2.0 * tr;
            }
        }
        const int rr_ = (k == 0) ? 0 : 2 * k - 1, ri_ = 2 * k;      // rows of the cos / -sin coefficient
#pragma unroll
        for (int kind = 0; kind < 5; kind++) {
            const int c = pc.colof[v][kind];
            if (c < 0) continue;
            double xr, xi;
            if (kind < 3) { xr = cr[kind]; xi = ci[kind]; }
            else if (kind == 3) { xr = -(double)k * ci[0]; xi = (double)k * cr[0]; }
            else { xr = -((double)k * k) * cr[0]; xi = -((double)k * k) * ci[0]; }
            C[rr_ * CSTP + c] = xr;
            if (k > 0) C[ri_ * CSTP + c] = xi;
        }
    }
    __syncthreads();
    const int i = lane & 15, kk = lane >> 4;
    const bool is_cos = kk & 1;
    for (int mt = part * nw + wave; mt * 16 < L; mt += 2 * nw * nparts) {
        const int mtb = mt + nw * nparts;
        const bool two = mtb * 16 < L;
        const int l0 = min(mt * 16 + i, L - 1), l1 = min(mtb * 16 + i, L - 1);
        int k = (kk + 1) >> 1;
        int m0 = (int)(((int64_t)k * l0) % L), m1 = (int)(((int64_t)k * l1) % L);
        int s0 = 2 * l0, s1 = 2 * l1;
        if (s0 >= L) s0 -= L;
        if (s1 >= L) s1 -= L;
        dft_d4 a00 = {0.0, 0.0, 0.0, 0.0}, a10 = a00;                            // the two row tiles
        for (int js = 0; js < J4 / 4; js++) {
            const double2 t0 = twl[m0], t1 = twl[m1];
            double a0 = is_cos ? t0.x : -t0.y, a1 = is_cos ? t1.x : -t1.y;
            if (js == 0 && kk == 0) { a0 = 1.0; a1 = 1.0; }
            const double b0 = C[(4 * js + kk) * CSTP + i];
            a00 = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b0, a00, 0, 0, 0);
            if (two) a10 = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b0, a10, 0, 0, 0);
            m0 += s0;
            if (m0 >= L) m0 -= L;
            m1 += s1;
            if (m1 >= L) m1 -= L;
        }
#pragma unroll
        for (int half = 0; half < 2; half++) {
            if ((half && !two) || i >= pc.n) continue;
            const dft_d4 &acc = half ? a10 : a00;
            const int vv = pc.v[i], sl = pc.slot[i];
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const int lo = (half ? mtb : mt) * 16 + kk + 4 * r;
                if (lo >= L) continue;
                if (sl == 0) phys.val[(int64_t)vv * N + p0 + lo] = acc[r];
                else phys.der[((int64_t)(sl - 1) * V + vv) * N + p0 + lo] = (ST)acc[r];
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------ forward
// Workgroup = (16 levels, variable, ring).  The ring's points are staged LCH at a time as X[l][level]; wave w owns the
// wavenumber tiles nt = w, w + nw, ... (16 columns j each, at most NTW per wave) and keeps their accumulators across chunks.
constexpr int LCH = 256;      // ring points per staged chunk
constexpr int NTW = 5;        // column tiles per wave: 8 waves x 5 x 16 = 640 columns >= 2 kmax + 2 for kmax <= 319

__global__ void __launch_bounds__(512)
k_fl_forward_dft(const double *__restrict__ np1, double *__restrict__ Fl, const int *__restrict__ Lr,
                 const int *__restrict__ kmaxr, const int64_t *__restrict__ pstart, const int64_t *__restrict__ twoff,
                 const double2 *__restrict__ tw, const int64_t *__restrict__ phoff, const double2 *__restrict__ ph, int V, int nz,
                 int K2, int64_t N, int ring0, int lcap, int planes, int ntw) {
    // planes = 0: the 16 MFMA rows are vertical levels of variable blockIdx.y (RLZ / RZ: z innermost in var_np1).
    // planes = 1: grids without a vertical dimension - the rows are the V variables of the ring (var_np1 is [v][point]).
    extern __shared__ double sm[];
    const int ring = ring0 + blockIdx.z, v = planes ? 0 : blockIdx.y, z0 = planes ? 0 : blockIdx.x * DZC;
    const int zc = planes ? V : min(DZC, nz - z0);
    const int L = Lr[ring], km = kmaxr[ring];
    const int J = 2 * km + 2;                                   // columns: blk 0 (k = 0), blk 1 (padding), Re / Im of k >= 1
    double2 *twl = reinterpret_cast<double2 *>(sm);             // [L]
    double *X = sm + 2 * (size_t)lcap;                          // [LCH][CST]
    const double2 *phr = ph + phoff[ring];
    const int64_t p0 = pstart[ring];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, nw = blockDim.x >> 6;
    for (int m = tid; m < L; m += blockDim.x) twl[m] = tw[twoff[ring] + m];
    const int n = lane & 15, kk = lane >> 4;
    // planes mode: the ring's column tiles are split over gridDim.y workgroups (a workgroup per ring would leave most of
    // the chip idle on RL grids); tile of (wave, q) = (part * ntw + q) * nw + wave, part = 0 otherwise
    const int tile0 = planes ? (int)blockIdx.y * ntw * nw : 0;      // ntw <= NTW column tiles per wave
    if (tile0 * 16 >= J) return;
    dft_d4 acc[NTW];
#pragma unroll
    for (int q = 0; q < NTW; q++) acc[q] = dft_d4{0.0, 0.0, 0.0, 0.0};
    const double *x = planes ? np1 + p0 : np1 + (int64_t)v * N + p0 * nz + z0;
    const int64_t sl = planes ? 1 : nz, szz = planes ? N : 1;        // strides of a ring point / of a row in var_np1

    for (int lc = 0; lc < L; lc += LCH) {
        const int ln = min(LCH, L - lc);                        // multiple of 4 (L is)
        __syncthreads();
        for (int o = tid; o < ln * DZC; o += blockDim.x) {
            const int zz = o & (DZC - 1), l = o >> 4;
            X[l * CST + zz] = (zz < zc) ? x[(int64_t)(lc + l) * sl + zz * szz] : 0.0;
        }
        __syncthreads();
#pragma unroll
        for (int q = 0; q < NTW; q++) {
            const int nt = tile0 + wave + q * nw;
            if (q >= ntw || nt * 16 >= J) continue;
            // this lane's B column: j = nt * 16 + n -> wavenumber k = j / 2 (cos for even j, -sin for odd j)
            const int j = nt * 16 + n;
            const int k = min(j >> 1, km);
            const bool is_cos = !(j & 1);
            int m = (int)(((int64_t)k * (lc + kk)) % L);        // angle index of (k, l = lc + kk); advances by 4 k per step
            int fourk = 4 * k;
            while (fourk >= L) fourk -= L;
            for (int ls = 0; ls < ln; ls += 4) {
                const double2 t = twl[m];
                const double b = is_cos ? t.x : -t.y;
                const double a = X[(ls + kk) * CST + n];        // A tile: row = level (lane & 15), column = ring point
                acc[q] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[q], 0, 0, 0);
                m += fourk;
                if (m >= L) m -= L;
            }
        }
    }
    // D tile: lane holds column n (= j within the tile), rows kk + 4 r (levels).  (Re, Im) of a wavenumber sit in
    // neighbouring lanes: rotate by the ring's phase reference e^{-i k off} and scale by 1 / L
    const double inv = 1.0 / L;
#pragma unroll
    for (int q = 0; q < NTW; q++) {
        const int nt = tile0 + wave + q * nw;
        if (q >= ntw || nt * 16 >= J) continue;
        const int j = nt * 16 + n;
        const int k = j >> 1;
        const double2 w = phr[min(k, km)];
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const double mine = acc[q][r];
            const double other = __shfl_xor(mine, 1);           // partner column (j ^ 1) of the same level
            const int zz = kk + 4 * r;
            double out;
            if (j == 0) out = mine * inv;
            else if (j == 1) out = 0.0;
            else if (!(j & 1)) out = (mine * w.x + other * w.y) * inv;      // Re: sr w.x + si w.y
            else out = (mine * w.x - other * w.y) * inv;                    // Im: si w.x - sr w.y
            // Fl [ring][v][z][blk]: row zz is level z0 + zz of variable v, or (planes) variable zz of a grid with nz = 1
            if (j < J && zz < zc) Fl[(planes ? (int64_t)ring * V + zz : ((int64_t)ring * V + v) * nz + z0 + zz) * K2 + j] = out;
        }
    }
}

// ------------------------------------------------------------------------------------------------ launchers
// grids without a vertical dimension: columns = (variable, plane)
static bool dft_planes(const sx_handle *h) { return !h->has_z; }

bool dft_mfma_ok(const sx_handle *h) {
    if (!h->has_l || fft_path_ok(h) || h->kmax_max > 319) return false;
    if (dft_planes(h) ? (h->V > 8 || h->D > 5) : h->nz < 8) return false;
    static const bool off = getenv("SX_DFT_MFMA") && atoi(getenv("SX_DFT_MFMA")) == 0;       // scalar kernels instead (debugging)
    if (off) return false;
    return h->L_all_mult4;
}

// Rings are launched in classes of growing size so that the small ones do not pay the LDS footprint (and with it the
// occupancy) of the largest: class c covers rings [c n/4, (c+1) n/4), sized for its last ring.
template <class F>
static void for_ring_classes(sx_handle *h, int n_rings, F f, int max_classes = 4) {
    const int ncls = n_rings >= 16 ? max_classes : 1;
    for (int c = 0; c < ncls; c++) {
        const int r0 = (int)((int64_t)n_rings * c / ncls), r1 = (int)((int64_t)n_rings * (c + 1) / ncls);
        if (r1 <= r0) continue;
        int lcap = 0, kcap = 0;
        for (int i = r0; i < r1; i++) { lcap = std::max(lcap, h->hL[i]); kcap = std::max(kcap, h->hkmax[i]); }
        f(r0, r1 - r0, lcap, kcap);
    }
}

static void launch_rl_inverse_dft_planes(sx_handle *h, bool full) {
    const std::vector<int> &mask = full ? h->hmask_full : h->hmask_eq;
    // kinds in slot order for RL grids: u, r, rr, l, ll  (slot[0..4])
    PlaneGroups pgs;
    pgs.ng = 0;
    PlaneCols pc;
    auto reset = [&]() { pc.n = 0; for (auto &row : pc.colof) for (int &c : row) c = -1; };
    auto push = [&]() { if (pgs.ng < 3) pgs.g[pgs.ng++] = pc; reset(); };
    reset();
    for (int v = 0; v < h->V; v++) {
        int need = 0;
        for (int kind = 0; kind < 5; kind++) need += (h->slot[kind] >= 0 && ((mask[v] >> h->slot[kind]) & 1));
        if (pc.n + need > 16) push();
        for (int kind = 0; kind < 5; kind++) {
            const int sl = h->slot[kind];
            if (sl < 0 || !((mask[v] >> sl) & 1)) continue;
            pc.colof[v][kind] = pc.n; pc.v[pc.n] = v; pc.slot[pc.n] = sl; pc.n++;
        }
    }
    if (pc.n > 0) push();                       // V <= 8 variables x 5 planes = 40 columns at most: 3 groups
    if (pgs.ng == 0) return;
    const double *a = h->d_A + (int64_t)h->cell0 * h->C;
    // two launch classes only: each launch is as long as its largest ring's workgroup, so more classes mostly add tails
    for_ring_classes(h, h->nrings, [&](int r0, int nr, int lcap, int kcap) {
        const size_t lds = sizeof(double) * (2 * (size_t)lcap + (size_t)((2 * kcap + 4) & ~3) * CSTP);
        const int nsplit = std::min(8, std::max(1, ((lcap + 15) / 16 + 15) / 16));       // 8 waves x 2 row tiles per workgroup pass
#define DFT_INVP(ST)                                                                                                                 \
        {                                                                                                                            \
            auto kern = k_rl_inverse_dft_planes<ST>;                                                                                 \
            HIPCHK3(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); \
            hipLaunchKernelGGL(kern, dim3(nr, nsplit, pgs.ng), dim3(512), lds, h->stream, a, planes_of<ST>(h->d_phys, h->V, h->N),   \
                               h->d_phi, h->d_L, h->d_kmax, h->d_pstart, h->d_twoff, h->d_tw, h->d_phoff, h->d_ph, h->V, h->K2,      \
                               h->nrings, h->N, h->C, pgs, r0, lcap);                                                                \
        }
        if (h->f32) DFT_INVP(float) else DFT_INVP(double)
#undef DFT_INVP
        HIPCHK3(hipGetLastError());
    }, 2);
}

void launch_rl_inverse_dft(sx_handle *h, const int *d_mask) {
    const int id = timer_id(h, "k_rl_inverse");
    timer_begin(h, id);
    if (dft_planes(h)) {
        launch_rl_inverse_dft_planes(h, d_mask == h->d_mask_full);
        timer_end(h);
        return;
    }
    const double *az = h->has_z ? h->d_Az : h->d_A + (int64_t)h->cell0 * h->C;
    const int64_t azrow = h->has_z ? (int64_t)h->V * 3 * h->nz * h->K2 : h->C;
    for_ring_classes(h, h->nrings, [&](int r0, int nr, int lcap, int kcap) {
        const size_t lds = sizeof(double) * (2 * (size_t)lcap + (size_t)((2 * kcap + 4) & ~3) * CST);
        dim3 g((h->nz + DZC - 1) / DZC, h->V, nr);
#define DFT_INV(ST)                                                                                                                  \
        {                                                                                                                            \
            auto kern = k_rl_inverse_dft<ST>;                                                                                        \
            HIPCHK3(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); \
            hipLaunchKernelGGL(kern, g, dim3(512), lds, h->stream, az, planes_of<ST>(h->d_phys, h->V, h->N), h->d_phi, h->d_L,       \
                               h->d_kmax, h->d_pstart, h->d_twoff, h->d_tw, h->d_phoff, h->d_ph, h->V, h->nz, h->nsz, h->K2,         \
                               h->nrings, h->N, azrow, h->slot[0], h->slot[1], h->slot[2], h->slot[3], h->slot[4], h->slot[5],       \
                               h->slot[6], d_mask, r0, lcap);                                                                        \
        }
        if (h->f32) DFT_INV(float) else DFT_INV(double)
#undef DFT_INV
        HIPCHK3(hipGetLastError());
    });
    timer_end(h);
}

void launch_fl_forward_dft(sx_handle *h) {
    const int id = timer_id(h, "k_fl_forward");
    timer_begin(h, id);
    const int planes = dft_planes(h) ? 1 : 0;
    for_ring_classes(h, h->nrings, [&](int r0, int nr, int lcap, int) {
        const size_t lds = sizeof(double) * (2 * (size_t)lcap + (size_t)LCH * CST);
        // planes: one column tile per wave, the ring's (2 kmax + 2) / 16 tiles spread over gridDim.y workgroups of 8 waves
        const int ntw = planes ? 1 : NTW;
        int kcap = 0;
        for (int i = r0; i < r0 + nr; i++) kcap = std::max(kcap, h->hkmax[i]);
        dim3 g(planes ? 1 : (h->nz + DZC - 1) / DZC, planes ? ((2 * kcap + 2 + 15) / 16 + 7) / 8 : h->V, nr);
        HIPCHK3(hipFuncSetAttribute(reinterpret_cast<const void *>(k_fl_forward_dft), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(k_fl_forward_dft, g, dim3(512), lds, h->stream, h->d_np1, h->d_Fl, h->d_L, h->d_kmax, h->d_pstart,
                           h->d_twoff, h->d_tw, h->d_phoff, h->d_ph, h->V, h->nz, h->K2, h->N, r0, lcap, planes, ntw);
        HIPCHK3(hipGetLastError());
    }, planes ? 2 : 4);
    timer_end(h);
}

}  // namespace sx
