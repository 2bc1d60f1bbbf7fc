// HIP kernels (gfx950) for the Scythe.jl spectral-transform time-stepping path.
//
// Device data layout:
//   spectral   A, B     [radial node m][col]            col = (v * Zb + zm) * K2 + blk   (blk fastest)
//                       blk = 0: k = 0;  1: unused (zero);  2k: Re k;  2k+1: Im k   => (Re, Im) pairs are 16-B aligned
//   Az                  [tile node j][v][sz][z][blk]    sz = value, d/dz, d2/dz2 (z already inverted)
//   physical            [slot][v][point]                point = (pstart[ring] + l) * nz + z   (reference layout)
//   var_np1, expdot_*   [v][point]
//   Fl                  [ring][v][z][blk]               ring spectra of var_np1
//   Bz                  [tile node j][v][z][blk]        radial inner products before the vertical transform
// The radial node is the slowest index of every spectral array so that (a) the banded solve runs one lane per
// right-hand side with perfectly coalesced rows, (b) radial evaluation / inner products stream whole rows,
// (c) a tile's halo (3 nodes) is one contiguous block.
#include "sx_internal.hpp"
#include <cmath>

namespace sx {

#define HIPCHK(x)                                                                                   \
    do {                                                                                            \
        hipError_t e_ = (x);                                                                        \
        if (e_ != hipSuccess) set_error(std::string(#x) + ": " + hipGetErrorString(e_) + " (" __FILE__ ":" + std::to_string(__LINE__) + ")"); \
    } while (0)

constexpr int ZC = 16;   // z-levels per workgroup in the ring kernels: 16 * 8 B = one 128-B line per (ring point)

// ------------------------------------------------------------------------------------------------ vertical transforms
// Dense Chebyshev collocation products on spectral-sized data, one "job" per (variable, operator):
//   out[row][job.out_off + o*K2 + blk] = sum_i M[job.mat_off + o*n_in + i] * in[row0 + row][job.in_off + i*K2 + blk]
// inverse: in = A rows, M = Mz[v][sz] (b -> value / d/dz / d2/dz2 incl. BC projection), out = Az
// forward: in = Bz rows, M = CB (values -> truncated b), out = B
// Workgroup = 64 wavenumber blocks x 4 output groups; the input tile sits in LDS, the operator entries are
// wave-uniform (scalar loads), every thread accumulates 4 outputs per pass over the input.
__global__ void __launch_bounds__(256)
k_colmat(const double *__restrict__ in, double *__restrict__ out, const double *__restrict__ mats,
         const ColJob *__restrict__ jobs, int n_in, int n_out, int K2, int64_t in_row, int64_t out_row, int row0) {
    extern __shared__ double As[];
    const int lane = threadIdx.x;
    const int g = __builtin_amdgcn_readfirstlane(threadIdx.y);     // blockDim.x == 64: one wave per g => operator
    const int blk = blockIdx.x * 64 + lane;                         // entries become scalar loads
    const ColJob job = jobs[blockIdx.y];
    const double *src = in + (int64_t)(row0 + blockIdx.z) * in_row + job.in_off;
    double *dst = out + (int64_t)blockIdx.z * out_row + job.out_off;
    const double *M = mats + job.mat_off;
    const bool ok = blk < K2;
    for (int i = g; i < n_in; i += 4) As[i * 64 + lane] = ok ? src[(int64_t)i * K2 + blk] : 0.0;
    __syncthreads();
    for (int o0 = g * 4; o0 < n_out; o0 += 16) {
        const double *m0 = M + (int64_t)o0 * n_in;
        const double *m1 = M + (int64_t)min(o0 + 1, n_out - 1) * n_in;
        const double *m2 = M + (int64_t)min(o0 + 2, n_out - 1) * n_in;
        const double *m3 = M + (int64_t)min(o0 + 3, n_out - 1) * n_in;
        double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
        for (int i = 0; i < n_in; i++) {
            const double x = As[i * 64 + lane];
            a0 += m0[i] * x;
            a1 += m1[i] * x;
            a2 += m2[i] * x;
            a3 += m3[i] * x;
        }
        if (ok) {
            dst[(int64_t)o0 * K2 + blk] = a0;
            if (o0 + 1 < n_out) dst[(int64_t)(o0 + 1) * K2 + blk] = a1;
            if (o0 + 2 < n_out) dst[(int64_t)(o0 + 2) * K2 + blk] = a2;
            if (o0 + 3 < n_out) dst[(int64_t)(o0 + 3) * K2 + blk] = a3;
        }
    }
}

// The same product on the f64 matrix cores for n_out a multiple of 16 (zDim 32 / 64 / 128): wave = (16 wavenumber blocks) x
// all n_out outputs, K = n_in in steps of 4.  B comes straight from the A-coefficient rows (16 consecutive blocks = one
// 128-byte line per k), the operator fragments are shared by every wave of the launch (L1 / L2 resident), the result
// tile is stored as 128-byte lines: no LDS, no barrier, and the scalar operator loads of k_colmat (its limiter) are gone.
typedef double colmat_d4 __attribute__((ext_vector_type(4)));

// CT = column tiles (16 wavenumber blocks each) per wave: a tile's operator fragments are fetched once per wave and row tile
// and serve CT column tiles.  At 128 levels every wave pulls the whole 87 KB operator through L2 (6.7 GB per step at config 5
// with CT = 1, three times the kernel's HBM bytes): CT = 2 there.
template <int MT, class OT = double, int CT = 1>          // n_out / 16; OT = float: the fp32 spectral-intermediate mode (storage_f32 = 2)
__global__ void __launch_bounds__(256)
k_colmat_mfma(const double *__restrict__ in, OT *__restrict__ out, const double *__restrict__ mats,
              const ColJob *__restrict__ jobs, int n_in, int K2, int64_t in_row, int64_t out_row, int row0) {
    constexpr int n_out = MT * 16;
    constexpr int KSTEPS = MT * 4;          // K = n_in <= n_out in steps of 4, fully unrolled (rows beyond n_in contribute zeros)
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int n = lane & 15, kk = lane >> 4;
    const int blk0 = (blockIdx.x * 4 + wave) * 16 * CT;
    if (blk0 >= K2) return;
    const ColJob job = jobs[blockIdx.y];
    const double *src = in + (int64_t)(row0 + blockIdx.z) * in_row + job.in_off;
    OT *dst = out + (int64_t)blockIdx.z * out_row + job.out_off;
    const double *MTr = mats + job.mat_off;                // operator transposed: [n_in][n_out]
    // Every B element (coefficient row k, block n) of this wave is requested before the first MFMA, and each tile's operator
    // fragments (L2-resident) before that tile's chain: with the loads inside the K loop every step of 4 waited for its own
    // round trip (11 in a row at b_zDim 43).
    double b[CT][KSTEPS];
#pragma unroll
    for (int c = 0; c < CT; c++) {
        const int blk = min(blk0 + c * 16 + n, K2 - 1);
#pragma unroll
        for (int ks = 0; ks < KSTEPS; ks++) {
            const int k = 4 * ks + kk;
            b[c][ks] = (k < n_in) ? src[(int64_t)k * K2 + blk] : 0.0;
        }
    }
#pragma unroll
    for (int t = 0; t < MT; t++) {
        double a[KSTEPS];
#pragma unroll
        for (int ks = 0; ks < KSTEPS; ks++) {
            const int k = 4 * ks + kk;
            a[ks] = (k < n_in) ? MTr[(int64_t)k * n_out + t * 16 + n] : 0.0;      // A[m = lane & 15][k], 128-byte rows
        }
#pragma unroll
        for (int c = 0; c < CT; c++) {
            colmat_d4 acc = colmat_d4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int ks = 0; ks < KSTEPS; ks++)
                if (4 * ks < n_in) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a[ks], b[c][ks], acc, 0, 0, 0);
            if (blk0 + c * 16 + n < K2) {
#pragma unroll
                for (int r = 0; r < 4; r++) dst[(int64_t)(t * 16 + kk + 4 * r) * K2 + blk0 + c * 16 + n] = (OT)acc[r];
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------ radial + azimuthal inverse
// One workgroup per (z-chunk, variable, ring): radial evaluation (4 rows of Az), phase reference, truncated inverse
// DFT with lambda-derivatives, stores straight into the reference physical layout (z innermost => 128-B lines).
template <class ST>
__global__ void __launch_bounds__(256)
k_rl_inverse(const double *__restrict__ Az, Planes<ST> phys, const double *__restrict__ phi,
             const int *__restrict__ Lr, const int *__restrict__ kmaxr, const int64_t *__restrict__ pstart,
             const int64_t *__restrict__ twoff, const double2 *__restrict__ tw, const int64_t *__restrict__ phoff,
             const double2 *__restrict__ ph, int V, int nz, int nsz, int K2, int nrings, int64_t N, int64_t azrow,
             int s_u, int s_r, int s_rr, int s_l, int s_ll, int s_z, int s_zz, int has_l, int cstride,
             const int *__restrict__ slotmask) {
    extern __shared__ double sm[];
    const int ring = blockIdx.z, v = blockIdx.y, z0 = blockIdx.x * ZC;
    const int mask = slotmask[v];
    const int zc = min(ZC, nz - z0);
    const int L = Lr[ring], km = has_l ? kmaxr[ring] : 0;
    const int j0 = ring / MUBAR;
    double *cR = sm, *cI = sm + (size_t)ZC * cstride;
    const double2 *twr = tw + twoff[ring];
    const double2 *phr = ph + phoff[ring];
    const int64_t p0 = pstart[ring];
    const int tid = threadIdx.x;

    for (int q = 0; q < 5; q++) {
        // coefficient line q: (sz, d) = (0,0) (0,1) (0,2) (1,0) (2,0)
        const int sz = q < 3 ? 0 : q - 2, d = q < 3 ? q : 0;
        if (sz >= nsz) break;
        const int slot0 = (q == 0) ? s_u : (q == 1) ? s_r : (q == 2) ? s_rr : (q == 3) ? s_z : s_zz;
        const bool need0 = (mask >> slot0) & 1;
        const bool needl = (q == 0) && has_l && ((mask >> s_l) & 1), needll = (q == 0) && has_l && ((mask >> s_ll) & 1);
        if (!need0 && !needl && !needll) continue;
        const double *p = phi + ((int64_t)d * nrings + ring) * 4;
        const double f0 = p[0], f1 = p[1], f2 = p[2], f3 = p[3];
        __syncthreads();
        for (int e = tid; e < zc * (km + 1); e += blockDim.x) {
            const int k = e % (km + 1), zz = e / (km + 1);
            const double *a = Az + (int64_t)j0 * azrow + (((int64_t)v * nsz + sz) * nz + (z0 + zz)) * K2;
            double cr, ci = 0.0;
            if (k == 0) {
                cr = f0 * a[0] + f1 * a[azrow] + f2 * a[2 * azrow] + f3 * a[3 * azrow];
            } else {
                const int b = 2 * k;
                cr = f0 * a[b] + f1 * a[azrow + b] + f2 * a[2 * azrow + b] + f3 * a[3 * azrow + b];
                ci = f0 * a[b + 1] + f1 * a[azrow + b + 1] + f2 * a[2 * azrow + b + 1] + f3 * a[3 * azrow + b + 1];
                const double2 w = phr[k];          // e^{+i k off}
                const double tr = cr * w.x - ci * w.y;
                ci = cr * w.y + ci * w.x;
                cr = 2.0 * tr;
                ci = 2.0 * ci;
            }
            cR[zz * cstride + k] = cr;
            cI[zz * cstride + k] = ci;
        }
        __syncthreads();
        const bool lamder = needl || needll;
        for (int o = tid; o < L * zc; o += blockDim.x) {
            const int zz = o % zc, l = o / zc;
            const double *r = cR + zz * cstride, *im = cI + zz * cstride;
            double a0 = r[0], a1 = 0.0, a2 = 0.0;
            int idx = 0;
            for (int k = 1; k <= km; k++) {
                idx += l;
                if (idx >= L) idx -= L;
                const double2 t = twr[idx];
                const double val = r[k] * t.x - im[k] * t.y;
                a0 += val;
                if (lamder) {
                    a1 -= k * (im[k] * t.x + r[k] * t.y);
                    a2 -= (double)k * k * val;
                }
            }
            const int64_t pt = (p0 + l) * nz + z0 + zz;
            if (need0) {
                if (slot0 == 0) phys.val[(int64_t)v * N + pt] = a0;
                else phys.der[((int64_t)(slot0 - 1) * V + v) * N + pt] = (ST)a0;
            }
            if (needl) phys.der[((int64_t)(s_l - 1) * V + v) * N + pt] = (ST)a1;
            if (needll) phys.der[((int64_t)(s_ll - 1) * V + v) * N + pt] = (ST)a2;
        }
    }
}

// ------------------------------------------------------------------------------------------------ forward azimuthal
// Fl[ring][v][z][blk] = (1/L) sum_l var_np1[v][(pstart + l) nz + z] e^{-ik lambda_l}
__global__ void __launch_bounds__(256)
k_fl_forward(const double *__restrict__ np1, double *__restrict__ Fl, const int *__restrict__ Lr,
             const int *__restrict__ kmaxr, const int64_t *__restrict__ pstart, const int64_t *__restrict__ twoff,
             const double2 *__restrict__ tw, const int64_t *__restrict__ phoff, const double2 *__restrict__ ph, int V,
             int nz, int K2, int64_t N, int has_l, int xstride) {
    extern __shared__ double sm[];
    const int ring = blockIdx.z, v = blockIdx.y, z0 = blockIdx.x * ZC;
    const int zc = min(ZC, nz - z0);
    const int L = Lr[ring], km = has_l ? kmaxr[ring] : 0;
    const double2 *twr = tw + twoff[ring];
    const double2 *phr = ph + phoff[ring];
    const int64_t p0 = pstart[ring];
    const int tid = threadIdx.x;
    for (int o = tid; o < L * zc; o += blockDim.x) {
        const int zz = o % zc, l = o / zc;
        sm[zz * xstride + l] = np1[(int64_t)v * N + (p0 + l) * nz + z0 + zz];
    }
    __syncthreads();
    const double inv = 1.0 / L;
    for (int e = tid; e < zc * (km + 1); e += blockDim.x) {
        const int k = e % (km + 1), zz = e / (km + 1);
        const double *x = sm + zz * xstride;
        double sr = 0.0, si = 0.0;
        int idx = 0;
        for (int l = 0; l < L; l++) {
            const double2 t = twr[idx];
            sr += x[l] * t.x;
            si -= x[l] * t.y;
            idx += k;
            if (idx >= L) idx -= L;
        }
        double *out = Fl + (((int64_t)ring * V + v) * nz + z0 + zz) * K2;
        if (k == 0) {
            out[0] = sr * inv;
        } else {
            const double2 w = phr[k];              // multiply by e^{-i k off}
            out[2 * k] = (sr * w.x + si * w.y) * inv;
            out[2 * k + 1] = (si * w.x - sr * w.y) * inv;
        }
    }
}

// ------------------------------------------------------------------------------------------------ radial inner products
// Bz[j][e] = sum over the rings of cells j-3..j of wq * phi0 * Fl[ring][e],   e = (v, z, blk) flattened
__global__ void k_sb(const double *__restrict__ Fl, double *__restrict__ Bz, const double *__restrict__ phi,
                     const double *__restrict__ wq, int ncells, int64_t plane) {
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int j = blockIdx.y;
    if (e >= plane) return;
    double s = 0.0;
    for (int c = max(0, j - 3); c <= min(ncells - 1, j); c++) {
        const int jj = j - c;
        for (int mu = 0; mu < MUBAR; mu++) {
            const int ring = c * MUBAR + mu;
            s += wq[ring] * phi[(int64_t)ring * 4 + jj] * Fl[(int64_t)ring * plane + e];
        }
    }
    Bz[(int64_t)j * plane + e] = s;
}

// ------------------------------------------------------------------------------------------------ radial inner products + vertical forward
// Fused k_sb + vertical forward transform (RZ / RLZ): the [z][64 blocks] tile of radial inner products of node j stays in
// LDS and is contracted with CB right away, so Bz never goes to HBM.
//   B[j][v][zm][blk] = sum_z CB[zm][z] * sum_{rings of cells j-3..j} wq * phi0 * Fl[ring][v][z][blk]
__global__ void __launch_bounds__(256)
k_sbz(const double *__restrict__ Fl, double *__restrict__ B, const double *__restrict__ phi, const double *__restrict__ wq,
      const double *__restrict__ CB, int ncells, int V, int nz, int Zb, int K2, int64_t C) {
    extern __shared__ double As[];          // [nz][64]
    const int lane = threadIdx.x;
    const int g = __builtin_amdgcn_readfirstlane(threadIdx.y);
    const int blk = blockIdx.x * 64 + lane;
    const int v = blockIdx.y, j = blockIdx.z;
    const bool ok = blk < K2;
    const int64_t plane = (int64_t)V * nz * K2;
    const int c0 = max(0, j - 3), c1 = min(ncells - 1, j);
    for (int z = g; z < nz; z += 4) {
        double s = 0.0;
        if (ok) {
            const int64_t e = ((int64_t)v * nz + z) * K2 + blk;
            for (int c = c0; c <= c1; c++) {
                const int jj = j - c;
#pragma unroll
                for (int mu = 0; mu < MUBAR; mu++) {
                    const int ring = c * MUBAR + mu;
                    s += wq[ring] * phi[(int64_t)ring * 4 + jj] * Fl[(int64_t)ring * plane + e];
                }
            }
        }
        As[z * 64 + lane] = s;
    }
    __syncthreads();
    double *dst = B + (int64_t)j * C + (int64_t)v * Zb * K2;
    for (int o0 = g * 4; o0 < Zb; o0 += 16) {
        const double *m0 = CB + (int64_t)o0 * nz;
        const double *m1 = CB + (int64_t)min(o0 + 1, Zb - 1) * nz;
        const double *m2 = CB + (int64_t)min(o0 + 2, Zb - 1) * nz;
        const double *m3 = CB + (int64_t)min(o0 + 3, Zb - 1) * nz;
        double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
        for (int i = 0; i < nz; i++) {
            const double x = As[i * 64 + lane];
            a0 += m0[i] * x;
            a1 += m1[i] * x;
            a2 += m2[i] * x;
            a3 += m3[i] * x;
        }
        if (ok) {
            dst[(int64_t)o0 * K2 + blk] = a0;
            if (o0 + 1 < Zb) dst[(int64_t)(o0 + 1) * K2 + blk] = a1;
            if (o0 + 2 < Zb) dst[(int64_t)(o0 + 2) * K2 + blk] = a2;
            if (o0 + 3 < Zb) dst[(int64_t)(o0 + 3) * K2 + blk] = a3;
        }
    }
}

// Sliding-window form of k_sbz for zDim = NZ (multiple of 8): a workgroup walks a run of consecutive radial cells for
// one (variable, 64 wavenumber blocks) and keeps the partial inner products of the 4 nodes a cell touches in registers,
// so every Fl value enters the CU once (k_sbz re-reads it for each of its 4 nodes: 4x the L2 -> L1 traffic, which is
// what bounds it).  Node c is complete once cell c has been added (cells c-3..c); its [NZ][64] tile then goes through
// LDS into the vertical contraction with CB.  A segment starts 3 cells early to warm up its first nodes.
// Summation order per node (cells ascending, mish points ascending) is the same as k_sbz's.
#ifdef SX_PHASES
__device__ long long *g_sbw_dbg = nullptr;     // [workgroup][8]: cycles in load+accumulate, barrier 1, LDS write + barrier 2, contraction + stores; cells; total
#define SBW_T0() long long st_ = (long long)__builtin_readcyclecounter(); const long long st0_ = st_; long long sacc_[4] = {0, 0, 0, 0}; int scells_ = 0
#define SBW_LAP(i) do { const long long n_ = (long long)__builtin_readcyclecounter(); sacc_[i] += n_ - st_; st_ = n_; } while (0)
#define SBW_END() do { if (threadIdx.x == 0 && g_sbw_dbg) { long long *d_ = g_sbw_dbg + ((int64_t)(blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) * 8; \
        d_[0] = sacc_[0]; d_[1] = sacc_[1]; d_[2] = sacc_[2]; d_[3] = sacc_[3]; d_[4] = scells_; d_[5] = (long long)__builtin_readcyclecounter() - st0_; } } while (0)
#else
#define SBW_T0() do { } while (0)
#define SBW_LAP(i) do { } while (0)
#define SBW_END() do { } while (0)
#endif

template <int NZ, bool PREFETCH>
__global__ void __launch_bounds__(512, 2)       // second argument: waves per SIMD (one 512-thread workgroup per CU)
k_sbw(const double *__restrict__ Fl, double *__restrict__ B, const double *__restrict__ phi, const double *__restrict__ wq,
      const double *__restrict__ CB, int ncells, int V, int Zb, int K2, int64_t C, int cps) {
    constexpr int ZPT = NZ / 8;
    __shared__ double As[NZ * 64];
    const int lane = threadIdx.x & 63;
    const int g = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int blk = blockIdx.x * 64 + lane;
    const int v = blockIdx.y;
    const bool ok = blk < K2;
    const int ca = blockIdx.z * cps, cb = min(ca + cps, ncells);
    const int cend = (cb == ncells) ? ncells + 3 : cb;          // the last segment also owns the 3 trailing nodes
    const int cstart = max(0, ca - 3);
    const int64_t plane = (int64_t)V * NZ * K2;
    const double *base = Fl + ((int64_t)v * NZ + g) * K2 + (ok ? blk : 0);
    double *dst0 = B + (int64_t)v * Zb * K2 + blk;
    double acc[4][ZPT];
#pragma unroll
    for (int q = 0; q < 4; q++)
#pragma unroll
        for (int i = 0; i < ZPT; i++) acc[q][i] = 0.0;
    // The ring spectra of cell c + 1 are requested as soon as cell c has been accumulated, BEFORE its node is contracted: the
    // loads then fly during the LDS contraction instead of starting after it (a workgroup used to have at most ~11 loads
    // per wave in flight and none at all during the contraction; read once: non-temporal, the caches stay with B and the
    // history).
    double xn[PREFETCH ? MUBAR : 1][ZPT];
    auto fetch = [&](int c) {
        if (!PREFETCH || c < cstart || c >= cend || c >= ncells) return;
#pragma unroll
        for (int mu = 0; mu < MUBAR; mu++) {
            const double *src = base + (int64_t)(c * MUBAR + mu) * plane;
#pragma unroll
            for (int i = 0; i < ZPT; i++) xn[mu][i] = __builtin_nontemporal_load(src + (int64_t)(8 * i) * K2);
        }
    };
    fetch(cstart);
    SBW_T0();
    for (int c4 = cstart & ~3; c4 < cend; c4 += 4) {
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const int c = c4 + u;
            if (c < cstart || c >= cend) continue;
#ifdef SX_PHASES
            scells_++;
#endif
            if (c < ncells) {
#pragma unroll
                for (int mu = 0; mu < MUBAR; mu++) {
                    const int ring = c * MUBAR + mu;
                    const double w = wq[ring];
                    const double w0 = w * phi[(int64_t)ring * 4], w1 = w * phi[(int64_t)ring * 4 + 1];
                    const double w2 = w * phi[(int64_t)ring * 4 + 2], w3 = w * phi[(int64_t)ring * 4 + 3];
                    double x[ZPT];
                    if (PREFETCH) {
#pragma unroll
                        for (int i = 0; i < ZPT; i++) x[i] = xn[mu][i];
                    } else {
                        const double *src = base + (int64_t)ring * plane;
#pragma unroll
                        for (int i = 0; i < ZPT; i++) x[i] = __builtin_nontemporal_load(src + (int64_t)(8 * i) * K2);
                    }
#pragma unroll
                    for (int i = 0; i < ZPT; i++) {
                        acc[u][i] += w0 * x[i];
                        acc[(u + 1) & 3][i] += w1 * x[i];
                        acc[(u + 2) & 3][i] += w2 * x[i];
                        acc[(u + 3) & 3][i] += w3 * x[i];
                    }
                }
                fetch(c + 1);           // in flight while node c goes through LDS and the vertical contraction
            }
            SBW_LAP(0);
            if (c >= ca) {                      // node c is complete: vertical forward transform and store
                __syncthreads();                // the previous node's tile has been consumed
                SBW_LAP(1);
#pragma unroll
                for (int i = 0; i < ZPT; i++) As[(g + 8 * i) * 64 + lane] = acc[u][i];
                __syncthreads();
                SBW_LAP(2);
                double *dst = dst0 + (int64_t)c * C;
                for (int o0 = g * 4; o0 < Zb; o0 += 32) {
                    const double *m0 = CB + (int64_t)o0 * NZ;
                    const double *m1 = CB + (int64_t)min(o0 + 1, Zb - 1) * NZ;
                    const double *m2 = CB + (int64_t)min(o0 + 2, Zb - 1) * NZ;
                    const double *m3 = CB + (int64_t)min(o0 + 3, Zb - 1) * NZ;
                    double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
#pragma unroll 8
                    for (int i = 0; i < NZ; i++) {
                        const double xx = As[i * 64 + lane];
                        a0 += m0[i] * xx;
                        a1 += m1[i] * xx;
                        a2 += m2[i] * xx;
                        a3 += m3[i] * xx;
                    }
                    if (ok) {
                        dst[(int64_t)o0 * K2] = a0;
                        if (o0 + 1 < Zb) dst[(int64_t)(o0 + 1) * K2] = a1;
                        if (o0 + 2 < Zb) dst[(int64_t)(o0 + 2) * K2] = a2;
                        if (o0 + 3 < Zb) dst[(int64_t)(o0 + 3) * K2] = a3;
                    }
                }
                SBW_LAP(3);
            }
#pragma unroll
            for (int i = 0; i < ZPT; i++) acc[u][i] = 0.0;      // the slot now belongs to node c + 4
        }
    }
    SBW_END();
}

// The same with the vertical contraction on the f64 matrix cores and the next cell's ring spectra in flight meanwhile
// (zDim 64 / 32; b_zDim <= 64).  Phase stamps of k_sbw (profiles/r02/phases_sbw.txt): 57 % of a workgroup's time was the
// contraction - 11,000 cycles per node against ~1,500 of arithmetic: its operator entries arrive as scalar loads, two
// dependent batches per 8 terms, and with 43 output rows over 8 waves x 4 rows three waves ran a second pass while five
// waited.  Here the operator lives in LDS in MFMA-fragment order for the whole kernel: wave w owns column tile w & 3 (16 wavenumber blocks)
// and the row tiles of its half (w < 4: the first ceil(MT / 2) tiles of 16 modes, else the rest), 16 K-steps of
// v_mfma_f64_16x16x4_f64 per tile with B = the node's [level][block] tile in LDS (row stride 80 doubles: the 4 levels a
// K-step reads fall in disjoint bank halves).  One 512-thread workgroup per CU; the grid is one round.
// Summation order differs from k_sbw / k_sbz (K in blocks of 4): results agree to rounding, not bitwise.
// BW = wavenumber blocks per workgroup: 64 (zDim 32 / 64), or 32 for zDim 128, where the operator fragments (6 row tiles x
// 32 K steps = 96 KB) and the node tile (128 levels x 32 blocks) have to share the 160 KB; then wave w owns column tile
// w & 1 and the row tiles (w >> 1), (w >> 1) + 4.
template <int NZ, int BW = 64, int THREADS = 512, class FT = double>      // FT = float: fp32-stored ring spectra (storage_f32 = 2)
__global__ void __launch_bounds__(THREADS, 2)
k_sbw_mfma(const FT *__restrict__ Fl, double *__restrict__ B, const double *__restrict__ phi, const double *__restrict__ wq,
           const double *__restrict__ CB, int ncells, int V, int Zb, int K2, int64_t C, int cps) {
    constexpr int NG = THREADS / BW;                  // level groups: thread = (group g, block lb), levels z = g + NG i
    constexpr int ZPT = NZ / NG, KS = NZ / 4, LS = BW + 16;      // LS: the 4 levels a K step reads fall in disjoint bank halves
    constexpr int MTMAX = BW == 64 ? 4 : 6;           // row tiles of 16 modes: b_zDim <= 64 / <= 96
    __shared__ double As[NZ * LS];
    const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lb = threadIdx.x & (BW - 1);
    const int g = BW == 64 ? wv : (int)(threadIdx.x / BW);
    const int blk = blockIdx.x * BW + lb;
    const int v = blockIdx.y;
    const bool ok = blk < K2;
    const int ca = blockIdx.z * cps, cb = min(ca + cps, ncells);
    const int cend = (cb == ncells) ? ncells + 3 : cb;          // the last segment also owns the 3 trailing nodes
    const int cstart = max(0, ca - 3);
    const int64_t plane = (int64_t)V * NZ * K2;
    const FT *base = Fl + ((int64_t)v * NZ + g) * K2 + (ok ? blk : 0);
    // operator fragments: A[m][k] = CB[m][k], lane supplies m = 16 mt + (lane & 15), k = 4 js + (lane >> 4)
    const int MT = (Zb + 15) / 16, mhalf = (MT + 1) / 2;
    // this wave's column tile nt and row tiles mt0, mt1 (nmt of them)
    const int nt = BW == 64 ? (wv & 3) : (wv & 1);
    const int mt0 = BW == 64 ? (wv < 4 ? 0 : mhalf) : (wv >> 1);
    const int mt1 = BW == 64 ? mt0 + 1 : mt0 + THREADS / 128;      // BW = 32: two column tiles, the waves of a column tile share the row tiles
    const int nmt = BW == 64 ? (wv < 4 ? mhalf : MT - mhalf) : (mt0 >= MT ? 0 : mt1 < MT ? 2 : 1);
    const int n = lane & 15, kk = lane >> 4;
    // (kept in LDS in fragment order [row tile][K step][lane]: a conflict-free 8-byte read per MFMA; in registers the two
    // tiles' 64 VGPRs pushed the kernel into spills)
    __shared__ double Af[MTMAX * KS * 64];
    for (int e = threadIdx.x; e < MT * KS * 64; e += blockDim.x) {
        const int l = e & 63, js = (e >> 6) % KS, mt = e / (64 * KS);
        const int m = mt * 16 + (l & 15);
        Af[e] = (m < Zb) ? CB[(int64_t)m * NZ + 4 * js + (l >> 4)] : 0.0;
    }
    const double *af0 = Af + (size_t)(nmt > 0 ? mt0 : 0) * KS * 64 + lane, *af1 = Af + (size_t)(nmt > 1 ? mt1 : 0) * KS * 64 + lane;
    double acc[4][ZPT];
#pragma unroll
    for (int q = 0; q < 4; q++)
#pragma unroll
        for (int i = 0; i < ZPT; i++) acc[q][i] = 0.0;
    FT xn[MUBAR][ZPT];
    auto fetch = [&](int c) {
        if (c < cstart || c >= cend || c >= ncells) return;
#pragma unroll
        for (int mu = 0; mu < MUBAR; mu++) {
            const FT *src = base + (int64_t)(c * MUBAR + mu) * plane;
#pragma unroll
            for (int i = 0; i < ZPT; i++) xn[mu][i] = __builtin_nontemporal_load(src + (int64_t)(NG * i) * K2);
        }
    };
    fetch(cstart);
    SBW_T0();
    for (int c4 = cstart & ~3; c4 < cend; c4 += 4) {
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const int c = c4 + u;
            if (c < cstart || c >= cend) continue;
#ifdef SX_PHASES
            scells_++;
#endif
            if (c < ncells) {
#pragma unroll
                for (int mu = 0; mu < MUBAR; mu++) {
                    const int ring = c * MUBAR + mu;
                    const double w = wq[ring];
                    const double w0 = w * phi[(int64_t)ring * 4], w1 = w * phi[(int64_t)ring * 4 + 1];
                    const double w2 = w * phi[(int64_t)ring * 4 + 2], w3 = w * phi[(int64_t)ring * 4 + 3];
#pragma unroll
                    for (int i = 0; i < ZPT; i++) {
                        const double xv = (double)xn[mu][i];
                        acc[u][i] += w0 * xv;
                        acc[(u + 1) & 3][i] += w1 * xv;
                        acc[(u + 2) & 3][i] += w2 * xv;
                        acc[(u + 3) & 3][i] += w3 * xv;
                    }
                }
                fetch(c + 1);           // in flight while node c goes through LDS and the matrix cores
            }
            SBW_LAP(0);
            if (c >= ca) {                      // node c is complete: vertical forward transform and store
                __syncthreads();                // the previous node's tile has been consumed
                SBW_LAP(1);
#pragma unroll
                for (int i = 0; i < ZPT; i++) As[(g + NG * i) * LS + lb] = acc[u][i];
                __syncthreads();
                SBW_LAP(2);
                colmat_d4 o0 = {0.0, 0.0, 0.0, 0.0}, o1 = o0;
                const double *xb = As + kk * LS + nt * 16 + n;
                if (nmt > 0) {
#pragma unroll
                    for (int js = 0; js < KS; js++) {
                        const double b = xb[(4 * js) * LS];
                        o0 = __builtin_amdgcn_mfma_f64_16x16x4f64(af0[js * 64], b, o0, 0, 0, 0);
                        if (nmt > 1) o1 = __builtin_amdgcn_mfma_f64_16x16x4f64(af1[js * 64], b, o1, 0, 0, 0);
                    }
                }
                // D[row = kk + 4 r][col = n]
                const int col = blockIdx.x * BW + nt * 16 + n;
                if (col < K2) {
                    double *dst = B + (int64_t)c * C + (int64_t)v * Zb * K2 + col;
#pragma unroll
                    for (int r = 0; r < 4; r++) {
                        const int m0 = mt0 * 16 + kk + 4 * r, m1 = mt1 * 16 + kk + 4 * r;
                        if (nmt > 0 && m0 < Zb) dst[(int64_t)m0 * K2] = o0[r];
                        if (nmt > 1 && m1 < Zb) dst[(int64_t)m1 * K2] = o1[r];
                    }
                }
                SBW_LAP(3);
            }
#pragma unroll
            for (int i = 0; i < ZPT; i++) acc[u][i] = 0.0;      // the slot now belongs to node c + 4
        }
    }
    SBW_END();
}

// ------------------------------------------------------------------------------------------------ B -> A banded SPD solve
// One lane per right-hand side (column); rows are contiguous across lanes so every load/store is coalesced.
// a = Gamma^T (L L^T)^-1 Gamma b with L banded (half-bandwidth 3) plus, for PERIODIC, three dense last rows.
// A wave covers 64 consecutive wavenumber blocks of one (variable, z-mode): its boundary-condition class is
// wave-uniform, so the factor entries are scalar loads. The k = 0 column (its own class) is handled by one extra
// block per (variable, z-mode) in which only lane 0 works.
// scalar / pair arithmetic so that one kernel body serves a lane that owns one column (k = 0) or the (Re, Im) pair of a
// wavenumber (two independent right-hand sides moved as one 16-byte access)
struct S1 { double x; };
struct S2 { double x, y; };
__device__ __forceinline__ S1 ld(const double *p, S1 *) { return S1{p[0]}; }
__device__ __forceinline__ S2 ld(const double *p, S2 *) { const double2 v = *reinterpret_cast<const double2 *>(p); return S2{v.x, v.y}; }
__device__ __forceinline__ void st(double *p, S1 v) { p[0] = v.x; }
__device__ __forceinline__ void st(double *p, S2 v) { *reinterpret_cast<double2 *>(p) = make_double2(v.x, v.y); }
__device__ __forceinline__ S1 operator+(S1 a, S1 b) { return S1{a.x + b.x}; }
__device__ __forceinline__ S2 operator+(S2 a, S2 b) { return S2{a.x + b.x, a.y + b.y}; }
__device__ __forceinline__ S1 operator-(S1 a, S1 b) { return S1{a.x - b.x}; }
__device__ __forceinline__ S2 operator-(S2 a, S2 b) { return S2{a.x - b.x, a.y - b.y}; }
__device__ __forceinline__ S1 operator*(double c, S1 a) { return S1{c * a.x}; }
__device__ __forceinline__ S2 operator*(double c, S2 a) { return S2{c * a.x, c * a.y}; }
__device__ __forceinline__ S1 zero(S1 *) { return S1{0.0}; }
__device__ __forceinline__ S2 zero(S2 *) { return S2{0.0, 0.0}; }

// row m of the right-hand side from the LDS offset table: first source + (second source if the global table has one)
template <class T>
__device__ __forceinline__ T brow2(const double *__restrict__ B, const int64_t *sofs, int m, int64_t col) {
    const T x = ld(B + sofs[m * 4 + 0] + col, (T *)nullptr), y = ld(B + sofs[m * 4 + 1] + col, (T *)nullptr);
    return x + (sofs[m * 4 + 1] != sofs[m * 4 + 0] ? 1.0 : 0.0) * y;
}

// Row m of the right-hand side is Bsrc[boffA[m] + col] (+ Bsrc[boffB[m] + col] where a second tile overlaps, boffB >= 0);
// row m of the solution goes to A[aoffA[m] + col] and, in the final sweep, also to A[aoffB[m] + col] if aoffB >= 0.
// SOLVE_U rows per batch: 8 when the launch fills the chip (single-tile solve: bandwidth-bound, A/B 750 -> 755 steps/s),
// 16 for the transposed multi-GPU solve, whose few waves are latency-bound (8 there: 0.058 -> 0.098 ms)
template <class T, bool LINEAR, int SOLVE_U>
__device__ __forceinline__ void solve_columns(const double *__restrict__ Bsrc, const int64_t *__restrict__ boffA,
                                              const int64_t *__restrict__ boffB, double *__restrict__ A,
                                              const int64_t *__restrict__ aoffA, const int64_t *__restrict__ aoffB,
                                              const int *__restrict__ cmeta, const double *__restrict__ gl,
                                              const double *__restrict__ gr, const double *__restrict__ Lband,
                                              const double *__restrict__ Ldinv, const double *__restrict__ Larrow, int nb, int c,
                                              int64_t col, int64_t stride, bool active) {
    const int n = cmeta[c * 4 + 0], per = cmeta[c * 4 + 1], rl = cmeta[c * 4 + 2], rr = cmeta[c * 4 + 3];
    const double *Lb = Lband + (int64_t)c * nb * 4;
    const double *Ld = Ldinv + (int64_t)c * nb;
    // The factor rows (l0, l1, l2, 1 / diagonal) of this boundary-condition class go to LDS once per workgroup: as scalar
    // loads from memory they could not be fetched a batch ahead (16 rows x 5 doubles exceed the scalar registers), and
    // every couple of rows waited ~500 cycles for its own scalar load - the whole run time of the kernel.
    extern __shared__ double sfac[];              // [nb][4] factor rows, then (not LINEAR) [nb][4] row offsets
    int64_t *sofs = reinterpret_cast<int64_t *>(sfac + (size_t)nb * 4);
    for (int e = threadIdx.x; e < n; e += blockDim.x) {
        const double4 l4 = *reinterpret_cast<const double4 *>(Lb + (int64_t)e * 4);
        *reinterpret_cast<double4 *>(sfac + (size_t)e * 4) = make_double4(l4.x, l4.y, l4.z, Ld[e]);
    }
    if (!LINEAR)
        for (int e = threadIdx.x; e < nb; e += blockDim.x) {
            const int64_t o1 = boffA[e], o2 = boffB[e], a1 = aoffA[e], a2 = aoffB[e];
            // no second source / destination: repeat the first (its contribution is masked, the store is idempotent)
            sofs[e * 4 + 0] = o1; sofs[e * 4 + 1] = o2 >= 0 ? o2 : o1; sofs[e * 4 + 2] = a1; sofs[e * 4 + 3] = a2 >= 0 ? a2 : a1;
        }
    __syncthreads();
    if (!active) return;
    const double *La = Larrow + (int64_t)c * 3 * nb;
    const double *g_l = gl + c * 6, *g_r = gr + c * 6;
    T *tp = nullptr;
    // LINEAR: one contiguous [row][col] array on each side (row offset = m * stride), no offset tables to fetch
#define BROW(m) (LINEAR ? ld(Bsrc + (int64_t)(m) * stride + col, tp) : brow2<T>(Bsrc, sofs, (m), col))
#define AROW(m) ld(A + (LINEAR ? (int64_t)(m) * stride : sofs[(m) * 4 + 2]) + col, tp)
#define ASET(m, val) st(A + (LINEAR ? (int64_t)(m) * stride : sofs[(m) * 4 + 2]) + col, (val))
#define AFIN(m, val)                                                     \
    do {                                                                 \
        const T v_ = (val);                                              \
        st(A + (LINEAR ? (int64_t)(m) * stride : sofs[(m) * 4 + 2]) + col, v_);   \
        if (!LINEAR) st(A + sofs[(m) * 4 + 3] + col, v_);   /* no second destination: the same address again */ \
    } while (0)
    if (!per) {
        // forward substitution; the free unknown i lives in row rl + i of A
        T y1 = zero(tp), y2 = zero(tp), y3 = zero(tp);     // y[i-1], y[i-2], y[i-3]
        T bl0 = zero(tp), bl1 = zero(tp), br0 = zero(tp), br1 = zero(tp);
        for (int q = 0; q < rl; q++) { const T bq = BROW(q); bl0 = bl0 + g_l[q * 2] * bq; bl1 = bl1 + g_l[q * 2 + 1] * bq; }
        for (int q = 0; q < rr; q++) { const T bq = BROW(nb - 1 - q); br0 = br0 + g_r[q * 2] * bq; br1 = br1 + g_r[q * 2 + 1] * bq; }
        // Full batches of SOLVE_U rows run without any control flow (row index tests are selects), so the compiler hoists the
        // wave-uniform factor loads of a whole batch in front of its dependent multiply-adds; with a branch per row every row
        // waited for its own scalar loads (~700 cycles per row, the whole kernel).  The remainder rows take the simple loop.
        // interior rows: nothing but the three-term recurrence (7 multiply / add per column and row); the boundary-condition
        // contributions only touch the first and last two rows, which take the general form outside the batches
        auto fwd_plain = [&](int i, T s) {
            const double4 l = *reinterpret_cast<const double4 *>(sfac + (size_t)i * 4);
            s = s - (l.z * y1 + l.y * y2 + l.x * y3);
            s = l.w * s;
            y3 = y2; y2 = y1; y1 = s;
            ASET(rl + i, s);
        };
        auto fwd_edge = [&](int i, T s) {
            if (i == 0) s = s + bl0;
            if (i == 1) s = s + bl1;
            if (i == n - 1) s = s + br0;
            if (i == n - 2) s = s + br1;
            fwd_plain(i, s);
        };
        const int f_lo = min(2, n), f_hi = max(f_lo, n - 2);                      // interior rows [f_lo, f_hi)
        const int f_full = f_lo + ((f_hi - f_lo) / SOLVE_U) * SOLVE_U;
        for (int i = 0; i < f_lo; i++) fwd_edge(i, BROW(rl + i));
        // The rows of batch b + 1 are requested BEFORE batch b is computed and stored: the memory counter of this hardware
        // retires loads and stores in issue order, so a load issued after a batch's stores would also wait for those
        // stores to complete (and its own latency would be exposed once per batch).
        {
            T rhs[SOLVE_U], nxt[SOLVE_U];
            if (f_lo < f_full) {
#pragma unroll
                for (int u = 0; u < SOLVE_U; u++) rhs[u] = BROW(rl + f_lo + u);
            }
            for (int i0 = f_lo; i0 < f_full; i0 += SOLVE_U) {
                const bool more = i0 + SOLVE_U < f_full;
#pragma unroll
                for (int u = 0; u < SOLVE_U; u++) nxt[u] = BROW(rl + (more ? i0 + SOLVE_U + u : i0 + u));
#pragma unroll
                for (int u = 0; u < SOLVE_U; u++) fwd_plain(i0 + u, rhs[u]);
#pragma unroll
                for (int u = 0; u < SOLVE_U; u++) rhs[u] = nxt[u];
            }
        }
        for (int i = f_full; i < n; i++) fwd_edge(i, BROW(rl + i));
        // back substitution
        T x1 = zero(tp), x2 = zero(tp), x3 = zero(tp);     // x[i+1], x[i+2], x[i+3]
        T xl0 = zero(tp), xl1 = zero(tp), xr0 = zero(tp), xr1 = zero(tp);
        auto bwd_plain = [&](int i, T s) {                 // rows i <= n - 4: all three super-diagonal terms exist
            s = s - (sfac[(size_t)(i + 1) * 4 + 2] * x1 + sfac[(size_t)(i + 2) * 4 + 1] * x2 + sfac[(size_t)(i + 3) * 4 + 0] * x3);
            s = sfac[(size_t)i * 4 + 3] * s;
            x3 = x2; x2 = x1; x1 = s;
            AFIN(rl + i, s);
        };
        auto bwd_edge = [&](int i, T s) {
            if (i + 1 < n) s = s - Lb[(int64_t)(i + 1) * 4 + 2] * x1;
            if (i + 2 < n) s = s - Lb[(int64_t)(i + 2) * 4 + 1] * x2;
            if (i + 3 < n) s = s - Lb[(int64_t)(i + 3) * 4 + 0] * x3;
            s = Ld[i] * s;
            x3 = x2; x2 = x1; x1 = s;
            AFIN(rl + i, s);
            if (i == n - 1) xr0 = s;
            if (i == n - 2) xr1 = s;
            if (i == 1) xl1 = s;
            if (i == 0) xl0 = s;
        };
        int ib = n - 1;
        for (; ib >= max(n - 3, 0); ib--) bwd_edge(ib, AROW(rl + ib));               // last three rows
        {                                                                            // interior rows down to row 2
            T rhs[SOLVE_U], nxt[SOLVE_U];
            if (ib >= SOLVE_U + 1) {
#pragma unroll
                for (int u = 0; u < SOLVE_U; u++) rhs[u] = AROW(rl + ib - u);
            }
            for (; ib >= SOLVE_U + 1; ib -= SOLVE_U) {
                const bool more = ib - SOLVE_U >= SOLVE_U + 1;
#pragma unroll
                for (int u = 0; u < SOLVE_U; u++) nxt[u] = AROW(rl + (more ? ib - SOLVE_U - u : ib - u));
#pragma unroll
                for (int u = 0; u < SOLVE_U; u++) bwd_plain(ib - u, rhs[u]);
#pragma unroll
                for (int u = 0; u < SOLVE_U; u++) rhs[u] = nxt[u];
            }
        }
        for (; ib >= 0; ib--) bwd_edge(ib, AROW(rl + ib));
        for (int q = 0; q < rl; q++) AFIN(q, g_l[q * 2] * xl0 + g_l[q * 2 + 1] * xl1);
        for (int q = 0; q < rr; q++) AFIN(nb - 1 - q, g_r[q * 2] * xr0 + g_r[q * 2 + 1] * xr1);
    } else {
        // periodic: unknown i <-> row i + 1; rows 0, nb-2, nb-1 fold onto unknowns n-1, 0, 1
        T y1 = zero(tp), y2 = zero(tp), y3 = zero(tp);
        T acc0 = zero(tp), acc1 = zero(tp), acc2 = zero(tp);        // arrow-row dot products
        for (int i = 0; i < n - 3; i++) {
            T s = BROW(i + 1);
            if (i == 0) s = s + BROW(nb - 2);
            if (i == 1) s = s + BROW(nb - 1);
            const double *l = Lb + (int64_t)i * 4;
            s = s - (l[2] * y1 + l[1] * y2 + l[0] * y3);
            s = Ld[i] * s;
            y3 = y2; y2 = y1; y1 = s;
            ASET(i + 1, s);
            acc0 = acc0 + La[i] * s;
            acc1 = acc1 + La[nb + i] * s;
            acc2 = acc2 + La[2 * nb + i] * s;
        }
        const T t0 = (1.0 / La[n - 3]) * (BROW(n - 2) - acc0);
        const T t1 = (1.0 / La[nb + n - 2]) * (BROW(n - 1) - acc1 - La[nb + n - 3] * t0);
        const T t2 = (1.0 / La[2 * nb + n - 1]) * (BROW(n) + BROW(0) - acc2 - La[2 * nb + n - 3] * t0 - La[2 * nb + n - 2] * t1);
        // back substitution of the dense 3x3 corner
        const T u2 = (1.0 / La[2 * nb + n - 1]) * t2;
        const T u1 = (1.0 / La[nb + n - 2]) * (t1 - La[2 * nb + n - 2] * u2);
        const T u0 = (1.0 / La[n - 3]) * (t0 - La[nb + n - 3] * u1 - La[2 * nb + n - 3] * u2);
        AFIN(n - 2, u0);
        AFIN(n - 1, u1);
        AFIN(n, u2);
        T x1 = zero(tp), x2 = zero(tp), x3 = zero(tp);
        T first0 = zero(tp), first1 = zero(tp);
        for (int i = n - 4; i >= 0; i--) {
            T s = AROW(i + 1);
            if (i + 1 < n - 3) s = s - Lb[(int64_t)(i + 1) * 4 + 2] * x1;
            if (i + 2 < n - 3) s = s - Lb[(int64_t)(i + 2) * 4 + 1] * x2;
            if (i + 3 < n - 3) s = s - Lb[(int64_t)(i + 3) * 4 + 0] * x3;
            s = s - (La[i] * u0 + La[nb + i] * u1 + La[2 * nb + i] * u2);
            s = Ld[i] * s;
            x3 = x2; x2 = x1; x1 = s;
            AFIN(i + 1, s);
            if (i == 0) first0 = s;
            if (i == 1) first1 = s;
        }
        AFIN(0, u2);            // a_{-1} = a_{n-1}
        AFIN(nb - 2, first0);   // a_{n}  = a_0
        AFIN(nb - 1, first1);   // a_{n+1} = a_1
    }
#undef BROW
#undef AROW
#undef ASET
#undef AFIN
}

// One lane per wavenumber: the (Re, Im) columns of a wavenumber k >= 1 form one 16-byte aligned pair and share a
// boundary-condition class, so a lane solves both with double2 loads/stores; rows are contiguous across lanes, so every
// access is coalesced. The k = 0 column (own class, single column) is handled by one extra block per (variable,
// z-mode) in which only lane 0 works.
// PAIR = false: one column per lane (twice the waves, half the dependent arithmetic per row): used when the launch has too
// few wavenumbers to occupy the chip - the transposed solve of a multi-GPU run - where the kernel time is the latency
// of one wave's row recurrence.
template <bool LINEAR, bool PAIR = true, int SOLVE_U = LINEAR ? 8 : 16>
__global__ void __launch_bounds__(64)
k_solve(const double *__restrict__ Bsrc, const int64_t *__restrict__ boffA, const int64_t *__restrict__ boffB,
        double *__restrict__ A, const int64_t *__restrict__ aoffA, const int64_t *__restrict__ aoffB,
        const int *__restrict__ cls, const int *__restrict__ cmeta, const double *__restrict__ gl,
        const double *__restrict__ gr, const double *__restrict__ Lband, const double *__restrict__ Ldinv,
        const double *__restrict__ Larrow, int nb, int Zb, int K2, int vz0, int64_t stride) {
    const int vz = blockIdx.y;                      // local (v, zm) group; vz0 + vz is the patch-level group
    const int v = (vz0 + vz) / Zb;
    const bool k0 = (blockIdx.x == gridDim.x - 1);  // the last block in x handles the k = 0 column
    // every lane takes part in staging the factor rows; lanes without a column leave after that (`active`)
    if (k0) {
        solve_columns<S1, LINEAR, SOLVE_U>(Bsrc, boffA, boffB, A, aoffA, aoffB, cmeta, gl, gr, Lband, Ldinv, Larrow, nb, cls[v * 2 + 0],
                                  (int64_t)vz * K2, stride, threadIdx.x == 0);
    } else if (PAIR) {
        const int k = 1 + blockIdx.x * 64 + threadIdx.x;      // wavenumber; its columns are blocks 2k and 2k + 1
        const bool act = 2 * k + 1 < K2;
        solve_columns<S2, LINEAR, SOLVE_U>(Bsrc, boffA, boffB, A, aoffA, aoffB, cmeta, gl, gr, Lband, Ldinv, Larrow, nb, cls[v * 2 + 1],
                                  (int64_t)vz * K2 + (act ? 2 * k : 2), stride, act);
    } else {
        const int c = 2 + blockIdx.x * 64 + threadIdx.x;      // column (Re or Im of a wavenumber >= 1)
        const bool act = c < K2;
        solve_columns<S1, LINEAR, SOLVE_U>(Bsrc, boffA, boffB, A, aoffA, aoffB, cmeta, gl, gr, Lband, Ldinv, Larrow, nb, cls[v * 2 + 1],
                                  (int64_t)vz * K2 + (act ? c : 2), stride, act);
    }
}

// Transposed (all-to-all) patch solve, tile side: split the tile's [row][col] arrays by destination column range.
//   pack:   send[soff[d] + j * cw[d] + (col - cs[d])] = B[j][col]
//   unpack: A[(cell0 + j)][col] = recv[soff[d] + j * cw[d] + (col - cs[d])]        d = owner of col's (v, z-mode) group
__global__ void k_a2a_pack(const double *__restrict__ B, double *__restrict__ send, const int *__restrict__ owner,
                           const int64_t *__restrict__ soff, const int64_t *__restrict__ cw, const int64_t *__restrict__ cs,
                           int K2, int64_t C, int unpack, int64_t brow0) {
    const int64_t col = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int j = blockIdx.y;
    if (col >= C) return;
    const int d = owner[col / K2];
    const int64_t o = soff[d] + (int64_t)j * cw[d] + (col - cs[d]);
    if (unpack) const_cast<double *>(B)[(brow0 + j) * C + col] = send[o];
    else send[o] = B[(brow0 + j) * C + col];
}

__global__ void k_halo_add(double *__restrict__ B, const double *__restrict__ recv, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) B[i] += recv[i];
}

__global__ void k_nan_check(const double *__restrict__ x, int64_t n, int *flag) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    int bad = 0;
    for (; i < n; i += stride) bad |= (x[i] != x[i]);
    if (bad) atomicOr(flag, 1);
}

// max |x[v][p]| per variable: the bit pattern of a non-negative double orders like an unsigned integer, and every NaN
// pattern (sign cleared) orders above +Inf - so a NaN anywhere in the field comes out as NaN, as Julia's maximum(abs, x) does
__global__ void k_max_abs(const double *__restrict__ x, int64_t N, unsigned long long *__restrict__ out) {
    const int v = blockIdx.y;
    const double *xv = x + (int64_t)v * N;
    unsigned long long m = 0ull;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < N; i += (int64_t)gridDim.x * blockDim.x) {
        const unsigned long long a = (unsigned long long)__double_as_longlong(xv[i]) & 0x7fffffffffffffffull;
        if (a > m) m = a;
    }
    for (int o = 32; o > 0; o >>= 1) {
        const unsigned long long t = __shfl_xor(m, o);
        if (t > m) m = t;
    }
    if ((threadIdx.x & 63) == 0) atomicMax(out + v, m);
}

// ------------------------------------------------------------------------------------------------ equation sets
template <class ST>
struct PhysArgsT {
    Planes<ST> P;         // physical
    double *En;           // expdot_n  [V][N]
    double *E1, *E2;      // expdot_nm1 / nm2 (read)
    double *In;           // impdot_n
    double *np1;          // var_np1
    const double *r, *cosl, *sinl, *z;
    const double *MintT, *MdzT;
    int64_t N;
    int V, nz, t, eq;
    int s_u, s_r, s_rr, s_l, s_ll, s_z, s_zz;
    double ts;
    double par[SX_NPARAMS];
    // column range of this launch and, for the node-space variant, the node transforms G [slot][v][NG] + basis weights
    int64_t col0, col1;
    Planes<ST> G;
    const double *phi;
    const double *ref;    // ReferenceState [3][3][nz] (Euler_test)
    int write_w;          // store the diagnostic w into physical[:, 6, 1] (src/shallowWaterModels.jl:66-67, 426-429): only the
                          // stand-alone sx_physics needs it there; inside sx_advance nothing reads that plane again
    int64_t NG;
    int L, nrings;
    // cell-independent constants of the cell-wise kernel (scalar registers, no loads): basis weights phi / phi' / phi'' of a
    // cell's 3 Gauss points at its 4 nodes, and what it takes to recompute r exactly as sx_create tabulates it
    double phiw[3][MUBAR][4];
    double xmin, DX, goff[MUBAR];
    int gcell0;           // patch index of the tile's first cell
    long long *dbg;       // phase stamps [workgroup][8] of the diagnostic build (-DSX_PHASES, profiles/phases.sh); otherwise null
};

// A diagnostic variable has expdot == 0 for ever (src/shallowWaterModels.jl:69, 185, 430): explicit_timestep reduces to
// var_np1 = value, and its (all-zero) tendency history is neither read nor written.
template <class A>
__device__ __forceinline__ void diag_step(const A &a, int v, int64_t p, double u) { a.np1[(int64_t)v * a.N + p] = u; }

// explicit_timestep (src/semiimplicit.jl:672-698); history arrays are rotated by the host instead of copied
template <class A>
__device__ __forceinline__ double ab_step(const A &a, int v, int64_t p, double u, double en) {
    const int64_t o = (int64_t)v * a.N + p;
    a.En[o] = en;
    double un;
    if (a.t == 1) un = u + (a.ts * en);
    else if (a.t == 2) un = u + (0.5 * a.ts) * ((3.0 * en) - a.E1[o]);
    else un = u + ((a.ts / 12.0) * ((23.0 * en) - (16.0 * a.E1[o]) + (5.0 * a.E2[o])));
    a.np1[o] = un;
    return un;
}

// value of variable v / derivative slot s (>= 1) of variable v at point p
#define PSV(v) a.P.val[(int64_t)(v) * a.N + p]
// Moist thermodynamics of Euler_test (src/thermodynamics.jl; constants :2-17, :31-32)
namespace thermo {
constexpr double Rd = 287.04, Rv = 461.50, Cvd = 716.96, Cvv = 1410.0, gravity = 9.81, L_v0 = 2.501e6, T_0 = 273.16, p_0 = 1000.0,
                 q0 = 1.0e-7;
constexpr double rho_d0 = 100.0 * p_0 / (T_0 * Rd);
// rho_v0 = 100 sat_pressure_liquid(T_0) / (T_0 Rv), sat_pressure_liquid(T) = 6.112 exp(17.67 Tc / (Tc + 243.5)) (:19-23, :32)
__device__ __forceinline__ double rho_v0() { const double Tc = T_0 - 273.15; return 100.0 * (6.112 * exp(17.67 * Tc / (Tc + 243.5))) / (T_0 * Rv); }
__device__ __forceinline__ double ahyp(double mu) { return mu < 0.0 ? 0.0 : sqrt(mu * mu + q0 * q0) + mu - q0; }            // :190-198
__device__ __forceinline__ double dmudq(double mu, double q_v) { return ((q_v + q0) - mu) / (q_v + q0); }                    // :200-203
__device__ __forceinline__ double dry_density(double xi) { return rho_d0 * exp(xi); }                                        // :205-208
__device__ __forceinline__ double temperature(double s, double rho_d, double q_v) {                                          // :67-80
    const double Cf = Cvd + (q_v * Cvv);
    double qf = 1.0;
    if (q_v != 0.0) qf = pow(rho_d * q_v / rho_v0(), (q_v * Rv) / Cf);
    const double rf = pow(rho_d / rho_d0, Rd / Cf);
    const double Tf = exp((s - (q_v * L_v0 / T_0)) / Cf);
    return T_0 * Tf * rf * qf;
}
__device__ __forceinline__ double P_s(double Tk, double rho_d, double q_v) {                                                 // :215-219
    return Tk * ((rho_d * Rd) + (q_v * rho_d * Rv)) / (Cvd + (q_v * Cvv));
}
__device__ __forceinline__ double P_xi(double Tk, double rho_d, double q_v) {                                                // :221-224
    return (Rd + (q_v * rho_d * Rv)) * ((rho_d * Tk) + P_s(Tk, rho_d, q_v));
}
__device__ __forceinline__ double P_qv(double Tk, double rho_d, double q_v) {                                                // :232-242
    if (q_v == 0.0) return 0.0;
    const double rho_v = q_v * rho_d;
    double qf = Rv * (1 + log(rho_v / rho_v0())) - (Cvv * log(Tk / T_0)) - L_v0 / T_0;
    qf *= P_s(Tk, rho_d, q_v);
    return (rho_d * Rv * Tk) + qf;
}
__device__ __forceinline__ double pressure_gradient(double Tk, double rho_d, double q_v, double s_x, double xi_x, double qv_x) {   // :250-258
    return (P_s(Tk, rho_d, q_v) * s_x) + (P_xi(Tk, rho_d, q_v) * xi_x) + (P_qv(Tk, rho_d, q_v) * qv_x);
}
}  // namespace thermo

#define PS(v, s) ((double)a.P.der[((int64_t)((s) - 1) * a.V + (v)) * a.N + p])

template <class ST>
__global__ void k_phys_pointwise(PhysArgsT<ST> a) {
    const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= a.N) return;
    const double *par = a.par;
    const double r = a.r[p / a.nz];
    switch (a.eq) {
        case SX_EQ_NONE:
            for (int v = 0; v < a.V; v++) a.np1[(int64_t)v * a.N + p] = PSV(v);
            return;
        case SX_EQ_LINEAR_ADVECTION_1D: {      // src/testModels.jl:15
            const double e = -(par[SX_P_C0] * PS(0, a.s_r)) + (par[SX_P_K] * PS(0, a.s_rr));
            ab_step(a, 0, p, PSV(0), e);
            for (int v = 1; v < a.V; v++) ab_step(a, v, p, PSV(v), 0.0);
        } break;
        case SX_EQ_LINEAR_ADVECTION_RZ: {      // src/testModels.jl:40
            const double hr = PS(0, a.s_r);
            const double e = (-PSV(1) * hr) + (-PSV(3) * PS(0, a.s_z)) +
                             (par[SX_P_K] * ((hr / r) + PS(0, a.s_rr) + PS(0, a.s_zz)));
            ab_step(a, 0, p, PSV(0), e);
            for (int v = 1; v < a.V; v++) ab_step(a, v, p, PSV(v), 0.0);
        } break;
        case SX_EQ_LINEAR_ADVECTION_RL:        // src/testModels.jl:62-68
        case SX_EQ_LINEAR_ADVECTION_RLZ: {     // src/testModels.jl:93
            const double hr = PS(0, a.s_r), hl = PS(0, a.s_l);
            double e = (-PSV(1) * hr) - (PSV(2) * (hl / r));
            if (a.eq == SX_EQ_LINEAR_ADVECTION_RLZ || par[SX_P_K] > 0.0)
                e += par[SX_P_K] * ((hr / r) + PS(0, a.s_rr) + (PS(0, a.s_ll) / (r * r)));
            ab_step(a, 0, p, PSV(0), e);
            for (int v = 1; v < a.V; v++) ab_step(a, v, p, PSV(v), 0.0);
        } break;
        case SX_EQ_ONEWAY_SW_SLAB:             // src/shallowWaterModels.jl:60-108
        case SX_EQ_TWOWAY_SW_SLAB: {           // src/shallowWaterModels.jl:176-228
            const double g = par[SX_P_G], K = par[SX_P_K], Cd = par[SX_P_CD], Hfree = par[SX_P_HFREE],
                         Hb = par[SX_P_HB], f = par[SX_P_F];
            const double h = PSV(0), hr = PS(0, a.s_r), hl = PS(0, a.s_l);
            const double ug = PSV(1), ugr = PS(1, a.s_r), ugl = PS(1, a.s_l);
            const double vg = PSV(2), vgr = PS(2, a.s_r), vgl = PS(2, a.s_l);
            const double ub = PSV(3), ubr = PS(3, a.s_r), ubrr = PS(3, a.s_rr), ubl = PS(3, a.s_l), ubll = PS(3, a.s_ll);
            const double vb = PSV(4), vbr = PS(4, a.s_r), vbrr = PS(4, a.s_rr), vbl = PS(4, a.s_l), vbll = PS(4, a.s_ll);
            const double U = 0.78 * sqrt((ub * ub) + (vb * vb));
            const double w = -Hb * ((ub / r) + ubr + (vbl / r));
            if (a.write_w) a.P.val[(int64_t)5 * a.N + p] = w;
            const double w_ = 0.5 * fabs(w) - w;
            double e0 = ((-vg * hl / r) + (-ug * hr)) + (-(Hfree + h) * ((ug / r) + ugr + (vgl / r)));
            if (a.eq == SX_EQ_TWOWAY_SW_SLAB) e0 += -(Hfree + h) * w * par[SX_P_S1];
            const double e1 = ((-vg * ugl / r) + (-ug * ugr)) + (-g * hr) + (vg * (f + (vg / r)));
            const double e2 = ((-vg * vgl / r) + (-ug * vgr)) + (-g * (hl / r)) + (-ug * (f + (vg / r)));
            const double e3 = ((-vb * ubl / r) + (-ub * ubr)) + (-g * hr) + (vb * (f + (vb / r))) + (-(Cd * U * ub / Hb)) +
                              (w_ * (ug - ub) / Hb) +
                              (K * ((ubr / r) + ubrr - (ub / (r * r)) + (ubll / (r * r)) - (2.0 * vbl / (r * r))));
            const double e4 = ((-vb * vbl / r) + (-ub * vbr)) + (-g * (hl / r)) + (-ub * (f + (vb / r))) + (-(Cd * U * vb / Hb)) +
                              (w_ * (vg - vb) / Hb) +
                              (K * ((vbr / r) + vbrr - (vb / (r * r)) + (vbll / (r * r)) + (2.0 * ubl / (r * r))));
            ab_step(a, 0, p, h, e0);
            ab_step(a, 1, p, ug, e1);
            ab_step(a, 2, p, vg, e2);
            ab_step(a, 3, p, ub, e3);
            ab_step(a, 4, p, vb, e4);
            diag_step(a, 5, p, w);
            for (int v = 6; v < a.V; v++) ab_step(a, v, p, PSV(v), 0.0);
        } break;
        case SX_EQ_LINEAR_ACOUSTIC_RZ: {
            const double K = par[SX_P_K], pxi = par[SX_P_PXI_BAR];
            const double u = PSV(3), w = PSV(4);
            double e[5];
            for (int v = 0; v < 5; v++) e[v] = (-u * PS(v, a.s_r)) + (-w * PS(v, a.s_z));
            const double d0 = K * (PS(0, a.s_rr) + PS(0, a.s_zz)), d2 = K * (PS(2, a.s_rr) + PS(2, a.s_zz));
            const double d3 = K * (PS(3, a.s_rr) + PS(3, a.s_zz)), d4 = K * (PS(4, a.s_rr) + PS(4, a.s_zz));
            const double xir = PS(1, a.s_r), xiz = PS(1, a.s_z), wz = PS(4, a.s_z);
            e[0] = e[0] + d0;
            e[1] = e[1] - PS(3, a.s_r) - wz;
            e[2] = e[2] + d2;
            e[3] = e[3] + (-(pxi * xir)) + d3;
            e[4] = e[4] + (-(pxi * xiz)) + d4;
            for (int v = 0; v < 5; v++) {
                ab_step(a, v, p, PSV(v), e[v]);
                if (a.In) a.In[(int64_t)v * a.N + p] = (v == 1) ? -wz : (v == 4) ? -(pxi * xiz) : 0.0;
            }
        } break;
        case SX_EQ_EULER_TEST: {               // src/testModels.jl:100-215
            const double K = par[SX_P_K], pxi = par[SX_P_PXI_BAR];
            const int k = (int)(p % a.nz), nz = a.nz;
            // ReferenceState rows: sbar, sbar_z, sbar_zz, xibar, xibar_z, xibar_zz, mubar, mubar_z, mubar_zz
            const double sbar = a.ref[k], sbar_z = a.ref[nz + k], xibar = a.ref[3 * nz + k], xibar_z = a.ref[4 * nz + k];
            const double mubar = a.ref[6 * nz + k], mubar_z = a.ref[7 * nz + k];
            const double s_x = PS(0, a.s_r), s_z = PS(0, a.s_z), xi_x = PS(1, a.s_r), xi_z = PS(1, a.s_z);
            const double mu = PSV(2), mu_x = PS(2, a.s_r), mu_z = PS(2, a.s_z);
            const double u = PSV(3), u_x = PS(3, a.s_r), u_z = PS(3, a.s_z), w = PSV(4), w_x = PS(4, a.s_r), w_z = PS(4, a.s_z);
            const double q_v = thermo::ahyp(mu + mubar);
            const double rho_d = thermo::dry_density(PSV(1) + xibar);
            const double Tk = thermo::temperature(PSV(0) + sbar, rho_d, q_v);
            const double rho_t = rho_d * (1.0 + q_v);
            const double dm = thermo::dmudq(mu + mubar, q_v);
            const double qvp_x = mu_x / dm, qvp_z = mu_z / dm;
            const double rhobar = thermo::dry_density(xibar) * (1.0 + thermo::ahyp(mubar));
            const double rho_p = rho_t - rhobar;
            double e[5];
            e[0] = ((-u * s_x) + (-w * (s_z + sbar_z))) + (K * (PS(0, a.s_rr) + PS(0, a.s_zz)));
            e[1] = ((-u * xi_x) + (-w * (xi_z + xibar_z))) - u_x - w_z;
            e[2] = ((-u * mu_x) + (-w * (mu_z + mubar_z))) + (K * (PS(2, a.s_rr) + PS(2, a.s_zz)));
            e[3] = ((-u * u_x) + (-w * u_z)) + (-(thermo::pressure_gradient(Tk, rho_d, q_v, s_x, xi_x, qvp_x) / rho_t)) +
                   (K * (PS(3, a.s_rr) + PS(3, a.s_zz)));
            e[4] = ((-u * w_x) + (-w * w_z)) +
                   (-(thermo::gravity * rho_p / rho_t) - (thermo::pressure_gradient(Tk, rho_d, q_v, s_z, xi_z, qvp_z) / rho_t)) +
                   (K * (PS(4, a.s_rr) + PS(4, a.s_zz)));
            for (int v = 0; v < 5; v++) {
                ab_step(a, v, p, PSV(v), e[v]);
                if (a.In) a.In[(int64_t)v * a.N + p] = (v == 1) ? -w_z : (v == 4) ? -(pxi * xi_z) : 0.0;      // impdot: only kept when semi-implicit
            }
            for (int v = 5; v < a.V; v++) ab_step(a, v, p, PSV(v), 0.0);
        } break;
        default: break;
    }
}

// Oneway_ShallowWater_HeightResolvedBL (src/shallowWaterModels.jl:346-511). One workgroup handles `cpb` columns;
// thread (c, k) owns level k of column c. The three per-column Chebyshev operators (integral of the divergence,
// derivative of the two vertical fluxes) are dense nz x nz mat-vecs with the operands staged in LDS.
template <class ST>
__global__ void __launch_bounds__(256) k_phys_hrbl(PhysArgsT<ST> a, int cpb) {
    extern __shared__ double sm[];
    const int nz = a.nz;
    const int k = threadIdx.x % nz, cl = threadIdx.x / nz;
    const int64_t col = (int64_t)blockIdx.x * cpb + cl;
    const int64_t ncol = a.N / nz;
    const bool live = (cl < cpb) && (col < ncol);
    double *sdiv = sm, *sfu = sm + (size_t)cpb * nz, *sfv = sm + (size_t)2 * cpb * nz;
    double *sub = sm + (size_t)3 * cpb * nz, *svb = sm + (size_t)4 * cpb * nz;
    const double *par = a.par;
    const double g = par[SX_P_G], Kh = par[SX_P_KH], Hfree = par[SX_P_HFREE], f = par[SX_P_F];
    const int64_t p = live ? col * nz + k : 0;
    double r = 1.0, h = 0, hr = 0, hl = 0, ug = 0, ugr = 0, ugl = 0, vg = 0, vgr = 0, vgl = 0;
    double ub = 0, ubr = 0, ubrr = 0, ubl = 0, ubll = 0, ubz = 0, vb = 0, vbr = 0, vbrr = 0, vbl = 0, vbll = 0, vbz = 0;
    if (live) {
        r = a.r[col];
        h = PSV(0); hr = PS(0, a.s_r); hl = PS(0, a.s_l);
        ug = PSV(1); ugr = PS(1, a.s_r); ugl = PS(1, a.s_l);
        vg = PSV(2); vgr = PS(2, a.s_r); vgl = PS(2, a.s_l);
        ub = PSV(3); ubr = PS(3, a.s_r); ubrr = PS(3, a.s_rr); ubl = PS(3, a.s_l); ubll = PS(3, a.s_ll); ubz = PS(3, a.s_z);
        vb = PSV(4); vbr = PS(4, a.s_r); vbrr = PS(4, a.s_rr); vbl = PS(4, a.s_l); vbll = PS(4, a.s_ll); vbz = PS(4, a.s_z);
        const double S = sqrt((ubz * ubz) + (vbz * vbz));
        const double l = 1.0 / ((1.0 / (0.4 * a.z[k])) + (1.0 / 80.0));
        const double Kv = (l * l) * S;
        sdiv[cl * nz + k] = -((ub / r) + ubr + (vbl / r));
        sfu[cl * nz + k] = Kv * ubz;
        sfv[cl * nz + k] = Kv * vbz;
        sub[cl * nz + k] = ub;
        svb[cl * nz + k] = vb;
    }
    __syncthreads();
    if (live && k == 0) {
        const double Um = par[SX_P_UM], Vm = par[SX_P_VM];
        const double cs = a.cosl[col], sn = a.sinl[col];
        const double sfcu = (Um * cs) + (Vm * sn), sfcv = (Vm * cs) - (Um * sn);
        const double u10 = sub[cl * nz + 1] + sfcu, v10 = svb[cl * nz + 1] + sfcv;
        const double U10 = sqrt(u10 * u10 + v10 * v10);
        double Cd = par[SX_P_CD];
        if (U10 < 5.2) Cd = 1.0e-3;
        else if (U10 < 33.6) Cd = 4.4e-4 * sqrt(U10);
        sfu[cl * nz] = Cd * U10 * u10;
        sfv[cl * nz] = Cd * U10 * v10;
    }
    __syncthreads();
    if (!live) return;
    double wb = 0.0, vdu = 0.0, vdv = 0.0;
    const double *xd = sdiv + cl * nz, *xu = sfu + cl * nz, *xv = sfv + cl * nz;
    for (int j = 0; j < nz; j++) {
        const double mi = a.MintT[(int64_t)j * nz + k], md = a.MdzT[(int64_t)j * nz + k];
        wb += mi * xd[j];
        vdu += md * xu[j];
        vdv += md * xv[j];
    }
    if (a.write_w) a.P.val[(int64_t)5 * a.N + p] = wb;
    const double e0 = ((-vg * hl / r) + (-ug * hr)) + (-(Hfree + h) * ((ug / r) + ugr + (vgl / r)));
    const double e1 = ((-vg * ugl / r) + (-ug * ugr)) + (-g * hr) + (vg * (f + (vg / r)));
    const double e2 = ((-vg * vgl / r) + (-ug * vgr)) + (-g * (hl / r)) + (-ug * (f + (vg / r)));
    const double e3 = ((-vb * ubl / r) + (-ub * ubr) + (-wb * ubz)) + (-g * hr) + (vb * (f + (vb / r))) + vdu +
                      (Kh * ((ubr / r) + ubrr - (ub / (r * r)) + (ubll / (r * r)) - (2.0 * vbl / (r * r))));
    const double e4 = ((-vb * vbl / r) + (-ub * vbr) + (-wb * vbz)) + (-g * (hl / r)) + (-ub * (f + (vb / r))) + vdv +
                      (Kh * ((vbr / r) + vbrr - (vb / (r * r)) + (vbll / (r * r)) + (2.0 * ubl / (r * r))));
    ab_step(a, 0, p, h, e0);
    ab_step(a, 1, p, ug, e1);
    ab_step(a, 2, p, vg, e2);
    ab_step(a, 3, p, ub, e3);
    ab_step(a, 4, p, vb, e4);
    diag_step(a, 5, p, wb);
    for (int v = 6; v < a.V; v++) ab_step(a, v, p, PSV(v), 0.0);
}

// MFMA variant of the same equation set for zDim = NZ (multiple of 16): 16 columns per workgroup. The three column
// operators are genuine contractions  Y[NZ x 16] = M[NZ x NZ] * X[NZ x 16]  and run on the f64 matrix cores
// (v_mfma_f64_16x16x4_f64): wave w < 12 owns (operand w / 4, row tile w % 4); A comes straight from the
// (L2-resident) operator, B and the result tiles live in LDS, column-major with a 2-double pad (bank-conflict free).
typedef double mfma_d4 __attribute__((ext_vector_type(4)));

// 16 bytes per lane for streams that are 8 bytes per point.  A wave owns 64 consecutive doubles of every stream; lanes
// 0-31 fetch TWO consecutive elements of stream a, lanes 32-63 of stream b (one global_load_dwordx4 instead of two
// dwordx2), and one v_permlane32_swap per dword leaves (a[e], b[e]) in every lane with e = 2 (lane & 31) + (lane >> 5) -
// which is therefore the element (level) a lane works on.  Stores run the same exchange backwards.  The load and the
// exchange are SEPARATE steps (RawPair): an exchange right behind its load makes the wave wait for that load alone, and a
// handful of such round trips in a row was most of this kernel's time (phase stamps, profiles/r02/phases_*.txt).
typedef double dbl2v __attribute__((ext_vector_type(2)));
typedef float flt2v __attribute__((ext_vector_type(2)));
template <class T> struct Vec2;
template <> struct Vec2<double> { typedef dbl2v type; };
template <> struct Vec2<float> { typedef flt2v type; };
__device__ __forceinline__ int wide_elem(int lane) { return 2 * (lane & 31) + (lane >> 5); }

// pa / pb point at the lane's pair of stream a / b
template <bool NT, class T>
__device__ __forceinline__ typename Vec2<T>::type issue_pair(const T *pa, const T *pb, int lane) {
    typedef typename Vec2<T>::type V;
    const V *p = reinterpret_cast<const V *>(lane < 32 ? pa : pb);
    return NT ? __builtin_nontemporal_load(p) : *p;
}
__device__ __forceinline__ void take_pair(dbl2v t, double &xa, double &xb) {
    const auto r0 = __builtin_amdgcn_permlane32_swap(__double2loint(t.x), __double2loint(t.y), false, false);
    const auto r1 = __builtin_amdgcn_permlane32_swap(__double2hiint(t.x), __double2hiint(t.y), false, false);
    xa = __hiloint2double(r1[0], r0[0]);
    xb = __hiloint2double(r1[1], r0[1]);
}
__device__ __forceinline__ void take_pair(flt2v t, double &xa, double &xb) {
    const auto r0 = __builtin_amdgcn_permlane32_swap(__float_as_uint(t.x), __float_as_uint(t.y), false, false);
    xa = (double)__uint_as_float(r0[0]);
    xb = (double)__uint_as_float(r0[1]);
}
// every lane hands over its element of streams a and b; lanes 0-31 then store two consecutive elements of a, lanes 32-63 of b
__device__ __forceinline__ void store_pair_nt(double *pa, double *pb, int lane, double xa, double xb) {
    const auto r0 = __builtin_amdgcn_permlane32_swap(__double2loint(xa), __double2loint(xb), false, false);
    const auto r1 = __builtin_amdgcn_permlane32_swap(__double2hiint(xa), __double2hiint(xb), false, false);
    dbl2v t;
    t.x = __hiloint2double(r1[0], r0[0]);
    t.y = __hiloint2double(r1[1], r0[1]);
    __builtin_nontemporal_store(t, reinterpret_cast<dbl2v *>(lane < 32 ? pa : pb));
}

// In-kernel phase stamps (s_memtime) of the diagnostic build only; the stamps go to a buffer nothing else reads.
#ifdef SX_PHASES
#define SX_STAMP(i) do { if (threadIdx.x == 0 && a.dbg) a.dbg[(int64_t)blockIdx.x * 8 + (i)] = (long long)__builtin_readcyclecounter(); } while (0)
#else
#define SX_STAMP(i) do { } while (0)
#endif
// keeps the loads in front of it in front of the loads behind it (the memory counter retires in issue order: what is
// needed first must be issued first)
#define SX_LOAD_FENCE() asm volatile("" ::: "memory")

// CPB columns per workgroup (<= 16, the MFMA tile width).  Used for the rings on the ring-wise path (all of them without
// the node-space inverse, the inner ones with it; k_phys_hrbl_cell takes the rest).
// Load discipline as in k_phys_hrbl_cell: one burst at entry, oldest = needed first (the memory counter retires in issue
// order); the tendency history and the second half of the planes are consumed only after the column operators; all operator
// fragments of a wave's jobs are requested before its first MFMA.  WIDE: 16-byte-per-lane pairs (issue_pair / take_pair).
template <int NZ, int CPB, class ST, bool WIDE>
__global__ void __launch_bounds__(CPB * NZ, 4) k_phys_hrbl_mfma(PhysArgsT<ST> a) {      // 4 waves per SIMD: <= 128 VGPRs, two 512-thread workgroups per CU
    constexpr int CS = NZ + 2;                     // column stride in LDS
    __shared__ double X[3][16 * CS];               // div, Kv*ubz, Kv*vbz   -> inputs (columns >= CPB unused)
    // wb, d/dz(...), d/dz(...) -> outputs.  At zDim = 128 the two tile sets would exceed the 64 KB of static LDS: the results
    // then wait in the accumulators until every wave has finished reading X and are written over it.
    constexpr bool ALIAS = (NZ > 64);
    __shared__ double Ysep[ALIAS ? 1 : 3][ALIAS ? 1 : 16 * CS];
    double (*Y)[16 * CS] = ALIAS ? X : reinterpret_cast<double (*)[16 * CS]>(&Ysep[0][0]);
    __shared__ double s1[2][16];                   // ub, vb at level 1 ("10 m")
    const int lane = threadIdx.x & 63, wbase = threadIdx.x & ~63;
    const int elem = WIDE ? wbase + wide_elem(lane) : (int)threadIdx.x;      // element of the workgroup's CPB x NZ block
    const int k = elem % NZ, cl = elem / NZ;
    const int64_t col = a.col0 + (int64_t)blockIdx.x * CPB + cl;
    const bool live = col < a.col1;
    const double *par = a.par;
    const double g = par[SX_P_G], Kh = par[SX_P_KH], Hfree = par[SX_P_HFREE], f = par[SX_P_F];
    const int64_t p = live ? col * NZ + k : 0;
    // this lane's PAIR (elements 2i, 2i + 1 of the wave's 64: same column as its own element since NZ is even)
    const int64_t pw = live ? (a.col0 + (int64_t)blockIdx.x * CPB) * NZ + wbase + 2 * (lane & 31) : 0;
    typedef typename Vec2<ST>::type SV;
    double xd = 0.0, xu = 0.0, xv = 0.0;
    // ---- the burst.  Small per-column values first (they come back first), then the planes the column operators need,
    // then the rest.
    double r = 1.0, zk = 1.0, cs_d = 0.0, sn_d = 0.0;
    if (live) { r = a.r[col]; zk = a.z[k]; }
    if (live && k == 0) { cs_d = a.cosl[col]; sn_d = a.sinl[col]; }
    // planes as (value-type) v0: ub | v1: vb | v2: h | v3: ug | v4: vg and (derivative-type) pairs
    double ub = 0, vb = 0, h = 0, ug = 0, vg = 0;
    double ubr = 0, vbl = 0, ubz = 0, vbz = 0, hr = 0, hl = 0, ugr = 0, ugl = 0, vgr = 0, vgl = 0, ubrr = 0, ubl = 0, ubll = 0, vbr = 0, vbrr = 0, vbll = 0;
    dbl2v rv0, rv1;              // (ub, vb), (h, ug); vg travels alone
    SV rd[8];                    // (ubr, vbl) (ubz, vbz) | (hr, hl) (ugr, ugl) (vgr, vgl) (ubrr, ubl) (ubll, vbr) (vbrr, vbll)
#define PV(v) (a.P.val + (int64_t)(v) * a.N)
#define PD(v, s) (a.P.der + ((int64_t)((s) - 1) * a.V + (v)) * a.N)
#define LD2V(raw, x, y, va, vb_) { if (WIDE) raw = issue_pair<false>(PV(va) + pw, PV(vb_) + pw, lane); else { x = PV(va)[p]; y = PV(vb_)[p]; } }
#define LD2D(raw, x, y, va, sa, vb_, sb) { if (WIDE) raw = issue_pair<false>(PD(va, sa) + pw, PD(vb_, sb) + pw, lane); else { x = (double)PD(va, sa)[p]; y = (double)PD(vb_, sb)[p]; } }
    if (live) {
        LD2V(rv0, ub, vb, 3, 4)
        LD2D(rd[0], ubr, vbl, 3, a.s_r, 4, a.s_l)
        LD2D(rd[1], ubz, vbz, 3, a.s_z, 4, a.s_z)
        SX_LOAD_FENCE();
        LD2V(rv1, h, ug, 0, 1)
        vg = PV(2)[p];
        LD2D(rd[2], hr, hl, 0, a.s_r, 0, a.s_l)
        LD2D(rd[3], ugr, ugl, 1, a.s_r, 1, a.s_l)
        LD2D(rd[4], vgr, vgl, 2, a.s_r, 2, a.s_l)
        LD2D(rd[5], ubrr, ubl, 3, a.s_rr, 3, a.s_l)
        LD2D(rd[6], ubll, vbr, 3, a.s_ll, 4, a.s_r)
        LD2D(rd[7], vbrr, vbll, 4, a.s_rr, 4, a.s_ll)
        SX_LOAD_FENCE();
    }
    // tendency history of the five prognostic variables: requested behind the operator fragments (below)
    double e1h[5] = {0, 0, 0, 0, 0}, e2h[5] = {0, 0, 0, 0, 0};
    dbl2v rh[5];
    // 22 divisions by r / r^2 per point would make this kernel VALU-bound (an f64 division is ~25 instructions): the
    // reciprocal is formed once per thread and multiplied (differs from the reference's a / r by <= 1.5 ulp)
    double ri = 1.0, ri2 = 1.0;
    if (live) {
        if (WIDE) { take_pair(rv0, ub, vb); take_pair(rd[0], ubr, vbl); take_pair(rd[1], ubz, vbz); }
        ri = 1.0 / r;
        ri2 = ri * ri;
        const double S = sqrt((ubz * ubz) + (vbz * vbz));
        const double l = 1.0 / ((1.0 / (0.4 * zk)) + (1.0 / 80.0));
        const double Kv = (l * l) * S;
        xd = -((ub * ri) + ubr + (vbl * ri));
        xu = Kv * ubz;
        xv = Kv * vbz;
        if (k == 1) { s1[0][cl] = ub; s1[1][cl] = vb; }
    }
    __syncthreads();
    if (live && k == 0) {
        const double Um = par[SX_P_UM], Vm = par[SX_P_VM];
        const double cs = cs_d, sn = sn_d;
        const double sfcu = (Um * cs) + (Vm * sn), sfcv = (Vm * cs) - (Um * sn);
        const double u10 = s1[0][cl] + sfcu, v10 = s1[1][cl] + sfcv;
        const double U10 = sqrt(u10 * u10 + v10 * v10);
        double Cd = par[SX_P_CD];
        if (U10 < 5.2) Cd = 1.0e-3;
        else if (U10 < 33.6) Cd = 4.4e-4 * sqrt(U10);
        xu = Cd * U10 * u10;
        xv = Cd * U10 * v10;
    }
    X[0][cl * CS + k] = xd;
    X[1][cl * CS + k] = xu;
    X[2][cl * CS + k] = xv;
    __syncthreads();
    {
        const int wave = threadIdx.x >> 6;
        constexpr int RT = NZ / 16;                 // row tiles per operand
        constexpr int NW = CPB * NZ / 64;           // waves in the workgroup
        constexpr int JPW = (3 * RT + NW - 1) / NW;  // jobs per wave
        constexpr int KS = NZ / 4;                  // MFMA steps per job
        constexpr int KC = 8;                       // operator fragments requested at a time (register budget: 128 VGPRs)
        mfma_d4 acc[JPW];
#pragma unroll
        for (int jj = 0; jj < JPW; jj++) {
            const int job = wave + jj * NW;
            acc[jj] = mfma_d4{0.0, 0.0, 0.0, 0.0};
            const bool has = job < 3 * RT;
            const int op = has ? job / RT : 0, rt = has ? job % RT : 0;
            const double *MT = (op == 0) ? a.MintT : a.MdzT;      // MT[j][k] = M[k][j]
            const double *xb = X[op] + (lane & 15) * CS + (lane >> 4);
            const double *ma = MT + (int64_t)(lane >> 4) * NZ + rt * 16 + (lane & 15);
            for (int kc = 0; kc < KS; kc += KC) {
                double af[KC];
#pragma unroll
                for (int ks = 0; ks < KC; ks++) af[ks] = has ? ma[(int64_t)(kc + ks) * 4 * NZ] : 0.0;
                if (jj == 0 && kc == 0) {
                    // the history goes out BEHIND the first operator fragments: fragments issued after it would wait for its
                    // HBM latency before the first MFMA
                    SX_LOAD_FENCE();
                    if (live) {
#pragma unroll
                        for (int v = 0; v < 5; v++) {
                            if (WIDE) {
                                if (a.t >= 2) rh[v] = issue_pair<true>(a.E1 + (int64_t)v * a.N + pw, a.E2 + (int64_t)v * a.N + pw, lane);
                            } else {
                                if (a.t >= 2) e1h[v] = __builtin_nontemporal_load(a.E1 + (int64_t)v * a.N + p);
                                if (a.t >= 3) e2h[v] = __builtin_nontemporal_load(a.E2 + (int64_t)v * a.N + p);
                            }
                        }
                    }
                    SX_LOAD_FENCE();
                }
                if (has) {
#pragma unroll
                    for (int ks = 0; ks < KC; ks++)
                        acc[jj] = __builtin_amdgcn_mfma_f64_16x16x4f64(af[ks], xb[(kc + ks) * 4], acc[jj], 0, 0, 0);
                }
            }
        }
        if (ALIAS) __syncthreads();
#pragma unroll
        for (int jj = 0; jj < JPW; jj++) {
            const int job = wave + jj * NW;
            if (job < 3 * RT) {
                const int op = job / RT, rt = job % RT;
                double *yo = Y[op] + (lane & 15) * CS + rt * 16 + (lane >> 4);
                yo[0] = acc[jj][0]; yo[4] = acc[jj][1]; yo[8] = acc[jj][2]; yo[12] = acc[jj][3];
            }
        }
    }
    __syncthreads();
    if (!live) return;
    if (WIDE) {
        take_pair(rv1, h, ug);
        take_pair(rd[2], hr, hl); take_pair(rd[3], ugr, ugl); take_pair(rd[4], vgr, vgl);
        take_pair(rd[5], ubrr, ubl); take_pair(rd[6], ubll, vbr); take_pair(rd[7], vbrr, vbll);
#pragma unroll
        for (int v = 0; v < 5; v++)
            if (a.t >= 2) { take_pair(rh[v], e1h[v], e2h[v]); if (a.t < 3) e2h[v] = 0.0; }
    }
    const double wb = Y[0][cl * CS + k], vdu = Y[1][cl * CS + k], vdv = Y[2][cl * CS + k];
    if (a.write_w) a.P.val[(int64_t)5 * a.N + p] = wb;
    const double e0 = ((-vg * hl * ri) + (-ug * hr)) + (-(Hfree + h) * ((ug * ri) + ugr + (vgl * ri)));
    const double e1 = ((-vg * ugl * ri) + (-ug * ugr)) + (-g * hr) + (vg * (f + (vg * ri)));
    const double e2 = ((-vg * vgl * ri) + (-ug * vgr)) + (-g * (hl * ri)) + (-ug * (f + (vg * ri)));
    const double e3 = ((-vb * ubl * ri) + (-ub * ubr) + (-wb * ubz)) + (-g * hr) + (vb * (f + (vb * ri))) + vdu +
                      (Kh * ((ubr * ri) + ubrr - (ub * ri2) + (ubll * ri2) - (2.0 * vbl * ri2)));
    const double e4 = ((-vb * vbl * ri) + (-ub * vbr) + (-wb * vbz)) + (-g * (hl * ri)) + (-ub * (f + (vb * ri))) + vdv +
                      (Kh * ((vbr * ri) + vbrr - (vb * ri2) + (vbll * ri2) + (2.0 * ubl * ri2)));
    const double uu[5] = {h, ug, vg, ub, vb}, ee[5] = {e0, e1, e2, e3, e4};
#pragma unroll
    for (int v = 0; v < 5; v++) {          // explicit_timestep (src/semiimplicit.jl:672-698) with the prefetched history
        const int64_t o = (int64_t)v * a.N + p;
        double un;
        if (a.t == 1) un = uu[v] + (a.ts * ee[v]);
        else if (a.t == 2) un = uu[v] + (0.5 * a.ts) * ((3.0 * ee[v]) - e1h[v]);
        else un = uu[v] + ((a.ts / 12.0) * ((23.0 * ee[v]) - (16.0 * e1h[v]) + (5.0 * e2h[v])));
        if (WIDE) {
            store_pair_nt(a.En + (int64_t)v * a.N + pw, a.np1 + (int64_t)v * a.N + pw, lane, ee[v], un);
        } else {
            __builtin_nontemporal_store(ee[v], a.En + o);
            __builtin_nontemporal_store(un, a.np1 + o);
        }
    }
    __builtin_nontemporal_store(wb, a.np1 + (int64_t)5 * a.N + p);
    for (int v = 6; v < a.V; v++) ab_step(a, v, p, PSV(v), 0.0);
#undef PV
#undef PD
#undef LD2V
#undef LD2D
}

// Cell-wise node-space variant ("radial last", uniform rings): one workgroup = LAM azimuths x NZ levels of ONE radial
// cell, i.e. the 3 rings that share the same 4 spline nodes.  Each thread loads the 14 node transforms of its
// (lambda, z) at the 4 nodes once (56 values, kept in registers) and evaluates all 3 rings from them, so a node value
// enters the CU once instead of three times (the ring-wise grouping was bound by L1 fill rate, not by HBM).
// The column operators of the 3 x LAM columns run as one f64-MFMA batch; the fields are re-formed from the registers
// after it, ring by ring, for the tendencies.
// Load schedule (what the phase stamps asked for): every load a workgroup needs is issued in ONE burst at entry, oldest =
// needed first; nothing small is fetched on its own later.  The per-ring constants come without memory traffic: the
// basis weights phi / phi' / phi'' at a cell's three Gauss points are the same for every cell (kernel arguments, scalar
// registers) and r is recomputed from the cell index exactly as sx_create tabulates it.
// WIDE: 16-byte-per-lane loads / stores with the lane <-> level map of issue_pair (needs 64 | LAM * NZ, always true here).
template <int NZ, int LAM, class ST, bool WIDE>
__global__ void __launch_bounds__(LAM * NZ, 2) k_phys_hrbl_cell(PhysArgsT<ST> a, int cell0) {
    constexpr int CS = NZ + 2;
    constexpr int NCOL = 3 * LAM, NT = (NCOL + 15) / 16;
    __shared__ double X[3][NT * 16 * CS];
    // results of the column operators; at zDim = 128 they wait in the accumulators and are written over X (64 KB of static LDS)
    constexpr bool ALIAS = (NZ > 64);
    __shared__ double Ysep[ALIAS ? 1 : 3][ALIAS ? 1 : NT * 16 * CS];
    double (*Y)[NT * 16 * CS] = ALIAS ? X : reinterpret_cast<double (*)[NT * 16 * CS]>(&Ysep[0][0]);
    __shared__ double s1[2][NCOL];
    SX_STAMP(0);
    const int lane = threadIdx.x & 63, wbase = threadIdx.x & ~63;
    const int elem = WIDE ? wbase + wide_elem(lane) : (int)threadIdx.x;      // element of the workgroup's LAM x NZ block
    const int k = elem % NZ, ll = elem / NZ;
    const int nlb = a.L / LAM;
    const int cell = cell0 + blockIdx.x / nlb;
    const int lam = (blockIdx.x % nlb) * LAM + ll;
    // offset of this lane's PAIR inside a stream of the workgroup's block (WIDE)
    const int64_t pairo = (int64_t)(blockIdx.x % nlb) * LAM * NZ + wbase + 2 * (lane & 31);
    const double *par = a.par;
    const double g = par[SX_P_G], Kh = par[SX_P_KH], Hfree = par[SX_P_HFREE], f = par[SX_P_F];
    const int64_t gp = ((int64_t)cell * a.L + lam) * NZ + k;
    const int64_t gs = (int64_t)a.L * NZ;
    const int64_t gw = (int64_t)cell * a.L * NZ + pairo;                      // this lane's pair at node 0 of the cell
    const int64_t pc = ((int64_t)(cell * MUBAR) * a.L + lam) * NZ + k;        // this lane's point on ring mu = 0; + mu * gs
    const int64_t pw = (int64_t)(cell * MUBAR) * a.L * NZ + pairo;            // this lane's pair on ring mu = 0

    // ---- the one burst of loads, oldest first: level height and surface-drag angles (one small load each, L2-resident)
    const double zk = a.z[k];
    double cs_d = 0.0, sn_d = 0.0;
    if (k < MUBAR) {       // the surface-drag lanes (k < 3: one ring each)
        const int64_t col = (int64_t)(cell * MUBAR + k) * a.L + lam;
        cs_d = a.cosl[col]; sn_d = a.sinl[col];
    }
    // node transforms [transform][node].  WIDE keeps the raw 16-byte pairs (nodes 0|1 and 2|3) until they are needed.
    typedef typename Vec2<ST>::type SV;
    double qh[4], qhl[4], qug[4], qugl[4], qvg[4], qvgl[4];
    double qub[4], qubl[4], qubll[4], qubz[4], qvb[4], qvbl[4], qvbll[4], qvbz[4];
    dbl2v rv[5][2];         // value planes: ub, vb | h, ug, vg
    SV rd[9][2];            // derivative planes: ubz, vbz, vbl | hl, ugl, vgl, ubl, ubll, vbll
#define NODE_V(dst, raw, v)                                                                        \
    {                                                                                              \
        const double *gq = a.G.val + (int64_t)(v) * a.NG;                                          \
        if (WIDE) { raw[0] = issue_pair<false>(gq + gw, gq + gw + gs, lane); raw[1] = issue_pair<false>(gq + gw + 2 * gs, gq + gw + 3 * gs, lane); } \
        else { dst[0] = gq[gp]; dst[1] = gq[gp + gs]; dst[2] = gq[gp + 2 * gs]; dst[3] = gq[gp + 3 * gs]; } \
    }
#define NODE_D(dst, raw, v, s)                                                                     \
    {                                                                                              \
        const ST *gq = a.G.der + ((int64_t)((s) - 1) * a.V + (v)) * a.NG;                          \
        if (WIDE) { raw[0] = issue_pair<false>(gq + gw, gq + gw + gs, lane); raw[1] = issue_pair<false>(gq + gw + 2 * gs, gq + gw + 3 * gs, lane); } \
        else { dst[0] = gq[gp]; dst[1] = gq[gp + gs]; dst[2] = gq[gp + 2 * gs]; dst[3] = gq[gp + 3 * gs]; } \
    }
#define NODE_TAKE(dst, raw) { if (WIDE) { take_pair(raw[0], dst[0], dst[1]); take_pair(raw[1], dst[2], dst[3]); } }
    // what the column operators' inputs need ...
    NODE_V(qub, rv[0], 3) NODE_D(qubz, rd[0], 3, a.s_z) NODE_D(qvbz, rd[1], 4, a.s_z) NODE_D(qvbl, rd[2], 4, a.s_l) NODE_V(qvb, rv[1], 4)
    SX_LOAD_FENCE();
    // ... then everything else, needed only after the column operators
    NODE_V(qh, rv[2], 0) NODE_D(qhl, rd[3], 0, a.s_l) NODE_V(qug, rv[3], 1) NODE_D(qugl, rd[4], 1, a.s_l) NODE_V(qvg, rv[4], 2) NODE_D(qvgl, rd[5], 2, a.s_l)
    NODE_D(qubl, rd[6], 3, a.s_l) NODE_D(qubll, rd[7], 3, a.s_ll) NODE_D(qvbll, rd[8], 4, a.s_ll)
    // tendency history: ring 0 with the entry burst, ring 1 behind the first operator fragments (in flight during the MFMA
    // phase and ring 0), ring 2 at the start of the final phase; the fragments come in two chunks of 8 so that all of
    // this fits the 256 registers of a two-waves-per-SIMD kernel.  WIDE:
    // expdot_nm1 / nm2 of a variable travel as one pair; before step 3 the buffers exist but hold no history yet.
    double e1h[MUBAR][5], e2h[MUBAR][5];
    dbl2v rh[MUBAR][5];
#define HIST(mu)                                                                                   \
    _Pragma("unroll") for (int v = 0; v < 5; v++) {                                                \
        if (WIDE) {                                                                                \
            if (a.t >= 2) rh[mu][v] = issue_pair<true>(a.E1 + (int64_t)v * a.N + pw + (mu) * gs, a.E2 + (int64_t)v * a.N + pw + (mu) * gs, lane);   \
        } else {                                                                                   \
            e1h[mu][v] = (a.t >= 2) ? __builtin_nontemporal_load(a.E1 + (int64_t)v * a.N + pc + (mu) * gs) : 0.0;   \
            e2h[mu][v] = (a.t >= 3) ? __builtin_nontemporal_load(a.E2 + (int64_t)v * a.N + pc + (mu) * gs) : 0.0;   \
        }                                                                                          \
    }
#define HIST_TAKE(mu)                                                                              \
    _Pragma("unroll") for (int v = 0; v < 5; v++) {                                                \
        if (WIDE) {                                                                                \
            if (a.t >= 2) { take_pair(rh[mu][v], e1h[mu][v], e2h[mu][v]); if (a.t < 3) e2h[mu][v] = 0.0; }   \
            else { e1h[mu][v] = 0.0; e2h[mu][v] = 0.0; }                                           \
        }                                                                                          \
    }
    HIST(0)
    SX_LOAD_FENCE();

    // ---- inputs of the column operators
    NODE_TAKE(qub, rv[0]) NODE_TAKE(qubz, rd[0]) NODE_TAKE(qvbz, rd[1]) NODE_TAKE(qvbl, rd[2]) NODE_TAKE(qvb, rv[1])
#define DOT(w, q) ((w)[0] * q[0] + (w)[1] * q[1] + (w)[2] * q[2] + (w)[3] * q[3])
    const double lmix = 1.0 / ((1.0 / (0.4 * zk)) + (1.0 / 80.0));
    double rinv[MUBAR];
#pragma unroll
    for (int mu = 0; mu < MUBAR; mu++) {
        // r of the ring, as sx_create tabulates it (xmin + DX (c + 0.5 + offset of the Gauss point))
        rinv[mu] = 1.0 / (a.xmin + a.DX * ((a.gcell0 + cell) + 0.5 + a.goff[mu]));
        const double *w0 = a.phiw[0][mu], *w1 = a.phiw[1][mu];
        const double ri = rinv[mu];
        const double ub = DOT(w0, qub), ubr = DOT(w1, qub), vbl = DOT(w0, qvbl), ubz = DOT(w0, qubz), vbz = DOT(w0, qvbz);
        const double S = sqrt((ubz * ubz) + (vbz * vbz));
        const double Kv = (lmix * lmix) * S;
        const int c = mu * LAM + ll;
        X[0][c * CS + k] = -((ub * ri) + ubr + (vbl * ri));
        X[1][c * CS + k] = Kv * ubz;
        X[2][c * CS + k] = Kv * vbz;
        if (k == 1) { s1[0][c] = ub; s1[1][c] = DOT(w0, qvb); }
    }
    SX_STAMP(1);
    __syncthreads();
    if (k < MUBAR) {       // surface drag replaces the level-0 flux (src/shallowWaterModels.jl:463-482); lane k takes ring k
        const double Um = par[SX_P_UM], Vm = par[SX_P_VM];
        const int c = k * LAM + ll;
        const double sfcu = (Um * cs_d) + (Vm * sn_d), sfcv = (Vm * cs_d) - (Um * sn_d);
        const double u10 = s1[0][c] + sfcu, v10 = s1[1][c] + sfcv;
        const double U10 = sqrt(u10 * u10 + v10 * v10);
        double Cd = par[SX_P_CD];
        if (U10 < 5.2) Cd = 1.0e-3;
        else if (U10 < 33.6) Cd = 4.4e-4 * sqrt(U10);
        X[1][c * CS] = Cd * U10 * u10;
        X[2][c * CS] = Cd * U10 * v10;
    }
    __syncthreads();
    SX_STAMP(2);
    {
        const int wave = threadIdx.x >> 6;
        constexpr int RT = NZ / 16, NW = LAM * NZ / 64;
        constexpr int UPW = (RT * NT + NW - 1) / NW;         // (row tile, column tile) units per wave
        constexpr int KC = 8;                                // operator fragments fetched per chunk (register budget)
        mfma_d4 c0[UPW], c1[UPW], c2[UPW];
#pragma unroll
        for (int uu = 0; uu < UPW; uu++) {
            const int unit = wave + uu * NW;
            c0[uu] = mfma_d4{0.0, 0.0, 0.0, 0.0}; c1[uu] = c0[uu]; c2[uu] = c0[uu];
            if (unit < RT * NT) {
                const int rt = unit % RT, nt = unit / RT;
                const int64_t ao = (int64_t)(lane >> 4) * NZ + rt * 16 + (lane & 15);      // MT[j][k] = M[k][j]
                const int xo = (nt * 16 + (lane & 15)) * CS + (lane >> 4);
                for (int kc = 0; kc < NZ / 4; kc += KC) {
                    double ai[KC], ad[KC];
#pragma unroll
                    for (int ks = 0; ks < KC; ks++) {
                        ai[ks] = a.MintT[ao + (int64_t)(kc + ks) * 4 * NZ];
                        ad[ks] = a.MdzT[ao + (int64_t)(kc + ks) * 4 * NZ];
                    }
                    if (uu == 0 && kc == 0) {
                        // ring 1's history goes out BEHIND the first operator fragments: the memory counter retires in issue
                        // order, so fragments issued after it would wait for its HBM latency before the first MFMA
                        SX_LOAD_FENCE();
                        HIST(1)
                        SX_LOAD_FENCE();
                    }
#pragma unroll
                    for (int ks = 0; ks < KC; ks++) {
                        c0[uu] = __builtin_amdgcn_mfma_f64_16x16x4f64(ai[ks], X[0][xo + (kc + ks) * 4], c0[uu], 0, 0, 0);
                        c1[uu] = __builtin_amdgcn_mfma_f64_16x16x4f64(ad[ks], X[1][xo + (kc + ks) * 4], c1[uu], 0, 0, 0);
                        c2[uu] = __builtin_amdgcn_mfma_f64_16x16x4f64(ad[ks], X[2][xo + (kc + ks) * 4], c2[uu], 0, 0, 0);
                    }
                }
            }
        }
        if (RT * NT < NW && wave >= RT * NT) { HIST(1) }      // waves without a unit (never at the shipped shapes)
        SX_STAMP(3);
        if (ALIAS) __syncthreads();
#pragma unroll
        for (int uu = 0; uu < UPW; uu++) {
            const int unit = wave + uu * NW;
            if (unit < RT * NT) {
                const int rt = unit % RT, nt = unit / RT;
                const int yo = (nt * 16 + (lane & 15)) * CS + rt * 16 + (lane >> 4);
                Y[0][yo] = c0[uu][0]; Y[0][yo + 4] = c0[uu][1]; Y[0][yo + 8] = c0[uu][2]; Y[0][yo + 12] = c0[uu][3];
                Y[1][yo] = c1[uu][0]; Y[1][yo + 4] = c1[uu][1]; Y[1][yo + 8] = c1[uu][2]; Y[1][yo + 12] = c1[uu][3];
                Y[2][yo] = c2[uu][0]; Y[2][yo + 4] = c2[uu][1]; Y[2][yo + 8] = c2[uu][2]; Y[2][yo + 12] = c2[uu][3];
            }
        }
    }
    __syncthreads();
    SX_STAMP(4);
    HIST(2)
    SX_LOAD_FENCE();
    NODE_TAKE(qh, rv[2]) NODE_TAKE(qhl, rd[3]) NODE_TAKE(qug, rv[3]) NODE_TAKE(qugl, rd[4]) NODE_TAKE(qvg, rv[4]) NODE_TAKE(qvgl, rd[5])
    NODE_TAKE(qubl, rd[6]) NODE_TAKE(qubll, rd[7]) NODE_TAKE(qvbll, rd[8])
    double wb_keep = 0.0;
#pragma unroll
    for (int mu = 0; mu < MUBAR; mu++) {
        const int64_t p = pc + mu * gs;
        HIST_TAKE(mu)
        const double *w0 = a.phiw[0][mu], *w1 = a.phiw[1][mu], *w2 = a.phiw[2][mu];
        const double ri = rinv[mu], ri2 = ri * ri;
        const double h = DOT(w0, qh), hr = DOT(w1, qh), hl = DOT(w0, qhl);
        const double ug = DOT(w0, qug), ugr = DOT(w1, qug), ugl = DOT(w0, qugl);
        const double vg = DOT(w0, qvg), vgr = DOT(w1, qvg), vgl = DOT(w0, qvgl);
        const double ub = DOT(w0, qub), ubr = DOT(w1, qub), ubrr = DOT(w2, qub);
        const double ubl = DOT(w0, qubl), ubll = DOT(w0, qubll), ubz = DOT(w0, qubz);
        const double vb = DOT(w0, qvb), vbr = DOT(w1, qvb), vbrr = DOT(w2, qvb);
        const double vbl = DOT(w0, qvbl), vbll = DOT(w0, qvbll), vbz = DOT(w0, qvbz);
        const int c = mu * LAM + ll;
        const double wb = Y[0][c * CS + k], vdu = Y[1][c * CS + k], vdv = Y[2][c * CS + k];
        if (a.write_w) a.P.val[(int64_t)5 * a.N + p] = wb;
        const double e0 = ((-vg * hl * ri) + (-ug * hr)) + (-(Hfree + h) * ((ug * ri) + ugr + (vgl * ri)));
        const double e1 = ((-vg * ugl * ri) + (-ug * ugr)) + (-g * hr) + (vg * (f + (vg * ri)));
        const double e2 = ((-vg * vgl * ri) + (-ug * vgr)) + (-g * (hl * ri)) + (-ug * (f + (vg * ri)));
        const double e3 = ((-vb * ubl * ri) + (-ub * ubr) + (-wb * ubz)) + (-g * hr) + (vb * (f + (vb * ri))) + vdu +
                          (Kh * ((ubr * ri) + ubrr - (ub * ri2) + (ubll * ri2) - (2.0 * vbl * ri2)));
        const double e4 = ((-vb * vbl * ri) + (-ub * vbr) + (-wb * vbz)) + (-g * (hl * ri)) + (-ub * (f + (vb * ri))) + vdv +
                          (Kh * ((vbr * ri) + vbrr - (vb * ri2) + (vbll * ri2) + (2.0 * ubl * ri2)));
        const double uu[5] = {h, ug, vg, ub, vb}, ee[5] = {e0, e1, e2, e3, e4};
#pragma unroll
        for (int v = 0; v < 5; v++) {          // explicit_timestep (src/semiimplicit.jl:672-698)
            const int64_t o = (int64_t)v * a.N + p;
            double un;
            if (a.t == 1) un = uu[v] + (a.ts * ee[v]);
            else if (a.t == 2) un = uu[v] + (0.5 * a.ts) * ((3.0 * ee[v]) - e1h[mu][v]);
            else un = uu[v] + ((a.ts / 12.0) * ((23.0 * ee[v]) - (16.0 * e1h[mu][v]) + (5.0 * e2h[mu][v])));
            // expdot_n is read again only by the next step; var_np1 (0.4 GB per step) next by the forward transform, after
            // everything else of this kernel has gone through the caches: both non-temporal
            if (WIDE) {
                const int64_t ow = (int64_t)v * a.N + pw + mu * gs;
                store_pair_nt(a.En + ow, a.np1 + ow, lane, ee[v], un);
            } else {
                __builtin_nontemporal_store(ee[v], a.En + o);
                __builtin_nontemporal_store(un, a.np1 + o);
            }
        }
        if (WIDE) {        // the diagnostic w of rings 0 and 1 leaves as one pair, ring 2's on its own
            if (mu == 0) wb_keep = wb;
            else if (mu == 1) store_pair_nt(a.np1 + (int64_t)5 * a.N + pw, a.np1 + (int64_t)5 * a.N + pw + gs, lane, wb_keep, wb);
            else __builtin_nontemporal_store(wb, a.np1 + (int64_t)5 * a.N + p);
        } else {
            __builtin_nontemporal_store(wb, a.np1 + (int64_t)5 * a.N + p);
        }
        SX_STAMP(5 + mu);
    }
#undef DOT
#undef HIST
#undef HIST_TAKE
#undef NODE_V
#undef NODE_D
#undef NODE_TAKE
}

// semiimplicit_adjustment (src/semiimplicit.jl:521-597), one workgroup per group of columns
__global__ void __launch_bounds__(256) k_semiimplicit(SemiArgs a, int cpb) {
    extern __shared__ double sm[];
    const int nz = a.nz;
    const int k = threadIdx.x % nz, cl = threadIdx.x / nz;
    const int64_t col = (int64_t)blockIdx.x * cpb + cl;
    const bool live = (cl < cpb) && (col < a.N / nz);
    double *sw = sm, *sx_ = sm + (size_t)cpb * nz, *sg = sm + (size_t)2 * cpb * nz;
    const int64_t p = live ? col * nz + k : 0;
    const double ts = a.ts;
    if (live) {
        const int vv[2] = {a.wi, a.xi};
        double out[2];
        for (int q = 0; q < 2; q++) {
            const int64_t o = (int64_t)vv[q] * a.N + p;
            double x = a.np1[o];
            const double In = a.In[o];
            if (a.t == 1) x = x - (ts * In) + (ts * 0.5 * In);
            else if (a.t == 2) x = x - (0.5 * ts) * ((3.0 * In) - a.I1[o]) - (ts * In) + (ts * 0.75 * a.I1[o]);
            else x = x - ((ts / 12.0) * ((23.0 * In) - (16.0 * a.I1[o]) + (5.0 * a.I2[o]))) - (ts * In) + (ts * 0.75 * a.I1[o]);
            out[q] = x;
        }
        sw[cl * nz + k] = out[0];
        sx_[cl * nz + k] = out[1];
    }
    __syncthreads();
    double xrec = 0.0, xz = 0.0;
    if (live) {
        const double *x = sx_ + cl * nz;
        for (int j = 0; j < nz; j++) {
            xrec += a.MrecT[(int64_t)j * nz + k] * x[j];
            xz += a.MdzT[(int64_t)j * nz + k] * x[j];
        }
        // g = [0; 0; (tau Pxi xi*_z - w*)[2 : nz-1]]
        if (k >= 1 && k < nz - 1) sg[cl * nz + k + 1] = (a.tau * a.pxi * xz) - sw[cl * nz + k];
        if (k < 2) sg[cl * nz + k] = 0.0;
    }
    __syncthreads();
    if (!live) return;
    double wn = 0.0, wz = 0.0;
    const double *gv = sg + cl * nz;
    for (int j = 0; j < nz; j++) {
        wn += a.WT[(int64_t)j * nz + k] * gv[j];
        wz += a.XT[(int64_t)j * nz + k] * gv[j];
    }
    a.np1[(int64_t)a.wi * a.N + p] = wn;
    a.np1[(int64_t)a.xi * a.N + p] = xrec - (a.tau * wz);
}

// ------------------------------------------------------------------------------------------------ launchers
static inline dim3 grid1(int64_t n, int bs) { return dim3((unsigned)((n + bs - 1) / bs)); }

#ifdef SX_PHASES
// diagnostic build: stamps of the LAST cell-kernel launch, written to $SX_PHASES_OUT (binary int64 [nwg][8]) by sx_destroy
static long long *g_ph_buf = nullptr;
static int64_t g_ph_n = 0;
void phases_dump() {
    const char *path = getenv("SX_PHASES_OUT");
    if (!path || !g_ph_buf) return;
    std::vector<long long> hst((size_t)g_ph_n * 8);
    hipDeviceSynchronize();
    hipMemcpy(hst.data(), g_ph_buf, sizeof(long long) * hst.size(), hipMemcpyDeviceToHost);
    FILE *f = fopen(path, "wb");
    if (f) { fwrite(hst.data(), sizeof(long long), hst.size(), f); fclose(f); }
}
static long long *g_sbw_buf = nullptr;
static int64_t g_sbw_n = 0;
void sbw_phases_dump() {
    const char *path = getenv("SX_SBW_PHASES_OUT");
    if (!path || !g_sbw_buf) return;
    std::vector<long long> hst((size_t)g_sbw_n * 8);
    hipDeviceSynchronize();
    hipMemcpy(hst.data(), g_sbw_buf, sizeof(long long) * hst.size(), hipMemcpyDeviceToHost);
    FILE *f = fopen(path, "wb");
    if (f) { fwrite(hst.data(), sizeof(long long), hst.size(), f); fclose(f); }
}
static long long *phases_buffer(int64_t nwg) {
    if (!g_ph_buf) {
        hipMalloc(&g_ph_buf, sizeof(long long) * nwg * 8);
        hipMemset(g_ph_buf, 0, sizeof(long long) * nwg * 8);
        g_ph_n = nwg;
    }
    return nwg <= g_ph_n ? g_ph_buf : nullptr;
}
#endif

void launch_zinv(sx_handle *h, bool full) {
    if (!h->has_z || rz_fused(h)) return;          // RZ: the vertical inverse is part of k_rz_inverse (sx_rz.hip)
    const int id = timer_id(h, "k_zinv");
    timer_begin(h, id);
    const int njobs = full ? h->njobs_zinv_full : h->njobs_zinv_eq;
    h->last_zinv_jobs = njobs;
    if (njobs > 0) {
        // inside sx_advance on uniform rings the node-space units invert vertically inside their FFT kernel (sx_fft.hip, FUSE):
        // Az is then needed only for the nodes the ring-wise inner rings read (cells [0, R_in / 3) -> nodes 0 .. R_in / 3 + 2)
        const int rows = (!full && h->node_mode && fft_fused_zinv(h)) ? (h->R_in > 0 ? std::min(h->nbt, h->R_in / MUBAR + 3) : 0) : h->nbt;
        h->last_zinv_rows = rows;
        if (rows == 0) { timer_end(h); return; }
        static const int ct_env = getenv("SX_ZINV_CT") ? atoi(getenv("SX_ZINV_CT")) : 0;      // A/B: column tiles per wave at 128 levels
        // only CT = 1, 2, 4 are instantiated at 128 levels (1, 2 at 64): any other request takes the default, never a grid sized for a kernel that is not launched
        const int ct = h->nz == 128 ? ((ct_env == 1 || ct_env == 4) ? ct_env : 2) : (h->nz == 64 && ct_env == 2 ? 2 : 1);
        dim3 g((h->K2 + 64 * ct - 1) / (64 * ct), njobs, rows);
        const ColJob *jobs = full ? h->d_jobs_zinv_full : h->d_jobs_zinv_eq;
        const int64_t azrow = (int64_t)h->V * 3 * h->nz * h->K2;
#define ZINV(MT, OT, CT) hipLaunchKernelGGL((k_colmat_mfma<MT, OT, CT>), g, dim3(256), 0, h->stream, h->d_A, reinterpret_cast<OT *>(h->d_Az), h->d_MzT, jobs, h->Zb, h->K2, h->C, azrow, h->cell0)
        if (h->nz == 64 && ct == 2) { if (h->sp32) ZINV(4, float, 2); else ZINV(4, double, 2); }
        else if (h->nz == 64) { if (h->sp32) ZINV(4, float, 1); else ZINV(4, double, 1); }
        else if (h->nz == 32) { if (h->sp32) ZINV(2, float, 1); else ZINV(2, double, 1); }
        else if (h->nz == 128 && ct == 1) { if (h->sp32) ZINV(8, float, 1); else ZINV(8, double, 1); }
        else if (h->nz == 128 && ct == 4) { if (h->sp32) ZINV(8, float, 4); else ZINV(8, double, 4); }
        else if (h->nz == 128) { if (h->sp32) ZINV(8, float, 2); else ZINV(8, double, 2); }
#undef ZINV
        else
            hipLaunchKernelGGL(k_colmat, g, dim3(64, 4), sizeof(double) * 64 * h->Zb, h->stream, h->d_A, h->d_Az, h->d_Mz, jobs,
                               h->Zb, h->nz, h->K2, h->C, azrow, h->cell0);
        HIPCHK(hipGetLastError());
    }
    timer_end(h);
}

void launch_rl_inverse(sx_handle *h, bool full) {
    const int *mask = full ? h->d_mask_full : h->d_mask_eq;
    h->last_mask_full = full;
    h->node_active = (!full && h->node_mode);
    if (rz_fused(h)) { launch_rz_inverse(h, mask); return; }
    if (h->node_active) {            // sx_advance on uniform rings: node-space transforms + ring-wise inner rings only
        launch_node_fft(h);
        launch_rl_inverse_fft(h, mask, h->R_in);
        return;
    }
    if (fft_path_ok(h)) { launch_rl_inverse_fft(h, mask); return; }
    if (dft_mfma_ok(h)) { launch_rl_inverse_dft(h, mask); return; }
    const int id = timer_id(h, "k_rl_inverse");
    timer_begin(h, id);
    const int cstride = (h->kmax_max + 1) | 1;
    const size_t lds = sizeof(double) * 2 * ZC * cstride;
    if (lds > 64 * 1024) {
        set_error("azimuthal inverse transform: rings with " + std::to_string(h->kmax_max) + " wavenumbers are outside every transform path (power-of-two ring tables up to 512 points, ring lengths that are multiples of 4 up to 5120 points, otherwise kmax <= 255)");
        timer_end(h);
        return;
    }
    const double *az = h->has_z ? h->d_Az : h->d_A + (int64_t)h->cell0 * h->C;
    const int64_t azrow = h->has_z ? (int64_t)h->V * 3 * h->nz * h->K2 : h->C;
    dim3 g((h->nz + ZC - 1) / ZC, h->V, h->nrings);
#define RL_ARGS h->d_phi, h->d_L, h->d_kmax, h->d_pstart, h->d_twoff, h->d_tw, h->d_phoff, h->d_ph, h->V, h->nz, h->nsz, h->K2,   \
                h->nrings, h->N, azrow, h->slot[0], h->slot[1], h->slot[2], h->slot[3], h->slot[4], h->slot[5], h->slot[6],   \
                h->has_l, cstride, mask
    if (h->f32) hipLaunchKernelGGL(k_rl_inverse<float>, g, dim3(256), lds, h->stream, az, planes_of<float>(h->d_phys, h->V, h->N), RL_ARGS);
    else hipLaunchKernelGGL(k_rl_inverse<double>, g, dim3(256), lds, h->stream, az, planes_of<double>(h->d_phys, h->V, h->N), RL_ARGS);
#undef RL_ARGS
    HIPCHK(hipGetLastError());
    timer_end(h);
}

template <class ST>
static PhysArgsT<ST> phys_args(sx_handle *h, int t) {
    PhysArgsT<ST> a;
    a.P = planes_of<ST>(h->d_phys, h->V, h->N);
    a.En = h->d_E[h->rot % 3];
    a.E1 = h->d_E[(h->rot + 1) % 3];
    a.E2 = h->d_E[(h->rot + 2) % 3];
    a.In = h->d_I[0] ? h->d_I[h->rot % 3] : nullptr;
    a.np1 = h->d_np1;
    a.r = h->d_r; a.cosl = h->d_cosl; a.sinl = h->d_sinl; a.z = h->d_z;
    a.MintT = h->d_MintT; a.MdzT = h->d_MdzT; a.ref = h->d_ref; a.write_w = h->in_advance ? 0 : 1;
    a.N = h->N; a.V = h->V; a.nz = h->nz; a.t = t; a.eq = h->eq;
    a.s_u = h->slot[0]; a.s_r = h->slot[1]; a.s_rr = h->slot[2]; a.s_l = h->slot[3]; a.s_ll = h->slot[4];
    a.s_z = h->slot[5]; a.s_zz = h->slot[6];
    a.ts = h->ts;
    for (int i = 0; i < SX_NPARAMS; i++) a.par[i] = h->par[i];
    a.col0 = 0; a.col1 = h->Nh; a.G = Planes<ST>{nullptr, nullptr}; a.phi = nullptr; a.NG = 0; a.L = 1; a.nrings = h->nrings;
    a.dbg = nullptr;
    return a;
}

// History rotation replaces the copies of explicit_timestep: after step t the buffer written as expdot_n becomes
// expdot_nm1 and the previous nm1 becomes nm2. rot decreases by one (mod 3) per step.
constexpr int PCPB = 8;       // columns per workgroup of the ring-wise MFMA HRBL kernel: two resident 512-thread workgroups per CU
                              // (A/B on one box: 0.146 ms vs 0.154 ms with 16 columns / one workgroup per CU)

template <class ST>
static void launch_physics_t(sx_handle *h, int t, int part) {
    // part: 0 = everything; 1 = only the rings on the ring-wise path, 2 = only the node-space rings (the two halves of
    // launch_inverse_and_physics; the history rotation happens once, in part 1 before and in part 2 after)
    if (h->eq != SX_EQ_NONE && t == 1 && part != 2) h->rot = 0;
    PhysArgsT<ST> a = phys_args<ST>(h, t);
    if (h->eq == SX_EQ_ONEWAY_SW_HRBL && (h->nz == 64 || h->nz == 32 || h->nz == 128)) {
        // rings [0, R_in): ring-wise physical slots; rings [R_in, nrings): node-space transforms (node_mode only)
        const int64_t split = (h->node_mode && h->node_active) ? (int64_t)h->R_in * h->uniform_L : h->Nh;
        if (h->d_G) a.G = planes_of<ST>(h->d_G, h->V, h->NG);
        a.phi = h->d_phi; a.NG = h->NG; a.L = h->uniform_L; a.nrings = h->nrings;
        {
            double phi[4][MUBAR][4];
            basis_tables(h->DX, phi);
            for (int d = 0; d < 3; d++)
                for (int mu = 0; mu < MUBAR; mu++)
                    for (int j = 0; j < 4; j++) a.phiw[d][mu][j] = phi[d][mu][j];
            const double off3[MUBAR] = {-std::sqrt(3.0 / 5.0) / 2.0, 0.0, std::sqrt(3.0 / 5.0) / 2.0};     // as in sx_create
            for (int mu = 0; mu < MUBAR; mu++) a.goff[mu] = off3[mu];
            a.xmin = h->xmin; a.DX = h->DX; a.gcell0 = h->cell0;
        }
        if (split > 0 && part != 2) {
            const int id = timer_id(h, split < h->Nh ? "k_phys_hrbl_inner" : "k_phys_hrbl");
            timer_begin(h, id);
            a.col0 = 0; a.col1 = split;
#define RING_LAUNCH(NZ_, CPB_)                                                                                                     \
            do {                                                                                                                      \
                if (h->wide) hipLaunchKernelGGL((k_phys_hrbl_mfma<NZ_, CPB_, ST, true>), grid1(split, CPB_), dim3(CPB_ * NZ_), 0, h->stream, a);   \
                else hipLaunchKernelGGL((k_phys_hrbl_mfma<NZ_, CPB_, ST, false>), grid1(split, CPB_), dim3(CPB_ * NZ_), 0, h->stream, a);          \
            } while (0)
            if (h->nz == 64) RING_LAUNCH(64, PCPB);
            else if (h->nz == 32) RING_LAUNCH(32, PCPB);
            else RING_LAUNCH(128, 8);
#undef RING_LAUNCH
            HIPCHK(hipGetLastError());
            timer_end(h);
        }
        if (split < h->Nh && part != 1) {
            const int id = timer_id(h, "k_phys_hrbl");
            timer_begin(h, id);
            a.col0 = split; a.col1 = h->Nh;
            const int ncell = (h->nrings - h->R_in) / MUBAR;         // R_in is a multiple of 3 (sx_create)
#ifdef SX_PHASES
            a.dbg = phases_buffer((int64_t)ncell * (h->uniform_L / (h->nz == 32 ? 8 : 4)));
#endif
#define CELL_LAUNCH(NZ_, LAM_)                                                                                                     \
            do {                                                                                                                      \
                if (h->wide) hipLaunchKernelGGL((k_phys_hrbl_cell<NZ_, LAM_, ST, true>), dim3(ncell * (h->uniform_L / LAM_)), dim3(LAM_ * NZ_), 0, h->stream, a, h->R_in / MUBAR);   \
                else hipLaunchKernelGGL((k_phys_hrbl_cell<NZ_, LAM_, ST, false>), dim3(ncell * (h->uniform_L / LAM_)), dim3(LAM_ * NZ_), 0, h->stream, a, h->R_in / MUBAR);          \
            } while (0)
            if (h->nz == 64) CELL_LAUNCH(64, 4);      // LAM 2: 0.55 ms (6 of 16 MFMA columns, 1 KB chunks); 4: 0.41 ms
            else if (h->nz == 32) CELL_LAUNCH(32, 8);
            else CELL_LAUNCH(128, 4);   // 512 threads, one workgroup per CU: 12 of 16 MFMA columns (LAM 2: 6 of 16, 3.23 vs 2.58 ms at config 5)
#undef CELL_LAUNCH
            HIPCHK(hipGetLastError());
            timer_end(h);
        }
    } else if (h->eq == SX_EQ_ONEWAY_SW_HRBL) {
        const int id = timer_id(h, "k_phys_hrbl");
        timer_begin(h, id);
        const int cpb = h->nz >= 256 ? 1 : 256 / h->nz;
        const int bs = cpb * h->nz;
        const size_t lds = sizeof(double) * 5 * cpb * h->nz;
        hipLaunchKernelGGL(k_phys_hrbl<ST>, grid1(h->Nh, cpb), dim3(bs), lds, h->stream, a, cpb);
        HIPCHK(hipGetLastError());
        timer_end(h);
    } else {
        const int id = timer_id(h, "k_phys_pointwise");
        timer_begin(h, id);
        hipLaunchKernelGGL(k_phys_pointwise<ST>, grid1(h->N, 256), dim3(256), 0, h->stream, a);
        HIPCHK(hipGetLastError());
        timer_end(h);
    }
    if (h->semi && h->eq != SX_EQ_NONE) {
        const int id = timer_id(h, "k_semiimplicit");
        timer_begin(h, id);
        SemiArgs s;
        s.np1 = h->d_np1;
        s.In = h->d_I[h->rot % 3]; s.I1 = h->d_I[(h->rot + 1) % 3]; s.I2 = h->d_I[(h->rot + 2) % 3];
        const int which = (t == 1) ? 0 : 1;
        s.MrecT = h->d_MrecT; s.MdzT = h->d_MdzT; s.WT = h->d_WT[which]; s.XT = h->d_XT[which];
        s.N = h->N; s.nz = h->nz; s.t = t; s.wi = h->w_index - 1; s.xi = h->xi_index - 1;
        s.ts = h->ts; s.tau = h->tau[which]; s.pxi = h->par[SX_P_PXI_BAR];
        if (h->semi_mfma) launch_semi_mfma(h, s);       // the four column operators on the matrix cores (sx_rz.hip)
        else {
            const int cpb = h->nz >= 256 ? 1 : 256 / h->nz;
            const size_t lds = sizeof(double) * 3 * cpb * h->nz;
            hipLaunchKernelGGL(k_semiimplicit, grid1(h->Nh, cpb), dim3(cpb * h->nz), lds, h->stream, s, cpb);
        }
        HIPCHK(hipGetLastError());
        timer_end(h);
    }
    if (h->eq != SX_EQ_NONE && part != 1) h->rot = (h->rot + 2) % 3;
}

static void launch_physics_part(sx_handle *h, int t, int part) {
    if (h->f32) launch_physics_t<float>(h, t, part);
    else launch_physics_t<double>(h, t, part);
}

void launch_physics(sx_handle *h, int t) { launch_physics_part(h, t, 0); }

// sx_advance's inverse transform + equation set.  With the node-space inverse the tile has two independent chains -
// inner rings: ring-wise FFT -> ring-wise HRBL kernel; outer rings: node FFT -> cell-wise HRBL kernel - that touch
// disjoint points.  With SX_OVERLAP=1 the inner chain runs on a second (non-blocking) stream, forked after the vertical
// inverse and joined before the forward transform (measured gain 2.6 %: off by default, see sx_internal.hpp).
void launch_inverse_and_physics(sx_handle *h, int t) {
    struct Scope { sx_handle *h; Scope(sx_handle *x) : h(x) { h->in_advance = true; } ~Scope() { h->in_advance = false; } } scope(h);
    const bool two = h->node_mode && h->R_in > 0 && h->overlap && h->eq == SX_EQ_ONEWAY_SW_HRBL && !h->semi;
    if (!two) {
        launch_rl_inverse(h, false);
        launch_physics(h, t);
        return;
    }
    h->last_mask_full = false;
    h->node_active = 1;
    if (!h->stream2) {
        HIPCHK(hipStreamCreateWithFlags(&h->stream2, hipStreamNonBlocking));
        HIPCHK(hipEventCreateWithFlags(&h->ev_fork, hipEventDisableTiming));
        HIPCHK(hipEventCreateWithFlags(&h->ev_join, hipEventDisableTiming));
    }
    hipStream_t s0 = h->stream;
    HIPCHK(hipEventRecord(h->ev_fork, s0));
    HIPCHK(hipStreamWaitEvent(h->stream2, h->ev_fork, 0));
    h->stream = h->stream2;                                  // launchers and timers follow h->stream
    launch_rl_inverse_fft(h, h->d_mask_eq, h->R_in);
    launch_physics_part(h, t, 1);
    HIPCHK(hipEventRecord(h->ev_join, h->stream2));
    h->stream = s0;
    launch_node_fft(h);
    // overlap = 2: only the node FFT shares the chip with the inner chain; the cell-wise equation-set kernel (the dominant
    // one, whose event-timed duration is the roofline measurement) starts after the join and runs alone
    if (h->overlap == 2) HIPCHK(hipStreamWaitEvent(s0, h->ev_join, 0));
    launch_physics_part(h, t, 2);
    if (h->overlap != 2) HIPCHK(hipStreamWaitEvent(s0, h->ev_join, 0));
}

void launch_fl_forward(sx_handle *h) {
    if (rz_fused(h)) return;                        // RZ: k_rz_forward reads var_np1 itself (no azimuth, nothing to transform)
    if (fft_path_ok(h)) { launch_fl_forward_fft(h); return; }
    if (dft_mfma_ok(h)) { launch_fl_forward_dft(h); return; }
    const int id = timer_id(h, "k_fl_forward");
    timer_begin(h, id);
    const int xstride = h->L_max | 1;
    const size_t lds = sizeof(double) * ZC * xstride;
    if (lds > 64 * 1024) {       // the scalar kernel stages a whole ring; longer rings need the matrix-core DFT (lengths that are multiples of 4)
        set_error("azimuthal forward transform: rings of " + std::to_string(h->L_max) + " points are outside every transform path (power-of-two ring tables up to 512 points, ring lengths that are multiples of 4 up to 5120 points, any length up to 511 points)");
        return;
    }
    dim3 g((h->nz + ZC - 1) / ZC, h->V, h->nrings);
    hipLaunchKernelGGL(k_fl_forward, g, dim3(256), lds, h->stream, h->d_np1, h->d_Fl, h->d_L, h->d_kmax, h->d_pstart,
                       h->d_twoff, h->d_tw, h->d_phoff, h->d_ph, h->V, h->nz, h->K2, h->N, h->has_l, xstride);
    HIPCHK(hipGetLastError());
    timer_end(h);
}

void launch_sb(sx_handle *h) {
    if (rz_fused(h)) { launch_rz_forward(h); return; }
    if (h->has_z) {        // fused with the vertical forward transform
        const int id = timer_id(h, "k_sbz");
        timer_begin(h, id);
        if (h->nz == 64 || h->nz == 32 || h->nz == 128) {
            // cells per workgroup (+3 warm-up cells).  With the prefetch (zDim <= 64: 177 VGPRs, one 512-thread workgroup per
            // CU) the grid is ONE round of at most 256 workgroups; without it (zDim 128: 16 values per thread and ring leave no
            // registers for a second set; or SX_SBW_PF=0) about 1.5 workgroups per CU as before.  On large tiles never fewer
            // than 6 cells so that the warm-up stays below half of the reads
            const bool mf = h->sbw_mfma && (h->nz <= 64 ? h->Zb <= 64 : h->Zb <= 96);      // matrix-core contraction + prefetch (k_sbw_mfma)
            // zDim 64: 256-thread workgroups of 32 blocks, two per CU - one loads while the other contracts (0.127 -> 0.118 ms;
            // SX_SBW_T256=0 restores the 512-thread form)
            static const bool t256_env = !(getenv("SX_SBW_T256") && atoi(getenv("SX_SBW_T256")) == 0);
            const bool t256 = t256_env && mf && h->nz == 64;
            const int bw = ((mf && h->nz == 128) || t256) ? 32 : 64;         // wavenumber blocks per workgroup
            const int groups = ((h->K2 + bw - 1) / bw) * h->v_cnt;
            const bool pf = (h->sbw_prefetch && h->nz <= 64) || mf;
            static const int seg_env = getenv("SX_SBW_SEG") ? atoi(getenv("SX_SBW_SEG")) : 0;      // experiments: segments per (block group, variable)
            const int nseg = seg_env > 0 ? seg_env : std::max(1, ((mf && h->nz == 128) || t256 ? 512 : pf || h->nz == 128 ? 256 : 384) / groups);
            // small tiles (multi-GPU strong scaling): the kernel is then one workgroup's latency chain, which is proportional
            // to the cells it walks, so short segments (down to 2 cells + 3 warm-up) beat the saved re-reads
            const int cps = std::max(h->ncells <= 64 ? 2 : 6, (h->ncells + nseg - 1) / nseg);
            dim3 gw((h->K2 + bw - 1) / bw, h->v_cnt, (h->ncells + cps - 1) / cps);      // variable window: see sx_internal.hpp
            const int64_t flo = (int64_t)h->v_lo * h->nz * h->K2, blo = (int64_t)h->v_lo * h->Zb * h->K2;
#ifdef SX_PHASES
            if (!g_sbw_buf) {
                g_sbw_n = (int64_t)gw.x * gw.y * gw.z;
                hipMalloc(&g_sbw_buf, sizeof(long long) * g_sbw_n * 8);
                hipMemset(g_sbw_buf, 0, sizeof(long long) * g_sbw_n * 8);
                hipMemcpyToSymbol(HIP_SYMBOL(g_sbw_dbg), &g_sbw_buf, sizeof(g_sbw_buf));
            }
#endif
            auto kern = h->nz == 64 ? (pf ? k_sbw<64, true> : k_sbw<64, false>) : h->nz == 32 ? (pf ? k_sbw<32, true> : k_sbw<32, false>) : k_sbw<128, false>;
            if (mf) kern = h->nz == 64 ? k_sbw_mfma<64> : h->nz == 32 ? k_sbw_mfma<32> : k_sbw_mfma<128, 32>;
            if (t256) kern = k_sbw_mfma<64, 32, 256>;
            if (h->sp32) {             // fp32-stored ring spectra (storage_f32 = 2; sx_create guarantees the matrix-core kernel applies)
                auto kf = t256 ? k_sbw_mfma<64, 32, 256, float> : h->nz == 64 ? k_sbw_mfma<64, 64, 512, float>
                          : h->nz == 32 ? k_sbw_mfma<32, 64, 512, float> : k_sbw_mfma<128, 32, 512, float>;
                hipLaunchKernelGGL(kf, gw, dim3(t256 ? 256 : 512), 0, h->stream, reinterpret_cast<const float *>(h->d_Fl) + flo, h->d_Btile + blo, h->d_phi,
                                   h->d_wq, h->d_CB, h->ncells, h->V, h->Zb, h->K2, h->C, cps);
            } else
            hipLaunchKernelGGL(kern, gw, dim3(t256 ? 256 : 512), 0, h->stream, h->d_Fl + flo, h->d_Btile + blo, h->d_phi, h->d_wq, h->d_CB, h->ncells,
                               h->V, h->Zb, h->K2, h->C, cps);
            HIPCHK(hipGetLastError());
            timer_end(h);
            return;
        }
        dim3 g((h->K2 + 63) / 64, h->V, h->nbt);
        hipLaunchKernelGGL(k_sbz, g, dim3(64, 4), sizeof(double) * 64 * h->nz, h->stream, h->d_Fl, h->d_Btile, h->d_phi, h->d_wq,
                           h->d_CB, h->ncells, h->V, h->nz, h->Zb, h->K2, h->C);
        HIPCHK(hipGetLastError());
        timer_end(h);
        return;
    }
    const int id = timer_id(h, "k_sb");
    timer_begin(h, id);
    const int64_t plane = (int64_t)h->V * h->nz * h->K2;
    dim3 g((unsigned)((plane + 255) / 256), h->nbt);
    hipLaunchKernelGGL(k_sb, g, dim3(256), 0, h->stream, h->d_Fl, h->d_Btile, h->d_phi, h->d_wq, h->ncells, plane);
    HIPCHK(hipGetLastError());
    timer_end(h);
}


void launch_solve(sx_handle *h) {
    const int id = timer_id(h, "k_solve");
    timer_begin(h, id);
    dim3 g((h->K2 > 1 ? (h->K2 / 2 - 1 + 63) / 64 : 0) + 1, h->V * h->Zb);
    // few right-hand sides (the R grid's one column, RZ grids, small RL patches): LDS-staged parallel cyclic reduction (sx_pcr.hip)
    // instead of one wave's serial recurrence per 64 columns
    const bool contiguous = (h->d_Bsrc == h->d_Bfull);
    const int ngroups = contiguous ? h->v_cnt * h->Zb : h->V * h->Zb;
    if (pcr_wanted(h, (int64_t)ngroups * (h->K2 > 1 ? h->K2 - 1 : 1))) {
        const int64_t clo = contiguous ? (int64_t)h->v_lo * h->Zb * h->K2 : 0;
        launch_solve_pcr(h, contiguous, h->d_Bsrc + clo, h->d_rowoff, h->d_neg1, h->d_A + clo, h->d_aoff, h->d_neg1,
                         contiguous ? h->v_lo * h->Zb : 0, ngroups, h->C);
        timer_end(h);
        return;
    }
    if (contiguous) {   // internal contiguous B: no offset tables needed; the variable window through the base pointers
        const int64_t clo = (int64_t)h->v_lo * h->Zb * h->K2;
        g.y = h->v_cnt * h->Zb;
        hipLaunchKernelGGL(k_solve<true>, g, dim3(64), sizeof(double) * 4 * h->b_rDim, h->stream, h->d_Bsrc + clo, h->d_rowoff, h->d_neg1, h->d_A + clo, h->d_aoff, h->d_neg1,
                           h->d_cls, h->d_cmeta, h->d_gl, h->d_gr, h->d_Lband, h->d_Ldinv, h->d_Larrow, h->b_rDim, h->Zb, h->K2, h->v_lo * h->Zb,
                           h->C);
    }
    else
        hipLaunchKernelGGL(k_solve<false>, g, dim3(64), sizeof(double) * 8 * h->b_rDim, h->stream, h->d_Bsrc, h->d_rowoff, h->d_neg1, h->d_A, h->d_aoff,
                           h->d_neg1, h->d_cls, h->d_cmeta, h->d_gl, h->d_gr, h->d_Lband, h->d_Ldinv, h->d_Larrow, h->b_rDim, h->Zb,
                           h->K2, 0, h->C);
    HIPCHK(hipGetLastError());
    timer_end(h);
}

// transposed solve: my column groups [g0, g1), right-hand sides from the all-to-all receive buffer, solution rows into
// the all-to-all send buffer (both [tile][row][my columns]; a row shared by two tiles is summed on input, duplicated on output)
void launch_solve_a2a(sx_handle *h, const double *recv, double *send) {
    const int id = timer_id(h, "k_solve");
    timer_begin(h, id);
    const int ng = h->a2a_g1 - h->a2a_g0;
    if (ng > 0 && pcr_wanted(h, (int64_t)ng * (h->K2 > 1 ? h->K2 - 1 : 1))) {
        launch_solve_pcr(h, false, recv, h->d_a2a_offA, h->d_a2a_offB, send, h->d_a2a_offA, h->d_a2a_offB, h->a2a_g0, ng, 0);
        timer_end(h);
        return;
    }
    if (ng > 0) {
        const int bx_pair = (h->K2 > 1 ? (h->K2 / 2 - 1 + 63) / 64 : 0) + 1;
        const bool single = (int64_t)bx_pair * ng < 768 && h->K2 > 2;        // fewer waves than SIMDs: one column per lane
#define A2A_ARGS recv, h->d_a2a_offA, h->d_a2a_offB, send, h->d_a2a_offA, h->d_a2a_offB, h->d_cls, h->d_cmeta, h->d_gl, h->d_gr,      \
                 h->d_Lband, h->d_Ldinv, h->d_Larrow, h->b_rDim, h->Zb, h->K2, h->a2a_g0, (int64_t)0
        if (single) hipLaunchKernelGGL((k_solve<false, false>), dim3((h->K2 - 2 + 63) / 64 + 1, ng), dim3(64), sizeof(double) * 8 * h->b_rDim, h->stream, A2A_ARGS);
        else hipLaunchKernelGGL((k_solve<false, true>), dim3(bx_pair, ng), dim3(64), sizeof(double) * 8 * h->b_rDim, h->stream, A2A_ARGS);
#undef A2A_ARGS
        HIPCHK(hipGetLastError());
    }
    timer_end(h);
}

void launch_a2a_pack(sx_handle *h, double *buf, int unpack) {
    const int id = timer_id(h, unpack ? "k_a2a_unpack" : "k_a2a_pack");
    timer_begin(h, id);
    dim3 g((unsigned)((h->C + 255) / 256), h->nbt);
    const double *arr = unpack ? h->d_A : h->d_Btile;
    hipLaunchKernelGGL(k_a2a_pack, g, dim3(256), 0, h->stream, arr, buf, h->d_a2a_owner, h->d_a2a_soff, h->d_a2a_cw, h->d_a2a_cs,
                       h->K2, h->C, unpack, (int64_t)(unpack ? h->cell0 : 0));
    HIPCHK(hipGetLastError());
    timer_end(h);
}

void launch_halo_add(sx_handle *h, const double *recv) {
    const int id = timer_id(h, "k_halo_add");
    timer_begin(h, id);
    const int64_t n = 3 * h->C;
    hipLaunchKernelGGL(k_halo_add, grid1(n, 256), dim3(256), 0, h->stream, h->d_Btile, recv, n);
    HIPCHK(hipGetLastError());
    timer_end(h);
}

void launch_max_abs(sx_handle *h, unsigned long long *d_out) {
    HIPCHK(hipMemsetAsync(d_out, 0, sizeof(unsigned long long) * h->V, h->stream));
    hipLaunchKernelGGL(k_max_abs, dim3(512, h->V), dim3(256), 0, h->stream, h->d_np1, h->N, d_out);
    HIPCHK(hipGetLastError());
}

void launch_nan_check(sx_handle *h) {
    HIPCHK(hipMemsetAsync(h->d_flag, 0, sizeof(int), h->stream));
    // checkCFL (src/semiimplicit.jl:737-751) scans physical[:, v, 1] right after a tileTransform!.  Here the scan runs
    // over var_np1 = value + ts * tendency of the last step (the initial values before the first step): a NaN in any
    // value propagates into it, and unlike `physical` it is complete after every sx_advance (slot masks and the
    // node-space inverse leave parts of `physical` untouched between outputs).
    const int64_t n = (int64_t)h->V * h->N;
    hipLaunchKernelGGL(k_nan_check, dim3(2048), dim3(256), 0, h->stream, h->d_np1, n, h->d_flag);
    HIPCHK(hipGetLastError());
}

}  // namespace sx
