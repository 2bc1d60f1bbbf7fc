// Batched Stockham FFT ring kernels for power-of-two ring lengths (uniform ring tables, e.g. the 256-point
// "perf shape").  Rings of any other length use the direct truncated DFT in sx_kernels.hip.
//
// One workgroup = (z-chunk of 16 levels, variable, ring).  Two vertical levels are packed into one complex
// transform (level 2p -> real part, 2p+1 -> imaginary part), so a chunk needs 8 complex FFTs per derivative slot,
// all resident in LDS at once; results leave LDS as full 128-byte lines of the reference physical layout
// (z innermost).  Radix-4 autosort passes (+ one radix-2 pass when log2 L is odd), twiddles from an LDS table.
#include "sx_internal.hpp"

namespace sx {

#define HIPCHK2(x)                                                                                  \
    do {                                                                                            \
        hipError_t e_ = (x);                                                                        \
        if (e_ != hipSuccess) set_error(std::string(#x) + ": " + hipGetErrorString(e_));            \
    } while (0)

constexpr int FZC = 16;        // z levels per workgroup
constexpr int FNP = FZC / 2;   // complex transforms per slot and workgroup
constexpr int FTHREADS = 512;
constexpr int SKEW = 2;        // complex elements of skew between transform buffers (LDS bank spreading)

__device__ __forceinline__ double2 cmul(double2 a, double2 b) { return make_double2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x); }
__device__ __forceinline__ double2 cadd(double2 a, double2 b) { return make_double2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ double2 csub(double2 a, double2 b) { return make_double2(a.x - b.x, a.y - b.y); }

// All FNP transforms of the workgroup advance together; transform f uses threads [f*T, (f+1)*T), T = L/4.
// src/dst: base of the two buffer sets, transform f at offset f*(L+SKEW). Returns the set holding the result.
template <int SIGN>
__device__ double2 *stockham_pow2(double2 *src, double2 *dst, const double2 *tw, int L, int logL, int f, int t, bool active) {
    const int T = L >> 2;
    double2 *a = src + f * (L + SKEW), *b = dst + f * (L + SKEW);
    int Ns = 1;
    for (int p = 0; p < (logL >> 1); p++) {
        if (active) {
            const int k = t & (Ns - 1);
            double2 v0 = a[t], v1 = a[t + T], v2 = a[t + 2 * T], v3 = a[t + 3 * T];
            if (Ns > 1) {
                const int s = k * (L / (4 * Ns));
                double2 w1 = tw[s], w2 = tw[2 * s], w3 = tw[3 * s];
                if (SIGN < 0) { w1.y = -w1.y; w2.y = -w2.y; w3.y = -w3.y; }
                v1 = cmul(v1, w1); v2 = cmul(v2, w2); v3 = cmul(v3, w3);
            }
            const double2 t0 = cadd(v0, v2), t1 = csub(v0, v2), t2 = cadd(v1, v3), d = csub(v1, v3);
            const double2 t3 = SIGN > 0 ? make_double2(-d.y, d.x) : make_double2(d.y, -d.x);   // (+/- i) * d
            const int j0 = ((t - k) << 2) + k;
            b[j0] = cadd(t0, t2);
            b[j0 + Ns] = cadd(t1, t3);
            b[j0 + 2 * Ns] = csub(t0, t2);
            b[j0 + 3 * Ns] = csub(t1, t3);
        }
        __syncthreads();
        double2 *tmp = a; a = b; b = tmp;
        tmp = src; src = dst; dst = tmp;
        Ns <<= 2;
    }
    if (logL & 1) {       // final radix-2 pass, Ns = L/2
        if (active) {
            for (int q = 0; q < 2; q++) {
                const int j = t + q * T;
                double2 w = tw[j];
                if (SIGN < 0) w.y = -w.y;
                const double2 v0 = a[j], v1 = cmul(a[j + 2 * T], w);
                b[j] = cadd(v0, v1);
                b[j + Ns] = csub(v0, v1);
            }
        }
        __syncthreads();
        double2 *tmp = src; src = dst; dst = tmp;
    }
    return src;
}

// ------------------------------------------------------------------------------------------------ inverse
__global__ void __launch_bounds__(FTHREADS)
k_rl_inverse_fft(const double *__restrict__ Az, double *__restrict__ phys, const double *__restrict__ phi,
                 const int *__restrict__ kmaxr, const int64_t *__restrict__ pstart, const double2 *__restrict__ twg,
                 const int64_t *__restrict__ phoff, const double2 *__restrict__ ph, const int *__restrict__ slotmask,
                 int V, int nz, int nsz, int K2, int nrings, int64_t N, int64_t azrow, int L, int logL,
                 int s_u, int s_r, int s_rr, int s_l, int s_ll, int s_z, int s_zz) {
    extern __shared__ double2 smf[];
    const int ring = blockIdx.z, v = blockIdx.y, z0 = blockIdx.x * FZC;
    const int zc = min(FZC, nz - z0);
    const int km = kmaxr[ring];
    const int T = L >> 2, tid = threadIdx.x;
    const int f = tid / T, t = tid - f * T;
    const bool active = f < FNP;
    double2 *tw = smf;                              // [L]
    double2 *bufA = smf + L, *bufB = bufA + FNP * (L + SKEW);
    for (int j = tid; j < L; j += FTHREADS) tw[j] = twg[j];
    const int j0 = ring / MUBAR;
    const double2 *phr = ph + phoff[ring];
    const int64_t p0 = pstart[ring];
    const int mask = slotmask[v];
    // output slot table: (slot index, sz, radial derivative d, lambda derivative ld)
    const int slots[7] = {s_u, s_r, s_rr, s_l, s_ll, s_z, s_zz};
    const int szs[7] = {0, 0, 0, 0, 0, 1, 2}, ds[7] = {0, 1, 2, 0, 0, 0, 0}, lds[7] = {0, 0, 0, 1, 2, 0, 0};
    for (int q = 0; q < 7; q++) {
        const int slot = slots[q];
        if (slot < 0 || szs[q] >= nsz || !((mask >> slot) & 1)) continue;      // uniform across the workgroup
        const int ld = lds[q];
        __syncthreads();
        if (active) {
            const double *pf = phi + ((int64_t)ds[q] * nrings + ring) * 4;
            const double f0 = pf[0], f1 = pf[1], f2 = pf[2], f3 = pf[3];
            const int za = 2 * f, zb = 2 * f + 1;
            const bool hasa = za < zc, hasb = zb < zc;
            const double *a0 = Az + (int64_t)j0 * azrow + (((int64_t)v * nsz + szs[q]) * nz + (z0 + (hasa ? za : 0))) * K2;
            const double *b0 = Az + (int64_t)j0 * azrow + (((int64_t)v * nsz + szs[q]) * nz + (z0 + (hasb ? zb : 0))) * K2;
            double2 *X = bufA + f * (L + SKEW);
            for (int k = t; k <= L / 2; k += T) {
                double2 za_c = make_double2(0.0, 0.0), zb_c = make_double2(0.0, 0.0);
                if (k <= km) {
                    if (k == 0) {
                        if (ld == 0) {
                            if (hasa) za_c.x = f0 * a0[0] + f1 * a0[azrow] + f2 * a0[2 * azrow] + f3 * a0[3 * azrow];
                            if (hasb) zb_c.x = f0 * b0[0] + f1 * b0[azrow] + f2 * b0[2 * azrow] + f3 * b0[3 * azrow];
                        }
                    } else {
                        const int bb = 2 * k - 1;
                        if (hasa) {
                            za_c.x = f0 * a0[bb] + f1 * a0[azrow + bb] + f2 * a0[2 * azrow + bb] + f3 * a0[3 * azrow + bb];
                            za_c.y = f0 * a0[bb + 1] + f1 * a0[azrow + bb + 1] + f2 * a0[2 * azrow + bb + 1] + f3 * a0[3 * azrow + bb + 1];
                        }
                        if (hasb) {
                            zb_c.x = f0 * b0[bb] + f1 * b0[azrow + bb] + f2 * b0[2 * azrow + bb] + f3 * b0[3 * azrow + bb];
                            zb_c.y = f0 * b0[bb + 1] + f1 * b0[azrow + bb + 1] + f2 * b0[2 * azrow + bb + 1] + f3 * b0[3 * azrow + bb + 1];
                        }
                        const double2 w = phr[k];                 // e^{+ik off}
                        za_c = cmul(za_c, w);
                        zb_c = cmul(zb_c, w);
                        if (ld == 1) {                            // multiply by ik
                            za_c = make_double2(-k * za_c.y, k * za_c.x);
                            zb_c = make_double2(-k * zb_c.y, k * zb_c.x);
                        } else if (ld == 2) {                     // multiply by -k^2
                            const double kk = -(double)k * k;
                            za_c.x *= kk; za_c.y *= kk; zb_c.x *= kk; zb_c.y *= kk;
                        }
                    }
                }
                // W = Za + i Zb at bin k, conj(Za) + i conj(Zb) at bin L - k
                X[k] = make_double2(za_c.x - zb_c.y, za_c.y + zb_c.x);
                if (k > 0 && k < L - k) X[L - k] = make_double2(za_c.x + zb_c.y, zb_c.x - za_c.y);
            }
        }
        __syncthreads();
        double2 *res = stockham_pow2<+1>(bufA, bufB, tw, L, logL, f, t, active);
        // copy-out: full 128-byte lines (16 levels) per ring point
        double *out = phys + ((int64_t)slot * V + v) * N + p0 * nz + z0;
        for (int o = tid; o < L * FZC; o += FTHREADS) {
            const int zz = o & (FZC - 1), l = o >> 4;
            if (zz < zc) {
                const double2 r = res[(zz >> 1) * (L + SKEW) + l];
                out[(int64_t)l * nz + zz] = (zz & 1) ? r.y : r.x;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------ forward
__global__ void __launch_bounds__(FTHREADS)
k_fl_forward_fft(const double *__restrict__ np1, double *__restrict__ Fl, const int *__restrict__ kmaxr,
                 const int64_t *__restrict__ pstart, const double2 *__restrict__ twg, const int64_t *__restrict__ phoff,
                 const double2 *__restrict__ ph, int V, int nz, int K2, int64_t N, int L, int logL) {
    extern __shared__ double2 smf[];
    const int ring = blockIdx.z, v = blockIdx.y, z0 = blockIdx.x * FZC;
    const int zc = min(FZC, nz - z0);
    const int km = kmaxr[ring];
    const int T = L >> 2, tid = threadIdx.x;
    const int f = tid / T, t = tid - f * T;
    const bool active = f < FNP;
    double2 *tw = smf;
    double2 *bufA = smf + L, *bufB = bufA + FNP * (L + SKEW);
    for (int j = tid; j < L; j += FTHREADS) tw[j] = twg[j];
    const int64_t p0 = pstart[ring];
    const double *x = np1 + (int64_t)v * N + p0 * nz + z0;
    double *ba = (double *)bufA;
    for (int o = tid; o < L * FZC; o += FTHREADS) {
        const int zz = o & (FZC - 1), l = o >> 4;
        const double val = (zz < zc) ? x[(int64_t)l * nz + zz] : 0.0;
        ba[2 * ((zz >> 1) * (L + SKEW) + l) + (zz & 1)] = val;
    }
    __syncthreads();
    double2 *res = stockham_pow2<-1>(bufA, bufB, tw, L, logL, f, t, active);
    if (!active) return;
    const double2 *W = res + f * (L + SKEW);
    const double2 *phr = ph + phoff[ring];
    const double inv = 1.0 / L;
    const int za = 2 * f, zb = 2 * f + 1;
    double *oa = Fl + (((int64_t)ring * V + v) * nz + z0 + za) * K2;
    double *ob = oa + K2;
    for (int k = t; k <= km; k += T) {
        const double2 wk = W[k], wn = W[(L - k) & (L - 1)];
        // Xa = (W_k + conj W_{-k}) / 2,  Xb = (W_k - conj W_{-k}) / (2i)
        double2 xa = make_double2(0.5 * (wk.x + wn.x), 0.5 * (wk.y - wn.y));
        double2 xb = make_double2(0.5 * (wk.y + wn.y), -0.5 * (wk.x - wn.x));
        if (k == 0) {
            if (za < zc) oa[0] = xa.x * inv;
            if (zb < zc) ob[0] = xb.x * inv;
        } else {
            const double2 w = make_double2(phr[k].x, -phr[k].y);     // e^{-ik off}
            xa = cmul(xa, w);
            xb = cmul(xb, w);
            if (za < zc) { oa[2 * k - 1] = xa.x * inv; oa[2 * k] = xa.y * inv; }
            if (zb < zc) { ob[2 * k - 1] = xb.x * inv; ob[2 * k] = xb.y * inv; }
        }
    }
}

// ------------------------------------------------------------------------------------------------ launchers
static int ilog2(int n) { int l = 0; while ((1 << l) < n) l++; return l; }

bool fft_path_ok(const sx_handle *h) {
    const int L = h->uniform_L;
    return h->has_l && L >= 16 && L <= 256 && (L & (L - 1)) == 0;   // 8 transforms x L/4 threads <= 512
}

static size_t fft_lds(int L) { return sizeof(double2) * ((size_t)L + 2 * (size_t)FNP * (L + SKEW)); }

void launch_rl_inverse_fft(sx_handle *h, const int *d_mask) {
    const int id = timer_id(h, "k_rl_inverse");
    timer_begin(h, id);
    const int L = h->uniform_L;
    const double *az = h->has_z ? h->d_Az : h->d_A + (int64_t)h->cell0 * h->C;
    const int64_t azrow = h->has_z ? (int64_t)h->V * 3 * h->nz * h->K2 : h->C;
    dim3 g((h->nz + FZC - 1) / FZC, h->V, h->nrings);
    hipLaunchKernelGGL(k_rl_inverse_fft, g, dim3(FTHREADS), fft_lds(L), h->stream, az, h->d_phys, h->d_phi, h->d_kmax,
                       h->d_pstart, h->d_tw, h->d_phoff, h->d_ph, d_mask, h->V, h->nz, h->nsz, h->K2, h->nrings, h->N, azrow, L,
                       ilog2(L), h->slot[0], h->slot[1], h->slot[2], h->slot[3], h->slot[4], h->slot[5], h->slot[6]);
    HIPCHK2(hipGetLastError());
    timer_end(h);
}

void launch_fl_forward_fft(sx_handle *h) {
    const int id = timer_id(h, "k_fl_forward");
    timer_begin(h, id);
    const int L = h->uniform_L;
    dim3 g((h->nz + FZC - 1) / FZC, h->V, h->nrings);
    hipLaunchKernelGGL(k_fl_forward_fft, g, dim3(FTHREADS), fft_lds(L), h->stream, h->d_np1, h->d_Fl, h->d_kmax, h->d_pstart,
                       h->d_tw, h->d_phoff, h->d_ph, h->V, h->nz, h->K2, h->N, L, ilog2(L));
    HIPCHK2(hipGetLastError());
    timer_end(h);
}

}  // namespace sx
