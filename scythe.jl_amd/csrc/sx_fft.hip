// Batched Stockham FFT ring kernels for power-of-two ring lengths (uniform ring tables, e.g. the 256-point
// "perf shape").  Rings of any other length use the direct truncated DFT in sx_kernels.hip.
//
// One workgroup = (z-chunk of 16 levels, variable, ring).  Two vertical levels are packed into one complex
// transform (level 2p -> real part, 2p+1 -> imaginary part): 8 complex transforms per derivative slot and chunk.
// A transform of length L is owned by min(L/4, 64) lanes of ONE wave, so every pass is wave-local: in-place radix-4
// autosort passes through a 16 B x L LDS region (+ one radix-2 pass when log2 L is odd), twiddles held in registers,
// no workgroup barrier inside a transform (at L = 512 a lane carries two butterflies per pass).
//   inverse: the last pass stores straight to the reference physical layout (z innermost): each lane writes the
//            16-byte (z, z+1) pair of its ring points; the 8 waves of the workgroup complete every 128-byte line.
//   forward: the workgroup first stages the [ring point][16 levels] tile through LDS with full 128-byte loads.
#include "sx_internal.hpp"
#include <cstdlib>
#include <cstdio>
#include <vector>

namespace sx {

#define HIPCHK2(x)                                                                                  \
    do {                                                                                            \
        hipError_t e_ = (x);                                                                        \
        if (e_ != hipSuccess) set_error(std::string(#x) + ": " + hipGetErrorString(e_) + " (" __FILE__ ":" + std::to_string(__LINE__) + ")"); \
    } while (0)

// z levels per workgroup (FZC) and complex transforms per slot and workgroup (FNP = FZC / 2).  A transform of length L is
// owned by LPT = min(L / 4, 64) lanes of ONE wave, so every pass is wave-local; for L = 512 a lane carries NB = 2 radix-4
// butterflies per pass (and NK = 4 wavenumbers while staging) instead of spreading the transform over two waves that had to
// meet at a workgroup barrier after every pass.  16 levels per workgroup at every L, so that a ring point's levels leave and
// arrive as whole 128-byte lines (8 levels per workgroup - 64-byte pieces - held the L = 512 kernels at 3.2 TB/s).
// SKEW: complex elements between transform regions (bank spreading when L < 256; 0 at L = 512 so that the two LDS sets
// of the inverse are exactly 64 KB).
// HL = 1 halves the lanes per transform (two butterflies per lane and pass already at L = 256: 256-thread workgroups whose
// waves carry twice the independent work; used by the inverse kernels, see launch_inv).
template <int LOGL, int HL = 0> struct FftCfg {
    static constexpr int FZC = 16, FNP = FZC / 2, LOGZ = 4, SKEW = (LOGL <= 8) ? 2 : 0;
    static constexpr int L4 = (1 << LOGL) / 4;
    static constexpr int LPT = (L4 > 64 ? 64 : L4) >> HL;      // lanes per transform
    static constexpr int NB = L4 / LPT;                // radix-4 butterflies per lane and pass
    static constexpr int NK = 2 * NB;                  // wavenumbers (k < L / 2) per lane
};

// phase stamps of the diagnostic build (-DSX_PHASES): [workgroup][8] cycle counts of the NODE inverse kernel, dumped by sx_destroy
#ifdef SX_PHASES
__device__ long long *g_fft_dbg = nullptr;
#define FFT_STAMP(i) do { if (NODE && threadIdx.x == 0 && g_fft_dbg) g_fft_dbg[((int64_t)(blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) * 8 + (i)] = (long long)__builtin_readcyclecounter(); } while (0)
#else
#define FFT_STAMP(i) do { } while (0)
#endif

__device__ __forceinline__ double2 cmul(double2 a, double2 b) { return make_double2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x); }
__device__ __forceinline__ double2 cadd(double2 a, double2 b) { return make_double2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ double2 csub(double2 a, double2 b) { return make_double2(a.x - b.x, a.y - b.y); }
__device__ __forceinline__ double2 cconj(double2 a) { return make_double2(a.x, -a.y); }
__device__ __forceinline__ double2 cmuli(double2 a, int sign) { return sign > 0 ? make_double2(-a.y, a.x) : make_double2(a.y, -a.x); }

// orders LDS traffic between the lanes of one wave (DS instructions of a wave execute in issue order)
__device__ __forceinline__ void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// workgroup barrier that orders LDS traffic only: outstanding global stores (the previous slot's copy-out) and loads
// stay in flight across it (a plain __syncthreads() would also drain vmcnt)
__device__ __forceinline__ void lds_barrier() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}

template <int LOGL, int HL = 0>
struct Twiddles {
    static constexpr int L = 1 << LOGL, NP4 = LOGL / 2, LPT = FftCfg<LOGL, HL>::LPT, NB = FftCfg<LOGL, HL>::NB;
    double2 w1[NP4 > 1 ? NP4 - 1 : 1][NB]; // w2 = w1^2, w3 = w1^3 are formed on the fly (registers are the scarce resource)
    double2 r2[2 * NB];                    // final radix-2 pass (odd log2 L): butterflies j = t + c LPT
    // SIGN = +1: e^{+i...} (inverse), -1: forward.  t = lane within the transform.  The radix-4 twiddle of butterfly
    // j = t + b LPT depends on j mod Ns only: while Ns <= LPT all NB butterflies of a lane share it ([p][0]; the other
    // entries are never touched and cost no register), a pass with Ns > LPT keeps one per butterfly.
    template <int SIGN>
    __device__ void init(const double2 *__restrict__ twg, int t) {
        int Ns = 4;
#pragma unroll
        for (int p = 1; p < NP4; p++) {
#pragma unroll
            for (int b = 0; b < (Ns > LPT ? NB : 1); b++) {
                const int s = ((t + b * LPT) & (Ns - 1)) * (L / (4 * Ns));
                w1[p - 1][b] = twg[s];
                if (SIGN < 0) w1[p - 1][b].y = -w1[p - 1][b].y;
            }
            Ns <<= 2;
        }
        if (LOGL & 1) {
#pragma unroll
            for (int c = 0; c < 2 * NB; c++) {
                r2[c] = twg[t + c * LPT];
                if (SIGN < 0) r2[c].y = -r2[c].y;
            }
        }
    }
};

// In-place radix-4 passes on X (L complex, owned by LPT lanes of one wave, lane index t, NB butterflies per lane).
// All passes but the last are done here; `last` receives the outputs of the final pass:
//   last(index, value) for the 4 (radix-4 ending) or 2 (radix-2 ending) outputs of each butterfly of this lane.
struct NoSink {
    __device__ void operator()(int, double2) const {}
};

// TO_LDS = true: the final pass is written back to X as well (natural order), `last` is not called.
template <int LOGL, int SIGN, bool TO_LDS = false, class F = NoSink, int HL = 0>
__device__ __forceinline__ void fft_inplace(double2 *X, const Twiddles<LOGL, HL> &tw, int t, bool active, F last = F()) {
    constexpr int L = 1 << LOGL, NBF = L / 4, NP4 = LOGL / 2, LPT = FftCfg<LOGL, HL>::LPT, NB = FftCfg<LOGL, HL>::NB;
    int Ns = 1;
#pragma unroll
    for (int p = 0; p < NP4; p++) {
        double2 y0[NB], y1[NB], y2[NB], y3[NB];
        int j0[NB];
        if (active) {
#pragma unroll
            for (int b = 0; b < NB; b++) {
                const int j = t + b * LPT;
                double2 v0 = X[j], v1 = X[j + NBF], v2 = X[j + 2 * NBF], v3 = X[j + 3 * NBF];
                if (p > 0) {
                    const double2 w1 = tw.w1[p - 1][Ns > LPT ? b : 0], w2 = cmul(w1, w1), w3 = cmul(w2, w1);
                    v1 = cmul(v1, w1); v2 = cmul(v2, w2); v3 = cmul(v3, w3);
                }
                const double2 t0 = cadd(v0, v2), t1 = csub(v0, v2), t2 = cadd(v1, v3), t3 = cmuli(csub(v1, v3), SIGN);
                y0[b] = cadd(t0, t2); y1[b] = cadd(t1, t3); y2[b] = csub(t0, t2); y3[b] = csub(t1, t3);
                const int k = j & (Ns - 1);        // = t & (Ns - 1) while Ns <= LPT
                j0[b] = ((j - k) << 2) + k;
            }
        }
        if (!TO_LDS && p == NP4 - 1 && !(LOGL & 1)) {
            if (active) {
#pragma unroll
                for (int b = 0; b < NB; b++) { last(j0[b], y0[b]); last(j0[b] + Ns, y1[b]); last(j0[b] + 2 * Ns, y2[b]); last(j0[b] + 3 * Ns, y3[b]); }
            }
            return;
        }
        wave_sync();      // every lane has read its inputs before any lane overwrites them
        if (active) {
#pragma unroll
            for (int b = 0; b < NB; b++) { X[j0[b]] = y0[b]; X[j0[b] + Ns] = y1[b]; X[j0[b] + 2 * Ns] = y2[b]; X[j0[b] + 3 * Ns] = y3[b]; }
        }
        wave_sync();
        Ns <<= 2;
    }
    if (LOGL & 1) {       // final radix-2 pass, Ns = L/2: butterfly j works on X[j], X[j + L/2] in place
        double2 lo[2 * NB], hi[2 * NB];
        if (active) {
#pragma unroll
            for (int c = 0; c < 2 * NB; c++) {
                const int j = t + c * LPT;
                const double2 a0 = X[j], a1 = cmul(X[j + L / 2], tw.r2[c]);
                lo[c] = cadd(a0, a1); hi[c] = csub(a0, a1);
            }
        }
        if (TO_LDS) {
            wave_sync();
            if (active) {
#pragma unroll
                for (int c = 0; c < 2 * NB; c++) { X[t + c * LPT] = lo[c]; X[t + c * LPT + L / 2] = hi[c]; }
            }
            wave_sync();
        } else if (active) {
#pragma unroll
            for (int c = 0; c < 2 * NB; c++) { last(t + c * LPT, lo[c]); last(t + c * LPT + L / 2, hi[c]); }
        }
    }
}

// ------------------------------------------------------------------------------------------------ register-resident passes (L = 256)
// Round 4.  The four radix-4 passes of a 256-point transform without LDS in between: decimation in frequency, element index =
// (register q, lane); after each of the first three passes a 4 x 4 transpose between the register index and two lane bits brings
// the next digit into the registers - lane bits 5 / 4 by v_permlane32_swap / v_permlane16_swap (gfx950: one instruction swaps a
// dword both ways), bits 3 / 2 by DPP row_ror:8 / row_shl:4 + row_shr:4 under bank masks (one v_mov_dpp per dword), bits 1 / 0
// by quad_perm + select.  profiles/micro/fft_exchange.hip (profiles/r04/micro_fft_exchange.txt): 1,145 cycles per transform and
// wave against 2,850 through LDS (staging write and copy-out read included there), 1.8 x the transforms per second per CU.
//   in:  v[q] = x[lane + 64 q]      out: v[q0] = X[l],  l = (lane >> 4) + 4 ((lane >> 2) & 3) + 16 (lane & 3) + 64 q0
typedef unsigned int fft_u32;
struct FftQ { fft_u32 d[4]; };          // one complex double as 4 dwords
__device__ __forceinline__ FftQ fft_toq(double2 v) { return FftQ{{(fft_u32)__double2loint(v.x), (fft_u32)__double2hiint(v.x), (fft_u32)__double2loint(v.y), (fft_u32)__double2hiint(v.y)}}; }
__device__ __forceinline__ double2 fft_fromq(const FftQ &q) { return make_double2(__hiloint2double((int)q.d[1], (int)q.d[0]), __hiloint2double((int)q.d[3], (int)q.d[2])); }

// 2 x 2 step between the register pair (a: register bit clear, b: set) and lane bit BIT:
//   lanes with the bit clear: b <- partner's a;   lanes with the bit set: a <- partner's b
template <int BIT>
__device__ __forceinline__ void fft_swap2(double2 &a, double2 &b, int lane) {
    FftQ A = fft_toq(a), B = fft_toq(b);
#pragma unroll
    for (int i = 0; i < 4; i++) {
        if (BIT == 5) {
            const auto r = __builtin_amdgcn_permlane32_swap(A.d[i], B.d[i], false, false);     // A[32..63] <-> B[0..31]
            A.d[i] = r[0]; B.d[i] = r[1];
        } else if (BIT == 4) {
            const auto r = __builtin_amdgcn_permlane16_swap(A.d[i], B.d[i], false, false);     // A[odd rows of 16] <-> B[even rows]
            A.d[i] = r[0]; B.d[i] = r[1];
        } else if (BIT == 3) {           // row_ror:8 = lane ^ 8 within a row of 16; banks 0, 1 = lanes 0-7 of the row
            const fft_u32 nb = __builtin_amdgcn_update_dpp(B.d[i], A.d[i], 0x128, 0xF, 0x3, false);
            const fft_u32 na = __builtin_amdgcn_update_dpp(A.d[i], B.d[i], 0x128, 0xF, 0xC, false);
            A.d[i] = na; B.d[i] = nb;
        } else if (BIT == 2) {           // banks 0, 2 (bit 2 clear) read lane + 4 (row_shl:4), banks 1, 3 read lane - 4 (row_shr:4)
            const fft_u32 nb = __builtin_amdgcn_update_dpp(B.d[i], A.d[i], 0x104, 0xF, 0x5, false);
            const fft_u32 na = __builtin_amdgcn_update_dpp(A.d[i], B.d[i], 0x114, 0xF, 0xA, false);
            A.d[i] = na; B.d[i] = nb;
        } else {                         // quad_perm: lane ^ 2 = [2,3,0,1] (0x4E), lane ^ 1 = [1,0,3,2] (0xB1)
            constexpr int ctl = BIT == 1 ? 0x4E : 0xB1;
            const fft_u32 pa = __builtin_amdgcn_mov_dpp(A.d[i], ctl, 0xF, 0xF, false);
            const fft_u32 pb = __builtin_amdgcn_mov_dpp(B.d[i], ctl, 0xF, 0xF, false);
            const bool set = (lane >> BIT) & 1;
            const fft_u32 na = set ? pb : A.d[i], nb = set ? B.d[i] : pa;
            A.d[i] = na; B.d[i] = nb;
        }
    }
    a = fft_fromq(A); b = fft_fromq(B);
}
// 4 x 4 transpose: register bit 1 <-> lane bit HI, register bit 0 <-> lane bit HI - 1
template <int HI>
__device__ __forceinline__ void fft_transpose4(double2 (&v)[4], int lane) {
    fft_swap2<HI>(v[0], v[2], lane); fft_swap2<HI>(v[1], v[3], lane);
    fft_swap2<HI - 1>(v[0], v[1], lane); fft_swap2<HI - 1>(v[2], v[3], lane);
}
// decimation-in-frequency radix-4 butterfly, inverse sign: y_q = (sum_p a_p i^{pq}) w^q
__device__ __forceinline__ void fft_bfly_dif(double2 (&v)[4], double2 w1, bool tw) {
    const double2 t0 = cadd(v[0], v[2]), t1 = csub(v[0], v[2]), t2 = cadd(v[1], v[3]), t3 = cmuli(csub(v[1], v[3]), +1);
    const double2 y0 = cadd(t0, t2);
    double2 y1 = cadd(t1, t3), y2 = csub(t0, t2), y3 = csub(t1, t3);
    if (tw) {
        const double2 w2 = cmul(w1, w1), w3 = cmul(w2, w1);
        y1 = cmul(y1, w1); y2 = cmul(y2, w2); y3 = cmul(y3, w3);
    }
    v[0] = y0; v[1] = y1; v[2] = y2; v[3] = y3;
}
// w[0] = e^{+2 pi i lane / 256}, w[1] = e^{+2 pi i (lane & 15) / 64}, w[2] = e^{+2 pi i (lane & 3) / 16}
__device__ __forceinline__ void fft_reg256_inverse(double2 (&v)[4], const double2 (&w)[3], int lane) {
    fft_bfly_dif(v, w[0], true);
    fft_transpose4<5>(v, lane);
    fft_bfly_dif(v, w[1], true);
    fft_transpose4<3>(v, lane);
    fft_bfly_dif(v, w[2], true);
    fft_transpose4<1>(v, lane);
    fft_bfly_dif(v, w[0], false);
}
__device__ __forceinline__ double2 fft_bpermute(double2 v, int src_lane) {
    const FftQ q = fft_toq(v);
    FftQ r;
#pragma unroll
    for (int i = 0; i < 4; i++) r.d[i] = (fft_u32)__builtin_amdgcn_ds_bpermute(src_lane << 2, (int)q.d[i]);
    return fft_fromq(r);
}

// ------------------------------------------------------------------------------------------------ inverse
// The output planes are `physical` or the node-space array G: value slot fp64, derivative slots ST (Planes, sx_internal.hpp)
// (Re, Im) pair of a spectral intermediate stored as fp64 or fp32 (storage_f32 = 2), widened on load
__device__ __forceinline__ double2 ldpair(const double *p) { return *reinterpret_cast<const double2 *>(p); }
__device__ __forceinline__ double2 ldpair(const float *p) { const float2 v = *reinterpret_cast<const float2 *>(p); return make_double2((double)v.x, (double)v.y); }
__device__ __forceinline__ void stpair(double *p, double2 v) { *reinterpret_cast<double2 *>(p) = v; }
__device__ __forceinline__ void stpair(float *p, double2 v) { *reinterpret_cast<float2 *>(p) = make_float2((float)v.x, (float)v.y); }

// HL / SETS: lanes per transform halved (256-thread workgroups at L = 256) / LDS sets (2: the copy-out of a slot overlaps the
// next slot's transform inside the workgroup; 1: half the LDS, the overlap comes from more workgroups per CU instead).
// Only (0, 2) is launched: measured at the bench grid (round 3) HL = 1 with one set 0.148 / 0.153 ms (node / ring-wise), with two
// sets 0.160 / 0.173 ms, against 0.130 / 0.135 ms - 168 VGPRs with 18-40 spilled registers at three waves per SIMD; at 512 points
// (config 5) the node kernel with ONE set and 128 registers (two workgroups per CU instead of one): 1.60 against 1.22 ms.
// FUSE (node mode, L <= 256, b_zDim <= 64): the vertical inverse runs INSIDE this kernel - the workgroup forms its
// [16 levels x K2] coefficient slab as Mz[v][sz][16 x b_zDim] . A[node][v][b_zDim x K2] on the f64 matrix cores (8 waves x 2
// column tiles of 16 wavenumber blocks, operands straight from L2), through the LDS set the next slot is about to stage,
// into the lanes' registers; `Az` of the node-space units is never written or read (k_zinv then only serves the ring-wise rings).
typedef double fft_d4 __attribute__((ext_vector_type(4)));
template <int LOGL, int COPYOUT, bool NODE, class ST, class AT = double, int HL = 0, int SETS = 2, bool FUSE = false, bool REG = false>
__global__ void __launch_bounds__(512 >> HL, HL ? 3 : (LOGL <= 8 || SETS == 1) ? 4 : 2)      // waves per SIMD: at L = 512 one workgroup per CU with two LDS sets (2 x 64 KB, 4 wavenumbers per lane), two with one set
k_rl_inverse_fft(const AT *__restrict__ Az, Planes<ST> phys, const double *__restrict__ phi,
                 const int *__restrict__ kmaxr, const int64_t *__restrict__ pstart, const double2 *__restrict__ twg,
                 const int64_t *__restrict__ phoff, const double2 *__restrict__ ph, const int *__restrict__ slotmask,
                 int V, int nz, int nsz, int K2, int nrings, int64_t N, int64_t azrow,
                 int s_u, int s_r, int s_rr, int s_l, int s_ll, int s_z, int s_zz,
                 const double *__restrict__ Asrc = nullptr, int64_t Astride = 0, const double *__restrict__ MzT = nullptr, int Zb = 0,
                 int unit0 = 0) {
    // NODE: the "rings" are radial NODES (uniform ring tables only): one Az row per unit, no radial combination;
    // the output is the node-space array G that the equation-set kernel combines with the basis weights itself.
    constexpr int L = 1 << LOGL, T = FftCfg<LOGL, HL>::LPT, NK = FftCfg<LOGL, HL>::NK;
    constexpr int FZC = FftCfg<LOGL, HL>::FZC, FNP = FftCfg<LOGL, HL>::FNP, LOGZ = FftCfg<LOGL, HL>::LOGZ;
    // REG: the transform's LDS region only serves the copy-out; ring point l sits at l + (l >> 4), so that the 16 lanes of a row -
    // points 4 a + 16 b after the last register pass - fall into 16 different 16-byte bank groups (stride = 2 mod 16 for the readers)
    constexpr int SKEW = REG ? L / 16 + 2 : FftCfg<LOGL, HL>::SKEW;
    static_assert(!REG || ((LOGL == 8 || LOGL == 9) && HL == 0 && COPYOUT && !FUSE), "register-resident passes: 256- and 512-point transforms with copy-out");
    extern __shared__ double2 smf[];
    FFT_STAMP(0);
    int nslot = 0;
    // grid = (level chunk, ring or node, variable RANK): workgroups are dispatched in blockIdx order, so every unit of the
    // variable with the most requested slots goes first and the variables with few (or no) slots fill the tail of the launch
    const int ring = blockIdx.y + unit0, z0 = blockIdx.x * FZC;      // unit0: first ring / node of the launch
    int v = 0;
    for (int u = 0; u < V; u++) {
        const int cu = __popc(slotmask[u]);
        int before = 0;
        for (int w = 0; w < V; w++) { const int cw = __popc(slotmask[w]); before += (cw > cu || (cw == cu && w < u)) ? 1 : 0; }
        if (before == (int)blockIdx.z) v = u;
    }
    const int zc = min(FZC, nz - z0);
    const int km = kmaxr[ring];
    const int f = threadIdx.x / T, t = threadIdx.x - f * T;       // transform (z pair) and lane within it
    const int za = 2 * f, zb = 2 * f + 1;
    const bool active = (f < FNP) && (za < zc);
    const bool hasb = zb < zc;
    // two sets of FNP transform regions: slot i works in set i & 1, so the copy-out of slot i (global stores, never
    // waited for) overlaps the staging and transform of slot i + 1; one LDS-only workgroup barrier per slot
    int par = 0;
    // twiddles and phase factors are fetched BEHIND the first group's coefficient loads (below): the memory counter retires
    // in issue order, and the coefficients are what the first transform waits for.  (Phase stamps: a third of a
    // workgroup's time used to pass before its first transform started - three dependent round trips.)
    Twiddles<LOGL, HL> tw;
    bool setup_done = false;
    const int j0 = NODE ? ring : ring / MUBAR;
    const double2 *phr = ph + phoff[ring];
    const int64_t p0 = pstart[ring];
    const int mask = slotmask[v];
    // this lane's wavenumbers kq[q] = t + q T < L / 2 (two of them, four at L = 512)
    int kq[NK], kcq[NK];
    bool inq[NK];
    double2 phq[NK];                                               // node tables carry no phase offset
#pragma unroll
    for (int q = 0; q < NK; q++) {
        kq[q] = t + q * T;
        inq[q] = kq[q] <= km;
        kcq[q] = min(kq[q], K2 / 2 - 1);      // in-row address for lanes beyond the truncation (their values are dropped)
        phq[q] = make_double2(1.0, 0.0);
    }
    const bool pair_ok = ((nz & 1) == 0);                          // (z, z+1) pairs are 16-byte aligned

    // groups of output slots that share one radial combination: (sz, d) = (0,0): u, l, ll; (0,1): r; (0,2): rr;
    // (1,0): z; (2,0): zz.  The four radial rows are re-read (L2) per group so that only one combination is live.
    for (int grp = 0; grp < 5; grp++) {
        const int sz = grp < 3 ? 0 : grp - 2, d = grp < 3 ? grp : 0;
        if (sz >= nsz) break;
        // candidate slots of this group, indexed by the lambda-derivative order ld
        const int sl0 = grp == 0 ? s_u : grp == 1 ? s_r : grp == 2 ? s_rr : grp == 3 ? s_z : s_zz;
        const int sl1 = grp == 0 ? s_l : -1, sl2 = grp == 0 ? s_ll : -1;
        const bool n0 = sl0 >= 0 && ((mask >> sl0) & 1), n1 = sl1 >= 0 && ((mask >> sl1) & 1), n2 = sl2 >= 0 && ((mask >> sl2) & 1);
        if (!n0 && !n1 && !n2) continue;
        double2 aq[NK], bq[NK];
        if (FUSE) {
            // vertical inverse of this group's operator on the matrix cores: D[level][block] = sum_zm Mz[level][zm] A[zm][block]
            constexpr int KSM = 16, NWV = (FNP * T) / 64 > 0 ? (FNP * T) / 64 : 1;                 // K steps (b_zDim <= 64); waves of the workgroup
            constexpr int TPW = ((L / 16 > 0 ? L / 16 : 1) + NWV - 1) / NWV;                        // column tiles (16 wavenumber blocks) per wave
            constexpr int SST = 2 * (L + SKEW) * FNP / FZC;      // row stride (doubles) of the slab: the set's 16-byte elements as 16 rows
            const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, n16 = lane & 15, kk4 = lane >> 4;
            double *S = reinterpret_cast<double *>(smf + ((COPYOUT && SETS == 2) ? par * FNP * (L + SKEW) : 0));
            const double *arow = Asrc + (int64_t)j0 * Astride + (int64_t)v * Zb * K2;
            const double *mop = MzT + ((int64_t)v * 3 + sz) * Zb * nz + z0 + n16;
            constexpr int KH = 8;                              // operands in halves of 8 K steps: 32 VGPRs live instead of 64
#pragma unroll
            for (int tt = 0; tt < TPW; tt++) {
                const int tile = wave * TPW + tt;
                const int blk = min(tile * 16 + n16, K2 - 1);
                fft_d4 acc = {0.0, 0.0, 0.0, 0.0};
                // operand addresses walk forward by 4 rows per K step from an OPAQUE start: formed here, per tile and group
                // (as invariants of the group loop the 64 of them were hoisted and spilled)
                int64_t oa = (int64_t)kk4 * nz, ob = (int64_t)kk4 * K2 + blk;
                asm volatile("" : "+v"(oa), "+v"(ob));
                const double *pa = mop + oa, *pb = arow + ob;
#pragma unroll
                for (int k0 = 0; k0 < KSM; k0 += KH) {
                    if (4 * k0 >= Zb) break;
                    double am[KH], bm[KH];
#pragma unroll
                    for (int ks = 0; ks < KH; ks++) {
                        const int k = 4 * (k0 + ks) + kk4;
                        am[ks] = k < Zb ? *pa : 0.0;
                        bm[ks] = (k < Zb && tile * 16 < K2) ? *pb : 0.0;
                        pa += 4 * (int64_t)nz; pb += 4 * (int64_t)K2;
                    }
                    if (!setup_done) {                         // twiddles behind the first operands (see below)
                        asm volatile("" ::: "memory");
                        int tt2 = t;
                        asm volatile("" : "+v"(tt2));
                        tw.template init<+1>(twg, tt2);
                        asm volatile("" ::: "memory");
                        setup_done = true;
                    }
#pragma unroll
                    for (int ks = 0; ks < KH; ks++)
                        if (4 * (k0 + ks) < Zb) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(am[ks], bm[ks], acc, 0, 0, 0);
                }
                if (tile * 16 < K2) {
#pragma unroll
                    for (int r = 0; r < 4; r++) S[(kk4 + 4 * r) * SST + tile * 16 + n16] = acc[r];
                }
            }
            lds_barrier();
#pragma unroll
            for (int q = 0; q < NK; q++) {
                double2 a1 = make_double2(0.0, 0.0), b1 = a1;
                if (active && inq[q]) {
                    a1 = *reinterpret_cast<const double2 *>(S + za * SST + 2 * kcq[q]);
                    if (hasb) b1 = *reinterpret_cast<const double2 *>(S + zb * SST + 2 * kcq[q]);
                }
                if (kq[q] == 0) { a1.y = 0.0; b1.y = 0.0; }
                aq[q] = a1; bq[q] = b1;
            }
            lds_barrier();      // every lane holds its coefficients before the slab's set is staged
        } else
        {
            constexpr int R = NODE ? 1 : 4;                // radial rows combined per coefficient
            const double *pf = phi + ((int64_t)d * nrings + ring) * 4;
            const AT *a0 = Az + (int64_t)j0 * azrow + (((int64_t)v * nsz + sz) * nz + (z0 + (active ? za : 0))) * K2;
            const AT *b0 = a0 + ((active && hasb) ? K2 : 0);
            // two wavenumbers at a time: every load of the pair first (no branches: lanes beyond the truncation read a valid
            // address and drop the value) ...
#pragma unroll
            for (int q0 = 0; q0 < NK; q0 += 2) {
                double2 x1[R], y1[R], x2[R], y2[R];
#pragma unroll
                for (int r = 0; r < R; r++) {
                    x1[r] = ldpair(a0 + r * azrow + 2 * kcq[q0]);
                    y1[r] = ldpair(b0 + r * azrow + 2 * kcq[q0]);
                    x2[r] = ldpair(a0 + r * azrow + 2 * kcq[q0 + 1]);
                    y2[r] = ldpair(b0 + r * azrow + 2 * kcq[q0 + 1]);
                }
                if (!setup_done) {                             // ... then, once, the twiddles and phase factors behind them
                    asm volatile("" ::: "memory");
                    int tt = t;
                    asm volatile("" : "+v"(tt));               // opaque copy: the table addresses are formed HERE (hoisted out of the
                                                               // group loop they were spilled, and every reload drained the memory counter)
                    if (REG) {
                        // twiddles of the three register passes: kept in LDS behind the copy-out sets (3 KB, the same for every wave - each
                        // wave writes the whole table and reads back its own writes, no workgroup barrier), not in 12 registers: at the 128
                        // registers of a two-workgroups-per-CU kernel they cost a spilled address, and every reload of it drained the
                        // memory counter - i.e. waited for the previous slot's copy-out stores
                        double2 *wl = smf + SETS * FNP * (L + SKEW);
                        constexpr int S2 = L / 256;                     // the 256-point passes' angles in a table of L entries
                        wl[tt] = twg[S2 * tt]; wl[64 + tt] = twg[S2 * 4 * (tt & 15)]; wl[128 + tt] = twg[S2 * 16 * (tt & 3)];
                        if (LOGL == 9) {                                // the radix-2 pass in front: e^{+2 pi i (t + 64 q) / 512}
#pragma unroll
                            for (int q = 0; q < 4; q++) wl[192 + 64 * q + tt] = twg[tt + 64 * q];
                        }
                    } else tw.template init<+1>(twg, tt);
                    if (!NODE) {
#pragma unroll
                        for (int q = 0; q < NK; q++) if (inq[q]) phq[q] = phr[tt + q * T];
                    }
                    asm volatile("" ::: "memory");
                    setup_done = true;
                }
                double2 a1 = make_double2(0.0, 0.0), b1 = a1, a2 = a1, b2 = a1;
#pragma unroll
                for (int r = 0; r < R; r++) {
                    const double fr = pf[r];
                    if (inq[q0]) { a1.x += fr * x1[r].x; a1.y += fr * x1[r].y; b1.x += fr * y1[r].x; b1.y += fr * y1[r].y; }
                    if (inq[q0 + 1]) { a2.x += fr * x2[r].x; a2.y += fr * x2[r].y; b2.x += fr * y2[r].x; b2.y += fr * y2[r].y; }
                }
                if (kq[q0] == 0) { a1.y = 0.0; b1.y = 0.0; }      // block 0 is the real k = 0 coefficient, block 1 is padding
                if (!hasb) { b1 = make_double2(0.0, 0.0); b2 = b1; }
                if (!NODE) { a1 = cmul(a1, phq[q0]); b1 = cmul(b1, phq[q0]); a2 = cmul(a2, phq[q0 + 1]); b2 = cmul(b2, phq[q0 + 1]); }
                aq[q0] = a1; bq[q0] = b1; aq[q0 + 1] = a2; bq[q0 + 1] = b2;
            }
        }
        if (nslot == 0) FFT_STAMP(1);
        for (int ld = 0; ld < 3; ld++) {
            if (!(ld == 0 ? n0 : ld == 1 ? n1 : n2)) continue;
            const int slot = ld == 0 ? sl0 : ld == 1 ? sl1 : sl2;
            double2 *set = smf + ((COPYOUT && SETS == 2) ? par * FNP * (L + SKEW) : 0);
            double2 *X = set + f * (L + SKEW);
            par ^= 1;
            if (COPYOUT && SETS == 1 && nslot > 0) lds_barrier();     // the previous slot's copy-out has read this (only) set
            if constexpr (REG) {
                // packed spectrum W = Za + i Zb of this lane's wavenumbers k = t + 64 q (q < NK) and of their mirrors L - k
                double2 wk[NK], mir[NK];
#pragma unroll
                for (int q = 0; q < NK; q++) {
                    const int k = kq[q];
                    double2 c = aq[q], e = bq[q];
                    if (ld == 1) { c = make_double2(-k * aq[q].y, k * aq[q].x); e = make_double2(-k * bq[q].y, k * bq[q].x); }
                    else if (ld == 2) { const double qq = -(double)k * k; c.x *= qq; c.y *= qq; e.x *= qq; e.y *= qq; }
                    wk[q] = make_double2(c.x - e.y, c.y + e.x);
                    mir[q] = k > 0 ? make_double2(c.x + e.y, e.x - c.y) : make_double2(0.0, 0.0);      // k = 0: the Nyquist bin, zero
                }
                // the upper half of the input, x[t + 64 q] for q >= NK, are mirrors W[L - k] that lane 64 - t holds: its wavenumber
                // q' feeds q = 2 NK - 1 - q'; lane 0 keeps its own: x[L / 2] = 0 (Nyquist), x[64 (2 NK - q')] = W[L - 64 q']
                const int src = (64 - t) & 63;
                double2 r[NK];
#pragma unroll
                for (int q = 0; q < NK; q++) r[q] = fft_bpermute(mir[q], src);
                const double2 *wl = smf + SETS * FNP * (L + SKEW);
                const double2 wreg[3] = {wl[t], wl[64 + t], wl[128 + t]};
                const int lb = (t >> 4) + 4 * ((t >> 2) & 3) + 16 * (t & 3);
                if constexpr (LOGL == 8) {
                    double2 v[4] = {wk[0], wk[1], t == 0 ? r[0] : r[1], t == 0 ? r[1] : r[0]};
                    fft_reg256_inverse(v, wreg, t);
                    if (nslot == 0) FFT_STAMP(2); else if (nslot == 1) FFT_STAMP(5);
                    if (active) {
#pragma unroll
                        for (int q = 0; q < 4; q++) { const int l = lb + 64 * q; X[l + (l >> 4)] = v[q]; }
                    }
                } else {
                    // 512 points: one radix-2 pass in registers (x[m] +- x[m + 256], the difference turned by e^{+2 pi i m / 512}),
                    // then the even and the odd output points are two independent 256-point transforms
                    const double2 hi[4] = {t == 0 ? r[0] : r[3], t == 0 ? r[3] : r[2], t == 0 ? r[2] : r[1], t == 0 ? r[1] : r[0]};
                    double2 ev[4], od[4];
#pragma unroll
                    for (int q = 0; q < 4; q++) {
                        ev[q] = cadd(wk[q], hi[q]);
                        od[q] = cmul(csub(wk[q], hi[q]), wl[192 + 64 * q + t]);
                    }
                    fft_reg256_inverse(ev, wreg, t);
                    fft_reg256_inverse(od, wreg, t);
                    if (nslot == 0) FFT_STAMP(2); else if (nslot == 1) FFT_STAMP(5);
                    if (active) {
#pragma unroll
                        for (int q = 0; q < 4; q++) {
                            const int l = 2 * (lb + 64 * q);
                            X[l + (l >> 4)] = ev[q];
                            X[l + 1 + ((l + 1) >> 4)] = od[q];
                        }
                    }
                }
            } else
            if (active) {
#pragma unroll
                for (int q = 0; q < NK; q++) {
                    const int k = kq[q];
                    double2 c = aq[q], e = bq[q];
                    if (ld == 1) {                                     // multiply by ik
                        c = make_double2(-k * aq[q].y, k * aq[q].x); e = make_double2(-k * bq[q].y, k * bq[q].x);
                    } else if (ld == 2) {                              // multiply by -k^2
                        const double qq = -(double)k * k;
                        c.x *= qq; c.y *= qq; e.x *= qq; e.y *= qq;
                    }
                    // W = Za + i Zb at bin k, conj(Za) + i conj(Zb) at bin L - k; Nyquist bin is zero
                    X[k] = make_double2(c.x - e.y, c.y + e.x);
                    if (k > 0) X[L - k] = make_double2(c.x + e.y, e.x - c.y);
                    else X[L / 2] = make_double2(0.0, 0.0);
                }
            }
            if (!REG) wave_sync();
            if (COPYOUT) {
                // last pass goes back to LDS, then the whole workgroup writes full 128-byte lines
                if constexpr (!REG) {
                    fft_inplace<LOGL, +1, true, NoSink, HL>(X, tw, t, active);
                    if (nslot == 0) FFT_STAMP(2); else if (nslot == 1) FFT_STAMP(5);
                }
                lds_barrier();       // also orders the previous slot's copy-out reads (other set) before that set is restaged
                if (nslot == 0) FFT_STAMP(3);
                // thread -> (level pair zp, ring point l0 + (2 * 512 / FZC) i): a transform's LDS element l IS the pair of levels
                // (2 zp, 2 zp + 1) of ring point l, so it leaves as one 16-byte store (8 bytes for fp32-stored slots);
                // consecutive lanes cover the FZC levels of one point = one 128-byte line.  16 B per lane matters: at 8 B
                // per lane the vector-memory pipe of a CU moves ~7 B/clk, about half of what the kernel needs.
                const int zp = threadIdx.x & (FNP - 1);
                const double2 *src2 = set + zp * (L + SKEW);
                auto copy_out = [&](auto *out) {
                    using OT = decltype(+out[0]);
                    typedef OT ov2 __attribute__((ext_vector_type(2)));
                    if (pair_ok && 2 * zp + 1 < zc) {
#pragma unroll 4
                        for (int l = threadIdx.x >> (LOGZ - 1); l < L; l += (int)(blockDim.x >> (LOGZ - 1))) {
                            const double2 y = src2[REG ? l + (l >> 4) : l];
                            ov2 o2;
                            o2.x = (OT)y.x; o2.y = (OT)y.y;
                            __builtin_nontemporal_store(o2, reinterpret_cast<ov2 *>(out + (int64_t)l * nz + 2 * zp));
                        }
                    } else if (2 * zp < zc) {      // odd zDim (pairs not 16-byte aligned) or the last level of an odd chunk
                        const bool two = 2 * zp + 1 < zc;
                        for (int l = threadIdx.x >> (LOGZ - 1); l < L; l += (int)(blockDim.x >> (LOGZ - 1))) {
                            const double2 y = src2[REG ? l + (l >> 4) : l];
                            __builtin_nontemporal_store((OT)y.x, out + (int64_t)l * nz + 2 * zp);
                            if (two) __builtin_nontemporal_store((OT)y.y, out + (int64_t)l * nz + 2 * zp + 1);
                        }
                    }
                };
                if (slot == 0) copy_out(phys.val + (int64_t)v * N + p0 * nz + z0);
                else copy_out(phys.der + ((int64_t)(slot - 1) * V + v) * N + p0 * nz + z0);
                if (nslot == 0) FFT_STAMP(4); else if (nslot == 1) FFT_STAMP(6);
                nslot++;
            } else {
                const int64_t o0 = p0 * nz + z0 + za;
                auto sink = [&](int l, double2 y) {
                    if (slot == 0) {
                        double *o = phys.val + (int64_t)v * N + o0 + (int64_t)l * nz;
                        if (pair_ok && hasb) *reinterpret_cast<double2 *>(o) = y;
                        else { o[0] = y.x; if (hasb) o[1] = y.y; }
                    } else {
                        ST *o = phys.der + ((int64_t)(slot - 1) * V + v) * N + o0 + (int64_t)l * nz;
                        o[0] = (ST)y.x; if (hasb) o[1] = (ST)y.y;
                    }
                };
                fft_inplace<LOGL, +1, false, decltype(sink), HL>(X, tw, t, active, sink);
                wave_sync();      // the region is rewritten by the next slot
            }
        }
    }
    FFT_STAMP(7);
}

// ------------------------------------------------------------------------------------------------ forward
template <int LOGL, class FT = double>
__global__ void __launch_bounds__(512)
k_fl_forward_fft(const double *__restrict__ np1, FT *__restrict__ Fl, const int *__restrict__ kmaxr,
                 const int64_t *__restrict__ pstart, const double2 *__restrict__ twg, const int64_t *__restrict__ phoff,
                 const double2 *__restrict__ ph, int V, int nz, int K2, int64_t N) {
    constexpr int L = 1 << LOGL, T = FftCfg<LOGL>::LPT, PPT = L / T;       // lanes per transform; staged pairs per thread
    constexpr int FZC = FftCfg<LOGL>::FZC, FNP = FftCfg<LOGL>::FNP, LOGZ = FftCfg<LOGL>::LOGZ, SKEW = FftCfg<LOGL>::SKEW;
    extern __shared__ double2 smf[];
    const int ring = blockIdx.z, v = blockIdx.y, z0 = blockIdx.x * FZC;
    const int zc = min(FZC, nz - z0);
    const int km = kmaxr[ring];
    const int tid = threadIdx.x;
    const int f = tid / T, t = tid - f * T;
    const int za = 2 * f, zb = 2 * f + 1;
    const bool active = (f < FNP) && (za < zc);
    Twiddles<LOGL> tw;
    tw.template init<-1>(twg, t);
    const int64_t p0 = pstart[ring];
    const double *x = np1 + (int64_t)v * N + p0 * nz + z0;
    // stage the [ring point][FZC levels] tile: the (2 zp, 2 zp + 1) level pair of a point is one 16-byte load and one
    // LDS element of transform zp
    if ((nz & 1) == 0 && (int)blockDim.x == FNP * T) {
        // the workgroup has FNP * T threads, the tile FNP * L pairs: exactly PPT = L / T per thread (4; 8 at L = 512).  All of
        // them are issued before the first LDS write (rolled, every iteration was a load -> wait -> write round trip to HBM)
        typedef double dv2 __attribute__((ext_vector_type(2)));
        const int zp = tid & (FNP - 1);
        double2 val[PPT];
        if (zc == FZC) {                                   // full chunk (wave-uniform): straight-line loads
            dv2 t2[PPT];
#pragma unroll
            for (int i = 0; i < PPT; i++)
                t2[i] = __builtin_nontemporal_load(reinterpret_cast<const dv2 *>(x + (int64_t)((tid + i * FNP * T) >> (LOGZ - 1)) * nz + 2 * zp));
#pragma unroll
            for (int i = 0; i < PPT; i++) val[i] = make_double2(t2[i].x, t2[i].y);
        } else {
#pragma unroll
            for (int i = 0; i < PPT; i++) {
                const int l = (tid + i * FNP * T) >> (LOGZ - 1);
                val[i] = make_double2(0.0, 0.0);
                if (2 * zp + 1 < zc) {
                    const dv2 t2 = __builtin_nontemporal_load(reinterpret_cast<const dv2 *>(x + (int64_t)l * nz + 2 * zp));
                    val[i] = make_double2(t2.x, t2.y);
                } else if (2 * zp < zc) {
                    val[i].x = __builtin_nontemporal_load(x + (int64_t)l * nz + 2 * zp);
                }
            }
        }
#pragma unroll
        for (int i = 0; i < PPT; i++) smf[zp * (L + SKEW) + ((tid + i * FNP * T) >> (LOGZ - 1))] = val[i];
    } else {
        double *ba = (double *)smf;
        for (int o = tid; o < L * FZC; o += blockDim.x) {
            const int zz = o & (FZC - 1), l = o >> LOGZ;
            const double val = (zz < zc) ? __builtin_nontemporal_load(x + (int64_t)l * nz + zz) : 0.0;
            ba[2 * ((zz >> 1) * (L + SKEW) + l) + (zz & 1)] = val;
        }
    }
    __syncthreads();
    double2 *X = smf + (f < FNP ? f : 0) * (L + SKEW);
    // forward transform; the last pass leaves the spectrum in LDS (the untangling needs bins k and L - k)
    fft_inplace<LOGL, -1, true>(X, tw, t, active);
    if (!active) return;
    const double2 *phr = ph + phoff[ring];
    const double inv = 1.0 / L;
    FT *oa = Fl + (((int64_t)ring * V + v) * nz + z0 + za) * K2;
    FT *ob = oa + K2;
    const bool hasb = zb < zc;
    for (int k = t; k <= km; k += T) {
        const double2 wk = X[k], wn = X[(L - k) & (L - 1)];
        // Xa = (W_k + conj W_{-k}) / 2,  Xb = (W_k - conj W_{-k}) / (2i)
        double2 xa = make_double2(0.5 * (wk.x + wn.x), 0.5 * (wk.y - wn.y));
        double2 xb = make_double2(0.5 * (wk.y + wn.y), -0.5 * (wk.x - wn.x));
        if (k == 0) {
            stpair(oa, make_double2(xa.x * inv, 0.0));
            if (hasb) stpair(ob, make_double2(xb.x * inv, 0.0));
        } else {
            const double2 w = cconj(phr[k]);                        // e^{-ik off}
            xa = cmul(xa, w);
            xb = cmul(xb, w);
            stpair(oa + 2 * k, make_double2(xa.x * inv, xa.y * inv));
            if (hasb) stpair(ob + 2 * k, make_double2(xb.x * inv, xb.y * inv));
        }
    }
}

// ------------------------------------------------------------------------------------------------ launchers
static int ilog2(int n) { int l = 0; while ((1 << l) < n) l++; return l; }

bool fft_path_ok(const sx_handle *h) {
    const int L = h->uniform_L;
    return h->has_l && L >= 16 && L <= 512 && (L & (L - 1)) == 0;   // one transform = min(L/4, 64) lanes of one wave
}

// SX_FUSE_ZINV=1: node-space units take the vertical inverse inside the inverse FFT kernel (FUSE): 16-level chunks, b_zDim <= 64
// (16 K steps of the f64 MFMA), rings of at most 256 points (two column tiles per wave), fp64 intermediates.  OFF by default -
// measured at the bench grid (round 3, A/B/A/B on one box): k_zinv 0.065 -> 0.024 ms (it then only serves the ring-wise
// rings), Az of the 174 nodes never written or read, but k_node_fft 0.131 -> 0.201 ms: 968 against 987 steps/s.  The
// matrix-core prologue (per coefficient group four rounds of 16 L2 loads -> 8 MFMAs at the 128 registers the kernel has, an
// LDS round trip and two workgroup barriers) is serial time in a kernel whose limit is its serial time.
bool fft_fused_zinv(const sx_handle *h) {
    return h->fuse_zinv && h->node_mode && h->has_z && h->uniform_L <= 256 && h->Zb <= 64 && h->nz % 16 == 0 && !h->sp32;
}

static int fft_fzc(int) { return 16; }
static size_t fft_lds(int L, int sets = 1, bool reg = false) { return sizeof(double2) * ((size_t)sets * (fft_fzc(L) / 2) * (L + (reg ? L / 16 + 2 : L <= 256 ? 2 : 0)) + (reg ? (L == 512 ? 448 : 192) : 0)); }
static int fft_threads(int L, int hl = 0) { return std::max(64, (fft_fzc(L) / 2) * (std::min(L / 4, 64) >> hl)); }       // FNP transforms x LPT lanes

struct InvTarget {          // where an inverse ring launch writes and which unit tables it uses
    double *out;            // physical [slot][v][N] or node-space G [slot][v][NG]
    const double *phi;      // [3][n_phi][4]
    const int *kmax;
    const int64_t *pstart, *phoff;
    int n_units, n_phi;     // units launched (rings or nodes), ring count of the phi table
    int64_t N;              // plane size of `out`
    int node_mode;
    int unit0 = 0;          // first unit of the launch (the node-space launch skips the nodes only the ring-wise inner rings read)
};

template <int LOGL>
static void launch_inv(sx_handle *h, const int *d_mask, const InvTarget &tg, const double *az, int64_t azrow) {
    const int L = 1 << LOGL;
    dim3 g((h->nz + fft_fzc(L) - 1) / fft_fzc(L), tg.n_units, h->V);
#define INV_LAUNCH_V(NODE, ST, AT, HL, SETS)                                                                                         \
    do {                                                                                                                             \
        if (fft_lds(L, SETS) > 65536)                                                                                                \
            HIPCHK2(hipFuncSetAttribute(reinterpret_cast<const void *>(k_rl_inverse_fft<LOGL, 1, NODE, ST, AT, HL, SETS>),           \
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)fft_lds(L, SETS)));                        \
        hipLaunchKernelGGL((k_rl_inverse_fft<LOGL, 1, NODE, ST, AT, HL, SETS>), g, dim3(fft_threads(L, HL)), fft_lds(L, SETS), h->stream, \
                           reinterpret_cast<const AT *>(az), planes_of<ST>(tg.out, h->V, tg.N), tg.phi, tg.kmax, tg.pstart,          \
                           h->d_tw, tg.phoff, h->d_ph, d_mask, h->V, h->nz, h->nsz, h->K2, tg.n_phi, tg.N, azrow, h->slot[0],        \
                           h->slot[1], h->slot[2], h->slot[3], h->slot[4], h->slot[5], h->slot[6], nullptr, (int64_t)0, nullptr, 0,  \
                           tg.unit0);                                                                                                \
    } while (0)
#define INV_LAUNCH_REG(NODE, ST, AT)                                                                                                 \
    do {                                                                                                                             \
        constexpr int RS = 2;                                                                                                        \
        if (fft_lds(L, RS, true) > 65536)                                                                                            \
            HIPCHK2(hipFuncSetAttribute(reinterpret_cast<const void *>(k_rl_inverse_fft<LOGL, 1, NODE, ST, AT, 0, RS, false, true>), \
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)fft_lds(L, RS, true)));                    \
        hipLaunchKernelGGL((k_rl_inverse_fft<LOGL, 1, NODE, ST, AT, 0, RS, false, true>), g, dim3(fft_threads(L, 0)), fft_lds(L, RS, true), h->stream, \
                           reinterpret_cast<const AT *>(az), planes_of<ST>(tg.out, h->V, tg.N), tg.phi, tg.kmax, tg.pstart,          \
                           h->d_tw, tg.phoff, h->d_ph, d_mask, h->V, h->nz, h->nsz, h->K2, tg.n_phi, tg.N, azrow, h->slot[0],        \
                           h->slot[1], h->slot[2], h->slot[3], h->slot[4], h->slot[5], h->slot[6], nullptr, (int64_t)0, nullptr, 0,  \
                           tg.unit0);                                                                                                \
    } while (0)
#define INV_LAUNCH(NODE, ST, AT)                                                                                                     \
    do {                                                                                                                             \
        if constexpr (LOGL == 8 || LOGL == 9) { if (h->fft_reg) { INV_LAUNCH_REG(NODE, ST, AT); break; } }                                        \
        INV_LAUNCH_V(NODE, ST, AT, 0, 2);                                                                                            \
    } while (0)
#define INV_LAUNCH_FUSED(ST)                                                                                                         \
    hipLaunchKernelGGL((k_rl_inverse_fft<LOGL, 1, true, ST, double, 0, 2, true>), g, dim3(fft_threads(L)), fft_lds(L, 2), h->stream,  \
                       az, planes_of<ST>(tg.out, h->V, tg.N), tg.phi, tg.kmax, tg.pstart, h->d_tw, tg.phoff, h->d_ph, d_mask, h->V,   \
                       h->nz, h->nsz, h->K2, tg.n_phi, tg.N, azrow, h->slot[0], h->slot[1], h->slot[2], h->slot[3], h->slot[4],       \
                       h->slot[5], h->slot[6], h->d_A + (int64_t)h->cell0 * h->C, h->C, h->d_MzT, h->Zb, tg.unit0)
    if (tg.node_mode && fft_fused_zinv(h)) { if (LOGL <= 8) { if (h->f32) INV_LAUNCH_FUSED(float); else INV_LAUNCH_FUSED(double); } }
    else if (h->sp32) { if (tg.node_mode) INV_LAUNCH(true, float, float); else INV_LAUNCH(false, float, float); }     // storage_f32 = 2
    else if (h->f32) { if (tg.node_mode) INV_LAUNCH(true, float, double); else INV_LAUNCH(false, float, double); }
    else { if (tg.node_mode) INV_LAUNCH(true, double, double); else INV_LAUNCH(false, double, double); }
#undef INV_LAUNCH_FUSED
#undef INV_LAUNCH
#undef INV_LAUNCH_REG
#undef INV_LAUNCH_V
}

static void launch_inv_any(sx_handle *h, const int *d_mask, const InvTarget &tg) {
    if (tg.n_units <= 0) return;
    const double *az = h->has_z ? h->d_Az : h->d_A + (int64_t)h->cell0 * h->C;
    const int64_t azrow = h->has_z ? (int64_t)h->V * 3 * h->nz * h->K2 : h->C;
    switch (ilog2(h->uniform_L)) {
        case 4: launch_inv<4>(h, d_mask, tg, az, azrow); break;
        case 5: launch_inv<5>(h, d_mask, tg, az, azrow); break;
        case 6: launch_inv<6>(h, d_mask, tg, az, azrow); break;
        case 7: launch_inv<7>(h, d_mask, tg, az, azrow); break;
        case 8: launch_inv<8>(h, d_mask, tg, az, azrow); break;
        default: launch_inv<9>(h, d_mask, tg, az, azrow); break;
    }
    HIPCHK2(hipGetLastError());
}

template <int LOGL>
static void launch_fwd(sx_handle *h, dim3 g) {
    const int L = 1 << LOGL;
    if (h->sp32)
        hipLaunchKernelGGL((k_fl_forward_fft<LOGL, float>), g, dim3(fft_threads(L)), fft_lds(L), h->stream, h->d_np1 + (int64_t)h->v_lo * h->N,
                           reinterpret_cast<float *>(h->d_Fl) + (int64_t)h->v_lo * h->nz * h->K2,
                           h->d_kmax, h->d_pstart, h->d_tw, h->d_phoff, h->d_ph, h->V, h->nz, h->K2, h->N);
    else       // the variable window [v_lo, v_lo + v_cnt) (sx_internal.hpp): blockIdx.y counts from v_lo through the base pointers
        hipLaunchKernelGGL((k_fl_forward_fft<LOGL, double>), g, dim3(fft_threads(L)), fft_lds(L), h->stream, h->d_np1 + (int64_t)h->v_lo * h->N,
                           h->d_Fl + (int64_t)h->v_lo * h->nz * h->K2, h->d_kmax,
                           h->d_pstart, h->d_tw, h->d_phoff, h->d_ph, h->V, h->nz, h->K2, h->N);
}

// ring-wise inverse of the first n_rings rings of the tile (all of them by default)
void launch_rl_inverse_fft(sx_handle *h, const int *d_mask, int n_rings) {
    const int id = timer_id(h, "k_rl_inverse");
    timer_begin(h, id);
    InvTarget tg{h->d_phys, h->d_phi, h->d_kmax, h->d_pstart, h->d_phoff, n_rings < 0 ? h->nrings : n_rings, h->nrings, h->N, 0};
    launch_inv_any(h, d_mask, tg);
    timer_end(h);
}

// node-space inverse ("radial last", uniform rings): one transform set per radial node instead of per ring
#ifdef SX_PHASES
static long long *g_fft_buf = nullptr;
static int64_t g_fft_n = 0;
void fft_phases_dump() {
    const char *path = getenv("SX_FFT_PHASES_OUT");
    if (!path || !g_fft_buf) return;
    std::vector<long long> hst((size_t)g_fft_n * 8);
    hipDeviceSynchronize();
    hipMemcpy(hst.data(), g_fft_buf, sizeof(long long) * hst.size(), hipMemcpyDeviceToHost);
    FILE *f = fopen(path, "wb");
    if (f) { fwrite(hst.data(), sizeof(long long), hst.size(), f); fclose(f); }
}
#endif

void launch_node_fft(sx_handle *h) {
    const int id = timer_id(h, "k_node_fft");
    timer_begin(h, id);
#ifdef SX_PHASES
    if (!g_fft_buf) {
        g_fft_n = (int64_t)((h->nz + fft_fzc(h->uniform_L) - 1) / fft_fzc(h->uniform_L)) * h->V * h->nbt;
        hipMalloc(&g_fft_buf, sizeof(long long) * g_fft_n * 8);
        hipMemset(g_fft_buf, 0, sizeof(long long) * g_fft_n * 8);
        hipMemcpyToSymbol(HIP_SYMBOL(g_fft_dbg), &g_fft_buf, sizeof(g_fft_buf));
    }
#endif
    // cell c of the node-space rings combines nodes c .. c + 3 and the first such cell is R_in / 3: the nodes below it feed the
    // ring-wise inner rings only (through Az), their node-space transforms would never be read (42 of 174 at the bench grid)
    const int j0 = h->R_in / MUBAR;
    InvTarget tg{h->d_G, h->d_nphi, h->d_nkmax, h->d_npstart, h->d_nphoff, h->nbt - j0, h->nbt, h->NG, 1, j0};
    launch_inv_any(h, h->d_mask_node, tg);
    timer_end(h);
}

void launch_fl_forward_fft(sx_handle *h) {
    const int id = timer_id(h, "k_fl_forward");
    timer_begin(h, id);
    const int fzc = fft_fzc(h->uniform_L);
    dim3 g((h->nz + fzc - 1) / fzc, h->v_cnt, h->nrings);
    switch (ilog2(h->uniform_L)) {
        case 4: launch_fwd<4>(h, g); break;
        case 5: launch_fwd<5>(h, g); break;
        case 6: launch_fwd<6>(h, g); break;
        case 7: launch_fwd<7>(h, g); break;
        case 8: launch_fwd<8>(h, g); break;
        default: launch_fwd<9>(h, g); break;
    }
    HIPCHK2(hipGetLastError());
    timer_end(h);
}

}  // namespace sx
