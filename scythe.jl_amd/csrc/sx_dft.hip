// Azimuthal transforms for ring lengths that are not powers of two - Springsteel's native ragged rings
// (ring ri has 4 + 4 ri points and keeps wavenumbers 0..ri, i.e. only about L/4 of them) - as dense truncated DFTs on the
// f64 matrix cores.  The spectrum is short (2 kmax + 1 ~ L/2 real coefficients), so the DFT of one ring is a genuine
// [L x (2 kmax + 1)] x [(2 kmax + 1) x columns] product with columns = vertical levels x derivative planes; an FFT of
// these lengths (4 * anything, including primes) would need Bluestein with three padded transforms per line.
// v_mfma_f64_16x16x4_f64 runs at the vector fp64 rate on MI355X, what it buys here is that one generated twiddle tile
// serves 16 levels (and up to three derivative planes) instead of one fused multiply-add.
//
// The ring's mirror symmetry halves the work: with theta = 2 pi / L, cos(k theta (L - l)) = cos(k theta l) and
// sin(k theta (L - l)) = -sin(k theta l), so only the points l = 0 .. L/2 are transformed, in a cosine and a sine part,
// and ONE twiddle (cos, sin)(k theta l) fetched per lane feeds both parts:
//   inverse  P[l] = sum_k Cc[k] cos(k theta l),  Q[l] = sum_k Cs[k] sin(k theta l)      x[l] = P - Q,  x[L - l] = P + Q
//            (Cc, Cs) = radial evaluation of Az (4 nodes), phase reference, factor 2 (same staging as k_rl_inverse);
//            d/dlambda: (Cc, Cs) -> (-k Cs, k Cc);  d2/dlambda2: -k^2 (Cc, Cs), formed while loading the B operand
//   forward  sr[k] = sum_{l <= L/2} (x[l] + x[L-l]) cos(k theta l),  si[k] = -sum (x[l] - x[L-l]) sin(k theta l)
//            (the end points l = 0, L/2 enter once), then the phase reference and 1 / L
//
// Rings with fewer than 8 levels (RL grids) stay on the scalar kernels of sx_kernels.hip: the MFMA N dimension is the
// vertical level.
#include "sx_internal.hpp"
#include <algorithm>
#include <array>
#include <cstdio>
#include <cstdlib>
#include <vector>

namespace sx {

#define HIPCHK3(x)                                                                                  \
    do {                                                                                            \
        hipError_t e_ = (x);                                                                        \
        if (e_ != hipSuccess) set_error(std::string(#x) + ": " + hipGetErrorString(e_) + " (" __FILE__ ":" + std::to_string(__LINE__) + ")"); \
    } while (0)

typedef double dft_d4 __attribute__((ext_vector_type(4)));

// phase stamps of the diagnostic build (-DSX_PHASES): per workgroup of k_rl_inverse_dft, as seen by wave 0:
// [0] total cycles, [1] staging of the coefficient sets (loads .. barrier), [2] matrix-core loops, [3] result stores,
// [4] row tiles done, [5] ring length, [6] matrix-core instructions issued, [7] total in 100 MHz real-time ticks
#ifdef SX_PHASES
__device__ long long *g_dft_dbg = nullptr;
#define DFT_NOW() ((long long)__builtin_readcyclecounter())
#else
#define DFT_NOW() 0ll
#endif

constexpr int DFT_LMAX = 5120;   // longest ring the LDS-resident twiddle table allows (native patches up to 426 cells, kmax 1279)
constexpr int DZC = 16;       // levels per workgroup = MFMA N
constexpr int CST = 17;       // row stride (doubles) of the LDS tiles: 4 consecutive rows land in different banks

// (m + step) mod L for 0 <= m, step < L as add, subtract, unsigned minimum (the wrapped candidate is huge when no wrap is due): one
// instruction fewer than compare-and-subtract in loops whose vector instructions are matrix-pipe time
__device__ __forceinline__ int wrap_add(int m, int step, int L) {
    const unsigned t = (unsigned)(m + step);
    return (int)min(t, t - (unsigned)L);
}

// ------------------------------------------------------------------------------------------------ inverse
// One row-tile pair of the half ring: rows l0 / l1 (clamped), angle index m = (k l) mod L advancing by 4 l per K step.
struct RowPair {
    int m0, m1, s0, s1;
    __device__ __forceinline__ void init(int l0, int l1, int kk, int L) {
        m0 = (int)(((int64_t)kk * l0) % L); m1 = (int)(((int64_t)kk * l1) % L);
        s0 = (int)(((int64_t)4 * l0) % L); s1 = (int)(((int64_t)4 * l1) % L);
    }
    __device__ __forceinline__ void step(int L) {
        m0 = wrap_add(m0, s0, L);
        m1 = wrap_add(m1, s1, L);
    }
};

template <class ST>
__global__ void __launch_bounds__(512)
k_rl_inverse_dft(const double *__restrict__ Az, Planes<ST> phys, const double *__restrict__ phi, const int *__restrict__ Lr,
                 const int *__restrict__ kmaxr, const int64_t *__restrict__ pstart, const int64_t *__restrict__ twoff,
                 const double2 *__restrict__ tw, const int64_t *__restrict__ phoff, const double2 *__restrict__ ph, int V,
                 int nz, int nsz, int K2, int nrings, int64_t N, int64_t azrow, int s_u, int s_r, int s_rr, int s_l, int s_ll,
                 int s_z, int s_zz, const int *__restrict__ slotmask, const int *__restrict__ items, int lcap, int kcap4) {
    extern __shared__ double sm[];
    // (ring, variable) from the work list: most expensive first - workgroups are dispatched in blockIdx order, a ring's work
    // grows with its square and a variable's with its planes, so the cheap items fill the tail of the launch
    const int ring = items[2 * blockIdx.y], v = items[2 * blockIdx.y + 1], z0 = blockIdx.x * DZC;
    const int mask = slotmask[v];
    const int zc = min(DZC, nz - z0);
    const int L = Lr[ring], km = kmaxr[ring], Lh = L / 2;
    const int K4 = (km + 1 + 3) & ~3;                           // wavenumber rows 0..km, padded to the MFMA K step
    double2 *twl = reinterpret_cast<double2 *>(sm);             // [L]   (cos, sin)(2 pi m / L)
    double *Cc = sm + 2 * (size_t)lcap;                         // [K4][CST] cosine coefficients
    double *Cs = Cc + (size_t)kcap4 * CST;                      // [K4][CST] sine coefficients (row 0 = 0)
    const int j0 = ring / MUBAR;
    const double2 *phr = ph + phoff[ring];
    const int64_t p0 = pstart[ring];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, nw = blockDim.x >> 6;
    [[maybe_unused]] const long long dbg_t0 = DFT_NOW();
    [[maybe_unused]] long long dbg_stage = 0, dbg_mm = 0, dbg_st = 0, dbg_tiles = 0, dbg_nmfma = 0;
#ifdef SX_PHASES
    const long long dbg_r0 = (long long)__builtin_amdgcn_s_memrealtime();
#endif
    for (int m = tid; m < L; m += blockDim.x) twl[m] = tw[twoff[ring] + m];

    // coefficient sets q: (sz, d) = (0,0) (0,1) (0,2) (1,0) (2,0); only those with a requested slot
    int qs[5], nq = 0;
    for (int q = 0; q < 5; q++) {
        const int sz = q < 3 ? 0 : q - 2;
        if (sz >= nsz) break;
        const int sl = (q == 0) ? s_u : (q == 1) ? s_r : (q == 2) ? s_rr : (q == 3) ? s_z : s_zz;
        if (((mask >> sl) & 1) || (q == 0 && (((mask >> s_l) | (mask >> s_ll)) & 1))) qs[nq++] = q;
    }
    // Staging of a set: thread -> (level zz, wavenumber kq + 32 b): the 4 radial rows of a wavenumber come as 16-byte (Re, Im)
    // pairs, KB wavenumbers per thread and batch (the rolled form waited for each wavenumber's 8 scalar loads on its own).
    // The FIRST batch of the next set (wavenumbers 0 .. 127: 64 registers) is requested before the matrix-core loops of
    // the current set and consumed after them: the rows come over the fabric (Az does not fit the L2s; every set of every
    // workgroup pulls up to 256 KB), which a quarter of a large ring's workgroup time used to wait for.
    constexpr int KB = 4;
    const int zz = tid >> 5, kq = tid & 31;                      // 512 threads = 16 levels x 32 wavenumbers
    const bool zin = zz < zc;
    auto issue = [&](int q, int k0, double2 (&raw)[KB][4]) {
        const int sz = q < 3 ? 0 : q - 2;
        const double *a = Az + (int64_t)j0 * azrow + (((int64_t)v * nsz + sz) * nz + (z0 + (zin ? zz : 0))) * K2;
#pragma unroll
        for (int b = 0; b < KB; b++) {
            const int kc = min(k0 + kq + 32 * b, km);            // valid address for the padding rows (value dropped)
#pragma unroll
            for (int r = 0; r < 4; r++) raw[b][r] = *reinterpret_cast<const double2 *>(a + r * azrow + 2 * kc);
        }
    };
    auto consume = [&](int q, int k0, const double2 (&raw)[KB][4]) {
        const int d = q < 3 ? q : 0;
        const double *pf = phi + ((int64_t)d * nrings + ring) * 4;
        const double f0 = pf[0], f1 = pf[1], f2 = pf[2], f3 = pf[3];
#pragma unroll
        for (int b = 0; b < KB; b++) {
            const int k = k0 + kq + 32 * b;
            if (k >= K4 || zz >= DZC) continue;
            double cr = 0.0, ci = 0.0;
            if (zin && k <= km) {
                cr = f0 * raw[b][0].x + f1 * raw[b][1].x + f2 * raw[b][2].x + f3 * raw[b][3].x;
                if (k > 0) {
                    ci = f0 * raw[b][0].y + f1 * raw[b][1].y + f2 * raw[b][2].y + f3 * raw[b][3].y;
                    const double2 w = phr[k];               // e^{+i k off}
                    const double tr = cr * w.x - ci * w.y;
                    ci = 2.0 * (cr * w.y + ci * w.x);
                    cr = 2.0 * tr;
                }
            }
            Cc[k * CST + zz] = cr;
            Cs[k * CST + zz] = ci;
        }
    };
    double2 raw0[KB][4];
    if (nq > 0) issue(qs[0], 0, raw0);

    for (int qi = 0; qi < nq; qi++) {
        const int q = qs[qi];
        const int slot0 = (q == 0) ? s_u : (q == 1) ? s_r : (q == 2) ? s_rr : (q == 3) ? s_z : s_zz;
        const bool need0 = (mask >> slot0) & 1;
        const bool needl = (q == 0) && ((mask >> s_l) & 1), needll = (q == 0) && ((mask >> s_ll) & 1);
        [[maybe_unused]] const long long dbg_t1 = DFT_NOW();
        __syncthreads();                                        // the previous set has been consumed (and twl is complete)
        consume(q, 0, raw0);
        for (int k0 = 32 * KB; k0 < K4; k0 += 32 * KB) {
            double2 raw[KB][4];
            issue(q, k0, raw);
            consume(q, k0, raw);
        }
        __syncthreads();
        if (qi + 1 < nq) {
            issue(qs[qi + 1], 0, raw0);
            asm volatile("" ::: "memory");                      // requested HERE, ahead of the loops
        }
        dbg_stage += DFT_NOW() - dbg_t1;

        const int i = lane & 15, kk = lane >> 4;
        // Quarter-wave form: with L a multiple of 4, cos(k theta (L/2 - l)) = (-1)^k cos(k theta l) and
        // sin(k theta (L/2 - l)) = -(-1)^k sin(k theta l), so only the points l = 0 .. L/4 are transformed, the even and the
        // odd wavenumbers in separate accumulators (Pe, Po, Qe, Qo), and each row gives FOUR ring points:
        //   x[l]       = (Pe + Po) - (Qe + Qo)        x[L - l]   = (Pe + Po) + (Qe + Qo)
        //   x[L/2 - l] = (Pe - Po) + (Qe - Qo)        x[L/2 + l] = (Pe - Po) - (Qe - Qo)
        // - half the matrix-core work of the half-ring form (this transform is compute-bound: DESIGN.md 7).
        const int Lq = L / 4;
        for (int mt = wave; mt * 16 <= Lq; mt += nw) {
            const int lrow = min(mt * 16 + i, Lq);                 // A operand: this lane's ring point
            [[maybe_unused]] const long long dbg_t2 = DFT_NOW();
            dbg_tiles++;
            dbg_nmfma += (long long)(((km / 2 + 1 + 3) >> 2) + (km >= 1 ? (((km - 1) / 2 + 1 + 3) >> 2) : 0)) * 2 * ((need0 ? 1 : 0) + (needl ? 1 : 0) + (needll ? 1 : 0));
            dft_d4 z4 = {0.0, 0.0, 0.0, 0.0};
            dft_d4 pu[2] = {z4, z4}, qu[2] = {z4, z4}, pl[2] = {z4, z4}, ql[2] = {z4, z4}, pll[2] = {z4, z4}, qll[2] = {z4, z4};
#pragma unroll
            for (int par = 0; par < 2; par++) {
                if (km < par) continue;
                const int nk = (km - par) / 2 + 1;                  // wavenumbers of this parity
                // this lane's wavenumber k = 8 js + 2 kk + par; angle index (k l) mod L advances by 8 l per step
                int m = (int)(((int64_t)(2 * kk + par) * lrow) % L);
                const int sm8 = (int)(((int64_t)8 * lrow) % L);
                double kd = (double)(2 * kk + par);
                for (int js = 0; js * 4 < nk; js++) {
                    const int k = 8 * js + 2 * kk + par;
                    const bool kin = k < K4;
                    const double2 t0 = twl[m];
                    const double bc = kin ? Cc[k * CST + i] : 0.0, bs = kin ? Cs[k * CST + i] : 0.0;
                    if (need0) {
                        pu[par] = __builtin_amdgcn_mfma_f64_16x16x4f64(t0.x, bc, pu[par], 0, 0, 0);
                        qu[par] = __builtin_amdgcn_mfma_f64_16x16x4f64(t0.y, bs, qu[par], 0, 0, 0);
                    }
                    if (needl) {                                    // i k (cr + i ci): cosine part -k ci, sine part k cr
                        pl[par] = __builtin_amdgcn_mfma_f64_16x16x4f64(t0.x, -kd * bs, pl[par], 0, 0, 0);
                        ql[par] = __builtin_amdgcn_mfma_f64_16x16x4f64(t0.y, kd * bc, ql[par], 0, 0, 0);
                    }
                    if (needll) {
                        const double k2 = -(kd * kd);
                        pll[par] = __builtin_amdgcn_mfma_f64_16x16x4f64(t0.x, k2 * bc, pll[par], 0, 0, 0);
                        qll[par] = __builtin_amdgcn_mfma_f64_16x16x4f64(t0.y, k2 * bs, qll[par], 0, 0, 0);
                    }
                    kd += 8.0;
                    m = wrap_add(m, sm8, L);
                }
            }
            // D tile: lane holds column n = lane & 15 (level), rows (lane >> 4) + 4 r (points l of the quarter ring)
#ifdef SX_PHASES
            asm volatile("s_nop 0" : "+v"(pu[0]), "+v"(qu[0]), "+v"(pu[1]), "+v"(qu[1]));      // the accumulators are final here
#endif
            [[maybe_unused]] const long long dbg_t3 = DFT_NOW();
            dbg_mm += dbg_t3 - dbg_t2;
            if (i < zc) {
                auto put = [&](int slot, int64_t pt, double val) {
                    if (slot == 0) phys.val[(int64_t)v * N + pt] = val;
                    else phys.der[((int64_t)(slot - 1) * V + v) * N + pt] = (ST)val;
                };
                auto put4 = [&](int slot, int lo, double Pe, double Po, double Qe, double Qo) {
                    const double Ps = Pe + Po, Pd = Pe - Po, Qs = Qe + Qo, Qd = Qe - Qo;
                    put(slot, (p0 + lo) * nz + z0 + i, Ps - Qs);
                    if (lo > 0) put(slot, (p0 + (L - lo)) * nz + z0 + i, Ps + Qs);
                    if (lo < Lq) {
                        put(slot, (p0 + (Lh - lo)) * nz + z0 + i, Pd + Qd);
                        if (lo > 0) put(slot, (p0 + (Lh + lo)) * nz + z0 + i, Pd - Qd);
                    }
                };
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    const int lo = mt * 16 + kk + 4 * r;
                    if (lo > Lq) continue;
                    if (need0) put4(slot0, lo, pu[0][r], pu[1][r], qu[0][r], qu[1][r]);
                    if (needl) put4(s_l, lo, pl[0][r], pl[1][r], ql[0][r], ql[1][r]);
                    if (needll) put4(s_ll, lo, pll[0][r], pll[1][r], qll[0][r], qll[1][r]);
                }
            }
            dbg_st += DFT_NOW() - dbg_t3;
        }
    }
#ifdef SX_PHASES
    if (tid == 0 && g_dft_dbg) {
        long long *o = g_dft_dbg + (((int64_t)ring * V + v) * gridDim.x + blockIdx.x) * 8;
        o[0] = DFT_NOW() - dbg_t0; o[1] = dbg_stage; o[2] = dbg_mm; o[3] = dbg_st; o[4] = dbg_tiles; o[5] = L; o[6] = dbg_nmfma;
        o[7] = (long long)__builtin_amdgcn_s_memrealtime() - dbg_r0;
    }
#endif
}


// ------------------------------------------------------------------------------------------------ inverse, merged passes (round 4)
// The kernel above stages ONE coefficient set at a time - value (serving u, d/dlambda, d2/dlambda2: 6 MFMAs per fetched twiddle),
// then d/dr, d2/dr2, d/dz each alone (2 MFMAs per twiddle) - with a barrier round per set, and spreads a ring's row tiles over its 8
// waves by whole tiles: 17 tiles take 3 rounds for 2.1 rounds of work.  Here
//   * a PASS holds up to two sets in LDS (when two fit beside the twiddle table: kmax <= ~270) and up to four planes in the
//     accumulators: h, ug, vg (u, d/dlambda, d/dr) in ONE pass of 6 MFMAs per twiddle; ub, vb (u, d/dlambda, d2/dlambda2, d/dz | d/dr,
//     d2/dr2) in two passes of 8 and 4 - half the staging rounds, no 2-MFMA pass left;
//   * the plane set of a unit of work is a TEMPLATE parameter (MA: which of plain / i k / -k^2 of set A, HASB: set B's plane), so the
//     inner loop carries no plane bookkeeping (a generic two-set loop was tried in round 2 and lost to exactly that);
//   * the tiles of the last, partial round are split by PLANES over the idle waves (different planes of one row tile need no
//     reduction): 17 tiles = 2 rounds + a quarter round instead of 3.
// SX_DFT_MERGE=0 (read at sx_create) keeps the kernel above.
struct DftUnitOut { int slot[4]; };      // output slots of the unit's planes: A plain, A ik, A -k^2, B (-1: not in this unit)

template <int MA, bool HASB, class ST>
__device__ __forceinline__ void dft_unit(const double2 *__restrict__ twl, const double *__restrict__ CcA, const double *__restrict__ CsA,
                                         const double *__restrict__ CcB, const double *__restrict__ CsB, Planes<ST> phys, int V, int v, int64_t N,
                                         int64_t p0, int nz, int z0, int zc, int L, int km, int K4, int mt, int lane, const DftUnitOut &o) {
    const int i = lane & 15, kk = lane >> 4;
    const int Lh = L / 2, Lq = L / 4;
    const int lrow = min(mt * 16 + i, Lq);
    constexpr int NP = ((MA & 1) ? 1 : 0) + ((MA & 2) ? 1 : 0) + ((MA & 4) ? 1 : 0) + (HASB ? 1 : 0);
    // accumulator index of each plane (compile-time: a running counter inside the loop put the arrays into scratch memory)
    constexpr int I1 = 0, I2 = (MA & 1) ? 1 : 0, I4 = I2 + ((MA & 2) ? 1 : 0), IB = I4 + ((MA & 4) ? 1 : 0);
    dft_d4 P[NP][2], Q[NP][2];
#pragma unroll
    for (int p = 0; p < NP; p++)
#pragma unroll
        for (int par = 0; par < 2; par++) { P[p][par] = dft_d4{0.0, 0.0, 0.0, 0.0}; Q[p][par] = P[p][par]; }
#pragma unroll
    for (int par = 0; par < 2; par++) {
        if (km < par) continue;
        const int nk = (km - par) / 2 + 1;                  // wavenumbers of this parity
        int m = (int)(((int64_t)(2 * kk + par) * lrow) % L);
        const int sm8 = (int)(((int64_t)8 * lrow) % L);
        double kd = (double)(2 * kk + par);
        for (int js = 0; js * 4 < nk; js++) {
            const int k = 8 * js + 2 * kk + par;                     // < K4 (the caller's Kz): rows km + 1 .. Kz - 1 hold zeros
            const double2 t0 = twl[m];
            if (MA != 0) {
                const double bc = CcA[k * CST + i], bs = CsA[k * CST + i];
                if (MA & 1) {
                    P[I1][par] = __builtin_amdgcn_mfma_f64_16x16x4f64(t0.x, bc, P[I1][par], 0, 0, 0);
                    Q[I1][par] = __builtin_amdgcn_mfma_f64_16x16x4f64(t0.y, bs, Q[I1][par], 0, 0, 0);
                }
                if (MA & 2) {                               // i k (cr + i ci): cosine part -k ci, sine part k cr
                    P[I2][par] = __builtin_amdgcn_mfma_f64_16x16x4f64(t0.x, -kd * bs, P[I2][par], 0, 0, 0);
                    Q[I2][par] = __builtin_amdgcn_mfma_f64_16x16x4f64(t0.y, kd * bc, Q[I2][par], 0, 0, 0);
                }
                if (MA & 4) {
                    const double k2 = -(kd * kd);
                    P[I4][par] = __builtin_amdgcn_mfma_f64_16x16x4f64(t0.x, k2 * bc, P[I4][par], 0, 0, 0);
                    Q[I4][par] = __builtin_amdgcn_mfma_f64_16x16x4f64(t0.y, k2 * bs, Q[I4][par], 0, 0, 0);
                }
            }
            if (HASB) {
                const double bc = CcB[k * CST + i], bs = CsB[k * CST + i];
                P[IB][par] = __builtin_amdgcn_mfma_f64_16x16x4f64(t0.x, bc, P[IB][par], 0, 0, 0);
                Q[IB][par] = __builtin_amdgcn_mfma_f64_16x16x4f64(t0.y, bs, Q[IB][par], 0, 0, 0);
            }
            kd += 8.0;
            m = wrap_add(m, sm8, L);
        }
    }
    if (i >= zc) return;
    auto put = [&](int slot, int64_t pt, double val) {
        if (slot == 0) phys.val[(int64_t)v * N + pt] = val;
        else phys.der[((int64_t)(slot - 1) * V + v) * N + pt] = (ST)val;
    };
    int slots[NP];
    if (MA & 1) slots[I1] = o.slot[0];
    if (MA & 2) slots[I2] = o.slot[1];
    if (MA & 4) slots[I4] = o.slot[2];
    if (HASB) slots[IB] = o.slot[3];
#pragma unroll
    for (int r = 0; r < 4; r++) {
        const int lo = mt * 16 + kk + 4 * r;
        if (lo > Lq) continue;
#pragma unroll
        for (int p = 0; p < NP; p++) {
            const double Ps = P[p][0][r] + P[p][1][r], Pd = P[p][0][r] - P[p][1][r], Qs = Q[p][0][r] + Q[p][1][r], Qd = Q[p][0][r] - Q[p][1][r];
            put(slots[p], (p0 + lo) * nz + z0 + i, Ps - Qs);
            if (lo > 0) put(slots[p], (p0 + (L - lo)) * nz + z0 + i, Ps + Qs);
            if (lo < Lq) {
                put(slots[p], (p0 + (Lh - lo)) * nz + z0 + i, Pd + Qd);
                if (lo > 0) put(slots[p], (p0 + (Lh + lo)) * nz + z0 + i, Pd - Qd);
            }
        }
    }
}

// Eighth-wave form of a unit (round 4).  With M = L / 4 the EVEN wavenumbers of the quarter ring are a series of period 2 M sampled on
// half a period, so they fold once more about M / 2:  cos(k theta (M - l)) = +cos(k theta l) for k = 0 mod 4, -cos for k = 2 mod 4, and
// sin(k theta (M - l)) = -sin / +sin.  (The odd wavenumbers do not fold: their cosine becomes a sine.)  A unit therefore owns a row tile
// l = 16 mt .. of the EIGHTH ring and the mirrored rows M - l:
//   even k: classes k = 0, 2 mod 4 at rows l only        E0c, E0s, E2c, E2s    (K / 4 wavenumbers each)
//   odd k : rows l and rows M - l                        Oc, Os, Oc', Os'      (K / 2 wavenumbers each)
//   rows l    :  Pe = E0c + E2c   Qe = E0s + E2s   Po = Oc    Qo = Os
//   rows M - l:  Pe = E0c - E2c   Qe = E2s - E0s   Po = Oc'   Qo = Os'          and the four ring points of a row as in the quarter-wave form
// - 3/4 of the quarter-wave form's matrix-core work (the even half of it halves), for eight accumulator tiles per plane instead of
// four: a unit carries at most two planes (NPM of the kernel).
template <int MA, bool HASB, class ST, bool HT>      // HT: the twiddle table holds half the ring, tw[m + L/2] = -tw[m]
__device__ __forceinline__ void dft_unit8(const double2 *__restrict__ twl, const double *__restrict__ CcA, const double *__restrict__ CsA,
                                          const double *__restrict__ CcB, const double *__restrict__ CsB, Planes<ST> phys, int V, int v, int64_t N,
                                          int64_t p0, int nz, int z0, int zc, int L, int km, int mt, int lane, const DftUnitOut &o,
                                          [[maybe_unused]] long long &dbg_loops) {
    const int i = lane & 15, kk = lane >> 4;
    const int Lh = L / 2, M = L / 4, Mh = M / 2;
    [[maybe_unused]] const long long dbg_l0 = DFT_NOW();
    const int lrow = min(mt * 16 + i, Mh);
    constexpr int NP = ((MA & 1) ? 1 : 0) + ((MA & 2) ? 1 : 0) + ((MA & 4) ? 1 : 0) + (HASB ? 1 : 0);
    constexpr int I1 = 0, I2 = (MA & 1) ? 1 : 0, I4 = I2 + ((MA & 2) ? 1 : 0), IB = I4 + ((MA & 4) ? 1 : 0);
    dft_d4 Ec[NP][2], Es[NP][2], Oc[NP][2], Os[NP][2];       // E*[plane][class 0 / 2], O*[plane][rows l / rows M - l]
#pragma unroll
    for (int p = 0; p < NP; p++)
#pragma unroll
        for (int c = 0; c < 2; c++) { Ec[p][c] = dft_d4{0.0, 0.0, 0.0, 0.0}; Es[p][c] = Ec[p][c]; Oc[p][c] = Ec[p][c]; Os[p][c] = Ec[p][c]; }
#define DFT8_MM(acc, a, b) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0)
    // Both loops are software-pipelined by hand, two K steps per iteration with alternating operand sets: the operands of step
    // js + 1 (two twiddles, the coefficient pairs) are requested from the LDS BEFORE the matrix-core instructions of step js are
    // issued and waited for BEHIND them (scheduling barriers and an opaque use pin both: the compiler's own loop reads, waits, then
    // issues; it rotated a plain one-step prefetch back into that form and sank the requests of a conditional second step into the
    // branch).  The read-wait-issue form ran at 0.905 or 1.47 ms depending on where ONE 4-byte instruction in front of the loops put
    // them in memory (profiles/r04/eighth_wave_alignment.txt); this form is 0.905-0.912 ms at eight different offsets.  An odd step
    // count runs one step on zero rows (the staged rows end at a multiple of 32); the last request goes one step beyond them (inside
    // the LDS allocation: the launcher adds 32 rows; the values are not used).
    // Twiddle fetch: index (k l) mod L advanced by add, subtract, unsigned minimum.  HT: the table holds -tw[m] for m < L / 2; with
    // u = m - L/2 the entry is min(m, u) (unsigned) and the value's sign the sign bit of u - an XOR mask applied to the pair once it
    // has arrived (in the matrix-core block, not at the request).  Every vector instruction here is time the matrix pipe stands still
    // (an MFMA holds its SIMD's vector issue port), see DESIGN.md 7.
    auto flip = [](double x, int mask) __attribute__((always_inline)) { return __hiloint2double(__double2hiint(x) ^ mask, __double2loint(x)); };
    auto advance = [&](int &m, int step) __attribute__((always_inline)) {
        const unsigned t = (unsigned)(m + step);
        m = (int)min(t, t - (unsigned)L);
    };
    auto tw_req = [&](int m, int &mask) __attribute__((always_inline)) {
        if constexpr (!HT) { mask = 0; return twl[m]; }
        else {
            const int u = m - Lh;
            mask = u & (int)0x80000000;
            return twl[min((unsigned)m, (unsigned)u)];
        }
    };
    struct OddOps { double2 t; int sm; double ac, as, bc, bs; };
    struct EvenOps { double2 t0, t2; int sm0, sm2; double ac0, as0, ac2, as2, bc0, bs0, bc2, bs2; };
    if (km >= 1) {                                            // odd wavenumbers k = 8 js + 2 kk + 1, rows l and M - l
        // The mirrored rows need no fetch of their own: k theta (M - l) = k pi / 2 - k theta l, so for odd k
        // (cos, sin)(k theta (M - l)) = s (sin, cos)(k theta l) with s = +1 for k = 1 mod 4 (kk even), -1 for k = 3 mod 4 (kk odd).
        const int nst = ((km - 1) / 2 + 1 + 3) / 4;
        int m = ((2 * kk + 1) * lrow) % L;
        const int s8 = (8 * lrow) % L;
        const int sgm = (kk & 1) ? (int)0x80000000 : 0;
        double kd = (double)(2 * kk + 1);
        int ko = (2 * kk + 1) * CST + i;                       // rows km + 1 .. Kz - 1 of the coefficient tiles hold zeros
        auto load = [&](OddOps &x) __attribute__((always_inline)) {
            x.t = tw_req(m, x.sm);
            if (MA != 0) { x.ac = CcA[ko]; x.as = CsA[ko]; }
            if (HASB) { x.bc = CcB[ko]; x.bs = CsB[ko]; }
            advance(m, s8);
            ko += 8 * CST;
        };
        auto hold = [&](OddOps &x) __attribute__((always_inline)) {      // the values exist from here on (an opaque use: the loads cannot move past it)
            asm volatile("" : "+v"(x.t.x), "+v"(x.t.y));
            if (MA != 0) asm volatile("" : "+v"(x.ac), "+v"(x.as));
            if (HASB) asm volatile("" : "+v"(x.bc), "+v"(x.bs));
        };
        auto mma = [&](const OddOps &x) __attribute__((always_inline)) {
            const double tx = HT ? flip(x.t.x, x.sm) : x.t.x, ty = HT ? flip(x.t.y, x.sm) : x.t.y;
            const double tmx = flip(x.t.y, x.sm ^ sgm), tmy = flip(x.t.x, x.sm ^ sgm);      // rows M - l
            if (MA & 1) {
                DFT8_MM(Oc[I1][0], tx, x.ac); DFT8_MM(Os[I1][0], ty, x.as); DFT8_MM(Oc[I1][1], tmx, x.ac); DFT8_MM(Os[I1][1], tmy, x.as);
            }
            if (MA & 2) {                                      // i k (cr + i ci): cosine part -k ci, sine part k cr
                const double b1 = -kd * x.as, b2 = kd * x.ac;
                DFT8_MM(Oc[I2][0], tx, b1); DFT8_MM(Os[I2][0], ty, b2); DFT8_MM(Oc[I2][1], tmx, b1); DFT8_MM(Os[I2][1], tmy, b2);
            }
            if (MA & 4) {
                const double k2 = -(kd * kd), b1 = k2 * x.ac, b2 = k2 * x.as;
                DFT8_MM(Oc[I4][0], tx, b1); DFT8_MM(Os[I4][0], ty, b2); DFT8_MM(Oc[I4][1], tmx, b1); DFT8_MM(Os[I4][1], tmy, b2);
            }
            if (HASB) {
                DFT8_MM(Oc[IB][0], tx, x.bc); DFT8_MM(Os[IB][0], ty, x.bs); DFT8_MM(Oc[IB][1], tmx, x.bc); DFT8_MM(Os[IB][1], tmy, x.bs);
            }
            kd += 8.0;
        };
        OddOps A{}, B{};
        load(A);
        hold(A);
        for (int js = 0; js < nst; js += 2) {
            load(B);
            __builtin_amdgcn_sched_barrier(0);                 // the requests stay in front of the matrix-core instructions ...
            mma(A);
            __builtin_amdgcn_sched_barrier(0);
            hold(B);                                           // ... and are waited for behind them
            load(A);
            __builtin_amdgcn_sched_barrier(0);
            mma(B);                                            // (an odd count's extra step multiplies zero rows)
            __builtin_amdgcn_sched_barrier(0);
            hold(A);
        }
    }
    {                                                         // even wavenumbers: k = 16 js + 4 kk (class 0) and + 2 (class 2), rows l
        const int nst = (km / 4 + 1 + 3) / 4;
        int m0 = (4 * kk * lrow) % L, m2 = ((4 * kk + 2) * lrow) % L;
        const int s16 = (16 * lrow) % L;
        double kd0 = (double)(4 * kk), kd2 = kd0 + 2.0;
        int ko = 4 * kk * CST + i;
        auto load = [&](EvenOps &x) __attribute__((always_inline)) {
            x.t0 = tw_req(m0, x.sm0); x.t2 = tw_req(m2, x.sm2);
            if (MA != 0) { x.ac0 = CcA[ko]; x.as0 = CsA[ko]; x.ac2 = CcA[ko + 2 * CST]; x.as2 = CsA[ko + 2 * CST]; }
            if (HASB) { x.bc0 = CcB[ko]; x.bs0 = CsB[ko]; x.bc2 = CcB[ko + 2 * CST]; x.bs2 = CsB[ko + 2 * CST]; }
            advance(m0, s16);
            advance(m2, s16);
            ko += 16 * CST;
        };
        auto hold = [&](EvenOps &x) __attribute__((always_inline)) {
            asm volatile("" : "+v"(x.t0.x), "+v"(x.t0.y), "+v"(x.t2.x), "+v"(x.t2.y));
            if (MA != 0) asm volatile("" : "+v"(x.ac0), "+v"(x.as0), "+v"(x.ac2), "+v"(x.as2));
            if (HASB) asm volatile("" : "+v"(x.bc0), "+v"(x.bs0), "+v"(x.bc2), "+v"(x.bs2));
        };
        auto mma = [&](const EvenOps &x) __attribute__((always_inline)) {
            const double t0x = HT ? flip(x.t0.x, x.sm0) : x.t0.x, t0y = HT ? flip(x.t0.y, x.sm0) : x.t0.y;
            const double t2x = HT ? flip(x.t2.x, x.sm2) : x.t2.x, t2y = HT ? flip(x.t2.y, x.sm2) : x.t2.y;
            if (MA & 1) {
                DFT8_MM(Ec[I1][0], t0x, x.ac0); DFT8_MM(Es[I1][0], t0y, x.as0); DFT8_MM(Ec[I1][1], t2x, x.ac2); DFT8_MM(Es[I1][1], t2y, x.as2);
            }
            if (MA & 2) {
                DFT8_MM(Ec[I2][0], t0x, -kd0 * x.as0); DFT8_MM(Es[I2][0], t0y, kd0 * x.ac0);
                DFT8_MM(Ec[I2][1], t2x, -kd2 * x.as2); DFT8_MM(Es[I2][1], t2y, kd2 * x.ac2);
            }
            if (MA & 4) {
                const double q0 = -(kd0 * kd0), q2 = -(kd2 * kd2);
                DFT8_MM(Ec[I4][0], t0x, q0 * x.ac0); DFT8_MM(Es[I4][0], t0y, q0 * x.as0);
                DFT8_MM(Ec[I4][1], t2x, q2 * x.ac2); DFT8_MM(Es[I4][1], t2y, q2 * x.as2);
            }
            if (HASB) {
                DFT8_MM(Ec[IB][0], t0x, x.bc0); DFT8_MM(Es[IB][0], t0y, x.bs0); DFT8_MM(Ec[IB][1], t2x, x.bc2); DFT8_MM(Es[IB][1], t2y, x.bs2);
            }
            kd0 += 16.0; kd2 += 16.0;
        };
        EvenOps A{}, B{};
        load(A);
        hold(A);
        for (int js = 0; js < nst; js += 2) {
            load(B);
            __builtin_amdgcn_sched_barrier(0);
            mma(A);
            __builtin_amdgcn_sched_barrier(0);
            hold(B);
            load(A);
            __builtin_amdgcn_sched_barrier(0);
            mma(B);
            __builtin_amdgcn_sched_barrier(0);
            hold(A);
        }
    }
#undef DFT8_MM
#ifdef SX_PHASES
    asm volatile("s_nop 0" : "+v"(Ec[0][0]), "+v"(Es[0][1]), "+v"(Oc[0][0]), "+v"(Os[0][1]));      // the accumulators are final here
#endif
    dbg_loops += DFT_NOW() - dbg_l0;
    if (i >= zc) return;
    int slots[NP];
    if (MA & 1) slots[I1] = o.slot[0];
    if (MA & 2) slots[I2] = o.slot[1];
    if (MA & 4) slots[I4] = o.slot[2];
    if (HASB) slots[IB] = o.slot[3];
    // Stores through a per-plane base pointer (this lane's level of the ring's point 0) and 32-bit point offsets: the 64-bit
    // multiply-add per stored value of the first version was vector-ALU time the matrix pipe stood still for.  (The caller runs the
    // plane chunks of a tile in a loop: an opaque copy of the tile index keeps the compiler from hoisting the offsets out of that
    // loop - it spilled them, and a spill reload waits for every store in flight.)
    int mto = mt;
    asm volatile("" : "+v"(mto));
    const int64_t ib = p0 * nz + z0 + i;
    const int nzL = L * nz, nzH = Lh * nz, nzM = M * nz;
#pragma unroll
    for (int p = 0; p < NP; p++) {
        const int slot = slots[p];
        double *bv = phys.val + (int64_t)v * N + ib;
        ST *bd = phys.der + ((int64_t)(slot > 0 ? slot - 1 : 0) * V + v) * N + ib;
        auto put = [&](int off, double val) __attribute__((always_inline)) {
            if (slot == 0) bv[off] = val;
            else bd[off] = (ST)val;
        };
        auto put4 = [&](int lo, int off, double Pe, double Po, double Qe, double Qo) __attribute__((always_inline)) {
            const double Ps = Pe + Po, Pd = Pe - Po, Qs = Qe + Qo, Qd = Qe - Qo;
            put(off, Ps - Qs);
            if (lo > 0) put(nzL - off, Ps + Qs);
            if (lo < M) {
                put(nzH - off, Pd + Qd);
                if (lo > 0) put(nzH + off, Pd - Qd);
            }
        };
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const int lo = mto * 16 + kk + 4 * r;
            if (lo > Mh) continue;
            const int off = lo * nz;
            put4(lo, off, Ec[p][0][r] + Ec[p][1][r], Oc[p][0][r], Es[p][0][r] + Es[p][1][r], Os[p][0][r]);
            if (M - lo != lo) put4(M - lo, nzM - off, Ec[p][0][r] - Ec[p][1][r], Oc[p][1][r], Es[p][1][r] - Es[p][0][r], Os[p][1][r]);
        }
    }
}

// HT (round 4, rings whose ONE coefficient set and HALF twiddle table fit 80 KB of LDS: kmax <= 255): launched with 256 threads, one set per
// pass, the table behind the set - TWO workgroups per CU, each on its own (ring, variable, level chunk): while one stages a set, drains its
// stores or waits at a barrier, the other has the matrix pipes to itself.  (One 512-thread workgroup owns a CU's whole LDS; nothing ran
// under the 45 % of its time that are not matrix-core loops.)
template <class ST, int NPM, bool HT = false>      // NPM = 0: quarter-wave units, any plane set; 2: eighth-wave units of at most NPM planes
__global__ void __launch_bounds__(512)
k_rl_inverse_dft_merged(const double *__restrict__ Az, Planes<ST> phys, const double *__restrict__ phi, const int *__restrict__ Lr,
                        const int *__restrict__ kmaxr, const int64_t *__restrict__ pstart, const int64_t *__restrict__ twoff,
                        const double2 *__restrict__ tw, const int64_t *__restrict__ phoff, const double2 *__restrict__ ph, int V,
                        int nz, int nsz, int K2, int nrings, int64_t N, int64_t azrow, int s_u, int s_r, int s_rr, int s_l, int s_ll,
                        int s_z, int s_zz, const int *__restrict__ slotmask, const int *__restrict__ items, int lcap, int kcap4, int two_sets) {
    constexpr bool E8 = NPM > 0;
    extern __shared__ double sm[];
    const int ring = items[2 * blockIdx.y], v = items[2 * blockIdx.y + 1], z0 = blockIdx.x * DZC;
    const int mask = slotmask[v];
    const int zc = min(DZC, nz - z0);
    const int L = Lr[ring], km = kmaxr[ring];
    // a K step of parity p touches wavenumbers 8 js + 2 kk + p, js < ceil(nk_p / 4): rows 0 .. Kz - 1 are staged (zeros beyond km), so the
    // loops carry neither a clamp nor a select
    // (eighth-wave units: classes k = 0 / 2 mod 4 advance by 16 per K step, two steps per loop iteration: rows up to the next multiple of 32)
    const int K4 = E8 ? 32 * ((km + 1 + 31) / 32) : 8 * ((km / 2 + 1 + 3) / 4);
    // [L] twiddles, then the sets; HT: the ONE set first, then [L / 2] twiddles (the pipelined loops' last request, one step beyond
    // the set's rows, then lands in the table instead of needing rows of its own)
    double2 *twl = reinterpret_cast<double2 *>(HT ? sm + (size_t)2 * kcap4 * CST : sm);
    double *Cset[2][2];                                         // [set A / B][cosine / sine] coefficient tiles [K4][CST]
    Cset[0][0] = HT ? sm : sm + 2 * (size_t)lcap;
    Cset[0][1] = Cset[0][0] + (size_t)kcap4 * CST;
    Cset[1][0] = Cset[0][1] + (size_t)kcap4 * CST;
    Cset[1][1] = Cset[1][0] + (size_t)kcap4 * CST;
    const int j0 = ring / MUBAR;
    const double2 *phr = ph + phoff[ring];
    const int64_t p0 = pstart[ring];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, nw = blockDim.x >> 6;
    for (int m = tid; m < (HT ? L / 2 : L); m += blockDim.x) {
        double2 t = tw[twoff[ring] + m];
        if (HT) { t.x = -t.x; t.y = -t.y; }                     // the half table holds -tw[m] (the fetch's sign mask is then one AND, see dft_unit8)
        twl[m] = t;
    }

    // stored sets q: (sz, d) = (0,0) value [planes u, d/dlambda, d2/dlambda2], (0,1) d/dr, (0,2) d2/dr2, (1,0) d/dz, (2,0) d2/dz2.
    // Everything about the passes lives in scalars (bit fields), not in indexed private arrays (those would sit in scratch memory).
    auto has = [&](int sl) { return sl >= 0 && ((mask >> sl) & 1); };
    const int ma_u = (has(s_u) ? 1 : 0) | (has(s_l) ? 2 : 0) | (has(s_ll) ? 4 : 0);
    // bit q of `need`: set q is needed; singles = sets 1 .. 4
    int need = (ma_u ? 1 : 0) | (has(s_r) ? 2 : 0) | (has(s_rr) ? 4 : 0) | ((nsz > 1 && has(s_z)) ? 8 : 0) | ((nsz > 2 && has(s_zz)) ? 16 : 0);
    auto slot_of = [&](int q) { return q == 1 ? s_r : q == 2 ? s_rr : q == 3 ? s_z : s_zz; };
    // Staging of a set (as in k_rl_inverse_dft): thread -> (level zz, wavenumber kq + 32 b), the 4 radial rows as 16-byte (Re, Im)
    // pairs, KB wavenumbers per thread and batch: 256 wavenumbers of a set - every ring of a <= 85-cell patch whole - are requested in ONE
    // burst (128 registers that are free while no accumulator lives).  (No request-ahead across passes here: its registers are the fourth plane's.)
    constexpr int KB = 4;      // (8 - a whole 256-wavenumber set in one burst - measured slower: 0.984 vs 0.956 ms)
    const int lgq = HT ? 4 : 5, NQ = 1 << lgq;                   // 512 threads = 16 levels x 32 wavenumbers (HT: 256 = 16 x 16)
    const int zz = tid >> lgq, kq = tid & (NQ - 1);
    const bool zin = zz < zc;
    auto stage = [&](int q, double *Cc, double *Cs) __attribute__((always_inline)) {
        const int sz = q < 3 ? 0 : q - 2, d = q < 3 ? q : 0;
        const double *a = Az + (int64_t)j0 * azrow + (((int64_t)v * nsz + sz) * nz + (z0 + (zin ? zz : 0))) * K2;
        const double *pf = phi + ((int64_t)d * nrings + ring) * 4;
        const double f0 = pf[0], f1 = pf[1], f2 = pf[2], f3 = pf[3];
        for (int k0 = 0; k0 < K4; k0 += NQ * KB) {
            double2 raw[KB][4];
#pragma unroll
            for (int b = 0; b < KB; b++) {
                const int kc = min(k0 + kq + NQ * b, km);
#pragma unroll
                for (int r = 0; r < 4; r++) raw[b][r] = *reinterpret_cast<const double2 *>(a + r * azrow + 2 * kc);
            }
#pragma unroll
            for (int b = 0; b < KB; b++) {
                const int k = k0 + kq + NQ * b;
                if (k >= K4 || zz >= DZC) continue;
                double cr = 0.0, ci = 0.0;
                if (zin && k <= km) {
                    cr = f0 * raw[b][0].x + f1 * raw[b][1].x + f2 * raw[b][2].x + f3 * raw[b][3].x;
                    if (k > 0) {
                        ci = f0 * raw[b][0].y + f1 * raw[b][1].y + f2 * raw[b][2].y + f3 * raw[b][3].y;
                        const double2 w = phr[k];               // e^{+i k off}
                        const double tr = cr * w.x - ci * w.y;
                        ci = 2.0 * (cr * w.y + ci * w.x);
                        cr = 2.0 * tr;
                    }
                }
                Cc[k * CST + zz] = cr;
                Cs[k * CST + zz] = ci;
            }
        }
    };
    const int T = E8 ? L / 8 / 16 + 1 : L / 4 / 16 + 1;         // row tiles of the quarter ring (E8: of the eighth ring, each with its mirrored rows)
    const int F = T - T % nw, R = T - F;                        // whole rounds of tiles, tiles of the last (partial) round
    [[maybe_unused]] const long long dbg_t0 = DFT_NOW();
    [[maybe_unused]] long long dbg_stage = 0, dbg_mm = 0, dbg_units = 0, dbg_wait = 0, dbg_loops = 0;
#ifdef SX_PHASES
    const long long dbg_r0 = (long long)__builtin_amdgcn_s_memrealtime();
#endif

    // passes: the lowest remaining set as A and - if two sets fit the LDS - the HIGHEST remaining single-plane set as B
    while (need) {
        const int qa = __ffs(need) - 1;
        need &= ~(1 << qa);
        int qb = -1;
        if (two_sets && (need & ~1)) {
            qb = 31 - __clz(need & ~1);
            need &= ~(1 << qb);
        }
        [[maybe_unused]] const long long dbg_t1 = DFT_NOW();
        __syncthreads();                                        // the previous pass has been consumed (and twl is complete)
        [[maybe_unused]] const long long dbg_t1b = DFT_NOW();
        dbg_wait += dbg_t1b - dbg_t1;
        stage(qa, Cset[0][0], Cset[0][1]);
        if (qb >= 0) stage(qb, Cset[1][0], Cset[1][1]);
        __syncthreads();
        [[maybe_unused]] const long long dbg_t2 = DFT_NOW();
        dbg_stage += dbg_t2 - dbg_t1b;
        const int maf = qa == 0 ? ma_u : 1;
        const bool hb = qb >= 0;
        DftUnitOut of;
        of.slot[0] = qa == 0 ? s_u : slot_of(qa); of.slot[1] = s_l; of.slot[2] = s_ll; of.slot[3] = hb ? slot_of(qb) : -1;
        // one unit = (row tile, plane subset): dispatch to the instance whose plane set is compiled in
        auto run = [&](int mt, int ma, bool b) {
#define DFT_U(MA_, B_)                                                                                                                 \
    do {                                                                                                                               \
        constexpr int np_ = ((MA_) & 1) + (((MA_) >> 1) & 1) + (((MA_) >> 2) & 1) + ((B_) ? 1 : 0);                                    \
        if constexpr (E8 && np_ <= NPM && !(HT && (B_)))                                                                               \
            dft_unit8<MA_, B_, ST, HT>(twl, Cset[0][0], Cset[0][1], Cset[1][0], Cset[1][1], phys, V, v, N, p0, nz, z0, zc, L, km, mt, lane, of, dbg_loops);   \
        else if constexpr (!E8)                                                                                                        \
            dft_unit<MA_, B_, ST>(twl, Cset[0][0], Cset[0][1], Cset[1][0], Cset[1][1], phys, V, v, N, p0, nz, z0, zc, L, km, K4, mt, lane, of);   \
    } while (0)
            switch ((ma << 1) | (b ? 1 : 0)) {
                case 1: DFT_U(0, true); break;
                case 2: DFT_U(1, false); break;
                case 3: DFT_U(1, true); break;
                case 4: DFT_U(2, false); break;
                case 5: DFT_U(2, true); break;
                case 6: DFT_U(3, false); break;
                case 7: DFT_U(3, true); break;
                case 8: DFT_U(4, false); break;
                case 9: DFT_U(4, true); break;
                case 10: DFT_U(5, false); break;
                case 11: DFT_U(5, true); break;
                case 12: DFT_U(6, false); break;
                case 13: DFT_U(6, true); break;
                case 14: DFT_U(7, false); break;
                case 15: DFT_U(7, true); break;
                default: break;
            }
#undef DFT_U
        };
        // This wave's units: one whole tile per full round; the R tiles of the last (partial) round as R x npl (tile, plane) pairs dealt
        // out in runs of ceil(R npl / 8) consecutive pairs (different planes of a row tile need no reduction; a run may span two tiles).
        // A unit runs in chunks of at most CH planes.  ONE loop with ONE call site, so that the dispatcher is inlined once.
        const int nfull = F / nw;
        const int npl = __popc(maf) + (hb ? 1 : 0);
        const int cper = (R * npl + nw - 1) / nw;
        int j = wave * cper;
        const int j1 = min(j + cper, R * npl);
        int u = 0, mt = 0, ma = 0;
        bool b = false;
        constexpr int CH = E8 ? NPM : 4;                        // eighth-wave units hold 8 accumulator tiles per plane
        for (;;) {
            if (!ma && !b) {                                    // next unit of this wave
                if (u < nfull) {
                    mt = wave + u * nw; ma = maf; b = hb;
                    u++;
                } else if (j < j1) {                            // planes in the order A plain, A ik, A -k^2, B
                    const int tile = j / npl, lo = j - tile * npl, hi = min(npl, lo + (j1 - j));
                    int idx = 0;
                    for (int bit = 1; bit <= 4; bit <<= 1)
                        if (maf & bit) { if (idx >= lo && idx < hi) ma |= bit; idx++; }
                    if (hb && idx >= lo && idx < hi) b = true;
                    mt = F + tile;
                    j += hi - lo;
                } else break;
                dbg_units++;
            }
            int cma = 0, cnt = 0;
            bool cb = false;
            for (int bit = 1; bit <= 4; bit <<= 1)
                if ((ma & bit) && cnt < CH) { cma |= bit; cnt++; }
            ma &= ~cma;
            if (b && cnt < CH) { cb = true; b = false; }
            run(mt, cma, cb);
        }
        dbg_mm += DFT_NOW() - dbg_t2;
    }
#ifdef SX_PHASES
    if (tid == 0 && g_dft_dbg) {       // [0] total, [1] staging, [2] units (loops + stores) of wave 0, [3] wait at the pass barrier, [4] units of wave 0, [5] L
        long long *o = g_dft_dbg + (((int64_t)ring * V + v) * gridDim.x + blockIdx.x) * 8;
        o[0] = DFT_NOW() - dbg_t0; o[1] = dbg_stage; o[2] = dbg_mm; o[3] = dbg_wait; o[4] = dbg_units + (dbg_loops << 8); o[5] = L;          // units in the low byte, cycles inside the matrix-core loops of the eighth-wave units above
        o[6] = (long long)__builtin_amdgcn_s_memrealtime() - dbg_r0; o[7] = dbg_r0;      // 100 MHz ticks: duration, start
    }
#endif
}

// The same transform for rings whose coefficient sets do not fit the LDS next to the twiddle table (kmax > 319, i.e. native
// patches of more than 106 cells: a 171-cell patch has rings of 2,052 points with kmax 512).  Loop order inverted: a
// workgroup owns ONE group of 8 row tiles (128 points of the quarter ring, blockIdx.z), its accumulators stay in registers
// while the wavenumbers pass through the LDS in chunks of KCH; every coefficient set is staged once per row-tile group.
// Simpler staging than the kernel above (no request-ahead across sets): these rings are few and large.
constexpr int KCH = 256;      // wavenumbers per staged chunk (a multiple of 8: a K step covers 8 wavenumbers of one parity)

template <class ST>
__global__ void __launch_bounds__(512)
k_rl_inverse_dft_big(const double *__restrict__ Az, Planes<ST> phys, const double *__restrict__ phi, const int *__restrict__ Lr,
                     const int *__restrict__ kmaxr, const int64_t *__restrict__ pstart, const int64_t *__restrict__ twoff,
                     const double2 *__restrict__ tw, const int64_t *__restrict__ phoff, const double2 *__restrict__ ph, int V,
                     int nz, int nsz, int K2, int nrings, int64_t N, int64_t azrow, int s_u, int s_r, int s_rr, int s_l, int s_ll,
                     int s_z, int s_zz, const int *__restrict__ slotmask, const int *__restrict__ items, int lcap) {
    extern __shared__ double sm[];
    const int ring = items[2 * blockIdx.y], v = items[2 * blockIdx.y + 1], z0 = blockIdx.x * DZC;
    const int mask = slotmask[v];
    const int zc = min(DZC, nz - z0);
    const int L = Lr[ring], km = kmaxr[ring], Lh = L / 2, Lq = L / 4;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, nw = blockDim.x >> 6;
    const int mt = blockIdx.z * nw + wave;                       // this wave's row tile of the quarter ring
    if ((int)blockIdx.z * nw * 16 > Lq) return;                  // no row tile group here (uniform for the workgroup)
    const bool rows = mt * 16 <= Lq;
    double2 *twl = reinterpret_cast<double2 *>(sm);             // [L]
    double *Cc = sm + 2 * (size_t)lcap;                         // [KCH][CST]
    double *Cs = Cc + (size_t)KCH * CST;
    const int j0 = ring / MUBAR;
    const double2 *phr = ph + phoff[ring];
    const int64_t p0 = pstart[ring];
    for (int m = tid; m < L; m += blockDim.x) twl[m] = tw[twoff[ring] + m];
    const int i = lane & 15, kk = lane >> 4;
    const int lrow = min(mt * 16 + i, Lq);
    const int sm8 = (int)(((int64_t)8 * lrow) % L);
    for (int q = 0; q < 5; q++) {
        const int sz = q < 3 ? 0 : q - 2, d = q < 3 ? q : 0;
        if (sz >= nsz) break;
        const int slot0 = (q == 0) ? s_u : (q == 1) ? s_r : (q == 2) ? s_rr : (q == 3) ? s_z : s_zz;
        const bool need0 = (mask >> slot0) & 1;
        const bool needl = (q == 0) && ((mask >> s_l) & 1), needll = (q == 0) && ((mask >> s_ll) & 1);
        if (!need0 && !needl && !needll) continue;
        const double *pf = phi + ((int64_t)d * nrings + ring) * 4;
        const double f0 = pf[0], f1 = pf[1], f2 = pf[2], f3 = pf[3];
        dft_d4 z4 = {0.0, 0.0, 0.0, 0.0};
        dft_d4 pu[2] = {z4, z4}, qu[2] = {z4, z4}, pl[2] = {z4, z4}, ql[2] = {z4, z4}, pll[2] = {z4, z4}, qll[2] = {z4, z4};
        for (int kc0 = 0; kc0 <= km; kc0 += KCH) {
            __syncthreads();                                    // the previous chunk has been consumed (and twl is complete)
            // stage wavenumbers kc0 .. kc0 + KCH - 1: thread -> (level, wavenumber), 4 radial rows as 16-byte (Re, Im) pairs
            for (int e = tid; e < KCH * DZC; e += blockDim.x) {
                const int zz = (e >> 5) & (DZC - 1), kl = (e & 31) + 32 * (e >> 9), k = kc0 + kl;     // 32 consecutive wavenumbers of a level per half wave
                double cr = 0.0, ci = 0.0;
                if (zz < zc && k <= km) {
                    const double *a = Az + (int64_t)j0 * azrow + (((int64_t)v * nsz + sz) * nz + (z0 + zz)) * K2 + 2 * k;
                    const double2 r0 = *reinterpret_cast<const double2 *>(a), r1 = *reinterpret_cast<const double2 *>(a + azrow),
                                  r2 = *reinterpret_cast<const double2 *>(a + 2 * azrow), r3 = *reinterpret_cast<const double2 *>(a + 3 * azrow);
                    cr = f0 * r0.x + f1 * r1.x + f2 * r2.x + f3 * r3.x;
                    if (k > 0) {
                        ci = f0 * r0.y + f1 * r1.y + f2 * r2.y + f3 * r3.y;
                        const double2 w = phr[k];               // e^{+i k off}
                        const double tr = cr * w.x - ci * w.y;
                        ci = 2.0 * (cr * w.y + ci * w.x);
                        cr = 2.0 * tr;
                    }
                }
                Cc[kl * CST + zz] = cr;
                Cs[kl * CST + zz] = ci;
            }
            __syncthreads();
            if (!rows) continue;
            const int kend = min(km, kc0 + KCH - 1);
#pragma unroll
            for (int par = 0; par < 2; par++) {
                // this lane's wavenumber k = kc0 + 8 js + 2 kk + par; angle index (k l) mod L advances by 8 l per step
                int k = kc0 + 2 * kk + par;
                int m = (int)(((int64_t)k * lrow) % L);
                double kd = (double)k;
                for (int js = 0; kc0 + 8 * js + par <= kend; js++, k += 8) {
                    const bool kin = k <= kend;
                    const double2 t0 = twl[m];
                    const double bc = kin ? Cc[(k - kc0) * CST + i] : 0.0, bs = kin ? Cs[(k - kc0) * CST + i] : 0.0;
                    if (need0) {
                        pu[par] = __builtin_amdgcn_mfma_f64_16x16x4f64(t0.x, bc, pu[par], 0, 0, 0);
                        qu[par] = __builtin_amdgcn_mfma_f64_16x16x4f64(t0.y, bs, qu[par], 0, 0, 0);
                    }
                    if (needl) {
                        pl[par] = __builtin_amdgcn_mfma_f64_16x16x4f64(t0.x, -kd * bs, pl[par], 0, 0, 0);
                        ql[par] = __builtin_amdgcn_mfma_f64_16x16x4f64(t0.y, kd * bc, ql[par], 0, 0, 0);
                    }
                    if (needll) {
                        const double k2 = -(kd * kd);
                        pll[par] = __builtin_amdgcn_mfma_f64_16x16x4f64(t0.x, k2 * bc, pll[par], 0, 0, 0);
                        qll[par] = __builtin_amdgcn_mfma_f64_16x16x4f64(t0.y, k2 * bs, qll[par], 0, 0, 0);
                    }
                    kd += 8.0;
                    m = wrap_add(m, sm8, L);
                }
            }
        }
        if (!rows || i >= zc) continue;
        auto put = [&](int slot, int64_t pt, double val) {
            if (slot == 0) phys.val[(int64_t)v * N + pt] = val;
            else phys.der[((int64_t)(slot - 1) * V + v) * N + pt] = (ST)val;
        };
        auto put4 = [&](int slot, int lo, double Pe, double Po, double Qe, double Qo) {
            const double Ps = Pe + Po, Pd = Pe - Po, Qs = Qe + Qo, Qd = Qe - Qo;
            put(slot, (p0 + lo) * nz + z0 + i, Ps - Qs);
            if (lo > 0) put(slot, (p0 + (L - lo)) * nz + z0 + i, Ps + Qs);
            if (lo < Lq) {
                put(slot, (p0 + (Lh - lo)) * nz + z0 + i, Pd + Qd);
                if (lo > 0) put(slot, (p0 + (Lh + lo)) * nz + z0 + i, Pd - Qd);
            }
        };
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const int lo = mt * 16 + kk + 4 * r;
            if (lo > Lq) continue;
            if (need0) put4(slot0, lo, pu[0][r], pu[1][r], qu[0][r], qu[1][r]);
            if (needl) put4(s_l, lo, pl[0][r], pl[1][r], ql[0][r], ql[1][r]);
            if (needll) put4(s_ll, lo, pll[0][r], pll[1][r], qll[0][r], qll[1][r]);
        }
    }
}

// ------------------------------------------------------------------------------------------------ inverse, RL grids
// Without a vertical dimension the MFMA columns are the requested (variable, derivative plane) pairs of the ring (19 for the
// shallow-water slab sets): all coefficient sets - value, d/dr, d2/dr2 by radial evaluation, d/dlambda, d2/dlambda2 by
// i k / -k^2 - are formed once while staging, so one fetched twiddle serves every plane.
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
                        int K2, int nrings, int64_t N, int64_t arow, PlaneGroups pgs, int ring0, int lcap, int kch4) {
    extern __shared__ double sm[];
    const PlaneCols &pc = pgs.g[blockIdx.z];
    // one ring has too little work per plane to fill the chip with a workgroup per ring (300 rings at config 2): the ring's
    // row tiles are split over gridDim.y workgroups, each staging the (small) coefficient tiles for itself
    const int ring = ring0 + blockIdx.x;
    const int part = blockIdx.y, nparts = gridDim.y;
    const int L = Lr[ring], km = kmaxr[ring], Lh = L / 2;
    if (part * (int)(blockDim.x >> 6) * 16 > Lh) return;        // nothing for this part (uniform for the workgroup)
    const int K4 = (km + 1 + 3) & ~3;
    double2 *twl = reinterpret_cast<double2 *>(sm);
    double *Cc = sm + 2 * (size_t)lcap;                         // [kch4][CSTP]: wavenumbers pass through in chunks of kch4
    double *Cs = Cc + (size_t)kch4 * CSTP;                      //   (one chunk for kmax <= 319; longer rings keep their accumulators across chunks)
    const int j0 = ring / MUBAR;
    const double2 *phr = ph + phoff[ring];
    const int64_t p0 = pstart[ring];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, nw = blockDim.x >> 6;
    for (int m = tid; m < L; m += blockDim.x) twl[m] = tw[twoff[ring] + m];
    const double *pf = phi + (int64_t)ring * 4;
    const int i = lane & 15, kk = lane >> 4;
    // ONE pair of row tiles per wave (the launcher gives the ring ceil(tiles / 16) workgroups of 8 waves)
    const int mt = part * nw + wave, mtb = mt + nw * nparts;
    const bool one = mt * 16 <= Lh, two = mtb * 16 <= Lh;
    RowPair rp;
    rp.init(min(mt * 16 + i, Lh), min(mtb * 16 + i, Lh), kk, L);
    dft_d4 p0a = {0.0, 0.0, 0.0, 0.0}, q0a = p0a, p1a = p0a, q1a = p0a;
    for (int kc0 = 0; kc0 < K4; kc0 += kch4) {
        const int nkc = min(kch4, K4 - kc0), kn = min(km + 1, kc0 + nkc) - kc0;      // rows of this chunk, of which wavenumbers <= km
        __syncthreads();                                        // the previous chunk has been consumed (and twl is complete)
        for (int e = tid; e < nkc * CSTP; e += blockDim.x) { Cc[e] = 0.0; Cs[e] = 0.0; }
        __syncthreads();
        for (int e = tid; e < V * kn; e += blockDim.x) {
            const int kl = e % kn, k = kc0 + kl, v = e / kn;
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
#pragma unroll
            for (int kind = 0; kind < 5; kind++) {
                const int c = pc.colof[v][kind];
                if (c < 0) continue;
                double xr, xi;
                if (kind < 3) { xr = cr[kind]; xi = ci[kind]; }
                else if (kind == 3) { xr = -(double)k * ci[0]; xi = (double)k * cr[0]; }
                else { xr = -((double)k * k) * cr[0]; xi = -((double)k * k) * ci[0]; }
                Cc[kl * CSTP + c] = xr;
                Cs[kl * CSTP + c] = xi;
            }
        }
        __syncthreads();
        if (!one) continue;
        for (int js = 0; js < nkc / 4; js++) {
            const double2 t0 = twl[rp.m0], t1 = twl[rp.m1];
            const double bc = Cc[(4 * js + kk) * CSTP + i], bs = Cs[(4 * js + kk) * CSTP + i];
            p0a = __builtin_amdgcn_mfma_f64_16x16x4f64(t0.x, bc, p0a, 0, 0, 0);
            q0a = __builtin_amdgcn_mfma_f64_16x16x4f64(t0.y, bs, q0a, 0, 0, 0);
            if (two) {
                p1a = __builtin_amdgcn_mfma_f64_16x16x4f64(t1.x, bc, p1a, 0, 0, 0);
                q1a = __builtin_amdgcn_mfma_f64_16x16x4f64(t1.y, bs, q1a, 0, 0, 0);
            }
            rp.step(L);
        }
    }
    if (one && i < pc.n) {
        const int vv = pc.v[i], sl = pc.slot[i];
        auto put = [&](int64_t pt, double val) {
            if (sl == 0) phys.val[(int64_t)vv * N + pt] = val;
            else phys.der[((int64_t)(sl - 1) * V + vv) * N + pt] = (ST)val;
        };
#pragma unroll
        for (int half = 0; half < 2; half++) {
            if (half && !two) break;
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const int lo = (half ? mtb : mt) * 16 + kk + 4 * r;
                if (lo > Lh) continue;
                const double P = half ? p1a[r] : p0a[r], Q = half ? q1a[r] : q0a[r];
                put(p0 + lo, P - Q);
                if (lo > 0 && lo < Lh) put(p0 + L - lo, P + Q);
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------ forward
// Workgroup = (16 rows, variable, ring); rows = vertical levels, or (planes) the variables of a grid without z.  The half
// ring's points are staged LCH at a time as Xs[l][row] = x[l] + x[L-l] and Xd[l][row] = x[l] - x[L-l]; wave w owns the
// wavenumber tiles nt = w, w + nw, ... (16 wavenumbers each, at most NTW per wave) and keeps their cosine / sine accumulators
// across chunks.  Re and Im of a wavenumber end in the same lane, so the result is stored as 16-byte pairs.
constexpr int LCH = 128;      // half-ring points per staged chunk
constexpr int NTW = 3;        // wavenumber tiles per wave: 8 waves x 3 x 16 = 384 wavenumbers > kmax <= 319

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
    const int L = Lr[ring], km = kmaxr[ring], Lh = L / 2;
    double2 *twl = reinterpret_cast<double2 *>(sm);             // [L]
    double *Xs = sm + 2 * (size_t)lcap;                         // [LCH][CST]
    double *Xd = Xs + (size_t)LCH * CST;
    const double2 *phr = ph + phoff[ring];
    const int64_t p0 = pstart[ring];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, nw = blockDim.x >> 6;
    // planes mode: the ring's wavenumber tiles are split over gridDim.y workgroups (a workgroup per ring would leave most
    // of the chip idle on RL grids); tile of (wave, q) = (part * ntw + q) * nw + wave, part = 0 otherwise
    const int tile0 = planes ? (int)blockIdx.y * ntw * nw : 0;      // ntw <= NTW wavenumber tiles per wave
    if (tile0 * 16 > km) return;
    for (int m = tid; m < L; m += blockDim.x) twl[m] = tw[twoff[ring] + m];
    const int n = lane & 15, kk = lane >> 4;
    dft_d4 ac[NTW], as[NTW];
#pragma unroll
    for (int q = 0; q < NTW; q++) { ac[q] = dft_d4{0.0, 0.0, 0.0, 0.0}; as[q] = ac[q]; }
    const double *x = planes ? np1 + p0 : np1 + (int64_t)v * N + p0 * nz + z0;
    const int64_t sl = planes ? 1 : nz, szz = planes ? N : 1;        // strides of a ring point / of a row in var_np1

    for (int lc = 0; lc <= Lh; lc += LCH) {
        const int ln = min(LCH, Lh + 1 - lc);                   // points lc .. lc + ln - 1 of the half ring
        const int ln4 = (ln + 3) & ~3;
        __syncthreads();
        for (int o = tid; o < ln4 * DZC; o += blockDim.x) {
            const int zz = o & (DZC - 1), l = lc + (o >> 4);
            double a = 0.0, b = 0.0;
            if (zz < zc && l <= Lh) {
                a = x[(int64_t)l * sl + zz * szz];
                if (l > 0 && l < Lh) b = x[(int64_t)(L - l) * sl + zz * szz];
            }
            Xs[(o >> 4) * CST + zz] = a + b;
            Xd[(o >> 4) * CST + zz] = (l > 0 && l < Lh) ? a - b : 0.0;
        }
        __syncthreads();
#pragma unroll
        for (int q = 0; q < NTW; q++) {
            const int nt = tile0 + wave + q * nw;
            if (q >= ntw || nt * 16 > km) continue;
            // this lane's B column: wavenumber k = nt * 16 + n; angle index of (k, l = lc + kk), advancing by 4 k per step
            const int k = min(nt * 16 + n, km);
            int m = (int)(((int64_t)k * (lc + kk)) % L);
            const int fourk = (int)(((int64_t)4 * k) % L);
            for (int ls = 0; ls < ln4; ls += 4) {
                const double2 t = twl[m];
                ac[q] = __builtin_amdgcn_mfma_f64_16x16x4f64(Xs[(ls + kk) * CST + n], t.x, ac[q], 0, 0, 0);
                as[q] = __builtin_amdgcn_mfma_f64_16x16x4f64(Xd[(ls + kk) * CST + n], t.y, as[q], 0, 0, 0);
                m = wrap_add(m, fourk, L);
            }
        }
    }
    // D tile: lane holds column n (wavenumber of the tile), rows kk + 4 r.  sr = ac, si = -as; phase reference e^{-i k off}
    const double inv = 1.0 / L;
#pragma unroll
    for (int q = 0; q < NTW; q++) {
        const int nt = tile0 + wave + q * nw;
        if (q >= ntw || nt * 16 > km) continue;
        const int k = nt * 16 + n;
        if (k > km) continue;
        const double2 w = phr[k];
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const int zz = kk + 4 * r;
            if (zz >= zc) continue;
            const double sr = ac[q][r], si = -as[q][r];
            double2 out;
            if (k == 0) out = make_double2(sr * inv, 0.0);
            else out = make_double2((sr * w.x + si * w.y) * inv, (si * w.x - sr * w.y) * inv);
            // Fl [ring][v][z][blk]: row zz is level z0 + zz of variable v, or (planes) variable zz of a grid with nz = 1
            double *dst = Fl + (planes ? (int64_t)ring * V + zz : ((int64_t)ring * V + v) * nz + z0 + zz) * K2 + 2 * k;
            *reinterpret_cast<double2 *>(dst) = out;
        }
    }
}

// Quarter-wave form of the forward transform (grids with a vertical dimension; every ring length is a multiple of 4):
// with theta = 2 pi / L the four points l, L - l, L/2 - l, L/2 + l share cos / sin(k theta l) up to signs that depend on
// the parity of k only, so the ring is folded onto l = 0 .. L/4 once per parity while staging,
//   even k:  Ce = a + b + c + d,  Se = a - b - c + d        a = x[l], b = x[L - l], c = x[L/2 - l], d = x[L/2 + l]
//   odd k:   Co = a + b - c - d,  So = a - b + c - d        (l = 0: a +- x[L/2], no sine part;  l = L/4: a +- b in Ce / So)
//   sr[k] = sum_l C[l] cos(k theta l),   si[k] = -sum_l S[l] sin(k theta l)
// - half the matrix-core work of the half-ring form.  Wave w transforms wavenumbers of parity w & 1 (tiles of 16: k = 2 (16 t
// + n) + parity, t = (w >> 1) + 4 q), so ONE fetch of the folded values feeds all its tiles (6 MFMAs per fetch with three
// tiles; the f64 matrix cores sustain 65 TFLOP/s that way against 50 at 2 per fetch, profiles/micro/mfma_f64_rate.hip).
constexpr int LCQ = 64;       // quarter-ring points per staged chunk (RL grids)
#ifndef SX_LCZ
#define SX_LCZ 64
#endif
constexpr int LCZ = SX_LCZ;   // the same for k_fl_forward_dft_q (RLZ grids; measured: 64: 0.265 ms, 96: 0.262, 128 - one workgroup per CU by LDS: 0.399)

__global__ void __launch_bounds__(512)
k_fl_forward_dft_q(const double *__restrict__ np1, double *__restrict__ Fl, const int *__restrict__ Lr,
                   const int *__restrict__ kmaxr, const int64_t *__restrict__ pstart, const int64_t *__restrict__ twoff,
                   const double2 *__restrict__ tw, const int64_t *__restrict__ phoff, const double2 *__restrict__ ph, int V, int nz,
                   int K2, int64_t N, const int *__restrict__ items, int lcap) {
    extern __shared__ double sm[];
    const int ring = items[2 * blockIdx.y], v = items[2 * blockIdx.y + 1], z0 = blockIdx.x * DZC;      // largest ring first (see the inverse)
    const int zc = min(DZC, nz - z0);
    const int L = Lr[ring], km = kmaxr[ring], Lh = L / 2, Lq = L / 4;
    double2 *twl = reinterpret_cast<double2 *>(sm);             // [L]
    double *F = sm + 2 * (size_t)lcap;                          // [parity][cosine / sine part][LCZ][CST]
    const double2 *phr = ph + phoff[ring];
    const int64_t p0 = pstart[ring];
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;      // (scalar: the per-tile tests below are then scalar branches, not exec masks)
    for (int m = tid; m < L; m += blockDim.x) twl[m] = tw[twoff[ring] + m];
    const int n = lane & 15, kk = lane >> 4;
    // rings with more than 4 x NTW tiles per parity (kmax > 383) spread their tiles over gridDim.z workgroups, each folding
    // the ring for itself; blockIdx.z = 0 is all there is for the others
    const int par = wave & 1, tq = (wave >> 1) + 4 * NTW * (int)blockIdx.z;     // this wave's parity and first tile
    const int nk = km >= par ? (km - par) / 2 + 1 : 0;              // wavenumbers of this parity
    const int ntile = (nk + 15) / 16;
    if (4 * NTW * (int)blockIdx.z * 16 >= (km / 2 + 1)) return;     // no tile of either parity in this part (uniform for the workgroup)
    dft_d4 ac[NTW], as[NTW];
    int kq[NTW], fourk[NTW];
#pragma unroll
    for (int q = 0; q < NTW; q++) {
        ac[q] = dft_d4{0.0, 0.0, 0.0, 0.0}; as[q] = ac[q];
        kq[q] = min(2 * (16 * (tq + 4 * q) + n) + par, km);         // this lane's B column (clamped: columns past km are dropped)
        fourk[q] = (int)(((int64_t)4 * kq[q]) % L);
    }
    const double *x = np1 + (int64_t)v * N + p0 * nz + z0;
    const double *Fc = F + (size_t)(2 * par) * LCZ * CST, *Fs = Fc + (size_t)LCZ * CST;

    // The four ring points of a (row, level) of chunk c + 1 are requested before the matrix-core steps of chunk c (round 4: 2 x 4
    // values per thread in registers), folded and written to the LDS after them: a chunk used to start with a full round trip to
    // HBM in front of its fold.  Clamped addresses, no branches in the request; the end rows l = 0, L/4 pick their two values below.
    constexpr int NIT = LCZ * DZC / 512;                        // (row, level) items per thread and chunk
    static_assert(LCZ * DZC % 512 == 0, "items per thread");
    double xa[NIT], xb[NIT], xc4[NIT], xd[NIT];
    auto request = [&](int lc) __attribute__((always_inline)) {
#pragma unroll
        for (int it = 0; it < NIT; it++) {
            const int o = tid + it * 512;
            const int zz = min(o & (DZC - 1), zc - 1), l = min(lc + (o >> 4), Lq);
            xa[it] = x[(int64_t)l * nz + zz];
            xb[it] = x[(int64_t)(l == 0 ? 0 : L - l) * nz + zz];
            xc4[it] = x[(int64_t)(Lh - l) * nz + zz];
            xd[it] = x[(int64_t)(Lh + l) * nz + zz];
        }
    };
    request(0);
    for (int lc = 0; lc <= Lq; lc += LCZ) {
        __syncthreads();
#pragma unroll
        for (int it = 0; it < NIT; it++) {
            const int o = tid + it * 512;
            const int zz = o & (DZC - 1), r = o >> 4, l = lc + r;
            double ce = 0.0, se = 0.0, co = 0.0, so = 0.0;
            if (zz < zc && l <= Lq) {
                const double a = xa[it], b = xb[it], c = xc4[it], d = xd[it];
                if (l == 0) {                                   // c = x[L/2]
                    ce = a + c; co = a - c;
                } else if (l == Lq) {                           // b = x[L - L/4]
                    ce = a + b; so = a - b;
                } else {
                    const double ab = a + b, cd = c + d, amb = a - b, dmc = d - c;
                    ce = ab + cd; co = ab - cd; se = amb + dmc; so = amb - dmc;
                }
            }
            F[(0 * LCZ + r) * CST + zz] = ce;
            F[(1 * LCZ + r) * CST + zz] = se;
            F[(2 * LCZ + r) * CST + zz] = co;
            F[(3 * LCZ + r) * CST + zz] = so;
        }
        if (lc + LCZ <= Lq) request(lc + LCZ);                  // in flight during this chunk's matrix-core steps
        __syncthreads();
        if (tq * 16 >= nk) continue;                                // nothing for this wave (uniform)
        const int ln4 = (min(LCZ, Lq + 1 - lc) + 3) & ~3;           // rows past Lq are staged as zeros
        int m[NTW];
#pragma unroll
        for (int q = 0; q < NTW; q++) m[q] = (int)(((int64_t)kq[q] * (lc + kk)) % L);
        for (int ls = 0; ls < ln4; ls += 4) {
            const double xc = Fc[(ls + kk) * CST + n], xs = Fs[(ls + kk) * CST + n];
#pragma unroll
            for (int q = 0; q < NTW; q++) {
                if (tq + 4 * q >= ntile) continue;                  // uniform
                const double2 t = twl[m[q]];
                ac[q] = __builtin_amdgcn_mfma_f64_16x16x4f64(xc, t.x, ac[q], 0, 0, 0);
                as[q] = __builtin_amdgcn_mfma_f64_16x16x4f64(xs, t.y, as[q], 0, 0, 0);
                const unsigned mn = (unsigned)(m[q] + fourk[q]);       // wrap as subtract + unsigned minimum: vector instructions are matrix-pipe
                m[q] = (int)min(mn, mn - (unsigned)L);                 // time (0.2625 -> 0.251 ms).  A branch-free instance per tile count with the
                                                                       // step's operands requested together needs 160 VGPRs: one workgroup per CU, 0.316 ms
            }
        }
    }
    // D tile: lane holds column n (wavenumber of the tile), rows kk + 4 r.  sr = ac, si = -as; phase reference e^{-i k off}
    const double inv = 1.0 / L;
#pragma unroll
    for (int q = 0; q < NTW; q++) {
        const int k = 2 * (16 * (tq + 4 * q) + n) + par;
        if (tq + 4 * q >= ntile || k > km) continue;
        const double2 w = phr[k];
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const int zz = kk + 4 * r;
            if (zz >= zc) continue;
            const double sr = ac[q][r], si = -as[q][r];
            double2 out;
            if (k == 0) out = make_double2(sr * inv, 0.0);
            else out = make_double2((sr * w.x + si * w.y) * inv, (si * w.x - sr * w.y) * inv);
            double *dst = Fl + (((int64_t)ring * V + v) * nz + z0 + zz) * K2 + 2 * k;
            *reinterpret_cast<double2 *>(dst) = out;
        }
    }
}

// ------------------------------------------------------------------------------------------------ RL grids, quarter-wave forms
// Round 4.  The two RL kernels above transform the HALF ring (l = 0 .. L/2) and were launched in two ring classes x column groups.
// Every native ring length is a multiple of 4, so - as on RLZ grids - only l = 0 .. L/4 needs transforming, the even and the odd
// wavenumbers in separate accumulators: half the matrix-core work.  One launch over a work list (ring, part), most expensive
// first; all column groups of a ring inside one workgroup, so a fetched twiddle feeds 2 MFMAs per group (4 for the 19 planes
// of the slab sets) and the coefficient tiles of a ring are staged once per part instead of once per part and group.
constexpr int KCHQ = 96;      // wavenumbers per staged chunk of the RL inverse (a multiple of 8)

template <class ST, int NG>
__global__ void __launch_bounds__(512)
k_rl_inverse_dft_planes_q(const double *__restrict__ A, Planes<ST> phys, const double *__restrict__ phi, const int *__restrict__ Lr,
                          const int *__restrict__ kmaxr, const int64_t *__restrict__ pstart, const int64_t *__restrict__ twoff,
                          const double2 *__restrict__ tw, const int64_t *__restrict__ phoff, const double2 *__restrict__ ph, int V,
                          int K2, int nrings, int64_t N, int64_t arow, PlaneGroups pgs, const int *__restrict__ items, int lcap) {
    extern __shared__ double sm[];
    const int ring = items[2 * blockIdx.x], part = items[2 * blockIdx.x + 1];
    const int L = Lr[ring], km = kmaxr[ring], Lh = L / 2, Lq = L / 4;
    double2 *twl = reinterpret_cast<double2 *>(sm);
    constexpr int CW = NG * CSTP;                              // row stride of the merged coefficient tile: NG groups of 16 (+1) columns
    double *Cc = sm + 2 * (size_t)lcap;                         // [KCHQ][CW]
    double *Cs = Cc + (size_t)KCHQ * CW;
    const int j0 = ring / MUBAR;
    const double2 *phr = ph + phoff[ring];
    const int64_t p0 = pstart[ring];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, nw = blockDim.x >> 6;
    for (int m = tid; m < L; m += blockDim.x) twl[m] = tw[twoff[ring] + m];
    const double *pf = phi + (int64_t)ring * 4;
    const int i = lane & 15, kk = lane >> 4;
    const int mt = part * nw + wave;                            // this wave's row tile of the quarter ring (points mt * 16 .. + 15 of 0 .. L/4)
    const bool rows = mt * 16 <= Lq;
    const int lrow = min(mt * 16 + i, Lq);
    const int sm8 = (int)(((int64_t)8 * lrow) % L);
    dft_d4 P[NG][2], Q[NG][2];
#pragma unroll
    for (int g = 0; g < NG; g++)
#pragma unroll
        for (int par = 0; par < 2; par++) { P[g][par] = dft_d4{0.0, 0.0, 0.0, 0.0}; Q[g][par] = P[g][par]; }
    for (int kc0 = 0; kc0 <= km; kc0 += KCHQ) {
        const int kend = min(km, kc0 + KCHQ - 1), kn = kend - kc0 + 1;
        __syncthreads();                                        // the previous chunk has been consumed (and twl is complete)
        for (int e = tid; e < KCHQ * CW; e += blockDim.x) { Cc[e] = 0.0; Cs[e] = 0.0; }
        __syncthreads();
        for (int e = tid; e < V * kn; e += blockDim.x) {
            const int kl = e % kn, k = kc0 + kl, v = e / kn;
            const double *a = A + (int64_t)j0 * arow + (int64_t)v * K2 + 2 * k;
            const double2 r0 = *reinterpret_cast<const double2 *>(a), r1 = *reinterpret_cast<const double2 *>(a + arow),
                          r2 = *reinterpret_cast<const double2 *>(a + 2 * arow), r3 = *reinterpret_cast<const double2 *>(a + 3 * arow);
            double cr[3], ci[3];
            const double2 w = phr[k];
#pragma unroll
            for (int d = 0; d < 3; d++) {
                const double *f = pf + (int64_t)d * nrings * 4;
                cr[d] = f[0] * r0.x + f[1] * r1.x + f[2] * r2.x + f[3] * r3.x;
                ci[d] = (k == 0) ? 0.0 : f[0] * r0.y + f[1] * r1.y + f[2] * r2.y + f[3] * r3.y;
                if (k > 0) {
                    const double tr = cr[d] * w.x - ci[d] * w.y;
                    ci[d] = 2.0 * (cr[d] * w.y + ci[d] * w.x);
                    cr[d] = 2.0 * tr;
                }
            }
#pragma unroll
            for (int g = 0; g < NG; g++) {
#pragma unroll
                for (int kind = 0; kind < 5; kind++) {
                    const int c = pgs.g[g].colof[v][kind];
                    if (c < 0) continue;
                    double xr, xi;
                    if (kind < 3) { xr = cr[kind]; xi = ci[kind]; }
                    else if (kind == 3) { xr = -(double)k * ci[0]; xi = (double)k * cr[0]; }
                    else { xr = -((double)k * k) * cr[0]; xi = -((double)k * k) * ci[0]; }
                    Cc[kl * CW + g * CSTP + c] = xr;
                    Cs[kl * CW + g * CSTP + c] = xi;
                }
            }
        }
        __syncthreads();
        if (!rows) continue;
#pragma unroll
        for (int par = 0; par < 2; par++) {
            // this lane's wavenumber k = kc0 + 8 js + 2 kk + par; angle index (k l) mod L advances by 8 l per step
            int k = kc0 + 2 * kk + par;
            int m = (int)(((int64_t)k * lrow) % L);
            for (int js = 0; kc0 + 8 * js + par <= kend; js++, k += 8) {
                const int kl = min(k - kc0, KCHQ - 1);          // rows past the chunk's last wavenumber hold zeros
                const double2 t0 = twl[m];
#pragma unroll
                for (int g = 0; g < NG; g++) {
                    const double bc = Cc[kl * CW + g * CSTP + i], bs = Cs[kl * CW + g * CSTP + i];
                    P[g][par] = __builtin_amdgcn_mfma_f64_16x16x4f64(t0.x, bc, P[g][par], 0, 0, 0);
                    Q[g][par] = __builtin_amdgcn_mfma_f64_16x16x4f64(t0.y, bs, Q[g][par], 0, 0, 0);
                }
                m = wrap_add(m, sm8, L);
            }
        }
    }
    if (!rows) return;
#pragma unroll
    for (int g = 0; g < NG; g++) {
        if (i >= pgs.g[g].n) continue;
        const int vv = pgs.g[g].v[i], sl = pgs.g[g].slot[i];
        auto put = [&](int64_t pt, double val) {
            if (sl == 0) phys.val[(int64_t)vv * N + pt] = val;
            else phys.der[((int64_t)(sl - 1) * V + vv) * N + pt] = (ST)val;
        };
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const int lo = mt * 16 + kk + 4 * r;
            if (lo > Lq) continue;
            const double Ps = P[g][0][r] + P[g][1][r], Pd = P[g][0][r] - P[g][1][r], Qs = Q[g][0][r] + Q[g][1][r], Qd = Q[g][0][r] - Q[g][1][r];
            put(p0 + lo, Ps - Qs);
            if (lo > 0) put(p0 + (L - lo), Ps + Qs);
            if (lo < Lq) {
                put(p0 + (Lh - lo), Pd + Qd);
                if (lo > 0) put(p0 + (Lh + lo), Pd - Qd);
            }
        }
    }
}

// forward, RL grids: the quarter-wave fold of k_fl_forward_dft_q with the MFMA rows = the V variables of the ring; a workgroup takes
// one PART of a ring's wavenumbers - wave w: parity w & 1, tile (w >> 1) + 4 part of 16 wavenumbers of that parity, i.e. 128
// wavenumbers per workgroup - and folds the ring for itself.
__global__ void __launch_bounds__(512)
k_fl_forward_dft_qp(const double *__restrict__ np1, double *__restrict__ Fl, const int *__restrict__ Lr,
                    const int *__restrict__ kmaxr, const int64_t *__restrict__ pstart, const int64_t *__restrict__ twoff,
                    const double2 *__restrict__ tw, const int64_t *__restrict__ phoff, const double2 *__restrict__ ph, int V,
                    int K2, int64_t N, const int *__restrict__ items, int lcap) {
    extern __shared__ double sm[];
    const int ring = items[2 * blockIdx.x], part = items[2 * blockIdx.x + 1];
    const int L = Lr[ring], km = kmaxr[ring], Lh = L / 2, Lq = L / 4;
    double2 *twl = reinterpret_cast<double2 *>(sm);             // [L]
    double *F = sm + 2 * (size_t)lcap;                          // [parity][cosine / sine part][LCQ][CST]
    const double2 *phr = ph + phoff[ring];
    const int64_t p0 = pstart[ring];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    for (int m = tid; m < L; m += blockDim.x) twl[m] = tw[twoff[ring] + m];
    const int n = lane & 15, kk = lane >> 4;
    const int par = wave & 1, tq = (wave >> 1) + 4 * part;         // this wave's parity and tile
    const int nk = km >= par ? (km - par) / 2 + 1 : 0;              // wavenumbers of this parity
    const bool live = tq * 16 < nk;
    dft_d4 ac = {0.0, 0.0, 0.0, 0.0}, as = ac;
    const int kq = min(2 * (16 * tq + n) + par, km);                // this lane's B column (clamped: columns past km are dropped)
    const int fourk = (int)(((int64_t)4 * kq) % L);
    const double *x = np1 + p0;
    const double *Fc = F + (size_t)(2 * par) * LCQ * CST, *Fs = Fc + (size_t)LCQ * CST;
    for (int lc = 0; lc <= Lq; lc += LCQ) {
        __syncthreads();
        for (int o = tid; o < LCQ * DZC; o += blockDim.x) {
            const int zz = o >> 6, r = o & (LCQ - 1), l = lc + r;       // ring points fastest across lanes: var_np1 is [v][point]
            double ce = 0.0, se = 0.0, co = 0.0, so = 0.0;
            if (zz < V && l <= Lq) {
                const double *xv = x + (int64_t)zz * N;
                const double a = xv[l];
                if (l == 0) {
                    const double c = xv[Lh];
                    ce = a + c; co = a - c;
                } else if (l == Lq) {
                    const double b = xv[L - l];
                    ce = a + b; so = a - b;
                } else {
                    const double b = xv[L - l], c = xv[Lh - l], d = xv[Lh + l];
                    const double ab = a + b, cd = c + d, amb = a - b, dmc = d - c;
                    ce = ab + cd; co = ab - cd; se = amb + dmc; so = amb - dmc;
                }
            }
            F[(0 * LCQ + r) * CST + zz] = ce;
            F[(1 * LCQ + r) * CST + zz] = se;
            F[(2 * LCQ + r) * CST + zz] = co;
            F[(3 * LCQ + r) * CST + zz] = so;
        }
        __syncthreads();
        if (!live) continue;
        const int ln4 = (min(LCQ, Lq + 1 - lc) + 3) & ~3;           // rows past Lq are staged as zeros
        int m = (int)(((int64_t)kq * (lc + kk)) % L);
        for (int ls = 0; ls < ln4; ls += 4) {
            const double xc = Fc[(ls + kk) * CST + n], xs = Fs[(ls + kk) * CST + n];
            const double2 t = twl[m];
            ac = __builtin_amdgcn_mfma_f64_16x16x4f64(xc, t.x, ac, 0, 0, 0);
            as = __builtin_amdgcn_mfma_f64_16x16x4f64(xs, t.y, as, 0, 0, 0);
            m = wrap_add(m, fourk, L);
        }
    }
    const int k = 2 * (16 * tq + n) + par;
    if (!live || k > km) return;
    const double inv = 1.0 / L;
    const double2 w = phr[k];
#pragma unroll
    for (int r = 0; r < 4; r++) {
        const int zz = kk + 4 * r;
        if (zz >= V) continue;
        const double sr = ac[r], si = -as[r];
        double2 out;
        if (k == 0) out = make_double2(sr * inv, 0.0);
        else out = make_double2((sr * w.x + si * w.y) * inv, (si * w.x - sr * w.y) * inv);
        *reinterpret_cast<double2 *>(Fl + ((int64_t)ring * V + zz) * K2 + 2 * k) = out;
    }
}

// ------------------------------------------------------------------------------------------------ launchers
// grids without a vertical dimension: columns = (variable, plane)
static bool dft_planes(const sx_handle *h) { return !h->has_z; }

bool dft_mfma_ok(const sx_handle *h) {
    if (!h->has_l || fft_path_ok(h)) return false;
    if (h->L_max > DFT_LMAX) return false;      // twiddle table (16 bytes per ring point) + staged chunks must fit 160 KB of LDS
    if (dft_planes(h) && (h->V > 8 || h->D > 5)) return false;
    // fewer than 8 levels leave most of the MFMA N dimension empty: the scalar kernels then, as far as THEY reach (511 points,
    // kmax 255); beyond that the matrix-core kernels run with a partial level chunk rather than refusing the grid
    if (!dft_planes(h) && h->nz < 8 && h->L_max <= 511 && h->kmax_max <= 255) return false;
    static const bool off = getenv("SX_DFT_MFMA") && atoi(getenv("SX_DFT_MFMA")) == 0;       // scalar kernels instead (debugging)
    if (off) return false;
    return h->L_all_mult4;
}

// Rings are launched in classes of growing size so that the small ones do not pay the LDS footprint (and with it the
// occupancy) of the largest: class c covers rings [c n/4, (c+1) n/4), sized for its last ring.
template <class F>
static void for_ring_classes(sx_handle *h, int n_rings, F f, int max_classes = 4) {
    static const int cls_env = getenv("SX_DFT_CLASSES") ? atoi(getenv("SX_DFT_CLASSES")) : 0;        // experiments
    const int ncls = n_rings >= 16 ? (cls_env > 0 ? cls_env : max_classes) : 1;
    for (int c = 0; c < ncls; c++) {
        const int r0 = (int)((int64_t)n_rings * c / ncls), r1 = (int)((int64_t)n_rings * (c + 1) / ncls);
        if (r1 <= r0) continue;
        int lcap = 0, kcap = 0;
        for (int i = r0; i < r1; i++) { lcap = std::max(lcap, h->hL[i]); kcap = std::max(kcap, h->hkmax[i]); }
        f(r0, r1 - r0, lcap, kcap);
    }
}

// (ring, part) work lists of the quarter-wave RL kernels, most expensive ring first: [0] inverse - a part is 8 row tiles of the
// quarter ring (one per wave); [1] forward - a part is 128 wavenumbers (4 tiles of 16 per parity)
static bool rlq_lists(sx_handle *h) {
    if (h->d_rlq_items[0]) return true;
    for (int which = 0; which < 2; which++) {
        std::vector<std::array<int64_t, 3>> it;       // (cost, ring, part)
        for (int r = 0; r < h->nrings; r++) {
            const int tiles = h->hL[r] / 4 / 16 + 1;
            const int parts = which == 0 ? (tiles + 7) / 8 : (h->hkmax[r] + 1 + 127) / 128;
            for (int p = 0; p < parts; p++) it.push_back({(int64_t)h->hL[r], r, p});
        }
        std::stable_sort(it.begin(), it.end(), [](const auto &a, const auto &b) { return a[0] > b[0]; });
        std::vector<int> flat;
        for (const auto &e : it) { flat.push_back((int)e[1]); flat.push_back((int)e[2]); }
        void *d = nullptr;
        if (hipMalloc(&d, sizeof(int) * flat.size()) != hipSuccess || hipMemcpy(d, flat.data(), sizeof(int) * flat.size(), hipMemcpyHostToDevice) != hipSuccess) {
            set_error("RL work list upload failed");
            return false;
        }
        h->allocs.push_back(d);
        h->n_rlq_items[which] = (int)it.size();
        h->d_rlq_items[which] = (int *)d;
    }
    return true;
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
    if (h->rl_quarter && pgs.ng <= 3) {
        // one launch over (ring, part) items, most expensive first; a part = 8 row tiles of the quarter ring (one per wave)
        if (!rlq_lists(h)) return;
        const double *a = h->d_A + (int64_t)h->cell0 * h->C;
        const int lcap = h->L_max;
#define DFT_INVQ(ST, NG)                                                                                                             \
        {                                                                                                                            \
            const size_t lds = sizeof(double) * (2 * (size_t)lcap + (size_t)2 * KCHQ * NG * CSTP);                                   \
            auto kern = k_rl_inverse_dft_planes_q<ST, NG>;                                                                           \
            HIPCHK3(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); \
            hipLaunchKernelGGL(kern, dim3(h->n_rlq_items[0]), dim3(512), lds, h->stream, a, planes_of<ST>(h->d_phys, h->V, h->N),    \
                               h->d_phi, h->d_L, h->d_kmax, h->d_pstart, h->d_twoff, h->d_tw, h->d_phoff, h->d_ph, h->V, h->K2,      \
                               h->nrings, h->N, h->C, pgs, h->d_rlq_items[0], lcap);                                                 \
        }
        if (h->f32) { if (pgs.ng == 1) DFT_INVQ(float, 1) else if (pgs.ng == 2) DFT_INVQ(float, 2) else DFT_INVQ(float, 3) }
        else { if (pgs.ng == 1) DFT_INVQ(double, 1) else if (pgs.ng == 2) DFT_INVQ(double, 2) else DFT_INVQ(double, 3) }
#undef DFT_INVQ
        HIPCHK3(hipGetLastError());
        return;
    }
    const double *a = h->d_A + (int64_t)h->cell0 * h->C;
    // two launch classes only: each launch is as long as its largest ring's workgroup, so more classes mostly add tails
    for_ring_classes(h, h->nrings, [&](int r0, int nr, int lcap, int kcap) {
        const int kcap4 = std::min((kcap + 1 + 3) & ~3, KCH);                               // wavenumbers staged per chunk
        const size_t lds = sizeof(double) * (2 * (size_t)lcap + (size_t)2 * kcap4 * CSTP);
        const int nsplit = std::max(1, ((lcap / 2 + 16) / 16 + 15) / 16);                   // 8 waves x ONE pair of row tiles of the half ring each
#define DFT_INVP(ST)                                                                                                                 \
        {                                                                                                                            \
            auto kern = k_rl_inverse_dft_planes<ST>;                                                                                 \
            HIPCHK3(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); \
            hipLaunchKernelGGL(kern, dim3(nr, nsplit, pgs.ng), dim3(512), lds, h->stream, a, planes_of<ST>(h->d_phys, h->V, h->N),   \
                               h->d_phi, h->d_L, h->d_kmax, h->d_pstart, h->d_twoff, h->d_tw, h->d_phoff, h->d_ph, h->V, h->K2,      \
                               h->nrings, h->N, h->C, pgs, r0, lcap, kcap4);                                                                \
        }
        if (h->f32) DFT_INVP(float) else DFT_INVP(double)
#undef DFT_INVP
        HIPCHK3(hipGetLastError());
    }, 2);
}

#ifdef SX_PHASES
static long long *g_dft_buf = nullptr;
static int64_t g_dft_n = 0;
void dft_phases_dump() {
    const char *path = getenv("SX_DFT_PHASES_OUT");
    if (!path || !g_dft_buf) return;
    std::vector<long long> hst((size_t)g_dft_n * 8);
    hipDeviceSynchronize();
    hipMemcpy(hst.data(), g_dft_buf, sizeof(long long) * hst.size(), hipMemcpyDeviceToHost);
    FILE *f = fopen(path, "wb");
    if (f) { fwrite(hst.data(), sizeof(long long), hst.size(), f); fclose(f); }
}
#else
void dft_phases_dump() {}
#endif

void launch_rl_inverse_dft(sx_handle *h, const int *d_mask) {
    const int id = timer_id(h, "k_rl_inverse");
    timer_begin(h, id);
#ifdef SX_PHASES
    if (!g_dft_buf && !dft_planes(h)) {
        g_dft_n = (int64_t)((h->nz + DZC - 1) / DZC) * h->V * h->nrings;
        hipMalloc(&g_dft_buf, sizeof(long long) * g_dft_n * 8);
        hipMemset(g_dft_buf, 0, sizeof(long long) * g_dft_n * 8);
        hipMemcpyToSymbol(HIP_SYMBOL(g_dft_dbg), &g_dft_buf, sizeof(g_dft_buf));
    }
#endif
    if (dft_planes(h)) {
        launch_rl_inverse_dft_planes(h, d_mask == h->d_mask_full);
        timer_end(h);
        return;
    }
    const double *az = h->has_z ? h->d_Az : h->d_A + (int64_t)h->cell0 * h->C;
    const int64_t azrow = h->has_z ? (int64_t)h->V * 3 * h->nz * h->K2 : h->C;
    {
        // ONE launch over the (ring, variable) work list, most expensive first (four launches by ring size, rings in
        // increasing order: 1.63 ms; one launch, largest first: 1.41 ms - the large rings no longer form the tail)
        const int which = d_mask == h->d_mask_full ? 1 : 0;
        const int nbig = h->n_dft_big[which];
        if (nbig > 0) {      // rings with kmax > 319 (listed first): wavenumbers in chunks, one group of 8 row tiles per workgroup
            const int lcapb = h->L_max;
            const size_t ldsb = sizeof(double) * (2 * (size_t)lcapb + (size_t)2 * KCH * CST);
            dim3 gb((h->nz + DZC - 1) / DZC, nbig, (lcapb / 4 / 16 + 1 + 7) / 8);
#define DFT_INVB(ST)                                                                                                                 \
            {                                                                                                                        \
                auto kern = k_rl_inverse_dft_big<ST>;                                                                                \
                HIPCHK3(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb)); \
                hipLaunchKernelGGL(kern, gb, dim3(512), ldsb, h->stream, az, planes_of<ST>(h->d_phys, h->V, h->N), h->d_phi, h->d_L, \
                                   h->d_kmax, h->d_pstart, h->d_twoff, h->d_tw, h->d_phoff, h->d_ph, h->V, h->nz, h->nsz, h->K2,     \
                                   h->nrings, h->N, azrow, h->slot[0], h->slot[1], h->slot[2], h->slot[3], h->slot[4], h->slot[5],   \
                                   h->slot[6], d_mask, h->d_dft_items[which], lcapb);                                                \
            }
            if (h->f32) DFT_INVB(float) else DFT_INVB(double)
#undef DFT_INVB
            HIPCHK3(hipGetLastError());
        }
        const int lcap = h->dft_lcap_small, kcap4 = (h->dft_kcap_small + 1 + 3) & ~3;
        const size_t lds = sizeof(double) * (2 * (size_t)lcap + (size_t)2 * kcap4 * CST);
        dim3 g((h->nz + DZC - 1) / DZC, h->n_dft_items[which] - nbig, 1);
        if (g.y == 0) { timer_end(h); return; }
        if (h->dft_merge) {
            const bool e8 = h->dft_eighth != 0;
            const int kz = e8 ? 32 * ((h->dft_kcap_small + 1 + 31) / 32) : 8 * ((h->dft_kcap_small / 2 + 1 + 3) / 4);          // rows a K step can touch (see the kernel)
            const size_t pad = e8 ? sizeof(double) * 32 * CST : 0;     // the pipelined loops request one K step beyond the staged rows
            const size_t lds2 = sizeof(double) * (2 * (size_t)lcap + (size_t)4 * kz * CST) + pad;
            const int two = lds2 <= 160 * 1024 ? 1 : 0;            // two coefficient sets beside the twiddle table: kmax <= ~260
            const size_t ldsm = two ? lds2 : sizeof(double) * (2 * (size_t)lcap + (size_t)2 * kz * CST) + pad;
            // two 256-thread workgroups per CU (HT) where one set + half the twiddle table fit 80 KB
            const size_t ldsh = sizeof(double) * ((size_t)2 * kz * CST + (size_t)lcap);
            if (e8 && h->dft_half_wg && ldsh <= 80 * 1024) {
#define DFT_INVH(ST)                                                                                                                 \
                {                                                                                                                    \
                    auto kern = k_rl_inverse_dft_merged<ST, 2, true>;                                                                \
                    HIPCHK3(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsh)); \
                    hipLaunchKernelGGL(kern, g, dim3(256), ldsh, h->stream, az, planes_of<ST>(h->d_phys, h->V, h->N), h->d_phi, h->d_L, \
                                       h->d_kmax, h->d_pstart, h->d_twoff, h->d_tw, h->d_phoff, h->d_ph, h->V, h->nz, h->nsz, h->K2, \
                                       h->nrings, h->N, azrow, h->slot[0], h->slot[1], h->slot[2], h->slot[3], h->slot[4], h->slot[5], \
                                       h->slot[6], d_mask, h->d_dft_items[which] + 2 * nbig, lcap, kz, 0);                           \
                }
                if (h->f32) DFT_INVH(float) else DFT_INVH(double)
#undef DFT_INVH
                HIPCHK3(hipGetLastError());
                timer_end(h);
                return;
            }
#define DFT_INVM(ST)                                                                                                                 \
            {                                                                                                                        \
                auto kern = e8 ? k_rl_inverse_dft_merged<ST, 2> : k_rl_inverse_dft_merged<ST, 0>;                                    \
                HIPCHK3(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsm)); \
                hipLaunchKernelGGL(kern, g, dim3(512), ldsm, h->stream, az, planes_of<ST>(h->d_phys, h->V, h->N), h->d_phi, h->d_L,  \
                                   h->d_kmax, h->d_pstart, h->d_twoff, h->d_tw, h->d_phoff, h->d_ph, h->V, h->nz, h->nsz, h->K2,     \
                                   h->nrings, h->N, azrow, h->slot[0], h->slot[1], h->slot[2], h->slot[3], h->slot[4], h->slot[5],   \
                                   h->slot[6], d_mask, h->d_dft_items[which] + 2 * nbig, lcap, kz, two);                             \
            }
            if (h->f32) DFT_INVM(float) else DFT_INVM(double)
#undef DFT_INVM
            HIPCHK3(hipGetLastError());
            timer_end(h);
            return;
        }
#define DFT_INV(ST)                                                                                                                  \
        {                                                                                                                            \
            auto kern = k_rl_inverse_dft<ST>;                                                                                        \
            HIPCHK3(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); \
            hipLaunchKernelGGL(kern, g, dim3(512), lds, h->stream, az, planes_of<ST>(h->d_phys, h->V, h->N), h->d_phi, h->d_L,       \
                               h->d_kmax, h->d_pstart, h->d_twoff, h->d_tw, h->d_phoff, h->d_ph, h->V, h->nz, h->nsz, h->K2,         \
                               h->nrings, h->N, azrow, h->slot[0], h->slot[1], h->slot[2], h->slot[3], h->slot[4], h->slot[5],       \
                               h->slot[6], d_mask, h->d_dft_items[which] + 2 * nbig, lcap, kcap4);                                   \
        }
        if (h->f32) DFT_INV(float) else DFT_INV(double)
#undef DFT_INV
        HIPCHK3(hipGetLastError());
    }
    timer_end(h);
}

void launch_fl_forward_dft(sx_handle *h) {
    const int id = timer_id(h, "k_fl_forward");
    timer_begin(h, id);
    const int planes = dft_planes(h) ? 1 : 0;
    static const bool half = getenv("SX_DFT_HALF") && atoi(getenv("SX_DFT_HALF")) != 0;      // A/B: the half-ring kernel
    if (planes && h->rl_quarter && rlq_lists(h)) {      // RL grids: quarter-wave fold, one launch over (ring, part) items
        const int lcap = h->L_max;
        const size_t lds = sizeof(double) * (2 * (size_t)lcap + (size_t)4 * LCQ * CST);
        HIPCHK3(hipFuncSetAttribute(reinterpret_cast<const void *>(k_fl_forward_dft_qp), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(k_fl_forward_dft_qp, dim3(h->n_rlq_items[1]), dim3(512), lds, h->stream, h->d_np1, h->d_Fl, h->d_L, h->d_kmax, h->d_pstart,
                           h->d_twoff, h->d_tw, h->d_phoff, h->d_ph, h->V, h->K2, h->N, h->d_rlq_items[1], lcap);
        HIPCHK3(hipGetLastError());
        timer_end(h);
        return;
    }
    if (!planes && !half) {          // quarter-wave form; 4 waves x NTW tiles x 16 = 192 wavenumbers per parity >= (kmax <= 319) / 2 + 1
        // ONE launch over the work list (four launches by ring size, each with its own tail: 0.32 -> 0.29 ms)
        // rings with kmax > 319 (listed first) in their own launch: the wavenumber tiles of a ring spread over gridDim.z
        // workgroups (12 tiles of 16 wavenumbers per parity each); the others as before, sized for their largest ring
        const int nbig = h->n_dft_big[2];
        for (int part = 0; part < 2; part++) {
            const int n = part == 0 ? nbig : h->n_dft_items[2] - nbig;
            if (n == 0) continue;
            const int lcap = part == 0 ? h->L_max : h->dft_lcap_small;
            const int kparts = part == 0 ? ((h->kmax_max / 2 + 1 + 15) / 16 + 4 * NTW - 1) / (4 * NTW) : 1;
            const size_t lds = sizeof(double) * (2 * (size_t)lcap + (size_t)4 * LCZ * CST);
            dim3 g((h->nz + DZC - 1) / DZC, n, kparts);
            HIPCHK3(hipFuncSetAttribute(reinterpret_cast<const void *>(k_fl_forward_dft_q), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            hipLaunchKernelGGL(k_fl_forward_dft_q, g, dim3(512), lds, h->stream, h->d_np1, h->d_Fl, h->d_L, h->d_kmax, h->d_pstart,
                               h->d_twoff, h->d_tw, h->d_phoff, h->d_ph, h->V, h->nz, h->K2, h->N, h->d_dft_items[2] + (part == 0 ? 0 : 2 * nbig), lcap);
            HIPCHK3(hipGetLastError());
        }
        timer_end(h);
        return;
    }
    for_ring_classes(h, h->nrings, [&](int r0, int nr, int lcap, int) {
        const size_t lds = sizeof(double) * (2 * (size_t)lcap + (size_t)2 * LCH * CST);
        // planes: one wavenumber tile per wave, the ring's (kmax + 1) / 16 tiles spread over gridDim.y workgroups of 8 waves
        const int ntw = planes ? 1 : NTW;
        int kcap = 0;
        for (int i = r0; i < r0 + nr; i++) kcap = std::max(kcap, h->hkmax[i]);
        dim3 g(planes ? 1 : (h->nz + DZC - 1) / DZC, planes ? ((kcap + 1 + 15) / 16 + 7) / 8 : h->V, nr);
        HIPCHK3(hipFuncSetAttribute(reinterpret_cast<const void *>(k_fl_forward_dft), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(k_fl_forward_dft, g, dim3(512), lds, h->stream, h->d_np1, h->d_Fl, h->d_L, h->d_kmax, h->d_pstart,
                           h->d_twoff, h->d_tw, h->d_phoff, h->d_ph, h->V, h->nz, h->K2, h->N, r0, lcap, planes, ntw);
        HIPCHK3(hipGetLastError());
    }, planes ? 2 : 4);
    timer_end(h);
}

}  // namespace sx
