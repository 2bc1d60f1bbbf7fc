// Interface-only ("partitioned", SPIKE-type) form of the patch-level B -> A spline solve for radial tiles.
//
// The reference gathers every tile's B coefficients on the master and lets EVERY worker solve the whole patch
// (src/semiimplicit.jl:272-285, splineTransform! -> Springsteel SAtransform).  The transposed solve (sx_a2a_*) moves all
// n/N + 3 rows of a tile twice and still runs the full 2 x (nc + 3)-row recurrence per column on every rank.  Here every
// tile solves ITS OWN rows and only what couples tiles travels - per tile and column 10 values each way:
//
//     M a = b,  M = Gamma (P + eps_q Q) Gamma^T = D + R       D = diagonal blocks of the tiles (the free unknowns a tile
//                                                             owns), R = band entries that couple neighbouring tiles
//     b = b' + E f        b' = what a tile forms from rows only it holds; f = the <= 4 rows of a tile whose spline
//                         coefficient belongs to ANOTHER tile: its 3 halo rows (src/semiimplicit.jl:320-329) and, for
//                         PERIODIC, the wrap-around rows
//     y' = D^-1 b'        tile-local banded Cholesky solve, chain length n_t = n / N           (k_iface_local)
//     y_I = y'_I + W f    I = first 3 + last 3 unknowns of every tile (6 N "interface" unknowns)
//     a_I = T^-1 y_I      T = 1 + (D^-1 R)_II                                                  |
//     c   = (E f)_I - R_II a_I     correction of the right-hand side at a tile's 6 edge rows   |  one dense [10 N x 10 N]
//     x   = Gamma^T a_I            the <= 4 rows a tile evaluates but does not own             |  matrix per BC class,
//                                                                                              |  built here (k_iface_reduce)
//     a   = y' + Z c      Z = D_t^-1 restricted to the 6 edge columns: 6 multiply-adds per row (k_iface_apply)
//
// The matrices depend on the boundary-condition class and the tile table only; they are built once, in extended
// precision, from the very matrix build_spline_class factors for the one-patch solve.  The columns of the reduced system
// are split over the ranks like those of the transposed solve, so a step needs two all-to-alls of 10 rows instead of
// two of n / N + 3.  No pack / unpack kernels: k_iface_local writes its 10 rows straight into the send buffer, k_iface_apply
// reads the returned 10 rows straight from the receive buffer.
#include "sx_internal.hpp"
#include <algorithm>
#include <cmath>

namespace sx {

#define HIPCHK(x)                                                                                   \
    do {                                                                                            \
        hipError_t e_ = (x);                                                                        \
        if (e_ != hipSuccess) set_error(std::string(#x) + ": " + hipGetErrorString(e_) + " (" __FILE__ ":" + std::to_string(__LINE__) + ")"); \
    } while (0)

constexpr int IF_E = 6, IF_X = 4, IF_R = IF_E + IF_X;      // edge values + foreign rows = rows per tile that travel
constexpr int IF_META = 8;                                  // per class: nt, rowbase, nlead, ntrail, xrow[4]

struct IfaceState {
    int n = 0, me = 0, g0 = 0, g1 = 0, nt_max = 0;
    std::vector<int64_t> colstart;                          // [n + 1]
    int *d_meta = nullptr, *d_owner = nullptr;
    int64_t *d_soff = nullptr, *d_cw = nullptr, *d_cs = nullptr;
    double *d_Lf = nullptr, *d_Z = nullptr, *d_Q = nullptr, *d_Y = nullptr;
    std::vector<void *> bufs;
};

// ------------------------------------------------------------------------------------------------ device
struct P1 { double x; };
struct P2 { double x, y; };
__device__ __forceinline__ P1 ldp(const double *p, P1 *) { return P1{p[0]}; }
__device__ __forceinline__ P2 ldp(const double *p, P2 *) { const double2 v = *reinterpret_cast<const double2 *>(p); return P2{v.x, v.y}; }
__device__ __forceinline__ void stp(double *p, P1 v) { p[0] = v.x; }
__device__ __forceinline__ void stp(double *p, P2 v) { *reinterpret_cast<double2 *>(p) = make_double2(v.x, v.y); }
__device__ __forceinline__ P1 operator+(P1 a, P1 b) { return P1{a.x + b.x}; }
__device__ __forceinline__ P2 operator+(P2 a, P2 b) { return P2{a.x + b.x, a.y + b.y}; }
__device__ __forceinline__ P1 operator-(P1 a, P1 b) { return P1{a.x - b.x}; }
__device__ __forceinline__ P2 operator-(P2 a, P2 b) { return P2{a.x - b.x, a.y - b.y}; }
__device__ __forceinline__ P1 operator*(double c, P1 a) { return P1{c * a.x}; }
__device__ __forceinline__ P2 operator*(double c, P2 a) { return P2{c * a.x, c * a.y}; }
__device__ __forceinline__ P1 zerop(P1 *) { return P1{0.0}; }
__device__ __forceinline__ P2 zerop(P2 *) { return P2{0.0, 0.0}; }

// y' = D_t^-1 b' for the columns of one lane (one column, or the (Re, Im) pair of a wavenumber), rows coalesced across lanes.
// B: the tile's rows [nbt][C]; Y: [nbt][C] scratch holding y' at the tile-local row of each unknown; send: this column's
// place in the send buffer, slot s at send[s * cwd].
template <class T, int U>
__device__ __forceinline__ void iface_local_columns(const double *__restrict__ B, double *__restrict__ Y, double *__restrict__ send,
                                                    int64_t cwd, const int *__restrict__ meta, const double *__restrict__ gl,
                                                    const double *__restrict__ gr, const double *__restrict__ Lf, int nbt, int64_t C,
                                                    int64_t col, bool active) {
    const int nt = meta[0], rb = meta[1], nlead = meta[2], ntrail = meta[3];
    extern __shared__ double sfac[];                 // [nt + 3][4]: l0, l1, l2, 1 / diagonal; three zero rows behind the last
    for (int e = threadIdx.x; e < (nt + 3) * 4; e += blockDim.x) sfac[e] = e < nt * 4 ? Lf[e] : 0.0;
    __syncthreads();
    if (!active) return;
    T *tp = nullptr;
#define BROW(r) ldp(B + (int64_t)(r) * C + col, tp)
#define YROW(i) ldp(Y + (int64_t)(rb + (i)) * C + col, tp)
#define YSET(i, v) stp(Y + (int64_t)(rb + (i)) * C + col, (v))
    // rows of this tile that fold onto its first / last two unknowns (rank-1 / rank-2 boundary conditions at the patch edges)
    T bl0 = zerop(tp), bl1 = zerop(tp), br0 = zerop(tp), br1 = zerop(tp);
    for (int q = 0; q < nlead; q++) { const T b = BROW(q); bl0 = bl0 + gl[q * 2] * b; bl1 = bl1 + gl[q * 2 + 1] * b; }
    for (int q = 0; q < ntrail; q++) { const T b = BROW(nbt - 1 - q); br0 = br0 + gr[q * 2] * b; br1 = br1 + gr[q * 2 + 1] * b; }
    T y1 = zerop(tp), y2 = zerop(tp), y3 = zerop(tp);
    auto fwd = [&](int i, T s) {
        const double4 l = *reinterpret_cast<const double4 *>(sfac + (size_t)i * 4);
        s = s - (l.z * y1 + l.y * y2 + l.x * y3);
        s = l.w * s;
        y3 = y2; y2 = y1; y1 = s;
        YSET(i, s);
    };
    fwd(0, BROW(rb) + bl0);
    fwd(1, BROW(rb + 1) + bl1);
    {   // interior rows [2, nt - 2) in branch-free batches, the next batch requested before this one is computed and stored
        const int lo = 2, hi = nt - 2, full = lo + ((hi - lo) / U) * U;
        T rhs[U], nxt[U];
        if (lo < full) {
#pragma unroll
            for (int u = 0; u < U; u++) rhs[u] = BROW(rb + lo + u);
        }
        for (int i0 = lo; i0 < full; i0 += U) {
            const bool more = i0 + U < full;
#pragma unroll
            for (int u = 0; u < U; u++) nxt[u] = BROW(rb + (more ? i0 + U + u : i0 + u));
#pragma unroll
            for (int u = 0; u < U; u++) fwd(i0 + u, rhs[u]);
#pragma unroll
            for (int u = 0; u < U; u++) rhs[u] = nxt[u];
        }
        for (int i = full; i < hi; i++) fwd(i, BROW(rb + i));
    }
    fwd(nt - 2, BROW(rb + nt - 2) + br1);
    fwd(nt - 1, BROW(rb + nt - 1) + br0);
    // back substitution; rows nt .. nt + 2 of the factor are zero, so every row takes the same three-term form
    T x1 = zerop(tp), x2 = zerop(tp), x3 = zerop(tp);
    T e[IF_E];
    auto bwd = [&](int i, T s) {
        s = s - (sfac[(size_t)(i + 1) * 4 + 2] * x1 + sfac[(size_t)(i + 2) * 4 + 1] * x2 + sfac[(size_t)(i + 3) * 4 + 0] * x3);
        s = sfac[(size_t)i * 4 + 3] * s;
        x3 = x2; x2 = x1; x1 = s;
        YSET(i, s);
        return s;
    };
    e[5] = bwd(nt - 1, y1);
    e[4] = bwd(nt - 2, y2);
    e[3] = bwd(nt - 3, y3);
    {
        int ib = nt - 4;
        T rhs[U], nxt[U];
        if (ib - U + 1 >= 3) {
#pragma unroll
            for (int u = 0; u < U; u++) rhs[u] = YROW(ib - u);
        }
        for (; ib - U + 1 >= 3; ib -= U) {
            const bool more = ib - 2 * U + 1 >= 3;
#pragma unroll
            for (int u = 0; u < U; u++) nxt[u] = YROW(more ? ib - U - u : ib - u);
#pragma unroll
            for (int u = 0; u < U; u++) bwd(ib - u, rhs[u]);
#pragma unroll
            for (int u = 0; u < U; u++) rhs[u] = nxt[u];
        }
        for (; ib >= 3; ib--) bwd(ib, YROW(ib));
    }
    e[2] = bwd(2, YROW(2));
    e[1] = bwd(1, YROW(1));
    e[0] = bwd(0, YROW(0));
#pragma unroll
    for (int j = 0; j < IF_E; j++) stp(send + (int64_t)j * cwd, e[j]);
#pragma unroll
    for (int s = 0; s < IF_X; s++) {
        const int xr = meta[4 + s];
        stp(send + (int64_t)(IF_E + s) * cwd, xr >= 0 ? BROW(xr) : zerop(tp));
    }
#undef BROW
#undef YROW
#undef YSET
}

// The same for tiles of at most NT unknowns (the case that matters: 8 GPUs share 171 cells): EVERY right-hand-side row of
// the lane is requested in one burst and the whole recurrence runs in registers - one memory round trip for the loads, one
// burst of stores; the memory-resident form above pays a round trip per batch of 8 rows, twice (y' is re-read for the back
// substitution), which is all of its run time at this size.
template <class T, int NT>
__device__ __forceinline__ void iface_local_regs(const double *__restrict__ B, double *__restrict__ Y, double *__restrict__ send,
                                                 int64_t cwd, const int *__restrict__ meta, const double *__restrict__ gl,
                                                 const double *__restrict__ gr, const double *Lf, int nbt, int64_t C,
                                                 int64_t col, bool active) {
    const int nt = meta[0], rb = meta[1], nlead = meta[2], ntrail = meta[3];
    T *tp = nullptr;
    T y[NT], xv[IF_X];
    if (active) {
#pragma unroll
        for (int i = 0; i < NT; i++) y[i] = i < nt ? ldp(B + (int64_t)(rb + i) * C + col, tp) : zerop(tp);
    }
    // factor rows through LDS (requested behind the right-hand sides, arrive before them): as scalar loads they came in
    // batches of a few rows, each batch a round trip to L2 on the critical path of the recurrence
    extern __shared__ double sfac[];                 // [nt + 3][4]
    for (int e = threadIdx.x; e < (nt + 3) * 4; e += blockDim.x) sfac[e] = Lf[e];
    __syncthreads();
    if (!active) return;
    Lf = sfac;
#pragma unroll
    for (int s = 0; s < IF_X; s++) { const int xr = meta[4 + s]; xv[s] = xr >= 0 ? ldp(B + (int64_t)xr * C + col, tp) : zerop(tp); }
    T bl0 = zerop(tp), bl1 = zerop(tp), br0 = zerop(tp), br1 = zerop(tp);
    for (int q = 0; q < nlead; q++) { const T b = ldp(B + (int64_t)q * C + col, tp); bl0 = bl0 + gl[q * 2] * b; bl1 = bl1 + gl[q * 2 + 1] * b; }
    for (int q = 0; q < ntrail; q++) { const T b = ldp(B + (int64_t)(nbt - 1 - q) * C + col, tp); br0 = br0 + gr[q * 2] * b; br1 = br1 + gr[q * 2 + 1] * b; }
    y[0] = y[0] + bl0;
    y[1] = y[1] + bl1;
    T y1 = zerop(tp), y2 = zerop(tp), y3 = zerop(tp);
#pragma unroll
    for (int i = 0; i < NT; i++) {
        if (i < nt) {
            T s = y[i];
            if (i == nt - 2) s = s + br1;
            if (i == nt - 1) s = s + br0;
            s = s - (Lf[i * 4 + 2] * y1 + Lf[i * 4 + 1] * y2 + Lf[i * 4 + 0] * y3);
            s = Lf[i * 4 + 3] * s;
            y3 = y2; y2 = y1; y1 = s;
            y[i] = s;
        }
    }
    T x1 = zerop(tp), x2 = zerop(tp), x3 = zerop(tp);
    T e[IF_E];
#pragma unroll
    for (int j = 0; j < IF_E; j++) e[j] = zerop(tp);
#pragma unroll
    for (int i = NT - 1; i >= 0; i--) {
        if (i < nt) {
            T s = y[i];
            // rows nt .. nt + 2 of the factor array are zero (build_class_ops pads them), so the three-term form holds for every row
            s = s - (Lf[(i + 1) * 4 + 2] * x1 + Lf[(i + 2) * 4 + 1] * x2 + Lf[(i + 3) * 4 + 0] * x3);
            s = Lf[i * 4 + 3] * s;
            x3 = x2; x2 = x1; x1 = s;
            stp(Y + (int64_t)(rb + i) * C + col, s);
            if (i == nt - 1) e[5] = s;
            if (i == nt - 2) e[4] = s;
            if (i == nt - 3) e[3] = s;
            if (i < 3) e[i] = s;
        }
    }
#pragma unroll
    for (int j = 0; j < IF_E; j++) stp(send + (int64_t)j * cwd, e[j]);
#pragma unroll
    for (int s = 0; s < IF_X; s++) stp(send + (int64_t)(IF_E + s) * cwd, xv[s]);
}

constexpr int IF_NT_REGS = 32;     // tiles with at most this many unknowns take the register-resident kernels

// grid: x = waves of 64 wavenumbers + one block for the k = 0 column, y = (variable, z-mode) group; block = 64
template <bool REGS>
__global__ void __launch_bounds__(64)
k_iface_local(const double *__restrict__ B, double *__restrict__ Y, double *__restrict__ send, const int *__restrict__ owner,
              const int64_t *__restrict__ soff, const int64_t *__restrict__ cw, const int64_t *__restrict__ cs,
              const int *__restrict__ cls, const int *__restrict__ meta, const double *__restrict__ gl,
              const double *__restrict__ gr, const double *__restrict__ Lf, int nbt, int Zb, int K2, int64_t C) {
    const int g = blockIdx.y, v = g / Zb, d = owner[g];
    const bool k0 = (blockIdx.x == gridDim.x - 1);
    const int c = cls[v * 2 + (k0 ? 0 : 1)];
    const int k = 1 + blockIdx.x * 64 + threadIdx.x;
    const bool act = k0 ? threadIdx.x == 0 : 2 * k + 1 < K2;
    const int64_t col = (int64_t)g * K2 + (k0 ? 0 : (act ? 2 * k : 2));
    double *sp = send + soff[d] + (col - cs[d]);
    const double *Lc = Lf + (int64_t)c * (nbt + 3) * 4;
    if (REGS) {
        if (k0) iface_local_regs<P1, IF_NT_REGS>(B, Y, sp, cw[d], meta + c * IF_META, gl + c * 6, gr + c * 6, Lc, nbt, C, col, act);
        else iface_local_regs<P2, IF_NT_REGS>(B, Y, sp, cw[d], meta + c * IF_META, gl + c * 6, gr + c * 6, Lc, nbt, C, col, act);
    } else {
        if (k0) iface_local_columns<P1, 8>(B, Y, sp, cw[d], meta + c * IF_META, gl + c * 6, gr + c * 6, Lc, nbt, C, col, act);
        else iface_local_columns<P2, 8>(B, Y, sp, cw[d], meta + c * IF_META, gl + c * 6, gr + c * 6, Lc, nbt, C, col, act);
    }
}

// out = Q_class in, per column of this rank's share: in / out [tile][IF_R][cwm]; Qt = Q transposed, [class][input i][RNp outputs],
// RNp = RN rounded up to 16.  This is a plain dense product Out[RN x cols] = Q[RN x RN] In[RN x cols] (6,400 multiply-adds per
// column at N = 8) and runs on the f64 matrix cores: a wave owns 16 columns, its In fragments are requested in one burst,
// the operator fragments come from L2 per 16-row output tile (the scheme of k_colmat_mfma).  Operator entries fed from LDS
// broadcasts or from scalar loads both left the kernel waiting for its operands (0.03 ms for 8,256 columns).
// The k = 0 column of a group has its own boundary-condition class: the blocks behind the first ng * nchunk take one each,
// a thread per output; the 16-column tile that contains it skips that column.
typedef double iface_d4 __attribute__((ext_vector_type(4)));
template <int KS>                    // K steps of 4: K = RN <= 4 KS (KS = 20: up to 8 tiles, the operator fragments of the next
__global__ void __launch_bounds__(256)   // output tile are requested before the current tile's MFMA chain; KS = 40: up to 16)
k_iface_reduce(const double *__restrict__ in, double *__restrict__ out, const double *__restrict__ Qt, const int *__restrict__ cls,
               int g0, int ng, int nchunk, int Zb, int K2, int64_t cwm, int RN, int RNp) {
    if ((int)blockIdx.x >= ng * nchunk) {                       // the k = 0 column of group gl_: thread o = output o
        const int gl_ = blockIdx.x - ng * nchunk, v = (g0 + gl_) / Zb;
        const double *Qc = Qt + (int64_t)cls[v * 2 + 0] * RN * RNp;
        const int64_t j = (int64_t)gl_ * K2;
        __shared__ double sx_in[4 * KS];
        for (int i = threadIdx.x; i < 4 * KS; i += blockDim.x) sx_in[i] = i < RN ? in[(int64_t)i * cwm + j] : 0.0;
        __syncthreads();
        for (int o = threadIdx.x; o < RN; o += blockDim.x) {
            double acc[4] = {0.0, 0.0, 0.0, 0.0};
            for (int i0 = 0; i0 < RN; i0 += 20) {               // RN is a multiple of 10: 20 independent operator loads per batch
                double q[20];
#pragma unroll
                for (int u = 0; u < 20; u++) q[u] = (i0 + u < RN) ? Qc[(int64_t)(i0 + u) * RNp + o] : 0.0;
#pragma unroll
                for (int u = 0; u < 20; u++) acc[u & 3] = __builtin_fma(q[u], sx_in[i0 + u], acc[u & 3]);
            }
            out[(int64_t)o * cwm + j] = (acc[0] + acc[1]) + (acc[2] + acc[3]);
        }
        return;
    }
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, n = lane & 15, kk = lane >> 4;
    const int gl_ = blockIdx.x / nchunk, chunk = blockIdx.x - gl_ * nchunk, v = (g0 + gl_) / Zb;
    const int blk0 = (chunk * 4 + wave) * 16;
    if (blk0 >= K2) return;
    const int blk = min(blk0 + n, K2 - 1);
    const int64_t j = (int64_t)gl_ * K2 + blk;
    const double *Qc = Qt + (int64_t)cls[v * 2 + 1] * RN * RNp;
    const int ksteps = (RN + 3) / 4;
    double b[KS], a[KS], an[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ks++) {
        const int k = 4 * ks + kk;
        b[ks] = (ks < ksteps && k < RN) ? in[(int64_t)k * cwm + j] : 0.0;
    }
    auto load_a = [&](double *dst, int t) {
#pragma unroll
        for (int ks = 0; ks < KS; ks++) {
            const int k = 4 * ks + kk;
            dst[ks] = (ks < ksteps && k < RN) ? Qc[(int64_t)k * RNp + t * 16 + n] : 0.0;
        }
    };
    const bool okc = blk0 + n < K2 && blk0 + n >= 1;            // column 0 belongs to the k = 0 block
    const int ntile = (RN + 15) / 16;
    load_a(a, 0);
    for (int t = 0; t < ntile; t++) {
        if (KS <= 20) load_a(an, min(t + 1, ntile - 1));
        iface_d4 acc = iface_d4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int ks = 0; ks < KS; ks++)
            if (ks < ksteps) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a[ks], b[ks], acc, 0, 0, 0);
        if (okc) {
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const int o = t * 16 + kk + 4 * r;
                if (o < RN) out[(int64_t)o * cwm + j] = acc[r];
            }
        }
        if (KS <= 20) {
#pragma unroll
            for (int ks = 0; ks < KS; ks++) a[ks] = an[ks];
        } else if (t + 1 < ntile) {
            load_a(a, t + 1);
        }
    }
}

// a = y' + Z c for the tile's own unknowns, the rows that depend on them through a boundary condition, and the rows that
// came back from their owners; writes the patch A rows [cell0, cell0 + nbt).  Lane per column (or (Re, Im) pair) walking
// the tile's rows: the column's 10 returned values are loaded once, Z rows are wave-uniform (scalar loads).
// grid as k_iface_local.
template <class T>
__device__ __forceinline__ void iface_apply_columns(const double *__restrict__ Y, const double *__restrict__ rp, int64_t cwd,
                                                    double *__restrict__ A, const int *__restrict__ m, const double *__restrict__ gl,
                                                    const double *__restrict__ gr, const double *Zc, int nbt, int64_t C,
                                                    int64_t col, bool active) {
    const int nt = m[0], rb = m[1], nlead = m[2], ntrail = m[3];
    T *tp = nullptr;
    T cv[IF_R];
    if (active) {
#pragma unroll
        for (int jx = 0; jx < IF_R; jx++) cv[jx] = ldp(rp + (int64_t)jx * cwd, tp);
    }
    extern __shared__ double sz[];                   // [nt][6]
    for (int e = threadIdx.x; e < nt * IF_E; e += blockDim.x) sz[e] = Zc[e];
    __syncthreads();
    if (!active) return;
    Zc = sz;
    auto corr = [&](int i, T a) {
#pragma unroll
        for (int jx = 0; jx < IF_E; jx++) a = a + Zc[i * IF_E + jx] * cv[jx];
        return a;
    };
    T a0 = zerop(tp), a1 = zerop(tp), am1 = zerop(tp), am2 = zerop(tp);
    constexpr int U = 8;
    for (int i0 = 0; i0 < nt; i0 += U) {
        T yv[U];
#pragma unroll
        for (int u = 0; u < U; u++) yv[u] = ldp(Y + (int64_t)(rb + min(i0 + u, nt - 1)) * C + col, tp);
#pragma unroll
        for (int u = 0; u < U; u++) {
            const int i = i0 + u;
            if (i < nt) {
                const T a = corr(i, yv[u]);
                stp(A + (int64_t)(rb + i) * C + col, a);
                if (i == 0) a0 = a;
                if (i == 1) a1 = a;
                if (i == nt - 1) am1 = a;
                if (i == nt - 2) am2 = a;
            }
        }
    }
    for (int q = 0; q < nlead; q++) stp(A + (int64_t)q * C + col, gl[q * 2] * a0 + gl[q * 2 + 1] * a1);
    for (int q = 0; q < ntrail; q++) stp(A + (int64_t)(nbt - 1 - q) * C + col, gr[q * 2] * am1 + gr[q * 2 + 1] * am2);
#pragma unroll
    for (int s = 0; s < IF_X; s++)
        if (m[4 + s] >= 0) stp(A + (int64_t)m[4 + s] * C + col, cv[IF_E + s]);
}

__global__ void __launch_bounds__(64)
k_iface_apply(const double *__restrict__ Y, const double *__restrict__ recv, double *__restrict__ A, const int *__restrict__ owner,
              const int64_t *__restrict__ soff, const int64_t *__restrict__ cw, const int64_t *__restrict__ cs,
              const int *__restrict__ cls, const int *__restrict__ meta, const double *__restrict__ gl,
              const double *__restrict__ gr, const double *__restrict__ Z, int nbt, int Zb, int K2, int64_t C) {
    const int g = blockIdx.y, v = g / Zb, d = owner[g];
    const bool k0 = (blockIdx.x == gridDim.x - 1);
    const int c = cls[v * 2 + (k0 ? 0 : 1)];
    const int k = 1 + blockIdx.x * 64 + threadIdx.x;
    const bool act = k0 ? threadIdx.x == 0 : 2 * k + 1 < K2;
    const int64_t col = (int64_t)g * K2 + (k0 ? 0 : (act ? 2 * k : 2));
    const double *rp = recv + soff[d] + (col - cs[d]);
    if (k0) iface_apply_columns<P1>(Y, rp, cw[d], A, meta + c * IF_META, gl + c * 6, gr + c * 6, Z + (int64_t)c * nbt * IF_E, nbt, C, col, act);
    else iface_apply_columns<P2>(Y, rp, cw[d], A, meta + c * IF_META, gl + c * 6, gr + c * 6, Z + (int64_t)c * nbt * IF_E, nbt, C, col, act);
}

// ------------------------------------------------------------------------------------------------ host: operators
typedef long double xr;

// banded (half-bandwidth 3) Cholesky of a dense SPD block, extended precision; L[i][0..3] = L(i, i-3 .. i)
static bool band_cholesky(const std::vector<xr> &D, int n, std::vector<xr> &L) {
    L.assign((size_t)n * 4, 0.0L);
    auto Lij = [&](int i, int j) -> xr { return (i - j > 3 || j > i || j < 0) ? 0.0L : L[(size_t)i * 4 + (3 - (i - j))]; };
    for (int i = 0; i < n; i++)
        for (int j = std::max(0, i - 3); j <= i; j++) {
            xr s = D[(size_t)i * n + j];
            for (int k = std::max(0, i - 3); k < j; k++) s -= Lij(i, k) * Lij(j, k);
            if (i == j) {
                if (!(s > 0.0L)) return false;
                L[(size_t)i * 4 + 3] = sqrtl(s);
            } else {
                L[(size_t)i * 4 + (3 - (i - j))] = s / Lij(j, j);
            }
        }
    return true;
}

static void band_solve(const std::vector<xr> &L, int n, std::vector<xr> &x) {   // x <- (L L^T)^-1 x
    auto Lij = [&](int i, int j) -> xr { return L[(size_t)i * 4 + (3 - (i - j))]; };
    for (int i = 0; i < n; i++) {
        xr s = x[i];
        for (int k = std::max(0, i - 3); k < i; k++) s -= Lij(i, k) * x[k];
        x[i] = s / Lij(i, i);
    }
    for (int i = n - 1; i >= 0; i--) {
        xr s = x[i];
        for (int k = i + 1; k <= std::min(n - 1, i + 3); k++) s -= Lij(k, i) * x[k];
        x[i] = s / Lij(i, i);
    }
}

static bool invert(std::vector<xr> &A, int n) {           // Gauss-Jordan with partial pivoting, in place
    std::vector<xr> I((size_t)n * n, 0.0L);
    for (int i = 0; i < n; i++) I[(size_t)i * n + i] = 1.0L;
    for (int c = 0; c < n; c++) {
        int p = c;
        for (int r = c + 1; r < n; r++)
            if (fabsl(A[(size_t)r * n + c]) > fabsl(A[(size_t)p * n + c])) p = r;
        if (A[(size_t)p * n + c] == 0.0L) return false;
        if (p != c)
            for (int k = 0; k < n; k++) { std::swap(A[(size_t)p * n + k], A[(size_t)c * n + k]); std::swap(I[(size_t)p * n + k], I[(size_t)c * n + k]); }
        const xr piv = 1.0L / A[(size_t)c * n + c];
        for (int k = 0; k < n; k++) { A[(size_t)c * n + k] *= piv; I[(size_t)c * n + k] *= piv; }
        for (int r = 0; r < n; r++) {
            if (r == c) continue;
            const xr f = A[(size_t)r * n + c];
            if (f == 0.0L) continue;
            for (int k = 0; k < n; k++) { A[(size_t)r * n + k] -= f * A[(size_t)c * n + k]; I[(size_t)r * n + k] -= f * I[(size_t)c * n + k]; }
        }
    }
    A.swap(I);
    return true;
}

static std::vector<xr> mm(const std::vector<xr> &A, const std::vector<xr> &B, int n, int k, int m) {
    std::vector<xr> Cm((size_t)n * m, 0.0L);
    for (int i = 0; i < n; i++)
        for (int p = 0; p < k; p++) {
            const xr a = A[(size_t)i * k + p];
            if (a == 0.0L) continue;
            for (int j = 0; j < m; j++) Cm[(size_t)i * m + j] += a * B[(size_t)p * m + j];
        }
    return Cm;
}

struct ClassOps {
    int nt = 0, rowbase = 0, nlead = 0, ntrail = 0, xrow[IF_X] = {-1, -1, -1, -1};
    std::vector<double> Lf, Z, Q;       // [nbt + 3][4], [nbt][6], Q transposed and padded: [RN inputs][RNp outputs]
};

// Everything the three kernels need for one boundary-condition class and the tile table; `me` selects whose local factor
// and edge columns are kept.  Returns false with `err` set when the partition does not admit the interface form.
static bool build_class_ops(const SplineClass &sc, int nb, int n, int me, const int *cell0, const int *ncells, int nbt_me, ClassOps &out,
                            std::string &err) {
    const int nf = sc.nfree, sh = sc.periodic ? 1 : sc.rl, RN = IF_R * n;
    std::vector<int> u0(n), u1(n), own(n);
    for (int t = 0; t < n; t++) {
        own[t] = ncells[t] + (t == n - 1 ? 3 : 0);
        u0[t] = std::max(0, cell0[t] - sh);
        u1[t] = std::min(nf, cell0[t] + own[t] - sh);
        if (u1[t] - u0[t] < IF_E) { err = "interface-only solve: a tile owns fewer than 6 free spline coefficients (use more cells per tile or another exchange mode)"; return false; }
        if (t > 0 && u0[t] != u1[t - 1]) { err = "interface-only solve: tiles do not partition the unknowns"; return false; }
    }
    if (u0[0] != 0 || u1[n - 1] != nf) { err = "interface-only solve: tiles do not cover the unknowns"; return false; }
    auto tile_of = [&](int u) { for (int t = 0; t < n; t++) if (u >= u0[t] && u < u1[t]) return t; return -1; };
    // interface unknowns: first 3 and last 3 of every tile
    const int NI = IF_E * n;
    std::vector<int> I(NI);
    std::vector<int> idxI(nf, -1);
    for (int t = 0; t < n; t++)
        for (int j = 0; j < IF_E; j++) {
            I[t * IF_E + j] = j < 3 ? u0[t] + j : u1[t] - IF_E + j;
            idxI[I[t * IF_E + j]] = t * IF_E + j;
        }
    // Gamma by patch row
    std::vector<std::vector<std::pair<int, double>>> byrow(nb);
    for (int u = 0; u < nf; u++)
        for (auto &e : sc.Gam[u]) byrow[e.first].push_back({u, e.second});
    // rows of each tile whose coefficient belongs to another tile
    std::vector<std::vector<int>> xrows(n);
    for (int t = 0; t < n; t++)
        for (int r = cell0[t]; r < cell0[t] + ncells[t] + 3; r++) {
            int inside = 0, outside = 0;
            for (auto &e : byrow[r]) (e.first >= u0[t] && e.first < u1[t] ? inside : outside)++;
            if (outside && inside) { err = "interface-only solve: a spline row folds onto unknowns of two tiles"; return false; }
            if (outside) xrows[t].push_back(r);
        }
    for (int t = 0; t < n; t++)
        if ((int)xrows[t].size() > IF_X) { err = "interface-only solve: more than 4 foreign rows in a tile"; return false; }
    // tile-local factors and the edge columns of D_t^-1
    std::vector<std::vector<xr>> Zx(n);
    std::vector<xr> Lme;
    for (int t = 0; t < n; t++) {
        const int nt = u1[t] - u0[t];
        std::vector<xr> D((size_t)nt * nt), L;
        for (int i = 0; i < nt; i++)
            for (int j = 0; j < nt; j++) D[(size_t)i * nt + j] = (std::abs(i - j) <= 3) ? (xr)sc.Mdense[(size_t)(u0[t] + i) * nf + (u0[t] + j)] : 0.0L;
        if (!band_cholesky(D, nt, L)) { err = "interface-only solve: a tile's diagonal block is not positive definite"; return false; }
        Zx[t].assign((size_t)nt * IF_E, 0.0L);
        for (int j = 0; j < IF_E; j++) {
            std::vector<xr> x(nt, 0.0L);
            x[j < 3 ? j : nt - IF_E + j] = 1.0L;
            band_solve(L, nt, x);
            for (int i = 0; i < nt; i++) Zx[t][(size_t)i * IF_E + j] = x[i];
        }
        if (t == me) Lme = L;
    }
    // reduced system
    std::vector<xr> RII((size_t)NI * NI, 0.0L), DII((size_t)NI * NI, 0.0L);
    for (int a = 0; a < NI; a++)
        for (int b = 0; b < NI; b++) {
            const int ta = a / IF_E, tb = b / IF_E;
            if (ta != tb) RII[(size_t)a * NI + b] = sc.Mdense[(size_t)I[a] * nf + I[b]];
            else DII[(size_t)a * NI + b] = Zx[ta][(size_t)(I[a] - u0[ta]) * IF_E + (b % IF_E)];
        }
    // every coupling between tiles must run between interface unknowns
    for (int u = 0; u < nf; u++)
        for (int w = 0; w < nf; w++)
            if (sc.Mdense[(size_t)u * nf + w] != 0.0 && tile_of(u) != tile_of(w) && (idxI[u] < 0 || idxI[w] < 0)) {
                err = "interface-only solve: coupling outside the interface rows";
                return false;
            }
    std::vector<xr> T = mm(DII, RII, NI, NI, NI);
    for (int a = 0; a < NI; a++) T[(size_t)a * NI + a] += 1.0L;
    if (!invert(T, NI)) { err = "interface-only solve: singular reduced system"; return false; }
    const int NX = IF_X * n;
    std::vector<xr> EI((size_t)NI * NX, 0.0L);
    for (int t = 0; t < n; t++)
        for (size_t s = 0; s < xrows[t].size(); s++)
            for (auto &e : byrow[xrows[t][s]]) {
                if (idxI[e.first] < 0) { err = "interface-only solve: a foreign row folds onto a non-interface unknown"; return false; }
                EI[(size_t)idxI[e.first] * NX + (t * IF_X + s)] = e.second;
            }
    std::vector<xr> Gx((size_t)NX * NI);
    for (int a = 0; a < NI; a++)
        for (int b = 0; b < NX; b++) Gx[(size_t)b * NI + a] = EI[(size_t)a * NX + b];
    const std::vector<xr> W = mm(DII, EI, NI, NI, NX), TW = mm(T, W, NI, NI, NX);
    std::vector<xr> Cy = mm(RII, T, NI, NI, NI), Cf = mm(RII, TW, NI, NI, NX);
    const std::vector<xr> Xy = mm(Gx, T, NX, NI, NI), Xf = mm(Gx, TW, NX, NI, NX);
    const int RNp = (RN + 15) / 16 * 16;
    out.Q.assign((size_t)RN * RNp, 0.0);
    auto row_of = [&](int t, int j) { return t * IF_R + j; };
    for (int t = 0; t < n; t++)
        for (int j = 0; j < IF_R; j++)
            for (int t2 = 0; t2 < n; t2++)
                for (int j2 = 0; j2 < IF_R; j2++) {
                    xr q;
                    if (j < IF_E && j2 < IF_E) q = -Cy[(size_t)(t * IF_E + j) * NI + (t2 * IF_E + j2)];
                    else if (j < IF_E) q = EI[(size_t)(t * IF_E + j) * NX + (t2 * IF_X + j2 - IF_E)] - Cf[(size_t)(t * IF_E + j) * NX + (t2 * IF_X + j2 - IF_E)];
                    else if (j2 < IF_E) q = Xy[(size_t)(t * IF_X + j - IF_E) * NI + (t2 * IF_E + j2)];
                    else q = Xf[(size_t)(t * IF_X + j - IF_E) * NX + (t2 * IF_X + j2 - IF_E)];
                    out.Q[(size_t)row_of(t2, j2) * RNp + row_of(t, j)] = (double)q;
                }
    // this tile
    out.nt = u1[me] - u0[me];
    out.rowbase = u0[me] + sh - cell0[me];
    out.nlead = (!sc.periodic && me == 0) ? sc.rl : 0;
    out.ntrail = (!sc.periodic && me == n - 1) ? sc.rr : 0;
    for (int s = 0; s < IF_X; s++) out.xrow[s] = s < (int)xrows[me].size() ? xrows[me][s] - cell0[me] : -1;
    // what the kernels assume about the rows of this tile: unknown i sits in tile row rowbase + i, rows below rowbase /
    // above the last unknown are boundary-condition rows of the patch or foreign rows
    for (int i = 0; i < out.nt; i++) {
        bool found = false;
        for (auto &e : sc.Gam[u0[me] + i]) found |= (e.first == cell0[me] + out.rowbase + i && e.second == 1.0);
        if (!found) { err = "interface-only solve: unexpected row / unknown correspondence"; return false; }
    }
    if (out.rowbase < 0 || out.rowbase + out.nt > nbt_me) { err = "interface-only solve: unknowns outside the tile's rows"; return false; }
    out.Lf.assign((size_t)(nbt_me + 3) * 4, 0.0);        // three zero rows behind the last unknown (see the back substitutions)
    out.Z.assign((size_t)nbt_me * IF_E, 0.0);
    for (int i = 0; i < out.nt; i++) {
        for (int q = 0; q < 3; q++) out.Lf[(size_t)i * 4 + q] = (double)Lme[(size_t)i * 4 + q];
        out.Lf[(size_t)i * 4 + 3] = (double)(1.0L / Lme[(size_t)i * 4 + 3]);
        for (int j = 0; j < IF_E; j++) out.Z[(size_t)i * IF_E + j] = (double)Zx[me][(size_t)i * IF_E + j];
    }
    return true;
}

template <class T>
static bool up(sx_handle *h, IfaceState *st, T **p, const std::vector<T> &v) {
    void *d = nullptr;
    const size_t bytes = sizeof(T) * std::max<size_t>(v.size(), 1);
    if (hipMalloc(&d, bytes) != hipSuccess) { set_error("hipMalloc failed (interface solve)"); return false; }
    st->bufs.push_back(d);
    h->dev_bytes += bytes;
    if (!v.empty() && hipMemcpy(d, v.data(), sizeof(T) * v.size(), hipMemcpyHostToDevice) != hipSuccess) { set_error("hipMemcpy H2D failed"); return false; }
    *p = (T *)d;
    return true;
}

void iface_release(sx_handle *h) {
    IfaceState *st = (IfaceState *)h->iface_state;
    if (!st) return;
    (void)hipStreamSynchronize(h->stream);
    for (void *b : st->bufs) (void)hipFree(b);
    delete st;
    h->iface_state = nullptr;
}

}  // namespace sx

using namespace sx;

extern "C" {

int sx_iface_configure(sx_handle *h, int32_t n, int32_t me, const int32_t *cell0, const int32_t *ncells) {
    clear_error();
    if (!tile_table_ok(h, n, me, cell0, ncells)) return 1;
    if (n < 2) { set_error("interface-only solve needs at least 2 tiles (one tile: sx_spline_transform)"); return 1; }
    if (n > 16) { set_error("interface-only solve: at most 16 tiles"); return 1; }
    iface_release(h);
    IfaceState *st = new IfaceState();
    h->iface_state = st;
    st->n = n; st->me = me;
    std::string err;
    std::vector<int> meta((size_t)h->ncls * IF_META);
    const int RN = IF_R * n, RNp = (RN + 15) / 16 * 16;
    std::vector<double> Lf((size_t)h->ncls * (h->nbt + 3) * 4), Z((size_t)h->ncls * h->nbt * IF_E), Q((size_t)h->ncls * RN * RNp);
    int nt_max = 0;
    for (int c = 0; c < h->ncls; c++) {
        ClassOps co;
        if (!build_class_ops(h->classes[c], h->b_rDim, n, me, cell0, ncells, h->nbt, co, err)) {
            iface_release(h);
            set_error(err);
            return 1;
        }
        int *m = &meta[(size_t)c * IF_META];
        m[0] = co.nt; m[1] = co.rowbase; m[2] = co.nlead; m[3] = co.ntrail;
        for (int s = 0; s < IF_X; s++) m[4 + s] = co.xrow[s];
        nt_max = std::max(nt_max, co.nt);
        std::copy(co.Lf.begin(), co.Lf.end(), Lf.begin() + (size_t)c * (h->nbt + 3) * 4);
        std::copy(co.Z.begin(), co.Z.end(), Z.begin() + (size_t)c * h->nbt * IF_E);
        std::copy(co.Q.begin(), co.Q.end(), Q.begin() + (size_t)c * RN * RNp);
    }
    // columns split over the ranks by whole (variable, z-mode) groups, as in the transposed solve
    const int G = h->V * h->Zb;
    std::vector<int> owner(G);
    std::vector<int64_t> cs(n + 1), cw(n), soff(n);
    for (int d = 0; d <= n; d++) cs[d] = (int64_t)((int64_t)G * d / n) * h->K2;
    int64_t o = 0;
    for (int d = 0; d < n; d++) {
        cw[d] = cs[d + 1] - cs[d];
        for (int64_t g = cs[d] / h->K2; g < cs[d + 1] / h->K2; g++) owner[g] = d;
        soff[d] = o;
        o += (int64_t)IF_R * cw[d];
    }
    st->colstart = cs;
    st->nt_max = nt_max;
    st->g0 = (int)(cs[me] / h->K2);
    st->g1 = (int)(cs[me + 1] / h->K2);
    std::vector<int64_t> csn(cs.begin(), cs.begin() + n);
    std::vector<double> y((size_t)h->nbt * h->C, 0.0);
    if (!up(h, st, &st->d_meta, meta) || !up(h, st, &st->d_Lf, Lf) || !up(h, st, &st->d_Z, Z) || !up(h, st, &st->d_Q, Q) ||
        !up(h, st, &st->d_owner, owner) || !up(h, st, &st->d_soff, soff) || !up(h, st, &st->d_cw, cw) || !up(h, st, &st->d_cs, csn) ||
        !up(h, st, &st->d_Y, y)) {
        const std::string keep = sx_last_error();
        iface_release(h);
        set_error(keep);
        return 1;
    }
    return 0;
}

int sx_iface_col_starts(sx_handle *h, int64_t *out) {
    clear_error();
    IfaceState *st = h ? (IfaceState *)h->iface_state : nullptr;
    if (!st || !out) { set_error("sx_iface_configure has not been called"); return 1; }
    for (int d = 0; d <= st->n; d++) out[d] = st->colstart[d];
    return 0;
}

int sx_iface_local(sx_handle *h, void *dev_send) {
    clear_error();
    IfaceState *st = h ? (IfaceState *)h->iface_state : nullptr;
    if (!st || !dev_send) { set_error("sx_iface_local: invalid argument / not configured"); return 1; }
    const int id = timer_id(h, "k_iface_local");
    timer_begin(h, id);
    dim3 g((h->K2 > 1 ? (h->K2 / 2 - 1 + 63) / 64 : 0) + 1, h->V * h->Zb);
#define IFL_ARGS h->d_Btile, st->d_Y, (double *)dev_send, st->d_owner, st->d_soff, st->d_cw, st->d_cs, h->d_cls, st->d_meta, h->d_gl, h->d_gr, \
                 st->d_Lf, h->nbt, h->Zb, h->K2, h->C
    if (st->nt_max <= IF_NT_REGS) hipLaunchKernelGGL(k_iface_local<true>, g, dim3(64), sizeof(double) * 4 * (h->nbt + 3), h->stream, IFL_ARGS);
    else hipLaunchKernelGGL(k_iface_local<false>, g, dim3(64), sizeof(double) * 4 * (h->nbt + 3), h->stream, IFL_ARGS);
#undef IFL_ARGS
    HIPCHK(hipGetLastError());
    timer_end(h);
    return error_status();
}

int sx_iface_reduce(sx_handle *h, const void *dev_recv, void *dev_send) {
    clear_error();
    IfaceState *st = h ? (IfaceState *)h->iface_state : nullptr;
    if (!st || !dev_recv || !dev_send) { set_error("sx_iface_reduce: invalid argument / not configured"); return 1; }
    const int id = timer_id(h, "k_iface_reduce");
    timer_begin(h, id);
    const int ng = st->g1 - st->g0, RN = IF_R * st->n, RNp = (RN + 15) / 16 * 16;
    if (ng > 0) {
        const int nchunk = (h->K2 + 63) / 64;        // a block = 4 waves x 16 columns
#define IFR_ARGS dim3(ng * nchunk + ng), dim3(256), 0, h->stream, (const double *)dev_recv, (double *)dev_send, st->d_Q, h->d_cls, st->g0, ng, nchunk, \
                 h->Zb, h->K2, (int64_t)ng * h->K2, RN, RNp
        if (RN <= 80) hipLaunchKernelGGL(k_iface_reduce<20>, IFR_ARGS);
        else hipLaunchKernelGGL(k_iface_reduce<40>, IFR_ARGS);
#undef IFR_ARGS
        HIPCHK(hipGetLastError());
    }
    timer_end(h);
    return error_status();
}

int sx_iface_apply(sx_handle *h, const void *dev_recv) {
    clear_error();
    IfaceState *st = h ? (IfaceState *)h->iface_state : nullptr;
    if (!st || !dev_recv) { set_error("sx_iface_apply: invalid argument / not configured"); return 1; }
    const int id = timer_id(h, "k_iface_apply");
    timer_begin(h, id);
    hipLaunchKernelGGL(k_iface_apply, dim3((h->K2 > 1 ? (h->K2 / 2 - 1 + 63) / 64 : 0) + 1, h->V * h->Zb), dim3(64), sizeof(double) * IF_E * h->nbt, h->stream, st->d_Y, (const double *)dev_recv,
                       h->d_A + (int64_t)h->cell0 * h->C, st->d_owner, st->d_soff, st->d_cw, st->d_cs, h->d_cls, st->d_meta, h->d_gl, h->d_gr,
                       st->d_Z, h->nbt, h->Zb, h->K2, h->C);
    HIPCHK(hipGetLastError());
    timer_end(h);
    return error_status();
}

}  // extern "C"
