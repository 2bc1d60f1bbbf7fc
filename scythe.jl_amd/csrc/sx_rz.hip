// RZ grids (radius x Chebyshev column, no azimuth): tileTransform! and spectralTransform! as two fused matrix-core kernels.
//
// The general kernels keep the wavenumber block as the fastest index of every spectral array - on an RZ grid that index has
// extent ONE, so the vertical kernels (k_colmat_mfma, k_sbw_mfma: 16 wavenumber blocks per MFMA column tile) run with 1 of 16
// columns in use and one active lane per wave (config 3, RZ 513 x 128: k_sbz 0.20 ms, k_zinv 0.13 ms, k_rl_inverse 0.08 ms
// for 66 k points).  Here the RADIUS is the matrix-core dimension instead - this is the one place of the path where the work is a
// genuine dense contraction (the Chebyshev collocation operators, [zDim x b_zDim] per variable):
//   inverse  physical[slot][v][ring][z] = sum_zm M_sz[v][z][zm] * (sum_j phi_d[ring][j] A[node(ring) + j][v][zm])
//            workgroup = (16 rings, variable, group of 4 level tiles): the radial 4-term sums of the 16 rings go to LDS once
//            ([d][zm][ring], d = value / d/dr / d2/dr2), each wave takes one tile of 16 levels and runs the <= 5 requested
//            (d, sz) products as v_mfma_f64_16x16x4 chains (m = ring, n = level, k = zm; operator fragments from L2, coalesced),
//            writing 128-byte level runs of `physical`.  k_zinv + k_rl_inverse + the Az array are gone.
//   forward  B[node][v][zm] = sum_z CB[zm][z] * (sum_{rings of cells node-3 .. node} wq phi0[ring][node - cell] var_np1[v][ring][z])
//            workgroup = (16 nodes, variable, group of mode tiles): the radial inner products of the 16 nodes go to LDS
//            ([z][node]), each wave contracts them with CB^T for its 16-mode tiles (m = node, n = zm, k = z).
//            k_fl_forward (a copy on RZ grids) + k_sbz + the Fl array are gone.
// Any zDim / b_zDim <= 256 (partial tiles are masked).  SX_RZ_FUSED=0 (read at sx_create) keeps the general kernels.
#include "sx_internal.hpp"

namespace sx {

#define HIPCHK4(x)                                                                                  \
    do {                                                                                            \
        hipError_t e_ = (x);                                                                        \
        if (e_ != hipSuccess) set_error(std::string(#x) + ": " + hipGetErrorString(e_) + " (" __FILE__ ":" + std::to_string(__LINE__) + ")"); \
    } while (0)

typedef double rz_d4 __attribute__((ext_vector_type(4)));
constexpr int RZ_T = 16;          // rings (inverse) / nodes (forward) per workgroup: one MFMA row tile
constexpr int RZ_KC = 16;         // K steps whose operator fragments are in registers at once (32 VGPRs)

// One MFMA pass of the inverse: NA accumulators (the radial sums at LDS offsets aoff[]) against ONE vertical operator, K in chunks of
// RZ_KC steps whose operator fragments are requested together; no predicate anywhere in it - the LDS rows beyond b_zDim are zero
// (so whatever a clamped operator address returns is multiplied by zero) and lanes beyond zDim are dropped at the store.
template <int NA>
__device__ __forceinline__ void rz_pass(const double *__restrict__ sm, const int (&aoff)[3], const double *__restrict__ op, int64_t nz,
                                        int Zb, int nchunks, int lane, int kk, rz_d4 (&acc)[3]) {
#pragma unroll
    for (int a = 0; a < NA; a++) acc[a] = rz_d4{0.0, 0.0, 0.0, 0.0};
    for (int c = 0; c < nchunks; c++) {
        double b[RZ_KC];
#pragma unroll
        for (int q = 0; q < RZ_KC; q++) b[q] = op[(int64_t)min(4 * (c * RZ_KC + q) + kk, Zb - 1) * nz];      // B[k = zm][n = level]
#pragma unroll
        for (int q = 0; q < RZ_KC; q++) {
            const int o = (4 * (c * RZ_KC + q)) * RZ_T + lane;                                                // A[m = ring (lane & 15)][k = zm (lane >> 4)]
#pragma unroll
            for (int a = 0; a < NA; a++) acc[a] = __builtin_amdgcn_mfma_f64_16x16x4f64(sm[aoff[a] + o], b[q], acc[a], 0, 0, 0);
        }
    }
}

template <class ST>
__global__ void __launch_bounds__(256)
k_rz_inverse(const double *__restrict__ Arows /* tile rows [nbt][C] */, Planes<ST> phys, const double *__restrict__ phi /* [3][nrings][4] */,
             const double *__restrict__ MzT /* [v][sz][Zb][nz] */, const int *__restrict__ slotmask, int V, int nz, int Zb, int nrings,
             int64_t N, int64_t C, int s_u, int s_r, int s_rr, int s_z, int s_zz) {
    extern __shared__ double sm[];                      // [3][Zp][16]  radial sums; Zp = b_zDim rounded up to a whole chunk of K steps
    const int ring0 = blockIdx.x * RZ_T, v = blockIdx.y;
    const int mask = slotmask[v];
    const bool n_u = (mask >> s_u) & 1, n_r = (mask >> s_r) & 1, n_rr = (mask >> s_rr) & 1, n_z = (mask >> s_z) & 1, n_zz = (mask >> s_zz) & 1;
    if (!(n_u || n_r || n_rr || n_z || n_zz)) return;
    constexpr int CH = 4 * RZ_KC;                       // coefficient rows per chunk
    const int Zp = (Zb + CH - 1) / CH * CH, nchunks = Zp / CH;
    // ---- radial 4-term sums of this workgroup's 16 rings: Ar[d][zm][ring].  Thread = (ring t >> 4, 16 consecutive modes per step):
    // the ring's 12 basis weights are fetched once, every step reads 128-byte pieces of the four A rows
    {
        const int rl = threadIdx.x >> 4, l16 = threadIdx.x & 15;
        const int ring = min(ring0 + rl, nrings - 1);
        const double *p0 = phi + (int64_t)ring * 4, *p1 = p0 + (int64_t)nrings * 4, *p2 = p1 + (int64_t)nrings * 4;
        double w[3][4];
#pragma unroll
        for (int j = 0; j < 4; j++) { w[0][j] = p0[j]; w[1][j] = p1[j]; w[2][j] = p2[j]; }
        const double *a = Arows + (int64_t)(ring / MUBAR) * C + (int64_t)v * Zb;
#pragma unroll 4
        for (int zm = l16; zm < Zp; zm += 16) {
            const int zc = min(zm, Zb - 1);
            const double live = zm < Zb ? 1.0 : 0.0;
            const double a0 = live * a[zc], a1 = live * a[C + zc], a2 = live * a[2 * C + zc], a3 = live * a[3 * C + zc];
            sm[(0 * Zp + zm) * RZ_T + rl] = w[0][0] * a0 + w[0][1] * a1 + w[0][2] * a2 + w[0][3] * a3;
            sm[(1 * Zp + zm) * RZ_T + rl] = w[1][0] * a0 + w[1][1] * a1 + w[1][2] * a2 + w[1][3] * a3;
            sm[(2 * Zp + zm) * RZ_T + rl] = w[2][0] * a0 + w[2][1] * a1 + w[2][2] * a2 + w[2][3] * a3;
        }
    }
    __syncthreads();
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, n16 = lane & 15, kk = lane >> 4;
    const int nzt = (nz + 15) / 16;
    const int zt = blockIdx.z * 4 + wave;                // this wave's tile of 16 levels
    if (zt >= nzt) return;
    const int z = zt * 16 + n16;
    const bool zok = z < nz;
    const int zc = min(z, nz - 1);
    // D[m = ring kk + 4 r][n = level]: 16 consecutive levels of a ring = one 128-byte run
    auto store = [&](const rz_d4 &acc, int slot) {
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const int ring = ring0 + kk + 4 * r;
            if (ring < nrings && zok) {
                const int64_t pt = (int64_t)ring * nz + z;
                if (slot == 0) phys.val[(int64_t)v * N + pt] = acc[r];
                else phys.der[((int64_t)(slot - 1) * V + v) * N + pt] = (ST)acc[r];
            }
        }
    };
    rz_d4 acc[3];
    // pass 0 (operator "value"): the radial sums the mask asks for - value, d/dr, d2/dr2 - share every operator fragment
    {
        int aoff[3], slots[3], na = 0;
        if (n_u) { aoff[na] = 0; slots[na++] = s_u; }
        if (n_r) { aoff[na] = Zp * RZ_T; slots[na++] = s_r; }
        if (n_rr) { aoff[na] = 2 * Zp * RZ_T; slots[na++] = s_rr; }
        for (int a = na; a < 3; a++) { aoff[a] = 0; slots[a] = 0; }
        const double *op = MzT + ((int64_t)v * 3 + 0) * Zb * nz + zc;
        if (na == 3) rz_pass<3>(sm, aoff, op, nz, Zb, nchunks, lane, kk, acc);
        else if (na == 2) rz_pass<2>(sm, aoff, op, nz, Zb, nchunks, lane, kk, acc);
        else if (na == 1) rz_pass<1>(sm, aoff, op, nz, Zb, nchunks, lane, kk, acc);
        for (int a = 0; a < na; a++) store(acc[a], slots[a]);
    }
    // passes 1, 2 (d/dz, d2/dz2 of the value)
    for (int sz = 1; sz < 3; sz++) {
        if (!(sz == 1 ? n_z : n_zz)) continue;
        const int aoff[3] = {0, 0, 0};
        rz_pass<1>(sm, aoff, MzT + ((int64_t)v * 3 + sz) * Zb * nz + zc, nz, Zb, nchunks, lane, kk, acc);
        store(acc[0], sz == 1 ? s_z : s_zz);
    }
}

// Vertical transform FIRST, on the spline NODES (one third as many as rings): workgroup = (13 cells = the 16 nodes they touch, variable,
// group of 4 level tiles).  The A rows of the 16 nodes go to LDS ([zm][node]); each wave takes one tile of 16 levels and forms
// Az[sz][node][level] = M_sz A for the <= 3 operators the mask asks for (one LDS operand, three operator fragments: 3 MFMAs per LDS
// read); the result goes back to LDS and the same wave evaluates the 39 rings of the 13 cells from it - 4-term sums with the basis
// weights, value / d/dr / d2/dr2 from the "value" operator, d/dz and d2/dz2 from the other two - writing 128-byte level runs.
// 96 instead of 160 MFMAs per wave and a third of the workgroups of the ring-tile form above (k_rz_inverse, kept for SX_RZ_INV=0).
constexpr int RZ_CT = RZ_T - 3;       // cells per workgroup: their rings need nodes c .. c + 3, i.e. 16 nodes in all

template <int NB, int KC>
__device__ __forceinline__ void rz_ops_pass(const double *__restrict__ sa, const double *const (&op)[3], int64_t rs, int nrows, int nchunks,
                                            int lane, int kk, rz_d4 (&acc)[3]) {
#pragma unroll
    for (int a = 0; a < NB; a++) acc[a] = rz_d4{0.0, 0.0, 0.0, 0.0};
    for (int c = 0; c < nchunks; c++) {
        double b[NB][KC];
#pragma unroll
        for (int q = 0; q < KC; q++)
#pragma unroll
            for (int a = 0; a < NB; a++) b[a][q] = op[a][(int64_t)min(4 * (c * KC + q) + kk, nrows - 1) * rs];      // rows beyond nrows meet zeros
#pragma unroll
        for (int q = 0; q < KC; q++) {
            const double av = sa[(4 * (c * KC + q)) * RZ_T + lane];
#pragma unroll
            for (int a = 0; a < NB; a++) acc[a] = __builtin_amdgcn_mfma_f64_16x16x4f64(av, b[a][q], acc[a], 0, 0, 0);
        }
    }
}

template <class ST>
__global__ void __launch_bounds__(256)
k_rz_inverse_nodes(const double *__restrict__ Arows /* tile rows [nbt][C] */, Planes<ST> phys, const double *__restrict__ phi /* [3][nrings][4] */,
                   const double *__restrict__ MzT /* [v][sz][Zb][nz] */, const int *__restrict__ slotmask, int V, int nz, int Zb, int ncells,
                   int nbt, int64_t N, int64_t C, int s_u, int s_r, int s_rr, int s_z, int s_zz) {
    extern __shared__ double sm[];
    const int c0 = blockIdx.x * RZ_CT, v = blockIdx.y;   // first cell = first node of the workgroup
    const int mask = slotmask[v];
    const bool n_u = (mask >> s_u) & 1, n_r = (mask >> s_r) & 1, n_rr = (mask >> s_rr) & 1, n_z = (mask >> s_z) & 1, n_zz = (mask >> s_zz) & 1;
    if (!(n_u || n_r || n_rr || n_z || n_zz)) return;
    constexpr int KC = 8, CH = 4 * KC, ZS = 65;          // Az row stride: 64 levels per workgroup + 1
    const int Zp = (Zb + CH - 1) / CH * CH, nchunks = Zp / CH;
    const int nrings = ncells * MUBAR;
    double *sA = sm;                                     // [Zp][16]          A rows of the 16 nodes, zero beyond b_zDim / the tile
    double *sAz = sA + (size_t)Zp * RZ_T;                // [3][16][ZS]       vertically transformed node values of this workgroup's 64 levels
    double *sphi = sAz + (size_t)3 * RZ_T * ZS;          // [3][39][4]        basis weights of the 39 rings
    {
        const int nl = threadIdx.x >> 4, l16 = threadIdx.x & 15;
        const int node = min(c0 + nl, nbt - 1);
        const double nlive = c0 + nl < nbt ? 1.0 : 0.0;
        const double *a = Arows + (int64_t)node * C + (int64_t)v * Zb;
#pragma unroll 4
        for (int zm = l16; zm < Zp; zm += 16) sA[zm * RZ_T + nl] = (zm < Zb ? nlive : 0.0) * a[min(zm, Zb - 1)];
    }
    for (int e = threadIdx.x; e < 3 * RZ_CT * MUBAR * 4; e += blockDim.x) {
        const int j = e & 3, rl = (e >> 2) % (RZ_CT * MUBAR), d = (e >> 2) / (RZ_CT * MUBAR);
        const int ring = min(c0 * MUBAR + rl, nrings - 1);
        sphi[e] = phi[((int64_t)d * nrings + ring) * 4 + j];
    }
    __syncthreads();
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, n16 = lane & 15, kk = lane >> 4;
    const int nzt = (nz + 15) / 16;
    const int zt = blockIdx.z * 4 + wave;                // this wave's tile of 16 levels
    if (zt >= nzt) return;
    const int z = zt * 16 + n16;
    const bool zok = z < nz;
    const int zc = min(z, nz - 1);
    // operators this variable needs: sz = 0 (value, for u / r / rr), 1 (d/dz), 2 (d2/dz2)
    int szs[3], nb = 0;
    if (n_u || n_r || n_rr) szs[nb++] = 0;
    if (n_z) szs[nb++] = 1;
    if (n_zz) szs[nb++] = 2;
    for (int a = nb; a < 3; a++) szs[a] = szs[0];
    const double *const ops[3] = {MzT + ((int64_t)v * 3 + szs[0]) * Zb * nz + zc, MzT + ((int64_t)v * 3 + szs[1]) * Zb * nz + zc,
                                  MzT + ((int64_t)v * 3 + szs[2]) * Zb * nz + zc};
    rz_d4 acc[3];
    if (nb == 3) rz_ops_pass<3, KC>(sA, ops, nz, Zb, nchunks, lane, kk, acc);
    else if (nb == 2) rz_ops_pass<2, KC>(sA, ops, nz, Zb, nchunks, lane, kk, acc);
    else rz_ops_pass<1, KC>(sA, ops, nz, Zb, nchunks, lane, kk, acc);
    // D[m = node kk + 4 r][n = level]: into this wave's 16 columns of the Az tile
    for (int a = 0; a < nb; a++)
#pragma unroll
        for (int r = 0; r < 4; r++) sAz[(szs[a] * RZ_T + kk + 4 * r) * ZS + wave * 16 + n16] = acc[a][r];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    // the 39 rings of the 13 cells: lane = (ring sub-index kk, level n16); 16 consecutive levels of a ring = one 128-byte run
    const int zl = wave * 16 + n16;
    for (int rl = kk; rl < RZ_CT * MUBAR; rl += 4) {
        const int ring = c0 * MUBAR + rl;
        if (ring >= nrings) break;
        const int cl = rl / MUBAR;                       // local cell = first of its 4 nodes
        const int64_t pt = (int64_t)ring * nz + z;
        auto put = [&](int slot, double val) {
            if (!zok) return;
            if (slot == 0) phys.val[(int64_t)v * N + pt] = val;
            else phys.der[((int64_t)(slot - 1) * V + v) * N + pt] = (ST)val;
        };
        if (n_u || n_r || n_rr) {
            const double *g = sAz + (size_t)(0 * RZ_T + cl) * ZS + zl;
            const double g0 = g[0], g1 = g[ZS], g2 = g[2 * ZS], g3 = g[3 * ZS];
            const double *p0 = sphi + (size_t)rl * 4, *p1 = p0 + RZ_CT * MUBAR * 4, *p2 = p1 + RZ_CT * MUBAR * 4;
            if (n_u) put(s_u, p0[0] * g0 + p0[1] * g1 + p0[2] * g2 + p0[3] * g3);
            if (n_r) put(s_r, p1[0] * g0 + p1[1] * g1 + p1[2] * g2 + p1[3] * g3);
            if (n_rr) put(s_rr, p2[0] * g0 + p2[1] * g1 + p2[2] * g2 + p2[3] * g3);
        }
        if (n_z || n_zz) {
            const double *p0 = sphi + (size_t)rl * 4;
#pragma unroll
            for (int sz = 1; sz < 3; sz++) {
                if (!(sz == 1 ? n_z : n_zz)) continue;
                const double *g = sAz + (size_t)(sz * RZ_T + cl) * ZS + zl;
                put(sz == 1 ? s_z : s_zz, p0[0] * g[0] + p0[1] * g[ZS] + p0[2] * g[2 * ZS] + p0[3] * g[3 * ZS]);
            }
        }
    }
}

// Radial inner products of 16 nodes: the weights wq * phi0 of the 19 cells (57 rings) that touch them go to LDS first (zero for
// cells outside the tile), then a thread walks the 11 cells (33 rings) of its half of the nodes for ONE level - 33 loads requested
// together - and keeps the 8 node sums in registers.
__global__ void __launch_bounds__(256)
k_rz_forward(const double *__restrict__ np1, double *__restrict__ B /* tile rows [nbt][C] */, const double *__restrict__ phi /* [nrings][4] */,
             const double *__restrict__ wq, const double *__restrict__ CBT /* [nz][Zb] */, int ncells, int nbt, int V, int nz, int Zb,
             int64_t N, int64_t C, int mt_per_wg) {
    extern __shared__ double sm[];                      // [Np][16] radial inner products (Np = zDim rounded up to a whole chunk), then [57][4] weights
    const int node0 = blockIdx.x * RZ_T, v = blockIdx.y;
    constexpr int CH = 4 * RZ_KC;
    const int Np = (nz + CH - 1) / CH * CH, nchunks = Np / CH;
    double *wt = sm + (size_t)Np * RZ_T;
    const int nrings = ncells * MUBAR;
    for (int e = threadIdx.x; e < 19 * MUBAR * 4; e += blockDim.x) {
        const int jj = e & 3, rr = e >> 2;               // ring rr of the window: cell node0 - 3 + rr / 3
        const int c = node0 - 3 + rr / MUBAR, ring = c * MUBAR + rr % MUBAR;
        wt[e] = (c >= 0 && c < ncells) ? wq[ring] * phi[(int64_t)ring * 4 + jj] : 0.0;
    }
    __syncthreads();
    const double *x = np1 + (int64_t)v * N;
    for (int e = threadIdx.x; e < Np * 2; e += blockDim.x) {
        const int z = e % Np, half = e / Np;
        const int zc = min(z, nz - 1);
        const double live = z < nz ? 1.0 : 0.0;
        double xv[11 * MUBAR];
#pragma unroll
        for (int q = 0; q < 11 * MUBAR; q++) {
            const int ring = min(max((node0 + 8 * half - 3) * MUBAR + q, 0), nrings - 1);      // outside the tile: any finite value, its weights are zero
            xv[q] = x[(int64_t)ring * nz + zc];
        }
        double acc[8];
#pragma unroll
        for (int q = 0; q < 8; q++) acc[q] = 0.0;
#pragma unroll
        for (int cr = 0; cr < 11; cr++)
#pragma unroll
            for (int mu = 0; mu < MUBAR; mu++) {
                const double *w = wt + ((8 * half + cr) * MUBAR + mu) * 4;
                const double xx = live * xv[cr * MUBAR + mu];
#pragma unroll
                for (int jj = 0; jj < 4; jj++) {
                    const int q = cr - 3 + jj;           // node 8 half + q = cell (8 half - 3 + cr) + jj
                    if (q >= 0 && q < 8) acc[q] += w[jj] * xx;
                }
            }
#pragma unroll
        for (int q = 0; q < 8; q++) sm[z * RZ_T + 8 * half + q] = acc[q];
    }
    __syncthreads();
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, n16 = lane & 15, kk = lane >> 4;
    const int nmt = (Zb + 15) / 16;
    for (int t = wave; t < mt_per_wg; t += 4) {
        const int mt = blockIdx.z * mt_per_wg + t;       // tile of 16 Chebyshev modes
        if (mt >= nmt) break;
        const int zm = mt * 16 + n16;
        const bool mok = zm < Zb;
        const double *op = CBT + min(zm, Zb - 1);
        rz_d4 acc = {0.0, 0.0, 0.0, 0.0};
        for (int c = 0; c < nchunks; c++) {
            double b[RZ_KC];
#pragma unroll
            for (int q = 0; q < RZ_KC; q++) b[q] = op[(int64_t)min(4 * (c * RZ_KC + q) + kk, nz - 1) * Zb];   // B[k = level][n = mode]; rows beyond zDim meet zeros
#pragma unroll
            for (int q = 0; q < RZ_KC; q++)
                acc = __builtin_amdgcn_mfma_f64_16x16x4f64(sm[(4 * (c * RZ_KC + q)) * RZ_T + lane], b[q], acc, 0, 0, 0);   // A[m = node][k = level]
        }
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const int node = node0 + kk + 4 * r;
            if (node < nbt && mok) B[(int64_t)node * C + (int64_t)v * Zb + zm] = acc[r];
        }
    }
}

// ------------------------------------------------------------------------------------------------ semi-implicit adjustment
// semiimplicit_adjustment (src/semiimplicit.jl:521-597) for 16 columns per workgroup, its four column operators as matrix-core
// products (m = column, n = level, k = level; operators transposed in memory, so a fragment is a 128-byte run):
//   xi*, w*  = the explicit step minus its implicit part, plus the AI2* off-centring     (:544-558, element-wise, staged into LDS)
//   xrec     = Mrec xi*,  xz = Mdz xi*;   g = [0; 0; (tau Pxi xz - w*)[2 : nz-1]]        (:569-584)
//   w        = W g  (= T H^-1 g: the Helmholtz solve folded with the collocation matrix, build_helmholtz),  wz = X g   (:586-592)
//   xi       = xrec - tau wz                                                             (:593-596)
// The scalar kernel k_semiimplicit ran these as 4 x zDim dependent multiply-adds per thread: 28 us for 513 columns x 128 levels.
template <int NB>        // operators sharing one LDS operand
__device__ __forceinline__ void semi_pass(const double *__restrict__ sa, const double *const (&op)[2], int64_t nz, int nchunks, int lane, int kk,
                                          rz_d4 (&acc)[2]) {
#pragma unroll
    for (int a = 0; a < NB; a++) acc[a] = rz_d4{0.0, 0.0, 0.0, 0.0};
    for (int c = 0; c < nchunks; c++) {
        double b[NB][RZ_KC];
#pragma unroll
        for (int q = 0; q < RZ_KC; q++)
#pragma unroll
            for (int a = 0; a < NB; a++) b[a][q] = op[a][(int64_t)min(4 * (c * RZ_KC + q) + kk, (int)nz - 1) * nz];      // rows beyond zDim meet zeros
#pragma unroll
        for (int q = 0; q < RZ_KC; q++) {
            const double av = sa[(4 * (c * RZ_KC + q)) * RZ_T + lane];
#pragma unroll
            for (int a = 0; a < NB; a++) acc[a] = __builtin_amdgcn_mfma_f64_16x16x4f64(av, b[a][q], acc[a], 0, 0, 0);
        }
    }
}

__global__ void __launch_bounds__(512)
k_semi_mfma(SemiArgs a, int64_t ncol) {
    extern __shared__ double sm[];                      // four [Kp][16] column tiles: w*, xi*, g, xrec
    const int nz = a.nz;
    constexpr int CH = 4 * RZ_KC;
    const int Kp = (nz + CH - 1) / CH * CH, nchunks = Kp / CH;
    double *sw = sm, *sx_ = sm + (size_t)Kp * RZ_T, *sg = sm + (size_t)2 * Kp * RZ_T, *srec = sm + (size_t)3 * Kp * RZ_T;
    const int64_t col0 = (int64_t)blockIdx.x * RZ_T;
    const double ts = a.ts;
    for (int e = threadIdx.x; e < Kp * RZ_T; e += blockDim.x) {
        const int k = e % Kp, cl = e / Kp;               // levels fastest across lanes: coalesced column reads
        double ow = 0.0, ox = 0.0;
        if (k < nz && col0 + cl < ncol) {
            const int64_t p = (col0 + cl) * nz + k;
            double out[2];
#pragma unroll
            for (int q = 0; q < 2; q++) {
                const int64_t o = (int64_t)(q == 0 ? a.wi : a.xi) * a.N + p;
                double x = a.np1[o];
                const double In = a.In[o];
                if (a.t == 1) x = x - (ts * In) + (ts * 0.5 * In);
                else if (a.t == 2) x = x - (0.5 * ts) * ((3.0 * In) - a.I1[o]) - (ts * In) + (ts * 0.75 * a.I1[o]);
                else x = x - ((ts / 12.0) * ((23.0 * In) - (16.0 * a.I1[o]) + (5.0 * a.I2[o]))) - (ts * In) + (ts * 0.75 * a.I1[o]);
                out[q] = x;
            }
            ow = out[0]; ox = out[1];
        }
        sw[k * RZ_T + cl] = ow;
        sx_[k * RZ_T + cl] = ox;
        sg[k * RZ_T + cl] = 0.0;                          // rows 0, 1 and the padding stay zero
    }
    __syncthreads();
    const int wave = threadIdx.x >> 6, nw = blockDim.x >> 6, lane = threadIdx.x & 63, n16 = lane & 15, kk = lane >> 4;
    const int nzt = (nz + 15) / 16;
    rz_d4 acc[2];
    for (int kt = wave; kt < nzt; kt += nw) {
        const int k = kt * 16 + n16, kc = min(k, nz - 1);
        const double *const ops[2] = {a.MrecT + kc, a.MdzT + kc};
        semi_pass<2>(sx_, ops, nz, nchunks, lane, kk, acc);
        // D[m = column kk + 4 r][n = level k]
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const int cl = kk + 4 * r;
            if (k < nz) {
                srec[k * RZ_T + cl] = acc[0][r];
                if (k >= 1 && k < nz - 1) sg[(k + 1) * RZ_T + cl] = (a.tau * a.pxi * acc[1][r]) - sw[k * RZ_T + cl];
            }
        }
    }
    __syncthreads();
    for (int kt = wave; kt < nzt; kt += nw) {
        const int k = kt * 16 + n16, kc = min(k, nz - 1);
        const double *const ops[2] = {a.WT + kc, a.XT + kc};
        semi_pass<2>(sg, ops, nz, nchunks, lane, kk, acc);
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const int cl = kk + 4 * r;
            if (k < nz && col0 + cl < ncol) {
                const int64_t p = (col0 + cl) * nz + k;
                a.np1[(int64_t)a.wi * a.N + p] = acc[0][r];
                a.np1[(int64_t)a.xi * a.N + p] = srec[k * RZ_T + cl] - (a.tau * acc[1][r]);
            }
        }
    }
}

void launch_semi_mfma(sx_handle *h, const SemiArgs &a) {
    const int Kp = (h->nz + 4 * RZ_KC - 1) / (4 * RZ_KC) * (4 * RZ_KC);
    const size_t lds = sizeof(double) * 4 * Kp * RZ_T;
    if (lds > 65536) HIPCHK4(hipFuncSetAttribute(reinterpret_cast<const void *>(k_semi_mfma), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    const int nzt = (h->nz + 15) / 16;
    const int threads = 64 * std::min(8, nzt);
    hipLaunchKernelGGL(k_semi_mfma, dim3((unsigned)((h->Nh + RZ_T - 1) / RZ_T)), dim3(threads), lds, h->stream, a, h->Nh);
}

bool rz_fused(const sx_handle *h) { return h->rz_fused && h->geom == SX_GEOM_RZ && !h->sp32; }

void launch_rz_inverse(sx_handle *h, const int *d_mask) {
    const int id = timer_id(h, "k_rz_inverse");
    timer_begin(h, id);
    static const bool by_nodes = !(getenv("SX_RZ_INV") && atoi(getenv("SX_RZ_INV")) == 0);      // A/B: 0 = the ring-tile form
    if (by_nodes) {
        const int Zpn = (h->Zb + 31) / 32 * 32;
        dim3 gn((h->ncells + RZ_CT - 1) / RZ_CT, h->V, ((h->nz + 15) / 16 + 3) / 4);
        const size_t ldsn = sizeof(double) * ((size_t)Zpn * RZ_T + 3 * RZ_T * 65 + 3 * RZ_CT * MUBAR * 4);
        const double *arows = h->d_A + (int64_t)h->cell0 * h->C;
#define RZ_INVN(ST) hipLaunchKernelGGL(k_rz_inverse_nodes<ST>, gn, dim3(256), ldsn, h->stream, arows, planes_of<ST>(h->d_phys, h->V, h->N), h->d_phi, h->d_MzT, \
                                       d_mask, h->V, h->nz, h->Zb, h->ncells, h->nbt, h->N, h->C, h->slot[0], h->slot[1], h->slot[2], h->slot[5], h->slot[6])
        if (ldsn <= 65536) {
            if (h->f32) RZ_INVN(float); else RZ_INVN(double);
            HIPCHK4(hipGetLastError());
            timer_end(h);
            return;
        }
#undef RZ_INVN
    }
    const int Zp = (h->Zb + 4 * RZ_KC - 1) / (4 * RZ_KC) * (4 * RZ_KC);
    dim3 g((h->nrings + RZ_T - 1) / RZ_T, h->V, ((h->nz + 15) / 16 + 3) / 4);
    const size_t lds = sizeof(double) * 3 * Zp * RZ_T;
    if (lds > 65536) {
        HIPCHK4(hipFuncSetAttribute(reinterpret_cast<const void *>(k_rz_inverse<double>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        HIPCHK4(hipFuncSetAttribute(reinterpret_cast<const void *>(k_rz_inverse<float>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    }
    const double *arows = h->d_A + (int64_t)h->cell0 * h->C;
#define RZ_INV(ST) hipLaunchKernelGGL(k_rz_inverse<ST>, g, dim3(256), lds, h->stream, arows, planes_of<ST>(h->d_phys, h->V, h->N), h->d_phi, h->d_MzT, \
                                      d_mask, h->V, h->nz, h->Zb, h->nrings, h->N, h->C, h->slot[0], h->slot[1], h->slot[2], h->slot[5], h->slot[6])
    if (h->f32) RZ_INV(float); else RZ_INV(double);
#undef RZ_INV
    HIPCHK4(hipGetLastError());
    timer_end(h);
}

void launch_rz_forward(sx_handle *h) {
    const int id = timer_id(h, "k_rz_forward");
    timer_begin(h, id);
    const int Np = (h->nz + 4 * RZ_KC - 1) / (4 * RZ_KC) * (4 * RZ_KC), nmt = (h->Zb + 15) / 16;
    const int mt_per_wg = std::min(nmt, 4);              // one 16-mode tile per wave; more workgroups beat fewer restagings at this size
    dim3 g((h->nbt + RZ_T - 1) / RZ_T, h->v_cnt, (nmt + mt_per_wg - 1) / mt_per_wg);
    const size_t lds = sizeof(double) * (Np * RZ_T + 19 * MUBAR * 4);
    hipLaunchKernelGGL(k_rz_forward, g, dim3(256), lds, h->stream, h->d_np1 + (int64_t)h->v_lo * h->N, h->d_Btile + (int64_t)h->v_lo * h->Zb,
                       h->d_phi, h->d_wq, h->d_CBT, h->ncells, h->nbt, h->V, h->nz, h->Zb, h->N, h->C, mt_per_wg);
    HIPCHK4(hipGetLastError());
    timer_end(h);
}

}  // namespace sx
