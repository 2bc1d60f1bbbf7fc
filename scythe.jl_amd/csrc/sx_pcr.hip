// B -> A banded SPD solve (splineTransform!, src/semiimplicit.jl:237, 285) as an LDS-staged PARALLEL CYCLIC REDUCTION, for launches
// with few right-hand sides: the R grid (one column), RZ grids (V x b_zDim columns), small RL patches, a rank's share of the
// transposed multi-GPU solve.  There the lane-per-column kernel (k_solve, sx_kernels.hip) is one wave's serial recurrence over
// 2 x (num_cells + 3) rows - 50-80 us whatever the size; here a right-hand side's rows are worked on side by side:
//   workgroup = R columns of one boundary-condition class; thread = (block of 3 unknowns i, column c)
//   1. rhs = Gamma b          gathered straight from the B rows into LDS [unknown][column]
//   2. `levels` (<= 6) reduction levels  r_i <- r_i - alpha r_{i-s} - gamma r_{i+s},  s = 1, 2, 4, ...: two 3 x 3 products per
//      thread and level on neighbours read from LDS, ONE workgroup barrier per level (double buffer), the elimination blocks of
//      the next level requested before the barrier.  The blocks belong to the constant matrix Gamma (P + eps_q Q) Gamma^T and
//      are worked out once, in extended precision, by build_pcr_tables (sx_setup.cpp): the kernel only applies them.
//      The couplings fall off as rho^(2^l) with rho ~ 0.02 per block, so the system is block diagonal to < 1e-34 after
//      5-6 levels whatever the patch size (exactly after ceil(log2(blocks)) levels).
//   3. x_i = D_i^-1 r_i;  PERIODIC classes add the rank-6 corner correction x -= G (E^T x)
//   4. a = Gamma^T x          written to the A rows (and, transposed solve, to the second tile that shares the row)
// Same interface as k_solve: row m of the right-hand side is Bsrc[boffA[m] + col] (+ Bsrc[boffB[m] + col]), row m of the solution
// goes to A[aoffA[m] + col] (and A[aoffB[m] + col]); LINEAR: m * stride + col on both sides.
// Results agree with the Cholesky solve to rounding (tests/test_pcr_tables.py on the host, tests/test_gpu_parity.py on the device).
#include "sx_internal.hpp"
#include <cstdlib>
#include <map>

namespace sx {

#define HIPCHK3(x)                                                                                  \
    do {                                                                                            \
        hipError_t e_ = (x);                                                                        \
        if (e_ != hipSuccess) set_error(std::string(#x) + ": " + hipGetErrorString(e_) + " (" __FILE__ ":" + std::to_string(__LINE__) + ")"); \
    } while (0)

struct PcrClassDev {
    int n, nblk, levels, periodic;
    const double *coef, *dinv, *gin_w, *gout_w, *G;
    const int *gin_row, *gout_j;
};

struct PcrSeg {          // columns col0 + c * cstride, c < ncols (<= R), all of boundary-condition class cls
    int64_t col0;
    int ncols, cstride, cls, pad;
};

constexpr int PCR_IPT = 4;     // row items (patch row, column) per thread while loading B and storing A: nb * R <= 4 * blockDim (launcher)

template <bool LINEAR>
__global__ void __launch_bounds__(1024)
k_solve_pcr(const double *__restrict__ Bsrc, const int64_t *__restrict__ boffA, const int64_t *__restrict__ boffB,
            double *__restrict__ A, const int64_t *__restrict__ aoffA, const int64_t *__restrict__ aoffB,
            const PcrClassDev *__restrict__ classes, const PcrSeg *__restrict__ segs, int nb, int logR, int64_t stride) {
    extern __shared__ double sm[];
    const PcrSeg sg = segs[blockIdx.x];
    const PcrClassDev cd = classes[sg.cls];
    const int R = 1 << logR, np = 3 * cd.nblk, rows = max(np, nb);
    double *cur = sm, *nxt = sm + (size_t)rows * R, *ye = sm + (size_t)2 * rows * R;
    const int tid = threadIdx.x, nt = blockDim.x;
    // this thread's reduction item: block row bi, column ci (at most one per thread: the launcher keeps nblk * R <= blockDim)
    const int ci = tid & (R - 1), bi = tid >> logR;
    const bool item = bi < cd.nblk;
    // ---- everything this thread will need from memory is requested NOW, in one burst, nothing depends on anything else:
    // its <= 4 right-hand-side values, the Gamma tables of its unknowns / of the rows it will store, level 0's elimination blocks
    double braw[PCR_IPT];
#pragma unroll
    for (int u = 0; u < PCR_IPT; u++) {
        const int e = tid + u * nt, c = e & (R - 1), m = min(e >> logR, nb - 1);
        const int64_t col = sg.col0 + (int64_t)min(c, sg.ncols - 1) * sg.cstride;
        if (LINEAR) braw[u] = Bsrc[(int64_t)m * stride + col];
        else { const int64_t o1 = boffA[m], o2 = boffB[m]; braw[u] = Bsrc[o1 + col] + (o2 >= 0 ? Bsrc[o2 + col] : 0.0); }
    }
    int gi[3][4], go[PCR_IPT][2];
    double gw[3][4], gow[PCR_IPT][2];
#pragma unroll
    for (int p = 0; p < 3; p++)
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const int j = min(3 * bi + p, np - 1);
            gi[p][q] = cd.gin_row[j * 4 + q]; gw[p][q] = cd.gin_w[j * 4 + q];
        }
#pragma unroll
    for (int u = 0; u < PCR_IPT; u++) {
        const int m = min((tid + u * nt) >> logR, nb - 1);
#pragma unroll
        for (int q = 0; q < 2; q++) { go[u][q] = cd.gout_j[m * 2 + q]; gow[u][q] = cd.gout_w[m * 2 + q]; }
    }
    double cf[18];
    if (item) {
        const double *cp = cd.levels > 0 ? cd.coef + (size_t)bi * 18 : cd.dinv + (size_t)bi * 9;
        const int ncf = cd.levels > 0 ? 18 : 9;
#pragma unroll
        for (int q = 0; q < 18; q++) if (q < ncf) cf[q] = cp[q];
    }
    // ---- 1. the B rows to LDS as they are ([row][column]), then rhs = Gamma b gathered from there
#pragma unroll
    for (int u = 0; u < PCR_IPT; u++) {
        const int e = tid + u * nt;
        if (e < nb * R) nxt[e] = braw[u];
    }
    __syncthreads();
    if (item) {
#pragma unroll
        for (int p = 0; p < 3; p++) {
            double s = 0.0;
#pragma unroll
            for (int q = 0; q < 4; q++) if (gi[p][q] >= 0) s += gw[p][q] * nxt[gi[p][q] * R + ci];
            cur[(3 * bi + p) * R + ci] = s;
        }
    }
    __syncthreads();
    // ---- 2. reduction levels
    for (int l = 0; l < cd.levels; l++) {
        const int s = 1 << l;
        if (item) {
            double v0 = cur[(3 * bi) * R + ci], v1 = cur[(3 * bi + 1) * R + ci], v2 = cur[(3 * bi + 2) * R + ci];
            if (bi - s >= 0) {
                const double *lo = cur + (size_t)(3 * (bi - s)) * R + ci;
                const double a0 = lo[0], a1 = lo[R], a2 = lo[2 * R];
                v0 -= cf[0] * a0 + cf[1] * a1 + cf[2] * a2;
                v1 -= cf[3] * a0 + cf[4] * a1 + cf[5] * a2;
                v2 -= cf[6] * a0 + cf[7] * a1 + cf[8] * a2;
            }
            if (bi + s < cd.nblk) {
                const double *hi = cur + (size_t)(3 * (bi + s)) * R + ci;
                const double a0 = hi[0], a1 = hi[R], a2 = hi[2 * R];
                v0 -= cf[9] * a0 + cf[10] * a1 + cf[11] * a2;
                v1 -= cf[12] * a0 + cf[13] * a1 + cf[14] * a2;
                v2 -= cf[15] * a0 + cf[16] * a1 + cf[17] * a2;
            }
            nxt[(3 * bi) * R + ci] = v0; nxt[(3 * bi + 1) * R + ci] = v1; nxt[(3 * bi + 2) * R + ci] = v2;
            // the next level's blocks (or the final diagonal inverses) are requested before the barrier
            const double *cp = (l + 1 < cd.levels) ? cd.coef + ((size_t)(l + 1) * cd.nblk + bi) * 18 : cd.dinv + (size_t)bi * 9;
            const int ncf = (l + 1 < cd.levels) ? 18 : 9;
#pragma unroll
            for (int q = 0; q < 18; q++) if (q < ncf) cf[q] = cp[q];
        }
        __syncthreads();
        double *t = cur; cur = nxt; nxt = t;
    }
    // ---- 3. x_i = D_i^-1 r_i
    if (item) {
        const double a0 = cur[(3 * bi) * R + ci], a1 = cur[(3 * bi + 1) * R + ci], a2 = cur[(3 * bi + 2) * R + ci];
        nxt[(3 * bi) * R + ci] = cf[0] * a0 + cf[1] * a1 + cf[2] * a2;
        nxt[(3 * bi + 1) * R + ci] = cf[3] * a0 + cf[4] * a1 + cf[5] * a2;
        nxt[(3 * bi + 2) * R + ci] = cf[6] * a0 + cf[7] * a1 + cf[8] * a2;
    }
    __syncthreads();
    double *x = nxt;
    if (cd.periodic) {
        // corner blocks of the cyclic matrix: x -= G (E^T x) over the edge unknowns 0, 1, 2, n-3, n-2, n-1
        if (tid < 6 * R) {
            const int c = tid & (R - 1), q = tid >> logR;
            ye[q * R + c] = x[(q < 3 ? q : cd.n - 6 + q) * R + c];
        }
        __syncthreads();
        for (int e = tid; e < cd.n * R; e += nt) {
            const int c = e & (R - 1), i = e >> logR;
            const double *g = cd.G + (size_t)i * 6;
            double v = x[e];
#pragma unroll
            for (int q = 0; q < 6; q++) v -= g[q] * ye[q * R + c];
            cur[e] = v;
        }
        __syncthreads();
        x = cur;
    }
    // ---- 4. a = Gamma^T x
#pragma unroll
    for (int u = 0; u < PCR_IPT; u++) {
        const int e = tid + u * nt, c = e & (R - 1), m = e >> logR;
        if (m >= nb || c >= sg.ncols) continue;
        const int64_t col = sg.col0 + (int64_t)c * sg.cstride;
        double s = 0.0;
#pragma unroll
        for (int q = 0; q < 2; q++) if (go[u][q] >= 0) s += gow[u][q] * x[go[u][q] * R + c];
        if (LINEAR) A[(int64_t)m * stride + col] = s;
        else { A[aoffA[m] + col] = s; const int64_t o2 = aoffB[m]; if (o2 >= 0) A[o2 + col] = s; }
    }
}

// ---- host side ---------------------------------------------------------------------------------------------------------
struct PcrLaunch {
    PcrSeg *d_segs = nullptr;
    int nsegs = 0, logR = 0, threads = 0;
    size_t lds = 0;
};

struct PcrState {
    bool ok = false;                 // tables built and uploaded
    int nblk_max = 0, np_max = 0;
    PcrClassDev *d_classes = nullptr;
    std::vector<void *> allocs;
    std::map<std::pair<int, int>, PcrLaunch> launches;      // (first patch-level group, groups) -> segment list
};

template <class T>
static bool pcr_upload(PcrState *st, const T **dst, const std::vector<T> &v) {
    void *d = nullptr;
    if (hipMalloc(&d, std::max<size_t>(v.size(), 1) * sizeof(T)) != hipSuccess) return false;
    st->allocs.push_back(d);
    if (!v.empty() && hipMemcpy(d, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice) != hipSuccess) return false;
    *dst = (const T *)d;
    return true;
}

static PcrState *pcr_state(sx_handle *h) {
    if (h->pcr_state) return (PcrState *)h->pcr_state;
    PcrState *st = new PcrState();
    h->pcr_state = st;
    std::vector<PcrClassDev> cls(h->classes.size());
    bool good = true;
    for (size_t c = 0; c < h->classes.size() && good; c++) {
        PcrTables t;
        std::string err;
        if (!build_pcr_tables(h->classes[c], h->b_rDim, t, err)) { good = false; break; }
        PcrClassDev &d = cls[c];
        d.n = t.n; d.nblk = t.nblk; d.levels = t.levels; d.periodic = t.periodic;
        good = pcr_upload(st, &d.coef, t.coef) && pcr_upload(st, &d.dinv, t.dinv) && pcr_upload(st, &d.gin_w, t.gin_w) &&
               pcr_upload(st, &d.gout_w, t.gout_w) && pcr_upload(st, &d.G, t.G) && pcr_upload(st, &d.gin_row, t.gin_row) &&
               pcr_upload(st, &d.gout_j, t.gout_j);
        st->nblk_max = std::max(st->nblk_max, t.nblk);
    }
    st->np_max = 3 * st->nblk_max;
    if (good) {
        const PcrClassDev *dc = nullptr;
        good = pcr_upload(st, &dc, cls);
        st->d_classes = const_cast<PcrClassDev *>(dc);
    }
    st->ok = good && st->nblk_max <= 1024;
    return st;
}

void pcr_release(sx_handle *h) {
    PcrState *st = (PcrState *)h->pcr_state;
    if (!st) return;
    for (void *p : st->allocs) hipFree(p);
    delete st;
    h->pcr_state = nullptr;
}

// Should this launch (ncols right-hand sides in all) take the parallel-cyclic-reduction kernel?  SX_SOLVE_PCR=0 / 1 (read at sx_create)
// forces the answer (1: wherever the tables exist); by default launches of up to SX_PCR_MAXCOLS (16384) columns do - above that the
// lane-per-column kernel fills the chip and streams its rows at the HBM rate.
bool pcr_wanted(sx_handle *h, int64_t ncols) {
    if (h->solve_pcr == 0) return false;
    if (h->solve_pcr != 1 && ncols > h->pcr_maxcols) return false;
    return pcr_state(h)->ok;
}

// vz0: patch-level index of the launch's first (variable, z-mode) group, ng groups; columns are numbered from the launch's first
// column (group g local, block blk -> g * K2 + blk), as in k_solve
void launch_solve_pcr(sx_handle *h, bool linear, const double *Bsrc, const int64_t *boffA, const int64_t *boffB, double *A,
                      const int64_t *aoffA, const int64_t *aoffB, int vz0, int ng, int64_t stride) {
    PcrState *st = pcr_state(h);
    if (!st->ok || ng <= 0) return;
    auto key = std::make_pair(vz0, ng);
    auto it = st->launches.find(key);
    if (it == st->launches.end()) {
        PcrLaunch pl;
        const int64_t total = (int64_t)ng * (h->K2 > 1 ? h->K2 - 1 : 1);
        static const int r_env = getenv("SX_PCR_R") ? atoi(getenv("SX_PCR_R")) : 0;
        int R = r_env > 0 ? r_env : total <= 4096 ? 4 : total <= 16384 ? 8 : 16;
        while (R > 1 && (int64_t)st->nblk_max * R > 1024) R >>= 1;
        int logR = 0;
        while ((1 << (logR + 1)) <= R) logR++;
        R = 1 << logR;
        std::vector<PcrSeg> segs;
        auto add = [&](int64_t col0, int64_t count, int cstride, int cls) {
            for (int64_t c = 0; c < count; c += R)
                segs.push_back(PcrSeg{col0 + c * cstride, (int)std::min<int64_t>(R, count - c), cstride, cls, 0});
        };
        // k = 0 columns (block 0 of every group; the only column when there is no azimuth): the groups of one variable are
        // consecutive and share a class -> one run with column stride K2
        for (int g = 0; g < ng;) {
            const int v = (vz0 + g) / h->Zb;
            int g1 = g;
            while (g1 < ng && (vz0 + g1) / h->Zb == v) g1++;
            add((int64_t)g * h->K2, g1 - g, h->K2, h->hcls[(size_t)v * 2 + 0]);
            g = g1;
        }
        // wavenumbers k >= 1: blocks 2 .. K2 - 1 of every group
        if (h->K2 > 2)
            for (int g = 0; g < ng; g++) add((int64_t)g * h->K2 + 2, h->K2 - 2, 1, h->hcls[(size_t)((vz0 + g) / h->Zb) * 2 + 1]);
        pl.nsegs = (int)segs.size();
        pl.logR = logR;
        pl.threads = std::min(1024, ((std::max(st->nblk_max * R, (h->b_rDim * R + PCR_IPT - 1) / PCR_IPT) + 63) / 64) * 64);
        pl.lds = sizeof(double) * ((size_t)2 * std::max(st->np_max, h->b_rDim) * R + 6 * R);
        void *d = nullptr;
        if (hipMalloc(&d, sizeof(PcrSeg) * segs.size()) != hipSuccess ||
            hipMemcpy(d, segs.data(), sizeof(PcrSeg) * segs.size(), hipMemcpyHostToDevice) != hipSuccess) {
            set_error("launch_solve_pcr: segment table upload failed");
            return;
        }
        st->allocs.push_back(d);
        pl.d_segs = (PcrSeg *)d;
        it = st->launches.emplace(key, pl).first;
    }
    const PcrLaunch &pl = it->second;
    if (linear)
        hipLaunchKernelGGL(k_solve_pcr<true>, dim3(pl.nsegs), dim3(pl.threads), pl.lds, h->stream, Bsrc, boffA, boffB, A, aoffA, aoffB,
                           st->d_classes, pl.d_segs, h->b_rDim, pl.logR, stride);
    else
        hipLaunchKernelGGL(k_solve_pcr<false>, dim3(pl.nsegs), dim3(pl.threads), pl.lds, h->stream, Bsrc, boffA, boffB, A, aoffA, aoffB,
                           st->d_classes, pl.d_segs, h->b_rDim, pl.logR, stride);
    HIPCHK3(hipGetLastError());
}

}  // namespace sx
