// C ABI of libscythe_hip.so (declared in include/scythe_hip.h).
#include "sx_internal.hpp"
#include <algorithm>
#include <array>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <map>
#include <type_traits>

namespace sx {

static thread_local std::string g_err;
static thread_local bool g_err_set = false;

void set_error(const std::string &msg) {
    if (!g_err_set) g_err = msg;
    g_err_set = true;
}
void clear_error() { g_err_set = false; }
static int status() { return g_err_set ? 1 : 0; }
int error_status() { return status(); }

#define HIPOK(x)                                                                          \
    do {                                                                                  \
        hipError_t e_ = (x);                                                              \
        if (e_ != hipSuccess) {                                                           \
            set_error(std::string(#x) + ": " + hipGetErrorString(e_));                    \
            return 1;                                                                     \
        }                                                                                 \
    } while (0)

// doubles to allocate for `count` elements of the storage type (fp64, or fp32 when storage_f32 is set)
// for a [D][V][n] plane array whose derivative slots (1..D-1) are fp32 when storage_f32 is set (value slot stays fp64)
static size_t plane_count(const sx_handle *h, size_t n) {
    const size_t val = (size_t)h->V * n, der = (size_t)(h->D - 1) * h->V * n;
    return val + (h->f32 ? (der + 1) / 2 : der);
}

template <class T>
static bool dalloc(sx_handle *h, T **p, size_t count, bool zero = true) {
    void *d = nullptr;
    size_t bytes = std::max<size_t>(count, 1) * sizeof(T);
    hipError_t e = hipMalloc(&d, bytes);
    if (e != hipSuccess) {
        set_error(std::string("hipMalloc failed: ") + hipGetErrorString(e));
        return false;
    }
    if (zero && hipMemset(d, 0, bytes) != hipSuccess) {
        set_error("hipMemset failed");
        return false;
    }
    h->allocs.push_back(d);
    h->dev_bytes += bytes;
    *p = (T *)d;
    return true;
}

template <class T>
static bool upload(sx_handle *h, T **p, const std::vector<T> &v) {
    if (!dalloc(h, p, v.size(), false)) return false;
    if (!v.empty() && hipMemcpy(*p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice) != hipSuccess) {
        set_error("hipMemcpy H2D failed");
        return false;
    }
    return true;
}

static std::vector<double> transpose(const std::vector<double> &m, int n) {
    std::vector<double> t((size_t)n * n);
    for (int i = 0; i < n; i++)
        for (int j = 0; j < n; j++) t[(size_t)j * n + i] = m[(size_t)i * n + j];
    return t;
}

// ---- timers
int timer_id(sx_handle *h, const char *name) {
    for (size_t i = 0; i < h->timers.size(); i++)
        if (h->timers[i].name == name || !std::strcmp(h->timers[i].name, name)) return (int)i;
    Timer t;
    t.name = name;
    h->timers.push_back(t);
    return (int)h->timers.size() - 1;
}

static hipEvent_t get_event(sx_handle *h) {
    if (!h->event_pool.empty()) {
        hipEvent_t e = h->event_pool.back();
        h->event_pool.pop_back();
        return e;
    }
    hipEvent_t e;
    if (hipEventCreate(&e) != hipSuccess) set_error("hipEventCreate failed");
    return e;
}

void timer_begin(sx_handle *h, int id) {
    h->timer_skip = false;
    if (!h->timers_on) return;
    if (!h->timer_only.empty() && h->timer_only != h->timers[id].name) { h->timer_skip = true; return; }
    if (h->pending.size() >= 8192) timers_flush(h);
    PendingEvent p;
    p.timer = id;
    p.a = get_event(h);
    p.b = get_event(h);
    hipEventRecord(p.a, h->stream);
    h->pending.push_back(p);
}

void timer_end(sx_handle *h) {
    if (!h->timers_on || h->timer_skip || h->pending.empty()) return;
    hipEventRecord(h->pending.back().b, h->stream);
}

void timers_flush(sx_handle *h) {
    for (auto &p : h->pending) {
        float ms = 0.f;
        hipEventSynchronize(p.b);
        if (hipEventElapsedTime(&ms, p.a, p.b) == hipSuccess) {
            h->timers[p.timer].ms += ms;
            h->timers[p.timer].calls += 1;
        }
        h->event_pool.push_back(p.a);
        h->event_pool.push_back(p.b);
    }
    h->pending.clear();
}

static const int DERIV_SLOTS[4][7] = {
    /* u r rr l ll z zz */
    {0, 1, 2, -1, -1, -1, -1},   // R
    {0, 1, 2, -1, -1, 3, 4},     // RZ
    {0, 1, 2, 3, 4, -1, -1},     // RL
    {0, 1, 2, 3, 4, 5, 6},       // RLZ
};

static int default_bzdim(int zDim) {
    int b = (int)std::floor((2.0 * zDim - 1.0) / 3.0) + 1;
    return std::min(zDim, b);
}

// reference layout index of (zm, blk, node) inside one variable's spectral column
static inline int64_t ref_index(int zm, int blk, int node, int K2, int nb) { return ((int64_t)zm * K2 + blk) * nb + node; }
// device block index of reference block b (0: k = 0, 2k-1: Re k, 2k: Im k)
static inline int dev_blk(int b) { return b == 0 ? 0 : b + 1; }

}  // namespace sx

using namespace sx;

extern "C" {

const char *sx_last_error(void) { return g_err.c_str(); }
int sx_abi_version(void) { return SX_ABI_VERSION; }

int sx_equation_set_id(const char *name) {
    static const std::map<std::string, int> ids = {
        {"LinearAdvection1D", SX_EQ_LINEAR_ADVECTION_1D},
        {"LinearAdvectionRZ", SX_EQ_LINEAR_ADVECTION_RZ},
        {"LinearAdvectionRL", SX_EQ_LINEAR_ADVECTION_RL},
        {"LinearAdvectionRLZ", SX_EQ_LINEAR_ADVECTION_RLZ},
        {"Oneway_ShallowWater_Slab", SX_EQ_ONEWAY_SW_SLAB},
        {"Twoway_ShallowWater_Slab", SX_EQ_TWOWAY_SW_SLAB},
        {"Oneway_ShallowWater_HeightResolvedBL", SX_EQ_ONEWAY_SW_HRBL},
        {"LinearAcousticRZ", SX_EQ_LINEAR_ACOUSTIC_RZ},
        {"Euler_test", SX_EQ_EULER_TEST},
        {"None", SX_EQ_NONE},
    };
    if (!name) return -1;
    auto it = ids.find(name);
    return it == ids.end() ? -1 : it->second;
}

static int tile_sizes(const sx_grid_desc *g, int n, std::vector<int> &cells) {
    // calcTileSizes: R / RZ split the cells evenly, RL / RLZ balance gridpoints (SURVEY.md 8(c) "Layouts").
    const int nc = g->num_cells;
    if (n < 1 || nc < 3 * n) {
        set_error("calcTileSizes: need at least 3 cells per tile");
        return 1;
    }
    const bool has_l = (g->geometry == SX_GEOM_RL || g->geometry == SX_GEOM_RLZ);
    cells.assign(n, 0);
    if (!has_l || g->ring_uniform_L > 0) {
        for (int t = 0; t < n; t++) cells[t] = nc / n + (t < nc % n ? 1 : 0);
        return 0;
    }
    std::vector<double> cum(nc + 1, 0.0);
    for (int c = 0; c < nc; c++) {
        double pts = 0;
        for (int mu = 0; mu < MUBAR; mu++) pts += 4.0 + 4.0 * (c * MUBAR + mu + 1);
        cum[c + 1] = cum[c] + pts;
    }
    int c0 = 0;
    for (int t = 0; t < n; t++) {
        int c1;
        if (t == n - 1) {
            c1 = nc;
        } else {
            const double target = cum[nc] * (t + 1) / n;
            c1 = c0 + 3;
            while (c1 < nc - 3 * (n - 1 - t) && cum[c1] < target) c1++;
            if (c1 > c0 + 3 && (cum[c1] - target) > (target - cum[c1 - 1])) c1--;
        }
        cells[t] = c1 - c0;
        c0 = c1;
    }
    return 0;
}

int sx_calc_tile_sizes(const sx_grid_desc *g, int32_t n, double *out) {
    clear_error();
    if (!g || !out) { set_error("null argument"); return 1; }
    std::vector<int> cells;
    if (tile_sizes(g, n, cells)) return 1;
    const double DX = (g->xmax - g->xmin) / g->num_cells;
    const bool has_l = (g->geometry == SX_GEOM_RL || g->geometry == SX_GEOM_RLZ);
    const bool has_z = (g->geometry == SX_GEOM_RZ || g->geometry == SX_GEOM_RLZ);
    int c0 = 0;
    for (int t = 0; t < n; t++) {
        double pts = 0;
        for (int r = c0 * MUBAR; r < (c0 + cells[t]) * MUBAR; r++) {
            int L, km;
            double off;
            ring_table(has_l, g->ring_uniform_L, r + 1, L, km, off);
            pts += L;
        }
        if (has_z) pts *= g->zDim;
        out[t * 5 + 0] = g->xmin + c0 * DX;
        out[t * 5 + 1] = g->xmin + (c0 + cells[t]) * DX;
        out[t * 5 + 2] = cells[t];
        out[t * 5 + 3] = c0 + 1;
        out[t * 5 + 4] = pts;
        c0 += cells[t];
    }
    return 0;
}

int sx_create(const sx_grid_desc *g, const sx_model_desc *m, sx_handle **out) {
    clear_error();
    if (!g || !m || !out) { set_error("null argument"); return 1; }
    if (g->abi_version != SX_ABI_VERSION) { set_error("sx_grid_desc.abi_version mismatch"); return 1; }
    if (g->geometry < SX_GEOM_R || g->geometry > SX_GEOM_RLZ) { set_error("Unknown geometry"); return 1; }
    if (g->num_cells < 3 || g->nvars < 1 || !(g->xmax > g->xmin)) { set_error("invalid grid parameters"); return 1; }
    if (g->tile_cell0 < 0 || g->tile_num_cells < 1 || g->tile_cell0 + g->tile_num_cells > g->num_cells) {
        set_error("tile range outside the patch");
        return 1;
    }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) {
        set_error("no HIP device available: libscythe_hip has no CPU fallback");
        return 1;
    }
    sx_handle *h = new sx_handle();
    std::string err;
    h->geom = g->geometry;
    h->has_l = (g->geometry == SX_GEOM_RL || g->geometry == SX_GEOM_RLZ);
    h->has_z = (g->geometry == SX_GEOM_RZ || g->geometry == SX_GEOM_RLZ);
    h->xmin = g->xmin; h->xmax = g->xmax; h->nc = g->num_cells;
    h->DX = (g->xmax - g->xmin) / g->num_cells;
    h->l_q = g->l_q > 0 ? g->l_q : 2.0;
    h->V = g->nvars;
    h->rDim = MUBAR * h->nc; h->b_rDim = h->nc + 3;
    h->uniform_L = h->has_l ? g->ring_uniform_L : 0;
    h->f32 = g->storage_f32 ? 1 : 0;
    h->sp32 = g->storage_f32 == 2 ? 1 : 0;
    if (g->storage_f32 < 0 || g->storage_f32 > 2) { set_error("storage_f32 must be 0, 1 or 2"); delete h; return 1; }
    h->overlap = getenv("SX_OVERLAP") ? atoi(getenv("SX_OVERLAP")) : 0;
    h->wide = !(getenv("SX_WIDE") && atoi(getenv("SX_WIDE")) == 0);
    h->fuse_zinv = getenv("SX_FUSE_ZINV") && atoi(getenv("SX_FUSE_ZINV")) != 0;
    h->sbw_prefetch = getenv("SX_SBW_PF") && atoi(getenv("SX_SBW_PF")) != 0;
    h->sbw_mfma = !(getenv("SX_SBW_MFMA") && atoi(getenv("SX_SBW_MFMA")) == 0);
    h->rz_fused = !(getenv("SX_RZ_FUSED") && atoi(getenv("SX_RZ_FUSED")) == 0);
    h->semi_mfma = !(getenv("SX_SEMI_MFMA") && atoi(getenv("SX_SEMI_MFMA")) == 0);
    h->use_graph = getenv("SX_GRAPH") && atoi(getenv("SX_GRAPH")) != 0;
    h->fft_reg = !(getenv("SX_FFT_REG") && atoi(getenv("SX_FFT_REG")) == 0);
    h->dft_merge = !(getenv("SX_DFT_MERGE") && atoi(getenv("SX_DFT_MERGE")) == 0);
    if (getenv("SX_DFT_HALFWG")) h->dft_half_wg = atoi(getenv("SX_DFT_HALFWG")) != 0;
    if (getenv("SX_DFT_EIGHTH")) h->dft_eighth = atoi(getenv("SX_DFT_EIGHTH")) != 0 ? 2 : 0;
    h->rl_quarter = !(getenv("SX_DFT_RLQ") && atoi(getenv("SX_DFT_RLQ")) == 0);
    h->solve_pcr = getenv("SX_SOLVE_PCR") ? atoi(getenv("SX_SOLVE_PCR")) : -1;
    if (getenv("SX_PCR_MAXCOLS")) h->pcr_maxcols = atoll(getenv("SX_PCR_MAXCOLS"));
    h->cell0 = g->tile_cell0; h->ncells = g->tile_num_cells; h->tile_num = g->tile_num;
    h->nrings = MUBAR * h->ncells; h->nbt = h->ncells + 3;
    for (int i = 0; i < 7; i++) h->slot[i] = DERIV_SLOTS[h->geom][i];
    h->D = h->geom == SX_GEOM_R ? 3 : h->geom == SX_GEOM_RLZ ? 7 : 5;
    h->ncoord = 1 + h->has_l + h->has_z;
    if (h->has_z) {
        h->nz = g->zDim;
        h->Zb = g->b_zDim > 0 ? g->b_zDim : default_bzdim(g->zDim);
        h->zmin = g->zmin; h->zmax = g->zmax;
        h->nsz = 3;
        if (h->nz < 4 || h->nz > 256 || h->Zb > h->nz || !(g->zmax > g->zmin)) {
            set_error("invalid vertical grid (need 4 <= zDim <= 256, b_zDim <= zDim, zmax > zmin)");
            delete h;
            return 1;
        }
    }
    if (h->uniform_L && (h->uniform_L < 4 || (h->uniform_L & 1))) { set_error("ring_uniform_L must be even and >= 4"); delete h; return 1; }
    auto bcv = [&](const int32_t *p, std::vector<int> &dst) {
        dst.assign(h->V, SX_BC_R0);
        if (p) for (int v = 0; v < h->V; v++) dst[v] = p[v];
    };
    bcv(g->bcl, h->bcl); bcv(g->bcl_k0 ? g->bcl_k0 : g->bcl, h->bcl0); bcv(g->bcr, h->bcr); bcv(g->bcb, h->bcb); bcv(g->bct, h->bct);
    // model
    h->ts = m->ts; h->eq = m->equation_set; h->semi = m->semiimplicit;
    h->w_index = m->w_index; h->xi_index = m->xi_index; h->col_var = m->col_var > 0 ? m->col_var : 1;
    if (m->params) std::memcpy(h->par, m->params, sizeof(double) * SX_NPARAMS);
    {
        const int eq = h->eq;
        const int need_geom = (eq == SX_EQ_LINEAR_ADVECTION_1D) ? SX_GEOM_R
                            : (eq == SX_EQ_LINEAR_ADVECTION_RZ || eq == SX_EQ_LINEAR_ACOUSTIC_RZ || eq == SX_EQ_EULER_TEST) ? SX_GEOM_RZ
                            : (eq == SX_EQ_LINEAR_ADVECTION_RL || eq == SX_EQ_ONEWAY_SW_SLAB || eq == SX_EQ_TWOWAY_SW_SLAB) ? SX_GEOM_RL
                            : (eq == SX_EQ_LINEAR_ADVECTION_RLZ || eq == SX_EQ_ONEWAY_SW_HRBL) ? SX_GEOM_RLZ : -1;
        const int need_vars = (eq == SX_EQ_LINEAR_ADVECTION_RZ) ? 4 : (eq == SX_EQ_LINEAR_ADVECTION_RL || eq == SX_EQ_LINEAR_ADVECTION_RLZ) ? 3
                            : (eq == SX_EQ_ONEWAY_SW_SLAB || eq == SX_EQ_TWOWAY_SW_SLAB || eq == SX_EQ_ONEWAY_SW_HRBL) ? 6
                            : (eq == SX_EQ_LINEAR_ACOUSTIC_RZ || eq == SX_EQ_EULER_TEST) ? 5 : 1;
        if (eq != SX_EQ_NONE && need_geom < 0) { set_error("equation set not in scope"); delete h; return 1; }
        if (eq != SX_EQ_NONE && (need_geom != h->geom || h->V < need_vars)) {
            set_error("equation set does not match the grid geometry / variable count");
            delete h;
            return 1;
        }
        if (h->semi && (!h->has_z || h->w_index < 1 || h->xi_index < 1 || h->w_index > h->V || h->xi_index > h->V || h->Zb != h->nz)) {
            set_error("semi-implicit adjustment needs an RZ/RLZ grid, w and xi variables and b_zDim == zDim");
            delete h;
            return 1;
        }
    }
#define FAIL()            \
    do {                  \
        sx_destroy(h);    \
        return 1;         \
    } while (0)

    // ---- ring tables (tile rings; kDim / K2 are patch-level)
    h->kDim = 0;
    for (int r = 0; r < h->rDim; r++) {
        int L, km;
        double off;
        ring_table(h->has_l, h->uniform_L, r + 1, L, km, off);
        h->kDim = std::max(h->kDim, km);
    }
    h->K2ref = 1 + 2 * h->kDim;
    h->K2 = h->has_l ? 2 * (h->kDim + 1) : 1;
    h->hL.resize(h->nrings); h->hkmax.resize(h->nrings); h->hoff.resize(h->nrings); h->hpstart.resize(h->nrings);
    std::vector<int64_t> twoff(h->nrings), phoff(h->nrings);
    std::vector<double2> tw, ph;
    int64_t pcount = 0;
    h->kDim_t = 0;
    std::map<int, int64_t> tw_of_L;
    for (int i = 0; i < h->nrings; i++) {
        int L, km;
        double off;
        ring_table(h->has_l, h->uniform_L, h->cell0 * MUBAR + i + 1, L, km, off);
        h->hL[i] = L; h->hkmax[i] = km; h->hoff[i] = off; h->hpstart[i] = pcount;
        pcount += L;
        h->kDim_t = std::max(h->kDim_t, km);
        h->L_max = std::max(h->L_max, L);
        auto it = tw_of_L.find(L);
        if (it == tw_of_L.end()) {
            tw_of_L[L] = (int64_t)tw.size();
            twoff[i] = (int64_t)tw.size();
            // angles formed in extended precision and rounded once (the Float64 product k * off alone loses k ulp: 1e-13 at k = 300)
            for (int j = 0; j < L; j++) tw.push_back(make_double2((double)cosl(2.0L * M_PIl * j / L), (double)sinl(2.0L * M_PIl * j / L)));
        } else {
            twoff[i] = it->second;
        }
        phoff[i] = (int64_t)ph.size();
        for (int k = 0; k <= km; k++) ph.push_back(make_double2((double)cosl((long double)k * (long double)off), (double)sinl((long double)k * (long double)off)));
    }
    h->kmax_max = h->kDim_t;
    h->L_all_mult4 = true;
    for (int i = 0; i < h->nrings; i++) if (h->hL[i] % 4) h->L_all_mult4 = false;
    h->K2t = 1 + 2 * h->kDim_t;
    h->Nh = pcount;
    h->N = pcount * h->nz;
    h->C = (int64_t)h->V * h->Zb * h->K2;
    h->S_patch = (int64_t)h->Zb * h->K2ref * h->b_rDim;
    h->S_tile = (int64_t)h->Zb * h->K2t * h->nbt;

    // ---- radial tables
    double phi[4][MUBAR][4], wq[MUBAR];
    basis_tables(h->DX, phi);
    quad_weights(h->DX, wq);
    std::vector<double> hphi((size_t)3 * h->nrings * 4), hwq(h->nrings), hr(h->Nh), hcos(h->Nh), hsin(h->Nh);
    const double off3[MUBAR] = {-std::sqrt(3.0 / 5.0) / 2.0, 0.0, std::sqrt(3.0 / 5.0) / 2.0};
    for (int i = 0; i < h->nrings; i++) {
        const int mu = i % MUBAR, c = h->cell0 + i / MUBAR;
        for (int d = 0; d < 3; d++)
            for (int j = 0; j < 4; j++) hphi[((size_t)d * h->nrings + i) * 4 + j] = phi[d][mu][j];
        hwq[i] = wq[mu];
        const double r = h->xmin + h->DX * (c + 0.5 + off3[mu]);
        for (int l = 0; l < h->hL[i]; l++) {
            const double lam = h->hoff[i] + 2.0 * M_PI * l / h->hL[i];
            hr[h->hpstart[i] + l] = r;
            hcos[h->hpstart[i] + l] = std::cos(lam);
            hsin[h->hpstart[i] + l] = std::sin(lam);
        }
    }
    if (!upload(h, &h->d_phi, hphi) || !upload(h, &h->d_wq, hwq) || !upload(h, &h->d_r, hr) || !upload(h, &h->d_cosl, hcos) ||
        !upload(h, &h->d_sinl, hsin) || !upload(h, &h->d_L, h->hL) || !upload(h, &h->d_kmax, h->hkmax) ||
        !upload(h, &h->d_pstart, h->hpstart) || !upload(h, &h->d_twoff, twoff) || !upload(h, &h->d_phoff, phoff) ||
        !upload(h, &h->d_tw, tw) || !upload(h, &h->d_ph, ph))
        FAIL();

    // ---- spline classes
    std::vector<SplineClass> classes;
    std::vector<int> cls((size_t)h->V * 2);
    for (int v = 0; v < h->V; v++)
        for (int q = 0; q < 2; q++) {
            const int bl = q == 0 ? h->bcl0[v] : h->bcl[v], br = h->bcr[v];
            int found = -1;
            for (size_t c = 0; c < classes.size(); c++)
                if (classes[c].bcl == bl && classes[c].bcr == br) found = (int)c;
            if (found < 0) {
                SplineClass sc;
                if (!build_spline_class(h->nc, h->DX, h->l_q, bl, br, sc, err)) { set_error(err); FAIL(); }
                classes.push_back(sc);
                found = (int)classes.size() - 1;
            }
            cls[(size_t)v * 2 + q] = found;
        }
    h->ncls = (int)classes.size();
    h->classes = classes;
    h->hcls = cls;
    {
        const int nb = h->b_rDim;
        std::vector<int> cmeta((size_t)h->ncls * 4);
        std::vector<double> gl((size_t)h->ncls * 6), gr((size_t)h->ncls * 6), Lb((size_t)h->ncls * nb * 4), La((size_t)h->ncls * 3 * nb);
        std::vector<double> Ldinv((size_t)h->ncls * nb, 0.0);
        for (int c = 0; c < h->ncls; c++) {
            const SplineClass &s = classes[c];
            cmeta[c * 4 + 0] = s.nfree; cmeta[c * 4 + 1] = s.periodic; cmeta[c * 4 + 2] = s.rl; cmeta[c * 4 + 3] = s.rr;
            for (int i = 0; i < 3; i++)
                for (int j = 0; j < 2; j++) { gl[c * 6 + i * 2 + j] = s.gl[i][j]; gr[c * 6 + i * 2 + j] = s.gr[i][j]; }
            std::copy(s.Lband.begin(), s.Lband.end(), Lb.begin() + (size_t)c * nb * 4);
            std::copy(s.Larrow.begin(), s.Larrow.end(), La.begin() + (size_t)c * 3 * nb);
            for (int i = 0; i < s.nfree; i++) Ldinv[(size_t)c * nb + i] = 1.0 / s.Lband[(size_t)i * 4 + 3];
        }
        if (!upload(h, &h->d_cls, cls) || !upload(h, &h->d_cmeta, cmeta) || !upload(h, &h->d_gl, gl) || !upload(h, &h->d_gr, gr) ||
            !upload(h, &h->d_Lband, Lb) || !upload(h, &h->d_Larrow, La) || !upload(h, &h->d_Ldinv, Ldinv))
            FAIL();
    }

    // ---- Chebyshev operators
    if (h->has_z) {
        const int nz = h->nz, Zb = h->Zb;
        std::vector<double> Mz((size_t)h->V * 3 * nz * Zb), MzT(Mz.size());
        std::vector<ChebOps> ops(h->V);
        for (int v = 0; v < h->V; v++) {
            if (!build_cheb_ops(h->zmin, h->zmax, nz, Zb, h->bcb[v], h->bct[v], ops[v], err)) { set_error(err); FAIL(); }
            for (int d = 0; d < 3; d++) {
                const size_t o = ((size_t)v * 3 + d) * nz * Zb;
                std::copy(ops[v].M[d].begin(), ops[v].M[d].end(), Mz.begin() + o);
                for (int i = 0; i < nz; i++)                     // [Zb][nz] copy for the matrix-core kernel (k_colmat_mfma)
                    for (int k = 0; k < Zb; k++) MzT[o + (size_t)k * nz + i] = ops[v].M[d][(size_t)i * Zb + k];
            }
        }
        if (!upload(h, &h->d_MzT, MzT)) FAIL();
        const ChebOps &cop = ops[h->col_var - 1 < h->V ? h->col_var - 1 : 0];
        std::vector<double> zv = cop.z;
        if (!upload(h, &h->d_Mz, Mz) || !upload(h, &h->d_CB, ops[0].CB) || !upload(h, &h->d_z, zv)) FAIL();
        {
            std::vector<double> cbt((size_t)nz * Zb);          // [nz][Zb] for the fused RZ forward kernel (sx_rz.hip)
            for (int k = 0; k < Zb; k++)
                for (int i = 0; i < nz; i++) cbt[(size_t)i * Zb + k] = ops[0].CB[(size_t)k * nz + i];
            if (!upload(h, &h->d_CBT, cbt)) FAIL();
        }
        if (h->semi) {
            const ChebOps &ox = ops[h->xi_index - 1], &ow = ops[h->w_index - 1];
            if (!upload(h, &h->d_MrecT, transpose(ox.Mrec, nz)) || !upload(h, &h->d_MdzT, transpose(ox.Mdz, nz))) FAIL();
            h->tau[0] = 0.5 * h->ts;     // first step: trapezoidal (src/semiimplicit.jl:544-548)
            h->tau[1] = 1.25 * h->ts;    // AI2*        (src/semiimplicit.jl:549-558, 96)
            for (int q = 0; q < 2; q++) {
                std::vector<double> W, X;
                if (!build_helmholtz(ow, h->par[SX_P_PXI_BAR], h->tau[q], W, X, err)) { set_error(err); FAIL(); }
                if (!upload(h, &h->d_WT[q], transpose(W, nz)) || !upload(h, &h->d_XT[q], transpose(X, nz))) FAIL();
            }
        } else {
            if (!upload(h, &h->d_MintT, transpose(cop.Mint, nz)) || !upload(h, &h->d_MdzT, transpose(cop.Mdz, nz))) FAIL();
        }
    } else {
        std::vector<double> one(1, 0.0);
        if (!upload(h, &h->d_z, one)) FAIL();
    }

    if (h->eq == SX_EQ_EULER_TEST) {
        if (!m->ref_state) { set_error("Euler_test needs sx_model_desc.ref_state (ReferenceState)"); FAIL(); }
        std::vector<double> ref(m->ref_state, m->ref_state + (size_t)9 * h->nz);
        if (!upload(h, &h->d_ref, ref)) FAIL();
    }

    // ---- state
    const int64_t C = h->C, N = h->N;
    if (!dalloc(h, &h->d_A, (size_t)h->b_rDim * C) || !dalloc(h, &h->d_Bfull, (size_t)h->b_rDim * C) ||
        !dalloc(h, &h->d_rowoff, (size_t)h->b_rDim) || !dalloc(h, &h->d_aoff, (size_t)h->b_rDim) || !dalloc(h, &h->d_neg1, (size_t)h->b_rDim) || !dalloc(h, &h->d_phys, plane_count(h, (size_t)N)) ||
        !dalloc(h, &h->d_np1, (size_t)h->V * N) || !dalloc(h, &h->d_flag, 1))
        FAIL();
    for (int i = 0; i < 3; i++) {
        if (!dalloc(h, &h->d_E[i], (size_t)h->V * N)) FAIL();
        if (h->semi && !dalloc(h, &h->d_I[i], (size_t)h->V * N)) FAIL();
    }
    if (h->sp32 && !(fft_path_ok(h) && h->has_z && (h->nz == 32 || h->nz == 64 || h->nz == 128) && h->sbw_mfma &&
                     (h->nz <= 64 ? h->Zb <= 64 : h->Zb <= 96))) {
        set_error("storage_f32 = 2 (fp32 spectral intermediates) needs an RLZ / RZ grid on a uniform power-of-two ring table with zDim 32, 64 or 128");
        FAIL();
    }
    if (!dalloc(h, &h->d_Fl, (size_t)h->nrings * h->V * h->nz * h->K2)) FAIL();
    if (h->has_z) {
        if (!dalloc(h, &h->d_Az, (size_t)h->nbt * h->V * 3 * h->nz * h->K2)) FAIL();
    }
    if (h->ncells == h->nc) {
        h->d_Btile = h->d_Bfull;                      // one-tile patch: the tile's B rows are the patch's B rows
    } else {
        if (!dalloc(h, &h->d_Btile_own, (size_t)h->nbt * C)) FAIL();
        h->d_Btile = h->d_Btile_own;
    }
    {   // derivative slots each equation set reads (everything else is skipped by sx_advance's inverse transform)
        const int u = 1 << h->slot[0], r = 1 << h->slot[1], rr = 1 << h->slot[2];
        const int l = h->has_l ? 1 << h->slot[3] : 0, ll = h->has_l ? 1 << h->slot[4] : 0;
        const int z = h->has_z ? 1 << h->slot[5] : 0, zz = h->has_z ? 1 << h->slot[6] : 0;
        std::vector<int> full(h->V, (1 << h->D) - 1), eq(h->V, u);
        switch (h->eq) {
            case SX_EQ_LINEAR_ADVECTION_1D: eq[0] = u | r | rr; break;
            case SX_EQ_LINEAR_ADVECTION_RZ: eq[0] = u | r | rr | z | zz; break;
            case SX_EQ_LINEAR_ADVECTION_RL: case SX_EQ_LINEAR_ADVECTION_RLZ: eq[0] = u | r | rr | l | ll; break;
            case SX_EQ_ONEWAY_SW_SLAB: case SX_EQ_TWOWAY_SW_SLAB: case SX_EQ_ONEWAY_SW_HRBL:
                eq[0] = eq[1] = eq[2] = u | r | l;
                eq[3] = eq[4] = u | r | rr | l | ll | (h->eq == SX_EQ_ONEWAY_SW_HRBL ? z : 0);
                eq[5] = 0;        // w is diagnostic: written by the equation set before it is read
                break;
            case SX_EQ_LINEAR_ACOUSTIC_RZ: case SX_EQ_EULER_TEST:
                eq[0] = eq[2] = eq[3] = eq[4] = u | r | rr | z | zz;
                eq[1] = u | r | z;
                break;
            default: break;
        }
        for (int v = 0; v < h->V; v++) {
            h->mask_full_bits += __builtin_popcount(full[v]);
            h->mask_eq_bits += __builtin_popcount(eq[v]);
            h->mask_full_val += (full[v] & u) ? 1 : 0;
            h->mask_eq_val += (eq[v] & u) ? 1 : 0;
        }
        h->hmask_full = full; h->hmask_eq = eq;
        if (!upload(h, &h->d_mask_full, full) || !upload(h, &h->d_mask_eq, eq)) FAIL();
        if (h->has_z && dft_mfma_ok(h)) {
            // Work lists of the native-ring DFT kernels: one launch over all rings, workgroups dispatched in order of
            // decreasing cost (ring length squared x planes of the variable) so that the cheap ones fill the tail;
            // variables without a requested slot get no workgroup at all.
            for (int which = 0; which < 3; which++) {
                std::vector<std::array<int64_t, 3>> it;          // (cost, ring, v)
                for (int r = 0; r < h->nrings; r++)
                    for (int v = 0; v < h->V; v++) {
                        const int planes = which == 2 ? 1 : __builtin_popcount(which == 0 ? eq[v] : full[v]);
                        if (planes == 0) continue;
                        it.push_back({(int64_t)h->hL[r] * h->hL[r] * planes, r, v});
                    }
                std::stable_sort(it.begin(), it.end(), [](const auto &a, const auto &b) { return a[0] > b[0]; });
                // rings whose coefficient sets do not fit the LDS in one piece (kmax > 319) go first and take the chunked
                // kernels (sx_dft.hip); the rest keeps the single-pass kernels, sized for ITS largest ring
                std::stable_partition(it.begin(), it.end(), [&](const auto &e) { return h->hkmax[e[1]] > DFT_KMAX_SINGLE; });
                std::vector<int> flat;
                int nbig = 0;
                for (const auto &e : it) {
                    flat.push_back((int)e[1]); flat.push_back((int)e[2]);
                    if (h->hkmax[e[1]] > DFT_KMAX_SINGLE) nbig++;
                    else { h->dft_lcap_small = std::max(h->dft_lcap_small, h->hL[e[1]]); h->dft_kcap_small = std::max(h->dft_kcap_small, h->hkmax[e[1]]); }
                }
                h->n_dft_big[which] = nbig;
                h->n_dft_items[which] = (int)it.size();
                if (!upload(h, &h->d_dft_items[which], flat)) FAIL();
            }
        }
        // node-space ("radial last") inverse: uniform power-of-two rings + the MFMA HRBL kernel
        if (h->eq == SX_EQ_ONEWAY_SW_HRBL && h->V == 6 && fft_path_ok(h) && h->has_z && (h->nz == 64 || h->nz == 32 || h->nz == 128) &&
            !(getenv("SX_NODE_MODE") && atoi(getenv("SX_NODE_MODE")) == 0)) {
            h->node_mode = 1;
            h->R_in = 0;
            while (h->R_in < h->nrings && h->hkmax[h->R_in] < h->kDim) h->R_in++;
            h->R_in = std::min(h->nrings, (h->R_in + MUBAR - 1) / MUBAR * MUBAR);   // whole cells on either side
            for (int i = h->R_in; i < h->nrings; i++)
                if (h->hkmax[i] != h->kDim) h->node_mode = 0;          // truncation must be ring-independent beyond R_in
            if (h->R_in == h->nrings) h->node_mode = 0;                // no ring on the node-space path
        }
        if (h->node_mode) {
            const int L = h->uniform_L;
            h->NG = (int64_t)h->nbt * L * h->nz;
            std::vector<int> nkmax(h->nbt, h->kDim), nmask(h->V, 0);
            std::vector<int64_t> npstart(h->nbt), nphoff(h->nbt, phoff[h->nrings - 1]);   // zero phase offset: all (1, 0)
            std::vector<double> nphi((size_t)3 * h->nbt * 4, 0.0);
            for (int j = 0; j < h->nbt; j++) { npstart[j] = (int64_t)j * L; nphi[(size_t)j * 4] = 1.0; }
            for (int v = 0; v < h->V; v++) {
                if (eq[v] & (u | r | rr)) nmask[v] |= u;
                nmask[v] |= eq[v] & (l | ll | z | zz);
                h->mask_node_bits += __builtin_popcount(nmask[v]);
                h->mask_node_val += (nmask[v] & u) ? 1 : 0;
            }
            if (!upload(h, &h->d_nkmax, nkmax) || !upload(h, &h->d_npstart, npstart) || !upload(h, &h->d_nphoff, nphoff) ||
                !upload(h, &h->d_nphi, nphi) || !upload(h, &h->d_mask_node, nmask) ||
                !dalloc(h, &h->d_G, plane_count(h, (size_t)h->NG)))
                FAIL();
        }
        if (h->has_z) {   // vertical-transform job lists: only the (variable, operator) pairs some requested slot needs
            const int horiz = u | r | rr | l | ll;
            std::vector<ColJob> jf, je, jz;
            for (int v = 0; v < h->V; v++) {
                for (int sz = 0; sz < 3; sz++) {
                    ColJob j;
                    j.in_off = (int64_t)v * h->Zb * h->K2;
                    j.out_off = ((int64_t)v * 3 + sz) * h->nz * h->K2;
                    j.mat_off = ((int64_t)v * 3 + sz) * h->nz * h->Zb;
                    jf.push_back(j);
                    const int need = sz == 0 ? horiz : sz == 1 ? z : zz;
                    if (eq[v] & need) je.push_back(j);
                }
                ColJob f;
                f.in_off = (int64_t)v * h->nz * h->K2;
                f.out_off = (int64_t)v * h->Zb * h->K2;
                f.mat_off = 0;
                jz.push_back(f);
            }
            h->njobs_zinv_full = (int)jf.size();
            h->njobs_zinv_eq = (int)je.size();
            if (!upload(h, &h->d_jobs_zinv_full, jf) || !upload(h, &h->d_jobs_zinv_eq, je) || !upload(h, &h->d_jobs_zf, jz)) FAIL();
        }
    }
    {
        std::vector<int64_t> ao(h->b_rDim), neg(h->b_rDim, -1);
        for (int m = 0; m < h->b_rDim; m++) ao[m] = (int64_t)m * h->C;
        if (hipMemcpy(h->d_aoff, ao.data(), sizeof(int64_t) * ao.size(), hipMemcpyHostToDevice) != hipSuccess ||
            hipMemcpy(h->d_neg1, neg.data(), sizeof(int64_t) * neg.size(), hipMemcpyHostToDevice) != hipSuccess) {
            set_error("hipMemcpy H2D failed");
            FAIL();
        }
    }
    h->v_lo = 0; h->v_cnt = h->V;
    // deferred diagnostic variable (sx_internal.hpp): one-tile HRBL runs whose forward path is the FFT + matrix-core kernels
    h->defer_diag = getenv("SX_DEFER_DIAG") && atoi(getenv("SX_DEFER_DIAG")) != 0 && h->eq == SX_EQ_ONEWAY_SW_HRBL && h->V == 6 &&
                    h->ncells == h->nc && fft_path_ok(h) && h->has_z && (h->nz == 32 || h->nz == 64 || h->nz == 128) && h->sbw_mfma &&
                    (h->nz <= 64 ? h->Zb <= 64 : h->Zb <= 96);
    if (sx_bind_patch_b(h, nullptr, nullptr)) FAIL();
    *out = h;
    return status();
#undef FAIL
}

int sx_destroy(sx_handle *h) {
    if (!h) return 0;
    hipDeviceSynchronize();
#ifdef SX_PHASES
    phases_dump();
    fft_phases_dump();
    sbw_phases_dump();
    dft_phases_dump();
#endif
    graphs_release(h);
    if (h->graph_stream) hipStreamDestroy(h->graph_stream);
    comm_release(h);
    iface_release(h);
    pcr_release(h);
    for (auto &p : h->pending) { hipEventDestroy(p.a); hipEventDestroy(p.b); }
    for (auto e : h->event_pool) hipEventDestroy(e);
    if (h->ev_fork) hipEventDestroy(h->ev_fork);
    if (h->ev_join) hipEventDestroy(h->ev_join);
    if (h->stream2) hipStreamDestroy(h->stream2);
    for (void *p : h->allocs) hipFree(p);
    delete h;
    return 0;
}

int sx_cheb_column_ops(double zmin, double zmax, int32_t zDim, int32_t b_zDim, int32_t bcb, int32_t bct, double *z,
                       double *rec, double *dz, double *dzz, double *integ) {
    clear_error();
    if (zDim < 2 || !(zmax > zmin)) { set_error("sx_cheb_column_ops: need zDim >= 2 and zmax > zmin"); return 1; }
    const int Zb = b_zDim > 0 ? std::min(b_zDim, zDim) : std::min(zDim, (2 * zDim - 1) / 3 + 1);
    ChebOps o;
    std::string err;
    if (!build_cheb_ops(zmin, zmax, zDim, Zb, bcb, bct, o, err)) { set_error(err); return 1; }
    const size_t n2 = (size_t)zDim * zDim;
    if (z) std::copy(o.z.begin(), o.z.end(), z);
    if (rec) std::copy(o.Mrec.begin(), o.Mrec.begin() + n2, rec);
    if (dz) std::copy(o.Mdz.begin(), o.Mdz.begin() + n2, dz);
    if (dzz) std::copy(o.Mdzz.begin(), o.Mdzz.begin() + n2, dzz);
    if (integ) std::copy(o.Mint.begin(), o.Mint.begin() + n2, integ);
    return 0;
}

int sx_spline_solve_check(int32_t num_cells, double xmin, double xmax, double l_q, int32_t bcl, int32_t bcr, const double *b,
                          double *a_pcr, double *a_chol, int32_t *levels) {
    clear_error();
    if (!b || num_cells < 3 || !(xmax > xmin)) { set_error("sx_spline_solve_check: invalid argument"); return 1; }
    SplineClass sc;
    PcrTables t;
    std::string err;
    const int nb = num_cells + 3;
    if (!build_spline_class(num_cells, (xmax - xmin) / num_cells, l_q > 0 ? l_q : 2.0, bcl, bcr, sc, err) || !build_pcr_tables(sc, nb, t, err)) {
        set_error(err);
        return 1;
    }
    if (a_pcr) pcr_apply_host(t, nb, b, a_pcr);
    if (a_chol) cholesky_apply_host(sc, nb, b, a_chol);
    if (levels) *levels = t.levels;
    return 0;
}

int sx_get_dims(const sx_handle *h, sx_dims *o) {
    clear_error();
    if (!h || !o) { set_error("null argument"); return 1; }
    o->n_points = h->N; o->n_hpoints = h->Nh; o->n_vars = h->V; o->n_derivs = h->D; o->n_coord = h->ncoord;
    o->rDim = h->rDim; o->b_rDim = h->b_rDim; o->tile_rDim = h->nrings; o->tile_b_rDim = h->nbt;
    o->zDim = h->has_z ? h->nz : 0; o->b_zDim = h->has_z ? h->Zb : 0;
    o->kDim = h->kDim; o->n_blocks = h->K2ref; o->tile_kDim = h->kDim_t; o->tile_n_blocks = h->K2t;
    o->s_patch = h->S_patch; o->s_tile = h->S_tile; o->n_cols = h->C;
    return 0;
}

int sx_set_stream(sx_handle *h, void *s) {
    clear_error();
    if (!h) { set_error("null handle"); return 1; }
    if (h->stream != (hipStream_t)s) graphs_release(h);      // captured on the old stream
    h->stream = (hipStream_t)s;
    return 0;
}

int sx_synchronize(sx_handle *h) {
    clear_error();
    if (!h) { set_error("null handle"); return 1; }
    HIPOK(hipStreamSynchronize(h->stream));
    return status();
}

int sx_get_gridpoints(const sx_handle *h, double *out) {
    clear_error();
    if (!h || !out) { set_error("null argument"); return 1; }
    const double off3[MUBAR] = {-std::sqrt(3.0 / 5.0) / 2.0, 0.0, std::sqrt(3.0 / 5.0) / 2.0};
    std::vector<double> z(h->nz, 0.0);
    if (h->has_z)
        for (int n = 0; n < h->nz; n++)
            z[n] = std::cos(n * M_PI / (h->nz - 1)) * (-0.5 * (h->zmax - h->zmin)) + 0.5 * (h->zmin + h->zmax);
    int64_t p = 0;
    for (int i = 0; i < h->nrings; i++) {
        const int mu = i % MUBAR, c = h->cell0 + i / MUBAR;
        const double r = h->xmin + h->DX * (c + 0.5 + off3[mu]);
        for (int l = 0; l < h->hL[i]; l++) {
            const double lam = h->hoff[i] + 2.0 * M_PI * l / h->hL[i];
            for (int k = 0; k < h->nz; k++, p++) {
                int col = 0;
                out[p] = r;
                if (h->has_l) out[(int64_t)(++col) * h->N + p] = lam;
                if (h->has_z) out[(int64_t)(++col) * h->N + p] = z[k];
            }
        }
    }
    return 0;
}

int sx_set_physical_values(sx_handle *h, const double *values) {
    clear_error();
    if (!h || !values) { set_error("null argument"); return 1; }
    HIPOK(hipMemcpyAsync(h->d_np1, values, sizeof(double) * h->V * h->N, hipMemcpyHostToDevice, h->stream));
    HIPOK(hipStreamSynchronize(h->stream));
    return status();
}

int sx_get_physical(sx_handle *h, double *out) {
    clear_error();
    if (!h || !out) { set_error("null argument"); return 1; }
    const size_t n = (size_t)h->D * h->V * h->N;
    if (h->f32) {          // derivative slots are stored as fp32: widen on the host
        const size_t nv = (size_t)h->V * h->N, nd = n - nv;
        std::vector<float> tmp(nd);
        HIPOK(hipMemcpyAsync(out, h->d_phys, sizeof(double) * nv, hipMemcpyDeviceToHost, h->stream));
        HIPOK(hipMemcpyAsync(tmp.data(), h->d_phys + nv, sizeof(float) * nd, hipMemcpyDeviceToHost, h->stream));
        HIPOK(hipStreamSynchronize(h->stream));
        for (size_t i = 0; i < nd; i++) out[nv + i] = (double)tmp[i];
        return status();
    }
    HIPOK(hipMemcpyAsync(out, h->d_phys, sizeof(double) * n, hipMemcpyDeviceToHost, h->stream));
    HIPOK(hipStreamSynchronize(h->stream));
    return status();
}

int sx_get_var_np1(sx_handle *h, double *out) {
    clear_error();
    if (!h || !out) { set_error("null argument"); return 1; }
    HIPOK(hipMemcpyAsync(out, h->d_np1, sizeof(double) * h->V * h->N, hipMemcpyDeviceToHost, h->stream));
    HIPOK(hipStreamSynchronize(h->stream));
    return status();
}

// restart blob: [header 4 doubles: magic, nbt * C, V * N, semi][A rows of the tile][E_nm1][E_nm2]([I_nm1][I_nm2])
static const double SX_STATE_MAGIC = 5.3171e7;

int sx_state_size(const sx_handle *h, int64_t *n_doubles) {
    clear_error();
    if (!h || !n_doubles) { set_error("null argument"); return 1; }
    *n_doubles = 4 + (int64_t)h->nbt * h->C + (int64_t)(h->semi ? 4 : 2) * h->V * h->N;
    return 0;
}

int sx_get_state(sx_handle *h, double *out) {
    clear_error();
    if (!h || !out) { set_error("null argument"); return 1; }
    flush_diag(h);
    const size_t na = (size_t)h->nbt * h->C, nv = (size_t)h->V * h->N;
    out[0] = SX_STATE_MAGIC; out[1] = (double)na; out[2] = (double)nv; out[3] = h->semi ? 1.0 : 0.0;
    double *p = out + 4;
    HIPOK(hipMemcpyAsync(p, h->d_A + (int64_t)h->cell0 * h->C, sizeof(double) * na, hipMemcpyDeviceToHost, h->stream));
    p += na;
    // after a step the buffer written as expdot_n has become nm1: E1 = d_E[(rot + 1) % 3], E2 = d_E[(rot + 2) % 3]
    for (int q = 1; q <= 2; q++, p += nv)
        HIPOK(hipMemcpyAsync(p, h->d_E[(h->rot + q) % 3], sizeof(double) * nv, hipMemcpyDeviceToHost, h->stream));
    if (h->semi)
        for (int q = 1; q <= 2; q++, p += nv)
            HIPOK(hipMemcpyAsync(p, h->d_I[(h->rot + q) % 3], sizeof(double) * nv, hipMemcpyDeviceToHost, h->stream));
    HIPOK(hipStreamSynchronize(h->stream));
    return status();
}

int sx_set_state(sx_handle *h, const double *in) {
    clear_error();
    if (!h || !in) { set_error("null argument"); return 1; }
    const size_t na = (size_t)h->nbt * h->C, nv = (size_t)h->V * h->N;
    if (in[0] != SX_STATE_MAGIC || in[1] != (double)na || in[2] != (double)nv || in[3] != (h->semi ? 1.0 : 0.0)) {
        set_error("sx_set_state: the blob does not belong to a handle with these dimensions");
        return 1;
    }
    h->diag_dirty = false;          // the (validated) blob carries every variable's coefficients
    const double *p = in + 4;
    HIPOK(hipMemcpyAsync(h->d_A + (int64_t)h->cell0 * h->C, p, sizeof(double) * na, hipMemcpyHostToDevice, h->stream));
    p += na;
    h->rot = 0;
    for (int q = 1; q <= 2; q++, p += nv)
        HIPOK(hipMemcpyAsync(h->d_E[q], p, sizeof(double) * nv, hipMemcpyHostToDevice, h->stream));
    if (h->semi)
        for (int q = 1; q <= 2; q++, p += nv)
            HIPOK(hipMemcpyAsync(h->d_I[q], p, sizeof(double) * nv, hipMemcpyHostToDevice, h->stream));
    HIPOK(hipStreamSynchronize(h->stream));
    return status();
}

int sx_get_tile_spectral(sx_handle *h, double *out) {
    clear_error();
    if (!h || !out) { set_error("null argument"); return 1; }
    flush_diag(h);
    std::vector<double> tmp((size_t)h->nbt * h->C);
    HIPOK(hipMemcpyAsync(tmp.data(), h->d_Btile, sizeof(double) * tmp.size(), hipMemcpyDeviceToHost, h->stream));
    HIPOK(hipStreamSynchronize(h->stream));
    for (int v = 0; v < h->V; v++)
        for (int zm = 0; zm < h->Zb; zm++)
            for (int blk = 0; blk < h->K2t; blk++)
                for (int j = 0; j < h->nbt; j++)
                    out[(int64_t)v * h->S_tile + ref_index(zm, blk, j, h->K2t, h->nbt)] =
                        tmp[(size_t)j * h->C + ((size_t)v * h->Zb + zm) * h->K2 + dev_blk(blk)];
    return status();
}

static int patch_to_device(sx_handle *h, const double *src, double *dst) {
    std::vector<double> tmp((size_t)h->b_rDim * h->C, 0.0);
    for (int v = 0; v < h->V; v++)
        for (int zm = 0; zm < h->Zb; zm++)
            for (int blk = 0; blk < h->K2ref; blk++)
                for (int m = 0; m < h->b_rDim; m++)
                    tmp[(size_t)m * h->C + ((size_t)v * h->Zb + zm) * h->K2 + dev_blk(blk)] =
                        src[(int64_t)v * h->S_patch + ref_index(zm, blk, m, h->K2ref, h->b_rDim)];
    HIPOK(hipMemcpyAsync(dst, tmp.data(), sizeof(double) * tmp.size(), hipMemcpyHostToDevice, h->stream));
    HIPOK(hipStreamSynchronize(h->stream));
    return status();
}

int sx_set_patch_spectral_b(sx_handle *h, const double *shared) {
    clear_error();
    if (!h || !shared) { set_error("null argument"); return 1; }
    h->diag_dirty = false;          // every variable's B arrives with the (non-null) argument
    return patch_to_device(h, shared, h->d_Bfull);
}

int sx_set_patch_spectral_a(sx_handle *h, const double *a) {
    clear_error();
    if (!h || !a) { set_error("null argument"); return 1; }
    h->diag_dirty = false;
    return patch_to_device(h, a, h->d_A);
}

int sx_get_patch_spectral_a(sx_handle *h, double *out) {
    clear_error();
    if (!h || !out) { set_error("null argument"); return 1; }
    flush_diag(h);
    std::vector<double> tmp((size_t)h->b_rDim * h->C);
    HIPOK(hipMemcpyAsync(tmp.data(), h->d_A, sizeof(double) * tmp.size(), hipMemcpyDeviceToHost, h->stream));
    HIPOK(hipStreamSynchronize(h->stream));
    for (int v = 0; v < h->V; v++)
        for (int zm = 0; zm < h->Zb; zm++)
            for (int blk = 0; blk < h->K2ref; blk++)
                for (int m = 0; m < h->b_rDim; m++)
                    out[(int64_t)v * h->S_patch + ref_index(zm, blk, m, h->K2ref, h->b_rDim)] =
                        tmp[(size_t)m * h->C + ((size_t)v * h->Zb + zm) * h->K2 + dev_blk(blk)];
    return status();
}

// calcPatchMap / calcHaloMap (src/semiimplicit.jl:79-86): where the entries of one variable's tile.spectral column live in
// the patch's column.  Tile block entry j <-> patch entry spectralIndexL - 1 + j; a tile OWNS the first num_cells entries
// of each (z-mode, wavenumber block) - the last tile all num_cells + 3 - and SENDS the remaining 3 to the next tile.
static void index_maps(const sx_handle *h, std::vector<int64_t> &po, std::vector<int64_t> &to, std::vector<int64_t> &ph,
                       std::vector<int64_t> &th) {
    const bool last = (h->cell0 + h->ncells == h->nc);
    const int owned = h->ncells + (last ? 3 : 0);
    for (int zm = 0; zm < h->Zb; zm++)
        for (int blk = 0; blk < h->K2t; blk++) {
            for (int j = 0; j < owned; j++) {
                po.push_back(1 + ref_index(zm, blk, h->cell0 + j, h->K2ref, h->b_rDim));
                to.push_back(1 + ref_index(zm, blk, j, h->K2t, h->nbt));
            }
            for (int j = owned; j < h->nbt; j++) {
                ph.push_back(1 + ref_index(zm, blk, h->cell0 + j, h->K2ref, h->b_rDim));
                th.push_back(1 + ref_index(zm, blk, j, h->K2t, h->nbt));
            }
        }
}

int sx_index_map_sizes(const sx_handle *h, int64_t *n_owned, int64_t *n_halo) {
    clear_error();
    if (!h || !n_owned || !n_halo) { set_error("null argument"); return 1; }
    const bool last = (h->cell0 + h->ncells == h->nc);
    const int64_t blocks = (int64_t)h->Zb * h->K2t;
    *n_owned = blocks * (h->ncells + (last ? 3 : 0));
    *n_halo = blocks * (last ? 0 : 3);
    return 0;
}

int sx_index_maps(const sx_handle *h, int64_t *patch_owned, int64_t *tile_owned, int64_t *patch_halo, int64_t *tile_halo) {
    clear_error();
    if (!h) { set_error("null handle"); return 1; }
    std::vector<int64_t> po, to, ph, th;
    index_maps(h, po, to, ph, th);
    if (patch_owned) std::copy(po.begin(), po.end(), patch_owned);
    if (tile_owned) std::copy(to.begin(), to.end(), tile_owned);
    if (patch_halo) std::copy(ph.begin(), ph.end(), patch_halo);
    if (tile_halo) std::copy(th.begin(), th.end(), tile_halo);
    return 0;
}

}  // extern "C"

namespace sx {
// SX_DEFER_DIAG: bring the diagnostic variable's B and A coefficients up to date (forward transform of var_np1[w], radial +
// vertical inner products, banded solve, for that one variable) - called by everything that lets A or B be observed
void flush_diag(sx_handle *h) {
    if (!h->diag_dirty) return;
    h->diag_dirty = false;
    h->v_lo = h->V - 1; h->v_cnt = 1;
    launch_fl_forward(h);
    launch_sb(h);
    launch_solve(h);
    h->v_lo = 0; h->v_cnt = h->V;
}
}  // namespace sx

extern "C" {

int sx_spectral_transform(sx_handle *h) {
    clear_error();
    if (!h) { set_error("null handle"); return 1; }
    h->diag_dirty = false;             // every variable is transformed here
    launch_fl_forward(h);
    launch_sb(h);
    return status();
}

int sx_spline_transform(sx_handle *h) {
    clear_error();
    if (!h) { set_error("null handle"); return 1; }
    if (h->diag_dirty) h->v_cnt = h->V - 1;      // after a deferred sx_advance: the prognostic variables only
    launch_solve(h);
    h->v_cnt = h->V;
    return status();
}

int sx_tile_transform(sx_handle *h) {
    clear_error();
    if (!h) { set_error("null handle"); return 1; }
    flush_diag(h);
    launch_zinv(h, true);
    launch_rl_inverse(h, true);
    return status();
}

int sx_physics(sx_handle *h, int32_t t) {
    clear_error();
    if (!h) { set_error("null handle"); return 1; }
    if (t < 1) { set_error("t is 1-based"); return 1; }
    launch_physics(h, t);
    return status();
}

int sx_advance(sx_handle *h, int32_t t) {
    clear_error();
    if (!h) { set_error("null handle"); return 1; }
    if (t < 1) { set_error("t is 1-based"); return 1; }
    launch_zinv(h, false);
    launch_inverse_and_physics(h, t);
    if (h->defer_diag) { h->v_cnt = h->V - 1; h->diag_dirty = true; }      // w's coefficients follow on demand (flush_diag)
    launch_fl_forward(h);
    launch_sb(h);
    h->v_cnt = h->V;
    return status();
}

}  // extern "C"

namespace sx {
void graphs_release(sx_handle *h) {
    for (auto &g : h->graph_exec) {
        if (g) hipGraphExecDestroy(g);
        g = nullptr;
    }
}

static void step_launches(sx_handle *h, int t) {          // = sx_advance + sx_spline_transform
    launch_zinv(h, false);
    launch_inverse_and_physics(h, t);
    if (h->defer_diag) { h->v_cnt = h->V - 1; h->diag_dirty = true; }
    launch_fl_forward(h);
    launch_sb(h);
    launch_solve(h);
    h->v_cnt = h->V;
}
}  // namespace sx

extern "C" {

int sx_step(sx_handle *h, int32_t t) {
    clear_error();
    if (!h) { set_error("null handle"); return 1; }
    if (t < 1) { set_error("t is 1-based"); return 1; }
    if (h->ncells != h->nc) { set_error("sx_step: one-tile patches only (tiles exchange between sx_advance and the solve)"); return 1; }
    // the first two steps a handle executes are always plain launches: Euler / AB2 arguments, and everything created lazily on a
    // first launch (work lists, elimination tables, function attributes) must exist before a capture, which may not allocate
    const bool graph = h->use_graph && !h->timers_on && t >= 3 && h->plain_steps >= 2 && h->eq != SX_EQ_NONE && !h->comm_state;
    if (!graph) {
        step_launches(h, t);
        h->plain_steps++;
        return status();
    }
    const int key = h->rot % 3;
    if (!h->graph_exec[key]) {
        // capture this rotation's launches.  The null stream cannot be captured: a handle that runs on it captures and replays on
        // a private BLOCKING stream, which the null stream's legacy semantics order against everything else the handle does
        hipStream_t user = h->stream;
        if (!user && !h->graph_stream && hipStreamCreate(&h->graph_stream) != hipSuccess) { h->use_graph = 0; step_launches(h, t); return status(); }
        hipStream_t cs = user ? user : h->graph_stream;
        hipGraph_t g = nullptr;
        hipGraphExec_t ex = nullptr;
        const int rot0 = h->rot;
        bool ok = hipStreamBeginCapture(cs, hipStreamCaptureModeThreadLocal) == hipSuccess;
        if (ok) {
            h->stream = cs;
            step_launches(h, t);
            h->stream = user;
            ok = hipStreamEndCapture(cs, &g) == hipSuccess && g && !error_status();
            if (ok) ok = hipGraphInstantiate(&ex, g, nullptr, nullptr, 0) == hipSuccess;
            if (g) hipGraphDestroy(g);
        }
        if (!ok) {          // no graphs on this handle from here on; redo the step with plain launches (nothing ran during the capture)
            (void)hipGetLastError();
            clear_error();
            h->use_graph = 0;
            h->rot = rot0;
            step_launches(h, t);
            return status();
        }
        h->graph_exec[key] = ex;
        h->rot = rot0;      // the capture advanced the host-side bookkeeping; the replay below does it again
    }
    hipStream_t ls = h->stream ? h->stream : h->graph_stream;
    HIPOK(hipGraphLaunch(h->graph_exec[key], ls));
    // what the launchers would have updated on the host
    h->rot = (h->rot + 2) % 3;
    if (h->defer_diag) h->diag_dirty = true;
    return status();
}

int sx_check_nan(sx_handle *h, int32_t *flag) {
    clear_error();
    if (!h || !flag) { set_error("null argument"); return 1; }
    launch_nan_check(h);
    int f = 0;
    HIPOK(hipMemcpyAsync(&f, h->d_flag, sizeof(int), hipMemcpyDeviceToHost, h->stream));
    HIPOK(hipStreamSynchronize(h->stream));
    *flag = f;
    return status();
}

int sx_max_abs(sx_handle *h, double *out) {
    clear_error();
    if (!h || !out) { set_error("null argument"); return 1; }
    // persistent device scratch (allocated on first use, freed with the handle): no hipMalloc / hipFree per call
    if (!h->d_maxabs && !dalloc(h, &h->d_maxabs, (size_t)h->V)) return 1;
    launch_max_abs(h, h->d_maxabs);
    std::vector<unsigned long long> bits(h->V);
    HIPOK(hipMemcpyAsync(bits.data(), h->d_maxabs, sizeof(unsigned long long) * h->V, hipMemcpyDeviceToHost, h->stream));
    HIPOK(hipStreamSynchronize(h->stream));
    for (int v = 0; v < h->V; v++) std::memcpy(&out[v], &bits[v], sizeof(double));
    return status();
}

int sx_tile_b_device(sx_handle *h, void **p, int64_t *rows, int64_t *cols) {
    clear_error();
    if (!h || !p) { set_error("null argument"); return 1; }
    flush_diag(h);
    *p = h->d_Btile;
    if (rows) *rows = h->nbt;
    if (cols) *cols = h->C;
    return 0;
}

int sx_bind_tile_b(sx_handle *h, void *p) {
    clear_error();
    if (!h) { set_error("null handle"); return 1; }
    h->d_Btile = p ? (double *)p : (h->d_Btile_own ? h->d_Btile_own : h->d_Bfull);
    return 0;
}

int sx_halo_add(sx_handle *h, const void *recv) {
    clear_error();
    if (!h || !recv) { set_error("null argument"); return 1; }
    launch_halo_add(h, (const double *)recv);
    return status();
}

int sx_bind_patch_b(sx_handle *h, const void *base, const int64_t *rowoff) {
    clear_error();
    if (!h) { set_error("null handle"); return 1; }
    std::vector<int64_t> ro(h->b_rDim);
    if (base && rowoff) {
        for (int m = 0; m < h->b_rDim; m++) ro[m] = rowoff[m];
        h->d_Bsrc = (const double *)base;
    } else {
        for (int m = 0; m < h->b_rDim; m++) ro[m] = (int64_t)m * h->C;
        h->d_Bsrc = h->d_Bfull;
    }
    HIPOK(hipMemcpy(h->d_rowoff, ro.data(), sizeof(int64_t) * ro.size(), hipMemcpyHostToDevice));
    return status();
}

int sx_patch_a_device(sx_handle *h, void **p, int64_t *rows, int64_t *cols) {
    clear_error();
    if (!h || !p) { set_error("null argument"); return 1; }
    flush_diag(h);
    *p = h->d_A;
    if (rows) *rows = h->b_rDim;
    if (cols) *cols = h->C;
    return 0;
}

// ---- transposed (all-to-all) patch solve ---------------------------------------------------------------------------
}  // extern "C"

namespace sx {
// The tile table every exchange protocol is configured from (calcTileSizes rows 4 and 3): entry `me` is this handle's tile,
// the tiles are contiguous, cover the patch and have at least 3 cells each (calcTileSizes' own rule, src/semiimplicit.jl:141-153;
// with fewer the 3 halo rows a tile sends would overlap the 3 rows that receive its predecessor's halo).
bool tile_table_ok(const sx_handle *h, int n, int me, const int32_t *cell0, const int32_t *ncells) {
    if (!h || !cell0 || !ncells || n < 1 || me < 0 || me >= n) { set_error("invalid tile table argument"); return false; }
    if (cell0[me] != h->cell0 || ncells[me] != h->ncells) { set_error("tile table does not match this handle"); return false; }
    int c = 0;
    for (int t = 0; t < n; t++) {
        if (cell0[t] != c || (ncells[t] < 3 && n > 1)) { set_error("tiles must be contiguous with at least 3 cells each"); return false; }
        c += ncells[t];
    }
    if (c != h->nc) { set_error("tiles do not cover the patch"); return false; }
    return true;
}
}  // namespace sx

extern "C" {
int sx_a2a_configure(sx_handle *h, int32_t n, int32_t me, const int32_t *cell0, const int32_t *ncells) {
    clear_error();
    if (!h || !cell0 || !ncells || n < 1 || me < 0 || me >= n) { set_error("invalid argument"); return 1; }
    if (!tile_table_ok(h, n, me, cell0, ncells)) return 1;
    h->a2a_n = n; h->a2a_me = me;
    h->a2a_cell0.assign(cell0, cell0 + n);
    h->a2a_ncells.assign(ncells, ncells + n);
    const int G = h->V * h->Zb;                               // column groups (variable, z-mode), K2 columns each
    std::vector<int> owner(G);
    std::vector<int64_t> cs(n + 1), cw(n), soff(n);
    for (int d = 0; d <= n; d++) cs[d] = (int64_t)((int64_t)G * d / n) * h->K2;
    for (int d = 0; d < n; d++) {
        cw[d] = cs[d + 1] - cs[d];
        for (int64_t g = cs[d] / h->K2; g < cs[d + 1] / h->K2; g++) owner[g] = d;
    }
    // tile-side buffers: [dest d][row j < nbt][cw[d]]
    int64_t o = 0;
    for (int d = 0; d < n; d++) { soff[d] = o; o += (int64_t)h->nbt * cw[d]; }
    h->a2a_colstart = cs;
    h->a2a_g0 = (int)(cs[me] / h->K2);
    h->a2a_g1 = (int)(cs[me + 1] / h->K2);
    // owner-side buffers: [tile t][row j < ncells[t] + 3][cw[me]]; row m of the patch lives in its owning tile and, for the
    // first three rows of tiles t >= 1, also in the previous tile (its halo rows)
    std::vector<int64_t> offA(h->b_rDim, 0), offB(h->b_rDim, -1), tbase(n);
    o = 0;
    for (int t = 0; t < n; t++) { tbase[t] = o; o += (int64_t)(ncells[t] + 3) * cw[me]; }
    for (int t = 0; t < n; t++) {
        const int owned = ncells[t] + (t == n - 1 ? 3 : 0);
        for (int j = 0; j < owned; j++) offA[cell0[t] + j] = tbase[t] + (int64_t)j * cw[me];
        if (t > 0)
            for (int j = 0; j < 3; j++) offB[cell0[t] + j] = tbase[t - 1] + (int64_t)(ncells[t - 1] + j) * cw[me];
    }
    auto up = [&](auto **p, const auto &v) {
        using T = typename std::remove_reference<decltype(v)>::type::value_type;
        if (!*p && !dalloc(h, p, v.size(), false)) return false;
        return hipMemcpy(*p, v.data(), sizeof(T) * v.size(), hipMemcpyHostToDevice) == hipSuccess;
    };
    std::vector<int64_t> csn(cs.begin(), cs.begin() + n);
    if (!up(&h->d_a2a_owner, owner) || !up(&h->d_a2a_soff, soff) || !up(&h->d_a2a_cw, cw) || !up(&h->d_a2a_cs, csn) ||
        !up(&h->d_a2a_offA, offA) || !up(&h->d_a2a_offB, offB)) {
        set_error("a2a table upload failed");
        return 1;
    }
    return status();
}

int sx_a2a_col_starts(sx_handle *h, int64_t *out) {
    clear_error();
    if (!h || !out || h->a2a_n < 1) { set_error("sx_a2a_configure has not been called"); return 1; }
    for (int d = 0; d <= h->a2a_n; d++) out[d] = h->a2a_colstart[d];
    return 0;
}

int sx_a2a_pack_b(sx_handle *h, void *dev_send) {
    clear_error();
    if (!h || !dev_send || h->a2a_n < 1) { set_error("invalid argument / not configured"); return 1; }
    launch_a2a_pack(h, (double *)dev_send, 0);
    return status();
}

int sx_a2a_solve(sx_handle *h, const void *dev_recv, void *dev_send) {
    clear_error();
    if (!h || !dev_recv || !dev_send || h->a2a_n < 1) { set_error("invalid argument / not configured"); return 1; }
    launch_solve_a2a(h, (const double *)dev_recv, (double *)dev_send);
    return status();
}

int sx_a2a_unpack_a(sx_handle *h, const void *dev_recv) {
    clear_error();
    if (!h || !dev_recv || h->a2a_n < 1) { set_error("invalid argument / not configured"); return 1; }
    launch_a2a_pack(h, (double *)const_cast<void *>(dev_recv), 1);
    return status();
}

int sx_enable_timers(sx_handle *h, int32_t on) {
    clear_error();
    if (!h) { set_error("null handle"); return 1; }
    if (!on) timers_flush(h);
    h->timers_on = on;
    return 0;
}

int sx_timer_only(sx_handle *h, const char *name) {
    clear_error();
    if (!h) { set_error("null handle"); return 1; }
    h->timer_only = name ? name : "";
    return 0;
}

int sx_reset_timers(sx_handle *h) {
    clear_error();
    if (!h) { set_error("null handle"); return 1; }
    timers_flush(h);
    for (auto &t : h->timers) { t.ms = 0; t.calls = 0; }
    return 0;
}

int sx_get_timers(sx_handle *h, int32_t max, const char **names, double *ms, int64_t *calls, int32_t *n) {
    clear_error();
    if (!h || !n) { set_error("null argument"); return 1; }
    timers_flush(h);
    int cnt = 0;
    for (auto &t : h->timers) {
        if (cnt >= max) break;
        if (names) names[cnt] = t.name;
        if (ms) ms[cnt] = t.ms;
        if (calls) calls[cnt] = t.calls;
        cnt++;
    }
    *n = cnt;
    return 0;
}

int sx_kernel_bytes(sx_handle *h, const char *name, double *bytes) {
    clear_error();
    if (!h || !name || !bytes) { set_error("null argument"); return 1; }
    // Algorithmic bytes per launch (fp64), counting each array once (DESIGN.md "Kernels and rooflines").
    // w = fp64; ws = width of the derivative slots of `physical` / G (4 bytes in the fp32-storage mode; value slots stay fp64)
    const double w = 8.0, ws = h->f32 ? 4.0 : 8.0, wi = h->sp32 ? 4.0 : 8.0, N = (double)h->N, V = h->V;     // wi: Az / Fl entries
    const double S_tile = (double)h->nbt * h->C, S_patch = (double)h->b_rDim * h->C;
    const bool fusedz = h->node_mode && h->node_active && fft_fused_zinv(h);       // node-space units invert vertically inside their FFT kernel
    const int zrows = h->last_zinv_rows > 0 || fusedz ? h->last_zinv_rows : h->nbt;
    const double az = h->has_z ? (double)zrows * h->last_zinv_jobs * h->nz * h->K2 : S_tile;
    const double fl = (double)h->nrings * h->V * h->nz * h->K2, bz = (double)h->nbt * h->V * h->nz * h->K2;
    std::string k(name);
    double b = 0;
    auto planes = [&](int bits, int val) { return w * val + ws * (bits - val); };        // bytes per point of a plane set
    const double out_planes = h->last_mask_full ? planes(h->mask_full_bits, h->mask_full_val) : planes(h->mask_eq_bits, h->mask_eq_val);
    const double eq_planes = planes(h->mask_eq_bits, h->mask_eq_val), node_planes = planes(h->mask_node_bits, h->mask_node_val);
    const bool node = h->node_mode && h->node_active;
    const double fin = node ? (double)h->R_in / h->nrings : 1.0;   // fraction of rings on the ring-wise path
    // node-space units actually transformed and read: cell c of the outer rings combines nodes c .. c + 3 and the first such cell is
    // R_in / 3, so nodes [R_in / 3, nbt) - 132 of 174 at the bench grid; the nodes below feed the ring-wise inner rings only
    const double n_units = node ? (double)(h->nbt - h->R_in / MUBAR) : 0.0;
    const double NGu = n_units * (double)h->uniform_L * h->nz;
    if (k == "k_rl_inverse") b = (N * out_planes + wi * az) * fin;   // write the requested physical planes, read Az
    else if (k == "k_node_fft") b = NGu * node_planes + (fusedz ? w * n_units * h->C : wi * n_units * h->last_zinv_jobs * h->nz * h->K2);
    else if (k == "k_phys_hrbl_inner") b = N * fin * (eq_planes + w * (4.0 * V - 3.0));
    else if (k == "k_phys_hrbl" && node)                            // node transforms (read once) + history + outputs
        b = NGu * node_planes + N * (1.0 - fin) * w * (4.0 * V - 3.0);
    else if (k == "k_zinv") b = w * S_tile * zrows / h->nbt + wi * az;
    else if (k == "k_phys_pointwise" || k == "k_phys_hrbl") {
        // read the requested slots, E_nm1, E_nm2; write E_n, var_np1; the SW sets also write the diagnostic w plane and
        // keep no tendency history for it
        const bool sw = (h->eq == SX_EQ_ONEWAY_SW_SLAB || h->eq == SX_EQ_TWOWAY_SW_SLAB || h->eq == SX_EQ_ONEWAY_SW_HRBL);
        b = N * (eq_planes + w * (4.0 * V + (sw ? -3.0 : 0.0)));        // inside sx_advance the diagnostic w plane is not stored
    }
    else if (k == "k_fl_forward") b = w * N * V + wi * fl;
    else if (k == "k_sb") b = w * (fl + bz);
    else if (k == "k_sbz") b = wi * fl + w * S_tile;
    else if (k == "k_solve") b = w * 4.0 * S_patch;                 // read B, write y, read y, write A
    else if (k == "k_semiimplicit") b = w * N * 2.0 * 5.0;
    else if (k == "k_rz_inverse") b = w * S_tile + N * out_planes;       // read the tile's A rows, write the requested physical planes
    else if (k == "k_rz_forward") b = w * N * V + w * S_tile;            // read var_np1, write the tile's B rows
    *bytes = b;
    return 0;
}

}  // extern "C"
