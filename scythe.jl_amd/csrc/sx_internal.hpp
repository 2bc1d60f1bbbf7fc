// Internal declarations of libscythe_hip.so (not part of the ABI).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>
#include <string>
#include <utility>
#include <vector>
#include "scythe_hip.h"

namespace sx {

constexpr int DFT_KMAX_SINGLE = 319;   // largest kmax whose coefficient sets fit the LDS beside the twiddle table (sx_dft.hip)
constexpr int MUBAR = 3;   // CubicBSpline.mubar: mish points per cell (src/spectralGrid.jl:24)

// ---- host-side operator construction (sx_setup.cpp) ---------------------------------------------------------------
struct SplineClass {
    int bcl = 0, bcr = 0;
    int nfree = 0, periodic = 0, rl = 0, rr = 0;
    double gl[3][2] = {}, gr[3][2] = {};   // dependent boundary coefficients in terms of the first/last two free ones
    std::vector<double> Lband;             // [b_rDim][4]  L[i][i-3..i]
    std::vector<double> Larrow;            // [3][b_rDim]  last three rows of L (periodic only)
    // kept for the interface-only (partitioned) patch solve, which factors the diagonal blocks of the tiles (sx_iface.hip)
    std::vector<double> Mdense;                             // [nfree][nfree]  Gamma (P + eps_q Q) Gamma^T before factoring
    std::vector<std::vector<std::pair<int, double>>> Gam;   // Gamma as sparse rows: free unknown -> (patch row, weight)
};

struct ChebOps {
    int bcb = 0, bct = 0;
    double zmin = 0, zmax = 0;
    std::vector<double> z;       // [nz] gridpoints, index 0 = bottom
    std::vector<double> T;       // [nz][nz]  coefficients -> values              (Chebyshev.dct_matrix)
    std::vector<double> Dc;      // [nz][nz]  coefficient-space d/dz
    std::vector<double> CB;      // [Zb][nz]  values -> truncated b
    std::vector<double> CA;      // [nz][Zb]  b -> a (padding + BC projection)
    std::vector<double> M[3];    // [nz][Zb]  b -> value, d/dz, d2/dz2
    std::vector<double> Mint;    // [nz][nz]  values -> (CB, CA, CIInt) integral from the bottom
    std::vector<double> Mdz;     // [nz][nz]  values -> (CB, CA, CIx) truncated derivative
    std::vector<double> Mrec;    // [nz][nz]  values -> (CB, CA, CI) truncated reconstruction
    std::vector<double> Mdzz;    // [nz][nz]  values -> (CB, CA, CIxx) truncated second derivative
};

// Parallel-cyclic-reduction form of a spline class's solve a = Gamma^T (Gamma (P + eps_q Q) Gamma^T)^-1 Gamma b (sx_pcr.hip): the free
// unknowns in blocks of 3 (the half-bandwidth), block-tridiagonal; level l combines block row i with rows i -+ 2^l,
//   r_i <- r_i - alpha[l][i] r_{i - 2^l} - gamma[l][i] r_{i + 2^l},
// with the elimination blocks alpha = L D^-1, gamma = U D^-1 of the CONSTANT matrix worked out once, here, in extended precision;
// after `levels` steps the system is block diagonal (x_i = dinv[i] r_i).  PERIODIC classes: the band part M_b by the same
// elimination, the corner blocks by a rank-6 correction x = y - G (E^T y) over the first and last three unknowns.
struct PcrTables {
    int n = 0, nblk = 0, levels = 0, periodic = 0;
    std::vector<double> coef;     // [levels][nblk][18]  alpha (3 x 3 row-major), gamma (3 x 3)
    std::vector<double> dinv;     // [nblk][9]
    std::vector<int> gin_row;     // [3 nblk][4]  patch rows m whose b enters free unknown j (Gamma b), -1 = none
    std::vector<double> gin_w;    // [3 nblk][4]
    std::vector<int> gout_j;      // [nb][2]      free unknowns that make patch row m (Gamma^T x), -1 = none
    std::vector<double> gout_w;   // [nb][2]
    std::vector<double> G;        // periodic: [3 nblk][6]
};
bool build_pcr_tables(const SplineClass &sc, int nb, PcrTables &out, std::string &err);
// the same arithmetic the kernel applies (double, level by level), on the host: b / a are patch rows [nb]
void pcr_apply_host(const PcrTables &t, int nb, const double *b, double *a);
// the serial statement of the same solve (banded Cholesky factors of build_spline_class), on the host
void cholesky_apply_host(const SplineClass &sc, int nb, const double *b, double *a);
void basis_tables(double DX, double phi[4][MUBAR][4]);
void quad_weights(double DX, double w[MUBAR]);
int bc_rank(int bc);
bool build_spline_class(int nc, double DX, double l_q, int bcl, int bcr, SplineClass &out, std::string &err);
bool build_cheb_ops(double zmin, double zmax, int nz, int Zb, int bcb, int bct, ChebOps &out, std::string &err);
// Helmholtz operator of calc_Helmholtz_semiimplicit_matrix (src/semiimplicit.jl:768-781) folded with the
// collocation matrices:  Wmat = T H^-1,  Xmat = T Dc H^-1   (both [nz][nz], acting on the shifted right-hand side)
bool build_helmholtz(const ChebOps &w, double pxi_bar, double tau, std::vector<double> &Wmat, std::vector<double> &Xmat,
                     std::string &err);
void ring_table(int has_l, int uniform_L, int ri /*1-based patch ring*/, int &L, int &kmax, double &off);

// ---- device-side tables handed to the kernels -----------------------------------------------------------------------
// `physical` [slot][v][N] and the node-space transforms G [slot][v][NG] as the kernels see them: the VALUE slot (slot 0)
// is always fp64 - it is time-stepping state - while the derivative slots 1..D-1 are ST = double, or float in the
// fp32-storage mode (sx_grid_desc.storage_f32).  For ST = double `der` = `val` + V * N: the plain [D][V][N] array.
template <class ST>
struct Planes {
    double *val;          // [V][n]
    ST *der;              // [D-1][V][n]
};
template <class ST>
__host__ __device__ inline Planes<ST> planes_of(double *base, int V, int64_t n) {
    return Planes<ST>{base, reinterpret_cast<ST *>(base + (int64_t)V * n)};
}

struct ColJob {
    int64_t in_off, out_off, mat_off;
};

// semiimplicit_adjustment (src/semiimplicit.jl:521-597): what its kernels need
struct SemiArgs {
    double *np1;
    const double *In, *I1, *I2;
    const double *MrecT, *MdzT, *WT, *XT;
    int64_t N;
    int nz, t, wi, xi;
    double ts, tau, pxi;
};

struct Timer {
    const char *name;
    double ms = 0.0;
    int64_t calls = 0;
};

struct PendingEvent {
    int timer;
    hipEvent_t a, b;
};

}  // namespace sx

struct sx_handle {
    // geometry
    int geom = 0, has_l = 0, has_z = 0;
    double xmin = 0, xmax = 0, DX = 0, l_q = 2.0, zmin = 0, zmax = 0;
    int nc = 0, V = 0, D = 0, ncoord = 1, rDim = 0, b_rDim = 0, nz = 1, Zb = 1, nsz = 1;
    int cell0 = 0, ncells = 0, nrings = 0, nbt = 0, tile_num = 0, uniform_L = 0;
    int kDim = 0, K2 = 1, kDim_t = 0, K2t = 1, kmax_max = 0, L_max = 1;   // K2: device row width (padded, see sx_kernels.hip)
    int K2ref = 1;                                                         // reference block count 1 + 2 kDim
    int64_t N = 0, Nh = 0, C = 0, S_patch = 0, S_tile = 0;
    int slot[7] = {-1, -1, -1, -1, -1, -1, -1};
    std::vector<int> hL, hkmax;
    std::vector<double> hoff;
    std::vector<int64_t> hpstart;
    std::vector<int> bcl, bcl0, bcr, bcb, bct;
    // model
    double ts = 0;
    int eq = SX_EQ_NONE, semi = 0, w_index = 0, xi_index = 0, col_var = 0;
    double par[SX_NPARAMS] = {};
    int rot = 0;   // rotation of the expdot / impdot history buffers
    // device
    hipStream_t stream = nullptr;
    double *d_A = nullptr, *d_Bfull = nullptr, *d_Btile = nullptr, *d_Btile_own = nullptr;
    const double *d_Bsrc = nullptr;
    int64_t *d_rowoff = nullptr, *d_aoff = nullptr, *d_neg1 = nullptr;
    // transposed (all-to-all) solve
    int a2a_n = 0, a2a_me = 0, a2a_g0 = 0, a2a_g1 = 0;
    std::vector<int64_t> a2a_colstart;          // [n + 1] column (not group) starts
    std::vector<int> a2a_cell0, a2a_ncells;
    int *d_a2a_owner = nullptr;
    int64_t *d_a2a_soff = nullptr, *d_a2a_cw = nullptr, *d_a2a_cs = nullptr, *d_a2a_offA = nullptr, *d_a2a_offB = nullptr;
    // second stream for the inner-ring chain of sx_advance (launch_inverse_and_physics); off unless SX_OVERLAP=1: measured
    // +2.6 % (669 vs 653 steps/s) - the two chains compete for the same HBM bandwidth - and it blurs the per-kernel timers
    hipStream_t stream2 = nullptr;
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    int overlap = 0;
    // native-ring DFT launches (sx_dft.hip): (ring, variable) work items, most expensive first; [0] inverse with the
    // equation-set slot mask, [1] inverse with every slot, [2] forward
    int *d_dft_items[3] = {nullptr, nullptr, nullptr};
    int n_dft_items[3] = {0, 0, 0};
    int n_dft_big[3] = {0, 0, 0};            // of which (listed first) rings with kmax > DFT_KMAX_SINGLE: chunked kernels
    int dft_lcap_small = 0, dft_kcap_small = 0;   // largest ring length / kmax among the other rings
    // hipGraph replay of a one-tile step (sx_step; SX_GRAPH=1): one instantiated graph per history rotation
    int use_graph = 0, plain_steps = 0;
    hipGraphExec_t graph_exec[3] = {nullptr, nullptr, nullptr};
    hipStream_t graph_stream = nullptr;      // capture / replay stream when the handle runs on the (uncapturable) null stream
    int fft_reg = 1;                         // 256-point inverse transforms: register-resident passes with lane swaps (SX_FFT_REG=0: every pass through LDS)
    int dft_half_wg = 1;                     // eighth-wave kernel as two 256-thread workgroups per CU where one set + half a twiddle table fit 80 KB (SX_DFT_HALFWG=0: one 512-thread workgroup)
    int dft_eighth = 2;                      // merged kernel: eighth-wave units of two planes (round 4; SX_DFT_EIGHTH=0: quarter-wave units of up to four)
    int dft_merge = 1;                       // RLZ native rings: merged-pass inverse DFT kernel (SX_DFT_MERGE=0: one set per pass, whole tiles per wave)
    int rl_quarter = 1;                      // RL grids: quarter-wave DFT kernels over one work list (SX_DFT_RLQ=0: the half-ring kernels in two ring classes)
    int *d_rlq_items[2] = {nullptr, nullptr};    // (ring, part) items of the RL inverse / forward launch, most expensive first
    int n_rlq_items[2] = {0, 0};
    // SX_DEFER_DIAG=1 (one-tile HRBL runs on the FFT path): the diagnostic variable w is written by the equation set before it is
    // read (src/shallowWaterModels.jl:69, 430), so its spline coefficients are consumed by OUTPUT only; sx_advance then sends
    // the five prognostic variables through the forward transform and the solve, and w's follow on demand (flush_diag) when
    // something reads A or B.  v_lo / v_cnt: the variable window the forward-path launchers cover.
    int defer_diag = 0, v_lo = 0, v_cnt = 0;
    bool diag_dirty = false;
    int fuse_zinv = 0;      // SX_FUSE_ZINV=1: vertical inverse inside the node FFT kernel (measured slower: sx_fft.hip)
    int sbw_mfma = 1;       // k_sbw_mfma (matrix-core vertical contraction, operator in registers) for zDim 64 / 32 (SX_SBW_MFMA=0: k_sbw)
    int sbw_prefetch = 0;   // k_sbw requests the next cell's ring spectra before contracting the current node (SX_SBW_PF=0: off)
    int wide = 1;    // 16-byte-per-lane loads / stores in the equation-set kernels (SX_WIDE=0: the 8-byte forms, A/B timing)
    std::vector<int> hmask_full, hmask_eq;       // host copies of d_mask_full / d_mask_eq
    bool in_advance = false;    // set while sx_advance launches the equation set (the diagnostic w plane is then not stored)
    bool L_all_mult4 = false;   // every ring length is a multiple of 4 (native rings are): the MFMA DFT kernels apply
    double *d_ref = nullptr;    // ReferenceState [3][3][nz] (Euler_test)
    int f32 = 0;   // fp32 storage of the derivative slots of d_phys / d_G (typed by the launchers)
    int sp32 = 0;  // storage_f32 = 2: the spectral transform intermediates d_Az and d_Fl are fp32 as well (fp64 accumulation)
    double *d_Az = nullptr, *d_phys = nullptr, *d_np1 = nullptr, *d_E[3] = {}, *d_I[3] = {};
    double *d_Fl = nullptr;
    double *d_phi = nullptr, *d_wq = nullptr;
    int *d_L = nullptr, *d_kmax = nullptr;
    int64_t *d_pstart = nullptr, *d_twoff = nullptr, *d_phoff = nullptr;
    double2 *d_tw = nullptr, *d_ph = nullptr;
    double *d_MzT = nullptr;    // operators of d_Mz transposed: [v][sz][Zb][nz]
    double *d_Mz = nullptr, *d_CB = nullptr, *d_MintT = nullptr, *d_MdzT = nullptr, *d_MrecT = nullptr;
    double *d_WT[2] = {}, *d_XT[2] = {};
    double tau[2] = {0, 0};
    int *d_cls = nullptr, *d_cmeta = nullptr;   // cmeta [ncls][4] = nfree, periodic, rl, rr
    double *d_gl = nullptr, *d_gr = nullptr, *d_Lband = nullptr, *d_Larrow = nullptr, *d_Ldinv = nullptr;
    double *d_r = nullptr, *d_cosl = nullptr, *d_sinl = nullptr, *d_z = nullptr;
    int *d_flag = nullptr;
    unsigned long long *d_maxabs = nullptr;   // [V] scratch of sx_max_abs
    void *comm_state = nullptr;               // RCCL exchange state (sx_comm.cpp)
    void *iface_state = nullptr;              // interface-only patch solve (sx_iface.hip)
    void *pcr_state = nullptr;                // parallel-cyclic-reduction tables and launch lists (sx_pcr.hip)
    int solve_pcr = -1;                       // SX_SOLVE_PCR read at sx_create: 0 never, 1 wherever the tables exist, -1 by column count
    int64_t pcr_maxcols = 16384;              // SX_PCR_MAXCOLS
    int rz_fused = 1;                         // RZ grids: the fused radius-on-the-matrix-cores kernels of sx_rz.hip (SX_RZ_FUSED=0: the general kernels)
    double *d_CBT = nullptr;                  // CB transposed [nz][Zb] (sx_rz.hip)
    int semi_mfma = 1;                        // semi-implicit column operators on the matrix cores (SX_SEMI_MFMA=0: k_semiimplicit)
    std::vector<sx::SplineClass> classes;     // host copies of the spline classes (d_cls indexes them)
    std::vector<int> hcls;                    // host copy of d_cls: [v][2] -> class of (k = 0, k >= 1)
    int *d_mask_full = nullptr, *d_mask_eq = nullptr;   // per-variable bit masks of derivative slots to produce
    int mask_eq_bits = 0, mask_full_bits = 0;            // total number of (variable, slot) planes in each mask
    int mask_eq_val = 0, mask_full_val = 0, mask_node_val = 0;   // of which value-slot planes (always fp64)
    bool last_mask_full = true;
    // node-space ("radial last") inverse for uniform rings: rings [0, R_in) keep the ring-wise path (their wavenumber
    // truncation depends on the ring), rings [R_in, nrings) are evaluated from node-space transforms inside the equation set
    int node_mode = 0, node_active = 0, R_in = 0, mask_node_bits = 0;
    int64_t NG = 0;
    double *d_G = nullptr, *d_nphi = nullptr;
    int *d_nkmax = nullptr, *d_mask_node = nullptr;
    int64_t *d_npstart = nullptr, *d_nphoff = nullptr;
    sx::ColJob *d_jobs_zinv_full = nullptr, *d_jobs_zinv_eq = nullptr, *d_jobs_zf = nullptr;
    int njobs_zinv_full = 0, njobs_zinv_eq = 0, last_zinv_jobs = 0, last_zinv_rows = 0;
    int ncls = 0;
    size_t dev_bytes = 0;
    std::vector<void *> allocs;
    // timers
    int timers_on = 0;
    bool timer_skip = false;          // the launch in progress is not timed (sx_timer_only)
    std::string timer_only;           // when set, only this kernel gets its event pair
    std::vector<sx::Timer> timers;
    std::vector<sx::PendingEvent> pending;
    std::vector<hipEvent_t> event_pool;
};

namespace sx {
// kernel launchers (sx_kernels.hip); each returns hipError_t from the launch
void launch_zinv(sx_handle *h, bool full);
void launch_rl_inverse(sx_handle *h, bool full);
bool fft_path_ok(const sx_handle *h);
bool fft_fused_zinv(const sx_handle *h);
bool dft_mfma_ok(const sx_handle *h);
void launch_rl_inverse_dft(sx_handle *h, const int *d_mask);
void launch_fl_forward_dft(sx_handle *h);
void launch_rl_inverse_fft(sx_handle *h, const int *d_mask, int n_rings = -1);
void launch_node_fft(sx_handle *h);
void launch_fl_forward_fft(sx_handle *h);
void launch_physics(sx_handle *h, int t);
void launch_inverse_and_physics(sx_handle *h, int t);
void launch_copy_slot0(sx_handle *h);
void launch_fl_forward(sx_handle *h);
void launch_sb(sx_handle *h);
void launch_solve(sx_handle *h);
void launch_solve_a2a(sx_handle *h, const double *recv, double *send);
void launch_a2a_pack(sx_handle *h, double *buf, int unpack);
void launch_halo_add(sx_handle *h, const double *recv);
void launch_nan_check(sx_handle *h);
void launch_max_abs(sx_handle *h, unsigned long long *d_out);
int timer_id(sx_handle *h, const char *name);
void timer_begin(sx_handle *h, int id);
void timer_end(sx_handle *h);
void timers_flush(sx_handle *h);
void set_error(const std::string &msg);
void clear_error();
int error_status();   // 1 if set_error has been called since the last clear_error
void comm_release(sx_handle *h);
void flush_diag(sx_handle *h);
void graphs_release(sx_handle *h);
void iface_release(sx_handle *h);
void pcr_release(sx_handle *h);
bool rz_fused(const sx_handle *h);
void launch_rz_inverse(sx_handle *h, const int *d_mask);
void launch_rz_forward(sx_handle *h);
void launch_semi_mfma(sx_handle *h, const SemiArgs &a);
bool pcr_wanted(sx_handle *h, int64_t ncols);
void launch_solve_pcr(sx_handle *h, bool linear, const double *Bsrc, const int64_t *boffA, const int64_t *boffB, double *A,
                      const int64_t *aoffA, const int64_t *aoffB, int vz0, int ng, int64_t stride);
bool tile_table_ok(const sx_handle *h, int n, int me, const int32_t *cell0, const int32_t *ncells);
#ifdef SX_PHASES
void phases_dump();
void fft_phases_dump();
void sbw_phases_dump();
void dft_phases_dump();
#endif
}  // namespace sx
