/*
 * scythe_hip.h — C ABI of libscythe_hip.so: the MI355X (gfx950) implementation of Scythe.jl's per-time-step
 * spectral-transform hot path.  Plain pointers and sizes only; no torch / C++ types cross this boundary.
 *
 * The reference (Julia) has no FFI for this path; the seam is the pair of remote calls the master issues per step
 * (SURVEY.md 8(b)).  Every entry point cites the reference call it replaces (paths relative to the reference tree).
 * The Julia-side `ccall` glue a maintainer would add is shown in INTEGRATION.md.
 *
 * Conventions
 *   - every function returns 0 on success, non-zero on error; sx_last_error() gives the message
 *     (the reference signals errors with Julia exceptions: src/Scythe.jl:40, src/semiimplicit.jl:745,
 *      src/spectralGrid.jl:88-91).
 *   - host arrays use the reference's own layouts (column-major as in Julia):
 *        physical[point, var, deriv]   point = (ring-major, lambda, z innermost)   src/semiimplicit.jl:48, 64
 *        spectral[index, var]          index = (z-mode, wavenumber block, radial node), node fastest
 *     host pointers are borrowed for the duration of the call only.
 *   - one handle per tile (= per worker process / GPU, src/semiimplicit.jl:179-184); calls on one handle are serial.
 *   - t is the 1-based step counter of model_loop (src/semiimplicit.jl:268); it selects Euler/AB2/AB3.
 */
#ifndef SCYTHE_HIP_H
#define SCYTHE_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SX_ABI_VERSION 2

/* geometry  (GridParameters.geometry, src/spectralGrid.jl:21, 63-94) */
enum { SX_GEOM_R = 0, SX_GEOM_RZ = 1, SX_GEOM_RL = 2, SX_GEOM_RLZ = 3 };

/* radial / vertical boundary-condition codes (CubicBSpline.* / Chebyshev.* Dicts, models/ *.jl) */
enum { SX_BC_R0 = 0, SX_BC_R1T0 = 1, SX_BC_R1T1 = 2, SX_BC_R1T2 = 3, SX_BC_R2T10 = 4, SX_BC_R2T20 = 5, SX_BC_R3 = 6,
       SX_BC_PERIODIC = 7 };

/* equation sets, selected by name in the reference (src/semiimplicit.jl:357-363) */
enum {
    SX_EQ_LINEAR_ADVECTION_1D = 0,   /* src/testModels.jl:1-20   */
    SX_EQ_LINEAR_ADVECTION_RZ = 1,   /* src/testModels.jl:22-45  */
    SX_EQ_LINEAR_ADVECTION_RL = 2,   /* src/testModels.jl:47-73  */
    SX_EQ_LINEAR_ADVECTION_RLZ = 3,  /* src/testModels.jl:75-98  */
    SX_EQ_ONEWAY_SW_SLAB = 4,        /* src/shallowWaterModels.jl:1-113   */
    SX_EQ_TWOWAY_SW_SLAB = 5,        /* src/shallowWaterModels.jl:115-233 */
    SX_EQ_ONEWAY_SW_HRBL = 6,        /* src/shallowWaterModels.jl:346-511 */
    SX_EQ_LINEAR_ACOUSTIC_RZ = 7,    /* synthetic vehicle for semiimplicit_adjustment (src/semiimplicit.jl:521-597);
                                        Euler_test's variable layout and implicit terms (src/testModels.jl:188-205)
                                        with a linearised pressure-gradient force */
    SX_EQ_EULER_TEST = 8,            /* src/testModels.jl:100-215: moist Euler RZ (s, xi, mu, u, w) about a reference state,
                                        semi-implicit in the vertical acoustic terms; needs sx_model_desc.ref_state */
    SX_EQ_NONE = 99                  /* transforms only: sx_advance copies physical[:, :, 1] into var_np1 */
};

/* physical_params, fixed positions (model.physical_params Dict, models/ *.jl) */
enum { SX_P_G = 0, SX_P_K, SX_P_CD, SX_P_HFREE, SX_P_HB, SX_P_F, SX_P_S1, SX_P_C0, SX_P_KH, SX_P_UM, SX_P_VM,
       SX_P_PXI_BAR, SX_NPARAMS };

/* GridParameters flattened (src/spectralGrid.jl:20-45; tile construction src/semiimplicit.jl:155-169).
 * The patch fields describe the whole domain; the tile fields select this handle's radial range. */
typedef struct sx_grid_desc {
    int32_t abi_version;      /* SX_ABI_VERSION */
    int32_t geometry;         /* SX_GEOM_* */
    double xmin, xmax;        /* patch extent */
    int32_t num_cells;        /* patch cells; rDim = 3*num_cells, b_rDim = num_cells + 3 */
    double l_q;               /* spline filter length (default 2.0) */
    int32_t nvars;
    const int32_t *bcl;       /* [nvars] patch left BC for wavenumbers k >= 1 (and for R / RZ grids) */
    const int32_t *bcl_k0;    /* [nvars] patch left BC for wavenumber 0, or NULL = same as bcl */
    const int32_t *bcr;       /* [nvars] patch right BC */
    double zmin, zmax;
    int32_t zDim, b_zDim;     /* b_zDim <= 0 selects min(zDim, floor((2 zDim - 1) / 3) + 1) */
    const int32_t *bcb;       /* [nvars] bottom BC or NULL = R0 */
    const int32_t *bct;       /* [nvars] top BC or NULL = R0 */
    int32_t ring_uniform_L;   /* 0: native rings (4 + 4 ri points, kmax = ri); L > 0: every ring has L points,
                                 kmax = min(ri, L/2 - 1), zero phase offset (SURVEY.md 8(d) "perf shape") */
    int32_t tile_cell0;       /* first patch cell of this tile  (= spectralIndexL - 1) */
    int32_t tile_num_cells;   /* cells in this tile */
    int32_t tile_num;         /* informational (GridParameters.tile_num) */
    int32_t storage_f32;      /* 0: everything fp64 (the reference's arithmetic, `const real = Float64`,
                                 src/spectralGrid.jl:12).  1: the DERIVATIVE slots of `physical` (and of the node-space
                                 transforms) are stored as fp32; the value slot, the arithmetic, every spectral array, the
                                 banded solve and the time-stepping state stay fp64, so no fp32 rounding ever enters the
                                 model state directly - only the tendencies see it.
                                 2: additionally the SPECTRAL transform intermediates - the vertically inverted coefficients
                                 (between the Chebyshev and the Fourier stage of tileTransform!) and the ring spectra (between
                                 the Fourier and the B-spline stage of spectralTransform!) - are stored as fp32, every sum still
                                 accumulated in fp64: "fp32 mixed-precision transforms".  Rounding is relative to each spectral
                                 coefficient (6e-8), so the derivative operators do not amplify it, but it enters the state
                                 once per step.  Uniform power-of-two ring tables with zDim 32 / 64 / 128 only. */
} sx_grid_desc;

/* ModelParameters subset the step needs (src/Scythe.jl:8-21) */
typedef struct sx_model_desc {
    double ts;
    int32_t equation_set;     /* SX_EQ_* */
    int32_t semiimplicit;     /* options[:semiimplicit] */
    const double *params;     /* [SX_NPARAMS] */
    int32_t w_index, xi_index;/* 1-based variable indices of "w" and "xi" (semi-implicit only), 0 = absent */
    int32_t col_var;          /* 1-based variable whose vertical BCs the column operators of HRBL use ("h",
                                 src/shallowWaterModels.jl:423), 0 = variable 1 */
    const double *ref_state;  /* ReferenceState (src/reference_state.jl:4-10) for Euler_test, or NULL:
                                 [3][3][zDim] = (sbar, xibar, mubar) x (value, d/dz, d2/dz2) x level (0 = bottom);
                                 Pxi_bar travels in params[SX_P_PXI_BAR] */
} sx_model_desc;

typedef struct sx_dims {
    int64_t n_points;         /* tile gridpoints N (incl. z) */
    int64_t n_hpoints;        /* horizontal points (N / zDim) */
    int32_t n_vars, n_derivs, n_coord;  /* V, D, columns of getGridpoints */
    int32_t rDim, b_rDim;     /* patch */
    int32_t tile_rDim, tile_b_rDim;
    int32_t zDim, b_zDim;
    int32_t kDim, n_blocks;   /* patch kDim, 1 + 2 kDim */
    int32_t tile_kDim, tile_n_blocks;
    int64_t s_patch;          /* patch spectral entries per variable */
    int64_t s_tile;           /* tile spectral entries per variable (reference tile layout) */
    int64_t n_cols;           /* internal: V * b_zDim * n_blocks columns of the [node][col] spectral arrays */
} sx_dims;

typedef struct sx_handle sx_handle;

/* --- lifetime -------------------------------------------------------------------------------------------------- */
/* createGrid(GridParameters(...)) + createModelTile  (src/semiimplicit.jl:155-169, 44-124) */
int sx_create(const sx_grid_desc *grid, const sx_model_desc *model, sx_handle **out);
int sx_destroy(sx_handle *h);
const char *sx_last_error(void);
int sx_abi_version(void);
/* getfield(Scythe, Symbol(equation_set)) (src/semiimplicit.jl:359-361): name -> SX_EQ_*, -1 if not in scope */
int sx_equation_set_id(const char *name);
int sx_get_dims(const sx_handle *h, sx_dims *out);
/* launch on this hipStream_t (NULL = default stream) */
int sx_set_stream(sx_handle *h, void *hip_stream);
int sx_synchronize(sx_handle *h);

/* --- geometry ---------------------------------------------------------------------------------------------------- */
/* getGridpoints(tile) (src/semiimplicit.jl:59): out[n_points, n_coord] column-major; columns r[, lambda][, z] */
int sx_get_gridpoints(const sx_handle *h, double *out);
/* calcTileSizes(patch, n) (src/semiimplicit.jl:141): out[5, n] column-major rows = xmin, xmax, num_cells,
 * spectralIndexL, gridpoint count.  Pure host helper, no handle needed. */
int sx_calc_tile_sizes(const sx_grid_desc *patch, int32_t n_tiles, double *out);

/* Chebyshev column operators (Springsteel CBtransform! -> CAtransform! -> CItransform! / CIxtransform / CIxxtransform /
 * CIInttransform(C0 = 0); call sites src/reference_state.jl:95-117, 143-167, src/shallowWaterModels.jl:426-429) as dense
 * [zDim x zDim] row-major collocation matrices acting on the values of one column (index 0 = bottom), and the levels z.
 * Pure host helper for one-time set-up work such as the reference state; any output pointer may be NULL. */
int sx_cheb_column_ops(double zmin, double zmax, int32_t zDim, int32_t b_zDim, int32_t bcb, int32_t bct, double *z,
                       double *rec, double *dz, double *dzz, double *integ);

/* splineTransform!'s arithmetic for ONE right-hand side on the host, two ways: through the parallel-cyclic-reduction tables the
 * device kernel k_solve_pcr applies (csrc/sx_pcr.hip: elimination blocks of the constant matrix Gamma (P + eps_q Q) Gamma^T worked
 * out once in extended precision, applied level by level in double) and through the banded Cholesky factors the serial kernel
 * k_solve applies.  b, a_pcr, a_chol: [num_cells + 3] patch rows of one spectral column; levels: reduction levels of the tables.
 * Pure host helper (no handle, no device): it validates the table construction where there is no GPU; never on the step path. */
int sx_spline_solve_check(int32_t num_cells, double xmin, double xmax, double l_q, int32_t bcl, int32_t bcr, const double *b,
                          double *a_pcr, double *a_chol, int32_t *levels);

/* --- state in / out (host pointers, reference layouts) ------------------------------------------------------------- */
/* read_physical_grid -> physical[:, v, 1]  (src/semiimplicit.jl:134): values[n_points, n_vars] */
int sx_set_physical_values(sx_handle *h, const double *values);
/* tile.physical (src/semiimplicit.jl:305, 290): out[n_points, n_vars, n_derivs] */
int sx_get_physical(sx_handle *h, double *out);
/* mtile.var_np1 (src/semiimplicit.jl:21, 731): out[n_points, n_vars] */
int sx_get_var_np1(sx_handle *h, double *out);
/* tile.spectral after calcTendency (src/semiimplicit.jl:323): out[s_tile, n_vars] B coefficients */
int sx_get_tile_spectral(sx_handle *h, double *out);
/* sharedSpectral -> device (src/semiimplicit.jl:285 input): shared[s_patch, n_vars] B coefficients */
int sx_set_patch_spectral_b(sx_handle *h, const double *shared);
/* mtile.patchSpectral (src/semiimplicit.jl:289): out[s_patch, n_vars] A coefficients */
int sx_get_patch_spectral_a(sx_handle *h, double *out);
int sx_set_patch_spectral_a(sx_handle *h, const double *a);

/* calcPatchMap / calcHaloMap (src/semiimplicit.jl:79-86) as 1-based linear indices into ONE variable's column of the
 * reference layouts (add (v - 1) * s_patch resp. (v - 1) * s_tile for variable v):
 *   patch_owned[i] <-> tile_owned[i]   sharedSpectral[patchIndexMap] .= tileView            (:323)   n_owned entries
 *   patch_halo[i]  <-> tile_halo[i]    put!(haloSend, haloSendView) on this tile (:320);  on the NEXT tile
 *                                      sharedSpectral[haloReceiveIndexMap] .+= buffer (:326-329) with haloReceiveIndexMap =
 *                                      this tile's patch_halo                                         n_halo entries
 * The last tile owns all its rows (n_halo = 0).  Any output pointer may be NULL. */
int sx_index_map_sizes(const sx_handle *h, int64_t *n_owned, int64_t *n_halo);
int sx_index_maps(const sx_handle *h, int64_t *patch_owned, int64_t *tile_owned, int64_t *patch_halo, int64_t *tile_halo);

/* --- restart state (the reference can only restart from a physical_out CSV, which loses the
 * Adams-Bashforth history; this blob lets a run continue bit-identically) ------------------------------------------- */
/* The tile's A coefficients (its own radial nodes) and the tendency history expdot_nm1 / expdot_nm2 (and impdot_nm1 / nm2
 * when semi-implicit) as of the last completed step, in the library's device layout: an opaque blob for a handle created
 * from the same descriptors.  n_doubles from sx_state_size; sx_set_state must be followed by sx_advance(h, t + 1) with the
 * t of the step the state was taken after (t >= 2; earlier steps have no full history and restart from the values). */
int sx_state_size(const sx_handle *h, int64_t *n_doubles);
int sx_get_state(sx_handle *h, double *out);
int sx_set_state(sx_handle *h, const double *in);

/* --- the hot path -------------------------------------------------------------------------------------------------- */
/* spectralTransform!(tile) on var_np1 (calcTendency, src/semiimplicit.jl:728-735) -> tile B coefficients */
int sx_spectral_transform(sx_handle *h);
/* splineTransform!(patchSplines, patchSpectral, gp, sharedSpectral, tile) (src/semiimplicit.jl:237, 285) */
int sx_spline_transform(sx_handle *h);
/* tileTransform!(patchSplines, patchSpectral, gp, tile, splineBuffer) (src/semiimplicit.jl:241, 305) */
int sx_tile_transform(sx_handle *h);
/* advanceTimestep up to and including calcTendency (src/semiimplicit.jl:305-317):
 * tileTransform! -> equation set -> explicit_timestep [-> semiimplicit_adjustment] -> spectralTransform!
 * Only the derivative planes the equation set reads are produced, and on uniform rings part of them never leaves the
 * node-space form: after sx_advance the contents of `physical` are unspecified - call sx_tile_transform (the output path,
 * src/semiimplicit.jl:289-290) before sx_get_physical. */
int sx_advance(sx_handle *h, int32_t t);
/* model_loop's body for a ONE-tile patch (src/semiimplicit.jl:268-285 with a single worker): advanceTimestep followed by
 * splineTransform!, i.e. sx_advance(h, t) + sx_spline_transform(h) in one call.  With SX_GRAPH=1 in the environment at sx_create the
 * step's kernel launches are captured into a hipGraph - once per rotation of the tendency-history buffers, from the third step
 * this handle executes (Adams-Bashforth-3: the launch arguments then repeat with period 3) - and replayed: ONE graph launch per
 * step instead of 5-9 kernel launches, for grids whose step is shorter than the host takes to enqueue it (R, RZ and small RL
 * grids).  Bit-identical to the plain launches; timers on, or a failed capture, fall back to them. */
int sx_step(sx_handle *h, int32_t t);
/* physical_model only (src/semiimplicit.jl:357-363) on the current tile.physical */
int sx_physics(sx_handle *h, int32_t t);
/* checkCFL (src/semiimplicit.jl:737-751): flag = 1 if any NaN in the model state (scans var_np1, which every
 * sx_advance / sx_set_physical_values leaves complete; a NaN in physical[:, v, 1] always reaches it) */
int sx_check_nan(sx_handle *h, int32_t *flag);
/* on-device diagnostic: out[n_vars] = max |var_np1[:, v]| after the last step; with the grid spacing
 * the caller forms the advective CFL number without pulling a field to the host.  A NaN anywhere in a variable makes its
 * maximum NaN (like maximum(abs, x) in Julia).  No allocation per call: the scratch lives with the handle. */
int sx_max_abs(sx_handle *h, double *out);

/* --- tile <-> patch exchange on the device (src/semiimplicit.jl:320-329, 272-285) ---------------------------------- */
/* The tile's B coefficients live in a [tile_b_rDim][n_cols] row-major device array (row = radial node).
 * Rows [0, tile_num_cells) are owned (patchIndexMap), rows [tile_num_cells, +3) are the halo sent to the next
 * tile (haloSendIndexMap); the last tile owns all its rows. */
int sx_tile_b_device(sx_handle *h, void **dev_ptr, int64_t *n_rows, int64_t *n_cols);
/* Make sx_spectral_transform / sx_advance write the tile's B rows at dev_ptr (e.g. inside an all-gather buffer). */
int sx_bind_tile_b(sx_handle *h, void *dev_ptr);
/* sharedSpectral[haloReceiveIndexMap] .+= haloReceiveBuffer (src/semiimplicit.jl:329): B rows 0..2 += recv[3][n_cols] */
int sx_halo_add(sx_handle *h, const void *dev_recv);
/* Source of the patch-level B for sx_spline_transform: row m of the patch is at dev_base + row_offset[m] doubles.
 * dev_base = NULL restores the internal buffer (filled by sx_set_patch_spectral_b or, for a one-tile patch,
 * by sx_spectral_transform). */
int sx_bind_patch_b(sx_handle *h, const void *dev_base, const int64_t *row_offset /* [b_rDim] host */);
/* device pointer of the patch A array [b_rDim][n_cols] */
int sx_patch_a_device(sx_handle *h, void **dev_ptr, int64_t *n_rows, int64_t *n_cols);

/* --- transposed (all-to-all) patch solve -----------------------------------------------------------------------------
 * Scalable alternative to halo + all-gather + redundant solve (src/semiimplicit.jl:320-329, 285) for n tiles = n GPUs.
 * The columns of the [node][col] arrays are split into n contiguous ranges of whole (variable, z-mode) groups
 * (sx_a2a_col_starts). Per step, after sx_advance:
 *   1. sx_a2a_pack_b      tile B rows -> send buffer [dest d][row j < tile_b_rDim][cols of d]
 *   2. all-to-all         every tile's rows for my columns arrive as [tile t][row j][my cols]
 *   3. sx_a2a_solve       sums the rows two tiles share (the reference's halo add), solves my columns for the whole
 *                         patch, writes the solution rows back in the same [tile t][row j][my cols] layout
 *   4. all-to-all         reverse direction
 *   5. sx_a2a_unpack_a    [owner d][row j][cols of d] -> the patch A rows this tile evaluates
 * tile_cell0 / tile_num_cells describe all n tiles (calcTileSizes rows 4 and 3, 0-based cell0). */
int sx_a2a_configure(sx_handle *h, int32_t n_tiles, int32_t my_tile, const int32_t *tile_cell0, const int32_t *tile_num_cells);
int sx_a2a_col_starts(sx_handle *h, int64_t *out /* [n_tiles + 1] */);
int sx_a2a_pack_b(sx_handle *h, void *dev_send);
int sx_a2a_solve(sx_handle *h, const void *dev_recv, void *dev_send);
int sx_a2a_unpack_a(sx_handle *h, const void *dev_recv);

/* --- interface-only (partitioned) patch solve ---------------------------------------------------------------------------
 * instead of the redundant whole-patch solve of src/semiimplicit.jl:285 every tile solves its OWN rows
 * (a chain of n / N rows) and only what couples tiles travels: per tile and column its 6 edge values and the <= 4 rows whose
 * coefficient another tile owns (the 3 halo rows of src/semiimplicit.jl:320-329, PERIODIC wrap rows) go to the owner of
 * the column's reduced system, 6 right-hand-side corrections and the <= 4 foreign coefficients come back.  Per step, after
 * sx_advance:
 *   1. sx_iface_local     tile-local banded solve of the tile's B rows; its 10 rows -> send buffer [dest d][10][cols of d]
 *   2. all-to-all         arrive as [tile t][10][my cols]
 *   3. sx_iface_reduce    one dense [10 N x 10 N] operator per boundary-condition class (built at configure) per column
 *   4. all-to-all         reverse direction
 *   5. sx_iface_apply     a = y' + Z c  ->  the patch A rows this tile evaluates
 * Column split and table arguments as for sx_a2a_*; needs >= 2 tiles with >= 6 free coefficients each. */
int sx_iface_configure(sx_handle *h, int32_t n_tiles, int32_t my_tile, const int32_t *tile_cell0, const int32_t *tile_num_cells);
int sx_iface_col_starts(sx_handle *h, int64_t *out /* [n_tiles + 1] */);
int sx_iface_local(sx_handle *h, void *dev_send);
int sx_iface_reduce(sx_handle *h, const void *dev_recv, void *dev_send);
int sx_iface_apply(sx_handle *h, const void *dev_recv);

/* --- exchange over RCCL, inside the library --------------------------------------------------------------------------
 * One process per GPU, one tile per process (src/semiimplicit.jl:179-184).  The reference's per-step exchange - the halo
 * chain tile -> tile + 1 (src/semiimplicit.jl:203-219, 320-329), the shared sum on the master (:272-282) and the patch solve
 * on every worker (:285) - runs here as ncclSend / ncclRecv / ncclAllGather on the handle's stream, ordered with the
 * kernels by that stream alone.  librccl is bound with dlopen on first use (SX_RCCL_LIB overrides the search; a copy the
 * process has already mapped is reused).  A Julia host needs only ccall:
 *   rank 0: sx_comm_unique_id(id) -> hand the 128 bytes to every worker (the master's RemoteChannels will do)
 *   all   : sx_comm_init(h, n_tiles, my_tile, cell0, ncells, mode, id)      (collective; after hipSetDevice / sx_create)
 *   step  : sx_advance(h, t); sx_exchange(h);                               (replaces :320-329, :272-285)
 * mode 0 = transposed solve (two all-to-alls of B / A rows, each rank solves its share of the columns for the whole
 * patch: scales), mode 1 = the reference's protocol (halo rows to the next tile, all-gather of the owned rows, redundant
 * patch solve), mode 2 = interface-only solve (tile-local solves, two all-to-alls of 10 rows per tile: least traffic and the
 * shortest recurrence; with one tile it is the plain solve).  tile_cell0 / tile_num_cells describe all n tiles (calcTileSizes rows 4 and 3, 0-based cell0).
 * sx_comm_attach does the same with a communicator the host already owns (ncclComm_t, e.g. from NCCL.jl); it is not
 * destroyed with the handle.  After sx_exchange the patch A coefficients this tile evaluates are in place for the next
 * sx_advance / sx_tile_transform.
 * sx_comm_prepare is the part of sx_comm_init / sx_comm_attach that can fail on one rank alone (binding librccl, checking
 * the tile table, allocating the exchange buffers) and is NOT collective: a host that wants to fall back cleanly calls it on
 * every rank, agrees on the outcome through its own channel, and enters the collective sx_comm_init on all ranks or on none
 * (sx_comm_init prepares by itself when this was not called).  A failed set-up leaves the handle without exchange state. */
int sx_comm_unique_id(char *out128);
int sx_comm_prepare(sx_handle *h, int32_t n_tiles, int32_t my_tile, const int32_t *tile_cell0, const int32_t *tile_num_cells,
                    int32_t mode);
int sx_comm_init(sx_handle *h, int32_t n_tiles, int32_t my_tile, const int32_t *tile_cell0, const int32_t *tile_num_cells,
                 int32_t mode, const char *id128);
int sx_comm_attach(sx_handle *h, int32_t n_tiles, int32_t my_tile, const int32_t *tile_cell0, const int32_t *tile_num_cells,
                   int32_t mode, void *nccl_comm);
int sx_exchange(sx_handle *h);
/* The same exchange with all n tiles in ONE process on one GPU (handles hs[0..n-1] = tiles 0..n-1): identical buffer geometry
 * and offset tables, every send / receive pair replaced by a device-to-device copy.  For single-GPU multi-tile runs and for
 * testing the exchange without RCCL (which refuses two ranks on one device). */
int sx_comm_init_local(sx_handle **hs, int32_t n_tiles, const int32_t *tile_cell0, const int32_t *tile_num_cells, int32_t mode);
int sx_exchange_local(sx_handle **hs, int32_t n_tiles);

/* --- measurement ---------------------------------------------------------------------------------------------------- */
/* hipEvent timers around every kernel on the handle's stream (off by default). */
int sx_enable_timers(sx_handle *h, int32_t on);
int sx_reset_timers(sx_handle *h);
/* restrict the event pairs to the kernel with this timer name (NULL = every kernel): two event records per launch cost
 * ~4 us each on the stream, so a timed region that only needs its dominant kernel's duration should not pay for all */
int sx_timer_only(sx_handle *h, const char *name);
/* names[i] borrowed static strings; ms[i] accumulated milliseconds; calls[i] launches. Returns count via n. */
int sx_get_timers(sx_handle *h, int32_t max, const char **names, double *ms, int64_t *calls, int32_t *n);
/* algorithmic bytes of one launch of the named kernel (SURVEY.md 8(d) accounting), 0 if unknown */
int sx_kernel_bytes(sx_handle *h, const char *name, double *bytes);

#ifdef __cplusplus
}
#endif
#endif /* SCYTHE_HIP_H */
