/* A torch-free host on the C ABI of libscythe_hip.so: the closest stand-in this image has for the Julia `ccall` host of
 * INTEGRATION.md (there is no julia here).  Plain C, compiled by tests/test_gpu_abi_host.py with gcc against
 * include/scythe_hip.h; the library is dlopen'ed, so the only HIP runtime / RCCL in the process are the ones the library
 * itself resolves from /opt/rocm (no Python, no torch, no bundled libamdhip64).
 *
 *   abi_host <libscythe_hip.so> <steps> <tiles> <out.bin>
 *
 * Runs the reference's LinearAdvection1D model (models/LinearAdvection1D.jl:1-21; the notebook's known-answer case,
 * notebooks/LinearAdvection_example.ipynb cells 2-6: R grid, 100 cells on [-50, 50], periodic, u0 = exp(-(x/20)^2),
 * ts = 0.05) for <steps> steps through exactly the calls the Julia glue makes -
 *   sx_create -> sx_get_gridpoints -> sx_set_physical_values -> sx_spectral_transform -> [exchange] -> sx_spline_transform
 *   -> per step: sx_advance(h, t), [exchange], sx_spline_transform -> sx_tile_transform -> sx_get_physical
 * - and writes x[300], u[300] (physical[:, 1, 1]) as raw doubles for the test to compare with the oracle.
 *   tiles = 1: one handle, and additionally the in-library RCCL exchange with a ONE-rank communicator (sx_comm_unique_id /
 *              sx_comm_init / sx_exchange): librccl is then bound by the library from /opt/rocm, without torch in the process.
 *   tiles = 2: two handles in this process, exchange through the library's loopback transport (sx_comm_init_local /
 *              sx_exchange_local), i.e. the reference's 2-worker run (src/semiimplicit.jl:203-219, 320-329).
 * Exit status 0 on success; any ABI error prints sx_last_error() and exits 1. */
#include <dlfcn.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "scythe_hip.h"

#define BIND(name) \
    __typeof__(&name) p_##name = (__typeof__(&name))dlsym(lib, #name); \
    if (!p_##name) { fprintf(stderr, "missing symbol %s\n", #name); return 1; }
#define OK(call) \
    do { if ((call) != 0) { fprintf(stderr, "%s failed: %s\n", #call, p_sx_last_error()); return 1; } } while (0)

/* abi_host case <libscythe_hip.so> <case.bin> <out.bin>: ANY grid / equation set, described by a small binary file the test
 * writes (what the Julia glue's createHipModelTile fills in from GridParameters / ModelParameters):
 *   int32  geometry, num_cells, nvars, zDim, ring_uniform_L, equation_set, semiimplicit, steps, w_index, xi_index, col_var
 *   double xmin, xmax, zmin, zmax, ts, params[SX_NPARAMS]
 *   int32  bcl[nvars], bcr[nvars], bcb[nvars], bct[nvars]
 *   int64  n_points;  double values[n_points * nvars]      (column-major [point, var]: the initial condition)
 * One tile; writes physical[n_points, nvars, n_derivs] after <steps> steps and sx_max_abs's nvars values behind it. */
static int run_case(int argc, char **argv) {
    if (argc != 5) { fprintf(stderr, "usage: abi_host case lib case.bin out.bin\n"); return 2; }
    void *lib = dlopen(argv[2], RTLD_NOW | RTLD_GLOBAL);
    if (!lib) { fprintf(stderr, "dlopen: %s\n", dlerror()); return 1; }
    BIND(sx_last_error) BIND(sx_create) BIND(sx_destroy) BIND(sx_get_dims) BIND(sx_set_physical_values) BIND(sx_spectral_transform)
    BIND(sx_spline_transform) BIND(sx_advance) BIND(sx_tile_transform) BIND(sx_get_physical) BIND(sx_check_nan) BIND(sx_max_abs)
    FILE *f = fopen(argv[3], "rb");
    if (!f) { fprintf(stderr, "cannot read %s\n", argv[3]); return 1; }
    int32_t hd[11];
    double dd[5 + SX_NPARAMS];
    if (fread(hd, sizeof(int32_t), 11, f) != 11 || fread(dd, sizeof(double), 5 + SX_NPARAMS, f) != 5 + SX_NPARAMS) return 1;
    const int nv = hd[2];
    int32_t *bc = malloc(sizeof(int32_t) * 4 * nv);
    int64_t npts = 0;
    if (fread(bc, sizeof(int32_t), 4 * nv, f) != (size_t)(4 * nv) || fread(&npts, sizeof(int64_t), 1, f) != 1) return 1;
    double *vals = malloc(sizeof(double) * npts * nv);
    if (fread(vals, sizeof(double), npts * nv, f) != (size_t)(npts * nv)) return 1;
    fclose(f);
    sx_grid_desc gd;
    memset(&gd, 0, sizeof gd);
    gd.abi_version = SX_ABI_VERSION; gd.geometry = hd[0]; gd.num_cells = hd[1]; gd.nvars = nv; gd.zDim = hd[3]; gd.ring_uniform_L = hd[4];
    gd.xmin = dd[0]; gd.xmax = dd[1]; gd.zmin = dd[2]; gd.zmax = dd[3]; gd.l_q = 2.0;
    gd.bcl = bc; gd.bcr = bc + nv; gd.bcb = bc + 2 * nv; gd.bct = bc + 3 * nv;
    gd.tile_cell0 = 0; gd.tile_num_cells = hd[1]; gd.tile_num = 2;
    sx_model_desc md;
    memset(&md, 0, sizeof md);
    md.ts = dd[4]; md.equation_set = hd[5]; md.semiimplicit = hd[6]; md.params = dd + 5;
    md.w_index = hd[8]; md.xi_index = hd[9]; md.col_var = hd[10];
    sx_handle *h = 0;
    OK(p_sx_create(&gd, &md, &h));
    sx_dims d;
    OK(p_sx_get_dims(h, &d));
    if (d.n_points != npts || d.n_vars != nv) { fprintf(stderr, "case file does not match the grid: %ld points\n", (long)d.n_points); return 1; }
    OK(p_sx_set_physical_values(h, vals));
    OK(p_sx_spectral_transform(h));
    OK(p_sx_spline_transform(h));
    for (int s = 1; s <= hd[7]; s++) { OK(p_sx_advance(h, s)); OK(p_sx_spline_transform(h)); }
    int32_t flag = 0;
    OK(p_sx_check_nan(h, &flag));
    if (flag) { fprintf(stderr, "NaN in the model state\n"); return 1; }
    double *mx = malloc(sizeof(double) * nv);
    OK(p_sx_max_abs(h, mx));
    OK(p_sx_tile_transform(h));
    const size_t np = (size_t)npts * nv * d.n_derivs;
    double *phys = malloc(sizeof(double) * np);
    OK(p_sx_get_physical(h, phys));
    f = fopen(argv[4], "wb");
    if (!f || fwrite(phys, sizeof(double), np, f) != np || fwrite(mx, sizeof(double), nv, f) != (size_t)nv) { fprintf(stderr, "cannot write\n"); return 1; }
    fclose(f);
    OK(p_sx_destroy(h));
    printf("abi_host case ok: %d steps, %ld points x %d vars x %d slots\n", hd[7], (long)npts, nv, d.n_derivs);
    return 0;
}

int main(int argc, char **argv) {
    if (argc >= 2 && strcmp(argv[1], "case") == 0) return run_case(argc, argv);
    if (argc != 5) { fprintf(stderr, "usage: abi_host lib steps tiles out.bin | abi_host case lib case.bin out.bin\n"); return 2; }
    const int steps = atoi(argv[2]), ntiles = atoi(argv[3]);
    void *lib = dlopen(argv[1], RTLD_NOW | RTLD_GLOBAL);
    if (!lib) { fprintf(stderr, "dlopen: %s\n", dlerror()); return 1; }
    BIND(sx_last_error) BIND(sx_abi_version) BIND(sx_create) BIND(sx_destroy) BIND(sx_get_dims) BIND(sx_get_gridpoints)
    BIND(sx_set_physical_values) BIND(sx_spectral_transform) BIND(sx_spline_transform) BIND(sx_advance) BIND(sx_tile_transform)
    BIND(sx_get_physical) BIND(sx_equation_set_id) BIND(sx_check_nan) BIND(sx_comm_unique_id) BIND(sx_comm_init) BIND(sx_exchange)
    BIND(sx_comm_init_local) BIND(sx_exchange_local) BIND(sx_synchronize)
    if (p_sx_abi_version() != SX_ABI_VERSION) { fprintf(stderr, "ABI version mismatch\n"); return 1; }
    if (ntiles < 1 || ntiles > 4) { fprintf(stderr, "tiles must be 1..4\n"); return 2; }

    const int32_t bc[1] = {SX_BC_PERIODIC};
    double par[SX_NPARAMS];
    memset(par, 0, sizeof par);
    par[SX_P_C0] = 1.0;
    par[SX_P_K] = 0.0;
    sx_model_desc md;
    memset(&md, 0, sizeof md);
    md.ts = 0.05;
    md.equation_set = p_sx_equation_set_id("LinearAdvection1D");
    md.params = par;
    if (md.equation_set != SX_EQ_LINEAR_ADVECTION_1D) { fprintf(stderr, "equation set lookup failed\n"); return 1; }

    const int nc = 100;
    sx_handle *h[4] = {0};
    int32_t cell0[4], ncells[4];
    int64_t npts[4], first[4], total = 0;
    for (int t = 0; t < ntiles; t++) {                       /* calcTileSizes for an R grid: cells split evenly */
        cell0[t] = (int)((long)nc * t / ntiles);
        ncells[t] = (int)((long)nc * (t + 1) / ntiles) - cell0[t];
        sx_grid_desc gd;
        memset(&gd, 0, sizeof gd);
        gd.abi_version = SX_ABI_VERSION;
        gd.geometry = SX_GEOM_R;
        gd.xmin = -50.0; gd.xmax = 50.0; gd.num_cells = nc; gd.l_q = 2.0; gd.nvars = 1;
        gd.bcl = bc; gd.bcr = bc;
        gd.tile_cell0 = cell0[t]; gd.tile_num_cells = ncells[t]; gd.tile_num = t + 2;
        OK(p_sx_create(&gd, &md, &h[t]));
        sx_dims d;
        OK(p_sx_get_dims(h[t], &d));
        npts[t] = d.n_points;
        first[t] = total;
        total += d.n_points;
        if (d.n_vars != 1 || d.n_derivs != 3) { fprintf(stderr, "unexpected dims\n"); return 1; }
    }
    if (total != 300) { fprintf(stderr, "expected 300 gridpoints, got %ld\n", (long)total); return 1; }
    double *x = malloc(sizeof(double) * total), *u = malloc(sizeof(double) * total), *phys = malloc(sizeof(double) * total * 3);

    int use_rccl = 0;
    if (ntiles == 1) {
        /* the in-library exchange with a one-rank communicator: dlopen of librccl from /opt/rocm, ncclCommInitRank, grouped
         * send / recv to self around the patch solve - what a one-worker Julia run would execute */
        char id[128];
        OK(p_sx_comm_unique_id(id));
        OK(p_sx_comm_init(h[0], 1, 0, cell0, ncells, 0, id));
        use_rccl = 1;
    } else {
        OK(p_sx_comm_init_local(h, ntiles, cell0, ncells, 0));
    }
#define EXCHANGE_AND_SOLVE()                                             \
    do {                                                                 \
        if (use_rccl) OK(p_sx_exchange(h[0]));                           \
        else OK(p_sx_exchange_local(h, ntiles));                         \
    } while (0)

    for (int t = 0; t < ntiles; t++) {                      /* initialize_model (src/semiimplicit.jl:126-136) */
        OK(p_sx_get_gridpoints(h[t], x + first[t]));
        for (int64_t i = 0; i < npts[t]; i++) u[first[t] + i] = exp(-(x[first[t] + i] / 20.0) * (x[first[t] + i] / 20.0));
        OK(p_sx_set_physical_values(h[t], u + first[t]));
        OK(p_sx_spectral_transform(h[t]));
    }
    EXCHANGE_AND_SOLVE();
    for (int s = 1; s <= steps; s++) {                      /* model_loop (src/semiimplicit.jl:268-297) */
        for (int t = 0; t < ntiles; t++) OK(p_sx_advance(h[t], s));
        EXCHANGE_AND_SOLVE();
    }
    for (int t = 0; t < ntiles; t++) {
        int32_t flag = 0;
        OK(p_sx_check_nan(h[t], &flag));
        if (flag) { fprintf(stderr, "NaN in the model state\n"); return 1; }
        OK(p_sx_tile_transform(h[t]));
        OK(p_sx_get_physical(h[t], phys));                  /* [n_points, 1, 3] column-major: the first n_points are u */
        memcpy(u + first[t], phys, sizeof(double) * npts[t]);
    }
    FILE *f = fopen(argv[4], "wb");
    if (!f || fwrite(x, sizeof(double), total, f) != (size_t)total || fwrite(u, sizeof(double), total, f) != (size_t)total) {
        fprintf(stderr, "cannot write %s\n", argv[4]);
        return 1;
    }
    fclose(f);
    for (int t = 0; t < ntiles; t++) OK(p_sx_destroy(h[t]));
    printf("abi_host ok: %d steps, %d tile(s), exchange %s\n", steps, ntiles, use_rccl ? "rccl(1 rank)" : "loopback");
    return 0;
}
