"""-m gpu: BASELINE.json's parity-test configurations at their full size, plus size-independent properties of the
RLZ 513 x 256 x 64 bench workload and one direct comparison with the C oracle on that grid."""
import os

import numpy as np
import pytest

from tests import cases

pytestmark = pytest.mark.gpu
TOL = 1e-10        # the model fields (derivative slot 1 = values) and the spectral state (A coefficients)


_per_slot, _report, _check = cases.per_slot_errors, cases.report_slots, cases.check_full


def test_config2_rl_slab_literal_100_cells():
    case = cases.config2_literal()
    hip = cases.HipModel(case)
    orc = cases.OracleModel(case)
    alt = cases.OracleModel(case, numpy_twin=True)
    assert hip.run.tiles[0].N == 181800
    for _ in range(3):
        hip.step()
        orc.step()
        alt.step()
    _check(hip, orc, [0, 1, 2, 30, 149, 150, 151, 297, 298, 299], "config 2 (RL 100 cells, native rings, kmax 300), 3 steps",
           orc_alt=alt)


def test_config2_two_tiles_balance_gridpoints():
    case = cases.config2_literal(twoway=True)
    one = cases.HipModel(case)
    two = cases.HipModel(case, num_tiles=2)
    n = [g.N for g in two.run.tiles]
    assert sum(n) == 181800 and abs(n[0] - n[1]) / 181800 < 0.02       # calcTileSizes balances points, not cells
    lay = two.run.layout
    o1 = cases.OracleModel(case)
    o2 = cases.OracleModel(case, tiles=list(zip(lay.cell0, lay.ncells)))
    for _ in range(3):
        for m in (one, two, o1, o2):
            m.step()
    a, b = two.physical(), one.physical()
    assert cases.rel_err(a[:, :, :1], b[:, :, :1]) < TOL
    # two tilings of the HIP path differ by no more than the same two tilings of the fp64 oracle do (x 10)
    d, floor = _per_slot(a, b), _per_slot(o2.physical(), o1.physical())
    _report("config 2, two tiles vs one tile", o1.g, [("HIP", d), ("fp64 oracle", floor)])
    assert (d <= 10.0 * floor + 1e-12).all(), (d, floor)


def test_config3_rz_513x128_semiimplicit_three_way():
    """Config 3 at full size, the semi-implicit column solve three ways:
      HIP        - the product (Helmholtz operator inverted once in extended precision at sx_create, csrc/sx_setup.cpp)
      LU oracle  - the reference's own arithmetic: Float64 matrix, LAPACK getrf / getrs per column (src/semiimplicit.jl:768-781, 589)
      extended   - the oracle's extended-precision inverse: what the column solve returns in exact arithmetic
    Reported: HIP vs LU, HIP vs extended, LU vs extended, for the fields and for every derivative slot."""
    case = cases.config3_rz()
    hip = cases.HipModel(case)
    ext = cases.OracleModel(case)
    lu = cases.OracleModel(case, helmholtz="lu")
    assert hip.run.tiles[0].N == 513 * 128
    for _ in range(6):
        hip.step()
        ext.step()
        lu.step()
    rings = list(range(0, 513, 19))
    alt = cases.OracleModel(case, numpy_twin=True)
    for _ in range(6):
        alt.step()
    _check(hip, ext, rings, "config 3 (RZ 513 x 128, semi-implicit), 6 steps, HIP vs extended-precision Helmholtz oracle", orc_alt=alt)
    ph, pe, pl = hip.physical(), ext.physical(), lu.physical()
    val = lambda x, y: max(np.abs(x[:, v, 0] - y[:, v, 0]).max() / np.abs(y[:, v, 0]).max() for v in range(y.shape[1]))
    three = {"HIP vs LU oracle": (val(ph, pl), cases.rel_err_per_var(ph, pl)),
             "HIP vs extended": (val(ph, pe), cases.rel_err_per_var(ph, pe)),
             "LU oracle vs extended": (val(pl, pe), cases.rel_err_per_var(pl, pe))}
    print("\nconfig 3 three-way (fields, all derivative slots):")
    for k, (a, b) in three.items():
        print("  %-24s %.2e  %.2e" % (k, a, b))
    # the model fields agree to 1e-10 in all three pairs: the 1e-10 claim holds against the reference's LU arithmetic too
    for k, (a, _) in three.items():
        assert a < TOL, (k, a)
    # derivative slots: the HIP path is as close to the reference's arithmetic as the exact solve is (N^4-amplified rounding)
    assert three["HIP vs LU oracle"][1] <= 2.0 * three["LU oracle vs extended"][1] + 1e-12


def _bench_model(num_tiles, exchange="a2a", split="reference", workload="rlz_513x256x64", storage="f64", impl="torch"):
    import bench
    import scythe_jl_amd as S
    kw, L = bench.grid_kwargs(workload)
    gp = S.GridParameters(ring_uniform_L=L, storage=storage, **kw)
    mp = S.ModelParameters(ts=bench.TS_OF.get(workload, bench.TS), equation_set="Oneway_ShallowWater_HeightResolvedBL", grid_params=gp,
                           physical_params=dict(bench.PAR))
    run = S.ModelRun(mp, num_tiles=num_tiles, device="cuda", exchange=exchange, split=split, impl=impl)
    run.set_initial_conditions([bench.initial_condition(S.getGridpoints(g)) for g in run.tiles])
    return run


def test_config4_full_size_tiling_invariance():
    """RLZ 513 x 256 x 64, 6 variables: 3 even radial tiles and 4 cost-balanced ones (transposed solve, Python-side stand-in
    for the exchange), and 8 cost-balanced tiles with the exchange done INSIDE the library (the RCCL path's buffers and
    offset tables through the loopback transport, all three protocols incl. the interface-only solve) reproduce the one-tile run."""
    fields = []
    for nt, split, impl, exch in ((1, "reference", "torch", "a2a"), (3, "reference", "torch", "a2a"), (4, "cost", "torch", "a2a"),
                                  (8, "cost", "lib", "a2a"), (8, "cost", "lib", "gather"), (8, "cost", "lib", "iface"),
                                  (4, "reference", "torch", "iface")):
        run = _bench_model(nt, split=split, impl=impl, exchange=exch)
        for _ in range(3):
            run.step()
        vals = []
        for g in run.tiles:
            g.tileTransform_()
            assert not g.check_nan()
            vals.append(g.physical[:, :, 0])
        fields.append(np.concatenate(vals, axis=0))
        run.close()
    a = fields[0]
    for b in fields[1:]:
        for v in range(a.shape[1]):
            sc = np.abs(a[:, v]).max()
            assert np.abs(a[:, v] - b[:, v]).max() <= 1e-11 * max(sc, 1e-300)


def test_config4_full_size_two_steps_against_the_c_oracle():
    """The bench workload's grid at its full size - RLZ 513 x 256 x 64 on the uniform 256-point ring table, 6 variables,
    7 derivative slots, Oneway_ShallowWater_HeightResolvedBL - stepped twice by the HIP path and by the C oracle (OpenMP;
    about a second per step on the box's host cores): spectral state and model fields to 1e-10, and every derivative slot
    of the HIP run no further from the extended-precision evaluation of its own coefficients than the oracle's (sampled
    rings: innermost, the last truncated ring, the first full-spectrum ring, outermost)."""
    case = cases.rlz_hrbl(num_cells=171, zDim=64, ring_L=256)
    case["ts"] = 0.2                     # the bench's step: the explicit set is diffusion-limited at r = 198 m
    hip = cases.HipModel(case)
    assert hip.run.tiles[0].N == 513 * 256 * 64
    orc = cases.OracleModel(case)
    for _ in range(2):
        hip.step()
        orc.step()
    # (whole-grid derivative slots are REPORTED here, bounded only through the sampled rings against extended precision: the
    #  fp64 noise floor of check_full needs a second, independently written fp64 evaluation, and the numpy oracle does not step
    #  8.4 M points in test time; a two-tile split of the C oracle only re-orders the halo sum and differs from the one-tile
    #  run by 5e-14 (values) .. 1.8e-10 (d2/dlambda2), 5-18 x less than two different algorithms do - measured, round 3)
    _check(hip, orc, [0, 1, 125, 126, 127, 300, 512], "config 4 shape (RLZ 513 x 256 x 64, uniform rings), 2 steps")


def test_config4_native_equivalent_full_size_against_the_c_oracle():
    """The native-equivalent shape of config 4 (bench.py's `native_equivalent`): 85 cells = 255 ragged rings of
    4 + 4 ri points keeping ri wavenumbers, 64 levels, 8.4 M points - the matrix-core DFT kernels at the lengths they are
    timed on (8 .. 1024 points, kmax up to 255) against the C oracle, 2 steps."""
    case = cases.rlz_hrbl(num_cells=85, zDim=64)
    case["ts"] = 0.2
    hip = cases.HipModel(case)
    assert hip.run.tiles[0].N == 8421120                 # 131,580 horizontal points (rings of 8 .. 1024) x 64 levels
    orc = cases.OracleModel(case)
    for _ in range(2):
        hip.step()
        orc.step()
    _check(hip, orc, [0, 1, 127, 128, 253, 254], "config 4, native-equivalent shape (255 ragged rings x 64 levels, kmax 255), 2 steps")


def _bench_case(num_cells=None, native=False):
    """bench.py's own workload (grid, boundary conditions, parameters, time step, initial condition) as a test case."""
    import bench
    kw, L = bench.grid_kwargs("rlz_513x256x64")
    if num_cells:
        kw["num_cells"] = num_cells
    if not native:
        kw["ring_L"] = L
    return dict(name="bench", grid=kw, eq="Oneway_ShallowWater_HeightResolvedBL", ts=bench.TS, par=dict(bench.PAR), ic=bench.initial_condition)


def _long_run_against_the_c_oracle(case, nsteps, rings, title, npoints):
    """HIP path and C oracle side by side for `nsteps` steps: the spectral state (A coefficients) is compared after EVERY step - the
    measured growth of the difference per step is printed - and state + value slot are held to 1e-10 at the end; the derivative
    slots by check_full's extended-precision criterion on the sampled rings."""
    hip = cases.HipModel(case)
    assert hip.run.tiles[0].N == npoints
    orc = cases.OracleModel(case)
    hist = []
    for t in range(1, nsteps + 1):
        hip.step()
        orc.step()
        hist.append(cases.rel_err(hip.A, orc.A))
        assert np.isfinite(hist[-1]) and hist[-1] < TOL, (t, hist)
    growth = (hist[-1] / max(hist[0], 1e-300)) ** (1.0 / max(nsteps - 1, 1))
    print("\n%s: max relative difference of the A coefficients after step t" % title)
    print("  " + "  ".join("t=%d %.1e" % (t + 1, e) for t, e in enumerate(hist) if t in (0, 1, 4, 9, 14, 19, 24, nsteps - 1)))
    print("  geometric growth per step over %d steps: %.3f" % (nsteps, growth))
    _check(hip, orc, rings, "%s, %d steps" % (title, nsteps))
    return hist


def test_config4_full_size_25_steps_against_the_c_oracle():
    """Parity at the bench's own length: bench.py times steps 6-25 (driver: --warmup 5 --steps 20) of exactly this configuration -
    RLZ 513 x 256 x 64, uniform ring table, bench.py's parameters, boundary conditions, time step and initial condition.  25 steps of
    the HIP path against 25 steps of the C oracle (about a second per step on the box's host cores): A coefficients after every
    step and the value slot of every variable at the end within 1e-10 (north_star: "fields within 1e-10 rel-err of CPU reference")."""
    _long_run_against_the_c_oracle(_bench_case(), 25, [0, 1, 125, 126, 127, 300, 512],
                                   "bench workload (RLZ 513 x 256 x 64, uniform rings) vs the C oracle", 513 * 256 * 64)


def test_config4_native_equivalent_10_steps_against_the_c_oracle():
    """The same over 10 steps on the native-equivalent shape (bench.py's `native_equivalent`: 85 cells = 255 ragged rings x 64 levels,
    bench.py's parameters and initial condition): the matrix-core DFT kernels over the length of a short timing run."""
    _long_run_against_the_c_oracle(_bench_case(num_cells=85, native=True), 10, [0, 1, 127, 128, 253, 254],
                                   "native-equivalent shape (255 ragged rings x 64 levels) vs the C oracle", 8421120)


def test_native_rings_of_a_171_cell_patch_beyond_kmax_319():
    """The "512-ring" problem on Springsteel's NATIVE layout: 171 cells = 513 ragged rings of 8 .. 2,052 points keeping up to
    512 wavenumbers (SURVEY.md 8(c) "Layouts"; any num_cells is legal, src/semiimplicit.jl:155-169).  Rings with kmax > 319
    take the chunked matrix-core DFT kernels (coefficient sets pass through the LDS in chunks of 256 wavenumbers, the
    forward transform spreads a ring's wavenumber tiles over several workgroups); 8 levels, 2 steps against the C oracle."""
    case = cases.rlz_hrbl(num_cells=171, zDim=8)
    case["ts"] = 0.2
    hip = cases.HipModel(case)
    assert hip.run.tiles[0].N == 529416 * 8                # sum of 4 + 4 ri points over 513 rings, x 8 levels
    orc = cases.OracleModel(case)
    for _ in range(2):
        hip.step()
        orc.step()
    _check(hip, orc, [0, 1, 318, 319, 320, 321, 400, 511, 512], "171-cell native RLZ patch (513 ragged rings, kmax up to 512) x 8 levels, 2 steps")


def test_native_rl_patch_of_171_cells_beyond_kmax_319():
    """The same rings without a vertical dimension (RL slab set: the MFMA columns are (variable, derivative plane) pairs)."""
    case = cases.rl_slab(num_cells=171)
    hip = cases.HipModel(case)
    assert hip.run.tiles[0].N == 529416
    orc = cases.OracleModel(case)
    for _ in range(2):
        hip.step()
        orc.step()
    _check(hip, orc, [0, 1, 319, 320, 321, 512], "171-cell native RL patch (kmax up to 512), 2 steps")


def test_config4_full_size_forward_transform_is_linear():
    import scythe_jl_amd as S
    run = _bench_model(1)
    g = run.tiles[0]
    rng = np.random.default_rng(11)
    x = rng.standard_normal((g.N, g.V))
    y = rng.standard_normal((g.N, g.V))

    def fwd(vals):
        g.set_physical_values(vals)
        g.spectralTransform_()
        return g.spectral.copy()

    bx, by, bxy = fwd(x), fwd(y), fwd(2.0 * x - 3.0 * y)
    assert np.abs(bxy - (2.0 * bx - 3.0 * by)).max() <= 1e-12 * np.abs(bxy).max()
    run.close()


def test_config4_full_size_node_space_path_equals_ring_wise_path(monkeypatch):
    """The node-space ("radial last") inverse + cell-wise equation-set kernel against the ring-wise kernels on the whole
    513 x 256 x 64 grid: same fields after 3 steps to rounding."""
    a = _bench_model(1)
    monkeypatch.setenv("SX_NODE_MODE", "0")
    b = _bench_model(1)
    for _ in range(3):
        a.step()
        b.step()
    fa, fb = a.tiles[0].var_np1, b.tiles[0].var_np1
    for v in range(fa.shape[1]):
        assert np.abs(fa[:, v] - fb[:, v]).max() <= 1e-11 * max(np.abs(fb[:, v]).max(), 1e-300)
    a.close()
    b.close()


@pytest.mark.parametrize("mode", ["1", "2"])
def test_config4_full_size_two_stream_modes_are_bit_identical(monkeypatch, mode):
    """SX_OVERLAP=1 / 2 (the inner-ring chain on a second stream) only reorders kernels that touch disjoint
    points: after 4 steps the fields are bit-identical to the one-stream run."""
    a = _bench_model(1)
    monkeypatch.setenv("SX_OVERLAP", mode)
    b = _bench_model(1)
    for _ in range(4):
        a.step()
        b.step()
    fa, fb = a.tiles[0].var_np1, b.tiles[0].var_np1
    assert not a.tiles[0].check_nan() and not b.tiles[0].check_nan()
    assert np.array_equal(fa, fb)
    a.close()
    b.close()


def test_config4_full_size_azimuthal_derivative_slots_are_the_spectral_derivatives():
    """tileTransform! at full size: on every sampled ring and level, the d/dlambda and d2/dlambda2 slots equal the
    FFT derivatives (numpy, on the host) of the value slot - the three slots come from one spectrum."""
    run = _bench_model(1)
    g = run.tiles[0]
    g.tileTransform_()
    phys = g.physical                                    # [N, V, D] with D = u, r, rr, l, ll, z, zz
    L, nz = 256, 64
    k = np.fft.rfftfreq(L, 1.0 / L)
    for ring in (0, 7, 130, 512):
        for z in (0, 31, 63):
            for v in (0, 2, 4):
                idx = (ring * L + np.arange(L)) * nz + z
                u = phys[idx, v, 0]
                spec = np.fft.rfft(u)
                dl = np.fft.irfft(1j * k * spec, L)
                dll = np.fft.irfft(-(k ** 2) * spec, L)
                scale = max(np.abs(u).max(), 1e-300)
                assert np.abs(phys[idx, v, 3] - dl).max() <= 1e-9 * max(np.abs(dl).max(), scale)
                assert np.abs(phys[idx, v, 4] - dll).max() <= 1e-8 * max(np.abs(dll).max(), scale)
    run.close()


# ----------------------------------------------------------------------------- config 5 at its workload
# RLZ 1023 x 512 x 128 (341 cells), fp32-stored derivative planes: the two-wave 512-point FFT with a full spectrum
# (kmax 255), the 128-level cell-wise and ring-wise equation-set kernels and the 128-level sliding-window inner products at
# full size: the size-independent properties of the config-4 tests, one comparison of the spectral state with the C oracle
# (7 s per step at this size), plus (in test_gpu_parity.py) an oracle comparison at 90 cells x 512-point rings where every
# FFT bin carries signal.
C5 = "rlz_1023x512x128"


def _np1_fields(run):
    out = []
    for g in run.tiles:
        assert not g.check_nan()
        out.append(g.var_np1)
    return np.concatenate(out, axis=0)


def _max_rel(a, b):
    return max(np.abs(a[:, v] - b[:, v]).max() / max(np.abs(b[:, v]).max(), 1e-300) for v in range(a.shape[1]))


def test_config5_full_size_f32_storage_finite_and_close_to_f64_storage():
    """20 steps at full size: no NaN, and the fp32-stored-derivative run stays within the declared fp32-mode tolerance
    (1e-6 of each variable's scale, tests/test_gpu_parity.py) of the all-fp64 run of the same library."""
    a = _bench_model(1, workload=C5, storage="f32")
    assert a.tiles[0].N == 1023 * 512 * 128
    b = _bench_model(1, workload=C5, storage="f64")
    for _ in range(20):
        a.step()
        b.step()
    fa, fb = _np1_fields(a), _np1_fields(b)
    err = _max_rel(fa, fb)
    print("\nconfig 5 full size, 20 steps: fp32-stored derivative planes vs all-fp64, max relative field difference %.2e" % err)
    assert 0.0 < err < 1e-6, err
    a.close()
    b.close()


@pytest.mark.skipif(not os.environ.get("SCYTHE_SLOW_TESTS"), reason="a MEASUREMENT (4 minutes, 45 GB of host arrays), recorded in "
                    "profiles/r03/config5_f32x_full_size_20_steps.txt: fp32 transform intermediates miss the declared bars at 128 levels")
def test_config5_full_size_fp32_spectral_intermediates_20_steps():
    """BASELINE.json configs[4] / SURVEY.md 8(d) item 5 at FULL size over 20 steps: storage "f32x" (fp32-stored derivative planes
    AND fp32-stored transform intermediates - vertically inverted coefficients, ring spectra - with fp64 accumulation)
    against the all-fp64 run of the same library.  Declared: fields within 1e-6 of each variable's scale, every derivative
    slot (tileTransform! after the 20 steps) within 5e-5 of the slot's scale.  It does not hold them (see the 64-level
    case in tests/test_gpu_parity.py for the mechanism); run with SCYTHE_SLOW_TESTS=1 to reproduce the numbers - the assertions
    below are on the RECORDED outcome."""
    a = _bench_model(1, workload=C5, storage="f32x")
    b = _bench_model(1, workload=C5, storage="f64")
    for _ in range(20):
        a.step()
        b.step()
    fa, fb = _np1_fields(a), _np1_fields(b)
    err = _max_rel(fa, fb)
    del fa, fb
    # derivative slots on sampled rings (the whole physical array is 22 GB per run)
    worst = np.zeros(7)
    ga, gb = a.tiles[0], b.tiles[0]
    ga.tileTransform_()
    gb.tileTransform_()
    pa, pb = ga.physical, gb.physical
    g = cases.oracle_grid(cases.config5_case())
    idx = cases.ring_points(g, [0, 1, 253, 254, 600, 1022])
    worst = cases.per_slot_errors(pa[idx], pb[idx])
    print("\nconfig 5 full size, 20 steps, fp32 spectral intermediates vs all-fp64: fields %.2e; slots on sampled rings %s"
          % (err, " ".join("%.1e" % x for x in worst)))
    # the recorded outcome (profiles/r03/config5_f32x_full_size_20_steps.txt): the mode MISSES its declared bars - fields 1.3e-6
    # against 1e-6, d2/dz2 5.9e-4 against 5e-5 - which is why it is not the config-5 default.  The test passes when that
    # measurement reproduces, and fails if the mode ever starts to hold the bars (then the decision has to be revisited).
    assert 1e-6 < err < 5e-6, err
    assert worst[6] > 5e-5 and (worst[:5] < 5e-5).all(), worst
    a.close()
    b.close()


def test_config5_full_size_against_the_sampled_oracle_fixture():
    """Config 5 at FULL size (RLZ 1023 x 512 x 128, 67 M points, all-fp64 storage), 2 steps, against the C oracle's full-size
    run as sampled by tests/golden/make_config5_fixture.py (run once in the build container; committed: tests/golden/
    config5_sampled.npz).  (1) STATE: the A coefficients of 48 (z-mode, wavenumber) columns, all radial nodes, to 1e-10 of
    the variable's largest coefficient; (2) VALUES on sampled points of cells 0 / 84 / 85 / 340 to 1e-10; (3) every
    DERIVATIVE slot on rings 0 / 253 / 254 / 256 / 1022 by check_full's criterion - the HIP path's error against the
    EXTENDED-precision evaluation of its own coefficients is no larger than twice the fp64 oracle's against its own."""
    fx = np.load(os.path.join(os.path.dirname(__file__), "golden", "config5_sampled.npz"))
    case = cases.config5_case()
    assert case["ts"] == float(fx["ts"])
    hip = cases.HipModel(case)
    tile = hip.run.tiles[0]
    assert tile.N == 1023 * 512 * 128
    for _ in range(int(fx["steps"])):
        hip.step()
    assert not tile.check_nan()
    g = cases.oracle_grid(case)
    A = np.asarray(hip.A)
    eA = 0.0
    for i, (zm, blk) in enumerate(fx["cols"]):
        for v in range(g.V):
            got = A[:, v].reshape(g.b_zDim, g.K2, g.b_rDim)[zm, blk]
            eA = max(eA, float(np.abs(got - fx["A_cols"][i, :, v]).max() / fx["A_scale"][v]))
    tile.tileTransform_()
    phys = tile.physical
    got = phys[fx["pt_idx"]]
    e_pts = cases.per_slot_errors(got, fx["pt_val"])       # slot scale floored by the operator gain x the variable's magnitude
    rings = [int(r) for r in fx["rings"]]
    e_hip = cases.slot_errors_vs_extended_rings(g, {r: phys[cases.ring_points(g, [r])] for r in rings}, A, rings)
    e_orc = fx["e_orc"]
    cases.report_slots("config 5 at full size (RLZ 1023 x 512 x 128, fp64), 2 steps (A coefficients %.1e on 48 columns)" % eA, g,
                       [("HIP vs extended precision (own A)", e_hip), ("fp64 oracle vs extended precision (own A; fixture)", e_orc),
                        ("HIP vs fp64 oracle, sampled points", e_pts)])
    assert eA < TOL and e_pts[0] < TOL, (eA, e_pts)
    assert (e_hip <= 2.0 * e_orc + 2e-15).all(), (e_hip, e_orc)


@pytest.mark.skipif(not os.environ.get("SCYTHE_SLOW_TESTS"), reason="5 minutes (the C oracle steps 67 M points at 7 s per step): "
                    "run with SCYTHE_SLOW_TESTS=1; last result in profiles/r02/config5_full_size_state_parity.txt")
def test_config5_full_size_state_against_the_c_oracle():
    """Config 5's grid at its full size (RLZ 1023 x 512 x 128, 67 M points, all-fp64 storage) stepped twice by the HIP path and
    by the C oracle: the spectral state (A coefficients) to 1e-10, and the model fields on sampled cells (innermost, the last
    truncated one, the first full-spectrum one, outermost) against the oracle's evaluation of its own coefficients.  The
    oracle's whole `physical` array (22 GB) is never formed."""
    from oracle import oracle_c as OC
    case = cases.rlz_hrbl(num_cells=341, zDim=128, ring_L=512)
    case["ts"] = 0.02                    # bench.TS_OF: 0.3 m end spacing of the 128-level Chebyshev column
    hip = cases.HipModel(case)
    assert hip.run.tiles[0].N == 1023 * 512 * 128
    orc = cases.OracleModel(case)
    for _ in range(2):
        hip.step()
        orc.step()
    eA = cases.rel_err(hip.A, orc.A)
    g = orc.g
    tile = hip.run.tiles[0]
    tile.tileTransform_()
    assert not tile.check_nan()
    phys = tile.physical
    worst = np.zeros(phys.shape[2])
    for cell in (0, 84, 85, 340):
        ref = OC.TileOracle(g, cell, 1).inverse(orc.A)
        idx = cases.ring_points(g, [3 * cell, 3 * cell + 1, 3 * cell + 2])
        worst = np.maximum(worst, cases.per_slot_errors(phys[idx], ref))
    cases.report_slots("config 5 at full size (RLZ 1023 x 512 x 128, fp64), 2 steps (A coefficients %.1e)" % eA, g,
                       [("HIP vs fp64 oracle on cells 0, 84, 85, 340", worst)])
    # (the derivative slots are reported, not bounded: on ONE cell the slot's own scale is local, and d2/dr2 of two states
    #  that differ by 1e-14 differs by that times 1 / DX^2 - see check_full for how the other full-size tests treat them)
    assert eA < TOL and worst[0] < TOL, (eA, worst)


def test_config5_full_size_node_space_equals_ring_wise_and_tiling_invariance(monkeypatch):
    """fp32 storage at full size: (1) the node-space inverse + cell-wise kernel against the ring-wise kernels everywhere
    (SX_NODE_MODE=0); (2) two radial tiles (transposed solve) against one tile."""
    one = _bench_model(1, workload=C5, storage="f32")
    two = _bench_model(2, workload=C5, storage="f32")
    for _ in range(3):
        one.step()
        two.step()
    f1, f2 = _np1_fields(one), _np1_fields(two)
    e_tiles = _max_rel(f2, f1)
    two.close()
    del f2
    monkeypatch.setenv("SX_NODE_MODE", "0")
    ring = _bench_model(1, workload=C5, storage="f32")
    for _ in range(3):
        ring.step()
    e_ring = _max_rel(_np1_fields(ring), f1)
    print("\nconfig 5 full size, 3 steps (fp32-stored derivative planes): 2 tiles vs 1 tile %.2e; ring-wise vs node-space %.2e" % (e_tiles, e_ring))
    # the state path is fp64 in both arrangements; tiles change only the summation order of B (rounding), the two inverse
    # paths round their fp32-stored derivative planes at different places (ring values vs node values): fp32-mode tolerance
    assert e_tiles < 1e-11, e_tiles
    assert e_ring < 1e-6, e_ring
    one.close()
    ring.close()


def test_config5_full_size_azimuthal_derivative_slots_are_the_spectral_derivatives():
    """tileTransform! at full size with fp32-stored derivative planes: on sampled rings / levels the d/dlambda and d2/dlambda2
    slots equal the FFT derivatives (numpy) of the fp64 value slot to fp32 rounding of the slot (512-point rings, kmax 255)."""
    run = _bench_model(1, workload=C5, storage="f32")
    g = run.tiles[0]
    g.tileTransform_()
    phys = g.physical
    L, nz = 512, 128
    k = np.fft.rfftfreq(L, 1.0 / L)
    worst = 0.0
    for ring in (0, 9, 300, 766, 1022):
        for z in (0, 64, 127):
            for v in (0, 2, 4):
                idx = (ring * L + np.arange(L)) * nz + z
                u = phys[idx, v, 0]
                spec = np.fft.rfft(u)
                dl = np.fft.irfft(1j * k * spec, L)
                dll = np.fft.irfft(-(k ** 2) * spec, L)
                scale = max(np.abs(u).max(), 1e-300)
                e1 = np.abs(phys[idx, v, 3] - dl).max() / max(np.abs(dl).max(), scale)
                e2 = np.abs(phys[idx, v, 4] - dll).max() / max(np.abs(dll).max(), scale)
                worst = max(worst, e1, e2)
    print("\nconfig 5 full size: lambda-derivative slots vs numpy FFT derivative of the value slot, worst %.2e (fp32 storage: 6e-8)" % worst)
    assert worst < 5e-7, worst
    run.close()
