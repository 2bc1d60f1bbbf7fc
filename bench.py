#!/usr/bin/env python3
"""bench.py - model steps/sec of the RLZ 512x256x64 shallow-water configuration on N MI355X.

  python bench.py --gpus N --steps K --warmup W
  N > 1: either launched by `python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...`
  (RANK / WORLD_SIZE in the environment), or invoked plainly - then this process starts that launcher itself as a CHILD
  process before anything here touches the GPU, relays rank 0's JSON line and exits with the launcher's status.

One "step" is one pass of model_loop's body (src/semiimplicit.jl:268-297): tileTransform! -> equation set
(Oneway_ShallowWater_HeightResolvedBL) -> explicit_timestep -> spectralTransform! -> halo/sum -> splineTransform!.
Workload (SURVEY.md 8(d) "perf shape"): 171 radial cells -> 513 rings x 256 azimuthal points x 64 Chebyshev levels,
6 variables, 7 derivative slots, fp64, synthetic vortex initial condition resident in HBM before the timed region.
For N > 1 the 171 cells are split into N radial tiles, one per GPU (strong scaling); the patch-level B -> A solve runs in its
interface-only form (every rank solves its own rows, two RCCL all-to-alls of 10 rows per tile around a small reduced system;
--exchange a2a: the transposed solve, --exchange gather: the reference's protocol - halo rank -> rank+1, all-gather of owned
rows, redundant patch solve).

Prints ONE JSON line on rank 0 with the driver's contract plus "roofline" (dominant kernel, live hipEvent timing) and
"cpu_baseline" (the C oracle "port" on a bounded radial sample of the same workload, rank 0, N = 1 only).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0     # MI355X_MICROARCH.md: HBM3E peak 8.0 TB/s (spec)
HBM_ACHIEVABLE_GBS = 6300.0   # same guide: 6.29 TB/s measured for a float4 copy (79 % of spec)
PROFILE_ROUND = "r04"     # profiles/<round>/pmc_traffic_<workload>.json: the PMC passes of the current build

WORKLOADS = {
    # name: (num_cells, ring_L, zDim)
    "rlz_513x256x64": (171, 256, 64),
    "rlz_small": (24, 64, 16),
    # SURVEY.md 8(d) config 5 (use with --storage f32): 341 cells -> 1023 rings x 512 x 128
    "rlz_1023x512x128": (341, 512, 128),
}
# BASELINE.json configs 2 and 3 (bench_configs.py; one GPU, not the headline): reported under "other_configs" of the headline
# line, or as the line's own workload with --workload
OTHER_WORKLOADS = {
    "r_linear_advection_1d": "config1_r",      # models/LinearAdvection1D.jl (the reference's own CPU-runnable case: plumbing, launch-bound here)
    "rl_cha_bell2024": "config2_literal",      # models/cha_bell2024/Oneway_ShallowWater_Slab.jl:1-40, 100 cells, native rings
    "rz_513x128_semi": "config3_rz",           # RZ 513 x 128, Chebyshev vertical + semiimplicit_adjustment (src/semiimplicit.jl:521-597)
}
PARITY_CLAIM = ("fields within 1e-10 of the CPU oracle: the A coefficients and the VALUE slot of every variable, at this grid's full size over "
                "25 steps (tests/test_gpu_configs.py::test_config4_full_size_25_steps_against_the_c_oracle); the derivative slots "
                "(d/dr .. d2/dz2) are held to 'no less accurate than the fp64 oracle against an extended-precision evaluation of the same "
                "coefficients' and to 10x the spread between two fp64 oracles - two correct fp64 runs differ there by k^2 / N^4 times the last bit")
TS_OF = {"rlz_1023x512x128": 0.02}     # 128 levels: 0.3 m end spacing of the Chebyshev column
VARS6 = {"h": 1, "u": 2, "v": 3, "ub": 4, "vb": 5, "wb": 6}
BCL6 = {"h": "R1T1", "u": "R1T0", "v": "R1T0", "ub": "R1T0", "vb": "R1T0", "wb": "R1T1"}
BCR6 = {"h": "R0", "u": "R1T1", "v": "R0", "ub": "R1T1", "vb": "R0", "wb": "R0"}
BCT6 = {"ub": "R1T1", "vb": "R1T1"}      # boundary-layer winds: zero vertical gradient at the model top
PAR = dict(g=9.81, Kh=5000.0, Cd=2.4e-3, Hfree=2000.0, f=5.0e-5, Um=0.0, Vm=0.0)
# Time step: the equation set is explicit in everything, so ts is bound by (i) horizontal diffusion at the innermost
# ring (Kh ts c / r_min^2 < 0.5 with r_min = 0.11 DX = 198 m -> ts < 1 s) and (ii) the nonlinear vertical mixing on a
# 64-level Chebyshev column whose end spacing is 1.2 m. ts = 0.2 s ran 6000 steps without a NaN (0.25 s: 2000+; 0.5 s
# blows up after 1600). Throughput does not depend on ts.
TS = 0.2


def initial_condition(pts):
    """Rankine vortex (Rmax 50 km, Vmax 50 m/s) with a wave-2 height perturbation (notebooks/Cha_Bell_WCD2024_
    initialization.ipynb cells 5, 10), extended in z with an Ekman-like decay of the boundary-layer winds."""
    r, l, z = pts[:, 0], pts[:, 1], pts[:, 2]
    Rmax, V0 = 5.0e4, 50.0 / 5.0e4
    vbar = np.where(r < Rmax, V0 * r, Rmax * Rmax * V0 / np.maximum(r, 1.0))
    dec = 1.0 - np.exp(-(z + 50.0) / 300.0)
    h = 100.0 * np.exp(-(r / 1.0e5) ** 2) * (1.0 + 0.05 * np.cos(2.0 * l))
    u = 0.5 * np.sin(l) * r / 3.0e5
    return np.stack([h, u, vbar * (1.0 + 0.02 * np.cos(l)), (u - 2.0 * r / 3.0e5) * dec, 0.7 * vbar * dec, 0.0 * r], axis=1)


def grid_kwargs(workload):
    nc, L, nz = WORKLOADS[workload]
    return dict(geometry="RLZ", xmin=0.0, xmax=3.0e5, num_cells=nc, vars=VARS6, BCL=BCL6, BCR=BCR6, BCT=BCT6, zmin=0.0,
                zmax=2000.0, zDim=nz), L


def cpu_baseline(workload, sample_cells, steps):
    """Time the C oracle (oracle/scythe_oracle.c, OpenMP; a restatement, NOT Julia) on the workload.
    sample_cells = 0: the FULL grid (every radial cell), `steps` timed steps after one warm-up step - a measurement.
    sample_cells > 0: that many cells from the middle of the patch with the full azimuthal x vertical extent, plus the
    full-patch B->A solve, scaled to the grid - an extrapolation, kept for quick runs and reported beside the measurement."""
    from oracle import oracle_np as O, oracle_c as OC
    # the CPUs this container may use (the GPU box: a quota of 16 out of 256 visible); an OMP_NUM_THREADS given by the caller wins
    OC.lib().orc_set_num_threads(int(os.environ.get("OMP_NUM_THREADS", OC.usable_cpus())))
    kw, L = grid_kwargs(workload)
    g = O.Grid(kw.pop("geometry"), kw.pop("xmin"), kw.pop("xmax"), kw.pop("num_cells"), kw.pop("vars"), ring_L=L, **kw)

    def timed(ncell):
        c0 = (g.nc - ncell) // 2
        m = OC.ModelOracle(g, "Oneway_ShallowWater_HeightResolvedBL", TS, PAR, tiles=[(c0, ncell)])
        tl = m.tiles[0]
        shared = np.zeros((g.S_patch(), g.V), order="F")
        tl.add_to_shared(tl.forward(initial_condition(tl.pts)), shared)
        m.A = tl.spline_solve(shared)
        m.step()                                    # warm-up (page faults, thread pool)
        t0 = time.perf_counter()
        for _ in range(steps):
            m.step()
        dt = (time.perf_counter() - t0) / steps
        t0 = time.perf_counter()
        for _ in range(steps):
            tl.spline_solve(shared)                 # the full-patch B->A solve is not proportional to the sample
        return dt, (time.perf_counter() - t0) / steps

    cores = OC.lib().orc_num_threads()
    if sample_cells <= 0 or sample_cells >= g.nc:
        dt, _ = timed(g.nc)
        dts, dsolve = timed(18)
        extrap = (dts - dsolve) * g.nc / 18 + dsolve
        return {"value": 1.0 / dt, "unit": "steps/s", "cores": cores, "kind": "port",
                "sample": "the full grid (%d cells = %d rings x %d x %d points, %d vars), %d timed steps after 1 warm-up step: %.3f s/step "
                          "measured (C restatement with OpenMP, not Julia); for comparison an 18-cell sample scaled to the grid "
                          "predicts %.3f s/step" % (g.nc, 3 * g.nc, L, g.zDim, g.V, steps, dt, extrap)}
    dt, dt_solve = timed(sample_cells)
    full = (dt - dt_solve) * g.nc / sample_cells + dt_solve
    return {"value": 1.0 / full, "unit": "steps/s", "cores": cores, "kind": "port",
            "sample": "EXTRAPOLATED from %d of %d radial cells (%d rings x %d x %d points, %d vars), %d timed steps: %.3f s/step on the "
                      "sample of which %.3f s is the full-patch solve; full step = (%.3f - %.3f) * %d/%d + %.3f s"
                      % (sample_cells, g.nc, 3 * sample_cells, L, g.zDim, g.V, steps, dt, dt_solve, dt, dt_solve, g.nc,
                         sample_cells, dt_solve)}


def spawn_ranks(n):
    """`python bench.py --gpus N` without a launcher in front: start `torch.distributed.run` with N ranks of this very
    command line as a child process (never exec: this is called before torch or HIP are imported, and the parent never
    touches the GPU), pass the ranks' stderr through, relay ONE contract line and return the launcher's exit status.
    Watchdog: the in-library RCCL exchange has never run with more than one rank on real hardware (no multi-GPU node was
    available to the builder).  If the job has not finished after SX_BENCH_TIMEOUT seconds (default 420) the launcher's own
    process group - exactly the processes started here - is killed and the job is run ONCE more with the exchange done by
    torch.distributed (`--exchange-impl torch`); the same happens if the job exits with a non-zero status.  The record shows it:
    only the FINAL attempt's JSON line goes to stdout, and it carries the first attempt's outcome (exit status or timeout, the
    tail of its stderr, its own JSON line if it printed one) in config.lib_attempt and in config.exchange_impl.  A second
    failure is the failure."""
    import signal
    import socket
    import subprocess
    import threading

    def attempt(extra):
        s = socket.socket()
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
        s.close()
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
               "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:] + extra
        env = dict(os.environ)
        env.setdefault("OMP_NUM_THREADS", "1")       # what the launcher would set anyway, without its warning
        proc = subprocess.Popen(cmd, env=env, cwd=ROOT, start_new_session=True, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
        out_lines, err_tail = [], []

        def pump(stream, keep, forward):
            for line in stream:
                keep.append(line)
                if forward is not None:
                    forward.write(line)
                    forward.flush()
                    del keep[:-40]

        th = [threading.Thread(target=pump, args=(proc.stdout, out_lines, None), daemon=True),
              threading.Thread(target=pump, args=(proc.stderr, err_tail, sys.stderr), daemon=True)]
        for t in th:
            t.start()
        try:
            rc = proc.wait(timeout=float(os.environ.get("SX_BENCH_TIMEOUT", "420")))
        except subprocess.TimeoutExpired:
            for sig in (signal.SIGTERM, signal.SIGKILL):
                try:
                    os.killpg(proc.pid, sig)             # the session started above: the launcher and its ranks, nothing else
                except ProcessLookupError:
                    break
                try:
                    proc.wait(timeout=15)
                    break
                except subprocess.TimeoutExpired:
                    continue
            rc = None
        for t in th:
            t.join(5)
        line = None
        for ln in out_lines:                             # the contract line: the last stdout line that parses as a JSON object with "metric"
            try:
                d = json.loads(ln)
                if isinstance(d, dict) and "metric" in d:
                    line = d
            except ValueError:
                pass
        return rc, line, "".join(err_tail)[-1500:]

    rc, line, err = attempt([])
    first = None
    if rc != 0 and "--exchange-impl" not in sys.argv[1:]:
        first = {"outcome": "timed out" if rc is None else "exited with status %d" % rc, "stderr_tail": err, "json_line": line}
        print("bench.py: the %d-rank job %s; once more with --exchange-impl torch" % (n, first["outcome"]), file=sys.stderr, flush=True)
        rc, line, err = attempt(["--exchange-impl", "torch"])
    if line is not None:
        if first is not None:
            line.setdefault("config", {})["lib_attempt"] = first
            line["config"]["exchange_impl"] = "torch (retry: the run with the in-library exchange %s)" % first["outcome"]
        print(json.dumps(line), flush=True)
    return 124 if rc is None else rc


def kernel_table(tile, per_step_ms, launches_per_step, pmc):
    """Per-kernel roofline rows: algorithmic bytes of one launch (sx_kernel_bytes: SURVEY.md 8(d)'s accounting, every array a launch
    touches counted once), the hipEvent average of one launch, GB/s and the fraction of the 8 TB/s HBM peak; with PMC passes of
    this very library build (sha256 match) also the measured HBM bytes per launch."""
    rows = {}
    for k in sorted(per_step_ms):
        n = max(launches_per_step.get(k, 1.0), 1e-9)
        ms = per_step_ms[k] / n
        b = tile.kernel_bytes(k)
        row = {"ms_per_launch": ms, "launches_per_step": n, "algorithmic_bytes": b,
               "achieved_GBs": (b / (ms * 1e-3) / 1e9) if ms > 0 and b > 0 else None,
               "frac": (b / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if ms > 0 and b > 0 else None}
        if pmc and k in pmc:
            row["pmc_bytes"] = pmc[k]["hbm_bytes"]
            row["pmc_GBs"] = pmc[k]["hbm_bytes"] / (ms * 1e-3) / 1e9 if ms > 0 else None
        rows[k] = row
    return rows


def load_pmc(S, workload):
    """HBM bytes per launch from the committed rocprofv3 PMC passes of this workload (FETCH_SIZE / WRITE_SIZE, collected and corrected
    as profiles/summarize_pmc.py documents) - only if those passes ran with the very library loaded now: a stale number is worse than none."""
    tfile = os.path.join(ROOT, "profiles", PROFILE_ROUND, "pmc_traffic_%s.json" % workload)
    if not os.path.exists(tfile):
        return None
    import hashlib
    pmc = json.load(open(tfile))
    if pmc.get("_meta", {}).get("lib_sha256") != hashlib.sha256(open(S.LIB_PATH, "rb").read()).hexdigest():
        return None
    return pmc


def time_other_config(S, torch, dev, name, steps, warmup, graph_replay=True):
    """BASELINE.json config 2 / 3 on this GPU: `warmup` steps with an event pair around every kernel (the per-kernel table), then
    `steps` timed steps with no event on the stream (ms_per_step).  Returns the dictionary reported under other_configs."""
    import bench_configs as BC
    case = getattr(BC, OTHER_WORKLOADS[name])()
    mp = BC.model_parameters(S, case)
    run = S.ModelRun(mp, num_tiles=1, device=dev)
    tile = run.tiles[0]
    pts = S.getGridpoints(tile)
    run.set_initial_conditions([case["ic"](pts.reshape(len(pts), -1))])
    for _ in range(20):                               # clocks, caches, first-launch costs
        run.step()
    torch.cuda.synchronize()
    tile.enable_timers(True)
    tile.reset_timers()
    for _ in range(warmup):
        run.step()
    torch.cuda.synchronize()
    tm = tile.timers()
    tile.enable_timers(False)
    per_step = {k: v[0] / max(warmup, 1) for k, v in tm.items() if v[1] > 0}
    launches = {k: v[1] / max(warmup, 1) for k, v in tm.items() if v[1] > 0}
    t0 = time.perf_counter()
    for _ in range(steps):
        run.step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    # the same steps replayed from hipGraphs (sx_step with SX_GRAPH=1: one graph launch per step; bit-identical fields)
    graph = None
    try:
        if not graph_replay:
            raise RuntimeError("skipped (--no-graph-replay)")
        os.environ["SX_GRAPH"] = "1"
        rung = S.ModelRun(mp, num_tiles=1, device=dev)
        os.environ.pop("SX_GRAPH", None)
        rung.set_initial_conditions([case["ic"](pts.reshape(len(pts), -1))])
        for _ in range(20):
            rung.step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            rung.step()
        torch.cuda.synchronize()
        dtg = (time.perf_counter() - t0) / steps
        graph = {"steps_per_s": 1.0 / dtg, "ms_per_step": 1e3 * dtg, "nan": bool(rung.tiles[0].check_nan()),
                 "note": "sx_step with SX_GRAPH=1: one hipGraph launch per step instead of one launch per kernel"}
        rung.close()
    except Exception as e:
        os.environ.pop("SX_GRAPH", None)
        graph = {"steps_per_s": None, "error": repr(e)[:200]}
    rows = kernel_table(tile, per_step, launches, load_pmc(S, name))
    dom = max(per_step.items(), key=lambda kv: kv[1])[0] if per_step else None
    d = tile.dims
    # SURVEY.md 8(d)'s fixed byte model of a step: B_step = 8 V [N (2 D + 5) + 4 S]
    b_step = 8.0 * d.n_vars * (d.n_points * (2 * d.n_derivs + 5) + 4 * d.s_patch)
    out = {"workload": {"r_linear_advection_1d": "R grid, models/LinearAdvection1D.jl verbatim: 100 cells, PERIODIC, %d points, 1 var, 3 derivative slots, fp64" % d.n_points,
                        "rl_cha_bell2024": "RL two-layer shallow-water slab, models/cha_bell2024/Oneway_ShallowWater_Slab.jl verbatim: 100 cells -> 300 native "
                                           "ragged rings (4 + 4 ri points, kmax = ri), %d points x 6 vars, 5 derivative slots, fp64" % d.n_points,
                        "rz_513x128_semi": "RZ %d x %d (radius x Chebyshev levels, b_zDim = zDim), 5 vars, LinearAcousticRZ + semi-implicit adjustment "
                                           "(src/semiimplicit.jl:521-597), fp64" % (d.rDim, d.zDim)}[name],
           "steps_per_s": 1.0 / dt, "ms_per_step": 1e3 * dt, "steps": steps, "nan": bool(tile.check_nan()), "graph_replay": graph,
           "survey_bytes_per_step": b_step, "survey_model_frac_of_hbm_peak": b_step / dt / 1e9 / HBM_PEAK_GBS,
           "kernel_ms_per_step_sum": sum(per_step.values()),
           "roofline": ({"bound": "hbm", "kernel": dom, "achieved": rows[dom]["achieved_GBs"], "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": rows[dom]["frac"], "traffic": rows[dom].get("pmc_bytes"), "avg_launch_ms": rows[dom]["ms_per_launch"],
                         "algorithmic_bytes_per_launch": rows[dom]["algorithmic_bytes"]} if dom else None),
           "kernels_roofline": rows}
    run.close()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--workload", default="rlz_513x256x64", choices=sorted(WORKLOADS) + sorted(OTHER_WORKLOADS))
    ap.add_argument("--exchange", default="iface", choices=["iface", "a2a", "gather"],
                    help="multi-GPU patch solve: interface-only solve (default: tile-local solves, two all-to-alls of 10 rows per tile; "
                         "falls back to a2a when a tile has fewer than 9 cells), transposed all-to-all, or the reference's halo + gather protocol")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="gloo + --one-device rehearses the multi-rank path on a single GPU (not a performance mode)")
    ap.add_argument("--one-device", action="store_true", help="every rank uses cuda:0 (rehearsal only)")
    ap.add_argument("--tile-split", default="cost", choices=["cost", "reference"],
                    help="radial partition for N > 1: 'cost' gives the inner tiles (ring-wise kernels, ~1.6x per ring) fewer "
                         "cells; 'reference' is calcTileSizes' even split")
    ap.add_argument("--storage", default="f64", choices=["f64", "f32", "f32x"],
                    help="f32: derivative slots of `physical` stored as fp32; f32x: also the spectral transform intermediates "
                         "(config 5 'fp32 mixed-precision transforms'; not the headline metric, whose 1e-10 parity bar needs fp64 throughout)")
    ap.add_argument("--exchange-impl", default="lib", choices=["lib", "torch"],
                    help="N > 1: 'lib' = ncclSend/Recv/AllGather issued inside libscythe_hip.so on the tile's stream (sx_exchange), "
                         "'torch' = torch.distributed collectives on device tensors (always used with --backend gloo)")
    ap.add_argument("--schedule", default="overlap", choices=["overlap", "serial"],
                    help="N = 1 headline: 'overlap' (default) times `value` with the inner-ring chain on a second stream (SX_OVERLAP=1: the same "
                         "kernels and bit-identical fields, the latency-bound FFT kernels fill in under the bandwidth-bound equation-set kernels) "
                         "and takes `roofline` from a separate serial pass of the same length; 'serial' times `value` on one stream")
    ap.add_argument("--no-selfcheck", action="store_true", help="N > 1: skip the 2-step comparison of the two exchange implementations")
    ap.add_argument("--no-native", action="store_true", help="skip the native-ragged-ring run reported as native_equivalent")
    ap.add_argument("--no-other-configs", action="store_true", help="skip BASELINE.json configs 2 and 3 (other_configs)")
    ap.add_argument("--no-graph-replay", action="store_true", help="other configs: skip the second run that replays the steps from hipGraphs")
    ap.add_argument("--no-kernel-timers", action="store_true", help="diagnostic: no hipEvent pair per kernel anywhere (no roofline object)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample-cells", type=int, default=0, help="0 = time the C port on the full grid (default); n > 0 = extrapolate from n cells")
    ap.add_argument("--cpu-steps", type=int, default=3)
    args = ap.parse_args()

    # dmabuf IPC for RCCL / cross-process device buffers; must be in the environment before the HIP runtime starts
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if args.gpus > 1 and "RANK" not in os.environ:
        raise SystemExit(spawn_ranks(args.gpus))
    import torch
    import scythe_jl_amd as S

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d (launch with torch.distributed.run)" % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP path has no CPU fallback")
    if args.one_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    if args.workload in OTHER_WORKLOADS:
        # configs 2 / 3 as the line's own workload (one GPU): value = that configuration's steps/s, labelled as not the headline
        if world != 1:
            raise SystemExit("bench.py: --workload %s is a one-GPU configuration" % args.workload)
        r = time_other_config(S, torch, dev, args.workload, args.steps, max(args.warmup, 1), graph_replay=not args.no_graph_replay)
        out = {"metric": "model steps/sec, %s (BASELINE.json config, not the headline configuration)" % args.workload,
               "value": None if r["nan"] else r["steps_per_s"], "unit": "steps/s", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup,
               "ms_per_step": r["ms_per_step"], "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f64",
               "data": "synthetic", "config": {"workload": r["workload"], "nan": r["nan"], "survey_bytes_per_step": r["survey_bytes_per_step"],
                                                 "survey_model_frac_of_hbm_peak": r["survey_model_frac_of_hbm_peak"]},
               "roofline": r["roofline"], "kernels_roofline": r["kernels_roofline"]}
        print(json.dumps(out), flush=True)
        raise SystemExit(3 if r["nan"] else 0)

    dist = None
    if world > 1:
        import torch.distributed as dist
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group("gloo")

    kw, L = grid_kwargs(args.workload)
    gp = S.GridParameters(ring_uniform_L=L, storage=args.storage, **kw)
    mp = S.ModelParameters(ts=TS_OF.get(args.workload, TS), equation_set="Oneway_ShallowWater_HeightResolvedBL", grid_params=gp,
                           physical_params=dict(PAR))
    # the in-library exchange needs RCCL, i.e. one GPU per rank - or, for the one-GPU rehearsal of this very code path, the
    # stand-in transport of the tests (SX_RCCL_LIB=tests/fake_rccl.cpp's .so) together with --backend gloo
    impl = args.exchange_impl if (world > 1 and (args.backend == "nccl" or os.environ.get("SX_RCCL_LIB"))) else "torch"
    preheat = None
    if args.exchange == "iface" and world > 1:
        # the interface-only solve needs 6 free spline coefficients per tile (up to 3 more cells where a rank-3 boundary condition
        # takes rows away): decided from the tile table, i.e. identically on every rank
        if min(S.PatchLayout(gp, world, split=args.tile_split).ncells) < 9:
            args.exchange = "a2a"
    selfcheck = None

    def make_run(which, overlap=False):
        # SX_OVERLAP is read by sx_create: the handle keeps its schedule for life
        if overlap:
            os.environ["SX_OVERLAP"] = "1"
        r = None
        try:
            r = S.ModelRun(mp, num_tiles=world, rank=rank, device=dev, use_dist=world > 1, exchange=args.exchange, split=args.tile_split,
                           impl=which)
            r.set_initial_conditions([initial_condition(S.getGridpoints(r.tiles[0]))])
        except Exception:
            if r is not None:
                r.close()                    # a half-built run must not keep its tiles (and their device memory) alive
            raise
        finally:
            if overlap:
                os.environ.pop("SX_OVERLAP", None)
        return r

    if world > 1 and impl == "lib":
        # LibExchange itself makes the ranks agree before its first collective (probe on every rank -> all_gather of the
        # outcome -> unique-id broadcast -> ncclCommInitRank), so it raises on EVERY rank or on none and all ranks take the same
        # branch here; a failure inside ncclCommInitRank itself is fatal for the job, as for any RCCL program.
        run, err = None, ""
        try:
            run = make_run("lib")
        except RuntimeError as e:       # RCCL could not be bound / the exchange buffers not set up, on some rank
            err = str(e)[:160]
        if run is None:
            print("bench.py: in-library exchange unavailable (%s), using torch.distributed" % err, file=sys.stderr)
            impl = "torch (lib failed: %s)" % err
            run = make_run("torch")
        if impl == "lib" and not args.no_selfcheck:
            # both implementations of the exchange, two steps each from the same initial condition: same fields.  One after
            # the other with a device-wide wait in between: the library's communicator and torch's are never in flight at
            # the same time (two communicators whose kernels meet in different orders on different ranks can deadlock)
            chk = make_run("lib")
            for _ in range(2):
                chk.step()
            torch.cuda.synchronize()
            a = chk.tiles[0].var_np1
            chk.close()
            dist.barrier()
            other = make_run("torch")
            for _ in range(2):
                other.step()
            torch.cuda.synchronize()
            b = other.tiles[0].var_np1
            selfcheck = float(max(np.abs(a[:, v] - b[:, v]).max() / max(np.abs(b[:, v]).max(), 1e-300) for v in range(a.shape[1])))
            del a, b
            # `other` keeps stepping for a while and is closed only after the timed region: as at N = 1 (the native run below), the
            # device should be in its working state, not 5 ms out of idle, when the W warm-up steps of the timed run begin
            for _ in range(100):
                other.step()
            torch.cuda.synchronize()
            dist.barrier()
            preheat = other
    else:
        run = make_run(impl if world > 1 else "torch")
    tile = run.tiles[0]
    headline = (world == 1 and args.workload == "rlz_513x256x64")
    # N = 1 headline: the SAME work on two streams (SX_OVERLAP=1: inner-ring chain k_rl_inverse -> k_phys_hrbl_inner beside
    # k_node_fft -> k_phys_hrbl; bit-identical fields, tests/test_gpu_configs.py::test_config4_full_size_two_stream_modes_are_bit_identical) is the faster schedule and
    # what `value` is timed with; the dominant kernel then shares the chip, so its event-timed duration (the roofline measurement) comes
    # from a serial pass of the same K steps on `run`, after the timed region.  --schedule serial times `value` on `run` itself.
    run_ov = None
    if world == 1 and args.schedule == "overlap" and args.storage == "f64":
        run_ov = make_run("torch", overlap=True)
        # first-use costs of the second stream (its creation, the fork / join events, the first launches behind them: 200 us of host
        # time per step over the first steps, profiles/r04/overlap_steps.txt) are set-up, not throughput: 10 steps here, before anything is timed
        for _ in range(10):
            run_ov.step()
        torch.cuda.synchronize()

    # The same model on Springsteel's NATIVE ragged rings (SURVEY.md 8(d) "native-equivalent shape": 85 cells -> 255 rings of
    # 4 + 4 ri points, 131,580 horizontal points x 64 levels): the layout a drop-in must run; its azimuthal transforms are dense
    # truncated DFTs on the f64 matrix cores (sx_dft.hip).  Timed in this process BEFORE the headline loop, on purpose: with
    # the driver's `--warmup 5 --steps 20` the 20 timed steps would otherwise start 5 ms after the GPU left idle and run 4-5 %
    # below the steady state.  Measured step by step from an idle GPU (profiles/step_times.py, profiles/r03/step_times_from_idle.txt):
    # 1.02, 1.07, 1.16, 1.17, 1.18, 1.16, 1.13 ... ms, 1.00 ms from step 25 on, 0.99-1.01 ms steady - the device's power state,
    # not the code (929-942 steps/s at W = 5 against 977-980 at W = 50 or 200).  100 timed steps (a quarter of a second): with 30
    # the first bench run on a freshly acquired box still read 957 against 970-973 for the next ones; with 100, 971 like the rest.
    # After this run the GPU is in its working state
    # when the W warm-up steps begin.  `--no-native` skips it.
    native, runn = None, None
    if headline and rank == 0 and not args.no_native:
        try:
            kwn, _ = grid_kwargs(args.workload)
            kwn["num_cells"] = 85
            gpn = S.GridParameters(ring_uniform_L=0, storage=args.storage, **kwn)
            mpn = S.ModelParameters(ts=TS, equation_set="Oneway_ShallowWater_HeightResolvedBL", grid_params=gpn, physical_params=dict(PAR))
            runn = S.ModelRun(mpn, num_tiles=1, device=dev)
            runn.set_initial_conditions([initial_condition(S.getGridpoints(runn.tiles[0]))])
            for _ in range(10):
                runn.step()
            torch.cuda.synchronize()
            nsteps = 100 if args.steps >= 20 else max(5, args.steps)      # a quarter of a second of device time: also the pre-heat
            t1 = time.perf_counter()
            for _ in range(nsteps):
                runn.step()
            torch.cuda.synchronize()
            dtn = (time.perf_counter() - t1) / nsteps
            native = {"steps_per_s": 1.0 / dtn, "ms_per_step": 1e3 * dtn, "steps": nsteps, "nan": bool(runn.tiles[0].check_nan()),
                      "workload": "RLZ 85 cells -> 255 native ragged rings (4 + 4 ri points, kmax = ri), %d points x 6 vars" % runn.tiles[0].N}
            # (closed after the headline loop: freeing its 3 GB is a device-wide wait of tens of milliseconds)
        except Exception as e:
            native = {"steps_per_s": None, "error": repr(e)[:200]}

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    # Warm-up of `run` (serial schedule), with every kernel timed (rank 0): finds the dominant kernel and gives the per-kernel table.
    # An event pair costs ~4 us on the stream, 16 pairs per step were 6 % of the step - so a TIMED region carries the pair of the
    # dominant kernel only (the roofline object needs that one, measured live over K steps), or none at all (the overlapped schedule).
    use_timers = (rank == 0 and not args.no_kernel_timers)
    tile.enable_timers(use_timers)
    tile.reset_timers()
    for _ in range(args.warmup):
        run.step()
    barrier()
    warm = tile.timers() if use_timers else {}
    all_kernels = {k: v[0] / max(args.warmup, 1) for k, v in warm.items() if v[1] > 0}
    launches = {k: v[1] / max(args.warmup, 1) for k, v in warm.items() if v[1] > 0}
    dominant = max(all_kernels.items(), key=lambda kv: kv[1])[0] if all_kernels and args.warmup > 0 else None
    tile.timer_only(dominant)
    tile.reset_timers()

    serial = None
    if run_ov is not None:
        # ---- the serial pass FIRST: K steps on one stream with the dominant kernel's event pair -> roofline (and the device stays busy)
        t0 = time.perf_counter()
        for _ in range(args.steps):
            run.step()
        barrier()
        el_serial = time.perf_counter() - t0
        serial = {"steps_per_s": args.steps / el_serial, "ms_per_step": 1e3 * el_serial / args.steps, "steps": args.steps,
                  "note": "one stream, the dominant kernel's hipEvent pair on it: the pass `roofline` is measured in (it runs before the timed region)"}
        # ---- THE timed region: W warm-up steps, then exactly K steps of the two-stream schedule, no event on any stream
        for _ in range(args.warmup):
            run_ov.step()
        barrier()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            run_ov.step()
        barrier()
        elapsed = time.perf_counter() - t0
        nan = bool(run_ov.tiles[0].check_nan()) or bool(tile.check_nan())
    else:
        t0 = time.perf_counter()
        for _ in range(args.steps):
            run.step()
        barrier()
        elapsed = time.perf_counter() - t0
        nan = tile.check_nan()
    timers = {k: v for k, v in tile.timers().items() if v[1] > 0}
    if runn is not None:
        runn.close()
    if preheat is not None:
        preheat.close()
    if run_ov is not None:
        run_ov.close()
    tile.enable_timers(False)
    tile.timer_only(None)

    if dist is not None:
        tt = torch.tensor([elapsed], dtype=torch.float64, device="cuda" if args.backend == "nccl" else "cpu")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())

    if rank == 0:
        ms_per_step = 1e3 * elapsed / args.steps
        # dominant kernel by accumulated device time
        if not timers:                      # --no-kernel-timers (diagnostic)
            timers = {"none": (0.0, 0)}
        name, (ms, calls) = max(timers.items(), key=lambda kv: kv[1][0])
        avg_ms = ms / max(calls, 1)
        bytes_per_launch = tile.kernel_bytes(name) if calls else 0.0
        achieved = bytes_per_launch / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
        nc, _, nz = WORKLOADS[args.workload]
        pmc = load_pmc(S, args.workload) if (world == 1 and args.storage == "f64") else None
        traffic = pmc.get(name, {}).get("hbm_bytes") if pmc else None
        step_traffic = sum(pmc[k]["hbm_bytes"] * launches.get(k, 0.0) for k in all_kernels if k in pmc) if pmc else None
        rows = kernel_table(tile, all_kernels, launches, pmc) if all_kernels else {}
        if name in rows:                    # the dominant kernel's row from the timed pass (K steps), not from the warm-up
            rows[name].update({"ms_per_launch": avg_ms, "achieved_GBs": achieved, "frac": achieved / HBM_PEAK_GBS,
                               "pmc_GBs": (traffic / (avg_ms * 1e-3) / 1e9) if traffic and avg_ms > 0 else None, "timed_over": "%d steps" % args.steps})
        out = {
            "metric": ("model steps/sec, RLZ 512x256x64 shallow-water" if args.workload == "rlz_513x256x64"
                       else "model steps/sec, %s shallow-water (not the headline configuration)" % args.workload),
            "value": args.steps / elapsed,
            "unit": "steps/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": ms_per_step,
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": {"f64": "f64", "f32": "f64 arithmetic and state, fp32-stored derivative planes",
                      "f32x": "f64 arithmetic and state, fp32-stored derivative planes and spectral transform intermediates"}[args.storage],
            "data": "synthetic",
            "config": {"workload": "RLZ %dx%dx%d (rings x azimuth x levels), 6 vars, 7 derivative slots, "
                                   "Oneway_ShallowWater_HeightResolvedBL, uniform ring table kmax<=%d, b_zDim %d"
                                   % (3 * nc, L, nz, L // 2 - 1, int(tile.dims.b_zDim)),
                       "num_cells": nc, "tiles": world, "tile_cells": list(run.layout.ncells), "exchange": run.exchange_kind,
                       "exchange_impl": (impl if world > 1 else "none"), "exchange_selfcheck_max_rel_diff": selfcheck,
                       "schedule": ("two streams (SX_OVERLAP=1): the inner-ring chain beside the node-space chain; the same kernels, "
                                    "bit-identical fields" if run_ov is not None else "one stream"),
                       # what ran on the device right before the W warm-up steps (from an idle GPU the first ~25 steps of
                       # any run are up to 18 % slow)
                       "device_busy_before_warmup": ("native_equivalent run, then the serial roofline pass" if (native is not None and serial is not None) else "native_equivalent run" if native is not None else
                                                     "self-check's torch.distributed run, 100 steps" if preheat is not None else "nothing (cold start)"),
                       "ts": TS_OF.get(args.workload, TS), "nan": bool(nan),
                       "parity": PARITY_CLAIM if args.storage == "f64" else "declared tolerance of the fp32-storage mode: values 1e-6, derivative slots 5e-5 (DESIGN.md 3)"},
            "roofline": {"bound": "hbm", "kernel": name, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "avg_launch_ms": avg_ms, "algorithmic_bytes_per_launch": bytes_per_launch,
                         "measured_in": ("the serial pass (serial_schedule), %d steps, run right before the timed region" % args.steps if serial is not None else "the timed region"),
                         "achievable_peak": HBM_ACHIEVABLE_GBS, "frac_of_achievable": achieved / HBM_ACHIEVABLE_GBS,
                         # whole step: PMC bytes of every kernel of a step / step time (null without matching PMC passes)
                         "step_traffic": step_traffic,
                         "step_achieved": (step_traffic / (ms_per_step * 1e-3) / 1e9) if step_traffic else None,
                         "step_frac": (step_traffic / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS) if step_traffic else None},
            # every kernel of a step: algorithmic bytes, PMC bytes (matching build only), launch time, fraction of the HBM peak - from the
            # warm-up steps of the serial schedule (all event pairs on); the dominant kernel's row from its K timed steps
            "kernels_roofline": rows,
            "kernels_ms_per_step": {k: v for k, v in sorted(all_kernels.items())},
            "dominant_kernel_ms_timed_region": {k: v[0] / args.steps for k, v in sorted(timers.items())},
        }
        if serial is not None:
            out["serial_schedule"] = serial
        if native is not None:
            out["native_equivalent"] = native
        if headline and not args.no_native and not nan:
            # NOT the headline: the same model with SX_DEFER_DIAG=1 - the diagnostic variable w (written by the equation set
            # before it is read, so its spline coefficients are consumed by output only) skips the forward transform and the
            # solve inside the step and is brought up to date when something reads A or B; every observable is bit-identical
            # (tests/test_gpu_parity.py::test_deferred_diagnostic_*).  The reference transforms all six variables every step,
            # and so does the headline number above.
            try:
                os.environ["SX_DEFER_DIAG"] = "1"
                rund = S.ModelRun(mp, num_tiles=1, device=dev)
                del os.environ["SX_DEFER_DIAG"]
                rund.set_initial_conditions([initial_condition(S.getGridpoints(rund.tiles[0]))])
                for _ in range(20):
                    rund.step()
                torch.cuda.synchronize()
                nd = max(20, min(args.steps, 100))
                t1 = time.perf_counter()
                for _ in range(nd):
                    rund.step()
                torch.cuda.synchronize()
                dtd = (time.perf_counter() - t1) / nd
                out["deferred_diagnostic"] = {"steps_per_s": 1.0 / dtd, "ms_per_step": 1e3 * dtd, "steps": nd, "nan": bool(rund.tiles[0].check_nan()),
                                              "note": "opt-in (SX_DEFER_DIAG=1), not the headline: w's forward transform and solve on demand"}
                rund.close()
            except Exception as e:
                os.environ.pop("SX_DEFER_DIAG", None)
                out["deferred_diagnostic"] = {"steps_per_s": None, "error": repr(e)[:200]}
        if headline and not args.no_other_configs:
            # BASELINE.json configs 2 and 3 ("1 MI355X" configurations; parity: tests/test_gpu_configs.py) - reported, never `value`
            out["other_configs"] = {}
            for oname in sorted(OTHER_WORKLOADS):
                try:
                    out["other_configs"][oname] = time_other_config(S, torch, dev, oname, 200 if args.steps >= 20 else max(5, args.steps), 20)
                except Exception as e:
                    out["other_configs"][oname] = {"steps_per_s": None, "error": repr(e)[:200]}
        if world == 1 and not args.no_cpu_baseline:
            try:
                out["cpu_baseline"] = cpu_baseline(args.workload, args.cpu_sample_cells, args.cpu_steps)
            except Exception as e:   # the baseline is a reported side figure; never lose the GPU line over it
                out["cpu_baseline"] = {"value": None, "unit": "steps/s", "cores": 0, "kind": "port", "sample": "failed: %r" % (e,)}
        if nan:
            # a run that blew up is not a throughput measurement: no value, non-zero exit
            out["value"] = None
            out["error"] = "NaN in the model state after the timed steps (checkCFL)"
        if selfcheck is not None and not (selfcheck < 1e-10):
            out["value"] = None
            out["error"] = "in-library RCCL exchange and torch.distributed exchange disagree: %g" % selfcheck + (
                "; " + out["error"] if "error" in out else "")
            nan = True
        print(json.dumps(out), flush=True)
    run.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if nan:
        raise SystemExit(3)


if __name__ == "__main__":
    main()
