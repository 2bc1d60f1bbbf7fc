"""-m gpu: a torch-free C host on the C ABI (tests/abi_host.c), the stand-in for the Julia `ccall` host of INTEGRATION.md.

The C program is compiled here with gcc against include/scythe_hip.h, dlopens libscythe_hip.so and runs as a FRESH process
with no Python and no torch in it: the HIP runtime and librccl are the ones the library resolves from /opt/rocm.  It steps the
reference's LinearAdvection1D known-answer case; this test compares what it wrote with the CPU oracle (and, for the full 2000
steps, with the notebook's printed values)."""
import json
import os
import subprocess

import numpy as np
import pytest

from tests import cases

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KAT = json.load(open(os.path.join(ROOT, "tests", "golden", "linear_advection_kat.json")))
IDX = KAT["index_0based"]


def _build(tmp_path):
    exe = str(tmp_path / "abi_host")
    subprocess.run(["gcc", "-O1", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), "-o", exe,
                    os.path.join(ROOT, "tests", "abi_host.c"), "-ldl", "-lm"], check=True)
    return exe


def _run(exe, tmp_path, steps, tiles):
    out = str(tmp_path / ("out_%d_%d.bin" % (steps, tiles)))
    env = {k: v for k, v in os.environ.items() if not k.startswith("PYTHON")}
    env.pop("LD_PRELOAD", None)
    env.pop("LD_LIBRARY_PATH", None)          # nothing but the library's own RUNPATH (/opt/rocm) resolves HIP and RCCL
    p = subprocess.run([exe, os.path.join(ROOT, "scythe.jl_amd", "libscythe_hip.so"), str(steps), str(tiles), out],
                       capture_output=True, text=True, timeout=600, env=env)
    assert p.returncode == 0, p.stdout[-1000:] + p.stderr[-2000:]
    raw = np.fromfile(out, dtype=np.float64)
    assert raw.size == 600
    return raw[:300], raw[300:], p.stdout


@pytest.mark.gpu
@pytest.mark.parametrize("tiles", [1, 2])
def test_c_host_50_steps_against_the_oracle(tmp_path, tiles):
    """50 steps of the known-answer case from a plain C process.  tiles = 1 also runs the in-library RCCL exchange with a
    one-rank communicator (librccl bound from /opt/rocm with no torch in the process); tiles = 2 the loopback transport."""
    exe = _build(tmp_path)
    x, u, log = _run(exe, tmp_path, 50, tiles)
    assert ("rccl(1 rank)" if tiles == 1 else "loopback") in log
    assert np.max(np.abs(x[IDX] - np.array(KAT["gridpoints"]))) < 1e-13
    orc = cases.OracleModel(cases.kat_r())
    for _ in range(50):
        orc.step()
    ref = orc.physical()[:, 0, 0]
    assert np.max(np.abs(u - ref)) / np.max(np.abs(ref)) < 1e-12


@pytest.mark.gpu
def test_c_host_reproduces_the_notebook_known_answer(tmp_path):
    """All 2000 steps from the C host: the 25 values notebooks/LinearAdvection_example.ipynb prints."""
    exe = _build(tmp_path)
    _, u, _ = _run(exe, tmp_path, KAT["model"]["steps"], 1)
    assert np.max(np.abs(u[IDX] / np.array(KAT["final_u"]) - 1.0)) < 1e-11


def _write_case(path, case, steps):
    """The descriptors of `case` as tests/abi_host.c's `case` mode reads them (the same fields the Python mirror passes)."""
    import scythe_jl_amd as S
    from scythe_jl_amd import _lib as L
    gp, mp = cases.hip_params(case)
    names = gp.var_names()
    bc = lambda d, dflt: [L.BC[(d or {}).get(n, dflt)] for n in names]
    eq = L.load().sx_equation_set_id(mp.equation_set.encode())
    par = [float(mp.physical_params.get(k, 0.0)) for k in L.PARAM_ORDER]
    g = cases.oracle_grid(case)
    pts = g.gridpoints().reshape(-1, 1 + g.has_l + g.has_z)
    vals = np.asfortranarray(case["ic"](pts), dtype=np.float64)
    with open(path, "wb") as f:
        np.array([L.GEOM[gp.geometry], gp.num_cells, len(names), gp.zDim or 0, gp.ring_uniform_L or 0, eq, 0, steps,
                  gp.vars.get("w", 0), gp.vars.get("xi", 0), gp.vars.get("h", 0)], dtype=np.int32).tofile(f)
        np.array([gp.xmin, gp.xmax, gp.zmin, gp.zmax, mp.ts] + par, dtype=np.float64).tofile(f)
        np.array(bc(gp.BCL, "R0") + bc(gp.BCR, "R0") + bc(gp.BCB, "R0") + bc(gp.BCT, "R0"), dtype=np.int32).tofile(f)
        np.array([vals.shape[0]], dtype=np.int64).tofile(f)
        vals.T.ravel().tofile(f)                      # column-major [point, var]
    return vals.shape[0], len(names)


@pytest.mark.gpu
@pytest.mark.parametrize("maker,kw", [(cases.rlz_hrbl, {"num_cells": 6, "zDim": 32, "ring_L": 32}), (cases.rlz_hrbl, {"num_cells": 5, "zDim": 10}),
                                       (cases.rl_slab, {"num_cells": 8}), (cases.rz_advection, {})])
def test_c_host_runs_any_grid_from_descriptors(tmp_path, maker, kw):
    """The C host's `case` mode: RLZ (uniform rings through the node-space path, native ragged rings), RL and RZ grids with their
    equation sets, created from descriptors in a torch-free process, 3 steps, every derivative slot against the oracle, and
    sx_max_abs against the host-side maximum of the fields it wrote."""
    exe = _build(tmp_path)
    case = maker(**kw)
    cfile, out = str(tmp_path / "case.bin"), str(tmp_path / "case_out.bin")
    npts, nv = _write_case(cfile, case, 3)
    env = {k: v for k, v in os.environ.items() if not k.startswith("PYTHON")}
    env.pop("LD_PRELOAD", None)
    env.pop("LD_LIBRARY_PATH", None)
    p = subprocess.run([exe, "case", os.path.join(ROOT, "scythe.jl_amd", "libscythe_hip.so"), cfile, out], capture_output=True, text=True,
                       timeout=600, env=env)
    assert p.returncode == 0, p.stdout[-1000:] + p.stderr[-2000:]
    raw = np.fromfile(out, dtype=np.float64)
    orc = cases.OracleModel(case)
    for _ in range(3):
        orc.step()
    ref = orc.physical()
    phys = raw[:ref.size].reshape(ref.shape, order="F")
    assert cases.rel_err_per_var(phys, ref) < 1e-10
    assert raw.size == ref.size + nv
