"""-m gpu: two ranks on the one GPU, torch.distributed over gloo (device buffers staged through the host): the
whole multi-rank step path of bench.py - tile per rank, pack / all-to-all / solve / all-to-all / unpack, or halo +
all-gather - against the one-patch oracle; the in-library RCCL exchange with a one-rank communicator on the one GPU
(RCCL refuses two ranks on one device) and, where two GPUs exist, bench.py over RCCL with its exchange self-check."""
import os
import socket

import numpy as np
import pytest
import torch.multiprocessing as mp

from tests import cases

pytestmark = pytest.mark.gpu



def _join_all(procs, seconds):
    """Wait for every rank, then end exactly the ranks started here that are still alive (a dead or hung rank must not leave its
    peers spinning in a barrier with the GPU in their hands), and only then look at the exit codes."""
    for p in procs:
        p.join(seconds)
    for p in procs:
        if p.is_alive():
            p.terminate()
            p.join(20)
            if p.is_alive():
                p.kill()
                p.join(10)
    assert [p.exitcode for p in procs] == [0] * len(procs)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, exchange, q, case_kw=None):
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import scythe_jl_amd as S
        from tests import cases
        torch.cuda.set_device(0)
        case = cases.rlz_hrbl(**(case_kw or dict(num_cells=18 if exchange == "iface" else 8, zDim=12, ring_L=32)))
        gp, mp_ = cases.hip_params(case)
        run = S.ModelRun(mp_, num_tiles=world, rank=rank, device=torch.device("cuda", 0), use_dist=True, exchange=exchange)
        tile = run.tiles[0]
        pts = S.getGridpoints(tile)
        run.set_initial_conditions([case["ic"](pts)])
        for _ in range(3):
            run.step()
        phys = run.physical()
        orc = cases.OracleModel(case)
        for _ in range(3):
            orc.step()
        ref = orc.physical()
        p0 = sum(int(run.layout.tile_sizes[4, t]) for t in range(rank))
        err = cases.rel_err_per_var(phys, ref[p0:p0 + tile.N])
        q.put((rank, float(err)))
        run.close()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("exchange", ["a2a", "gather", "iface"])
def test_two_ranks_one_gpu(exchange):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, exchange, q)) for r in range(2)]
    for p in procs:
        p.start()
    _join_all(procs, 300)
    res = dict(q.get(timeout=5) for _ in range(2))
    assert max(res.values()) < 1e-10


def test_four_ranks_one_gpu_uneven_tiles_node_space_path():
    """Four ranks (the most a GPU box lets share one card with headroom), 14 cells -> tiles of 4, 4, 3, 3 cells, 32 levels
    on uniform rings: rank 0 runs ring-wise, the others partly or wholly through the node-space inverse."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    kw = dict(num_cells=14, zDim=32, ring_L=16)
    procs = [ctx.Process(target=_worker, args=(r, 4, port, "a2a", q, kw)) for r in range(4)]
    for p in procs:
        p.start()
    _join_all(procs, 600)
    res = dict(q.get(timeout=5) for _ in range(4))
    assert max(res.values()) < 1e-10


@pytest.mark.gpu
def test_bench_cli_two_ranks_on_one_device():
    """The driver's launch line for N = 2 (torch.distributed.run, one rank per process) with the gloo rehearsal switches:
    bench.py must print one JSON line with the contract's keys, n_gpus = 2 and a finite state."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
           "--workload", "rlz_small", "--backend", "gloo", "--one-device"]
    out = subprocess.run(cmd, cwd=root, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    line = [ln for ln in out.stdout.splitlines() if ln.startswith("{")][-1]
    d = json.loads(line)
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline"):
        assert key in d, key
    assert d["n_gpus"] == 2 and d["steps"] == 3 and d["value"] > 0 and d["config"]["nan"] is False
    assert d["config"]["exchange"] == "iface" and "cpu_baseline" not in d      # rlz_small: 24 cells, 12 per rank


@pytest.mark.gpu
def test_bench_cli_two_ranks_self_spawned():
    """`python3 bench.py --gpus 2 ...` exactly as the driver types it, with NO launcher in front: bench.py starts its own two
    ranks as child processes (before it touches the GPU) and hands back rank 0's contract line."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--workload", "rlz_small",
           "--backend", "gloo", "--one-device"]
    out = subprocess.run(cmd, cwd=root, capture_output=True, text=True, timeout=600, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 3 and d["value"] > 0 and d["config"]["nan"] is False and d["config"]["tiles"] == 2


@pytest.mark.gpu
@pytest.mark.parametrize("exchange", ["a2a", "gather", "iface"])
@pytest.mark.parametrize("maker,kw", [(cases.rlz_hrbl, {"num_cells": 8, "zDim": 32, "ring_L": 32}), (cases.rl_slab, {"num_cells": 9}),
                                       (cases.rz_semiimplicit, {"num_cells": 9})])
def test_in_library_rccl_exchange_world_size_one(maker, kw, exchange):
    """sx_comm_init / sx_exchange with a ONE-rank RCCL communicator on the one GPU: the whole code path of the in-library
    exchange (dlopen of librccl, ncclCommInitRank, grouped ncclSend / ncclRecv to self or the in-place ncclAllGather, pack /
    solve / unpack on the handle's stream) against the plain single-tile run.  RCCL refuses two ranks on one device, so
    more ranks need more GPUs (next test)."""
    import scythe_jl_amd as S
    case = maker(**kw)
    ref = cases.HipModel(case)
    gp, mp = cases.hip_params(case)
    run = S.ModelRun(mp, num_tiles=1, rank=0, device="cuda", use_dist=True, exchange=exchange, impl="lib",
                     unique_id=S.comm_unique_id())
    pts = S.getGridpoints(run.tiles[0])
    run.set_initial_conditions([case["ic"](pts.reshape(len(pts), -1))])
    for _ in range(4):
        ref.step()
        run.step()
    a, b = run.physical(), ref.physical()
    assert np.isfinite(a).all()
    assert cases.rel_err_per_var(a, b) < 1e-12
    run.close()


@pytest.mark.gpu
def test_bench_cli_two_ranks_rccl():
    """N = 2 over RCCL on two GPUs (skipped on the one-GPU box): bench.py's self-check steps the in-library exchange and the
    torch.distributed one side by side and reports their largest relative difference."""
    import json
    import subprocess
    import sys
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("needs 2 GPUs")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    for exchange in ("iface", "a2a", "gather"):
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
               "--master-port", str(port), os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
               "--workload", "rlz_small", "--exchange", exchange]
        out = subprocess.run(cmd, cwd=root, capture_output=True, text=True, timeout=600)
        assert out.returncode == 0, out.stderr[-2000:]
        d = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][-1])
        assert d["n_gpus"] == 2 and d["value"] > 0 and d["config"]["exchange_impl"] == "lib"
        assert d["config"]["exchange_selfcheck_max_rel_diff"] < 1e-12


def _drawn_case(seed, index):
    """draw number `index` of tests/test_gpu_fuzz.py's sequence for `seed` with more than one tile: (case, tiles, exchange)"""
    import numpy as np
    from tests import test_gpu_fuzz as F
    rng = np.random.default_rng(seed)
    k = -1
    while True:
        case, tiles, exchange, _ = F.draw(rng)
        if tiles > 1:
            k += 1
            if k == index:
                return case, tiles, exchange


# ----------------------------------------------------------------------------- the in-library exchange with MORE THAN ONE rank
def _fake_worker(rank, world, lib_path, uid, exchange, case_name, case_kw, q):
    """One process = one tile = one rank of sx_comm_init / sx_exchange; no torch.distributed anywhere: the only thing the ranks
    share is the 128-byte id, as in the Julia host of INTEGRATION.md 4."""
    os.environ["SX_RCCL_LIB"] = lib_path            # before the library binds its transport
    import scythe_jl_amd as S
    from tests import cases
    if case_name == "fuzz":                        # a drawn configuration (closures do not cross a spawn: every rank redraws it)
        case = _drawn_case(**case_kw)[0]
    else:
        case = getattr(cases, case_name)(**case_kw)
    gp, mp_ = cases.hip_params(case)
    run = S.ModelRun(mp_, num_tiles=world, rank=rank, device="cuda", use_dist=True, exchange=exchange, impl="lib", unique_id=uid)
    tile = run.tiles[0]
    pts = S.getGridpoints(tile)
    run.set_initial_conditions([case["ic"](pts.reshape(len(pts), -1))])
    for _ in range(3):
        run.step()
    phys = run.physical()
    orc = cases.OracleModel(case)
    for _ in range(3):
        orc.step()
    ref = orc.physical()
    p0 = sum(int(run.layout.tile_sizes[4, t]) for t in range(rank))
    q.put((rank, float(cases.rel_err_per_var(phys, ref[p0:p0 + tile.N]))))
    run.close()


@pytest.mark.gpu
@pytest.mark.parametrize("exchange,world,case_name,case_kw", [
    ("iface", 2, "rlz_hrbl", {"num_cells": 18, "zDim": 12, "ring_L": 32}),
    ("iface", 4, "rlz_hrbl", {"num_cells": 36, "zDim": 32, "ring_L": 16}),
    ("iface", 3, "kat_r", {"num_cells": 40}),                                  # PERIODIC: wrap-around rows between the first and the last rank
    ("a2a", 3, "rlz_hrbl", {"num_cells": 14, "zDim": 12, "ring_L": 32}),
    ("gather", 3, "rl_slab", {"num_cells": 12}),
])
def test_in_library_exchange_multi_rank_through_a_stand_in_transport(tmp_path, exchange, world, case_name, case_kw):
    """sx_comm_init / sx_exchange with 2-4 RANKS IN SEPARATE PROCESSES on the one GPU: every rank-dependent branch of the
    in-library exchange (csrc/sx_comm.cpp: offsets by rank, the grouped send / receive loops to every peer, the halo chain
    rank -> rank + 1, the in-place all-gather) against the one-patch oracle.  RCCL refuses two ranks on one device, so the
    library is pointed (SX_RCCL_LIB) at tests/fake_rccl.cpp, a stand-in with RCCL's entry points that moves the messages
    through shared memory - it validates this library's use of the API, not RCCL or xGMI."""
    import subprocess
    lib = str(tmp_path / "libfake_rccl.so")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    subprocess.run(["/opt/rocm/bin/hipcc", "-O1", "-shared", "-fPIC", "-o", lib, os.path.join(root, "tests", "fake_rccl.cpp"), "-lrt"],
                   check=True, capture_output=True)
    uid = ("/sxfake_%d_%s" % (os.getpid(), os.urandom(4).hex())).encode().ljust(128, b"\0")
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_fake_worker, args=(r, world, lib, uid, exchange, case_name, case_kw, q)) for r in range(world)]
    for p in procs:
        p.start()
    _join_all(procs, 600)
    res = dict(q.get(timeout=5) for _ in range(world))
    assert sorted(res) == list(range(world))
    assert max(res.values()) < 1e-10, res


@pytest.mark.gpu
def test_bench_cli_two_ranks_in_library_exchange_through_the_stand_in_transport(tmp_path):
    """bench.py's N = 2 path as the driver starts it (no launcher), with the exchange INSIDE the library: the probe / agree /
    unique-id broadcast of LibExchange, the sequential self-check against the torch.distributed exchange and the timed loop -
    on the one GPU, with tests/fake_rccl.cpp standing in for RCCL and gloo for torch's side."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    lib = str(tmp_path / "libfake_rccl.so")
    subprocess.run(["/opt/rocm/bin/hipcc", "-O1", "-shared", "-fPIC", "-o", lib, os.path.join(root, "tests", "fake_rccl.cpp"), "-lrt"],
                   check=True, capture_output=True)
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["SX_RCCL_LIB"] = lib
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--workload", "rlz_small",
           "--backend", "gloo", "--one-device", "--exchange-impl", "lib"]
    out = subprocess.run(cmd, cwd=root, capture_output=True, text=True, timeout=600, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    d = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][-1])
    assert d["n_gpus"] == 2 and d["value"] > 0 and d["config"]["exchange_impl"] == "lib" and d["config"]["exchange"] == "iface"
    assert d["config"]["exchange_selfcheck_max_rel_diff"] < 1e-12


@pytest.mark.gpu
@pytest.mark.parametrize("mode", [0, 1, 2])
def test_exchange_set_up_rejects_bad_tile_tables_in_every_mode(mode):
    """The tile table is validated before any offset table or buffer is built, whatever the protocol (a gap, an overlap or a
    tile of fewer than 3 cells used to reach the halo / gather offset arithmetic of mode 1 unchecked), a failed set-up leaves
    no exchange state behind, and sx_exchange without a communicator says so."""
    import ctypes as C
    import scythe_jl_amd as S
    from scythe_jl_amd import _lib as L
    case = cases.rl_slab(num_cells=24)
    gp, mp_ = cases.hip_params(case)
    lib = L.load()
    tiles = [S.Grid(gp, mp_, c0, n, t + 2) for t, (c0, n) in enumerate([(0, 12), (12, 12)])]
    hs = (C.c_void_p * 2)(*[g._h for g in tiles])
    arr = lambda v: (C.c_int32 * 2)(*v)
    try:
        for cell0, ncells, msg in (([0, 13], [12, 11], "contiguous"), ([0, 12], [12, 11], "cover the patch"), ([0, 11], [11, 13], "match this handle")):
            assert lib.sx_comm_init_local(hs, 2, arr(cell0), arr(ncells), mode) != 0
            assert msg in lib.sx_last_error().decode(), lib.sx_last_error()
        assert lib.sx_exchange_local(hs, 2) != 0 and b"sx_comm_init_local first" in lib.sx_last_error()
        assert lib.sx_comm_prepare(tiles[0]._h, 2, 0, arr([0, 12]), arr([12, 12]), mode) == 0       # non-collective part alone
        assert lib.sx_exchange(tiles[0]._h) != 0 and b"no communicator" in lib.sx_last_error()
        assert lib.sx_comm_init_local(hs, 2, arr([0, 12]), arr([12, 12]), mode) == 0                 # and a good table still works
        assert lib.sx_exchange_local(hs, 2) == 0
    finally:
        for g in tiles:
            g.close()


@pytest.mark.gpu
@pytest.mark.parametrize("index", range(int(os.environ.get("SCYTHE_FUZZ_RANKS", "4"))))
def test_drawn_configurations_with_ranks_in_separate_processes(tmp_path, index):
    """tests/test_gpu_fuzz.py's random configurations (2-3 tiles, any exchange protocol, random boundary conditions and grid
    sizes) with every tile a RANK in its own process, the exchange inside the library over the stand-in transport."""
    import subprocess
    import scythe_jl_amd as S
    from tests import cases
    case, world, exchange = _drawn_case(4711, index)
    try:       # the same tiles in THIS process first (loopback transport): a configuration the library refuses is not sent to ranks
        cases.HipModel(case, num_tiles=world, exchange=exchange, impl="lib").run.close()
    except S.ScytheHipError as e:
        pytest.skip("refused: %s" % e)
    lib = str(tmp_path / "libfake_rccl.so")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    subprocess.run(["/opt/rocm/bin/hipcc", "-O1", "-shared", "-fPIC", "-o", lib, os.path.join(root, "tests", "fake_rccl.cpp"), "-lrt"],
                   check=True, capture_output=True)
    uid = ("/sxfake_%d_%s" % (os.getpid(), os.urandom(4).hex())).encode().ljust(128, b"\0")
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_fake_worker, args=(r, world, lib, uid, exchange, "fuzz", {"seed": 4711, "index": index}, q)) for r in range(world)]
    for p in procs:
        p.start()
    _join_all(procs, 240)
    res = dict(q.get(timeout=5) for _ in range(world))
    print("\n%s %s cells=%d ranks=%d %s: %s" % (case["grid"]["geometry"], case["eq"], case["grid"]["num_cells"], world, exchange, res))
    assert max(res.values()) < 1e-8, res
