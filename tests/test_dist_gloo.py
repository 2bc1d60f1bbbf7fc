"""CPU-only, world_size 2 (gloo): the rank -> rank+1 halo and the in-place all-gather of owned B rows assemble
exactly the patch-level sum the reference builds in its SharedArray (src/semiimplicit.jl:320-329, 272-285)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import scythe_jl_amd as S
from oracle import oracle_c as OC
from tests import cases


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _to_rows(g, btile, K2t, nbt):
    """reference tile layout [S_tile, V] -> device layout [rows][cols], col = (v * Zb + zm) * K2 + blk."""
    out = np.zeros((nbt, g.V * g.b_zDim * g.K2))
    for v in range(g.V):
        b = btile[:, v].reshape(g.b_zDim, K2t, nbt)
        for zm in range(g.b_zDim):
            for blk in range(K2t):
                out[:, (v * g.b_zDim + zm) * g.K2 + blk] = b[zm, blk]
    return out


def _patch_rows(g, shared):
    out = np.zeros((g.b_rDim, g.V * g.b_zDim * g.K2))
    for v in range(g.V):
        out[:, v * g.b_zDim * g.K2:(v + 1) * g.b_zDim * g.K2] = shared[:, v].reshape(g.b_zDim * g.K2, g.b_rDim).T
    return out


def _worker(rank, world, port, case_name, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        case = getattr(cases, case_name)(num_cells=9)
        gp, _ = cases.hip_params(case)
        g = cases.oracle_grid(case)
        n_cols = g.V * g.b_zDim * g.K2
        lay = S.PatchLayout(gp, world, n_cols=n_cols)
        pts = g.gridpoints().reshape(-1, 1 + g.has_l + g.has_z)
        vals = case["ic"](pts)
        # expected: the reference's shared-array protocol over all tiles (oracle)
        shared = np.zeros((g.S_patch(), g.V), order="F")
        mine = None
        p0 = 0
        for t in range(world):
            tl = OC.TileOracle(g, lay.cell0[t], lay.ncells[t])
            b = tl.forward(vals[p0:p0 + tl.N])
            p0 += tl.N
            tl.add_to_shared(b, shared)
            if t == rank:
                mine = _to_rows(g, b, tl.og.K2t, lay.rows(t))
        ex = S.DistExchange(lay, None, "cpu")
        ex.my_rows()[:lay.rows(rank)] = torch.from_numpy(mine)

        def halo_add(recv):
            ex.my_rows()[:3] += recv

        ex.exchange(halo_add=halo_add)
        full = ex.buf.view(-1)[torch.from_numpy(lay.row_offsets())[:, None] + torch.arange(n_cols)[None, :]].numpy()
        err = np.abs(full - _patch_rows(g, shared)).max() / np.abs(shared).max()
        q.put((rank, float(err)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("case_name,world", [("rl_slab", 2), ("rlz_hrbl", 2), ("rl_slab", 3)])
def test_halo_and_gather_reproduce_shared_sum(case_name, world):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, case_name, q)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(180)
        assert p.exitcode == 0
    res = dict(q.get(timeout=5) for _ in range(world))
    assert sorted(res) == list(range(world))
    assert max(res.values()) < 1e-14


# ----------------------------------------------------------------------------- transposed (all-to-all) solve
def _a2a_worker(rank, world, port, case_name, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        case = getattr(cases, case_name)(num_cells=10)
        gp, _ = cases.hip_params(case)
        g = cases.oracle_grid(case)
        K2 = g.K2
        G = g.V * g.b_zDim
        n_cols = G * K2
        lay = S.PatchLayout(gp, world, n_cols=n_cols)
        col_starts = np.array([(G * d // world) * K2 for d in range(world + 1)], dtype=np.int64)   # rule of sx_a2a_configure
        pts = g.gridpoints().reshape(-1, 1 + g.has_l + g.has_z)
        vals = case["ic"](pts)
        shared = np.zeros((g.S_patch(), g.V), order="F")
        mine, p0 = None, 0
        for t in range(world):
            tl = OC.TileOracle(g, lay.cell0[t], lay.ncells[t])
            b = tl.forward(vals[p0:p0 + tl.N])
            p0 += tl.N
            tl.add_to_shared(b, shared)
            if t == rank:
                mine = _to_rows(g, b, tl.og.K2t, lay.rows(t))
        expected = _patch_rows(g, shared)                       # the reference's shared-array sum, [b_rDim][cols]
        ex = S.DistA2AExchange(lay, None, "cpu", col_starts=col_starts)
        al = ex.lay
        c0, c1 = col_starts[rank], col_starts[rank + 1]

        def pack(buf):
            buf[:] = torch.from_numpy(np.concatenate([mine[:, col_starts[d]:col_starts[d + 1]].ravel() for d in range(world)]))

        def solve(inp, out):
            # stand-in for the banded solve: identity on the summed rows (checks the overlap-sum and duplication layout)
            full = np.zeros((g.b_rDim, c1 - c0))
            o = 0
            for t in range(world):
                n = al.rows[t]
                full[lay.cell0[t]:lay.cell0[t] + n] += inp[o:o + n * (c1 - c0)].numpy().reshape(n, c1 - c0)
                o += n * (c1 - c0)
            out[:] = torch.from_numpy(np.concatenate([full[lay.cell0[t]:lay.cell0[t] + al.rows[t]].ravel() for t in range(world)]))

        got = {}

        def unpack(buf):
            rows = np.zeros((al.rows[rank], n_cols))
            o = 0
            for d in range(world):
                w = col_starts[d + 1] - col_starts[d]
                rows[:, col_starts[d]:col_starts[d + 1]] = buf[o:o + al.rows[rank] * w].numpy().reshape(al.rows[rank], w)
                o += al.rows[rank] * w
            got["rows"] = rows

        ex.exchange_and_solve(pack=pack, solve=solve, unpack=unpack)
        want = expected[lay.cell0[rank]:lay.cell0[rank] + al.rows[rank]]
        q.put((rank, float(np.abs(got["rows"] - want).max() / np.abs(expected).max())))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("case_name,world", [("rl_slab", 2), ("rlz_hrbl", 3)])
def test_all_to_all_transpose_layout(case_name, world):
    """Uneven all_to_all_single splits: every tile gets back exactly the patch rows it evaluates, with the rows two
    tiles share summed once (the reference's halo add) and delivered to both."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_a2a_worker, args=(r, world, port, case_name, q)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(180)
        assert p.exitcode == 0
    res = dict(q.get(timeout=5) for _ in range(world))
    assert sorted(res) == list(range(world))
    assert max(res.values()) < 1e-14


# ----------------------------------------------------------------------------- interface-only solve
def _iface_worker(rank, world, port, case_name, num_cells, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle.partitioned_np import InterfaceSolve
        case = getattr(cases, case_name)(num_cells=num_cells)
        gp, _ = cases.hip_params(case)
        g = cases.oracle_grid(case)
        K2, G = g.K2, g.V * g.b_zDim
        n_cols = G * K2
        lay = S.PatchLayout(gp, world, n_cols=n_cols)
        col_starts = np.array([(G * d // world) * K2 for d in range(world + 1)], dtype=np.int64)   # rule of sx_iface_configure
        pts = g.gridpoints().reshape(-1, 1 + g.has_l + g.has_z)
        vals = case["ic"](pts)
        shared = np.zeros((g.S_patch(), g.V), order="F")
        mine, p0 = None, 0
        for t in range(world):
            tl = OC.TileOracle(g, lay.cell0[t], lay.ncells[t])
            b = tl.forward(vals[p0:p0 + tl.N])
            p0 += tl.N
            tl.add_to_shared(b, shared)
            if t == rank:
                mine = _to_rows(g, b, tl.og.K2t, lay.rows(t))
                mytile = tl
        # expected: the reference's protocol - shared sum, then the one-patch solve (src/semiimplicit.jl:272-285)
        expected = _patch_rows(g, np.asfortranarray(mytile.spline_solve(shared)))
        # boundary-condition class of every column, one numpy interface solver per class
        cls_of, keys = np.zeros(n_cols, dtype=np.int64), []
        for v, name in enumerate(g.names):
            for zm in range(g.b_zDim):
                for blk in range(K2):
                    key = (g.BCL_k0[name] if blk == 0 else g.BCL[name], g.BCR[name])
                    if key not in keys:
                        keys.append(key)
                    cls_of[(v * g.b_zDim + zm) * K2 + blk] = keys.index(key)
        solvers = {i: InterfaceSolve(g.spline(*key), lay.cell0, lay.ncells) for i, key in enumerate(keys)}
        ex = S.DistA2AExchange(lay, None, "cpu", col_starts=col_starts, kind="iface")
        R = S.driver.IFACE_ROWS
        c0, c1 = col_starts[rank], col_starts[rank + 1]
        yprime = {}

        def local(buf):
            send = np.zeros((R, n_cols))
            for k, ps in solvers.items():
                cols = np.flatnonzero(cls_of == k)
                yprime[k], send[:, cols] = ps.local(rank, mine[:, cols])
            buf[:] = torch.from_numpy(np.concatenate([send[:, col_starts[d]:col_starts[d + 1]].ravel() for d in range(world)]))

        def reduce_(inp, out):
            w = c1 - c0
            x = inp.numpy()[:world * R * w].reshape(world, R, w)     # a rank without columns holds a 1-element placeholder
            y = np.zeros_like(x)
            for k, ps in solvers.items():
                cols = np.flatnonzero(cls_of[c0:c1] == k)
                if len(cols):
                    y[:, :, cols] = ps.reduce(x[:, :, cols])
            out[:y.size] = torch.from_numpy(y.ravel())

        got = {}

        def apply_(buf):
            recv = np.zeros((R, n_cols))
            o = 0
            for d in range(world):
                w = col_starts[d + 1] - col_starts[d]
                recv[:, col_starts[d]:col_starts[d + 1]] = buf[o:o + R * w].numpy().reshape(R, w)
                o += R * w
            rows = np.zeros((lay.rows(rank), n_cols))
            for k, ps in solvers.items():
                cols = np.flatnonzero(cls_of == k)
                rows[:, cols] = ps.apply(rank, yprime[k], recv[:, cols])
            got["rows"] = rows

        ex.exchange_and_solve(pack=local, solve=reduce_, unpack=apply_)
        want = expected[lay.cell0[rank]:lay.cell0[rank] + lay.rows(rank)]
        q.put((rank, float(np.abs(got["rows"] - want).max() / np.abs(expected).max())))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("case_name,num_cells,world", [("rl_slab", 14, 2), ("rlz_hrbl", 36, 3), ("kat_r", 40, 3)])
def test_interface_only_solve_over_all_to_all(case_name, num_cells, world):
    """The interface-only patch solve across ranks (gloo): every rank solves its own tile's rows, the uneven
    all_to_all_single splits carry 10 rows per tile to the owner of each column range and back, and every tile ends up
    with exactly the A rows the reference's shared sum + one-patch solve gives it (the device stages of sx_iface.hip are
    stood in for by their numpy statement, oracle/partitioned_np.py; PERIODIC wrap rows with kat_r)."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_iface_worker, args=(r, world, port, case_name, num_cells, q)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(300)
        assert p.exitcode == 0
    res = dict(q.get(timeout=5) for _ in range(world))
    assert sorted(res) == list(range(world))
    assert max(res.values()) < 1e-11
