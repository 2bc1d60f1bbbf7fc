"""Step driver: the master / worker protocol of src/semiimplicit.jl:126-332 with one radial tile per GPU.

The reference runs one Julia worker per tile, a unidirectional RemoteChannel chain for the 3-coefficient halo
(src/semiimplicit.jl:203-219, 320-329), a SharedArray for the patch-level sum (:229, :272-282) and a redundant
patch solve on every worker (:285).  Here a tile is a libscythe_hip handle and the exchange is done on device
buffers: halo rows by point-to-point send/recv, owned rows by an in-place all-gather, both over
torch.distributed (backend "nccl" = RCCL over xGMI on the GPU box, "gloo" in the CPU tests).
"""
import os
import time

import numpy as np

from .model import (Grid, GridParameters, ModelParameters, calcTileSizes, checkCFL, comm_unique_id, getGridpoints)


class PatchLayout:
    """Which patch rows (radial nodes) each tile computes, owns and sends (calcPatchMap / calcHaloMap,
    src/semiimplicit.jl:79-86), and where they sit in the all-gather buffer."""

    def __init__(self, patch: GridParameters, num_tiles: int, n_cols: int = 0, split="reference"):
        """split = "reference": calcTileSizes (cells split evenly, or gridpoints balanced on native rings).
        split = "cost" (uniform ring tables only): cells whose rings still have a growing wavenumber truncation
        (ring index < L/2 - 1) run the ring-wise kernels, which cost about 1.6x the node-space path per ring, so the inner
        tiles get proportionally fewer cells - the same idea as the reference's gridpoint balancing (src/semiimplicit.jl:144)."""
        ts = calcTileSizes(patch, num_tiles)
        if split == "cost" and num_tiles > 1 and patch.ring_uniform_L > 0 and "L" in patch.geometry:
            ts = _cost_balanced_tiles(patch, num_tiles, ts)
        elif split not in ("reference", "cost"):
            raise ValueError("split must be 'reference' or 'cost'")
        self.num_tiles = num_tiles
        self.ncells = [int(ts[2, t]) for t in range(num_tiles)]
        self.cell0 = [int(ts[3, t]) - 1 for t in range(num_tiles)]
        self.tile_sizes = ts
        self.b_rDim = patch.num_cells + 3
        self.max_rows = max(self.ncells) + 3
        self.n_cols = n_cols

    def rows(self, t):
        return self.ncells[t] + 3

    def owned_rows(self, t):
        return self.ncells[t] + (3 if t == self.num_tiles - 1 else 0)

    def row_offsets(self, n_cols=None):
        """Element offset of patch row m inside the [num_tiles][max_rows][n_cols] gather buffer."""
        n_cols = n_cols or self.n_cols
        off = np.zeros(self.b_rDim, dtype=np.int64)
        for t in range(self.num_tiles):
            for j in range(self.owned_rows(t)):
                off[self.cell0[t] + j] = (t * self.max_rows + j) * n_cols
        return off


def _cost_balanced_tiles(patch, n, ts_ref, inner_weight=1.6):
    nc = patch.num_cells
    kcap = patch.ring_uniform_L // 2 - 1
    w = np.array([inner_weight if 3 * c < kcap else 1.0 for c in range(nc)])
    cum = np.concatenate([[0.0], np.cumsum(w)])
    bounds = [0]
    for t in range(1, n):
        c = int(np.searchsorted(cum, cum[-1] * t / n))
        c = max(c, bounds[-1] + 3)                    # at least 3 cells per tile (calcTileSizes' rule)
        c = min(c, nc - 3 * (n - t))
        bounds.append(c)
    bounds.append(nc)
    DX = (patch.xmax - patch.xmin) / nc
    pts_per_cell = 3 * patch.ring_uniform_L * max(patch.zDim, 1) if "Z" in patch.geometry else 3 * patch.ring_uniform_L
    ts = np.zeros_like(ts_ref)
    for t in range(n):
        c0, c1 = bounds[t], bounds[t + 1]
        ts[:, t] = (patch.xmin + c0 * DX, patch.xmin + c1 * DX, c1 - c0, c0 + 1, (c1 - c0) * pts_per_cell)
    return ts


def _torch():
    import torch
    return torch


class LocalExchange:
    """All tiles live in this process on one device (used by single-GPU tests of the tile protocol)."""

    def __init__(self, layout: PatchLayout, tiles, device):
        torch = _torch()
        self.layout, self.tiles = layout, tiles
        C = tiles[0].n_cols
        self.buf = torch.zeros((layout.num_tiles, layout.max_rows, C), dtype=torch.float64, device=device)
        ro = layout.row_offsets(C)
        for t, g in enumerate(tiles):
            g.bind_tile_b(self.buf[t].data_ptr())
            g.bind_patch_b(self.buf.data_ptr(), ro)

    def exchange(self):
        lay = self.layout
        for t in range(1, lay.num_tiles):
            n = lay.ncells[t - 1]
            self.tiles[t].halo_add(self.buf[t - 1, n:n + 3].data_ptr())


class DistExchange:
    """One tile per rank. Halo rows travel rank -> rank + 1, owned rows are all-gathered in place."""

    def __init__(self, layout: PatchLayout, tile: Grid, device, group=None):
        torch = _torch()
        import torch.distributed as dist
        self.dist, self.group = dist, group
        self.rank, self.world = dist.get_rank(group), dist.get_world_size(group)
        assert self.world == layout.num_tiles
        self.layout, self.tile = layout, tile
        C = tile.n_cols if tile is not None else layout.n_cols
        self.buf = torch.zeros((self.world, layout.max_rows, C), dtype=torch.float64, device=device)
        self.halo = torch.zeros((3, C), dtype=torch.float64, device=device)
        if tile is not None:
            tile.bind_tile_b(self.buf[self.rank].data_ptr())
            tile.bind_patch_b(self.buf.data_ptr(), layout.row_offsets(C))
        self.stage_host = (dist.get_backend(group) == "gloo" and self.buf.is_cuda)   # one-GPU rehearsal only

    def my_rows(self):
        return self.buf[self.rank]

    def exchange(self, halo_add=None):
        """halo_add(recv_tensor) adds the received rows into rows 0..2 of this tile (device kernel on the GPU path)."""
        dist, r, W = self.dist, self.rank, self.world
        if W > 1:
            ops = []
            n = self.layout.ncells[r]
            send = self.buf[r, n:n + 3]
            recv = self.halo
            if self.stage_host:
                send, recv = send.cpu(), self.halo.cpu()
            if r < W - 1:
                ops.append(dist.P2POp(dist.isend, send, r + 1, self.group))
            if r > 0:
                ops.append(dist.P2POp(dist.irecv, recv, r - 1, self.group))
            for w in dist.batch_isend_irecv(ops):
                w.wait()
            if r > 0:
                if self.stage_host:
                    self.halo.copy_(recv)
                if halo_add is not None:
                    halo_add(self.halo)
                else:
                    self.tile.halo_add(self.halo.data_ptr())
            if self.stage_host:
                full = self.buf.cpu()
                dist.all_gather_into_tensor(full.view(-1), full[r].reshape(-1).clone(), group=self.group)
                self.buf.copy_(full)
            else:
                dist.all_gather_into_tensor(self.buf.view(-1), self.buf[r].reshape(-1), group=self.group)


class A2ALayout:
    """Buffer geometry of the transposed solve for tile `me` (see include/scythe_hip.h, sx_a2a_*)."""

    def __init__(self, layout: PatchLayout, col_starts, me, rows=None):
        """rows = None: every tile moves all its ncells + 3 rows (transposed solve); rows = 10: the interface-only solve."""
        self.n = layout.num_tiles
        self.rows = [layout.rows(t) if rows is None else rows for t in range(self.n)]
        self.cw = [int(col_starts[d + 1] - col_starts[d]) for d in range(self.n)]
        self.me = me
        # tile side: [dest d][row][cw[d]];  owner side: [tile t][row][cw[me]]
        self.tile_split = [self.rows[me] * self.cw[d] for d in range(self.n)]
        self.owner_split = [self.rows[t] * self.cw[me] for t in range(self.n)]
        self.tile_elems = sum(self.tile_split)
        self.owner_elems = sum(self.owner_split)


IFACE_ROWS = 10     # rows per tile that travel in the interface-only solve (6 edge values + 4 foreign rows, sx_iface.hip)


def _kernels_of(tile, kind):
    """(first, middle, last) device stages around the two all-to-alls: pack / solve / unpack of the transposed solve or
    local solve / reduced system / correction of the interface-only solve."""
    if kind == "iface":
        return tile.iface_local, tile.iface_reduce, tile.iface_apply
    return tile.a2a_pack_b, tile.a2a_solve, tile.a2a_unpack_a


class LocalA2AExchange:
    """Transposed solve (kind "a2a") or interface-only solve (kind "iface") with all tiles in this process (single-GPU test
    of the kernels on either side of the two all-to-alls)."""

    def __init__(self, layout: PatchLayout, tiles, device, kind="a2a"):
        torch = _torch()
        self.tiles, self.kind = tiles, kind
        self.lay = []
        for t, g in enumerate(tiles):
            if kind == "iface":
                self.lay.append(A2ALayout(layout, g.iface_configure(layout.cell0, layout.ncells, t), t, rows=IFACE_ROWS))
            else:
                self.lay.append(A2ALayout(layout, g.a2a_configure(layout.cell0, layout.ncells, t), t))
        z = lambda n: torch.zeros(max(n, 1), dtype=torch.float64, device=device)
        self.tile_buf = [z(l.tile_elems) for l in self.lay]      # pack output / unpack input
        self.own_in = [z(l.owner_elems) for l in self.lay]
        self.own_out = [z(l.owner_elems) for l in self.lay]

    def _all_to_all(self, src, src_splits, dst, dst_splits):
        n = len(src)
        for s in range(n):
            so = 0
            for d in range(n):
                cnt = src_splits[s][d]
                do = sum(dst_splits[d][:s])
                dst[d][do:do + cnt] = src[s][so:so + cnt]
                so += cnt

    def exchange_and_solve(self):
        for g, b in zip(self.tiles, self.tile_buf):
            _kernels_of(g, self.kind)[0](b.data_ptr())
        self._all_to_all(self.tile_buf, [l.tile_split for l in self.lay], self.own_in, [l.owner_split for l in self.lay])
        for g, i, o in zip(self.tiles, self.own_in, self.own_out):
            _kernels_of(g, self.kind)[1](i.data_ptr(), o.data_ptr())
        self._all_to_all(self.own_out, [l.owner_split for l in self.lay], self.tile_buf, [l.tile_split for l in self.lay])
        for g, b in zip(self.tiles, self.tile_buf):
            _kernels_of(g, self.kind)[2](b.data_ptr())


class DistA2AExchange:
    """Transposed solve (kind "a2a") or interface-only solve (kind "iface"), one tile per rank: two all_to_all_single calls
    per step (RCCL over xGMI on the GPU box)."""

    def __init__(self, layout: PatchLayout, tile, device, group=None, col_starts=None, kind="a2a"):
        torch = _torch()
        import torch.distributed as dist
        self.dist, self.group, self.tile, self.kind = dist, group, tile, kind
        self.rank, self.world = dist.get_rank(group), dist.get_world_size(group)
        assert self.world == layout.num_tiles
        if tile is not None:
            col_starts = (tile.iface_configure if kind == "iface" else tile.a2a_configure)(layout.cell0, layout.ncells, self.rank)
        self.lay = A2ALayout(layout, col_starts, self.rank, rows=IFACE_ROWS if kind == "iface" else None)
        z = lambda n: torch.zeros(max(n, 1), dtype=torch.float64, device=device)
        self.tile_buf, self.tile_buf2 = z(self.lay.tile_elems), z(self.lay.tile_elems)
        self.own_in, self.own_out = z(self.lay.owner_elems), z(self.lay.owner_elems)
        # gloo cannot move device tensors through all_to_all: stage through the host (rehearsal on one GPU only;
        # the production backend is "nccl" = RCCL, which takes the device buffers directly)
        self.stage_host = (dist.get_backend(group) == "gloo" and self.tile_buf.is_cuda)

    def _a2a(self, out, inp, out_split, in_split):
        out, inp = out[:sum(out_split)], inp[:sum(in_split)]        # a rank without columns keeps a 1-element placeholder buffer
        if self.stage_host:
            o = out.cpu()
            self.dist.all_to_all_single(o, inp.cpu(), out_split, in_split, group=self.group)
            out.copy_(o)
        else:
            self.dist.all_to_all_single(out, inp, out_split, in_split, group=self.group)

    def exchange_and_solve(self, pack=None, solve=None, unpack=None):
        """pack / solve / unpack default to the tile's device kernels; the CPU tests pass numpy stand-ins."""
        lay = self.lay
        k = _kernels_of(self.tile, self.kind) if self.tile is not None else (None, None, None)
        (pack or (lambda b: k[0](b.data_ptr())))(self.tile_buf)
        self._a2a(self.own_in, self.tile_buf, lay.owner_split, lay.tile_split)
        (solve or (lambda i, o: k[1](i.data_ptr(), o.data_ptr())))(self.own_in, self.own_out)
        self._a2a(self.tile_buf2, self.own_out, lay.tile_split, lay.owner_split)
        (unpack or (lambda b: k[2](b.data_ptr())))(self.tile_buf2)


class LibExchange:
    """One tile per rank, exchange done INSIDE libscythe_hip.so with RCCL on the tile's stream (sx_comm_init / sx_exchange):
    what a Julia host would use.  Only the 128-byte ncclUniqueId travels through the host-side launcher - here
    torch.distributed's store (any backend), in the reference's world the master's RemoteChannels."""

    def __init__(self, layout: PatchLayout, tile: Grid, mode, group=None, unique_id=None):
        self.tile, self.mode = tile, mode
        rank, world = 0, 1
        if unique_id is None:
            import torch.distributed as dist
            rank, world = dist.get_rank(group), dist.get_world_size(group)
            assert world == layout.num_tiles
            # Everything that can fail on one rank alone happens BEFORE the first collective, and the ranks agree on the outcome:
            # a rank that raised here while the others sat in the broadcast (or in ncclCommInitRank) would hang the job.
            err, uid = "", None
            try:
                tile.comm_prepare(layout.cell0, layout.ncells, rank, mode)      # librccl, tile table, buffers: not collective
                uid = comm_unique_id()                                          # ncclGetUniqueId: not collective either
            except Exception as e:
                err = str(e)
            oks = [None] * world
            dist.all_gather_object(oks, err, group=group)
            if any(oks):
                raise RuntimeError("in-library exchange unavailable on rank(s) %s: %s"
                                   % ([r for r, e in enumerate(oks) if e], next(e for e in oks if e)))
            box = [uid if rank == 0 else None]
            dist.broadcast_object_list(box, src=0, group=group)
            unique_id = box[0]
        else:
            rank = layout.cell0.index(tile.cell0) if layout.num_tiles > 1 else 0
            world = layout.num_tiles
        assert world == layout.num_tiles
        self.rank, self.world = rank, world
        tile.comm_init(layout.cell0, layout.ncells, rank, mode, unique_id)

    def exchange_and_solve(self):
        self.tile.exchange()


class LocalLibExchange:
    """All tiles in this process on one GPU, exchange done inside libscythe_hip.so with the loopback transport
    (sx_comm_init_local / sx_exchange_local): the RCCL path's buffer geometry and offsets with copies instead of sends."""

    def __init__(self, layout: PatchLayout, tiles, mode):
        import ctypes as C
        from . import _lib as L
        self._L, self.n = L, len(tiles)
        self.hs = (C.c_void_p * self.n)(*[g._h for g in tiles])
        c0 = (C.c_int32 * self.n)(*layout.cell0)
        nc = (C.c_int32 * self.n)(*layout.ncells)
        from .model import EXCHANGE_MODES
        L.check(L.load().sx_comm_init_local(self.hs, self.n, c0, nc, EXCHANGE_MODES[mode]))

    def exchange_and_solve(self):
        self._L.check(self._L.load().sx_exchange_local(self.hs, self.n))


class ModelRun:
    """initialize_model + run_model state for one process (src/semiimplicit.jl:126-256)."""

    def __init__(self, model: ModelParameters, num_tiles=1, rank=None, device=None, use_dist=False, exchange="a2a",
                 split="reference", impl="torch", unique_id=None):
        """exchange: "a2a" = transposed solve over all-to-all, "iface" = interface-only solve (tile-local solves, all-to-all of
        10 rows per tile around a small reduced system: least traffic, shortest recurrence), "gather" = the reference's
        protocol (halo chain + gather of owned rows + redundant patch solve on every tile), "auto" = "iface" where every tile
        has at least 9 cells, else "a2a".
        impl (use_dist only): "lib" = RCCL calls inside libscythe_hip.so on the tile's stream (sx_exchange; also valid with
        ONE tile, where every send is a send to self - the one-GPU self-test of that code path), "torch" =
        torch.distributed collectives on device tensors (also what the gloo rehearsals use)."""
        self.model = model
        patch = model.grid_params
        self.patch = patch
        self.num_tiles = num_tiles
        self.layout = PatchLayout(patch, num_tiles, split=split)
        if exchange == "auto":      # the interface-only solve where every tile is large enough for it (6 free coefficients: >= 9 cells
            exchange = "iface" if num_tiles > 1 and min(self.layout.ncells) >= 9 else "a2a"          # covers every boundary condition)
        self.use_dist = use_dist
        if use_dist:
            t = rank
            self.tiles = [Grid(patch, model, self.layout.cell0[t], self.layout.ncells[t], t + 2)]
            self.tile_ids = [t]
        else:
            self.tiles = [Grid(patch, model, self.layout.cell0[t], self.layout.ncells[t], t + 2)
                          for t in range(num_tiles)]
            self.tile_ids = list(range(num_tiles))
        self.exchange = None
        self.exchange_kind = exchange if num_tiles > 1 else "none"
        self.impl = impl if use_dist else "local"
        self._bind_streams(device)
        if use_dist and impl == "lib":
            self.exchange_kind = exchange
            self.exchange = LibExchange(self.layout, self.tiles[0], exchange, unique_id=unique_id)
        elif not use_dist and impl == "lib" and num_tiles > 1:
            self.impl = "lib"
            self.exchange = LocalLibExchange(self.layout, self.tiles, exchange)
        elif num_tiles > 1:
            if exchange in ("a2a", "iface"):
                self.exchange = (DistA2AExchange(self.layout, self.tiles[0], device, kind=exchange) if use_dist
                                 else LocalA2AExchange(self.layout, self.tiles, device, kind=exchange))
            elif exchange == "gather":
                self.exchange = (DistExchange(self.layout, self.tiles[0], device) if use_dist
                                 else LocalExchange(self.layout, self.tiles, device))
            else:
                raise ValueError("exchange must be 'a2a', 'iface' or 'gather'")
        self.t = 0

    def _bind_streams(self, device):
        """Kernels of a tile run on ITS stream; torch's collectives order themselves against torch's CURRENT stream.  Hand
        that stream to the library so that both see one queue (with the default stream this is stream 0, as before)."""
        self._stream = None
        try:
            torch = _torch()
            if device is not None and torch.cuda.is_available():
                self._stream = torch.cuda.current_stream(device).cuda_stream
                for g in self.tiles:
                    g.set_stream(self._stream)
        except ImportError:
            pass

    def _check_stream(self):
        # every implementation but the in-library one issues torch operations (collectives, or the copies of the Local*
        # exchanges) on torch's CURRENT stream: the tiles' kernels must follow it
        if self._stream is not None and self.impl != "lib":
            cur = _torch().cuda.current_stream().cuda_stream
            if cur != self._stream:          # the caller entered a torch.cuda.stream(...) context: follow it
                self._stream = cur
                for g in self.tiles:
                    g.set_stream(cur)

    def tile_points(self, t):
        return self.tiles[self.tile_ids.index(t)].N

    def set_initial_conditions(self, values_by_tile):
        """read_physical_grid + spectralTransform!(patch) + first splineTransform! (:134-136, :233-237)."""
        for g, v in zip(self.tiles, values_by_tile):
            g.set_physical_values(v)
            g.spectralTransform_()
        self._exchange_and_solve()

    def _exchange_and_solve(self):
        self._check_stream()
        if self.impl == "lib":
            self.exchange.exchange_and_solve()
            return
        if self.exchange_kind in ("a2a", "iface"):
            self.exchange.exchange_and_solve()
            return
        if self.exchange is not None:
            self.exchange.exchange()
        for g in self.tiles:
            g.splineTransform_()

    def step(self):
        """One pass of model_loop's body (src/semiimplicit.jl:268-297) without the output branch."""
        self.t += 1
        if self.num_tiles == 1 and self.exchange is None:
            self._check_stream()
            self.tiles[0].step(self.t)          # sx_step = sx_advance + sx_spline_transform (one hipGraph launch with SX_GRAPH=1)
            return
        for g in self.tiles:
            g.advance(self.t)
        self._exchange_and_solve()

    def physical(self):
        """tileTransform! on every local tile; returns the concatenated physical array."""
        out = []
        for g in self.tiles:
            g.tileTransform_()
            out.append(g.physical)
        return np.concatenate(out, axis=0)

    def patch_spectral(self):
        """mtile.patchSpectral as the master pulls it from a worker for output (src/semiimplicit.jl:288-293): the patch's A
        coefficients [s_patch, V].  With the reference's protocol every tile holds the whole patch; with the transposed solve a
        tile only holds the rows it evaluates, so the patch array is assembled from every tile's OWNED rows (output cadence
        only; one process per GPU: gathered to rank 0, other ranks return None)."""
        if self.exchange_kind not in ("a2a", "iface"):
            return self.tiles[0].patchSpectral
        nb = self.layout.b_rDim
        parts = []
        for t, g in zip(self.tile_ids, self.tiles):
            # reference layout of one variable's column: s = (z-mode * n_blocks + block) * b_rDim + node, the node FASTEST
            a = g.patchSpectral.reshape(nb, -1, g.V, order="F")             # [node, z-mode x block, var]
            c0 = self.layout.cell0[t]
            parts.append((c0, np.ascontiguousarray(a[c0:c0 + self.layout.owned_rows(t)])))
        if self.use_dist:
            import torch.distributed as dist
            box = [None] * dist.get_world_size() if dist.get_rank() == 0 else None
            dist.gather_object(parts, box, dst=0)
            if dist.get_rank() != 0:
                return None
            parts = [p for ps in box for p in ps]
        out = np.zeros((nb, parts[0][1].shape[1], parts[0][1].shape[2]))
        for c0, rows in parts:
            out[c0:c0 + rows.shape[0]] = rows
        return out.reshape(-1, out.shape[2], order="F")

    def synchronize(self):
        for g in self.tiles:
            g.synchronize()

    def save_checkpoint(self, path):
        """Restart file of the local tiles after step self.t (one .npz; with one process per GPU every rank writes its
        own path).  The reference restarts from a physical_out CSV only, i.e. with an Euler / AB2 start-up; this keeps
        the AB3 history so that the continued run is bit-identical to an uninterrupted one."""
        np.savez(path, t=self.t, tile_ids=np.array(self.tile_ids), **{"tile%d" % i: g.get_state() for i, g in zip(self.tile_ids, self.tiles)})

    def load_checkpoint(self, path):
        with np.load(path) as z:
            if list(z["tile_ids"]) != list(self.tile_ids):
                raise ValueError("checkpoint holds tiles %s, this run holds %s" % (list(z["tile_ids"]), self.tile_ids))
            for i, g in zip(self.tile_ids, self.tiles):
                g.set_state(z["tile%d" % i])
            self.t = int(z["t"])

    def close(self):
        for g in self.tiles:
            g.close()


def integrate_model(model: ModelParameters, num_tiles=1, verbose=False):
    """integrate_model(model) (src/Scythe.jl:37-62) on one GPU: initial conditions from CSV, time loop with
    output every output_interval, final output. Returns True like run_model."""
    from .io import read_physical_grid, write_output
    if not os.path.isdir(model.output_dir):
        os.makedirs(model.output_dir, exist_ok=True)
    log = open(os.path.join(model.output_dir, "scythe_out.log"), "w")
    device = None
    if num_tiles > 1:
        device = "cuda"
    run = ModelRun(model, num_tiles=num_tiles, device=device, exchange="auto")
    print("Initializing with %d workers and tiles" % num_tiles, file=log)
    vals = read_physical_grid(model.initial_conditions, model.grid_params, run)
    run.set_initial_conditions(vals)
    num_ts = int(round(model.integration_time / model.ts))
    output_int = int(round(model.output_interval / model.ts))
    print("Integrating %s sec increments for %d timesteps" % (model.ts, num_ts), file=log)
    write_output(run, model, 0.0)
    t0 = time.time()
    for t in range(1, num_ts + 1):
        if verbose:
            print("ts: %s" % (t * model.ts), file=log)
        run.step()
        if output_int > 0 and t % output_int == 0 and t != num_ts:
            write_output(run, model, t * model.ts)
            for g in run.tiles:
                checkCFL(g)
    print("%.6f seconds" % (time.time() - t0), file=log)
    write_output(run, model, model.integration_time)
    for g in run.tiles:
        checkCFL(g)
    print("Model complete!", file=log)
    log.close()
    run.close()
    return True
