"""-m gpu: two ranks on the one GPU, torch.distributed over gloo (device buffers staged through the host): the
whole multi-rank step path of bench.py - tile per rank, pack / all-to-all / solve / all-to-all / unpack, or halo +
all-gather - against the one-patch oracle. RCCL itself cannot be exercised on a one-GPU box."""
import os
import socket

import numpy as np
import pytest
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, exchange, q):
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import scythe_jl_amd as S
        from tests import cases
        torch.cuda.set_device(0)
        case = cases.rlz_hrbl(num_cells=8, zDim=12, ring_L=32)
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


@pytest.mark.parametrize("exchange", ["a2a", "gather"])
def test_two_ranks_one_gpu(exchange):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, exchange, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(300)
        assert p.exitcode == 0
    res = dict(q.get(timeout=5) for _ in range(2))
    assert max(res.values()) < 1e-10
