"""Per-step device time (events on the tile's stream) of the first 40 steps of a FRESH run of the bench workload, one-stream and
two-stream (SX_OVERLAP=1) schedule, after a pre-heating run - what bench.py's `--warmup 5 --steps 20` samples (steps 6-25)."""
import os, sys, time
sys.path.insert(0, os.getcwd())
import torch
import bench as B, scythe_jl_amd as S

dev = torch.device("cuda", 0)
kw, L = B.grid_kwargs("rlz_513x256x64")
mp = S.ModelParameters(ts=B.TS, equation_set="Oneway_ShallowWater_HeightResolvedBL", grid_params=S.GridParameters(ring_uniform_L=L, **kw), physical_params=dict(B.PAR))


def make(overlap):
    if overlap:
        os.environ["SX_OVERLAP"] = "1"
    r = S.ModelRun(mp, num_tiles=1, device=dev)
    os.environ.pop("SX_OVERLAP", None)
    r.set_initial_conditions([B.initial_condition(S.getGridpoints(r.tiles[0]))])
    return r


pre = make(False)
for _ in range(150):
    pre.step()
torch.cuda.synchronize()
for rep in range(2):
    for name, ov in (("serial ", False), ("overlap", True)):
        r = make(ov)
        for _ in range(60):       # keep the device busy while the new run was being set up
            pre.step()
        n = 40
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(n + 1)]
        t0 = time.perf_counter()
        ev[0].record()
        for i in range(n):
            r.step()
            ev[i + 1].record()
        host = time.perf_counter() - t0
        torch.cuda.synchronize()
        ms = [ev[i].elapsed_time(ev[i + 1]) for i in range(n)]
        print(rep, name, "host enqueue %.1f us/step;" % (1e6 * host / n), "steps 6-25: %.4f ms/step;" % (sum(ms[5:25]) / 20), " ".join("%.3f" % x for x in ms), flush=True)
        r.close()
