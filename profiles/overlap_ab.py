"""Step time of the one-stream and the two-stream (SX_OVERLAP=1) schedule of the bench workload, block by block from the state
bench.py leaves the device in before its W warm-up steps (after a native-ring run).  Usage: python profiles/overlap_ab.py"""
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


def blocks(run, n, k=5):
    out = []
    for _ in range(n):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(k):
            run.step()
        torch.cuda.synchronize()
        out.append(1e3 * (time.perf_counter() - t0) / k)
    return out


a, b = make(False), make(True)
for r in (a, b):
    for _ in range(3):
        r.step()
torch.cuda.synchronize()
for rep in range(3):
    for name, r in (("serial ", a), ("overlap", b)):
        print(rep, name, " ".join("%.3f" % x for x in blocks(r, 12)), flush=True)
# the same without a host synchronisation between blocks: 60 steps in one go
for name, r in (("serial ", a), ("overlap", b), ("serial ", a), ("overlap", b)):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(60):
        r.step()
    torch.cuda.synchronize()
    print("60 steps", name, "%.4f ms/step" % (1e3 * (time.perf_counter() - t0) / 60), flush=True)
