"""Kernel time per tile when the bench grid is split into 2 / 4 / 8 radial tiles, all resident on ONE GPU (the exchange is a local
copy): what each rank of a multi-GPU run has to execute between its two all-to-alls.  python profiles/tile_kernel_times.py"""
import sys, time, json
sys.path.insert(0, ".")
import numpy as np, torch
import bench
import scythe_jl_amd as S
kw, L = bench.grid_kwargs("rlz_513x256x64")
gp = S.GridParameters(ring_uniform_L=L, **kw)
mp = S.ModelParameters(ts=bench.TS, equation_set="Oneway_ShallowWater_HeightResolvedBL", grid_params=gp, physical_params=dict(bench.PAR))
for nt in (2, 4, 8):
  for split in ("reference", "cost"):
    run = S.ModelRun(mp, num_tiles=nt, device="cuda", split=split)
    run.set_initial_conditions([bench.initial_condition(S.getGridpoints(g)) for g in run.tiles])
    for _ in range(5):
        run.step()
    torch.cuda.synchronize()
    for g in run.tiles:
        g.enable_timers(True); g.reset_timers()
    t0 = time.perf_counter()
    for _ in range(20):
        run.step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 20
    per = []
    for g in run.tiles:
        tm = g.timers()
        per.append(round(sum(v[0] for v in tm.values()) / 20, 3))
    print(nt, "tiles", split, "split", run.layout.ncells, ": wall %.3f ms/step, per-tile kernel ms:" % (1e3 * dt), per, "max %.3f" % max(per), flush=True)
    run.close()
