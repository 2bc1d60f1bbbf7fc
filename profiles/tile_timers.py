"""What ONE rank of an N-way radial split executes per step, measured on one GPU: the N tiles of the bench workload live on
the same device and exchange through the library's loopback transport (sx_exchange_local), so every kernel of a tile runs
on the whole GPU exactly as it would on that rank's own GPU - only the wire time is missing.
    python3 profiles/tile_timers.py N [a2a|gather|iface] [cost|reference] [notimers]
prints, per tile, the hipEvent time of every kernel per step and their sum (each figure carries the ~5 us an event pair costs on
the stream).  With `notimers` no events are recorded: run it under `rocprofv3 --kernel-trace --stats` for the kernels' own durations."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import scythe_jl_amd as S
import bench as B

n = int(sys.argv[1]) if len(sys.argv) > 1 else 8
kind = sys.argv[2] if len(sys.argv) > 2 else "a2a"
split = sys.argv[3] if len(sys.argv) > 3 else "cost"
timers = not (len(sys.argv) > 4 and sys.argv[4] == "notimers")
steps = 20
kw, L = B.grid_kwargs("rlz_513x256x64")
gp = S.GridParameters(ring_uniform_L=L, **kw)
mp = S.ModelParameters(ts=B.TS, equation_set="Oneway_ShallowWater_HeightResolvedBL", grid_params=gp, physical_params=dict(B.PAR))
run = S.ModelRun(mp, num_tiles=n, device="cuda", exchange=kind, impl="lib", split=split)
run.set_initial_conditions([B.initial_condition(S.getGridpoints(g)) for g in run.tiles])
for _ in range(3):
    run.step()
torch.cuda.synchronize()
for g in run.tiles:
    g.enable_timers(timers)
    g.reset_timers()
for _ in range(steps):
    run.step()
torch.cuda.synchronize()
out = []
for i, g in enumerate(run.tiles):
    tm = {k: round(v[0] / steps, 4) for k, v in sorted(g.timers().items())}
    out.append({"tile": i, "cells": run.layout.ncells[i], "sum_ms": round(sum(tm.values()), 4), "kernels_ms_per_step": tm})
    print(json.dumps(out[-1]))
print(json.dumps({"tiles": n, "exchange": kind, "max_tile_ms": max(o["sum_ms"] for o in out), "nan": any(bool(g.check_nan()) for g in run.tiles)}))
run.close()
