"""Per-kernel times of the native-ragged-ring equivalent of the bench workload (bench.py's `native_equivalent` run):
    python3 profiles/native_timers.py [steps] [num_cells]   # hipEvent timers of the library (85 cells by default; 171 = the
                                                           # "512-ring" problem on native rings, kmax up to 512: chunked DFT kernels)
    rocprofv3 --kernel-trace --stats ... -- python3 profiles/native_timers.py    # per launch class of the DFT kernels
"""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import scythe_jl_amd as S
import bench as B

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
kw, _ = B.grid_kwargs("rlz_513x256x64")
kw["num_cells"] = int(sys.argv[2]) if len(sys.argv) > 2 else 85
gp = S.GridParameters(ring_uniform_L=0, storage="f64", **kw)
mp = S.ModelParameters(ts=B.TS, equation_set="Oneway_ShallowWater_HeightResolvedBL", grid_params=gp, physical_params=dict(B.PAR))
run = S.ModelRun(mp, num_tiles=1, device="cuda")
run.set_initial_conditions([B.initial_condition(S.getGridpoints(run.tiles[0]))])
for _ in range(3):
    run.step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(steps):
    run.step()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / steps
tile = run.tiles[0]
tile.enable_timers(True)
tile.reset_timers()
for _ in range(steps):
    run.step()
torch.cuda.synchronize()
tm = {k: round(v[0] / steps, 4) for k, v in sorted(tile.timers().items())}
print(json.dumps({"num_cells": kw["num_cells"], "points": int(tile.N), "native_steps_per_s": round(1.0 / dt, 1), "ms_per_step": round(1e3 * dt, 4), "kernels_ms_per_step": tm, "nan": bool(tile.check_nan())}))
run.close()
