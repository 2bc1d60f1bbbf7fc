import sys, os, time
sys.path.insert(0, os.getcwd())
import torch, numpy as np
import bench as B, scythe_jl_amd as S
kw, L = B.grid_kwargs("rlz_513x256x64")
gp = S.GridParameters(ring_uniform_L=L, **kw)
mp = S.ModelParameters(ts=B.TS, equation_set="Oneway_ShallowWater_HeightResolvedBL", grid_params=gp, physical_params=dict(B.PAR))
run = S.ModelRun(mp, num_tiles=1, device=torch.device("cuda", 0))
run.set_initial_conditions([B.initial_condition(S.getGridpoints(run.tiles[0]))])
torch.cuda.synchronize()
time.sleep(float(sys.argv[1]) if len(sys.argv) > 1 else 0.0)
n = 80
ev = [torch.cuda.Event(enable_timing=True) for _ in range(n + 1)]
ev[0].record()
for i in range(n):
    run.step()
    ev[i + 1].record()
torch.cuda.synchronize()
ms = [ev[i].elapsed_time(ev[i + 1]) for i in range(n)]
print(" ".join("%.3f" % x for x in ms))
