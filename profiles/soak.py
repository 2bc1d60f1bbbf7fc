"""Sustained run of the bench workload: N steps in blocks of 2000, steps/s and max |value| per variable of every block, NaN check.
    python3 profiles/soak.py [steps] [ts]      -> profiles/r03/soak_20000_steps.txt was written from its output"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench as B, scythe_jl_amd as S

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
ts = float(sys.argv[2]) if len(sys.argv) > 2 else B.TS
kw, L = B.grid_kwargs("rlz_513x256x64")
gp = S.GridParameters(ring_uniform_L=L, **kw)
mp = S.ModelParameters(ts=ts, equation_set="Oneway_ShallowWater_HeightResolvedBL", grid_params=gp, physical_params=dict(B.PAR))
run = S.ModelRun(mp, num_tiles=1, device=torch.device("cuda", 0))
run.set_initial_conditions([B.initial_condition(S.getGridpoints(run.tiles[0]))])
tile = run.tiles[0]
print("ts = %g s" % ts)
print("step  steps/s  nan  max|h| max|ug| max|vg| max|ub| max|vb| max|wb|", flush=True)
done = 0
while done < steps:
    n = min(2000, steps - done)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        run.step()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    done += n
    print("%6d %7.1f %5s  %s" % (done, n / dt, bool(tile.check_nan()), " ".join("%.4g" % x for x in tile.max_abs())), flush=True)
run.close()
