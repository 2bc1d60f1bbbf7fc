import os, sys, json
sys.path.insert(0, "/root/repo")
import numpy as np, torch
import scythe_jl_amd as S
import bench as B
def run(env, steps):
    for k, v in env.items(): os.environ[k] = v
    kw, _ = B.grid_kwargs("rlz_513x256x64"); kw["num_cells"] = 85
    gp = S.GridParameters(ring_uniform_L=0, storage="f64", **kw)
    mp = S.ModelParameters(ts=B.TS, equation_set="Oneway_ShallowWater_HeightResolvedBL", grid_params=gp, physical_params=dict(B.PAR))
    r = S.ModelRun(mp, num_tiles=1, device="cuda")
    r.set_initial_conditions([B.initial_condition(S.getGridpoints(r.tiles[0]))])
    for _ in range(steps): r.step()
    torch.cuda.synchronize()
    f = np.array(r.tiles[0].var_np1); nan = bool(r.tiles[0].check_nan()); r.close()
    for k in env: del os.environ[k]
    return f, nan
steps = int(sys.argv[1])
a, na = run({}, steps)
b, nb = run({"SX_DFT_MERGE": "0"}, steps)
c, nc = run({"SX_DFT_HALFWG": "0"}, steps)
out = {"steps": steps, "nan": [na, nb, nc]}
for v in range(a.shape[1]):
    s = max(np.abs(b[:, v]).max(), 1e-300)
    out["var%d" % v] = [float(np.abs(a[:, v] - b[:, v]).max() / s), float(np.abs(c[:, v] - b[:, v]).max() / s)]
print(json.dumps(out))
