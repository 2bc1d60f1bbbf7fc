"""Step time of BASELINE.json's parity configurations on one MI355X (informational; bench.py measures config 4).
Run: python profiles/time_configs.py   (on a GPU box)"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from tests import cases  # noqa: E402


def run(name, case, steps=50, warmup=10):
    hip = cases.HipModel(case)
    g = hip.run.tiles[0]
    for _ in range(warmup):
        hip.step()
    torch.cuda.synchronize()
    g.enable_timers(True)
    g.reset_timers()
    t0 = time.perf_counter()
    for _ in range(steps):
        hip.step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    tm = {k: round(v[0] / steps, 4) for k, v in g.timers().items()}
    g.tileTransform_()
    out = {"config": name, "points": g.N, "vars": g.V, "ms_per_step": round(1e3 * dt, 4), "steps_per_s": round(1 / dt, 1),
           "nan": g.check_nan(), "kernels_ms": tm}
    print(json.dumps(out), flush=True)
    hip.run.close()


def config4_native_equivalent():
    """SURVEY.md 8(d) "native-equivalent shape": 85 cells -> 255 native ragged rings (4 + 4 ri points, kmax = ri),
    131,580 horizontal points x 64 levels, same equation set / parameters / initial condition as bench.py."""
    import bench
    kw, _ = bench.grid_kwargs("rlz_513x256x64")
    kw["num_cells"] = 85
    return dict(name="config4_native", grid=kw, eq="Oneway_ShallowWater_HeightResolvedBL", ts=bench.TS, par=dict(bench.PAR),
                ic=bench.initial_condition)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "native4":
        run("config 4 native-equivalent: RLZ 85 cells, native ragged rings x 64 levels (matrix-core DFT)",
            config4_native_equivalent(), steps=10, warmup=3)
        sys.exit(0)
    c2 = cases.config2_literal()
    run("config 2: RL cha_bell2024 Oneway slab, 100 cells, native ragged rings (matrix-core DFT)", c2)
    c2u = cases.config2_literal()
    c2u["grid"]["ring_L"] = 256
    run("config 2 on uniform 256-point rings (FFT path)", c2u)
    run("config 3: RZ 513 x 128, LinearAcousticRZ + semi-implicit adjustment", cases.config3_rz())
    run("config 1: R grid LinearAdvection1D, 100 cells (plumbing)", cases.kat_r())
    run("config 4 native-equivalent: RLZ 85 cells, native ragged rings x 64 levels (matrix-core DFT)",
        config4_native_equivalent(), steps=10, warmup=3)
