"""Extract the reference's only known-answer data (printed notebook outputs) into a small JSON fixture.

Source: /root/reference/notebooks/LinearAdvection_example.ipynb
  cell 2 output  : getGridpoints(R grid, 100 cells on [-50, 50])  (first 13 / last 12 of 300)
  cell 7 output  : final.u after 2000 steps                       (first 13 / last 12 of 300)
  cell 9 output  : l2_norm(initial.u - final.u)
  cell 6 output  : @time integrate_model wall-clock
Run in the build container only (the reference tree does not travel to the GPU box).
"""
import json, os, sys

NB = "/root/reference/notebooks/LinearAdvection_example.ipynb"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "linear_advection_kat.json")


def floats(lines):
    return [float(s) for s in lines if s.strip() and "⋮" not in s and "Vector" not in s]


def main():
    nb = json.load(open(NB))
    cells = nb["cells"]
    gp = floats("".join(cells[2]["outputs"][0]["data"]["text/plain"]).split("\n"))
    fu = floats("".join(cells[7]["outputs"][0]["data"]["text/plain"]).split("\n"))
    l2 = float("".join(cells[9]["outputs"][0]["data"]["text/plain"]))
    assert len(gp) == 25 and len(fu) == 25
    idx = list(range(13)) + list(range(288, 300))
    fix = {
        "source": "notebooks/LinearAdvection_example.ipynb (cells 2, 7, 9)",
        "model": {"geometry": "R", "xmin": -50.0, "xmax": 50.0, "num_cells": 100, "l_q": 2.0,
                  "BCL": "PERIODIC", "BCR": "PERIODIC", "ts": 0.05, "steps": 2000,
                  "equation_set": "LinearAdvection1D", "c_0": 1.0, "K": 0.0,
                  "ic": "exp(-(x/20)^2) at the mish points", "workers": 2},
        "index_0based": idx,
        "gridpoints": gp,
        "final_u": fu,
        "l2_norm": l2,
    }
    json.dump(fix, open(OUT, "w"), indent=1)
    print("wrote", OUT)


if __name__ == "__main__":
    sys.exit(main())
