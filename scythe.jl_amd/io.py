"""CSV I/O in the format the reference's notebooks read and write (write_output -> Springsteel write_grid,
src/io.py:3-13; read_physical_grid, src/semiimplicit.jl:134): one row per gridpoint, coordinate columns
(r[, l][, z]) followed by one column per variable, file name physical_out_<time>.csv."""
import os

import numpy as np

from .model import getGridpoints

_COORD = {"R": ["r"], "RZ": ["r", "z"], "RL": ["r", "l"], "RLZ": ["r", "l", "z"]}


def read_physical_grid(path, patch_params, run):
    """CSV -> list of [N_tile, V] arrays (one per local tile of `run`), variables matched by column name."""
    with open(path) as f:
        header = f.readline().strip().split(",")
    data = np.loadtxt(path, delimiter=",", skiprows=1, ndmin=2)
    names = patch_params.var_names()
    cols = [header.index(n) for n in names]
    vals = data[:, cols]
    out, p0 = [], 0
    starts = np.cumsum([0] + [int(run.layout.tile_sizes[4, t]) for t in range(run.num_tiles)])
    for g, t in zip(run.tiles, run.tile_ids):
        p0 = int(starts[t])
        out.append(np.asfortranarray(vals[p0:p0 + g.N]))
    if len(data) != int(starts[-1]):
        raise ValueError("initial conditions have %d rows, grid has %d points" % (len(data), int(starts[-1])))
    return out


def write_output(run, model, time):
    """physical_out_<time>.csv with the values (derivative slot 1) of every variable on the local tiles."""
    gp = model.grid_params
    names = gp.var_names()
    path = os.path.join(model.output_dir, "physical_out_%s.csv" % float(time))
    rows = []
    for g in run.tiles:
        g.tileTransform_()          # patch.spectral -> physical before every output (src/semiimplicit.jl:241, 290)
        pts = getGridpoints(g)
        pts = pts.reshape(len(pts), -1)
        rows.append(np.concatenate([pts, g.physical[:, :, 0]], axis=1))
    arr = np.concatenate(rows, axis=0)
    np.savetxt(path, arr, delimiter=",", header=",".join(_COORD[gp.geometry] + names), comments="", fmt="%.17g")
    return path
