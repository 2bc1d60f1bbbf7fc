"""CSV I/O in the format the reference's notebooks read and write (write_output -> Springsteel write_grid,
src/io.jl:3-13; read_physical_grid, src/semiimplicit.jl:134): one row per gridpoint, coordinate columns
(r[, l][, z]) followed by one column per variable (and per derivative slot, suffixed), file name
physical_out_<time>.csv with <time> = string(round(t; digits=2)) as Julia prints a Float64, plus the spectral
coefficients in spectral_out_<time>.csv.  The notebooks address columns by name (`initial.r`, `final.u`,
notebooks/LinearAdvection_example.ipynb:265-270), so the extra columns do not disturb them."""
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


def julia_float_string(x):
    """string(x::Float64) as Julia prints it (shortest round-trip digits; fixed notation for 1e-4 <= |x| < 1e6, otherwise
    d.ddde[-]X; always at least one decimal): 0.3 -> "0.3", 100.0 -> "100.0", 1e6 -> "1.0e6", 1e-5 -> "1.0e-5"."""
    x = float(x)
    if x != x:
        return "NaN"
    if x in (float("inf"), float("-inf")):
        return "Inf" if x > 0 else "-Inf"
    if x == 0.0:
        return "-0.0" if str(x).startswith("-") else "0.0"
    mant, exp = ("%r" % abs(x)), 0
    if "e" in mant:
        mant, e = mant.split("e")
        exp = int(e)
    ip, _, fp = mant.partition(".")
    digits = (ip + fp).lstrip("0")
    # decimal exponent of the first significant digit
    lead = len(ip.lstrip("0")) - 1 if ip.strip("0") else -(len(fp) - len(fp.lstrip("0")) + 1)
    e10 = lead + exp
    digits = digits.rstrip("0") or "0"
    sign = "-" if x < 0 else ""
    if -5 < e10 < 6:
        if e10 >= 0:
            whole = digits[: e10 + 1].ljust(e10 + 1, "0")
            frac = digits[e10 + 1:] or "0"
        else:
            whole, frac = "0", "0" * (-e10 - 1) + digits
        return sign + whole + "." + frac
    return sign + digits[0] + "." + (digits[1:] or "0") + "e" + str(e10)


def output_time_tag(t):
    """`time = string(round(t; digits=2))` (src/io.jl:5)."""
    return julia_float_string(round(float(t), 2))


_SUFFIX = {"u": "", "r": "_r", "rr": "_rr", "l": "_l", "ll": "_ll", "z": "_z", "zz": "_zz"}
_SLOTS = {"R": ["u", "r", "rr"], "RZ": ["u", "r", "rr", "z", "zz"], "RL": ["u", "r", "rr", "l", "ll"],
          "RLZ": ["u", "r", "rr", "l", "ll", "z", "zz"]}


def write_output(run, model, time, derivatives=True, spectral=True):
    """write_output(grid, model, t) (src/io.jl:3-13): physical_out_<tag>.csv with the values of every variable on the local
    tiles (and, with derivatives=True, the derivative slots as <var>_r, <var>_rr, ... columns) and spectral_out_<tag>.csv
    with the patch A coefficients (mtile.patchSpectral, src/semiimplicit.jl:289), one column per variable.  Springsteel's
    write_grid is not in the reference tree; its column naming is recalled, only `r` and the variable names are pinned by
    the notebooks."""
    gp = model.grid_params
    names = gp.var_names()
    tag = output_time_tag(time)
    path = os.path.join(model.output_dir, "physical_out_%s.csv" % tag)
    rows = []
    slots = _SLOTS[gp.geometry] if derivatives else ["u"]
    for g in run.tiles:
        g.tileTransform_()          # patch.spectral -> physical before every output (src/semiimplicit.jl:241, 290)
        pts = getGridpoints(g)
        pts = pts.reshape(len(pts), -1)
        ph = g.physical
        rows.append(np.concatenate([pts] + [ph[:, :, d] for d in range(len(slots))], axis=1))
    arr = np.concatenate(rows, axis=0)
    header = _COORD[gp.geometry] + [n + _SUFFIX[s] for s in slots for n in names]
    np.savetxt(path, arr, delimiter=",", header=",".join(header), comments="", fmt="%.17g")
    a = run.patch_spectral() if spectral else None   # transposed solve: assembled from the tiles' owned rows (rank 0 writes)
    if a is not None:
        idx = np.arange(1, a.shape[0] + 1, dtype=np.float64)[:, None]
        np.savetxt(os.path.join(model.output_dir, "spectral_out_%s.csv" % tag), np.concatenate([idx, a], axis=1), delimiter=",",
                   header=",".join(["i"] + names), comments="", fmt="%.17g")
    return path
