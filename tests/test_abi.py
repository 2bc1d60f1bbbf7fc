"""CPU-only: libscythe_hip.so loads, exports every symbol include/scythe_hip.h declares, and refuses to compute
without a GPU (no silent fallback)."""
import ctypes as C
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    src = open(os.path.join(ROOT, "include", "scythe_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(sx_[a-z0-9_]+)\s*\(", src)))


def test_header_and_binding_agree():
    import scythe_jl_amd as S
    from scythe_jl_amd import _lib
    declared = _declared_symbols()
    assert len(declared) >= 30
    assert sorted(_lib.SYMBOLS) == declared


def test_library_exports_every_declared_symbol():
    import scythe_jl_amd as S
    lib = C.CDLL(S.LIB_PATH)
    for name in _declared_symbols():
        assert hasattr(lib, name), name
    import re
    want = int(re.search(r"#define SX_ABI_VERSION (\d+)", open(os.path.join(ROOT, "include", "scythe_hip.h")).read()).group(1))
    assert S.load().sx_abi_version() == want


def test_equation_set_names():
    import scythe_jl_amd as S
    lib = S.load()
    for name, eid in [("LinearAdvection1D", 0), ("LinearAdvectionRZ", 1), ("LinearAdvectionRL", 2), ("LinearAdvectionRLZ", 3),
                      ("Oneway_ShallowWater_Slab", 4), ("Twoway_ShallowWater_Slab", 5),
                      ("Oneway_ShallowWater_HeightResolvedBL", 6)]:
        assert lib.sx_equation_set_id(name.encode()) == eid
    assert lib.sx_equation_set_id(b"Kepert2017_TCBL") == -1      # broken / out of scope in the reference too


def test_struct_sizes_match_the_header():
    """Compile a tiny C program against the header and compare sizeof() with the ctypes mirrors."""
    import subprocess, tempfile
    from scythe_jl_amd import _lib
    with tempfile.TemporaryDirectory() as d:
        src = os.path.join(d, "sz.c")
        open(src, "w").write('#include <stdio.h>\n#include "scythe_hip.h"\nint main(){printf("%zu %zu %zu\\n",'
                             'sizeof(sx_grid_desc),sizeof(sx_model_desc),sizeof(sx_dims));return 0;}')
        exe = os.path.join(d, "sz")
        subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), src, "-o", exe])
        a, b, c = map(int, subprocess.check_output([exe]).split())
    assert (a, b, c) == (C.sizeof(_lib.GridDesc), C.sizeof(_lib.ModelDesc), C.sizeof(_lib.Dims))


def test_create_fails_loudly_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    import scythe_jl_amd as S
    gp = S.GridParameters(geometry="R", xmin=-50.0, xmax=50.0, num_cells=100, vars={"u": 1})
    with pytest.raises(S.ScytheHipError, match="no HIP device"):
        S.createGrid(gp)


def test_error_behaviour_mirrors_reference():
    import scythe_jl_amd as S
    with pytest.raises(ValueError, match="Unknown geometry"):          # src/spectralGrid.jl:90
        S.createGrid(S.GridParameters(geometry="XY", xmin=0, xmax=1, num_cells=10))
    with pytest.raises(ValueError, match="not implemented"):           # src/spectralGrid.jl:87
        S.createGrid(S.GridParameters(geometry="Z", xmin=0, xmax=1, num_cells=10))
    mp = S.ModelParameters(ts=1.0, equation_set="Kepert2017_TCBL",
                           grid_params=S.GridParameters(geometry="R", xmin=0, xmax=1, num_cells=10))
    with pytest.raises(ValueError, match="not defined"):
        from scythe_jl_amd.model import model_desc
        model_desc(mp, mp.grid_params)


def test_calc_tile_sizes_host_helper():
    import scythe_jl_amd as S
    gp = S.GridParameters(geometry="R", xmin=-50.0, xmax=50.0, num_cells=100, vars={"u": 1})
    ts = S.calcTileSizes(gp, 3)
    assert ts.shape == (5, 3)
    assert ts[2].sum() == 100 and list(ts[3]) == [1, 35, 68]
    assert np.allclose(ts[1, :-1], ts[0, 1:]) and ts[0, 0] == -50.0 and ts[1, -1] == 50.0
    assert list(ts[4]) == [102, 99, 99]
    # RL: tiles balance gridpoints, not cells (src/semiimplicit.jl:144 prints row 5)
    gl = S.GridParameters(geometry="RL", xmin=0.0, xmax=3.0e5, num_cells=100, vars={"h": 1})
    tl = S.calcTileSizes(gl, 4)
    assert tl[2].sum() == 100 and tl[4].sum() == 2 * 300 ** 2 + 6 * 300
    assert tl[4].max() / tl[4].min() < 1.1 and tl[2, 0] > tl[2, -1]
    with pytest.raises(S.ScytheHipError):
        S.calcTileSizes(gp, 40)
