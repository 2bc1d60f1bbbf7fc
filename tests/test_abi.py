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


# ---------------------------------------------------------------------------------------------- Julia glue (julia/hipTile.jl)
_JL_TYPES = {"Int32": (4, 4), "Int64": (8, 8), "Float64": (8, 8), "Ptr": (8, 8), "Cint": (4, 4)}
_JL_TO_C = {"SxGridDesc": "sx_grid_desc", "SxModelDesc": "sx_model_desc", "SxDims": "sx_dims"}


def _julia_structs():
    """{struct name: [(field, julia type)]} parsed from the immutable `struct ... end` blocks of julia/hipTile.jl."""
    src = open(os.path.join(ROOT, "julia", "hipTile.jl")).read()
    out = {}
    for name, body in re.findall(r"^struct (\w+)[^\n]*\n(.*?)^end", src, flags=re.S | re.M):
        fields = []
        for line in body.splitlines():
            line = line.split("#")[0]
            fields += re.findall(r"(\w+)::(\w+)", line)
        out[name] = fields
    return out


def _julia_layout(fields):
    """Julia lays out an isbits struct like C: every field at the next multiple of its own alignment, the size rounded up to
    the largest alignment (https://docs.julialang.org/en/v1/manual/calling-c-and-fortran-code/#Struct-Type-Correspondences)."""
    off, amax, out = 0, 1, {}
    for f, t in fields:
        size, al = _JL_TYPES[t]
        off = (off + al - 1) // al * al
        out[f] = off
        off += size
        amax = max(amax, al)
    out["sizeof"] = (off + amax - 1) // amax * amax
    return out


def _c_offsets():
    """offsetof() of every field of the three ABI structs, from a C program compiled against include/scythe_hip.h."""
    import subprocess, tempfile
    hdr = open(os.path.join(ROOT, "include", "scythe_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    prog = ['#include <stdio.h>', '#include <stddef.h>', '#include "scythe_hip.h"', 'int main(){']
    names = {}
    for cname in _JL_TO_C.values():
        body = re.search(r"typedef struct %s \{(.*?)\} %s;" % (cname, cname), hdr, flags=re.S).group(1)
        fields = []
        for decl in body.split(";"):
            decl = decl.strip()
            if not decl:
                continue
            first, *rest = decl.split(",")
            fields.append(re.findall(r"(\w+)\s*$", first.strip())[0])
            fields += [r.strip().lstrip("*").strip() for r in rest]
        names[cname] = fields
        for f in fields:
            prog.append('printf("%s %s %%zu\\n", offsetof(%s, %s));' % (cname, f, cname, f))
        prog.append('printf("%s sizeof %%zu\\n", sizeof(%s));' % (cname, cname))
    prog.append("return 0;}")
    with tempfile.TemporaryDirectory() as d:
        src = os.path.join(d, "off.c")
        open(src, "w").write("\n".join(prog))
        exe = os.path.join(d, "off")
        subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), src, "-o", exe])
        lines = subprocess.check_output([exe]).decode().split("\n")
    out = {c: {} for c in names}
    for ln in lines:
        if ln.strip():
            c, f, o = ln.split()
            out[c][f] = int(o)
    return out, names


def test_julia_struct_field_offsets_equal_the_c_header():
    """What a ccall user can get wrong and no other test sees: the Julia mirror structs must have the header's fields in the
    header's order with types of the same size and alignment, i.e. the same OFFSETS (sizes alone would miss two swapped
    Int32 fields or an Int32 / Float64 pair exchanged inside one 16-byte slot)."""
    c_off, c_names = _c_offsets()
    jl = _julia_structs()
    for jname, cname in _JL_TO_C.items():
        assert [f for f, _ in jl[jname]] == c_names[cname], (jname, "field order / names differ from " + cname)
        lay = _julia_layout(jl[jname])
        assert lay == c_off[cname], (jname, {k: (lay.get(k), c_off[cname].get(k)) for k in set(lay) | set(c_off[cname]) if lay.get(k) != c_off[cname].get(k)})


def test_julia_offset_table_equals_the_c_header():
    """SX_ABI_OFFSETS in julia/hipTile.jl - the table sx_check_layout() holds fieldoffset() against when Julia loads the file -
    is the C compiler's offsetof() for every field."""
    c_off, _ = _c_offsets()
    src = open(os.path.join(ROOT, "julia", "hipTile.jl")).read()
    tab = re.search(r"const SX_ABI_OFFSETS = Dict\((.*?)^\)", src, flags=re.S | re.M).group(1)
    for jname, cname in _JL_TO_C.items():
        body = re.search(r":%s => \((.*?)\)," % jname, tab, flags=re.S).group(1)
        got = {k: int(v) for k, v in re.findall(r"(\w+) = (\d+)", body)}
        assert got == c_off[cname], (jname, got, c_off[cname])


def test_julia_file_is_the_text_of_integration_md():
    md = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    jl = set(l.strip() for l in open(os.path.join(ROOT, "julia", "hipTile.jl")).read().splitlines())
    blocks = re.findall(r"```julia\n(.*?)```", md, flags=re.S)
    assert len(blocks) == 4
    for b in blocks:
        for line in b.splitlines():
            if line.strip():
                assert line.strip() in jl, line
    # every C entry point the glue calls exists in the header
    src = open(os.path.join(ROOT, "julia", "hipTile.jl")).read()
    called = set(re.findall(r"ccall\(\(:(sx_\w+), libsx\)", src))
    assert called and called <= set(_declared_symbols()), called - set(_declared_symbols())


def test_ctypes_mirror_field_offsets_equal_the_c_header():
    """The same check for the Python host mirror's ctypes structures (sizes alone are pinned above)."""
    from scythe_jl_amd import _lib
    c_off, c_names = _c_offsets()
    for st, cname in ((_lib.GridDesc, "sx_grid_desc"), (_lib.ModelDesc, "sx_model_desc"), (_lib.Dims, "sx_dims")):
        assert [f[0] for f in st._fields_] == c_names[cname]
        for f in c_names[cname]:
            assert getattr(st, f).offset == c_off[cname][f], (cname, f)
