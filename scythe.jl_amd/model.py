"""Host-side mirror of the reference's interface for the spectral-transform time-stepping path.

Names, argument meaning and error behaviour follow the reference (Julia); `!` is spelled with a trailing
underscore.  Every object that computes anything is a thin wrapper over a libscythe_hip handle.

  GridParameters      src/spectralGrid.jl:20-45 (vestige of the live Springsteel definition)
  ModelParameters     src/Scythe.jl:8-21
  createGrid          src/spectralGrid.jl:63-94
  ModelTile           src/semiimplicit.jl:18-42
  createModelTile     src/semiimplicit.jl:44-124
  advanceTimestep     src/semiimplicit.jl:301-332
"""
import ctypes as C
import math
from dataclasses import dataclass, field
from typing import Dict, Optional

import numpy as np

from . import _lib as L


# ----------------------------------------------------------------------------- boundary-condition namespaces
class CubicBSpline:
    """Boundary-condition tags (Dicts in the reference: CubicBSpline.R0 etc., models/*.jl)."""
    mubar = 3
    R0 = {"R0": 0}
    R1T0 = {"α1": -4.0, "β1": -1.0}
    R1T1 = {"α1": 0.0, "β1": 1.0}
    R1T2 = {"α1": 2.0, "β1": -1.0}
    R2T10 = {"β1": 1.0, "β2": -0.5}
    R2T20 = {"β1": -1.0, "β2": 0.0}
    R3 = {"R3": 0}
    PERIODIC = {"PERIODIC": 0}


class Chebyshev:
    R0 = {"R0": 0}
    R1T0 = {"α0": 0.0}
    R1T1 = {"α1": 0.0}
    R1T2 = {"α2": 0.0}


_SPLINE_BCS = [(CubicBSpline.R0, "R0"), (CubicBSpline.R1T0, "R1T0"), (CubicBSpline.R1T1, "R1T1"),
               (CubicBSpline.R1T2, "R1T2"), (CubicBSpline.R2T10, "R2T10"), (CubicBSpline.R2T20, "R2T20"),
               (CubicBSpline.R3, "R3"), (CubicBSpline.PERIODIC, "PERIODIC")]
_CHEB_BCS = [(Chebyshev.R0, "R0"), (Chebyshev.R1T0, "R1T0"), (Chebyshev.R1T1, "R1T1"), (Chebyshev.R1T2, "R1T2")]


def bc_name(bc, table=_SPLINE_BCS):
    """Dict tag (or plain string) -> canonical BC name."""
    if isinstance(bc, str):
        if bc not in L.BC:
            raise ValueError("Unknown boundary condition %r" % (bc,))
        return bc
    for d, name in table:
        if bc == d:
            return name
    raise ValueError("Unknown boundary condition %r" % (bc,))


def _default_b_zDim(zDim):
    return int(min(zDim, math.floor(((2 * zDim) - 1) / 3) + 1)) if zDim > 0 else 0


# ----------------------------------------------------------------------------- parameters
@dataclass
class GridParameters:
    geometry: str = "R"
    xmin: float = 0.0
    xmax: float = 0.0
    num_cells: int = 0
    rDim: Optional[int] = None
    b_rDim: Optional[int] = None
    l_q: float = 2.0
    BCL: Dict = field(default_factory=dict)
    BCR: Dict = field(default_factory=dict)
    lDim: int = 0
    b_lDim: int = 0
    zmin: float = 0.0
    zmax: float = 0.0
    zDim: int = 0
    b_zDim: Optional[int] = None
    BCB: Dict = field(default_factory=dict)
    BCT: Dict = field(default_factory=dict)
    vars: Dict = field(default_factory=lambda: {"u": 1})
    spectralIndexL: int = 1
    spectralIndexR: Optional[int] = None
    patchOffsetL: Optional[int] = None
    patchOffsetR: Optional[int] = None
    tile_num: int = 0
    # extensions (not in the reference): uniform ring table and a separate k = 0 inner BC
    ring_uniform_L: int = 0
    BCL_k0: Optional[Dict] = None
    storage: str = "f64"       # "f32": derivative slots of `physical` stored as fp32; "f32x": also the spectral transform
                               # intermediates (sx_grid_desc.storage_f32 = 1 / 2)

    def __post_init__(self):
        if self.rDim is None:
            self.rDim = self.num_cells * CubicBSpline.mubar
        if self.b_rDim is None:
            self.b_rDim = self.num_cells + 3
        if self.b_zDim is None:
            self.b_zDim = _default_b_zDim(self.zDim)
        if self.spectralIndexR is None:
            self.spectralIndexR = self.spectralIndexL + self.b_rDim - 1
        if self.patchOffsetL is None:
            self.patchOffsetL = (self.spectralIndexL - 1) * 3
        if self.patchOffsetR is None:
            self.patchOffsetR = self.patchOffsetL + self.rDim

    def var_names(self):
        return [n for n, _ in sorted(self.vars.items(), key=lambda kv: kv[1])]


@dataclass
class ModelParameters:
    ts: float = 0.0
    integration_time: float = 1.0
    output_interval: float = 1.0
    equation_set: str = "LinearAdvection1D"
    initial_conditions: str = "ic.csv"
    output_dir: str = "./output/"
    ref_state_file: str = ""
    grid_params: GridParameters = None
    physical_params: Dict = field(default_factory=dict)
    options: Dict = field(default_factory=lambda: {"semiimplicit": False, "exact_reference_state": False})
    ref_state: object = None       # ReferenceState; built from ref_state_file when the equation set needs one and this is None


# ----------------------------------------------------------------------------- descriptors
def _i32(values):
    return (C.c_int32 * len(values))(*values)


def grid_desc(patch: GridParameters, tile_cell0=0, tile_num_cells=None, tile_num=0):
    """Flatten patch GridParameters (+ tile range) into the C descriptor. Returns (desc, keepalive)."""
    if patch.geometry not in L.GEOM:
        raise ValueError("Unknown geometry")          # DomainError(0, "Unknown geometry") src/spectralGrid.jl:90
    names = patch.var_names()
    d = L.GridDesc()
    keep = {}
    get = lambda dct, n, table: L.BC[bc_name((dct or {}).get(n, "R0"), table)]
    keep["bcl"] = _i32([get(patch.BCL, n, _SPLINE_BCS) for n in names])
    keep["bcr"] = _i32([get(patch.BCR, n, _SPLINE_BCS) for n in names])
    k0 = patch.BCL_k0 if patch.BCL_k0 is not None else patch.BCL
    keep["bcl0"] = _i32([get({**(patch.BCL or {}), **(k0 or {})}, n, _SPLINE_BCS) for n in names])
    keep["bcb"] = _i32([get(patch.BCB, n, _CHEB_BCS) for n in names])
    keep["bct"] = _i32([get(patch.BCT, n, _CHEB_BCS) for n in names])
    d.abi_version, d.geometry = L.SX_ABI_VERSION, L.GEOM[patch.geometry]
    d.xmin, d.xmax, d.num_cells, d.l_q, d.nvars = patch.xmin, patch.xmax, patch.num_cells, patch.l_q, len(names)
    d.bcl, d.bcl_k0, d.bcr = keep["bcl"], keep["bcl0"], keep["bcr"]
    d.zmin, d.zmax, d.zDim, d.b_zDim = patch.zmin, patch.zmax, patch.zDim, patch.b_zDim or 0
    d.bcb, d.bct = keep["bcb"], keep["bct"]
    d.ring_uniform_L = patch.ring_uniform_L
    d.tile_cell0 = tile_cell0
    d.tile_num_cells = patch.num_cells if tile_num_cells is None else tile_num_cells
    d.tile_num = tile_num
    if patch.storage not in ("f64", "f32", "f32x"):
        raise ValueError("GridParameters.storage must be 'f64', 'f32' or 'f32x'")
    d.storage_f32 = {"f64": 0, "f32": 1, "f32x": 2}[patch.storage]
    return d, keep


def model_desc(model: Optional[ModelParameters], patch: GridParameters):
    m = L.ModelDesc()
    keep = {}
    lib = L.load()
    if model is None:
        m.ts, m.equation_set, m.semiimplicit = 0.0, 99, 0
        keep["par"] = (C.c_double * len(L.PARAM_ORDER))()
    else:
        eq = lib.sx_equation_set_id(model.equation_set.encode())
        if eq < 0:
            # getfield(Scythe, Symbol(...)) raises UndefVarError for an unknown name (src/semiimplicit.jl:359-361)
            raise ValueError("equation set %r is not defined on the HIP path" % model.equation_set)
        pp = {(k if isinstance(k, str) else str(k)).lstrip(":"): v for k, v in model.physical_params.items()}
        opts = {str(k).lstrip(":"): v for k, v in (model.options or {}).items()}
        if model.equation_set == "Euler_test":
            # createModelTile builds mtile.ref_state from model.ref_state_file (src/semiimplicit.jl:44-124)
            if model.ref_state is None:
                from . import reference_state as RS
                build = RS.exact_reference_state if opts.get("exact_reference_state", False) else RS.interpolate_reference_file
                model.ref_state = build(model)
            keep["ref"] = model.ref_state.packed()
            m.ref_state = keep["ref"].ctypes.data_as(L.P_D)
            pp.setdefault("Pxi_bar", model.ref_state.Pxi_bar)
        keep["par"] = (C.c_double * len(L.PARAM_ORDER))(*[float(pp.get(k, 0.0)) for k in L.PARAM_ORDER])
        m.ts, m.equation_set = model.ts, eq
        m.semiimplicit = int(bool(opts.get("semiimplicit", False)))
    m.params = keep["par"]
    m.w_index = patch.vars.get("w", 0)
    m.xi_index = patch.vars.get("xi", 0)
    m.col_var = patch.vars.get("h", 0)
    return m, keep


def calcTileSizes(patch: GridParameters, num_tiles: int):
    """calcTileSizes(patch, n) -> 5 x n matrix: xmin, xmax, num_cells, spectralIndexL, gridpoints
    (src/semiimplicit.jl:141-144, 157-168)."""
    d, keep = grid_desc(patch)
    out = np.zeros((5, num_tiles), order="F")
    L.check(L.load().sx_calc_tile_sizes(C.byref(d), num_tiles, out.ctypes.data_as(L.P_D)))
    return out


# ----------------------------------------------------------------------------- grid / tile objects
class Grid:
    """A tile (or the whole patch) resident on the GPU: createGrid(GridParameters) + the ModelTile state."""

    def __init__(self, patch: GridParameters, model: Optional[ModelParameters] = None, tile_cell0=0,
                 tile_num_cells=None, tile_num=0):
        lib = L.load()
        self.patch_params = patch
        self.model = model
        gd, k1 = grid_desc(patch, tile_cell0, tile_num_cells, tile_num)
        md, k2 = model_desc(model, patch)
        h = C.c_void_p()
        L.check(lib.sx_create(C.byref(gd), C.byref(md), C.byref(h)))
        self._h = h
        self._lib = lib
        dims = L.Dims()
        L.check(lib.sx_get_dims(h, C.byref(dims)))
        self.dims = dims
        self.cell0 = tile_cell0
        self.ncells = patch.num_cells if tile_num_cells is None else tile_num_cells
        DX = (patch.xmax - patch.xmin) / patch.num_cells
        self.params = GridParameters(
            geometry=patch.geometry, xmin=patch.xmin + tile_cell0 * DX, xmax=patch.xmin + (tile_cell0 + self.ncells) * DX,
            num_cells=self.ncells, l_q=patch.l_q, BCL={k: CubicBSpline.R0 for k in patch.vars},
            BCR={k: CubicBSpline.R0 for k in patch.vars}, lDim=int(dims.n_hpoints) if "L" in patch.geometry else 0,
            zmin=patch.zmin, zmax=patch.zmax, zDim=patch.zDim, b_zDim=patch.b_zDim, BCB=patch.BCB, BCT=patch.BCT,
            vars=patch.vars, spectralIndexL=tile_cell0 + 1, tile_num=tile_num, ring_uniform_L=patch.ring_uniform_L,
            storage=patch.storage)

    def close(self):
        if getattr(self, "_h", None):
            self._lib.sx_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- shapes
    @property
    def N(self):
        return int(self.dims.n_points)

    @property
    def V(self):
        return int(self.dims.n_vars)

    @property
    def D(self):
        return int(self.dims.n_derivs)

    @property
    def n_cols(self):
        return int(self.dims.n_cols)

    # -- state access (reference layouts, Fortran order == Julia column-major)
    def set_physical_values(self, values):
        v = np.asfortranarray(values, dtype=np.float64)
        assert v.shape == (self.N, self.V), (v.shape, (self.N, self.V))
        L.check(self._lib.sx_set_physical_values(self._h, v.ctypes.data_as(L.P_D)))

    @property
    def physical(self):
        out = np.zeros((self.N, self.V, self.D), order="F")
        L.check(self._lib.sx_get_physical(self._h, out.ctypes.data_as(L.P_D)))
        return out

    def get_state(self):
        """Restart blob of this tile (A coefficients + Adams-Bashforth history), see sx_get_state."""
        n = C.c_int64(0)
        L.check(self._lib.sx_state_size(self._h, C.byref(n)))
        out = np.zeros(n.value)
        L.check(self._lib.sx_get_state(self._h, out.ctypes.data_as(L.P_D)))
        return out

    def set_state(self, blob):
        b = np.ascontiguousarray(blob, dtype=np.float64)
        L.check(self._lib.sx_set_state(self._h, b.ctypes.data_as(L.P_D)))

    @property
    def var_np1(self):
        out = np.zeros((self.N, self.V), order="F")
        L.check(self._lib.sx_get_var_np1(self._h, out.ctypes.data_as(L.P_D)))
        return out

    @property
    def spectral(self):
        """tile.spectral (B coefficients) in the reference tile layout."""
        out = np.zeros((int(self.dims.s_tile), self.V), order="F")
        L.check(self._lib.sx_get_tile_spectral(self._h, out.ctypes.data_as(L.P_D)))
        return out

    def set_patch_spectral_b(self, shared):
        s = np.asfortranarray(shared, dtype=np.float64)
        assert s.shape == (int(self.dims.s_patch), self.V)
        L.check(self._lib.sx_set_patch_spectral_b(self._h, s.ctypes.data_as(L.P_D)))

    def set_patch_spectral_a(self, a):
        s = np.asfortranarray(a, dtype=np.float64)
        assert s.shape == (int(self.dims.s_patch), self.V)
        L.check(self._lib.sx_set_patch_spectral_a(self._h, s.ctypes.data_as(L.P_D)))

    @property
    def patchSpectral(self):
        out = np.zeros((int(self.dims.s_patch), self.V), order="F")
        L.check(self._lib.sx_get_patch_spectral_a(self._h, out.ctypes.data_as(L.P_D)))
        return out

    # -- operators
    def spectralTransform_(self):
        L.check(self._lib.sx_spectral_transform(self._h))

    def splineTransform_(self):
        L.check(self._lib.sx_spline_transform(self._h))

    def tileTransform_(self):
        L.check(self._lib.sx_tile_transform(self._h))

    def advance(self, t):
        L.check(self._lib.sx_advance(self._h, int(t)))

    def step(self, t):
        """One-tile patch: advanceTimestep + splineTransform! in one call (sx_step; replayed from a hipGraph with SX_GRAPH=1)."""
        L.check(self._lib.sx_step(self._h, int(t)))

    def physics(self, t):
        L.check(self._lib.sx_physics(self._h, int(t)))

    def max_abs(self):
        """max |var_np1[:, v]| per variable, reduced on the device (sx_max_abs)."""
        out = np.zeros(self.V)
        L.check(self._lib.sx_max_abs(self._h, out.ctypes.data_as(L.P_D)))
        return out

    def check_nan(self):
        f = C.c_int32(0)
        L.check(self._lib.sx_check_nan(self._h, C.byref(f)))
        return bool(f.value)

    def synchronize(self):
        L.check(self._lib.sx_synchronize(self._h))

    def set_stream(self, stream_ptr):
        L.check(self._lib.sx_set_stream(self._h, C.c_void_p(stream_ptr)))

    # -- device exchange helpers
    def tile_b_device(self):
        p, r, c = C.c_void_p(), C.c_int64(), C.c_int64()
        L.check(self._lib.sx_tile_b_device(self._h, C.byref(p), C.byref(r), C.byref(c)))
        return p.value, r.value, c.value

    def patch_a_device(self):
        p, r, c = C.c_void_p(), C.c_int64(), C.c_int64()
        L.check(self._lib.sx_patch_a_device(self._h, C.byref(p), C.byref(r), C.byref(c)))
        return p.value, r.value, c.value

    def bind_tile_b(self, dev_ptr):
        L.check(self._lib.sx_bind_tile_b(self._h, C.c_void_p(dev_ptr)))

    def bind_patch_b(self, dev_ptr, row_offsets):
        ro = np.ascontiguousarray(row_offsets, dtype=np.int64)
        assert len(ro) == int(self.dims.b_rDim)
        L.check(self._lib.sx_bind_patch_b(self._h, C.c_void_p(dev_ptr), ro.ctypes.data_as(L.P_I64)))

    def halo_add(self, dev_ptr):
        L.check(self._lib.sx_halo_add(self._h, C.c_void_p(dev_ptr)))

    # -- transposed (all-to-all) patch solve
    def a2a_configure(self, cell0, ncells, my_tile):
        n = len(cell0)
        c0 = (C.c_int32 * n)(*cell0)
        nc = (C.c_int32 * n)(*ncells)
        L.check(self._lib.sx_a2a_configure(self._h, n, my_tile, c0, nc))
        cs = np.zeros(n + 1, dtype=np.int64)
        L.check(self._lib.sx_a2a_col_starts(self._h, cs.ctypes.data_as(L.P_I64)))
        return cs

    def a2a_pack_b(self, dev_send):
        L.check(self._lib.sx_a2a_pack_b(self._h, C.c_void_p(dev_send)))

    def a2a_solve(self, dev_recv, dev_send):
        L.check(self._lib.sx_a2a_solve(self._h, C.c_void_p(dev_recv), C.c_void_p(dev_send)))

    def a2a_unpack_a(self, dev_recv):
        L.check(self._lib.sx_a2a_unpack_a(self._h, C.c_void_p(dev_recv)))

    # -- interface-only (partitioned) patch solve (sx_iface.hip): same call pattern as the transposed solve, 10 rows per tile
    def iface_configure(self, cell0, ncells, my_tile):
        n = len(cell0)
        L.check(self._lib.sx_iface_configure(self._h, n, my_tile, (C.c_int32 * n)(*cell0), (C.c_int32 * n)(*ncells)))
        cs = np.zeros(n + 1, dtype=np.int64)
        L.check(self._lib.sx_iface_col_starts(self._h, cs.ctypes.data_as(L.P_I64)))
        return cs

    def iface_local(self, dev_send):
        L.check(self._lib.sx_iface_local(self._h, C.c_void_p(dev_send)))

    def iface_reduce(self, dev_recv, dev_send):
        L.check(self._lib.sx_iface_reduce(self._h, C.c_void_p(dev_recv), C.c_void_p(dev_send)))

    def iface_apply(self, dev_recv):
        L.check(self._lib.sx_iface_apply(self._h, C.c_void_p(dev_recv)))

    def index_maps(self):
        """calcPatchMap / calcHaloMap (src/semiimplicit.jl:79-86): 1-based linear indices into one variable's column:
        (patch_owned, tile_owned, patch_halo, tile_halo)."""
        no, nh = C.c_int64(), C.c_int64()
        L.check(self._lib.sx_index_map_sizes(self._h, C.byref(no), C.byref(nh)))
        a = [np.zeros(n, dtype=np.int64) for n in (no.value, no.value, nh.value, nh.value)]
        L.check(self._lib.sx_index_maps(self._h, *[x.ctypes.data_as(L.P_I64) for x in a]))
        return tuple(a)

    # -- exchange over RCCL inside the library (sx_comm.cpp)
    def comm_prepare(self, cell0, ncells, my_tile, mode):
        """The non-collective part of comm_init (librccl bound, tile table checked, exchange buffers allocated): raises on
        THIS rank alone if it cannot be done, so that the ranks can agree before the collective comm_init."""
        n = len(cell0)
        L.check(self._lib.sx_comm_prepare(self._h, n, my_tile, (C.c_int32 * n)(*cell0), (C.c_int32 * n)(*ncells), EXCHANGE_MODES[mode]))

    def comm_init(self, cell0, ncells, my_tile, mode, unique_id):
        """Collective over all tiles: ncclCommInitRank on this tile's device + exchange buffers. mode "a2a" or "gather"."""
        n = len(cell0)
        c0 = (C.c_int32 * n)(*cell0)
        nc = (C.c_int32 * n)(*ncells)
        assert len(unique_id) == 128
        L.check(self._lib.sx_comm_init(self._h, n, my_tile, c0, nc, EXCHANGE_MODES[mode], bytes(unique_id)))

    def exchange(self):
        """Halo / shared sum / patch solve of one step on the handle's stream (src/semiimplicit.jl:320-329, 272-285)."""
        L.check(self._lib.sx_exchange(self._h))

    # -- timers
    def enable_timers(self, on=True):
        L.check(self._lib.sx_enable_timers(self._h, int(on)))

    def timer_only(self, name=None):
        """Time only the kernel with this timer name (None: all)."""
        L.check(self._lib.sx_timer_only(self._h, name.encode() if name else None))

    def reset_timers(self):
        L.check(self._lib.sx_reset_timers(self._h))

    def timers(self):
        n = C.c_int32(0)
        names = (C.c_char_p * 32)()
        ms = (C.c_double * 32)()
        calls = (C.c_int64 * 32)()
        L.check(self._lib.sx_get_timers(self._h, 32, names, ms, calls, C.byref(n)))
        return {names[i].decode(): (ms[i], calls[i]) for i in range(n.value)}

    def kernel_bytes(self, name):
        b = C.c_double(0.0)
        L.check(self._lib.sx_kernel_bytes(self._h, name.encode(), C.byref(b)))
        return b.value


EXCHANGE_MODES = {"a2a": 0, "gather": 1, "iface": 2}


def comm_unique_id():
    """ncclGetUniqueId through the library: 128 bytes that rank 0 hands to every other rank before Grid.comm_init."""
    buf = C.create_string_buffer(128)
    L.check(L.load().sx_comm_unique_id(buf))
    return buf.raw


def createGrid(gp: GridParameters, model: Optional[ModelParameters] = None):
    """createGrid(gp): the whole patch as one device-resident grid (src/semiimplicit.jl:130)."""
    if gp.geometry == "Z":
        raise ValueError("Z column model not implemented yet")     # src/spectralGrid.jl:86-88
    return Grid(gp, model)


def getGridpoints(grid: Grid):
    """R: vector; RL: [:,1]=r,[:,2]=lambda; RZ: r,z; RLZ: r,lambda,z (src/semiimplicit.jl:59)."""
    n, nc = grid.N, int(grid.dims.n_coord)
    out = np.zeros((n, nc), order="F")
    L.check(grid._lib.sx_get_gridpoints(grid._h, out.ctypes.data_as(L.P_D)))
    return out[:, 0].copy() if nc == 1 else out


def num_columns(grid: Grid):
    return int(grid.dims.n_hpoints) if "Z" in grid.patch_params.geometry else 0


def checkCFL(grid: Grid):
    """checkCFL (src/semiimplicit.jl:737-751): error on NaN in physical[:, v, 1]."""
    if grid.check_nan():
        raise RuntimeError("NaN found in a model variable! CFL condition likely violated")
