"""ctypes binding of libscythe_hip.so (include/scythe_hip.h). No fallback: if the HIP library is missing or
cannot be loaded this module raises - the product path never routes through a CPU implementation."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# SCYTHE_HIP_LIB points at an alternative build of the same ABI (A/B timing of two builds on one box)
LIB_PATH = os.environ.get("SCYTHE_HIP_LIB") or os.path.join(_HERE, "libscythe_hip.so")

SX_ABI_VERSION = 2
GEOM = {"R": 0, "RZ": 1, "RL": 2, "RLZ": 3}
BC = {"R0": 0, "R1T0": 1, "R1T1": 2, "R1T2": 3, "R2T10": 4, "R2T20": 5, "R3": 6, "PERIODIC": 7}
PARAM_ORDER = ["g", "K", "Cd", "Hfree", "Hb", "f", "S1", "c_0", "Kh", "Um", "Vm", "Pxi_bar"]

P_I32 = C.POINTER(C.c_int32)
P_I64 = C.POINTER(C.c_int64)
P_D = C.POINTER(C.c_double)


class GridDesc(C.Structure):
    _fields_ = [("abi_version", C.c_int32), ("geometry", C.c_int32), ("xmin", C.c_double), ("xmax", C.c_double),
                ("num_cells", C.c_int32), ("l_q", C.c_double), ("nvars", C.c_int32),
                ("bcl", P_I32), ("bcl_k0", P_I32), ("bcr", P_I32),
                ("zmin", C.c_double), ("zmax", C.c_double), ("zDim", C.c_int32), ("b_zDim", C.c_int32),
                ("bcb", P_I32), ("bct", P_I32), ("ring_uniform_L", C.c_int32),
                ("tile_cell0", C.c_int32), ("tile_num_cells", C.c_int32), ("tile_num", C.c_int32),
                ("storage_f32", C.c_int32)]


class ModelDesc(C.Structure):
    _fields_ = [("ts", C.c_double), ("equation_set", C.c_int32), ("semiimplicit", C.c_int32), ("params", P_D),
                ("w_index", C.c_int32), ("xi_index", C.c_int32), ("col_var", C.c_int32), ("ref_state", P_D)]


class Dims(C.Structure):
    _fields_ = [("n_points", C.c_int64), ("n_hpoints", C.c_int64), ("n_vars", C.c_int32), ("n_derivs", C.c_int32),
                ("n_coord", C.c_int32), ("rDim", C.c_int32), ("b_rDim", C.c_int32), ("tile_rDim", C.c_int32),
                ("tile_b_rDim", C.c_int32), ("zDim", C.c_int32), ("b_zDim", C.c_int32), ("kDim", C.c_int32),
                ("n_blocks", C.c_int32), ("tile_kDim", C.c_int32), ("tile_n_blocks", C.c_int32),
                ("s_patch", C.c_int64), ("s_tile", C.c_int64), ("n_cols", C.c_int64)]


# every symbol include/scythe_hip.h declares: name -> (restype, argtypes)
_H = C.c_void_p
SYMBOLS = {
    "sx_create": (C.c_int, [C.POINTER(GridDesc), C.POINTER(ModelDesc), C.POINTER(_H)]),
    "sx_destroy": (C.c_int, [_H]),
    "sx_last_error": (C.c_char_p, []),
    "sx_abi_version": (C.c_int, []),
    "sx_equation_set_id": (C.c_int, [C.c_char_p]),
    "sx_get_dims": (C.c_int, [_H, C.POINTER(Dims)]),
    "sx_set_stream": (C.c_int, [_H, C.c_void_p]),
    "sx_synchronize": (C.c_int, [_H]),
    "sx_get_gridpoints": (C.c_int, [_H, P_D]),
    "sx_calc_tile_sizes": (C.c_int, [C.POINTER(GridDesc), C.c_int32, P_D]),
    "sx_spline_solve_check": (C.c_int, [C.c_int32, C.c_double, C.c_double, C.c_double, C.c_int32, C.c_int32, P_D, P_D, P_D, C.POINTER(C.c_int32)]),
    "sx_set_physical_values": (C.c_int, [_H, P_D]),
    "sx_get_physical": (C.c_int, [_H, P_D]),
    "sx_get_var_np1": (C.c_int, [_H, P_D]),
    "sx_cheb_column_ops": (C.c_int, [C.c_double, C.c_double, C.c_int32, C.c_int32, C.c_int32, C.c_int32, P_D, P_D, P_D, P_D, P_D]),
    "sx_index_map_sizes": (C.c_int, [_H, P_I64, P_I64]),
    "sx_index_maps": (C.c_int, [_H, P_I64, P_I64, P_I64, P_I64]),
    "sx_state_size": (C.c_int, [_H, P_I64]),
    "sx_get_state": (C.c_int, [_H, P_D]),
    "sx_set_state": (C.c_int, [_H, P_D]),
    "sx_get_tile_spectral": (C.c_int, [_H, P_D]),
    "sx_set_patch_spectral_b": (C.c_int, [_H, P_D]),
    "sx_get_patch_spectral_a": (C.c_int, [_H, P_D]),
    "sx_set_patch_spectral_a": (C.c_int, [_H, P_D]),
    "sx_spectral_transform": (C.c_int, [_H]),
    "sx_spline_transform": (C.c_int, [_H]),
    "sx_tile_transform": (C.c_int, [_H]),
    "sx_advance": (C.c_int, [_H, C.c_int32]),
    "sx_step": (C.c_int, [_H, C.c_int32]),
    "sx_physics": (C.c_int, [_H, C.c_int32]),
    "sx_check_nan": (C.c_int, [_H, P_I32]),
    "sx_max_abs": (C.c_int, [_H, P_D]),
    "sx_tile_b_device": (C.c_int, [_H, C.POINTER(C.c_void_p), P_I64, P_I64]),
    "sx_bind_tile_b": (C.c_int, [_H, C.c_void_p]),
    "sx_halo_add": (C.c_int, [_H, C.c_void_p]),
    "sx_bind_patch_b": (C.c_int, [_H, C.c_void_p, P_I64]),
    "sx_patch_a_device": (C.c_int, [_H, C.POINTER(C.c_void_p), P_I64, P_I64]),
    "sx_a2a_configure": (C.c_int, [_H, C.c_int32, C.c_int32, P_I32, P_I32]),
    "sx_a2a_col_starts": (C.c_int, [_H, P_I64]),
    "sx_a2a_pack_b": (C.c_int, [_H, C.c_void_p]),
    "sx_a2a_solve": (C.c_int, [_H, C.c_void_p, C.c_void_p]),
    "sx_a2a_unpack_a": (C.c_int, [_H, C.c_void_p]),
    "sx_iface_configure": (C.c_int, [_H, C.c_int32, C.c_int32, P_I32, P_I32]),
    "sx_iface_col_starts": (C.c_int, [_H, P_I64]),
    "sx_iface_local": (C.c_int, [_H, C.c_void_p]),
    "sx_iface_reduce": (C.c_int, [_H, C.c_void_p, C.c_void_p]),
    "sx_iface_apply": (C.c_int, [_H, C.c_void_p]),
    "sx_comm_unique_id": (C.c_int, [C.c_char_p]),
    "sx_comm_prepare": (C.c_int, [_H, C.c_int32, C.c_int32, P_I32, P_I32, C.c_int32]),
    "sx_comm_init": (C.c_int, [_H, C.c_int32, C.c_int32, P_I32, P_I32, C.c_int32, C.c_char_p]),
    "sx_comm_attach": (C.c_int, [_H, C.c_int32, C.c_int32, P_I32, P_I32, C.c_int32, C.c_void_p]),
    "sx_exchange": (C.c_int, [_H]),
    "sx_comm_init_local": (C.c_int, [C.POINTER(_H), C.c_int32, P_I32, P_I32, C.c_int32]),
    "sx_exchange_local": (C.c_int, [C.POINTER(_H), C.c_int32]),
    "sx_enable_timers": (C.c_int, [_H, C.c_int32]),
    "sx_reset_timers": (C.c_int, [_H]),
    "sx_timer_only": (C.c_int, [_H, C.c_char_p]),
    "sx_get_timers": (C.c_int, [_H, C.c_int32, C.POINTER(C.c_char_p), P_D, P_I64, P_I32]),
    "sx_kernel_bytes": (C.c_int, [_H, C.c_char_p, P_D]),
}

_lib = None


class ScytheHipError(RuntimeError):
    pass


def load():
    """Load libscythe_hip.so and bind every declared symbol. Raises if the library or a symbol is missing."""
    global _lib
    if _lib is not None:
        return _lib
    # One HIP runtime per process: PyTorch-ROCm bundles its own libamdhip64.so (same SONAME as /opt/rocm's). Import
    # torch first so that libscythe_hip.so binds to the runtime torch (and RCCL) already use; loading two copies
    # leaves the second one without a device.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    if not os.path.exists(LIB_PATH):
        raise ScytheHipError(
            "libscythe_hip.so not found at %s - build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(there is no CPU fallback)" % LIB_PATH)
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SYMBOLS.items():
        fn = getattr(lib, name)          # AttributeError if the symbol is not exported
        fn.restype = res
        fn.argtypes = args
    if lib.sx_abi_version() != SX_ABI_VERSION:
        raise ScytheHipError("ABI version mismatch between _lib.py and libscythe_hip.so")
    _lib = lib
    return lib


def check(rc):
    if rc != 0:
        raise ScytheHipError(load().sx_last_error().decode())
