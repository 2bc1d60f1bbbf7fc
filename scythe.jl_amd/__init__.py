"""scythe.jl_amd - MI355X-native spectral-transform time stepping for Scythe.jl (hot path only).

The directory name contains a dot, so import it through the `scythe_jl_amd` shim at the repo root."""
from ._lib import load, ScytheHipError, LIB_PATH
from .model import (CubicBSpline, Chebyshev, GridParameters, ModelParameters, Grid, createGrid, getGridpoints,
                    calcTileSizes, num_columns, checkCFL, comm_unique_id)
from .driver import (PatchLayout, LocalExchange, DistExchange, A2ALayout, LocalA2AExchange, DistA2AExchange, LibExchange, LocalLibExchange, ModelRun,
                     integrate_model)
from .io import read_physical_grid, write_output
from . import thermodynamics, reference_state
from .reference_state import ReferenceState, Chebyshev1D
