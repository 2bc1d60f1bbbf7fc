# hipTile.jl - binding libscythe_hip.so (include/scythe_hip.h) from Scythe.jl.
#
# The file a maintainer would add as src/hipTile.jl and include from src/Scythe.jl after semiimplicit.jl; it is the text of
# INTEGRATION.md sections 1-4.  NEVER EXECUTED: there is no julia in the image this repository is built and tested in.  What IS
# checked (tests/test_abi.py, CPU): the field offsets that Julia's layout rule for isbits structs (C layout: every field at
# the next multiple of its own alignment) gives the three structs below equal the C compiler's offsetof() for the header's
# structs, and so does the table SX_ABI_OFFSETS, which sx_check_layout() compares with fieldoffset() when the file is loaded.
#
# Replaces: createModelTile (src/semiimplicit.jl:44-124), advanceTimestep (:301-332, called :276), splineTransform! (called :285),
# the patchSpectral pull of the output path (:289-290).  integrate_model / initialize_model / run_model / model_loop stay as they are.

# ---- 1. descriptors ----------------------------------------------------------------------------------------------------------
const libsx = "libscythe_hip.so"            # on LD_LIBRARY_PATH, or an absolute path

struct SxGridDesc                           # mirrors sx_grid_desc (include/scythe_hip.h), field for field
    abi_version::Int32; geometry::Int32
    xmin::Float64; xmax::Float64
    num_cells::Int32
    l_q::Float64
    nvars::Int32
    bcl::Ptr{Int32}; bcl_k0::Ptr{Int32}; bcr::Ptr{Int32}
    zmin::Float64; zmax::Float64
    zDim::Int32; b_zDim::Int32
    bcb::Ptr{Int32}; bct::Ptr{Int32}
    ring_uniform_L::Int32
    tile_cell0::Int32; tile_num_cells::Int32; tile_num::Int32
    storage_f32::Int32                      # 0 = fp64 throughout (reference arithmetic); 1 = fp32-stored derivative planes; 2: see the header
end

struct SxModelDesc                          # mirrors sx_model_desc
    ts::Float64
    equation_set::Int32; semiimplicit::Int32
    params::Ptr{Float64}
    w_index::Int32; xi_index::Int32; col_var::Int32
    ref_state::Ptr{Float64}                 # Euler_test: [3][3][zDim] (sbar, xibar, mubar) x (value, z, zz), else C_NULL
end

const SX_GEOM = Dict("R" => 0, "RZ" => 1, "RL" => 2, "RLZ" => 3)
const SX_PARAMS = (:g, :K, :Cd, :Hfree, :Hb, :f, :S1, :c_0, :Kh, :Um, :Vm, :Pxi_bar)

function sx_bc(d::Dict)                     # CubicBSpline.* / Chebyshev.* Dict tag -> SX_BC_* code
    d == CubicBSpline.R0 && return Int32(0);  d == CubicBSpline.R1T0 && return Int32(1)
    d == CubicBSpline.R1T1 && return Int32(2); d == CubicBSpline.R1T2 && return Int32(3)
    d == CubicBSpline.R2T10 && return Int32(4); d == CubicBSpline.R2T20 && return Int32(5)
    d == CubicBSpline.R3 && return Int32(6);  d == CubicBSpline.PERIODIC && return Int32(7)
    throw(DomainError(0, "Unknown boundary condition"))
end

sxcheck(rc) = rc == 0 || error(unsafe_string(ccall((:sx_last_error, libsx), Cstring, ())))

# ---- 2. a GPU-resident ModelTile ---------------------------------------------------------------------------------------------
mutable struct HipModelTile                 # replaces ModelTile (src/semiimplicit.jl:18-42) on a worker
    handle::Ptr{Cvoid}
    model::ModelParameters
    tileSpectral::Matrix{Float64}           # host copy of tile.spectral [s_tile, nvars], filled after sx_advance (:323)
    patchSpectral::Matrix{Float64}          # host mirror [s_patch, nvars], filled only when the master asks for output (:289)
    # calcPatchMap / calcHaloMap (:79-86) as linear indices into ONE variable's column, from sx_index_maps:
    patchOwned::Vector{Int64}; tileOwned::Vector{Int64}      # sharedSpectral[patchOwned, v] .= tileSpectral[tileOwned, v]
    patchHalo::Vector{Int64};  tileHalo::Vector{Int64}       # what this tile sends (tile side) / where the NEXT tile adds it (patch side)
    haloRecvIndex::Vector{Int64}            # = the previous tile's patchHalo (handed over once at start-up, as :225 does)
end

struct SxDims                               # mirrors sx_dims
    n_points::Int64; n_hpoints::Int64
    n_vars::Int32; n_derivs::Int32; n_coord::Int32
    rDim::Int32; b_rDim::Int32; tile_rDim::Int32; tile_b_rDim::Int32
    zDim::Int32; b_zDim::Int32; kDim::Int32; n_blocks::Int32; tile_kDim::Int32; tile_n_blocks::Int32
    s_patch::Int64; s_tile::Int64; n_cols::Int64
end

# createModelTile(patch, tile, model, haloReceiveIndexMap)  (src/semiimplicit.jl:44-124, called at :179, :183)
function createHipModelTile(patch::AbstractGrid, tile_params::Matrix, model::ModelParameters, w::Int)
    gp = patch.params
    names = sort(collect(keys(gp.vars)); by = k -> gp.vars[k])
    bcl = Int32[sx_bc(gp.BCL[k]) for k in names]; bcr = Int32[sx_bc(gp.BCR[k]) for k in names]
    bcb = Int32[sx_bc(gp.BCB[k]) for k in names]; bct = Int32[sx_bc(gp.BCT[k]) for k in names]
    par = Float64[get(model.physical_params, p, 0.0) for p in SX_PARAMS]
    h = Ref{Ptr{Cvoid}}(C_NULL)
    GC.@preserve bcl bcr bcb bct par begin
        g = SxGridDesc(2, SX_GEOM[gp.geometry], gp.xmin, gp.xmax, gp.num_cells, gp.l_q, length(names),
                       pointer(bcl), C_NULL, pointer(bcr), gp.zmin, gp.zmax, gp.zDim, gp.b_zDim,
                       pointer(bcb), pointer(bct), 0,
                       Int32(tile_params[4, w] - 1), Int32(tile_params[3, w]), Int32(myid()), 0)
        eq = ccall((:sx_equation_set_id, libsx), Cint, (Cstring,), model.equation_set)
        eq < 0 && error("equation set $(model.equation_set) is not available on the HIP path")
        # Euler_test: ref = permutedims(cat(rs.sbar, rs.xibar, rs.mubar; dims = 3), (1, 2, 3)) flattened level-fastest,
        # kept alive in the GC.@preserve list; par[12] = rs.Pxi_bar (src/testModels.jl:171)
        m = SxModelDesc(model.ts, eq, model.options[:semiimplicit] ? 1 : 0, pointer(par),
                        get(gp.vars, "w", 0), get(gp.vars, "xi", 0), get(gp.vars, "h", 0), C_NULL)
        sxcheck(ccall((:sx_create, libsx), Cint, (Ref{SxGridDesc}, Ref{SxModelDesc}, Ref{Ptr{Cvoid}}), g, m, h))
    end
    d = Ref{SxDims}()
    sxcheck(ccall((:sx_get_dims, libsx), Cint, (Ptr{Cvoid}, Ref{SxDims}), h[], d))
    no = Ref{Int64}(0); nh = Ref{Int64}(0)
    sxcheck(ccall((:sx_index_map_sizes, libsx), Cint, (Ptr{Cvoid}, Ref{Int64}, Ref{Int64}), h[], no, nh))
    po = Vector{Int64}(undef, no[]); to = similar(po); ph = Vector{Int64}(undef, nh[]); th = similar(ph)
    sxcheck(ccall((:sx_index_maps, libsx), Cint, (Ptr{Cvoid}, Ptr{Int64}, Ptr{Int64}, Ptr{Int64}, Ptr{Int64}), h[], po, to, ph, th))
    mt = HipModelTile(h[], model, zeros(d[].s_tile, d[].n_vars), zeros(d[].s_patch, d[].n_vars), po, to, ph, th, Int64[])
    finalizer(t -> ccall((:sx_destroy, libsx), Cint, (Ptr{Cvoid},), t.handle), mt)
    return mt
end
# run_model (:222-226) hands every tile the map of what it receives, exactly as it does with haloReceiveIndexMap today:
#   mtile(w+1).haloRecvIndex = get_val_from(w, :(mtile.patchHalo))        (the first tile receives nothing)

# ---- layout self-check ------------------------------------------------------------------------------------------------------
# byte offsets of every field as the C compiler lays out include/scythe_hip.h on x86-64 / gfx950 hosts (LP64)
const SX_ABI_OFFSETS = Dict(
    :SxGridDesc => (abi_version = 0, geometry = 4, xmin = 8, xmax = 16, num_cells = 24, l_q = 32, nvars = 40, bcl = 48, bcl_k0 = 56,
                    bcr = 64, zmin = 72, zmax = 80, zDim = 88, b_zDim = 92, bcb = 96, bct = 104, ring_uniform_L = 112, tile_cell0 = 116,
                    tile_num_cells = 120, tile_num = 124, storage_f32 = 128, sizeof = 136),
    :SxModelDesc => (ts = 0, equation_set = 8, semiimplicit = 12, params = 16, w_index = 24, xi_index = 28, col_var = 32,
                     ref_state = 40, sizeof = 48),
    :SxDims => (n_points = 0, n_hpoints = 8, n_vars = 16, n_derivs = 20, n_coord = 24, rDim = 28, b_rDim = 32, tile_rDim = 36,
                tile_b_rDim = 40, zDim = 44, b_zDim = 48, kDim = 52, n_blocks = 56, tile_kDim = 60, tile_n_blocks = 64, s_patch = 72,
                s_tile = 80, n_cols = 88, sizeof = 96),
)

function sx_check_layout()
    for (T, name) in ((SxGridDesc, :SxGridDesc), (SxModelDesc, :SxModelDesc), (SxDims, :SxDims))
        want = SX_ABI_OFFSETS[name]
        for (i, f) in enumerate(fieldnames(T))
            fieldoffset(T, i) == getfield(want, f) || error("$name.$f is at byte $(fieldoffset(T, i)), the C header has it at $(getfield(want, f))")
        end
        sizeof(T) == want.sizeof || error("sizeof($name) = $(sizeof(T)), the C header says $(want.sizeof)")
    end
    ccall((:sx_abi_version, libsx), Cint, ()) == 2 || error("libscythe_hip.so has another SX_ABI_VERSION than this file (2)")
    return true
end

# ---- 3. the two per-step calls (host hop through the reference's SharedArray / RemoteChannel protocol) ----------------------
# advanceTimestep(mtile, sharedSpectral, haloSend, haloReceive, t)   (src/semiimplicit.jl:301-332)
# Drop-in that keeps the reference's SharedArray + RemoteChannel protocol (host hop; simplest to adopt):
function advanceTimestep(mtile::HipModelTile, sharedSpectral::SharedArray{Float64},
                         haloSend::RemoteChannel, haloReceive::RemoteChannel, t::Int64)
    sxcheck(ccall((:sx_advance, libsx), Cint, (Ptr{Cvoid}, Int32), mtile.handle, t))      # :305-317 on the GPU
    sxcheck(ccall((:sx_get_tile_spectral, libsx), Cint, (Ptr{Cvoid}, Ptr{Float64}), mtile.handle, mtile.tileSpectral))
    put!(haloSend, mtile.tileSpectral[mtile.tileHalo, :])                                    # :320  haloSendView
    sharedSpectral[mtile.patchOwned, :] .= view(mtile.tileSpectral, mtile.tileOwned, :)      # :323  patchIndexMap / tileView
    halo = take!(haloReceive)                                                                # :326
    isempty(mtile.haloRecvIndex) || (sharedSpectral[mtile.haloRecvIndex, :] .+= halo)        # :329  haloReceiveIndexMap
    return nothing
end

# splineTransform!(mtile.patchSplines, mtile.patchSpectral, gp, sharedSpectral, mtile.tile)  (:285)
function splineTransform!(mtile::HipModelTile, sharedSpectral::SharedArray{Float64})
    GC.@preserve sharedSpectral begin
        sxcheck(ccall((:sx_set_patch_spectral_b, libsx), Cint, (Ptr{Cvoid}, Ptr{Float64}),
                      mtile.handle, pointer(sharedSpectral)))
    end
    sxcheck(ccall((:sx_spline_transform, libsx), Cint, (Ptr{Cvoid},), mtile.handle))
end

# master output path: patch.spectral .= get_val_from(w, :(mtile.patchSpectral)); tileTransform!(...)  (:289-290)
function patchSpectral(mtile::HipModelTile)
    sxcheck(ccall((:sx_get_patch_spectral_a, libsx), Cint, (Ptr{Cvoid}, Ptr{Float64}),
                  mtile.handle, mtile.patchSpectral))
    return mtile.patchSpectral
end

# ---- 4. the exchange on the device (RCCL inside the library; what bench.py measures) -----------------------------------------
# once, after createHipModelTile on every worker.  cell0 / ncells = calcTileSizes rows 4 (spectralIndexL - 1) and 3 of every tile.
# mode 2: interface-only solve (least traffic; needs >= 9 cells per tile); 0: transposed solve; 1: the reference's halo + gather
function sx_comm_setup!(mtile::HipModelTile, ntiles::Integer, mytile::Integer, cell0::Vector{Int32}, ncells::Vector{Int32},
                        id::Vector{UInt8}; mode::Integer = 2)
    length(id) == 128 || error("the communicator id is 128 opaque bytes (sx_comm_unique_id on the first worker)")
    sxcheck(ccall((:sx_comm_init, libsx), Cint, (Ptr{Cvoid}, Int32, Int32, Ptr{Int32}, Ptr{Int32}, Int32, Ptr{UInt8}),
                  mtile.handle, ntiles, mytile - 1, cell0, ncells, mode, id))      # collective: ncclCommInitRank
end

function sx_comm_unique_id()                                # on ONE worker; hand the bytes to the others through any channel
    id = Vector{UInt8}(undef, 128)
    sxcheck(ccall((:sx_comm_unique_id, libsx), Cint, (Ptr{UInt8},), id))
    return id
end

# per step (replaces :320-329 and :285; nothing crosses PCIe):
function advanceTimestepDevice(mtile::HipModelTile, t::Int64)
    sxcheck(ccall((:sx_advance, libsx), Cint, (Ptr{Cvoid}, Int32), mtile.handle, t))
    sxcheck(ccall((:sx_exchange, libsx), Cint, (Ptr{Cvoid},), mtile.handle))
end
