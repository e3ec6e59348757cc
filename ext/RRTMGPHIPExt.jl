# RRTMGPHIPExt.jl — the Julia side of the drop-in: device methods of RRTMGP.jl for an
# MI355X back end, bound to libhip_rrtmgp.so through `ccall`.
#
# This file is what a maintainer adds to RRTMGP.jl next to ext/RRTMGPCUDAExt.jl (plus a
# `HIPRRTMGP` weak dependency that only has to provide `HIPRRTMGP.libpath`).  It takes the
# place of ext/RRTMGPCUDAExt.jl + ext/cuda/*.jl: the same 13 methods, dispatched on the
# device type, but instead of compiling the kernel bodies with CUDA.jl they call the
# hand-written HIP kernels.  No CUDA.jl, KernelAbstractions or AMDGPU.jl is involved.
#
# NOT EXERCISED IN THIS REPOSITORY'S CI: there is no Julia in the build image.  The C ABI
# it binds is exercised by the Python host mirror (rrtmgp.jl_amd/) with identical struct
# layouts; `__init__` checks those layouts against the library at load time.
#
# Arrays: `ClimaComms.array_type(::HIPDevice) = Array`, so every getter / broadcast of the
# reference (getters.jl, grid_adaptation.jl:228-256, Fluxes.jl:311,407,423) keeps working on
# host memory, and the library stages state -> HBM and fluxes -> host inside each solve
# (5-13 KB per column, SURVEY.md §8(d)).  A host model that already keeps its state in HBM
# passes device pointers instead (`mem = RRTMGP_MEM_DEVICE`), which is the path bench.py times.
module RRTMGPHIPExt

import ClimaComms
import RRTMGP
import RRTMGP.Parameters as RP
import RRTMGP.AngularDiscretizations: AngularDiscretization
import RRTMGP.Fluxes: FluxLW, FluxSW
import RRTMGP.Sources: SourceLWNoScat, SourceLW2Str, SourceSW2Str
import RRTMGP.BCs: LwBCs, SwBCs
import RRTMGP.Optics: OneScalar, TwoStream, compute_col_gas!, compute_relative_humidity!
import RRTMGP.AtmosphericStates: AtmosphericState, GrayAtmosphericState, CloudState, AerosolState,
    TransposedStateCache, GrayOpticalThicknessSchneider2004, GrayOpticalThicknessOGorman2008
import RRTMGP.VolumeMixingRatios: Vmr, VmrGM
import RRTMGP.LookUpTables: LookUpLW, LookUpSW, LookUpCld, LookUpAerosolMerra, LookUpMinor
import RRTMGP.RTESolver: rte_lw_noscat_solve!, rte_lw_2stream_solve!, rte_sw_noscat_solve!, rte_sw_2stream_solve!

export HIPDevice

"""
    HIPDevice(id = 0)

An MI355X reached through libhip_rrtmgp.so.  Not a CPU device, so the workspaces build a
`TransposedStateCache` by default (src/rte/RTE.jl:26-27); it is ignored here (the library
reads the caller's `(nlay, ncol)` slabs directly: one workgroup owns one column).
"""
struct HIPDevice <: ClimaComms.AbstractDevice
    id::Int
end
HIPDevice() = HIPDevice(0)
ClimaComms.array_type(::HIPDevice) = Array
ClimaComms.device_functional(::HIPDevice) = true

const libhip = Ref{String}("libhip_rrtmgp.so")   # set RRTMGP_HIP_LIBRARY to the built .so

# ---- C structs of include/rrtmgp_hip.h (field order and types must match) -------------------
const P = Ptr{Cvoid}
struct MinorDesc
    n_min_absrb::Int64; n_contrib::Int64
    bnd_st::Ptr{Int64}; gpt_st::Ptr{Int64}; gasdata::Ptr{Int64}; kminor::P
end
struct GasLookupDesc
    ftype::Int32; is_sw::Int32
    n_gpt::Int64; n_bnd::Int64; n_eta::Int64; n_p_ref::Int64; n_t_ref::Int64; n_gases::Int64; n_t_plnk::Int64
    idx_h2o::Int64
    p_ref_tropo::Float64; p_ref_min::Float64; t_ref_min::Float64; t_ref_max::Float64; solar_src_tot::Float64
    key_species::Ptr{Int64}; major_gpt2bnd::Ptr{Int64}
    kmajor::P; planck_fraction::P; t_planck::P; tot_planck::P; ln_p_ref::P; t_ref::P; vmr_ref::P
    minor_lower::MinorDesc; minor_upper::MinorDesc
    rayl_lower::P; rayl_upper::P; solar_src_scaled::P
end
struct CloudLookupDesc
    ftype::Int32; _pad::Int32
    nband::Int64; nrghice::Int64; nsize_liq::Int64; nsize_ice::Int64
    bounds::P; liqdata::P; icedata::P
end
struct AerosolLookupDesc
    ftype::Int32; _pad::Int32
    nband::Int64; nbin::Int64; nrh::Int64; iband_550nm::Int64
    size_bin_limits::P; rh_levels::P; dust::P; sea_salt::P; sulfate::P; black_carbon_rh::P; black_carbon::P
    organic_carbon_rh::P; organic_carbon::P
end
struct AtmosStateDesc
    mem::Int32; vmr_kind::Int32
    ncol::Int64; nlay::Int64; ngas::Int64
    layerdata::P; p_lev::P; t_lev::P; t_sfc::P; lat::P
    vmr_h2o::P; vmr_o3::P; vmr::P
    cld_r_eff_liq::P; cld_r_eff_ice::P; cld_path_liq::P; cld_path_ice::P; cld_frac::P
    cld_cover_lw::P; cld_cover_sw::P
    ice_rgh::Int64
    aero_size::P; aero_mass::P; aod_sw_ext::P; aod_sw_sca::P
end
struct LwBcsDesc
    mem::Int32; _pad::Int32; sfc_emis::P; inc_flux::P
end
struct SwBcsDesc
    mem::Int32; _pad::Int32; cos_zenith::P; toa_flux::P; sfc_alb_direct::P; sfc_alb_diffuse::P
end
struct FluxOutDesc
    mem::Int32; layout::Int32; flux_up::P; flux_dn::P; flux_net::P; flux_dn_dir::P
    band_flux_up::P; band_flux_dn::P; band_flux_net::P
    clear_flux_up::P; clear_flux_dn::P; clear_flux_net::P; clear_flux_dn_dir::P
end
struct SolveOpts
    n_gauss_angles::Int32; metric_mem::Int32; metric_scaling::P; seed::UInt64; col_offset::Int64
end
struct GrayStateDesc
    mem::Int32; otp_kind::Int32
    ncol::Int64; nlay::Int64
    lat::P; p_lay::P; p_lev::P; t_lay::P; t_lev::P; t_sfc::P
    otp::NTuple{5, Float64}; stefan::Float64
end
struct ParamsDesc
    grav::Float64; molmass_dryair::Float64; molmass_water::Float64; gas_constant::Float64
    kappa_d::Float64; stefan::Float64; avogad::Float64
end

struct PrepareOpts
    steps::Int32; interpolation::Int32; bottom_extrapolation::Int32; isothermal_boundary_layer::Int32
    z_mem::Int32; idx_h2o::Int32
    center_z::P; face_z::P
    p_min::Float64; t_min::Float64; t_max::Float64
end

const ABI_STRUCTS = (MinorDesc, GasLookupDesc, CloudLookupDesc, AerosolLookupDesc, AtmosStateDesc, LwBcsDesc,
                     SwBcsDesc, FluxOutDesc, SolveOpts, GrayStateDesc, ParamsDesc, PrepareOpts)

function __init__()
    libhip[] = get(ENV, "RRTMGP_HIP_LIBRARY", libhip[])
    for (i, T) in enumerate(ABI_STRUCTS)
        n = ccall((:rrtmgp_hip_abi_sizeof, libhip[]), Cint, (Cint,), i - 1)
        n == sizeof(T) || error("RRTMGPHIPExt: ABI mismatch for $T (library $n bytes, binding $(sizeof(T)))")
    end
end

function check(rc::Cint)
    rc == 0 && return nothing
    buf = Vector{UInt8}(undef, 1024)
    ccall((:rrtmgp_hip_last_error, libhip[]), Cint, (Ptr{UInt8}, Csize_t), buf, length(buf))
    error("libhip_rrtmgp: " * unsafe_string(pointer(buf)))
end

ptr(::Nothing) = C_NULL
ptr(a::AbstractArray) = Ptr{Cvoid}(pointer(a))
ftype(::Type{Float32}) = Int32(4)
ftype(::Type{Float64}) = Int32(8)

# ---- handles: created once per lookup / per (device, ncol, nlay, FT), reused every step ------
const LOOKUPS = IdDict{Any, Ptr{Cvoid}}()
const WORKSPACES = Dict{Tuple{Int, Int, Int, DataType}, Ptr{Cvoid}}()

minor_desc(m::LookUpMinor) =
    MinorDesc(size(m.gasdata, 2), size(m.kminor, 3), pointer(m.bnd_st), pointer(m.gpt_st), pointer(m.gasdata), ptr(m.kminor))

function lookup_handle(dev::HIPDevice, lkp::Union{LookUpLW{FT}, LookUpSW{FT}}) where {FT}
    get!(LOOKUPS, lkp) do
        is_sw = lkp isa LookUpSW
        n_eta, n_pp, n_t, n_gpt = size(lkp.kmajor)
        d = GasLookupDesc(
            ftype(FT), Int32(is_sw), n_gpt, size(lkp.key_species, 3), n_eta, n_pp - 1, n_t,
            size(lkp.ref_points.vmr_ref, 2), is_sw ? 0 : length(lkp.planck.t_planck), lkp.idx_h2o,
            lkp.p_ref_tropo, lkp.p_ref_min, lkp.t_ref_min, lkp.t_ref_max, is_sw ? lkp.solar_src_tot : 0.0,
            pointer(lkp.key_species), pointer(lkp.band_data.major_gpt2bnd), ptr(lkp.kmajor),
            is_sw ? C_NULL : ptr(lkp.planck.planck_fraction), is_sw ? C_NULL : ptr(lkp.planck.t_planck),
            is_sw ? C_NULL : ptr(lkp.planck.tot_planck), ptr(lkp.ref_points.ln_p_ref), ptr(lkp.ref_points.t_ref),
            ptr(lkp.ref_points.vmr_ref), minor_desc(lkp.minor_lower), minor_desc(lkp.minor_upper),
            is_sw ? ptr(lkp.rayl_lower) : C_NULL, is_sw ? ptr(lkp.rayl_upper) : C_NULL,
            is_sw ? ptr(lkp.solar_src_scaled) : C_NULL)
        h = Ref{Ptr{Cvoid}}()
        GC.@preserve lkp check(ccall((:rrtmgp_hip_gas_lookup_create, libhip[]), Cint,
                                     (Ref{GasLookupDesc}, Cint, Ref{Ptr{Cvoid}}), d, dev.id, h))
        h[]
    end
end
function lookup_handle(dev::HIPDevice, lkp::LookUpCld)
    get!(LOOKUPS, lkp) do
        FT = eltype(lkp.liqdata)
        d = CloudLookupDesc(ftype(FT), 0, lkp.dims[1], lkp.dims[2], lkp.dims[3], lkp.dims[4], ptr(lkp.bounds),
                            ptr(lkp.liqdata), ptr(lkp.icedata))
        h = Ref{Ptr{Cvoid}}()
        GC.@preserve lkp check(ccall((:rrtmgp_hip_cloud_lookup_create, libhip[]), Cint,
                                     (Ref{CloudLookupDesc}, Cint, Ref{Ptr{Cvoid}}), d, dev.id, h))
        h[]
    end
end
function lookup_handle(dev::HIPDevice, lkp::LookUpAerosolMerra)
    get!(LOOKUPS, lkp) do
        FT = eltype(lkp.dust)
        d = AerosolLookupDesc(ftype(FT), 0, size(lkp.dust, 3), size(lkp.size_bin_limits, 2), length(lkp.rh_levels),
                              lkp.iband_550nm, ptr(lkp.size_bin_limits), ptr(lkp.rh_levels), ptr(lkp.dust),
                              ptr(lkp.sea_salt), ptr(lkp.sulfate), ptr(lkp.black_carbon_rh), ptr(lkp.black_carbon),
                              ptr(lkp.organic_carbon_rh), ptr(lkp.organic_carbon))
        h = Ref{Ptr{Cvoid}}()
        GC.@preserve lkp check(ccall((:rrtmgp_hip_aerosol_lookup_create, libhip[]), Cint,
                                     (Ref{AerosolLookupDesc}, Cint, Ref{Ptr{Cvoid}}), d, dev.id, h))
        h[]
    end
end
lookup_handle(::HIPDevice, ::Nothing) = C_NULL

function workspace(dev::HIPDevice, ncol, nlay, ::Type{FT}) where {FT}
    get!(WORKSPACES, (dev.id, ncol, nlay, FT)) do
        h = Ref{Ptr{Cvoid}}()
        check(ccall((:rrtmgp_hip_workspace_create, libhip[]), Cint, (Cint, Int64, Int64, Int32, Ref{Ptr{Cvoid}}),
                    dev.id, ncol, nlay, ftype(FT), h))
        h[]
    end
end

# ---- descriptors of the caller-owned structs ----------------------------------------------------
vmr_fields(v::VmrGM) = (Int32(0), length(v.vmr), ptr(v.vmr_h2o), ptr(v.vmr_o3), ptr(v.vmr))
vmr_fields(v::Vmr) = (Int32(1), size(v.vmr, 1), C_NULL, C_NULL, ptr(v.vmr))

function state_desc(as::AtmosphericState, use_cld::Bool, use_aero::Bool)
    nlay, ncol = size(as.layerdata, 2), size(as.layerdata, 3)
    kind, ngas, ph2o, po3, pvmr = vmr_fields(as.vmr)
    cs = use_cld ? as.cloud_state : nothing
    ae = use_aero ? as.aerosol_state : nothing
    AtmosStateDesc(0, kind, ncol, nlay, ngas, ptr(as.layerdata), ptr(as.p_lev), ptr(as.t_lev), ptr(as.t_sfc),
                   ptr(as.lat), ph2o, po3, pvmr,
                   cs === nothing ? C_NULL : ptr(cs.cld_r_eff_liq), cs === nothing ? C_NULL : ptr(cs.cld_r_eff_ice),
                   cs === nothing ? C_NULL : ptr(cs.cld_path_liq), cs === nothing ? C_NULL : ptr(cs.cld_path_ice),
                   cs === nothing ? C_NULL : ptr(cs.cld_frac), cs === nothing ? C_NULL : ptr(cs.cld_cover_lw),
                   cs === nothing ? C_NULL : ptr(cs.cld_cover_sw), cs === nothing ? 1 : cs.ice_rgh,
                   ae === nothing ? C_NULL : ptr(ae.aero_size), ae === nothing ? C_NULL : ptr(ae.aero_mass),
                   ae === nothing ? C_NULL : ptr(ae.aod_sw_ext), ae === nothing ? C_NULL : ptr(ae.aod_sw_sca))
end

# FluxLW / FluxSW on a non-CPU device hold plain (ncol, nlev) arrays (Fluxes.jl:45-49)
# FluxBand keeps (nlev, ncol, n_bnd) (Fluxes.jl:170-186); its flux_net is filled later by
# update_net_fluxes! (update_fluxes.jl:198-201), so the solve is not asked for it.
band_ptrs(::Nothing) = (C_NULL, C_NULL, C_NULL)
band_ptrs(b) = (ptr(b.flux_up), ptr(b.flux_dn), C_NULL)
# The clear-sky slots stay NULL: the reference's L2 runs two solves (update_fluxes.jl:39-65), which
# works unchanged; a host that wants the one-pass form passes its clear-sky buffers here.
flux_desc(f::FluxLW, band = nothing) =
    FluxOutDesc(0, 0, ptr(f.flux_up), ptr(f.flux_dn), ptr(f.flux_net), C_NULL, band_ptrs(band)...,
                C_NULL, C_NULL, C_NULL, C_NULL)
flux_desc(f::FluxSW, band = nothing) =
    FluxOutDesc(0, 0, ptr(f.flux_up), ptr(f.flux_dn), ptr(f.flux_net), ptr(f.flux_dn_dir), band_ptrs(band)...,
                C_NULL, C_NULL, C_NULL, C_NULL)

# McICA: the host seeds Random (update_fluxes.jl:149-156); one draw from it keys the counter-based stream
opts(n_angles = 1) = SolveOpts(Int32(n_angles), 0, C_NULL, rand(UInt64), 0)

# ---- the device methods (same signatures as ext/cuda/*.jl) ---------------------------------------
function rte_lw_2stream_solve!(dev::HIPDevice, flux::FluxLW, flux_lw::FluxLW, band_flux, src_lw::SourceLW2Str,
                               bcs_lw::LwBCs, op::TwoStream, as::AtmosphericState, state_cache, lookup_lw::LookUpLW,
                               lookup_lw_cld = nothing, lookup_lw_aero = nothing)
    FT = eltype(flux_lw.flux_up)
    nlay, ncol = size(as.layerdata, 2), size(as.layerdata, 3)
    ws = workspace(dev, ncol, nlay, FT)
    GC.@preserve as bcs_lw flux_lw band_flux lookup_lw check(ccall(
        (:rrtmgp_hip_rte_lw_2stream_solve, libhip[]), Cint,
        (P, P, P, P, Ref{AtmosStateDesc}, Ref{LwBcsDesc}, Ref{FluxOutDesc}, Ref{SolveOpts}),
        ws, lookup_handle(dev, lookup_lw), lookup_handle(dev, lookup_lw_cld), lookup_handle(dev, lookup_lw_aero),
        state_desc(as, !isnothing(lookup_lw_cld), !isnothing(lookup_lw_aero)),
        LwBcsDesc(0, 0, ptr(bcs_lw.sfc_emis), ptr(bcs_lw.inc_flux)), flux_desc(flux_lw, band_flux), opts()))
    return nothing
end

function rte_lw_noscat_solve!(dev::HIPDevice, flux::FluxLW, flux_lw::FluxLW, src_lw::SourceLWNoScat, bcs_lw::LwBCs,
                              op::OneScalar, angle_disc::AngularDiscretization, as::AtmosphericState, state_cache,
                              lookup_lw::LookUpLW, lookup_lw_cld = nothing, lookup_lw_aero = nothing)
    FT = eltype(flux_lw.flux_up)
    nlay, ncol = size(as.layerdata, 2), size(as.layerdata, 3)
    ws = workspace(dev, ncol, nlay, FT)
    GC.@preserve as bcs_lw flux_lw lookup_lw check(ccall(
        (:rrtmgp_hip_rte_lw_noscat_solve, libhip[]), Cint,
        (P, P, P, P, Ref{AtmosStateDesc}, Ref{LwBcsDesc}, Ref{FluxOutDesc}, Ref{SolveOpts}),
        ws, lookup_handle(dev, lookup_lw), lookup_handle(dev, lookup_lw_cld), lookup_handle(dev, lookup_lw_aero),
        state_desc(as, !isnothing(lookup_lw_cld), !isnothing(lookup_lw_aero)),
        LwBcsDesc(0, 0, ptr(bcs_lw.sfc_emis), ptr(bcs_lw.inc_flux)), flux_desc(flux_lw),
        opts(angle_disc.n_gauss_angles)))
    return nothing
end

function rte_sw_2stream_solve!(dev::HIPDevice, flux::FluxSW, flux_sw::FluxSW, band_flux, op::TwoStream, bcs_sw::SwBCs,
                               src_sw::SourceSW2Str, as::AtmosphericState, state_cache, lookup_sw::LookUpSW,
                               lookup_sw_cld = nothing, lookup_sw_aero = nothing)
    FT = eltype(flux_sw.flux_up)
    nlay, ncol = size(as.layerdata, 2), size(as.layerdata, 3)
    ws = workspace(dev, ncol, nlay, FT)
    GC.@preserve as bcs_sw flux_sw band_flux lookup_sw check(ccall(
        (:rrtmgp_hip_rte_sw_2stream_solve, libhip[]), Cint,
        (P, P, P, P, Ref{AtmosStateDesc}, Ref{SwBcsDesc}, Ref{FluxOutDesc}, Ref{SolveOpts}),
        ws, lookup_handle(dev, lookup_sw), lookup_handle(dev, lookup_sw_cld), lookup_handle(dev, lookup_sw_aero),
        state_desc(as, !isnothing(lookup_sw_cld), !isnothing(lookup_sw_aero)),
        SwBcsDesc(0, 0, ptr(bcs_sw.cos_zenith), ptr(bcs_sw.toa_flux), ptr(bcs_sw.sfc_alb_direct),
                  ptr(bcs_sw.sfc_alb_diffuse)), flux_desc(flux_sw, band_flux), opts()))
    return nothing
end

function rte_sw_noscat_solve!(dev::HIPDevice, flux::FluxSW, flux_sw::FluxSW, op::OneScalar, bcs_sw::SwBCs,
                              as::AtmosphericState, state_cache, lookup_sw::LookUpSW)
    FT = eltype(flux_sw.flux_up)
    nlay, ncol = size(as.layerdata, 2), size(as.layerdata, 3)
    ws = workspace(dev, ncol, nlay, FT)
    GC.@preserve as bcs_sw flux_sw lookup_sw check(ccall(
        (:rrtmgp_hip_rte_sw_noscat_solve, libhip[]), Cint,
        (P, P, Ref{AtmosStateDesc}, Ref{SwBcsDesc}, Ref{FluxOutDesc}, Ref{SolveOpts}),
        ws, lookup_handle(dev, lookup_sw), state_desc(as, false, false),
        SwBcsDesc(0, 0, ptr(bcs_sw.cos_zenith), ptr(bcs_sw.toa_flux), C_NULL, C_NULL), flux_desc(flux_sw), opts()))
    return nothing
end

# ---- gray variants ---------------------------------------------------------------------------------
otp_fields(o::GrayOpticalThicknessSchneider2004) = (Int32(0), (Float64(o.α), Float64(o.te), Float64(o.tt), Float64(o.Δt), 0.0))
otp_fields(o::GrayOpticalThicknessOGorman2008) = (Int32(1), (Float64(o.α), Float64(o.fₗ), Float64(o.τₑ), Float64(o.τₚ), Float64(o.τ₀)))

function gray_desc(as::GrayAtmosphericState, stefan)
    nlay, ncol = size(as.p_lay)
    kind, otp = otp_fields(as.otp)
    GrayStateDesc(0, kind, ncol, nlay, ptr(as.lat), ptr(as.p_lay), ptr(as.p_lev), ptr(as.t_lay), ptr(as.t_lev),
                  ptr(as.t_sfc), otp, Float64(stefan))
end

function rte_lw_2stream_solve!(dev::HIPDevice, flux_lw::FluxLW, src_lw::SourceLW2Str, bcs_lw::LwBCs, op::TwoStream,
                               as::GrayAtmosphericState)
    FT = eltype(flux_lw.flux_up)
    nlay, ncol = size(as.p_lay)
    GC.@preserve as bcs_lw flux_lw check(ccall(
        (:rrtmgp_hip_rte_lw_2stream_solve_gray, libhip[]), Cint,
        (P, Ref{GrayStateDesc}, Ref{LwBcsDesc}, Ref{FluxOutDesc}, Ref{SolveOpts}),
        workspace(dev, ncol, nlay, FT), gray_desc(as, RP.Stefan(src_lw.param_set)),
        LwBcsDesc(0, 0, ptr(bcs_lw.sfc_emis), ptr(bcs_lw.inc_flux)), flux_desc(flux_lw, band_flux), opts()))
    return nothing
end

function rte_lw_noscat_solve!(dev::HIPDevice, flux_lw::FluxLW, src_lw::SourceLWNoScat, bcs_lw::LwBCs, op::OneScalar,
                              angle_disc::AngularDiscretization, as::GrayAtmosphericState)
    FT = eltype(flux_lw.flux_up)
    nlay, ncol = size(as.p_lay)
    GC.@preserve as bcs_lw flux_lw check(ccall(
        (:rrtmgp_hip_rte_lw_noscat_solve_gray, libhip[]), Cint,
        (P, Ref{GrayStateDesc}, Ref{LwBcsDesc}, Ref{FluxOutDesc}, Ref{SolveOpts}),
        workspace(dev, ncol, nlay, FT), gray_desc(as, RP.Stefan(src_lw.param_set)),
        LwBcsDesc(0, 0, ptr(bcs_lw.sfc_emis), ptr(bcs_lw.inc_flux)), flux_desc(flux_lw, band_flux), opts()))
    return nothing
end

function rte_sw_2stream_solve!(dev::HIPDevice, flux_sw::FluxSW, op::TwoStream, bcs_sw::SwBCs, src_sw::SourceSW2Str,
                               as::GrayAtmosphericState)
    FT = eltype(flux_sw.flux_up)
    nlay, ncol = size(as.p_lay)
    GC.@preserve as bcs_sw flux_sw check(ccall(
        (:rrtmgp_hip_rte_sw_2stream_solve_gray, libhip[]), Cint,
        (P, Ref{GrayStateDesc}, Ref{SwBcsDesc}, Ref{FluxOutDesc}, Ref{SolveOpts}),
        workspace(dev, ncol, nlay, FT), gray_desc(as, 0.0),
        SwBcsDesc(0, 0, ptr(bcs_sw.cos_zenith), ptr(bcs_sw.toa_flux), ptr(bcs_sw.sfc_alb_direct),
                  ptr(bcs_sw.sfc_alb_diffuse)), flux_desc(flux_sw, band_flux), opts()))
    return nothing
end

function rte_sw_noscat_solve!(dev::HIPDevice, flux_sw::FluxSW, op::OneScalar, bcs_sw::SwBCs, as::GrayAtmosphericState)
    FT = eltype(flux_sw.flux_up)
    nlay, ncol = size(as.p_lay)
    GC.@preserve as bcs_sw flux_sw check(ccall(
        (:rrtmgp_hip_rte_sw_noscat_solve_gray, libhip[]), Cint,
        (P, Ref{GrayStateDesc}, Ref{SwBcsDesc}, Ref{FluxOutDesc}, Ref{SolveOpts}),
        workspace(dev, ncol, nlay, FT), gray_desc(as, 0.0),
        SwBcsDesc(0, 0, ptr(bcs_sw.cos_zenith), ptr(bcs_sw.toa_flux), C_NULL, C_NULL), flux_desc(flux_sw), opts()))
    return nothing
end

# ---- state preparation ----------------------------------------------------------------------------
params_desc(ps) = ParamsDesc(RP.grav(ps), RP.molmass_dryair(ps), RP.molmass_water(ps), RP.gas_constant(ps),
                             RP.kappa_d(ps), RP.Stefan(ps), RP.avogad(ps))

function compute_col_gas!(dev::HIPDevice, p_lev::AbstractArray{FT, 2}, col_dry::AbstractArray{FT, 2}, param_set::RP.ARP,
                          vmr_h2o::Union{AbstractArray{FT, 2}, Nothing} = nothing,
                          lat::Union{AbstractArray{FT, 1}, Nothing} = nothing) where {FT}
    nlay, ncol = size(col_dry)
    # col_dry / vmr_h2o may be strided views of layerdata / Vmr: the C ABI wants dense (nlay, ncol) slabs
    cd = col_dry isa Array ? col_dry : Array(col_dry)
    h2o = vmr_h2o === nothing || vmr_h2o isa Array ? vmr_h2o : Array(vmr_h2o)
    GC.@preserve p_lev cd h2o lat check(ccall(
        (:rrtmgp_hip_compute_col_gas, libhip[]), Cint, (P, Int32, P, P, Ref{ParamsDesc}, P, P),
        workspace(dev, ncol, nlay, FT), 0, ptr(p_lev), ptr(cd), params_desc(param_set), ptr(h2o), ptr(lat)))
    cd === col_dry || copyto!(col_dry, cd)
    return nothing
end

function compute_relative_humidity!(dev::HIPDevice, rh::AbstractArray{FT, 2}, p_lay::AbstractArray{FT, 2},
                                    t_lay::AbstractArray{FT, 2}, param_set::RP.ARP,
                                    vmr_h2o::AbstractArray{FT, 2}) where {FT}
    nlay, ncol = size(p_lay)
    dense(a) = a isa Array ? a : Array(a)
    r, p, t, h = dense(rh), dense(p_lay), dense(t_lay), dense(vmr_h2o)
    GC.@preserve r p t h check(ccall(
        (:rrtmgp_hip_compute_relative_humidity, libhip[]), Cint, (P, Int32, P, P, P, Ref{ParamsDesc}, P),
        workspace(dev, ncol, nlay, FT), 0, ptr(r), ptr(p), ptr(t), params_desc(param_set), ptr(h)))
    r === rh || copyto!(rh, r)
    return nothing
end

# prepare_atmosphere! (src/api/update_fluxes.jl:252-281) as one launch instead of the broadcast
# cascade of src/api/grid_adaptation.jl.  Optional: with `array_type(::HIPDevice) = Array` the
# reference's own host cascade also works; a host model calls this instead to save those passes.
const INTERP_CODE = Dict(:NoInterpolation => 0, :ArithmeticMean => 1, :GeometricMean => 2, :UniformZ => 3,
                         :UniformP => 4, :BestFit => 5)
const BOTTOM_CODE = Dict(:SameAsInterpolation => 0, :UseSurfaceTempAtBottom => 1, :HydrostaticBottom => 2)

function hip_prepare_atmosphere!(dev::HIPDevice, as::AtmosphericState, param_set::RP.ARP, lookup_lw::LookUpLW;
                                 interpolation = RRTMGP.NoInterpolation(),
                                 bottom_extrapolation = RRTMGP.SameAsInterpolation(),
                                 isothermal_boundary_layer::Bool = false, center_z = nothing, face_z = nothing)
    FT = eltype(as.p_lev)
    nlay, ncol = size(as.layerdata, 2), size(as.layerdata, 3)
    icode = INTERP_CODE[nameof(typeof(interpolation))]
    steps = icode == 0 ? Int32(14) : Int32(15)  # RRTMGP_PREP_ALL without / with INTERPOLATE
    o = PrepareOpts(steps, icode, BOTTOM_CODE[nameof(typeof(bottom_extrapolation))], isothermal_boundary_layer, 0,
                    lookup_lw.idx_h2o, ptr(center_z), ptr(face_z), lookup_lw.p_ref_min, lookup_lw.t_ref_min,
                    lookup_lw.t_ref_max)
    GC.@preserve as center_z face_z check(ccall(
        (:rrtmgp_hip_prepare_atmosphere, libhip[]), Cint, (P, Ref{AtmosStateDesc}, Ref{ParamsDesc}, Ref{PrepareOpts}),
        workspace(dev, ncol, nlay, FT), state_desc(as, !isnothing(as.cloud_state), !isnothing(as.aerosol_state)),
        params_desc(param_set), o))
    return nothing
end

# update_profile_lw!, compute_gray_heating_rate! and setup_gray_as_pr_grid! (K11-K13 of the CUDA
# extension) are host-array loops of a few flops per column, used by the gray test driver only:
# with `array_type(::HIPDevice) = Array` they simply run the reference's CPU methods.
import RRTMGP.GrayAtmosphere: update_profile_lw!, compute_gray_heating_rate!
import RRTMGP.AtmosphericStates: setup_gray_as_pr_grid!
update_profile_lw!(::HIPDevice, args...) = update_profile_lw!(ClimaComms.CPUSingleThreaded(), args...)
compute_gray_heating_rate!(::HIPDevice, args...) = compute_gray_heating_rate!(ClimaComms.CPUSingleThreaded(), args...)
setup_gray_as_pr_grid!(::HIPDevice, ncol, args...) = setup_gray_as_pr_grid!(ClimaComms.CPUSingleThreaded(), ncol, args...)

end # module
