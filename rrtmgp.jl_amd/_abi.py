"""ctypes mirror of include/rrtmgp_hip.h (the C-ABI structs and constants).

Field order and types must match the header exactly; tests/test_abi.py checks the
struct sizes against the values compiled into the library.
"""
import ctypes as C

import numpy as np

F32 = 4
F64 = 8
MEM_HOST = 0
MEM_DEVICE = 1
LAYOUT_NCOL_NLEV = 0
LAYOUT_NLEV_NCOL = 1
VMR_GM = 0
VMR_FULL = 1
N_AEROSOLS = 15

OK = 0
ERRORS = {-1: "EINVAL", -2: "ENODEV", -3: "EHIP", -4: "ENOMEM", -5: "EUNSUPPORTED"}

i32, i64, f64, vp = C.c_int32, C.c_int64, C.c_double, C.c_void_p


class MinorDesc(C.Structure):
    _fields_ = [("n_min_absrb", i64), ("n_contrib", i64), ("bnd_st", vp), ("gpt_st", vp), ("gasdata", vp),
                ("kminor", vp)]


class GasLookupDesc(C.Structure):
    _fields_ = [("ftype", i32), ("is_sw", i32), ("n_gpt", i64), ("n_bnd", i64), ("n_eta", i64), ("n_p_ref", i64),
                ("n_t_ref", i64), ("n_gases", i64), ("n_t_plnk", i64), ("idx_h2o", i64), ("p_ref_tropo", f64),
                ("p_ref_min", f64), ("t_ref_min", f64), ("t_ref_max", f64), ("solar_src_tot", f64),
                ("key_species", vp), ("major_gpt2bnd", vp), ("kmajor", vp), ("planck_fraction", vp),
                ("t_planck", vp), ("tot_planck", vp), ("ln_p_ref", vp), ("t_ref", vp), ("vmr_ref", vp),
                ("minor_lower", MinorDesc), ("minor_upper", MinorDesc), ("rayl_lower", vp), ("rayl_upper", vp),
                ("solar_src_scaled", vp)]


class CloudLookupDesc(C.Structure):
    _fields_ = [("ftype", i32), ("_pad", i32), ("nband", i64), ("nrghice", i64), ("nsize_liq", i64),
                ("nsize_ice", i64), ("bounds", vp), ("liqdata", vp), ("icedata", vp)]


class AerosolLookupDesc(C.Structure):
    _fields_ = [("ftype", i32), ("_pad", i32), ("nband", i64), ("nbin", i64), ("nrh", i64), ("iband_550nm", i64),
                ("size_bin_limits", vp), ("rh_levels", vp), ("dust", vp), ("sea_salt", vp), ("sulfate", vp),
                ("black_carbon_rh", vp), ("black_carbon", vp), ("organic_carbon_rh", vp), ("organic_carbon", vp)]


class AtmosState(C.Structure):
    _fields_ = [("mem", i32), ("vmr_kind", i32), ("ncol", i64), ("nlay", i64), ("ngas", i64), ("layerdata", vp),
                ("p_lev", vp), ("t_lev", vp), ("t_sfc", vp), ("lat", vp), ("vmr_h2o", vp), ("vmr_o3", vp),
                ("vmr", vp), ("cld_r_eff_liq", vp), ("cld_r_eff_ice", vp), ("cld_path_liq", vp),
                ("cld_path_ice", vp), ("cld_frac", vp), ("cld_cover_lw", vp), ("cld_cover_sw", vp),
                ("ice_rgh", i64), ("aero_size", vp), ("aero_mass", vp), ("aod_sw_ext", vp), ("aod_sw_sca", vp)]


class LwBcs(C.Structure):
    _fields_ = [("mem", i32), ("inc_flux_ld", i32), ("sfc_emis", vp), ("inc_flux", vp)]


class SwBcs(C.Structure):
    _fields_ = [("mem", i32), ("_pad", i32), ("cos_zenith", vp), ("toa_flux", vp), ("sfc_alb_direct", vp),
                ("sfc_alb_diffuse", vp)]


class FluxOut(C.Structure):
    _fields_ = [("mem", i32), ("layout", i32), ("flux_up", vp), ("flux_dn", vp), ("flux_net", vp),
                ("flux_dn_dir", vp), ("band_flux_up", vp), ("band_flux_dn", vp), ("band_flux_net", vp),
                ("band_flux_ncol", i64), ("clear_flux_up", vp), ("clear_flux_dn", vp), ("clear_flux_net", vp), ("clear_flux_dn_dir", vp),
                ("flux_ncol", i64)]


class SolveOpts(C.Structure):
    _fields_ = [("n_gauss_angles", i32), ("metric_mem", i32), ("metric_scaling", vp), ("seed", C.c_uint64),
                ("col_offset", i64)]


class GrayState(C.Structure):
    _fields_ = [("mem", i32), ("otp_kind", i32), ("ncol", i64), ("nlay", i64), ("lat", vp), ("p_lay", vp),
                ("p_lev", vp), ("t_lay", vp), ("t_lev", vp), ("t_sfc", vp), ("otp", f64 * 5), ("stefan", f64)]


class Params(C.Structure):
    _fields_ = [("grav", f64), ("molmass_dryair", f64), ("molmass_water", f64), ("gas_constant", f64),
                ("kappa_d", f64), ("stefan", f64), ("avogad", f64)]


class PrepareOpts(C.Structure):
    _fields_ = [("steps", i32), ("interpolation", i32), ("bottom_extrapolation", i32),
                ("isothermal_boundary_layer", i32), ("z_mem", i32), ("idx_h2o", i32), ("center_z", vp),
                ("face_z", vp), ("p_min", f64), ("t_min", f64), ("t_max", f64)]


class View2D(C.Structure):
    """rrtmgp_view2d: element (i, j) at ptr[i * stride0 + j * stride1], strides in elements (Julia's `strides`)."""
    _fields_ = [("ptr", vp), ("stride0", i64), ("stride1", i64)]


class UpdateFluxesArgs(C.Structure):
    """rrtmgp_update_fluxes_args: the whole radiation step (update_fluxes!, src/api/update_fluxes.jl:223-233)."""
    _fields_ = [("lookup_lw", vp), ("lookup_sw", vp), ("lookup_lw_cld", vp), ("lookup_sw_cld", vp), ("lookup_lw_aero", vp),
                ("lookup_sw_aero", vp), ("as_", C.POINTER(AtmosState)), ("bcs_lw", C.POINTER(LwBcs)), ("bcs_sw", C.POINTER(SwBcs)),
                ("flux_lw", C.POINTER(FluxOut)), ("flux_sw", C.POINTER(FluxOut)), ("net_flux", vp), ("clear_net_flux", vp),
                ("params", C.POINTER(Params)), ("prepare", C.POINTER(PrepareOpts)), ("opts", C.POINTER(SolveOpts)),
                ("lw_solver", i32), ("_pad", i32)]


class UpdateFluxesGrayArgs(C.Structure):
    """rrtmgp_update_fluxes_gray_args: update_fluxes! for GrayRadiation in one call."""
    _fields_ = [("as_", C.POINTER(GrayState)), ("bcs_lw", C.POINTER(LwBcs)), ("bcs_sw", C.POINTER(SwBcs)),
                ("flux_lw", C.POINTER(FluxOut)), ("flux_sw", C.POINTER(FluxOut)), ("net_flux", vp),
                ("params", C.POINTER(Params)), ("prepare", C.POINTER(PrepareOpts)), ("opts", C.POINTER(SolveOpts)),
                ("lw_solver", i32), ("sw_twostream", i32)]


LW_TWOSTREAM, LW_NOSCAT = 1, 0
PREP_INTERPOLATE, PREP_ISOTHERMAL, PREP_CLIP, PREP_COL_DRY, PREP_ALL = 1, 2, 4, 8, 15
PREP_REL_HUM = 16   # optional extra step: relative humidity refreshed in the same launch
INTERP = {"none": 0, "arithmetic_mean": 1, "geometric_mean": 2, "uniform_z": 3, "uniform_p": 4, "best_fit": 5}
BOTTOM = {"same_as_interpolation": 0, "use_surface_temp_at_bottom": 1, "hydrostatic_bottom": 2}


def ftype_of(dtype) -> int:
    dtype = np.dtype(dtype)
    if dtype == np.float32:
        return F32
    if dtype == np.float64:
        return F64
    raise TypeError(f"FT must be float32 or float64, got {dtype}")


def dtype_of(ftype: int):
    return {F32: np.float32, F64: np.float64}[ftype]


def fptr(a, dtype=None):
    """Host pointer of a numpy array that must be Fortran-contiguous (Julia layout), or None."""
    if a is None:
        return None
    if not isinstance(a, np.ndarray):
        raise TypeError(f"expected numpy array, got {type(a)}")
    if dtype is not None and a.dtype != np.dtype(dtype):
        raise TypeError(f"expected dtype {np.dtype(dtype)}, got {a.dtype}")
    if a.ndim > 1 and not a.flags.f_contiguous:
        raise ValueError("array must be column-major (Fortran) contiguous, as the reference stores it")
    if a.ndim == 1 and not a.flags.c_contiguous:
        raise ValueError("1-D array must be contiguous")
    return a.ctypes.data
