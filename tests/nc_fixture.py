"""Writers of classic-NetCDF files in the rrtmgp-data v1.9 schema, filled from the
synthetic lookups (test infrastructure).  Variable names, dimension order and the
file-order minor-gas layout follow what ext/lookup_constructors.jl reads; the inverse
transforms here are written independently of rrtmgp_jl_amd.netcdf_io so the round trip
file -> lookup is a real check of the reader."""
import numpy as np
from scipy.io import netcdf_file

from rrtmgp_jl_amd.synthetic import GAS_NAMES

STRLEN = 32


def _put(nc, name, jl_arr, jl_dims, typ=None):
    """Store a Julia-shaped array: file dims are the reversed Julia dims."""
    a = np.asarray(jl_arr)
    for d, n in zip(jl_dims, a.shape):
        if d not in nc.dimensions:
            nc.createDimension(d, n)
    if typ is None:
        typ = "i" if a.dtype.kind in "iu" else "d"
    v = nc.createVariable(name, typ, tuple(reversed(jl_dims)))
    v[...] = np.ascontiguousarray(a.T)
    return v


def _put_scalar(nc, name, val, typ="d"):
    v = nc.createVariable(name, typ, ())
    v.data[()] = val   # assignValue() breaks on 0-d arrays with numpy 2


def _put_strings(nc, name, strs, dim):
    if "string_len" not in nc.dimensions:
        nc.createDimension("string_len", STRLEN)
    if dim not in nc.dimensions:
        nc.createDimension(dim, len(strs))
    v = nc.createVariable(name, "c", (dim, "string_len"))
    for i, s in enumerate(strs):
        v[i, :] = np.frombuffer(s.ljust(STRLEN).encode(), dtype="S1")


def _bnd_lims(gpt2bnd):
    n_bnd = int(gpt2bnd.max())
    lims = np.zeros((2, n_bnd), dtype=np.int64)
    for ib in range(n_bnd):
        w = np.nonzero(gpt2bnd == ib + 1)[0]
        lims[:, ib] = (w[0] + 1, w[-1] + 1)
    return lims


def _minor_file_form(minor, bnd_lims_gpt):
    """Invert the per-g-point re-ordering: returns (limits (2, n_min), kminor in file order)."""
    n_bnd = bnd_lims_gpt.shape[1]
    n_min = minor.gasdata.shape[1]
    lims = np.zeros((2, n_min), dtype=np.int64)
    for ib in range(n_bnd):
        for i in range(minor.bnd_st[ib] - 1, minor.bnd_st[ib + 1] - 1):
            lims[:, i] = bnd_lims_gpt[:, ib]
    start = np.concatenate([[0], np.cumsum(lims[1] - lims[0] + 1)])[:-1]   # file offset of each interval
    kfile = np.zeros_like(minor.kminor)
    for ib in range(n_bnd):
        for loc, igpt in enumerate(range(bnd_lims_gpt[0, ib], bnd_lims_gpt[1, ib] + 1)):
            for j, i in enumerate(range(minor.bnd_st[ib] - 1, minor.bnd_st[ib + 1] - 1)):
                kfile[:, :, start[i] + loc] = minor.kminor[:, :, minor.gpt_st[igpt - 1] - 1 + j]
    return lims, kfile


def write_gas_file(path, lk, seed=0):
    """`lk` is a float64 synthetic GasLookup.  Returns the (quiet, facular, sunspot, mg, sb)
    solar terms written for SW files (None for LW)."""
    rng = np.random.default_rng(seed)
    names = [""] * (lk.n_gases - 1)
    from rrtmgp_jl_amd.synthetic import IDX_GASES
    for g, i in IDX_GASES.items():
        names[i - 1] = g
    nc = netcdf_file(path, "w")
    lims_gpt = _bnd_lims(lk.major_gpt2bnd)
    _put_strings(nc, "gas_names", names, "absorber")
    ks = lk.key_species.copy()
    both2 = (ks[0] == 2) & (ks[1] == 2)
    ks[0][both2] = 0
    ks[1][both2] = 0
    _put(nc, "key_species", ks, ("pair", "atmos_layer", "bnd"))
    _put(nc, "kmajor", np.transpose(lk.kmajor, (3, 0, 1, 2)), ("gpt", "mixing_fraction", "pressure_interp", "temperature"))
    _put(nc, "bnd_limits_gpt", lims_gpt, ("pair", "bnd"))
    _put(nc, "bnd_limits_wavenumber", lk.bnd_lims_wn, ("pair", "bnd"))
    _put(nc, "press_ref", np.exp(lk.ln_p_ref), ("pressure",))
    _put(nc, "temp_ref", lk.t_ref, ("temperature",))
    _put_scalar(nc, "press_ref_trop", lk.p_ref_tropo)
    _put_scalar(nc, "absorption_coefficient_ref_T", 296.0)
    _put_scalar(nc, "absorption_coefficient_ref_P", 101325.0)
    _put(nc, "vmr_ref", lk.vmr_ref, ("atmos_layer", "absorber_ext", "temperature"))
    for reg, minor in (("lower", lk.minor_lower), ("upper", lk.minor_upper)):
        lims, kfile = _minor_file_form(minor, lims_gpt)
        idim = f"minor_absorber_intervals_{reg}"
        _put(nc, f"minor_limits_gpt_{reg}", lims, ("pair", idim))
        _put(nc, f"kminor_{reg}", np.transpose(kfile, (2, 0, 1)), (f"contributors_{reg}", "mixing_fraction", "temperature"))
        _put_strings(nc, f"minor_gases_{reg}", [names[g - 1] for g in minor.gasdata[0]], idim)
        _put_strings(nc, f"scaling_gas_{reg}", [names[g - 1] if g > 0 else "" for g in minor.gasdata[1]], idim)
        _put(nc, f"minor_scales_with_density_{reg}", minor.gasdata[2], (idim,))
        _put(nc, f"scale_by_complement_{reg}", minor.gasdata[3], (idim,))
        _put(nc, f"kminor_start_{reg}", np.arange(1, lims.shape[1] + 1), (idim,))
    solar = None
    if lk.is_sw:
        _put(nc, "rayl_lower", np.transpose(lk.rayl_lower, (2, 0, 1)), ("gpt", "mixing_fraction", "temperature"))
        _put(nc, "rayl_upper", np.transpose(lk.rayl_upper, (2, 0, 1)), ("gpt", "mixing_fraction", "temperature"))
        fac = rng.uniform(0.0, 2.0, lk.n_gpt)
        spot = rng.uniform(-1.0, 1.0, lk.n_gpt)
        mg, sb = 0.1567652, 902.71260 * 1e-6
        quiet = lk.solar_src_scaled * lk.solar_src_tot
        _put(nc, "solar_source_quiet", quiet, ("gpt",))
        _put(nc, "solar_source_facular", fac, ("gpt",))
        _put(nc, "solar_source_sunspot", spot, ("gpt",))
        _put_scalar(nc, "mg_default", mg)
        _put_scalar(nc, "sb_default", sb)
        solar = (quiet, fac, spot, mg, sb)
    else:
        _put(nc, "plank_fraction", np.transpose(lk.planck_fraction, (3, 0, 1, 2)),
             ("gpt", "mixing_fraction", "pressure_interp", "temperature"))
        _put(nc, "temperature_Planck", lk.t_planck, ("temperature_Planck",))
        _put(nc, "totplnk", lk.tot_planck, ("temperature_Planck", "bnd"))
    nc.close()
    return solar


def write_cloud_file(path, lk, band_wn):
    nc = netcdf_file(path, "w")
    nband, nrgh, nl, ni = (int(x) for x in lk.dims[:4])
    for k, name in enumerate(("ext", "ssa", "asy")):
        _put(nc, f"{name}liq", lk.liqdata[k * nl:(k + 1) * nl], ("nsize_liq", "nband"))
        _put(nc, f"{name}ice", lk.icedata[k * ni:(k + 1) * ni], ("nsize_ice", "nband", "nrghice"))
    _put(nc, "bnd_limits_wavenumber", band_wn, ("pair", "nband"))
    _put_scalar(nc, "radliq_lwr", lk.bounds[0])
    _put_scalar(nc, "radliq_upr", lk.bounds[1])
    _put_scalar(nc, "diamice_lwr", lk.bounds[2] * 2)
    _put_scalar(nc, "diamice_upr", lk.bounds[3] * 2)
    nc.close()


def write_aerosol_file(path, lk, band_wn):
    nc = netcdf_file(path, "w")
    _put(nc, "merra_aero_bin_lims", lk.size_bin_limits, ("pair", "nbin"))
    _put(nc, "aero_rh", lk.rh_levels, ("nrh",))
    _put(nc, "aero_dust_tbl", lk.dust, ("nval", "nbin", "nband"))
    _put(nc, "aero_salt_tbl", lk.sea_salt, ("nval", "nrh", "nbin", "nband"))
    _put(nc, "aero_sulf_tbl", lk.sulfate, ("nval", "nrh", "nband"))
    _put(nc, "aero_bcar_rh_tbl", lk.black_carbon_rh, ("nval", "nrh", "nband"))
    _put(nc, "aero_bcar_tbl", lk.black_carbon, ("nval", "nband"))
    _put(nc, "aero_ocar_rh_tbl", lk.organic_carbon_rh, ("nval", "nrh", "nband"))
    _put(nc, "aero_ocar_tbl", lk.organic_carbon, ("nval", "nband"))
    _put(nc, "bnd_limits_wavenumber", band_wn, ("pair", "nband"))
    nc.close()


# ---- example inputs / reference fluxes (rrtmgp-data examples/ layout) ---------------------
def write_rfmip_input(path, p_lev, p_lay, t_lev, t_lay, t_sfc, h2o, o3, gm, emis, alb, sza_deg, tsi, lat, lon,
                      n_expt=2, top_first=True):
    """Multi-experiment RFMIP-style input.  Arrays are (nlev|nlay, nsite) bottom-first; experiment e
    adds e Kelvin / scales e-dependent so the `expt_no` slicing is observable.  `gm` maps RFMIP
    variable name -> (value, units string)."""
    o = slice(None, None, -1) if top_first else slice(None)
    nc = netcdf_file(path, "w")
    ex = np.arange(n_expt, dtype=np.float64)
    _put(nc, "pres_level", p_lev[o], ("level", "site"))
    _put(nc, "pres_layer", p_lay[o], ("layer", "site"))
    _put(nc, "temp_level", t_lev[o][:, :, None] + ex, ("level", "site", "expt"))
    _put(nc, "temp_layer", t_lay[o][:, :, None] + ex, ("layer", "site", "expt"))
    _put(nc, "surface_temperature", t_sfc[:, None] + ex, ("site", "expt"))
    _put(nc, "water_vapor", h2o[o][:, :, None] * (1 + 0.1 * ex), ("layer", "site", "expt"))
    _put(nc, "ozone", o3[o][:, :, None] * (1 + 0.05 * ex), ("layer", "site", "expt"))
    _put(nc, "surface_emissivity", emis, ("site",))
    _put(nc, "surface_albedo", alb, ("site",))
    _put(nc, "solar_zenith_angle", sza_deg, ("site",))
    _put(nc, "total_solar_irradiance", tsi, ("site",))
    _put(nc, "lat", lat, ("site",))
    _put(nc, "lon", lon, ("site",))
    for var, (val, units) in gm.items():
        v = _put(nc, var, val * (1 + 0.01 * ex), ("expt",))
        v.units = units
    nc.close()


def write_rfmip_flux(path, var, flux, top_first=True):
    """(nlev, nsite, nexpt) bottom-first flux written the way the RFMIP reference files store it."""
    nc = netcdf_file(path, "w")
    _put(nc, var, flux[::-1] if top_first else flux, ("level", "site", "expt"))
    nc.close()


def write_allsky_input(path, p_lev, p_lay, t_lev, t_lay, h2o, o3, aero=None, top_first=True):
    """examples/all-sky input: Julia shape (col, lev).  `aero` = (type, size, mass) each (nlay, ncol)."""
    o = slice(None, None, -1) if top_first else slice(None)
    nc = netcdf_file(path, "w")
    for name, a, d in (("p_lev", p_lev, "lev"), ("t_lev", t_lev, "lev"), ("p_lay", p_lay, "lay"),
                       ("t_lay", t_lay, "lay"), ("h2o", h2o, "lay"), ("o3", o3, "lay")):
        _put(nc, name, a[o].T, ("col", d))
    if aero is not None:
        _put(nc, "aero_type", aero[0][o].T.astype(np.int32), ("col", "lay"))
        _put(nc, "aero_size", aero[1][o].T, ("col", "lay"))
        _put(nc, "aero_mass", aero[2][o].T, ("col", "lay"))
    nc.close()


def write_allsky_flux(path, lam, up, dn, top_first=True):
    """(nlev, ncol) bottom-first fluxes -> `{lam}_flux_up/dn` with Julia shape (col, lev)."""
    o = slice(None, None, -1) if top_first else slice(None)
    nc = netcdf_file(path, "w")
    _put(nc, f"{lam}_flux_up", up[o].T, ("col", "lev"))
    _put(nc, f"{lam}_flux_dn", dn[o].T, ("col", "lev"))
    nc.close()
