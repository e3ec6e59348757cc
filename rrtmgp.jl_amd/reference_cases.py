"""The reference's three data-driven parity cases, set up from rrtmgp-data's example files.

Host-side mirror of `test/read_clear_sky.jl`, `test/read_cloudy_sky.jl`,
`test/read_all_sky_with_aerosols.jl` and the comparison halves of
`test/clear_sky_utils.jl:164-258`, `cloudy_sky_utils.jl`, `all_sky_with_aerosols_utils.jl`:
the same inputs, orientation flip, column replication, gas lists and tolerances, driving
this package's solvers through the C ABI.  rrtmgp-data v1.9 is not reachable from the
build image, so these drivers are exercised on schema-faithful synthetic files
(tests/test_reference_cases.py); `tools/run_reference_parity.py` runs them on the real
files and is the step that turns "parity unpinned" into the reference's CI tolerances.
"""
from __future__ import annotations

import os
from dataclasses import dataclass
from typing import Callable, Dict, Optional

import numpy as np

from .netcdf_io import Dataset
from .states import AerosolState, AtmosphericState, CloudState, LwBCs, SwBCs, Vmr, VmrGM

# test/clear_sky.jl:7-9, test/cloudy_sky.jl:6-8, test/all_sky_with_aerosols.jl:6-8  [W/m2]
TOLERANCES = {
    "clear_sky": {"lw_noscat": {np.float64: 1e-4, np.float32: 5e-3},
                  "lw_2stream": {np.float64: 4.5, np.float32: 4.5},
                  "sw": {np.float64: 1e-3, np.float32: 0.04}},
    "cloudy_sky": {"lw_noscat": {np.float64: 1e-5, np.float32: 5e-3},
                   "lw_2stream": {np.float64: 5.0, np.float32: 5.0},
                   "sw": {np.float64: 1e-5, np.float32: 0.06}},
    "all_sky_with_aerosols": {"lw_noscat": {np.float64: 1e-5, np.float32: 5e-3},
                              "lw_2stream": {np.float64: 5.0, np.float32: 5.0},
                              "sw": {np.float64: 1e-5, np.float32: 0.06}},
}

# test/reference_files.jl:24-61, relative to the rrtmgp-data root
REFERENCE_FILES = {
    ("gas", "lw", "up"): "examples/rfmip-clear-sky/reference/rlu_Efx_RTE-RRTMGP-181204_rad-irf_r1i1p1f1_gn.nc",
    ("gas", "lw", "dn"): "examples/rfmip-clear-sky/reference/rld_Efx_RTE-RRTMGP-181204_rad-irf_r1i1p1f1_gn.nc",
    ("gas", "sw", "up"): "examples/rfmip-clear-sky/reference/rsu_Efx_RTE-RRTMGP-181204_rad-irf_r1i1p1f1_gn.nc",
    ("gas", "sw", "dn"): "examples/rfmip-clear-sky/reference/rsd_Efx_RTE-RRTMGP-181204_rad-irf_r1i1p1f1_gn.nc",
    ("gas_clouds", "lw"): "examples/all-sky/reference/rrtmgp-allsky-lw-no-aerosols.nc",
    ("gas_clouds", "sw"): "examples/all-sky/reference/rrtmgp-allsky-sw-no-aerosols.nc",
    ("gas_clouds_aerosols", "lw"): "examples/all-sky/reference/rrtmgp-allsky-lw.nc",
    ("gas_clouds_aerosols", "sw"): "examples/all-sky/reference/rrtmgp-allsky-sw.nc",
}

# read_clear_sky.jl:93-151: lookup gas name -> RFMIP variable (the `cf4` units are read from
# `hfc23_GM` there, :149-151; both are 1e-12 in the RFMIP file so it does not matter)
RFMIP_GM = {"co2": "carbon_dioxide_GM", "n2o": "nitrous_oxide_GM", "co": "carbon_monoxide_GM",
            "ch4": "methane_GM", "o2": "oxygen_GM", "n2": "nitrogen_GM", "ccl4": "carbon_tetrachloride_GM",
            "cfc11": "cfc11_GM", "cfc12": "cfc12_GM", "cfc22": "hcfc22_GM", "hfc143a": "hfc143a_GM",
            "hfc125": "hfc125_GM", "hfc23": "hfc23_GM", "hfc32": "hfc32_GM", "hfc134a": "hfc134a_GM",
            "cf4": "cf4_GM"}


@dataclass
class Case:
    as_: AtmosphericState
    bcs_lw: LwBCs
    bcs_sw: SwBCs
    bot_at_1: bool


def _tile(a, ncol):
    """`repeat(a, 1, nrepeat)[:, 1:ncol]` along the last (column) axis."""
    n = a.shape[-1]
    rep = -(-ncol // n)
    return np.asfortranarray(np.tile(a, (1,) * (a.ndim - 1) + (rep,))[..., :ncol])


def _layerdata(col_dry, p_lay, t_lay, rel_hum):
    ld = np.empty((4,) + p_lay.shape, dtype=p_lay.dtype, order="F")
    ld[0], ld[1], ld[2], ld[3] = col_dry, p_lay, t_lay, rel_hum
    return ld


def setup_clear_sky_as(ds_in: Dataset, idx_gases: Dict[str, int], expt_no: int, lookup_lw, ncol: int, FT,
                       col_gas: Callable, rel_hum_fn: Callable, vmr_type=VmrGM) -> Case:
    """setup_clear_sky_as (test/read_clear_sky.jl:7-193).  `expt_no` is 1-based.
    `col_gas(p_lev, vmr_h2o)` / `rel_hum_fn(p_lay, t_lay, vmr_h2o)` are the device routines."""
    FT = np.dtype(FT).type
    e = expt_no - 1
    nlay = ds_in.dim("layer")
    nlev = nlay + 1
    nbnd_lw = lookup_lw.n_bnd
    sfc_emis = _tile(np.repeat(ds_in.raw("surface_emissivity").astype(FT)[None, :], nbnd_lw, 0), ncol)
    sfc_alb = _tile(np.repeat(ds_in.raw("surface_albedo").astype(FT)[None, :], nbnd_lw, 0), ncol)
    zenith = (FT(np.pi) / FT(180)) * ds_in.raw("solar_zenith_angle").astype(FT)
    cos_zenith = _tile(np.cos(zenith), ncol)
    irrad = _tile(ds_in.raw("total_solar_irradiance").astype(FT), ncol)

    p_lev = ds_in.jl("pres_level").astype(np.float64)        # (level, site)
    bot_at_1 = bool(p_lev[0, 0] > p_lev[-1, 0])
    lev = slice(None) if bot_at_1 else slice(None, None, -1)
    top = nlev - 1 if bot_at_1 else 0
    p_lev[top, :] = lookup_lw.p_ref_min                        # :66
    p_lev = p_lev[lev]
    p_lay = ds_in.jl("pres_layer")[lev]
    t_lev = ds_in.jl("temp_level")[lev][:, :, e]
    t_lay = ds_in.jl("temp_layer")[lev][:, :, e]
    p_lev, p_lay, t_lev, t_lay = (_tile(x.astype(FT), ncol) for x in (p_lev, p_lay, t_lev, t_lay))
    t_sfc = _tile(ds_in.jl("surface_temperature")[:, e].astype(FT), ncol)
    vmr_h2o = _tile(ds_in.jl("water_vapor")[lev][:, :, e].astype(FT), ncol)
    vmr_o3 = _tile(ds_in.jl("ozone")[lev][:, :, e].astype(FT), ncol)
    vmrat = np.zeros(lookup_lw.n_gases - 1, dtype=FT)
    for gas, var in RFMIP_GM.items():
        vmrat[idx_gases[gas] - 1] = FT(ds_in.raw(var)[e]) * FT(float(ds_in.attr(var, "units")))
    col_dry = col_gas(p_lev, vmr_h2o)
    rel_hum = rel_hum_fn(p_lay, t_lay, vmr_h2o)
    if vmr_type is VmrGM:
        vmr = VmrGM(vmr_h2o, vmr_o3, vmrat)
    else:
        full = np.zeros((vmrat.shape[0], nlay, ncol), dtype=FT, order="F")
        full[:] = vmrat[:, None, None]
        full[idx_gases["h2o"] - 1] = vmr_h2o
        full[idx_gases["o3"] - 1] = vmr_o3
        vmr = Vmr(full)
    as_ = AtmosphericState(_layerdata(col_dry, p_lay, t_lay, rel_hum), p_lev, t_lev, t_sfc, vmr)
    return Case(as_, LwBCs(sfc_emis), SwBCs(cos_zenith, irrad, sfc_alb, sfc_alb.copy(order="F")), bot_at_1)


def _all_sky_common(ds_in, idx_gases, lkp_lw, lkp_sw, lkp_lw_cld, cldfrac, ncol, ncol_ds, FT, col_gas, rel_hum_fn):
    FT = np.dtype(FT).type
    nlay = ds_in.dim("lay")
    sfc_emis = np.full((lkp_lw.n_bnd, ncol), FT(0.98), dtype=FT, order="F")
    alb_dir = np.full((lkp_sw.n_bnd, ncol), FT(0.06), dtype=FT, order="F")
    alb_dif = alb_dir.copy(order="F")
    cos_zenith = np.full(ncol, FT(0.86), dtype=FT)
    irrad = np.full(ncol, FT(lkp_sw.solar_src_tot), dtype=FT)

    p_lev1 = ds_in.jl("p_lev")[0, :].astype(FT)                # first column only, (col, lev) in Julia
    bot_at_1 = bool(p_lev1[0] > p_lev1[-1])
    o = slice(None) if bot_at_1 else slice(None, None, -1)

    def col1(name):
        return np.asfortranarray(np.repeat(ds_in.jl(name)[0, :][o].astype(FT)[:, None], ncol, 1))
    p_lev, p_lay, t_lev, t_lay = col1("p_lev"), col1("p_lay"), col1("t_lev"), col1("t_lay")
    t_sfc = np.full(ncol, t_lev[0, 0], dtype=FT)
    ngas = lkp_lw.n_gases - 1
    vm = np.zeros((ngas, nlay), dtype=FT)
    vm[idx_gases["h2o"] - 1] = ds_in.jl("h2o")[0, :][o]
    vm[idx_gases["o3"] - 1] = ds_in.jl("o3")[0, :][o]
    for g, v in (("co2", 348e-6), ("ch4", 1650e-9), ("n2o", 306e-9), ("n2", 0.7808), ("o2", 0.2095), ("co", 0.0)):
        vm[idx_gases[g] - 1] = FT(v)
    vmrat = np.asfortranarray(np.repeat(vm[:, :, None], ncol, 2))
    vmr_h2o = np.asfortranarray(vmrat[idx_gases["h2o"] - 1])

    z = lambda: np.zeros((nlay, ncol), dtype=FT, order="F")   # noqa: E731
    cld_frac, reliq, reice, clwp, ciwp = z(), z(), z(), z(), z()
    b = lkp_lw_cld.bounds
    r_liq, r_ice = (b[0] + b[1]) / FT(2), (b[2] + b[3]) / FT(2)
    icol_ds = (np.arange(1, ncol + 1) - 1) % ncol_ds + 1        # read_cloudy_sky.jl:111-113
    in_cloud = (p_lay > FT(10000)) & (p_lay < FT(90000)) & ((icol_ds % 3) != 0)[None, :]
    cld_frac[in_cloud] = cldfrac
    liq = in_cloud & (t_lay > FT(263))
    ice = in_cloud & (t_lay < FT(273))
    clwp[liq], reliq[liq] = FT(10), r_liq
    ciwp[ice], reice[ice] = FT(10), r_ice
    cloud = CloudState(reliq, reice, clwp, ciwp, cld_frac, np.zeros(ncol, FT), np.zeros(ncol, FT), ice_rgh=2)
    col_dry = col_gas(p_lev, vmr_h2o)
    rel_hum = rel_hum_fn(p_lay, t_lay, vmr_h2o)
    as_ = AtmosphericState(_layerdata(col_dry, p_lay, t_lay, rel_hum), p_lev, t_lev, t_sfc, Vmr(vmrat),
                           cloud_state=cloud)
    return Case(as_, LwBCs(sfc_emis), SwBCs(cos_zenith, irrad, alb_dir, alb_dif), bot_at_1), o


def setup_cloudy_sky_as(ds_in, idx_gases, lkp_lw, lkp_sw, lkp_lw_cld, cldfrac, ncol, ncol_ds, FT,
                        col_gas, rel_hum_fn) -> Case:
    """setup_cloudy_sky_as (test/read_cloudy_sky.jl:7-186); `ncol_ds` = columns in the reference
    flux file (its `ncol_ds_all_sky()`), which sets the 2-in-3 cloudy pattern."""
    return _all_sky_common(ds_in, idx_gases, lkp_lw, lkp_sw, lkp_lw_cld, cldfrac, ncol, ncol_ds, FT,
                           col_gas, rel_hum_fn)[0]


def setup_allsky_with_aerosols_as(ds_in, idx_gases, idx_aerosol, idx_aerosize, lkp_lw, lkp_sw, lkp_lw_cld,
                                  cldfrac, ncol, ncol_ds, FT, col_gas, rel_hum_fn) -> Case:
    """setup_allsky_with_aerosols_as (test/read_all_sky_with_aerosols.jl:7-222)."""
    FT = np.dtype(FT).type
    case, o = _all_sky_common(ds_in, idx_gases, lkp_lw, lkp_sw, lkp_lw_cld, cldfrac, ncol, ncol_ds, FT,
                              col_gas, rel_hum_fn)
    a_type = ds_in.jl("aero_type")[:, o].T.astype(np.int64)    # (lay, col_ref)
    a_size = ds_in.jl("aero_size")[:, o].T.astype(FT)
    a_mass = ds_in.jl("aero_mass")[:, o].T.astype(FT)
    nlay, ncol_ref = a_type.shape
    n_aer, n_size = len(idx_aerosol), max(idx_aerosize.values())
    mass = np.zeros((n_aer, nlay, ncol_ref), dtype=FT, order="F")
    size = np.zeros((n_size, nlay, ncol_ref), dtype=FT, order="F")
    lay, col = np.nonzero(a_type > 0)
    t = a_type[lay, col]
    mass[t - 1, lay, col] = a_mass[lay, col]
    sized = np.isin(t, list(idx_aerosize))
    size[t[sized] - 1, lay[sized], col[sized]] = a_size[lay[sized], col[sized]]
    case.as_.aerosol_state = AerosolState(_tile(size, ncol), _tile(mass, ncol), np.zeros(ncol, FT), np.zeros(ncol, FT))
    return case


# ---- comparison data -------------------------------------------------------------------
def _orient(a, bot_at_1):
    return a if bot_at_1 else a[::-1]


def load_clear_sky_comparison(data_root: str, expt_no: int, bot_at_1: bool, ncol: int):
    """load_comparison_data (read_clear_sky.jl:197-242): (nlev, ncol) rlu, rld, rsu, rsd."""
    out = []
    for key, var in ((("gas", "lw", "up"), "rlu"), (("gas", "lw", "dn"), "rld"),
                     (("gas", "sw", "up"), "rsu"), (("gas", "sw", "dn"), "rsd")):
        with Dataset(os.path.join(data_root, REFERENCE_FILES[key])) as ds:
            a = ds.jl(var)[:, :, expt_no - 1]
        out.append(_tile(_orient(np.asarray(a, dtype=np.float64), bot_at_1), ncol))
    return tuple(out)


def load_all_sky_comparison(data_root: str, problem: str, bot_at_1: bool, ncol: int):
    """load_comparison_data (read_cloudy_sky.jl:190-226 / read_all_sky_with_aerosols.jl:226-262);
    `problem` is "gas_clouds" or "gas_clouds_aerosols".  The files store (col, lev) in Julia order."""
    out = []
    for lam in ("lw", "sw"):
        with Dataset(os.path.join(data_root, REFERENCE_FILES[(problem, lam)])) as ds:
            for d in ("up", "dn"):
                a = ds.jl(f"{lam}_flux_{d}").T
                out.append(_tile(_orient(np.asarray(a, dtype=np.float64), bot_at_1), ncol))
    return tuple(out)


def ncol_ds_all_sky(data_root: str, problem: str = "gas_clouds") -> int:
    with Dataset(os.path.join(data_root, REFERENCE_FILES[(problem, "lw")])) as ds:
        return ds.jl("lw_flux_up").shape[0]


def compare_fluxes(flux_up, flux_dn, comp_up, comp_dn, FT):
    """L-inf errors as clear_sky_utils.jl:172-190 computes them: up, dn, net and the
    relative net error (denominators below 10 eps(FT) are left absolute)."""
    flux_up, flux_dn = np.asarray(flux_up, dtype=np.float64), np.asarray(flux_dn, dtype=np.float64)
    net, cnet = flux_up - flux_dn, comp_up - comp_dn
    rel = np.abs(net - cnet)
    den = np.abs(cnet)
    big = den > 10 * np.finfo(FT).eps
    rel[big] /= den[big]
    return {"up": float(np.abs(flux_up - comp_up).max()), "dn": float(np.abs(flux_dn - comp_dn).max()),
            "net": float(np.abs(net - cnet).max()), "rel_net": float(rel.max())}


def night_columns_are_dark(flux_up, flux_dn, cos_zenith) -> bool:
    """clear_sky_utils.jl:205-217: SW fluxes vanish where cos_zenith <= 0."""
    night = np.asarray(cos_zenith) <= 0
    return bool(np.all(np.asarray(flux_up)[:, night] == 0) and np.all(np.asarray(flux_dn)[:, night] == 0))


# ---- device-backed helpers and the end-to-end runs -------------------------------------------
def hip_column_routines(ws, params):
    """(col_gas, rel_hum) closures over the C ABI (ext/cuda/optics.jl:2,35 equivalents)."""
    from . import rte

    def col_gas(p_lev, vmr_h2o, lat=None):
        return rte.compute_col_gas(ws, p_lev, params, vmr_h2o, lat)

    def rel_hum(p_lay, t_lay, vmr_h2o):
        return rte.compute_relative_humidity(ws, p_lay, t_lay, params, vmr_h2o)
    return col_gas, rel_hum


def solve_case(case: Case, lookups: dict, FT, lw_twostream: bool, clouds: bool, aerosols: bool, seed: int = 0,
               device: int = 0):
    """Run LW (+SW two-stream) for a Case on the GPU; returns (flux_lw, flux_sw) on the host."""
    from . import rte
    nlay, ncol = case.as_.layerdata.shape[1:]
    ws = rte.Workspace(ncol, nlay, FT, device)
    lw_cls = rte.TwoStreamLWRTE if lw_twostream else rte.NoScatLWRTE
    slv_lw = lw_cls(ncol, nlay, FT, case.bcs_lw, device, workspace=ws)
    slv_sw = rte.TwoStreamSWRTE(ncol, nlay, FT, case.bcs_sw, device, workspace=ws)
    f_lw = rte.solve_lw(slv_lw, case.as_, lookups["lw"], lookups.get("lw_cld") if clouds else None,
                        lookups.get("lw_aero") if aerosols else None, seed=seed)
    f_sw = rte.solve_sw(slv_sw, case.as_, lookups["sw"], lookups.get("sw_cld") if clouds else None,
                        lookups.get("sw_aero") if aerosols else None, seed=seed)
    ws.synchronize()
    return f_lw, f_sw


def check_against_reference(name: str, f_lw, f_sw, comp, FT, lw_twostream: bool, cos_zenith=None) -> dict:
    """Apply the reference's pass criteria; returns the error report with a `passed` flag."""
    FT = np.dtype(FT).type
    tol = TOLERANCES[name]
    t_lw = tol["lw_2stream" if lw_twostream else "lw_noscat"][FT]
    t_sw = tol["sw"][FT]
    e_lw = compare_fluxes(f_lw.flux_up, f_lw.flux_dn, comp[0], comp[1], FT)
    e_sw = compare_fluxes(f_sw.flux_up, f_sw.flux_dn, comp[2], comp[3], FT)
    ok = all(e_lw[k] <= t_lw for k in ("up", "dn", "net")) and all(e_sw[k] <= t_sw for k in ("up", "dn", "net"))
    if cos_zenith is not None:
        ok = ok and night_columns_are_dark(f_sw.flux_up, f_sw.flux_dn, cos_zenith)
    return {"case": name, "lw": e_lw, "sw": e_sw, "tol_lw": t_lw, "tol_sw": t_sw, "passed": bool(ok)}
