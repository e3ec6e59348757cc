"""Seeded synthetic lookup tables and atmospheric columns.

The real k-distribution (rrtmgp-data v1.9) and the RFMIP / all-sky inputs are not
available offline, so tests, smoke() and bench.py run on synthetic data with the
true dimensionality (9 x 60 x 14 x 256 etc.) and the same table structure the
reference constructors produce (ext/lookup_constructors.jl).  Everything here is
deterministic in (seed, column index): column j is identical whatever `ncol` is
or however columns are sharded across GPUs.

Column recipe: the analytic profiles of the reference's `standard_atmosphere`
(src/api/atmosphere_profile.jl:44-163) blended in latitude, with per-column
perturbations; clouds placed by the rule of test/read_cloudy_sky.jl:106-121;
aerosols one active species per (layer, column) as test/read_all_sky_with_aerosols.jl:84-102.
"""
from __future__ import annotations

import numpy as np

from . import _abi
from .lookups import GasLookup, LookUpAerosolMerra, LookUpCld, LookUpMinor, build_minor_index
from .states import (AerosolState, AtmosphericState, CloudState, LwBCs, RRTMGPParameters, SwBCs, TEST_PARAMETERS, Vmr,
                     VmrGM)

# gas order of the v1.9 files (h2o = 1, o3 = 3 asserted by lookup_constructors.jl:9-16)
GAS_NAMES = ["h2o", "co2", "o3", "n2o", "co", "ch4", "o2", "n2", "ccl4", "cfc11", "cfc12", "cfc22", "hfc143a",
             "hfc125", "hfc23", "hfc32", "hfc134a", "cf4", "no2"]
IDX_GASES = {n: i + 1 for i, n in enumerate(GAS_NAMES)}

LW_BAND_WN = np.array([[10, 250], [250, 500], [500, 630], [630, 700], [700, 820], [820, 980], [980, 1080],
                       [1080, 1180], [1180, 1390], [1390, 1480], [1480, 1800], [1800, 2080], [2080, 2250],
                       [2250, 2390], [2390, 2680], [2680, 3250]], dtype=np.float64).T
SW_BAND_WN = np.array([[820, 2680], [2680, 3250], [3250, 4000], [4000, 4650], [4650, 5150], [5150, 6150],
                       [6150, 7700], [7700, 8050], [8050, 12850], [12850, 16000], [16000, 22650], [22650, 29000],
                       [29000, 38000], [38000, 50000]], dtype=np.float64).T

# typical volume mixing ratios (lower, upper atmosphere) used for vmr_ref and as well-mixed values
_TYPICAL_VMR = {"h2o": (8e-3, 5e-6), "co2": (4e-4, 4e-4), "o3": (5e-8, 4e-6), "n2o": (3.2e-7, 2e-7),
                "co": (1e-7, 3e-8), "ch4": (1.8e-6, 1.2e-6), "o2": (0.209, 0.209), "n2": (0.781, 0.781)}


# ---- counter-based uniform numbers, vectorised over column index ------------
def _mix64(z):
    z = (z ^ (z >> np.uint64(30))) * np.uint64(0xbf58476d1ce4e5b9)
    z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94d049bb133111eb)
    return z ^ (z >> np.uint64(31))


def col_uniform(seed: int, cols: np.ndarray, stream: int, n: int = 1) -> np.ndarray:
    """U[0,1) of shape (n, len(cols)) that depends only on (seed, column index, stream, k)."""
    with np.errstate(over="ignore"):
        g = np.uint64(0x9e3779b97f4a7c15)
        k = _mix64(np.uint64(seed) + g * (cols.astype(np.uint64) + np.uint64(1)))
        k = _mix64(k ^ np.uint64(stream * 1000003 + 17))
        out = np.empty((n, cols.shape[0]), dtype=np.float64)
        for j in range(n):
            out[j] = (_mix64(k + g * np.uint64(j + 1)) >> np.uint64(11)).astype(np.float64) / 9007199254740992.0
    return out


# ---- lookup tables -------------------------------------------------------------
def _planck_band_integrals(t_planck, wn_lims):
    """Band-integrated Planck radiance [W/m^2/sr] for wavenumber limits in cm^-1."""
    h, c, kb = 6.62607015e-34, 2.99792458e8, 1.380649e-23
    out = np.zeros((t_planck.shape[0], wn_lims.shape[1]))
    for ib in range(wn_lims.shape[1]):
        nu = np.linspace(wn_lims[0, ib], wn_lims[1, ib], 400) * 100.0  # m^-1
        x = h * c * nu[None, :] / (kb * t_planck[:, None])
        b = 2 * h * c * c * nu[None, :] ** 3 / np.expm1(x)
        out[:, ib] = np.trapezoid(b, nu, axis=1)
    return out


def _make_minor(rng, n_bnd, gpt_per_bnd, n_eta, n_t_ref, gases, nrange, kscale):
    """Random but structurally valid minor-gas block in NetCDF form, then re-ordered exactly
    as ext/lookup_constructors.jl:220-311 does."""
    bnd_lims_gpt = np.zeros((2, n_bnd), dtype=np.int64)
    g0 = 1
    for ib in range(n_bnd):
        bnd_lims_gpt[:, ib] = (g0, g0 + gpt_per_bnd[ib] - 1)
        g0 += gpt_per_bnd[ib]
    lims, gas, sgas, dens, comp = [], [], [], [], []
    for ib in range(n_bnd):
        for _ in range(int(rng.integers(nrange[0], nrange[1] + 1))):
            lims.append(bnd_lims_gpt[:, ib])
            gas.append(int(rng.choice(gases)))
            d = int(rng.integers(0, 2))
            dens.append(d)
            s = int(rng.choice([0, 0] + list(gases)))
            sgas.append(s)
            comp.append(int(rng.integers(0, 2)) if s > 0 else 0)
    if not lims:  # keep at least one interval so the arrays are non-empty
        lims, gas, sgas, dens, comp = [bnd_lims_gpt[:, 0]], [gases[0]], [0], [0], [0]
    lims = np.asarray(lims, dtype=np.int64).T.copy()
    bnd_st, gpt_st, reorder = build_minor_index(bnd_lims_gpt, lims)
    n_contrib_file = int(np.sum(lims[1] - lims[0] + 1))
    # file-order kminor (n_eta, n_t_ref, n_contrib): smooth, positive
    eta = np.linspace(0, 1, n_eta)[:, None, None]
    tt = np.linspace(-1, 1, n_t_ref)[None, :, None]
    amp = kscale * np.exp(rng.uniform(-3.0, 2.0, size=(1, 1, n_contrib_file)))
    kfile = amp * np.exp(0.6 * rng.uniform(-1, 1, size=(1, 1, n_contrib_file)) * tt +
                         0.8 * rng.uniform(-1, 1, size=(1, 1, n_contrib_file)) * eta)
    kminor = np.asfortranarray(kfile[:, :, reorder - 1])
    gasdata = np.asfortranarray(np.vstack([gas, sgas, dens, comp]).astype(np.int64))
    return LookUpMinor(bnd_st, gpt_st, gasdata, kminor), bnd_lims_gpt


def make_gas_lookup(kind: str, dtype=np.float64, seed: int = 2026, n_bnd=None, gpt_per_bnd=16,
                    n_minor_lower=(2, 5), n_minor_upper=(0, 3)) -> GasLookup:
    """Synthetic LookUpLW (kind="lw") / LookUpSW ("sw") with the v1.9 dimensionality by default."""
    is_sw = kind == "sw"
    band_wn = SW_BAND_WN if is_sw else LW_BAND_WN
    if n_bnd is None:
        n_bnd = band_wn.shape[1]
    band_wn = band_wn[:, :n_bnd]
    if np.isscalar(gpt_per_bnd):
        gpt_per_bnd = [int(gpt_per_bnd)] * n_bnd
    n_gpt = int(sum(gpt_per_bnd))
    rng = np.random.default_rng([seed, 1 if is_sw else 0])
    n_eta, n_p_ref, n_t_ref, ngas = 9, 59, 14, len(GAS_NAMES)
    ln_p_ref = np.log(109663.31) - 0.2 * np.arange(n_p_ref)
    p_ref_tropo = 9948.431564193395
    t_ref = 160.0 + 15.0 * np.arange(n_t_ref)
    # vmr_ref (2, ngas+1, n_t_ref); slot 1 is dry air
    vmr_ref = np.ones((2, ngas + 1, n_t_ref))
    tvar = 1.0 + 0.3 * np.linspace(-1, 1, n_t_ref)
    for name, ig in IDX_GASES.items():
        lo, up = _TYPICAL_VMR.get(name, (1e-10, 1e-10))
        vmr_ref[0, ig, :] = lo * tvar
        vmr_ref[1, ig, :] = up * tvar
    # key species: pairs among the gases every state provides; (2,2) stands for the
    # (0,0) -> (2,2) rewrite of lookup_constructors.jl:175-182
    major = [1, 2, 3, 4, 6, 7]
    key_species = np.zeros((2, 2, n_bnd), dtype=np.int64)
    for ib in range(n_bnd):
        for it in range(2):
            if rng.uniform() < 0.15:
                key_species[:, it, ib] = 2
            else:
                a, b = rng.choice(major, size=2, replace=True)
                key_species[:, it, ib] = (a, b)
    major_gpt2bnd = np.repeat(np.arange(1, n_bnd + 1, dtype=np.int64), gpt_per_bnd)
    # kmajor (n_eta, n_p_ref+1, n_t_ref, n_gpt), values ~1e-27..1e-19 like the real tables
    eta = np.linspace(0, 1, n_eta)[:, None, None, None]
    # the table duplicates the tropopause level: entries 1..j_trop are the lower atmosphere
    lnp = np.sort(np.append(ln_p_ref, np.log(p_ref_tropo)))[::-1].copy()
    pp = (lnp - lnp[0])[None, :, None, None]
    tt = ((t_ref - 250.0) / 100.0)[None, None, :, None]
    gfrac = np.concatenate([np.linspace(0, 1, n) for n in gpt_per_bnd])
    a = (-58.0 + 14.0 * gfrac + rng.uniform(-1.0, 1.0, n_gpt))[None, None, None, :]
    # column amounts scale with pressure, so weight the strong lines toward low pressure
    b = rng.uniform(-0.35, 0.25, n_gpt)[None, None, None, :]
    c = rng.uniform(-1.0, 1.5, n_gpt)[None, None, None, :]
    d = rng.uniform(-1.5, 1.5, n_gpt)[None, None, None, :]
    # key-species vmr differs by orders of magnitude between bands: normalise so tau spans ~1e-4..1e2
    kscale = np.ones(n_gpt)
    for ig_ in range(n_gpt):
        ib = major_gpt2bnd[ig_] - 1
        v = max(vmr_ref[0, key_species[0, 0, ib], 7], 1e-7)
        kscale[ig_] = 3e-3 / v
    kmajor = np.exp(a + b * pp + c * tt + d * eta) * kscale[None, None, None, :]
    if is_sw:
        kmajor *= 0.05
    lower, bnd_lims_gpt = _make_minor(rng, n_bnd, gpt_per_bnd, n_eta, n_t_ref, [1, 2, 3, 4, 6, 7, 8],
                                      n_minor_lower, 2e-26 if not is_sw else 2e-27)
    upper, _ = _make_minor(rng, n_bnd, gpt_per_bnd, n_eta, n_t_ref, [2, 3, 4, 6, 7], n_minor_upper,
                           5e-24 if not is_sw else 5e-25)
    kw = dict(is_sw=is_sw, idx_h2o=1, p_ref_tropo=p_ref_tropo, p_ref_min=float(np.exp(ln_p_ref[-1])),
              t_ref_min=float(t_ref[0]), t_ref_max=float(t_ref[-1]), key_species=np.asfortranarray(key_species),
              kmajor=np.asfortranarray(kmajor), major_gpt2bnd=major_gpt2bnd, bnd_lims_wn=np.asfortranarray(band_wn),
              ln_p_ref=ln_p_ref, t_ref=t_ref, vmr_ref=np.asfortranarray(vmr_ref), minor_lower=lower,
              minor_upper=upper)
    if not is_sw:
        pf = np.exp(rng.uniform(-1.0, 1.0, n_gpt)[None, None, None, :] +
                    0.5 * rng.uniform(-1, 1, n_gpt)[None, None, None, :] * tt +
                    0.3 * rng.uniform(-1, 1, n_gpt)[None, None, None, :] * eta +
                    0.05 * rng.uniform(-1, 1, n_gpt)[None, None, None, :] * pp)
        for ib in range(n_bnd):
            sl = slice(bnd_lims_gpt[0, ib] - 1, bnd_lims_gpt[1, ib])
            pf[..., sl] /= pf[..., sl].sum(axis=3, keepdims=True)
        t_planck = 160.0 + np.arange(196.0)
        kw.update(planck_fraction=np.asfortranarray(pf), t_planck=t_planck,
                  tot_planck=np.asfortranarray(_planck_band_integrals(t_planck, band_wn)))
    else:
        wn_mid = band_wn.mean(axis=0)[major_gpt2bnd - 1]
        ray = 4e-28 * (wn_mid / 1e4) ** 4
        shape = 1.0 + 0.05 * np.linspace(-1, 1, n_eta)[:, None, None] + 0.02 * np.linspace(-1, 1, n_t_ref)[None, :, None]
        rayl_lower = shape * ray[None, None, :]
        rayl_upper = rayl_lower * 1.02
        solar = rng.uniform(0.2, 1.0, n_gpt) * np.exp(-((wn_mid - 15000.0) / 14000.0) ** 2)
        solar /= solar.sum()
        kw.update(solar_src_tot=1360.8583984375, rayl_lower=np.asfortranarray(rayl_lower),
                  rayl_upper=np.asfortranarray(rayl_upper), solar_src_scaled=solar)
    return GasLookup(**kw).astype(dtype)


def make_cloud_lookup(kind: str, nband: int, dtype=np.float64, seed: int = 2026) -> LookUpCld:
    """Synthetic LookUpCld: nsize_liq = 20, nsize_ice = 18, 3 roughness classes
    (docs/src/Optics.md:252-258); ice bounds halved as lookup_constructors.jl:741-743 does."""
    rng = np.random.default_rng([seed, 10 + (kind == "sw")])
    nl, ni, nr = 20, 18, 3
    bounds = np.array([2.5, 21.5, 10.0 / 2, 180.0 / 2])
    rl = np.linspace(bounds[0], bounds[1], nl)[:, None]
    ri = np.linspace(bounds[2], bounds[3], ni)[:, None, None]
    sw = kind == "sw"
    bfac = rng.uniform(0.8, 1.2, nband)[None, :]
    ext_l = 1.6 / rl * bfac
    ssa_l = np.clip((0.999 if sw else 0.55) - (0.02 if sw else 0.2) * rng.uniform(0, 1, nband)[None, :] * (rl / 21.5),
                    0.01, 0.999999)
    asy_l = np.clip(0.80 + 0.06 * (rl / 21.5) + 0.02 * rng.uniform(-1, 1, nband)[None, :], 0.0, 0.95)
    bfi = rng.uniform(0.8, 1.2, (nband, nr))[None, :, :]
    ext_i = 1.3 / ri * bfi
    ssa_i = np.clip((0.995 if sw else 0.5) - (0.05 if sw else 0.15) * rng.uniform(0, 1, (nband, nr))[None] * (ri / 90.0),
                    0.01, 0.999999)
    asy_i = np.clip(0.75 + 0.12 * (ri / 90.0) + 0.02 * rng.uniform(-1, 1, (nband, nr))[None], 0.0, 0.95)
    liq = np.concatenate([ext_l, ssa_l, asy_l], axis=0)
    ice = np.concatenate([ext_i, ssa_i, asy_i], axis=0)
    return LookUpCld(np.array([nband, nr, nl, ni, 2], dtype=np.int64), bounds, np.asfortranarray(liq),
                     np.asfortranarray(ice)).astype(dtype)


def make_aerosol_lookup(kind: str, band_wn: np.ndarray, dtype=np.float64, seed: int = 2026) -> LookUpAerosolMerra:
    """Synthetic LookUpAerosolMerra: 5 size bins, 36 RH levels; `iband_550nm` by the
    test of lookup_constructors.jl:41-44."""
    nband = band_wn.shape[1]
    rng = np.random.default_rng([seed, 20 + (kind == "sw")])
    nbin, nrh = 5, 36
    lims = np.array([[0.1, 1.0], [1.0, 1.8], [1.8, 3.0], [3.0, 6.0], [6.0, 10.0]]).T
    rh = np.concatenate([np.linspace(0.0, 0.80, 17), np.linspace(0.81, 0.99, 19)])
    sw = kind == "sw"

    def tab(shape_mid):
        ext = 1.0e3 * np.exp(rng.uniform(-1.5, 1.0, shape_mid))
        ssa = rng.uniform(0.75, 0.98, shape_mid) if sw else rng.uniform(0.05, 0.5, shape_mid)
        asy = rng.uniform(0.4, 0.8, shape_mid)
        return np.asfortranarray(np.stack([ext, ssa, asy], axis=0))

    def tab_rh(shape_tail):
        t = tab((1,) + shape_tail)
        grow = 1.0 + 2.5 * rh ** 3
        t = np.repeat(t, nrh, axis=1)
        t[0] *= grow.reshape((nrh,) + (1,) * len(shape_tail))
        return np.asfortranarray(t)

    i550 = 0
    for ib in range(nband):
        if 1.0 / (band_wn[1, ib] * 100.0) <= 550e-9 <= 1.0 / (band_wn[0, ib] * 100.0):
            i550 = ib + 1
    return LookUpAerosolMerra(np.asfortranarray(lims), rh, tab((nbin, nband)), tab_rh((nbin, nband)),
                              tab_rh((nband,)), tab_rh((nband,)), tab((nband,)), tab_rh((nband,)), tab((nband,)),
                              i550).astype(dtype)


# ---- columns -------------------------------------------------------------------
_KINDS = np.array([
    # t_sfc, z_trop, gamma_trop, gamma_strat, vmr_h2o_sfc, lat    (atmosphere_profile.jl:44-69)
    [300.0, 17.0e3, 6.5e-3, 2.2e-3, 2.3e-2, 0.0],
    [294.0, 13.0e3, 6.5e-3, 2.0e-3, 1.4e-2, 45.0],
    [257.0, 9.0e3, 5.0e-3, 1.5e-3, 1.6e-3, 65.0],
])


def compute_col_dry(p_lev, vmr_h2o, params: RRTMGPParameters, lat=None):
    """col_dry of src/optics/gas_optics.jl:16-47 in numpy (state preparation of the synthetic columns)."""
    ft = p_lev.dtype.type
    g0 = ft(params.grav) if lat is None else ft(params.grav) - ft(0.02586) * np.cos(ft(2) * ft(np.pi) * lat / ft(180))
    dp = p_lev[:-1, :] - p_lev[1:, :]
    m_air = ft(params.molmass_dryair) + ft(params.molmass_water) * vmr_h2o
    return np.asfortranarray(dp * ft(params.avogad) / (ft(100 * 100) * m_air * g0))


def compute_rel_hum(p_lay, t_lay, vmr_h2o, params: RRTMGPParameters):
    """relative humidity of src/optics/gas_optics.jl:58-80 in numpy."""
    ft = p_lay.dtype.type
    mwd = ft(params.molmass_water) / ft(params.molmass_dryair)
    mmr = vmr_h2o * mwd
    q = np.maximum(ft(1e-7), mmr / (ft(1) + mmr))
    es = np.exp((ft(17.67) * (t_lay - ft(273.16))) / (t_lay - ft(29.65)))
    return np.asfortranarray(np.maximum(ft(0.01) * (ft(0.263) * p_lay * q) / es, ft(0)))


def make_columns(ncol: int, nlay: int, dtype=np.float64, seed: int = 2026, col_offset: int = 0,
                 vmr_kind: str = "gm", clouds: bool = True, cld_frac: float = 1.0, aerosols: bool = False,
                 n_bnd_lw: int = 16, n_bnd_sw: int = 14, cos_zenith=None, night_fraction: float = 0.0,
                 params: RRTMGPParameters = TEST_PARAMETERS, cloud_bounds=(2.5, 21.5, 5.0, 90.0),
                 z_top: float = 45.0e3, inc_flux_ngpt: int = 0, random_cld_frac: bool = False):
    """Synthetic AtmosphericState + boundary conditions for global columns
    [col_offset, col_offset + ncol).  Returns (as, lw_bcs, sw_bcs)."""
    ft = np.dtype(dtype).type
    cols = np.arange(col_offset, col_offset + ncol, dtype=np.int64)
    u = col_uniform(seed, cols, 0, 8)
    lat = -80.0 + 160.0 * u[0]
    alat = np.abs(lat)
    prm = np.stack([np.interp(alat, _KINDS[:, 5], _KINDS[:, k]) for k in range(5)])  # (5, ncol)
    t_sfc0, z_trop, g_trop, g_strat, h2o_sfc = prm
    t_sfc_col = t_sfc0 + 15.0 * (2 * u[1] - 1)
    h2o_scale = 0.3 + 1.7 * u[2]
    p_sfc = 101325.0 * (1.0 + 0.03 * (2 * u[3] - 1))
    grav, r_d = params.grav, params.R_d
    nlev = nlay + 1
    z_lev = np.linspace(0.0, z_top, nlev)[:, None]
    z_lay = 0.5 * (z_lev[:-1] + z_lev[1:])

    def temp(z):
        t_trop = t_sfc_col - g_trop * z_trop
        return np.where(z <= z_trop, t_sfc_col - g_trop * z, t_trop + g_strat * (z - z_trop))

    def pres(z):
        t_trop = t_sfc_col - g_trop * z_trop
        p_trop = p_sfc * (t_trop / t_sfc_col) ** (grav / (r_d * g_trop))
        return np.where(z <= z_trop, p_sfc * (temp(np.minimum(z, z_trop)) / t_sfc_col) ** (grav / (r_d * g_trop)),
                        p_trop * (temp(np.maximum(z, z_trop)) / t_trop) ** (-grav / (r_d * g_strat)))

    t_lev = np.clip(temp(z_lev), 160.0, 355.0)
    t_lay = np.clip(temp(z_lay), 160.0, 355.0)
    p_lev, p_lay = pres(z_lev), pres(z_lay)
    vmr_h2o = np.maximum(h2o_sfc * h2o_scale * np.exp(-z_lay / 2.0e3), 4.0e-6)
    vmr_o3 = 3.0e-8 + 7.5e-6 * np.exp(-(np.log(p_lay / 1.2e3)) ** 2 / (2 * 1.2 ** 2))
    t_sfc = t_lev[0] + 2.0 * (2 * u[4] - 1)

    F = lambda a: np.asfortranarray(a, dtype=ft)
    p_lev, p_lay, t_lev, t_lay, vmr_h2o, vmr_o3 = map(F, (p_lev, p_lay, t_lev, t_lay, vmr_h2o, vmr_o3))
    lat_ft = lat.astype(ft)
    col_dry = compute_col_dry(p_lev, vmr_h2o, params, lat_ft)
    rel_hum = compute_rel_hum(p_lay, t_lay, vmr_h2o, params)
    layerdata = np.empty((4, nlay, ncol), dtype=ft, order="F")
    layerdata[0], layerdata[1], layerdata[2], layerdata[3] = col_dry, p_lay, t_lay, rel_hum

    ngas = len(GAS_NAMES)
    well_mixed = np.zeros(ngas, dtype=ft)
    for name, v in (("co2", 348e-6), ("ch4", 1650e-9), ("n2o", 306e-9), ("n2", 0.7808), ("o2", 0.2095), ("co", 1e-7)):
        well_mixed[IDX_GASES[name] - 1] = v
    if vmr_kind == "gm":
        vmr = VmrGM(vmr_h2o, vmr_o3, well_mixed)
    else:
        full = np.empty((ngas, nlay, ncol), dtype=ft, order="F")
        full[:] = well_mixed[:, None, None]
        full[IDX_GASES["h2o"] - 1] = vmr_h2o
        full[IDX_GASES["o3"] - 1] = vmr_o3
        vmr = Vmr(full)

    cloud_state = None
    if clouds:
        radliq_lwr, radliq_upr, radice_lwr, radice_upr = cloud_bounds
        r_liq, r_ice = (radliq_lwr + radliq_upr) / 2, (radice_lwr + radice_upr) / 2
        in_band = (p_lay > 10000) & (p_lay < 90000) & (((cols + 1) % 3) != 0)[None, :]
        if random_cld_frac:
            cf = col_uniform(seed, cols, 3, nlay)
        else:
            cf = np.full((nlay, ncol), cld_frac)
        cfrac = np.where(in_band, cf, 0.0)
        liq = in_band & (t_lay > 263)
        ice = in_band & (t_lay < 273)
        cloud_state = CloudState(F(np.where(liq, r_liq, 0.0)), F(np.where(ice, r_ice, 0.0)),
                                 F(np.where(liq, 10.0, 0.0)), F(np.where(ice, 10.0, 0.0)), F(cfrac),
                                 np.full(ncol, np.nan, dtype=ft), np.full(ncol, np.nan, dtype=ft), 2)
    aerosol_state = None
    if aerosols:
        na = _abi.N_AEROSOLS
        mass = np.zeros((na, nlay, ncol), dtype=ft, order="F")
        size = np.zeros((na, nlay, ncol), dtype=ft, order="F")
        ua = col_uniform(seed, cols, 5, 2 * nlay).reshape(2, nlay, ncol)
        species = (cols[None, :] + np.arange(nlay)[:, None]) % na  # 0-based
        active = p_lay > 70000
        m = 10.0 ** (-7.0 + 3.0 * ua[0])
        bins = np.array([[0.1, 1.0], [1.0, 1.8], [1.8, 3.0], [3.0, 6.0], [6.0, 10.0]])
        bin_of = {0: 0, 7: 1, 8: 2, 9: 3, 10: 4, 1: 0, 11: 1, 12: 2, 13: 3, 14: 4}
        li, ci = np.nonzero(active)
        sp = species[li, ci]
        mass[sp, li, ci] = m[li, ci]
        for s, b in bin_of.items():
            sel = sp == s
            size[s, li[sel], ci[sel]] = bins[b, 0] + (bins[b, 1] - bins[b, 0]) * ua[1][li[sel], ci[sel]]
        aerosol_state = AerosolState(size, mass, np.full(ncol, np.nan, dtype=ft), np.full(ncol, np.nan, dtype=ft))

    as_ = AtmosphericState(layerdata, p_lev, t_lev, t_sfc.astype(ft), vmr, lat_ft, cloud_state, aerosol_state)

    sfc_emis = np.full((n_bnd_lw, ncol), 0.98, dtype=ft, order="F")
    sfc_emis *= (1.0 - 0.05 * u[5]).astype(ft)[None, :]
    inc_flux = None
    if inc_flux_ngpt:
        inc_flux = np.asfortranarray(0.05 * col_uniform(seed, cols, 7, inc_flux_ngpt).T, dtype=ft)
    lw_bcs = LwBCs(sfc_emis, inc_flux)
    if cos_zenith is None:
        mu0 = np.cos(np.deg2rad(85.0 * u[6]))
    else:
        mu0 = np.full(ncol, float(cos_zenith))
    if night_fraction > 0:
        night = u[7] < night_fraction
        mu0 = np.where(night, -0.3 * u[6] - (u[6] < 0.2) * 0.0, mu0)
        mu0 = np.where(night & (u[6] < 0.25), 0.0, mu0)  # exercise mu0 == 0 exactly
    alb = (0.06 + 0.5 * u[5] * (u[3] > 0.7)).astype(ft)
    sw_bcs = SwBCs(mu0.astype(ft), np.full(ncol, 1360.8583984375, dtype=ft),
                   np.asfortranarray(np.repeat(alb[None, :], n_bnd_sw, 0)),
                   np.asfortranarray(np.repeat((alb * ft(0.9))[None, :], n_bnd_sw, 0)))
    return as_, lw_bcs, sw_bcs
