"""Parity evidence that does NOT pass through anyone's reading of the Julia sources: physics and closed-form mathematics.

Every other parity test compares the HIP path with `oracle/` (two transcriptions of the reference).  The expected values
here come from properties of the equations the path solves and from analytic answers, so a misreading shared by the C
oracle, the numpy twin and the kernels would still be caught:

  (a) LAYER SPLITTING.  Two-stream layer reflectance / transmittance are the exact solution of the two-stream ODE for a
      homogeneous layer and the adding method composes layers exactly, so a column whose every layer is cut into two halves
      (same p, T, vmr; col_dry, cloud paths, aerosol masses halved: every optical depth halves bit-exactly) must give the
      same fluxes at the original levels.  Shortwave on general cloudy + aerosol-laden columns (direct beam included);
      longwave two-stream and no-scattering on isothermal columns with a height-independent Planck fraction (the linear-in-
      tau source assumption is then exact).  src/rte/shortwave_2stream.jl:189-392, longwave_2stream.jl:149-334.
  (b) CONSERVATIVE SCATTERING.  A Rayleigh-only shortwave lookup (kmajor = 0, no minor absorbers) scatters with ssa = 1:
      nothing is absorbed, so flux_dn - flux_up is the same at every level and what enters at the top leaves through the
      top or into the surface.  With conservative clouds (ssa = 1 in the cloud table) the same holds through
      increment_2stream and the delta scaling.  (The solver's k_min clamp leaks sqrt(eps) tau^2 / 2 per layer; the bounds
      below are that leak.)
  (c) LINEAR-TABLE EXACTNESS.  With kmajor affine in (eta, ln p, T) trilinear interpolation is exact, so the optical depth
      of every (layer, g-point) has a closed form in the physical coordinates: tau = col_dry col_mix (A + B eta + C ln p +
      D T).  Purely absorbing shortwave: the direct beam is Beer-Lambert and the upwelling flux the surface reflection
      attenuated with diffusivity 2.  Longwave on an isothermal column over a black surface: flux_up = pi sum_b P_b(T),
      flux_dn(k) = pi sum_g pf_g P_b(T) (1 - exp(-D tau_above)).  Pins compute_interp_frac_{temp,press,eta} + interp3d
      (src/optics/gas_optics.jl:87-170, optics_utils.jl:136-181) and the transport of both longwave solvers.
  (d) OPTICALLY THIN / THICK LIMITS of the longwave sources (longwave_noscat.jl:171-205 incl. its small-tau series,
      longwave_2stream.jl:149-222): thin -> pi D sum tau B (layer source for no-scattering, the mean of the level sources
      for two-stream); thick -> the local Planck flux pi B(T_lev) in both directions.

Each case runs on the C oracle and the numpy twin here (CPU) and on the HIP path through the C ABI (`-m gpu`), Float64
tight and Float32 at the rounding budget, on the GPU at BASELINE config 3 / config 4 sizes (128 x 64, 4096 x 73)."""
import dataclasses

import numpy as np
import pytest

from oracle import np_oracle as T  # noqa: E402
from oracle import oracle as O  # noqa: E402
from rrtmgp_jl_amd import synthetic as S  # noqa: E402
from rrtmgp_jl_amd.lookups import LookUpMinor  # noqa: E402
from rrtmgp_jl_amd.states import AerosolState, AtmosphericState, CloudState, LwBCs, SwBCs, VmrGM  # noqa: E402

HIP = pytest.param("hip", marks=pytest.mark.gpu)
BACKENDS = ["oracle", "twin", HIP]
F = lambda a, dt=None: np.asfortranarray(a, dtype=dt)  # noqa: E731


# ---- one interface over the three implementations ------------------------------------------------------------------
def run_lw(backend, as_, bcs, lw, cld=None, aero=None, twostream=True, n_angles=1):
    """(flux_up, flux_dn) as float64 (nlev, ncol)."""
    if backend == "twin":
        assert as_.dtype == np.float64
        if twostream:
            return T.solve_lw_2stream(lw, as_, bcs, cld, lka=aero)
        Ds, w = O.angular_discretization(n_angles)
        up = dn = 0.0
        for s in range(n_angles):
            u, d = T.solve_lw_noscat(lw, as_, bcs, cld, Ds=Ds[s], w=w[s], lka=aero)
            up, dn = up + u, dn + d
        return up, dn
    if backend == "oracle":
        f = O.solve_lw(as_, bcs, lw, cld, aero, twostream=twostream, n_gauss_angles=n_angles)
    else:
        from rrtmgp_jl_amd import rte
        nlay, ncol = as_.dims
        cls = rte.TwoStreamLWRTE if twostream else rte.NoScatLWRTE
        f = rte.solve_lw(cls(ncol, nlay, as_.dtype, bcs, n_gauss_angles=n_angles), as_, lw, cld, aero)
    return f.flux_up.astype(np.float64), f.flux_dn.astype(np.float64)


def run_sw(backend, as_, bcs, sw, cld=None, aero=None):
    """(flux_up, flux_dn, flux_dn_dir) of the two-stream solver as float64 (nlev, ncol)."""
    if backend == "twin":
        assert as_.dtype == np.float64
        return T.solve_sw_2stream(sw, as_, bcs, cld, lka=aero)
    if backend == "oracle":
        f = O.solve_sw(as_, bcs, sw, cld, aero)
    else:
        from rrtmgp_jl_amd import rte
        nlay, ncol = as_.dims
        f = rte.solve_sw(rte.TwoStreamSWRTE(ncol, nlay, as_.dtype, bcs), as_, sw, cld, aero)
    return f.flux_up.astype(np.float64), f.flux_dn.astype(np.float64), f.flux_dn_dir.astype(np.float64)


def sizes(backend, big=(128, 64)):
    """(ncol, nlay): the twin loops over columns in Python, the oracle is a C loop, the GPU takes a BASELINE-size batch."""
    return {"twin": (3, 24), "oracle": (24, 40)}.get(backend, big)


# ---- (a) layer splitting ---------------------------------------------------------------------------------------------
def split_layers(as_):
    """The same column with every layer cut into two halves: p, T, RH, vmr and particle sizes repeated, column amounts
    (col_dry, cloud water paths, aerosol masses) halved — an exact operation in binary floating point."""
    ld = as_.layerdata
    nlay, ncol = ld.shape[1:]
    rep = lambda a: F(np.repeat(a, 2, axis=0))  # noqa: E731
    ld2 = np.empty((4, 2 * nlay, ncol), ld.dtype, order="F")
    for i in range(4):
        ld2[i] = np.repeat(ld[i], 2, axis=0)
    ld2[0] *= ld.dtype.type(0.5)

    def lev2(x, lay):
        out = np.empty((2 * nlay + 1, ncol), x.dtype, order="F")
        out[0::2], out[1::2] = x, lay
        return out
    v, cs, a = as_.vmr, as_.cloud_state, as_.aerosol_state
    half = ld.dtype.type(0.5)
    cs2 = None if cs is None else CloudState(rep(cs.cld_r_eff_liq), rep(cs.cld_r_eff_ice), F(rep(cs.cld_path_liq) * half),
                                             F(rep(cs.cld_path_ice) * half), rep(cs.cld_frac), cs.cld_cover_sw.copy(),
                                             cs.cld_cover_lw.copy(), cs.ice_rgh)
    a2 = None if a is None else AerosolState(F(np.repeat(a.aero_size, 2, axis=1)), F(np.repeat(a.aero_mass, 2, axis=1) * half),
                                             a.aod_sw_ext.copy(), a.aod_sw_sca.copy())
    return AtmosphericState(ld2, lev2(as_.p_lev, ld[1]), lev2(as_.t_lev, ld[2]), as_.t_sfc, VmrGM(rep(v.vmr_h2o), rep(v.vmr_o3), v.vmr),
                            as_.lat, cs2, a2)


@pytest.mark.parametrize("backend", BACKENDS)
@pytest.mark.parametrize("ft", [np.float64, np.float32])
@pytest.mark.parametrize("sky", ["clear", "clouds+aerosols"])
def test_layer_splitting_shortwave(tables64, backend, ft, sky):
    """SW two-stream, general columns (overcast McICA clouds: the sample is the same for both halves of a layer)."""
    if backend == "twin" and ft == np.float32:
        pytest.skip("the numpy twin computes in Float64")
    ncol, nlay = sizes(backend, big=(4096, 73) if sky != "clear" else (128, 64))   # config 4: 146 half-layers (deep-column mask path)
    full = sky != "clear"
    sw = tables64["sw"].astype(ft)
    cld = tables64["cld_sw"].astype(ft) if full else None
    aero = tables64["aero_sw"].astype(ft) if full else None
    as_, _, sb = S.make_columns(ncol, nlay, ft, seed=31, clouds=full, aerosols=full, cld_frac=1.0)
    one = run_sw(backend, as_, sb, sw, cld, aero)
    two = run_sw(backend, split_layers(as_), sb, sw, cld, aero)
    # Float64: composition is exact up to rounding (measured 3e-12 W/m2 on the oracle at 40 layers, 4e-9 on the GPU at 2 x 73 layers x 4096 columns with its 2-ulp Float64 exp); Float32: the rounding of twice as
    # many layers (measured 3e-3 on 1360 W/m2)
    tol = 2e-8 if ft == np.float64 else 2.5e-2
    assert one[1].max() > 100.0
    for name, a, b in zip(("flux_up", "flux_dn", "flux_dn_dir"), one, two):
        d = np.abs(a - b[0::2]).max()
        assert d < tol, (name, d)


def _const_planck_fraction(lw):
    """The lookup with a Planck fraction that does not depend on (eta, p, T): each g-point keeps its table mean, renormalised
    per band.  Level sources of an isothermal column are then the same at every level."""
    pf = np.broadcast_to(lw.planck_fraction.mean(axis=(0, 1, 2), keepdims=True), lw.planck_fraction.shape).copy()
    for b in range(lw.n_bnd):
        sel = lw.major_gpt2bnd == b + 1
        pf[..., sel] /= pf[..., sel].sum(axis=3, keepdims=True)
    return dataclasses.replace(lw, planck_fraction=F(pf, lw.dtype))


def _isothermal(as_, t0):
    ld = as_.layerdata.copy(order="F")
    ld[2] = t0
    return dataclasses.replace(as_, layerdata=ld, t_lev=F(np.full_like(as_.t_lev, t0)), t_sfc=np.full_like(as_.t_sfc, t0))


@pytest.mark.parametrize("backend", BACKENDS)
@pytest.mark.parametrize("ft", [np.float64, np.float32])
@pytest.mark.parametrize("solver", ["2stream", "noscat1", "noscat3"])
def test_layer_splitting_longwave_isothermal(tables64, backend, ft, solver):
    if backend == "twin" and ft == np.float32:
        pytest.skip("the numpy twin computes in Float64")
    ncol, nlay = sizes(backend)
    lw = _const_planck_fraction(tables64["lw"]).astype(ft)
    cld = tables64["cld_lw"].astype(ft)
    as_, lb, _ = S.make_columns(ncol, nlay, ft, seed=32, clouds=True, cld_frac=1.0)
    as_ = _isothermal(as_, ft(271.0))
    kw = dict(twostream=solver == "2stream", n_angles=3 if solver == "noscat3" else 1)
    one = run_lw(backend, as_, lb, lw, cld, **kw)
    two = run_lw(backend, split_layers(as_), lb, lw, cld, **kw)
    tol = 2e-8 if ft == np.float64 else 2e-3
    for name, a, b in zip(("flux_up", "flux_dn"), one, two):
        assert a.max() > 100.0
        d = np.abs(a - b[0::2]).max()
        assert d < tol, (name, d)


# ---- (b) conservative scattering -------------------------------------------------------------------------------------
def _zero_minor(m):
    return LookUpMinor(m.bnd_st, m.gpt_st, m.gasdata, F(np.zeros_like(m.kminor)))


def rayleigh_only(sw):
    return dataclasses.replace(sw, kmajor=F(np.zeros_like(sw.kmajor)), minor_lower=_zero_minor(sw.minor_lower),
                               minor_upper=_zero_minor(sw.minor_upper))


def conservative_clouds(cld):
    """The cloud lookup with single-scattering albedo 1 for liquid and ice (rows nsize..2 nsize of the tables)."""
    nl, ni = int(cld.dims[2]), int(cld.dims[3])
    liq, ice = cld.liqdata.copy(order="F"), cld.icedata.copy(order="F")
    liq[nl:2 * nl] = 1.0
    ice[ni:2 * ni] = 1.0
    return dataclasses.replace(cld, liqdata=liq, icedata=ice)


@pytest.mark.parametrize("backend", BACKENDS)
@pytest.mark.parametrize("ft,clouds", [(np.float64, False), (np.float64, True), (np.float32, False)])
def test_conservative_scattering_conserves_energy(tables64, backend, ft, clouds):
    if backend == "twin" and ft == np.float32:
        pytest.skip("the numpy twin computes in Float64")
    ncol, nlay = sizes(backend, big=(4096, 73) if clouds else (128, 64))
    sw = rayleigh_only(tables64["sw"]).astype(ft)
    cld = conservative_clouds(tables64["cld_sw"]).astype(ft) if clouds else None
    as_, _, sb = S.make_columns(ncol, nlay, ft, seed=33, clouds=clouds, cld_frac=1.0)
    up, dn, dr = run_sw(backend, as_, sb, sw, cld)
    mu0 = sb.cos_zenith.astype(np.float64)
    s0 = sb.toa_flux.astype(np.float64) * mu0 * float(tables64["sw"].solar_src_scaled.sum())
    net = dn - up
    # what enters at the top: the incident beam, nothing diffuse
    np.testing.assert_allclose(dn[-1], s0, rtol=1e-12 if ft == np.float64 else 2e-6)
    # ... leaves through the top or into the surface: no level absorbs.  The only sink is the k_min clamp of the
    # two-stream coefficients (k >= eps^(1/4) although gamma1 = gamma2): 1 - R - T = sqrt(eps) tau^2 / (2 (1 + gamma tau)) per
    # layer and g-point, i.e. < 1e-5 W/m2 per clear Float64 layer, < 1e-3 per cloudy one, ~2e-3 per clear Float32 layer
    leak = np.abs(net - net[-1:]).max()
    bound = {(np.float64, False): 2e-4, (np.float64, True): 5e-2, (np.float32, False): 0.4}[(ft, clouds)]
    assert leak < bound, leak
    # the leak is an absorption: the net flux never grows downward by more than rounding
    assert (np.diff(net, axis=0) > -(1e-9 if ft == np.float64 else 2e-2)).all()
    assert (up[-1] > 1.0).all() and (up[-1] < s0).all()       # something is scattered back to space, not everything
    # clear columns reflect more with a brighter surface, whatever the atmosphere: the energy went somewhere sensible
    absorbed_sfc = net[0]
    np.testing.assert_allclose(up[-1] + absorbed_sfc, s0, atol=bound)


# ---- (c) linear-table exactness --------------------------------------------------------------------------------------
def affine_lookup(lk, seed, kscale):
    """`lk` with kmajor = kscale_g (A + B eta + C ln p + D T) on the table's own axes (the p axis carries the tropopause level
    twice: lower-atmosphere planes first), vmr_ref independent of T (one binary-species parameter per layer), no minor
    absorbers, no Rayleigh scattering.  Returns (lookup, coefficient dict)."""
    r = np.random.default_rng(seed)
    n_eta, n_p1, n_t, n_gpt = lk.kmajor.shape
    co = dict(A=r.uniform(1.6, 2.6, n_gpt), B=r.uniform(-0.4, 0.4, n_gpt), C=r.uniform(-0.03, 0.03, n_gpt),
              D=r.uniform(-1.5e-3, 1.5e-3, n_gpt), s=kscale * np.exp(r.uniform(-3.0, 3.0, n_gpt)))
    eta = np.linspace(0.0, 1.0, n_eta)[:, None, None, None]
    lnp_axis = np.sort(np.append(lk.ln_p_ref.astype(np.float64), np.log(lk.p_ref_tropo)))[::-1]
    assert lnp_axis.shape == (n_p1,)
    k = co["s"] * (co["A"] + co["B"] * eta + co["C"] * lnp_axis[None, :, None, None] + co["D"] * lk.t_ref.astype(np.float64)[None, None, :, None])
    assert (k > 0).all()
    vref = np.repeat(lk.vmr_ref[:, :, 6:7], n_t, axis=2)
    kw = dict(kmajor=F(k), vmr_ref=F(vref), minor_lower=_zero_minor(lk.minor_lower), minor_upper=_zero_minor(lk.minor_upper))
    if lk.is_sw:
        kw.update(rayl_lower=F(np.zeros_like(lk.rayl_lower)), rayl_upper=F(np.zeros_like(lk.rayl_upper)))
    return dataclasses.replace(lk, **kw), co


def closed_form_tau(lk, co, as_):
    """tau[layer, column, g-point] of the affine lookup from the physical coordinates alone (Float64)."""
    ld = as_.layerdata.astype(np.float64)
    col_dry, p, t = ld[0], ld[1], ld[2]
    nlay, ncol = p.shape
    v = as_.vmr

    def vmr_of(ig):                       # get_vmr: 0 = dry air (1), h2o / o3 profiles, the others well mixed
        if ig == 0:
            return np.ones((nlay, ncol))
        if ig == 1:
            return v.vmr_h2o.astype(np.float64)
        if ig == 3:
            return v.vmr_o3.astype(np.float64)
        return np.full((nlay, ncol), float(v.vmr[ig - 1]))
    upper = ~(p > lk.p_ref_tropo)
    tau = np.empty((nlay, ncol, lk.n_gpt))
    for b in range(lk.n_bnd):
        sel = np.nonzero(lk.major_gpt2bnd == b + 1)[0]
        col_mix = np.empty((nlay, ncol)); eta = np.empty((nlay, ncol))
        for region in (0, 1):
            g1, g2 = (int(x) for x in lk.key_species[:, region, b])
            half = float(lk.vmr_ref[region, g1, 0]) / float(lk.vmr_ref[region, g2, 0])
            cm = vmr_of(g1) + half * vmr_of(g2)
            m = upper == bool(region)
            col_mix[m] = cm[m]
            eta[m] = (vmr_of(g1) / cm)[m]
        K = co["s"][sel] * (co["A"][sel] + co["B"][sel] * eta[..., None] + co["C"][sel] * np.log(p)[..., None] + co["D"][sel] * t[..., None])
        tau[:, :, sel] = (col_dry * col_mix)[..., None] * K
    return tau


def _cum_from_top(tau):
    """optical depth between level k and the top of the column, (nlev, ncol, ngpt)"""
    out = np.zeros((tau.shape[0] + 1,) + tau.shape[1:])
    out[:-1] = np.cumsum(tau[::-1], axis=0)[::-1]
    return out


def _cum_from_sfc(tau):
    out = np.zeros((tau.shape[0] + 1,) + tau.shape[1:])
    out[1:] = np.cumsum(tau, axis=0)
    return out


@pytest.mark.parametrize("backend", BACKENDS)
@pytest.mark.parametrize("ft", [np.float64, np.float32])
def test_linear_table_shortwave_beer_lambert(tables64, backend, ft):
    if backend == "twin" and ft == np.float32:
        pytest.skip("the numpy twin computes in Float64")
    ncol, nlay = sizes(backend)
    sw64, co = affine_lookup(tables64["sw"], seed=41, kscale=2e-23)
    sw = sw64.astype(ft)
    as_, _, sb = S.make_columns(ncol, nlay, ft, seed=34, clouds=False)
    up, dn, dr = run_sw(backend, as_, sb, sw)
    tau = closed_form_tau(sw, co, as_)
    assert 0.05 < np.median(tau.sum(axis=0)) < 50.0      # the columns are neither transparent nor black
    mu0 = sb.cos_zenith.astype(np.float64)
    top = (sb.toa_flux.astype(np.float64) * mu0)[:, None] * sw.solar_src_scaled.astype(np.float64)[None, :]   # (ncol, ngpt)
    beam = top[None] * np.exp(-_cum_from_top(tau) / mu0[None, :, None])
    bnd = sw.major_gpt2bnd - 1
    refl = beam[0] * sb.sfc_alb_direct.astype(np.float64)[bnd, :].T                       # (ncol, ngpt)
    want_up = (refl[None] * np.exp(-2.0 * _cum_from_sfc(tau))).sum(axis=2)
    want_dir = beam.sum(axis=2)
    tol = 1e-8 if ft == np.float64 else 4e-2      # Float32: interpolation fractions of ln p carry ~1e-6 relative
    assert np.abs(dr - want_dir).max() < tol, np.abs(dr - want_dir).max()
    assert np.abs(dn - want_dir).max() < tol      # nothing scatters: no diffuse downwelling
    assert np.abs(up - want_up).max() < tol, np.abs(up - want_up).max()


@pytest.mark.parametrize("backend", BACKENDS)
@pytest.mark.parametrize("ft", [np.float64, np.float32])
@pytest.mark.parametrize("solver", ["2stream", "noscat1", "noscat2"])
def test_linear_table_longwave_isothermal(tables64, backend, ft, solver):
    if backend == "twin" and ft == np.float32:
        pytest.skip("the numpy twin computes in Float64")
    ncol, nlay = sizes(backend)
    lw64, co = affine_lookup(_const_planck_fraction(tables64["lw"]), seed=42, kscale=3e-22)
    lw = lw64.astype(ft)
    t0 = 267.0                                        # a node of the Planck table (1 K), not of t_ref (15 K): fT = 0.133
    as_, lb, _ = S.make_columns(ncol, nlay, ft, seed=35, clouds=False)
    as_ = _isothermal(as_, ft(t0))
    lb = LwBCs(F(np.ones_like(lb.sfc_emis)), None)    # black surface at the air temperature, nothing incident
    n_angles = {"2stream": 1, "noscat1": 1, "noscat2": 2}[solver]
    up, dn = run_lw(backend, as_, lb, lw, twostream=solver == "2stream", n_angles=n_angles)
    tau = closed_form_tau(lw, co, as_)
    it = int(np.nonzero(lw.t_planck == t0)[0][0])
    bnd = lw.major_gpt2bnd - 1
    B = (lw64.tot_planck[it, bnd] * lw64.planck_fraction[0, 0, 0, :])[None, None, :]     # Planck source per g-point
    above = _cum_from_top(tau)
    if solver == "2stream":
        emitted = 1.0 - np.exp(-1.66 * above)
    else:
        Ds, w = O.angular_discretization(n_angles)
        emitted = sum(w[s] * (1.0 - np.exp(-Ds[s] * above)) for s in range(n_angles))
    want_dn = np.pi * (B * emitted).sum(axis=2)
    want_up = np.full_like(want_dn, np.pi * lw64.tot_planck[it].sum())
    assert 50.0 < want_dn[0].min() and want_dn[-1].max() == 0.0
    tol = 1e-8 if ft == np.float64 else 1e-3
    assert np.abs(up - want_up).max() < tol, np.abs(up - want_up).max()
    assert np.abs(dn - want_dn).max() < tol, np.abs(dn - want_dn).max()


def absorbing_particles(cld, aero, seed):
    """Cloud and aerosol lookups that only absorb (ssa = 0) with closed-form extinction: cloud extinction affine in the
    particle size (so the size interpolation of cloud_optics.jl:170-182 is exact), one extinction per band for every aerosol
    species, bin and humidity (so the sum over species does not depend on the MERRA index conventions).  Returns
    (cloud lookup, aerosol lookup, coefficients)."""
    r = np.random.default_rng(seed)
    nband, nrgh, nl, ni = (int(x) for x in cld.dims[:4])
    b = cld.bounds.astype(np.float64)
    rl, ri = np.linspace(b[0], b[1], nl), np.linspace(b[2], b[3], ni)
    co = dict(l0=r.uniform(0.02, 0.06, nband), l1=r.uniform(-1e-3, 2e-3, nband), i0=r.uniform(0.01, 0.04, (nband, nrgh)),
              i1=r.uniform(-1e-4, 3e-4, (nband, nrgh)), aer=r.uniform(2e2, 2e3, nband), bounds=b)
    liq, ice = np.zeros_like(cld.liqdata, dtype=np.float64), np.zeros_like(cld.icedata, dtype=np.float64)
    liq[:nl] = co["l0"][None, :] + co["l1"][None, :] * rl[:, None]
    ice[:ni] = co["i0"][None] + co["i1"][None] * ri[:, None, None]
    liq[2 * nl:], ice[2 * ni:] = 0.8, 0.7                       # asymmetry: irrelevant without scattering
    assert (liq[:nl] > 0).all() and (ice[:ni] > 0).all()
    cld2 = dataclasses.replace(cld, liqdata=F(liq), icedata=F(ice))
    kw = {}
    for name in ("dust", "sea_salt", "sulfate", "black_carbon_rh", "black_carbon", "organic_carbon_rh", "organic_carbon"):
        t = np.zeros_like(getattr(aero, name), dtype=np.float64)
        t[0] = co["aer"]                                        # (ext, ssa, asy) x ... x band: band is the last axis
        t[2] = 0.6
        kw[name] = F(t)
    return cld2, dataclasses.replace(aero, **kw), co


def closed_form_particle_tau(co, as_):
    """tau[layer, column, band] of the absorbing clouds (overcast where cloudy) and aerosols, from the state alone."""
    cs, a = as_.cloud_state, as_.aerosol_state
    eps = np.finfo(as_.dtype).eps
    b = co["bounds"]
    rl = np.clip(cs.cld_r_eff_liq.astype(np.float64), b[0], b[1])[..., None]
    ri = np.clip(cs.cld_r_eff_ice.astype(np.float64), b[2], b[3])[..., None]
    pl, pi = cs.cld_path_liq.astype(np.float64)[..., None], cs.cld_path_ice.astype(np.float64)[..., None]
    rg = int(cs.ice_rgh) - 1
    cloudy = (cs.cld_frac > 0)[..., None]
    tau = np.where(cloudy & (pl > eps), (co["l0"] + co["l1"] * rl) * pl, 0.0) + \
        np.where(cloudy & (pi > eps), (co["i0"][:, rg] + co["i1"][:, rg] * ri) * pi, 0.0)
    mass = np.where(a.aero_mass > 0, a.aero_mass, 0).astype(np.float64).sum(axis=0)      # (nlay, ncol)
    return tau + mass[..., None] * co["aer"]


@pytest.mark.parametrize("backend", BACKENDS)
@pytest.mark.parametrize("ft", [np.float64, np.float32])
@pytest.mark.parametrize("solver", ["sw", "lw2stream", "lwnoscat"])
def test_linear_tables_with_absorbing_clouds_and_aerosols(tables64, backend, ft, solver):
    """(c) with particles: affine gas tables + absorbing clouds (extinction affine in particle size, overcast McICA) + absorbing
    aerosols.  Every optical depth is closed-form, nothing scatters: Beer-Lambert shortwave, isothermal longwave.  Pins the
    cloud size interpolation, the cloud / aerosol increments of both optics flavours (TwoStream and OneScalar) and the McICA
    mask of overcast layers, without the oracle."""
    if backend == "twin" and ft == np.float32:
        pytest.skip("the numpy twin computes in Float64")
    ncol, nlay = sizes(backend, big=(4096, 73))
    kind = "sw" if solver == "sw" else "lw"
    base = tables64["sw"] if kind == "sw" else _const_planck_fraction(tables64["lw"])
    lk64, co = affine_lookup(base, seed=45, kscale=2e-23 if kind == "sw" else 3e-22)
    cld64, aero64, pco = absorbing_particles(tables64["cld_" + kind], tables64["aero_" + kind], seed=46)
    lk, cld, aero = lk64.astype(ft), cld64.astype(ft), aero64.astype(ft)
    as_, lb, sb = S.make_columns(ncol, nlay, ft, seed=38, clouds=True, cld_frac=1.0, aerosols=True)
    bnd = lk.major_gpt2bnd - 1
    if solver == "sw":
        up, dn, dr = run_sw(backend, as_, sb, lk, cld, aero)
        tau = closed_form_tau(lk, co, as_) + closed_form_particle_tau(pco, as_)[..., bnd]
        mu0 = sb.cos_zenith.astype(np.float64)
        top = (sb.toa_flux.astype(np.float64) * mu0)[:, None] * lk.solar_src_scaled.astype(np.float64)[None, :]
        beam = top[None] * np.exp(-_cum_from_top(tau) / mu0[None, :, None])
        refl = beam[0] * sb.sfc_alb_direct.astype(np.float64)[bnd, :].T
        want_up, want_dir = (refl[None] * np.exp(-2.0 * _cum_from_sfc(tau))).sum(axis=2), beam.sum(axis=2)
        tol = 5e-8 if ft == np.float64 else 4e-2    # (1.1e-8 on the GPU at 4 096 x 73: its Float64 exp is a 2-ulp form, 73 cumulative optical depths)
        assert (closed_form_particle_tau(pco, as_).sum(axis=0).max(axis=1) > 0.5).any()      # the clouds matter
        for name, got, want in (("dir", dr, want_dir), ("dn", dn, want_dir), ("up", up, want_up)):
            assert np.abs(got - want).max() < tol, (name, np.abs(got - want).max())
        return
    t0 = 267.0
    as_ = _isothermal(as_, ft(t0))
    lb = LwBCs(F(np.ones_like(lb.sfc_emis)), None)
    up, dn = run_lw(backend, as_, lb, lk, cld, aero, twostream=solver == "lw2stream")
    tau = closed_form_tau(lk, co, as_) + closed_form_particle_tau(pco, as_)[..., bnd]
    it = int(np.nonzero(lk.t_planck == t0)[0][0])
    B = (lk64.tot_planck[it, bnd] * lk64.planck_fraction[0, 0, 0, :])[None, None, :]
    D = 1.66 if solver == "lw2stream" else 1.0 / 0.6096748751
    want_dn = np.pi * (B * (1.0 - np.exp(-D * _cum_from_top(tau)))).sum(axis=2)
    want_up = np.full_like(want_dn, np.pi * lk64.tot_planck[it].sum())
    tol = 1e-8 if ft == np.float64 else 1e-3
    assert np.abs(up - want_up).max() < tol, np.abs(up - want_up).max()
    assert np.abs(dn - want_dn).max() < tol, np.abs(dn - want_dn).max()


# ---- (d) optically thin / thick limits of the longwave sources ----------------------------------------------------------
def _scaled(as_, factor):
    ld = as_.layerdata.copy(order="F")
    ld[0] *= ld.dtype.type(factor)
    return dataclasses.replace(as_, layerdata=ld)


@pytest.mark.parametrize("backend", BACKENDS)
@pytest.mark.parametrize("solver", ["2stream", "noscat1"])
def test_longwave_thin_limit(tables64, backend, solver):
    """tau -> 0 (Float64): flux_dn(k) = pi D sum_{layers above} tau B + O(tau^2), B the layer source (no-scattering:
    longwave_noscat.jl:178-181, through the small-tau series of `fact`) or the mean of the level sources (two-stream)."""
    ncol, nlay = sizes(backend)
    lw, co = affine_lookup(_const_planck_fraction(tables64["lw"]), seed=43, kscale=3e-22)
    as_, lb, _ = S.make_columns(ncol, nlay, np.float64, seed=36, clouds=False)
    as_ = _scaled(as_, 1e-6 / closed_form_tau(lw, co, as_).max())
    up, dn = run_lw(backend, as_, lb, lw, twostream=solver == "2stream")
    tau = closed_form_tau(lw, co, as_)
    assert tau.max() < 0.02 * np.sqrt(np.sqrt(np.finfo(np.float64).eps))     # every layer is on the series branch (tau D < eps^(1/4))
    bnd = lw.major_gpt2bnd - 1
    pf = lw.planck_fraction[0, 0, 0, :]
    planck = lambda temp: np.stack([np.interp(temp, lw.t_planck, lw.tot_planck[:, b]) for b in range(lw.n_bnd)], axis=-1)  # noqa: E731
    if solver == "noscat1":
        D = 1.0 / 0.6096748751
        src = planck(as_.layerdata[2])[..., bnd] * pf
    else:
        D = 1.66
        lev = planck(as_.t_lev)[..., bnd] * pf
        src = 0.5 * (lev[:-1] + lev[1:])
    want_dn = np.pi * D * _cum_from_top(tau * src).sum(axis=2)
    np.testing.assert_allclose(dn[:-1], want_dn[:-1], rtol=5e-6)
    assert (dn[-1] == 0).all()
    # the transparent atmosphere leaves the surface emission alone on its way up
    sfc = np.pi * (lb.sfc_emis.astype(np.float64)[bnd, :].T * (planck(as_.t_sfc)[..., bnd] * pf)).sum(axis=1)
    np.testing.assert_allclose(up[-1], sfc, rtol=1e-5)


@pytest.mark.parametrize("backend", BACKENDS)
@pytest.mark.parametrize("solver", ["2stream", "noscat1"])
def test_longwave_thick_limit(tables64, backend, solver):
    """tau -> infinity (Float64): both fluxes tend to the local Planck flux pi B(T_lev) at every interior level, with a
    correction of the order of the source difference across one optical depth, dB / tau."""
    ncol, nlay = sizes(backend)
    lw, co = affine_lookup(_const_planck_fraction(tables64["lw"]), seed=44, kscale=3e-22)
    as_, lb, _ = S.make_columns(ncol, nlay, np.float64, seed=37, clouds=False)
    tau1 = closed_form_tau(lw, co, as_)
    as_ = _scaled(as_, 2e4 / tau1.min())
    up, dn = run_lw(backend, as_, lb, lw, twostream=solver == "2stream")
    bnd = lw.major_gpt2bnd - 1
    pf = lw.planck_fraction[0, 0, 0, :]
    planck = lambda temp: np.stack([np.interp(temp, lw.t_planck, lw.tot_planck[:, b]) for b in range(lw.n_bnd)], axis=-1)  # noqa: E731
    local = np.pi * (planck(as_.t_lev)[..., bnd] * pf).sum(axis=2)
    np.testing.assert_allclose(up[1:-1], local[1:-1], rtol=2e-4)
    np.testing.assert_allclose(dn[1:-1], local[1:-1], rtol=2e-4)


# ---- (e) the other kernel instances under the same invariants -----------------------------------------------------------
@pytest.mark.gpu
@pytest.mark.parametrize("ft", [np.float64, np.float32])
@pytest.mark.parametrize("solver", ["sw", "lw"])
@pytest.mark.parametrize("instance", ["diag", "band"])      # the library rejects both in one launch (EINVAL)
def test_layer_splitting_on_the_diag_and_band_instances(tables64, ft, solver, instance):
    """The one-pass clear-sky twin (DIAG) and the per-band outputs (BAND) are separate template instances of the solve
    kernels (solve_lw.hip / solve_sw.hip): the layer-splitting invariant of (a) on their extra outputs, plus what the
    outputs mean — band fluxes sum to the broadband flux, the clear-sky pair equals a solve without the cloud lookup."""
    from rrtmgp_jl_amd import rte
    from rrtmgp_jl_amd.states import Flux
    ncol, nlay = 512, 64
    is_sw = solver == "sw"
    lk = (tables64["sw"] if is_sw else _const_planck_fraction(tables64["lw"])).astype(ft)
    cld = tables64["cld_sw" if is_sw else "cld_lw"].astype(ft)
    aero = tables64["aero_sw" if is_sw else "aero_lw"].astype(ft)
    as_, lb, sb = S.make_columns(ncol, nlay, ft, seed=41, clouds=True, aerosols=True, cld_frac=1.0)
    if not is_sw:
        as_ = _isothermal(as_, ft(268.0))
    cls, solve, bcs = (rte.TwoStreamSWRTE, rte.solve_sw, sb) if is_sw else (rte.TwoStreamLWRTE, rte.solve_lw, lb)
    names = ("flux_up", "flux_dn") + (("flux_dn_dir",) if is_sw else ())

    def run(state):
        nl, nc = state.dims
        slv = cls(nc, nl, ft, bcs, n_bnd_band_flux=lk.n_bnd if "band" in instance else 0)
        clear = Flux.allocate(nc, nl + 1, ft, sw=is_sw) if "diag" in instance else None
        solve(slv, state, lk, cld, aero, seed=9, clear_flux=clear)
        out = {"all": [np.asarray(getattr(slv.flux, n), np.float64) for n in names]}
        if clear is not None:
            out["clear"] = [np.asarray(getattr(clear, n), np.float64) for n in names]
        if slv.band_flux is not None:
            out["band"] = [np.asarray(getattr(slv.band_flux, n), np.float64) for n in ("flux_up", "flux_dn")]
        return out

    one, two = run(as_), run(split_layers(as_))
    # same budgets as (a): Float64 the rounding of 2 x 64 layers with the 2-ulp exp, Float32 the rounding of twice as many layers
    tol = 2e-8 if ft == np.float64 else (2.5e-2 if is_sw else 2e-3)
    assert one["all"][1].max() > 100.0
    for key in one:
        for a, b in zip(one[key], two[key]):
            d = np.abs(a - b[0::2]).max()
            assert d < tol, (key, d)
    sum_tol = 1e-9 if ft == np.float64 else 2e-3
    if "band" in one:
        for a, b in zip(one["band"], one["all"]):
            assert np.abs(a.sum(axis=2) - b).max() < sum_tol * max(1.0, b.max() / 100.0)
    if "clear" in one:
        plain = cls(ncol, nlay, ft, bcs)
        solve(plain, dataclasses.replace(as_, cloud_state=None), lk, None, aero, seed=9)
        for n, a in zip(names, one["clear"]):
            assert np.abs(a - np.asarray(getattr(plain.flux, n), np.float64)).max() < sum_tol * 15, n
        assert np.abs(one["clear"][0] - one["all"][0]).max() > 1.0     # the clouds do something
