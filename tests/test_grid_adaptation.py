"""N2: prepare_atmosphere! (src/api/grid_adaptation.jl, interpolation.jl, update_fluxes.jl:252-281).

CPU: the oracle restatement against the reference's own known answers — the closed forms of
test/interpolation_schemes.jl and the worked examples of test/grid_adaptation.jl:85-165 (values
copied as data).  GPU: the one-launch HIP cascade against the oracle, every scheme, both state
kinds, host- and device-resident arrays, and through RRTMGPSolver.update_fluxes."""
import copy

import numpy as np
import pytest

from rrtmgp_jl_amd import _abi, synthetic as S
from rrtmgp_jl_amd import grid_adaptation as GA
from rrtmgp_jl_amd.states import (AtmosphericState, GrayAtmosphericState, GrayOpticalThicknessSchneider2004,
                                  RRTMGPParameters, TEST_PARAMETERS, VmrGM)
from oracle import oracle

SCHEMES = [GA.ArithmeticMean, GA.GeometricMean, GA.UniformZ, GA.UniformP, GA.BestFit]
BOTTOMS = [GA.SameAsInterpolation, GA.UseSurfaceTempAtBottom, GA.HydrostaticBottom]


def _make_as(FT, nlay, ncol):
    """test/grid_adaptation.jl `_make_as`: zero-filled clear-sky state with VmrGM storage."""
    z = lambda *s: np.zeros(s, dtype=FT, order="F")   # noqa: E731
    return AtmosphericState(z(4, nlay, ncol), z(nlay + 1, ncol), z(nlay + 1, ncol), np.full(ncol, FT(300)),
                            VmrGM(z(nlay, ncol), z(nlay, ncol), z(3)))


def _two_layer(FT, p1, T1, p2, T2, z=None, ts=300.0):
    """A 2-layer, 1-column state: level 2 is interp(layer 1, layer 2); level 1 the bottom extrapolation
    from (layer 1, layer 2); level 3 the top extrapolation from (layer 2, layer 1)."""
    a = _make_as(FT, 2, 1)
    a.layerdata[1, :, 0] = (p1, p2)
    a.layerdata[2, :, 0] = (T1, T2)
    a.t_sfc[:] = ts
    kw = {}
    if z is not None:
        kw = dict(center_z=np.asfortranarray(np.array([[z[1]], [z[3]]], dtype=FT)),
                  face_z=np.asfortranarray(np.array([[z[0]], [z[2]], [z[4]]], dtype=FT)))
    return a, kw


@pytest.mark.parametrize("FT", [np.float32, np.float64])
def test_interp_closed_forms(FT):
    """test/interpolation_schemes.jl "interp! closed forms": pd/Td below, pu/Tu above."""
    rt = 1e-5 if FT is np.float32 else 1e-12
    for pd, Td, pu, Tu in ((100000, 300, 80000, 280), (90000, 290, 70000, 270)):
        a, _ = _two_layer(FT, pd, Td, pu, Tu)
        oracle.prepare_atmosphere(a, TEST_PARAMETERS, _abi.PREP_INTERPOLATE, interpolation=GA.UniformZ)
        assert a.t_lev[1, 0] == pytest.approx((Td + Tu) / 2, rel=rt)
        assert min(pd, pu) < a.p_lev[1, 0] < max(pd, pu)
        a, _ = _two_layer(FT, pd, Td, pu, Td)     # isothermal limit -> geometric mean of the pressures
        oracle.prepare_atmosphere(a, TEST_PARAMETERS, _abi.PREP_INTERPOLATE, interpolation=GA.UniformZ)
        assert a.t_lev[1, 0] == pytest.approx(Td, rel=rt) and a.p_lev[1, 0] == pytest.approx(np.sqrt(pd * pu), rel=rt)
        a, _ = _two_layer(FT, pd, Td, pu, Tu)
        oracle.prepare_atmosphere(a, TEST_PARAMETERS, _abi.PREP_INTERPOLATE, interpolation=GA.UniformP)
        assert a.p_lev[1, 0] == pytest.approx((pd + pu) / 2, rel=rt)
        assert min(Td, Tu) < a.t_lev[1, 0] < max(Td, Tu)
        a, _ = _two_layer(FT, pd, Td, pu, Td)
        oracle.prepare_atmosphere(a, TEST_PARAMETERS, _abi.PREP_INTERPOLATE, interpolation=GA.UniformP)
        assert a.t_lev[1, 0] == pytest.approx(Td, rel=rt) and a.p_lev[1, 0] == pytest.approx((pd + pu) / 2, rel=rt)
    # BestFit: T linear in z, p on the fitted power law, between the layer pressures
    a, kw = _two_layer(FT, 100000, 300, 80000, 280, z=(-500, 0, 500, 1000, 1500))
    oracle.prepare_atmosphere(a, TEST_PARAMETERS, _abi.PREP_INTERPOLATE, interpolation=GA.BestFit, **kw)
    assert a.t_lev[1, 0] == pytest.approx(290.0, rel=rt) and 80000 < a.p_lev[1, 0] < 100000
    # BestFit, isothermal: p1 (p2/p1)^((z - z1)/(z2 - z1))  (test/grid_adaptation.jl:58-59: 1, 3 at z = 1, 2 -> 1.5)
    a, kw = _two_layer(FT, 1.0, 280, 3.0, 280, z=(0.5, 1.0, 1.5, 2.0, 2.5))
    oracle.prepare_atmosphere(a, TEST_PARAMETERS, _abi.PREP_INTERPOLATE, interpolation=GA.BestFit, **kw)
    assert a.t_lev[1, 0] == pytest.approx(280.0, rel=rt) and a.p_lev[1, 0] == pytest.approx(3.0 ** 0.5, rel=rt)


@pytest.mark.parametrize("FT", [np.float32, np.float64])
def test_extrap_closed_forms(FT):
    """test/interpolation_schemes.jl "extrap! closed forms": nearest (p+, T+), next (p++, T++)."""
    rt = 2e-5 if FT is np.float32 else 1e-12
    ps = RRTMGPParameters()
    pp, Tp, ppp, Tpp, Ts = 90000.0, 285.0, 80000.0, 275.0, 300.0

    def bottom(interp, bot=GA.SameAsInterpolation, T2=Tpp, ts=Ts, z=None):
        a, kw = _two_layer(FT, pp, Tp, ppp, T2, z=z, ts=ts)
        oracle.prepare_atmosphere(a, ps, _abi.PREP_INTERPOLATE, interpolation=interp, bottom_extrapolation=bot, **kw)
        return float(a.p_lev[0, 0]), float(a.t_lev[0, 0])
    p, T = bottom(GA.GeometricMean)
    assert T / Tp == pytest.approx(np.sqrt(Tp / Tpp), rel=rt) and p / pp == pytest.approx(np.sqrt(pp / ppp), rel=rt)
    p, T = bottom(GA.ArithmeticMean)   # test/grid_adaptation.jl:75-76 pattern
    assert T == pytest.approx((3 * Tp - Tpp) / 2, rel=rt) and p == pytest.approx((3 * pp - ppp) / 2, rel=rt)
    p, T = bottom(GA.UniformZ)
    assert T == pytest.approx((3 * Tp - Tpp) / 2, rel=rt) and p > pp
    p, T = bottom(GA.UniformZ, T2=Tp)
    assert T == pytest.approx(Tp, rel=rt) and p == pytest.approx(np.sqrt(pp * ppp), rel=rt)
    p, T = bottom(GA.UniformP)
    assert p == pytest.approx((3 * pp - ppp) / 2, rel=rt) and T > Tp
    p, T = bottom(GA.UniformP, T2=Tp)
    assert T == pytest.approx(Tp, rel=rt) and p == pytest.approx((3 * pp - ppp) / 2, rel=rt)
    p, T = bottom(GA.ArithmeticMean, GA.UseSurfaceTempAtBottom)
    assert T == Ts and p == pytest.approx(pp * (Ts / Tp) ** (ps.cp_d / ps.R_d), rel=rt) and p > pp
    p, T = bottom(GA.ArithmeticMean, GA.UseSurfaceTempAtBottom, ts=Tp)
    assert p == pytest.approx(pp, rel=rt)
    zs = (0.0, 500.0, 1000.0, 1500.0, 2000.0)
    p, T = bottom(GA.ArithmeticMean, GA.HydrostaticBottom, z=zs)
    assert T == pytest.approx(Tp + ps.grav / ps.cp_d * (500.0 - 0.0), rel=rt) and p > pp
    p, T = bottom(GA.ArithmeticMean, GA.HydrostaticBottom, z=(500.0, 500.0, 1000.0, 1500.0, 2000.0))
    assert T == pytest.approx(Tp, rel=rt) and p == pytest.approx(pp, rel=rt)
    assert GA.requires_z(GA.BestFit) and GA.requires_z(GA.HydrostaticBottom)
    assert not GA.requires_z(GA.ArithmeticMean) and not GA.requires_z(GA.UniformZ) and not GA.requires_z(GA.NoInterpolation)
    with pytest.raises(ValueError, match="center_z"):
        bottom(GA.BestFit)


def test_interpolate_levels_reference_example():
    """test/grid_adaptation.jl:85-119, values verbatim."""
    nlay, ncol = 4, 2
    a = _make_as(np.float64, nlay, ncol)
    p_lay, t_lay = a.layerdata[1], a.layerdata[2]
    p_lay[:] = [[1000, 1100], [800, 850], [600, 620], [400, 410]]
    t_lay[:] = [[290, 292], [270, 272], [250, 252], [230, 232]]
    oracle.prepare_atmosphere(a, TEST_PARAMETERS, _abi.PREP_INTERPOLATE, interpolation=GA.ArithmeticMean)
    for k in range(1, nlay):
        np.testing.assert_allclose(a.p_lev[k], (p_lay[k - 1] + p_lay[k]) / 2)
        np.testing.assert_allclose(a.t_lev[k], (t_lay[k - 1] + t_lay[k]) / 2)
    np.testing.assert_allclose(a.t_lev[nlay], (3 * t_lay[nlay - 1] - t_lay[nlay - 2]) / 2)
    np.testing.assert_allclose(a.t_lev[0], (3 * t_lay[0] - t_lay[1]) / 2)
    before = copy.deepcopy(a)
    oracle.prepare_atmosphere(a, TEST_PARAMETERS, _abi.PREP_INTERPOLATE, interpolation=GA.NoInterpolation)
    np.testing.assert_array_equal(a.p_lev, before.p_lev)


def test_isothermal_layer_and_clip_reference_example():
    """test/grid_adaptation.jl:121-165, values verbatim."""
    nlay = 4
    a = _make_as(np.float64, nlay, 1)
    p_lay, t_lay, rh = a.layerdata[1], a.layerdata[2], a.layerdata[3]
    a.p_lev[:, 0] = [1000, 800, 600, 400, 200]
    a.t_lev[:, 0] = [300, 280, 260, 240, 220]
    rh[:3, 0] = [0.8, 0.6, 0.4]
    a.vmr.vmr_h2o[:, 0] = [0.05, 0.04, 0.03, 0.0]
    p_min = 10.0
    oracle.prepare_atmosphere(a, TEST_PARAMETERS, _abi.PREP_ISOTHERMAL, isothermal_boundary_layer=True, p_min=p_min)
    assert a.p_lev[-1, 0] == p_min and p_lay[-1, 0] == (a.p_lev[-2, 0] + p_min) / 2
    assert t_lay[-1, 0] == a.t_lev[-2, 0] == a.t_lev[-1, 0]
    assert rh[-1, 0] == rh[-2, 0] == 0.4 and a.vmr.vmr_h2o[-1, 0] == a.vmr.vmr_h2o[-2, 0] == 0.03
    p_lay[0, 0], a.vmr.vmr_h2o[0, 0] = -5.0, -1.0
    oracle.prepare_atmosphere(a, TEST_PARAMETERS, _abi.PREP_CLIP, p_min=p_min)
    assert p_lay[0, 0] == p_min and a.vmr.vmr_h2o[0, 0] == 0
    t_lay[0, 0], a.t_lev[-1, 0] = 120.0, 400.0
    oracle.prepare_atmosphere(a, TEST_PARAMETERS, _abi.PREP_CLIP, p_min=p_min, t_min=160.0, t_max=355.0)
    assert t_lay[0, 0] == 160.0 and a.t_lev[-1, 0] == 355.0
    t_lay[0, 0] = 120.0
    oracle.prepare_atmosphere(a, TEST_PARAMETERS, _abi.PREP_CLIP, p_min=p_min)   # no bounds: untouched
    assert t_lay[0, 0] == 120.0


def test_col_dry_step_equals_compute_col_gas(small_tables64):
    as_, _, _ = S.make_columns(5, 9, seed=2, n_bnd_lw=3, n_bnd_sw=3, vmr_kind="full")
    want = oracle.compute_col_gas(as_.p_lev, TEST_PARAMETERS, np.asfortranarray(as_.vmr.vmr[0]), as_.lat)
    as_.layerdata[0] = -1.0
    oracle.prepare_atmosphere(as_, TEST_PARAMETERS, _abi.PREP_COL_DRY, idx_h2o=1)
    np.testing.assert_array_equal(as_.layerdata[0], want)


# ---- GPU ---------------------------------------------------------------------------------------
def _perturbed_columns(FT, vmr_kind, iso, ncol=11, nlay=13, clouds=True, aerosols=True):
    as_, _, _ = S.make_columns(ncol, nlay, FT, seed=31, vmr_kind=vmr_kind, clouds=clouds, aerosols=aerosols,
                               random_cld_frac=True)
    h2o = as_.vmr.vmr_h2o if vmr_kind == "gm" else as_.vmr.vmr[0]
    h2o[2, 3] = -1e-3
    as_.layerdata[2][1, 0] = 100.0
    as_.layerdata[1][nlay - 2, 5] = 0.2
    rng = np.random.default_rng(1)
    zf = np.asfortranarray(np.sort(rng.uniform(0, 4.5e4, (nlay + 1, ncol)), axis=0).astype(FT))
    zc = np.asfortranarray((0.5 * (zf[:-1] + zf[1:])).astype(FT))
    return as_, zc, zf


def _assert_state_close(got, ref, rtol):
    for n in ("p_lev", "t_lev"):
        np.testing.assert_allclose(getattr(got, n), getattr(ref, n), rtol=rtol, atol=0, err_msg=n)
    np.testing.assert_allclose(got.layerdata[1:], ref.layerdata[1:], rtol=rtol, atol=0, err_msg="layerdata[2:4]")
    # col_dry ~ p_lev[k] - p_lev[k+1]: the subtraction amplifies the level-pressure round-off by p/dp (<~ 50)
    np.testing.assert_allclose(got.layerdata[0], ref.layerdata[0], rtol=50 * rtol, atol=0, err_msg="col_dry")
    for c in ("vmr", "cloud_state", "aerosol_state"):
        a, b = getattr(got, c), getattr(ref, c)
        if a is None:
            continue
        for f in a.__dataclass_fields__:
            x, y = getattr(a, f), getattr(b, f)
            if isinstance(x, np.ndarray) and x.dtype.kind == "f":
                np.testing.assert_allclose(x, y, rtol=rtol, atol=0, equal_nan=True, err_msg=f"{c}.{f}")


@pytest.mark.parametrize("vmr_kind", ["gm", "full"])
def test_numpy_restatement_of_prepare_atmosphere_agrees_with_the_c_oracle(tables64, vmr_kind):
    """N2 twice (round 5): oracle/np_oracle.py restates interpolate_levels! / add_isothermal_boundary_layer! / clip! /
    update_concentrations! from the Julia sources (grid_adaptation.jl:60-292, interpolation.jl:148-252) with no code shared
    with the C oracle; every interpolation scheme x bottom scheme x isothermal layer, both Vmr kinds, Float64."""
    from oracle import np_oracle as NP
    lw = tables64["lw"]
    for interp in SCHEMES + [GA.NoInterpolation]:
        for bot in BOTTOMS:
            for iso in (False, True):
                as_, zc, zf = _perturbed_columns(np.float64, vmr_kind, iso)
                ref = copy.deepcopy(as_)
                kw = dict(interpolation=interp, bottom_extrapolation=bot, isothermal_boundary_layer=iso, center_z=zc,
                          face_z=zf)
                oracle.prepare_atmosphere(ref, TEST_PARAMETERS, _abi.PREP_ALL, p_min=lw.p_ref_min, t_min=lw.t_ref_min,
                                          t_max=lw.t_ref_max, idx_h2o=lw.idx_h2o, **kw)
                NP.prepare_atmosphere(as_, TEST_PARAMETERS, interp, bot, iso, zc, zf, lw.p_ref_min, lw.t_ref_min,
                                      lw.t_ref_max, lw.idx_h2o)
                _assert_state_close(as_, ref, 1e-13)


@pytest.mark.gpu
@pytest.mark.parametrize("FT,rtol", [(np.float64, 1e-13), (np.float32, 1e-4)])
@pytest.mark.parametrize("vmr_kind", ["gm", "full"])
def test_hip_prepare_matches_oracle_all_schemes(tables64, FT, rtol, vmr_kind):
    """Float32: powf/logf differ by a few ulp between libm and the device library, and the power
    laws divide by log(T2/T1), which cancels for nearly isothermal layer pairs (measured 3.5e-5
    relative on p) -> 1e-4; Float64 agrees to 1e-13."""
    from rrtmgp_jl_amd import rte
    lw = tables64["lw"]
    for interp in SCHEMES:
        for bot in BOTTOMS:
            for iso in (False, True):
                as_, zc, zf = _perturbed_columns(FT, vmr_kind, iso)
                ref = copy.deepcopy(as_)
                kw = dict(interpolation=interp, bottom_extrapolation=bot, isothermal_boundary_layer=iso, center_z=zc,
                          face_z=zf)
                oracle.prepare_atmosphere(ref, TEST_PARAMETERS, _abi.PREP_ALL, p_min=lw.p_ref_min, t_min=lw.t_ref_min,
                                          t_max=lw.t_ref_max, idx_h2o=lw.idx_h2o, **kw)
                ws = rte.Workspace(as_.dims[1], as_.dims[0], FT)
                GA.prepare_atmosphere(ws, as_, TEST_PARAMETERS, lw, **kw)
                _assert_state_close(as_, ref, rtol)


@pytest.mark.gpu
def test_hip_prepare_steps_gray_and_device_memory(tables64):
    import torch
    from rrtmgp_jl_amd import rte
    lw = tables64["lw"]
    # separable steps == the fused cascade
    as_, zc, zf = _perturbed_columns(np.float64, "gm", True)
    fused = copy.deepcopy(as_)
    ws = rte.Workspace(as_.dims[1], as_.dims[0], np.float64)
    GA.prepare_atmosphere(ws, fused, TEST_PARAMETERS, lw, GA.UniformZ, GA.UseSurfaceTempAtBottom, True)
    GA.interpolate_levels(ws, as_, GA.UniformZ, GA.UseSurfaceTempAtBottom, TEST_PARAMETERS,
                          isothermal_boundary_layer=True)
    GA.add_isothermal_boundary_layer(ws, as_, lw.p_ref_min, TEST_PARAMETERS)
    GA.clip(ws, as_, lw.p_ref_min, TEST_PARAMETERS, lw.idx_h2o, lw.t_ref_min, lw.t_ref_max)
    GA.update_concentrations(ws, as_, TEST_PARAMETERS, lw.idx_h2o)
    _assert_state_close(as_, fused, 0)
    # device-resident state (torch tensors): same bits as the host-staged run
    dev, _, _ = _perturbed_columns(np.float64, "gm", True)
    dev = dev.to_device("cuda:0")
    GA.prepare_atmosphere(ws, dev, TEST_PARAMETERS, lw, GA.UniformZ, GA.UseSurfaceTempAtBottom, True)
    ws.synchronize()
    torch.cuda.synchronize()
    _assert_state_close(dev.to_host(), fused, 0)
    # gray state: interpolation + isothermal layer; pressures clipped at 0; no temperature clamp
    ncol, nlay = 6, 10
    rng = np.random.default_rng(3)
    p_lay = np.asfortranarray(np.sort(rng.uniform(1e3, 1e5, (nlay, ncol)), axis=0)[::-1].copy())
    t_lay = np.asfortranarray(np.linspace(290, 120, nlay)[:, None] + rng.uniform(-2, 2, (nlay, ncol)))
    z = lambda *s: np.zeros(s, order="F")   # noqa: E731
    gs = GrayAtmosphericState(z(ncol), p_lay, z(nlay + 1, ncol), t_lay, z(nlay + 1, ncol), z(nlay + 1, ncol),
                              np.full(ncol, 295.0), GrayOpticalThicknessSchneider2004())
    ref = copy.deepcopy(gs)
    oracle.prepare_atmosphere(ref, TEST_PARAMETERS, _abi.PREP_ALL, interpolation=GA.GeometricMean,
                              isothermal_boundary_layer=True, p_min=0.0)
    wsg = rte.Workspace(ncol, nlay, np.float64)
    GA.prepare_atmosphere(wsg, gs, TEST_PARAMETERS, None, GA.GeometricMean, isothermal_boundary_layer=True)
    for n in ("p_lay", "p_lev", "t_lay", "t_lev"):
        np.testing.assert_allclose(getattr(gs, n), getattr(ref, n), rtol=1e-13, err_msg=n)
    assert gs.t_lev.min() < 160.0 and np.all(gs.p_lev[-1] == 0.0)
    with pytest.raises(ValueError, match="lookup_lw"):
        GA.prepare_atmosphere(ws, as_, TEST_PARAMETERS)
