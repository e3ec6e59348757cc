"""Golden fixtures (tests/golden/*.npz, made by tests/golden/make_golden.py) and the
cross-check of the two independent CPU restatements (C oracle vs numpy)."""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "golden"))
import make_golden as G  # noqa: E402
from oracle import np_oracle as NP  # noqa: E402
from oracle import oracle as O  # noqa: E402

HERE = os.path.join(os.path.dirname(__file__), "golden")


@pytest.mark.parametrize("case", list(G.CASES))
def test_c_oracle_reproduces_golden(case):
    exp = np.load(os.path.join(HERE, f"{case}.npz"))
    got = G.run(case, O.solve_lw, O.solve_sw)
    assert set(exp.files) == set(got)
    for k in exp.files:
        np.testing.assert_allclose(got[k], exp[k], rtol=1e-12, atol=1e-12, err_msg=k)


@pytest.mark.parametrize("case", list(G.CASES))
def test_numpy_restatement_agrees_with_golden(case):
    """Two transcriptions of the reference that share no code must agree to rounding — every golden case: clear sky,
    clouds with the full Vmr, MERRA aerosols under a fractional-cloud McICA sample, ice roughness classes 1 and 3."""
    exp = np.load(os.path.join(HERE, f"{case}.npz"))
    t = G.tables()
    as_, lb, sb = G.inputs(case)
    cl = t["cld_lw"] if as_.cloud_state is not None else None
    cs = t["cld_sw"] if as_.cloud_state is not None else None
    al = t["aero_lw"] if G.CASES[case]["aero"] else None
    asw = t["aero_sw"] if G.CASES[case]["aero"] else None
    kw = dict(seed=11)   # make_golden.run
    up, dn = NP.solve_lw_2stream(t["lw"], as_, lb, cl, lka=al, **kw)
    np.testing.assert_allclose(up, exp["lw2s_up"], rtol=1e-11, atol=1e-10)
    np.testing.assert_allclose(dn, exp["lw2s_dn"], rtol=1e-11, atol=1e-10)
    up, dn = NP.solve_lw_noscat(t["lw"], as_, lb, cl, lka=al, **kw)
    np.testing.assert_allclose(up, exp["lwns_up"], rtol=1e-11, atol=1e-10)
    np.testing.assert_allclose(dn, exp["lwns_dn"], rtol=1e-11, atol=1e-10)
    # three Gauss-Jacobi-5 angles (literal table of src/optics/AngularDiscretizations.jl:47-49): the flux is the
    # weighted sum of the one-angle solves (longwave_noscat.jl:45-96)
    mu3 = (0.1024922169, 0.4417960320, 0.8633751621)
    w3 = (0.0437820218, 0.3875796738, 0.5686383044)
    up = dn = 0.0
    for mu, w in zip(mu3, w3):
        u, d = NP.solve_lw_noscat(t["lw"], as_, lb, cl, Ds=1.0 / mu, w=w, lka=al, **kw)
        up, dn = up + u, dn + d
    np.testing.assert_allclose(up, exp["lwns3_up"], rtol=1e-11, atol=1e-10)
    np.testing.assert_allclose(dn, exp["lwns3_dn"], rtol=1e-11, atol=1e-10)
    diag = {}
    up, dn, dr = NP.solve_sw_2stream(t["sw"], as_, sb, cs, lka=asw, diag=diag, **kw)
    np.testing.assert_allclose(up, exp["sw_up"], rtol=1e-11, atol=1e-10)
    np.testing.assert_allclose(dn, exp["sw_dn"], rtol=1e-11, atol=1e-10)
    np.testing.assert_allclose(dr, exp["sw_dir"], rtol=1e-11, atol=1e-10)
    if "cover_sw" in exp.files:      # McICA effective cloud cover: the same sample, g-point by g-point
        np.testing.assert_array_equal(diag["cover"], exp["cover_sw"])
    if "aod_ext" in exp.files:       # 550 nm AOD of aerosol_optics.jl:96-116
        np.testing.assert_allclose(diag["aod_ext"], exp["aod_ext"], rtol=1e-12)


def test_numpy_mcica_masks_equal_the_c_oracle_for_fractional_cloud_fractions():
    """cloud_optics.jl:264-334 twice: the same masks bit for bit, Float64 and Float32 cloud fractions (the comparison
    `draw >= FT(1) - cld_frac` is taken in the working precision's 1 - cld_frac), LW and SW streams, edge fractions."""
    rng = np.random.default_rng(3)
    for ft in (np.float64, np.float32):
        for trial in range(6):
            nlay = 19
            cf = rng.uniform(0, 1, nlay)
            cf[rng.uniform(size=nlay) < 0.4] = 0.0
            cf[rng.uniform(size=nlay) < 0.1] = 1.0
            if trial == 0:
                cf[:] = 0.0
            if trial == 1:
                cf[:] = 0.0; cf[7] = 1e-7
            cf = cf.astype(ft)
            for is_sw in (0, 1):
                got = NP.cloud_mask_column(cf, seed=5 + trial, gcol=40 + trial, ngpt=24, is_sw=is_sw)
                for g in range(24):
                    ref, any_ = O.build_cloud_mask(cf, 5 + trial, 40 + trial, g + 1, is_sw)
                    np.testing.assert_array_equal(got[:, g], ref, err_msg=f"{ft} {trial} {g}")
                    assert any_ == got[:, g].any()
    assert NP.mcica_uniform(9, 3, 7, 1, 2) == O.mcica_uniform(9, 3, 7, 1, 2)


def test_numpy_sw_noscat_gray_and_column_amounts_agree_with_the_c_oracle():
    """The remaining single-transcription pieces, twice: rte_sw_noscat! (shortwave_noscat.jl:120-148), the gray optics and
    the four gray solvers (gray_optics_kernels.jl:12-251), compute_col_gas / compute_relative_humidity
    (gas_optics.jl:16-80)."""
    from rrtmgp_jl_amd.states import (GrayOpticalThicknessOGorman2008, GrayOpticalThicknessSchneider2004, LwBCs,
                                      RRTMGPParameters, SwBCs)
    t = G.tables()
    as_, lb, sb = G.inputs("clear_gm")
    f = O.solve_sw(as_, sb, t["sw"], twostream=False)
    up, dn, dr = NP.solve_sw_noscat(t["sw"], as_, sb)
    np.testing.assert_allclose(dn, f.as_nlev_ncol("flux_dn"), rtol=1e-12, atol=1e-11)
    np.testing.assert_allclose(dr, f.as_nlev_ncol("flux_dn_dir"), rtol=1e-12, atol=1e-11)
    assert (f.as_nlev_ncol("flux_up") == 0).all() and (up == 0).all()
    params = RRTMGPParameters()
    ncol, nlay = 5, 30
    lat = np.linspace(-80.0, 80.0, ncol)
    lbg = LwBCs(np.full((1, ncol), 0.97, order="F"), None)
    mu0 = np.full(ncol, 0.6); mu0[1] = 0.0; mu0[2] = -0.3
    sbg = SwBCs(mu0, np.full(ncol, 1407.679), np.full((1, ncol), 0.1, order="F"), np.full((1, ncol), 0.2, order="F"))
    for otp in (GrayOpticalThicknessSchneider2004(), GrayOpticalThicknessOGorman2008()):
        gs = O.setup_gray_as_pr_grid(nlay, lat, 100000.0, 9000.0, otp, params, np.float64)
        for two in (True, False):
            ref = O.solve_lw_gray(gs, lbg, twostream=two)
            up, dn = NP.solve_lw_gray(gs, lbg, two)
            np.testing.assert_allclose(up, ref.as_nlev_ncol("flux_up"), rtol=1e-12, atol=1e-11)
            np.testing.assert_allclose(dn, ref.as_nlev_ncol("flux_dn"), rtol=1e-12, atol=1e-11)
            ref = O.solve_sw_gray(gs, sbg, twostream=two)
            up, dn, dr = NP.solve_sw_gray(gs, sbg, two)
            np.testing.assert_allclose(up, ref.as_nlev_ncol("flux_up"), rtol=1e-12, atol=1e-11)
            np.testing.assert_allclose(dn, ref.as_nlev_ncol("flux_dn"), rtol=1e-12, atol=1e-11)
            np.testing.assert_allclose(dr, ref.as_nlev_ncol("flux_dn_dir"), rtol=1e-12, atol=1e-11)
    h2o = as_.vmr.vmr_h2o
    np.testing.assert_allclose(NP.compute_col_gas(as_.p_lev, params, h2o, as_.lat),
                               O.compute_col_gas(as_.p_lev, params, h2o, as_.lat), rtol=1e-14)
    np.testing.assert_allclose(NP.compute_col_gas(as_.p_lev, params), O.compute_col_gas(as_.p_lev, params), rtol=1e-14)
    p_lay, t_lay = np.asfortranarray(as_.layerdata[1]), np.asfortranarray(as_.layerdata[2])
    np.testing.assert_allclose(NP.compute_relative_humidity(p_lay, t_lay, params, h2o),
                               O.compute_relative_humidity(p_lay, t_lay, params, h2o), rtol=1e-13)


@pytest.mark.gpu
@pytest.mark.parametrize("case", list(G.CASES))
def test_hip_reproduces_golden(case):
    from rrtmgp_jl_amd import rte

    def lw(as_, bcs, lk, cld, aero, twostream, seed, n_gauss_angles=1):
        nlay, ncol = as_.dims
        cls = rte.TwoStreamLWRTE if twostream else rte.NoScatLWRTE
        return rte.solve_lw(cls(ncol, nlay, as_.dtype, bcs, n_gauss_angles=n_gauss_angles), as_, lk, cld, aero, seed=seed)

    def sw(as_, bcs, lk, cld, aero, seed):
        nlay, ncol = as_.dims
        return rte.solve_sw(rte.TwoStreamSWRTE(ncol, nlay, as_.dtype, bcs), as_, lk, cld, aero, seed=seed)
    exp = np.load(os.path.join(HERE, f"{case}.npz"))
    got = G.run(case, lw, sw)
    for k in exp.files:
        np.testing.assert_allclose(got[k], exp[k], rtol=1e-10, atol=1e-8, err_msg=k)
