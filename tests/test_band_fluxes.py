"""N3: per-band fluxes (FluxBand, src/optics/Fluxes.jl:170-215; spectral_* getters,
src/api/getters.jl:398-470).  CPU: the oracle's band buffers against an independent
g-point-by-g-point numpy restatement and the sum-over-bands identity.  GPU: HIP parity with
the oracle through the C ABI, metric scaling, night columns, the L2 getters, and the
ragged bands (not whole 16-g-point groups) and the one remaining limit (padded bands must fit 256 lanes)."""
import numpy as np
import pytest

from rrtmgp_jl_amd import synthetic as S
from rrtmgp_jl_amd.states import FluxBand
from oracle import np_oracle, oracle

NCOL, NLAY = 7, 20


def _case(t, FT=np.float64, **kw):
    as_, lb, sb = S.make_columns(NCOL, NLAY, FT, seed=23, n_bnd_lw=t["lw"].n_bnd, n_bnd_sw=t["sw"].n_bnd,
                                 night_fraction=0.3, random_cld_frac=True, **kw)
    rng = np.random.default_rng(4)
    metric = np.asfortranarray(rng.uniform(0.95, 1.05, (NLAY + 1, NCOL)).astype(FT))
    return as_, lb, sb, metric


def test_oracle_band_fluxes_sum_to_broadband_and_match_numpy(small_tables64):
    t = small_tables64
    as_, lb, sb, metric = _case(t, clouds=False)
    for solve, lk, bcs, sw in ((oracle.solve_lw, t["lw"], lb, False), (oracle.solve_sw, t["sw"], sb, True)):
        bf = FluxBand.allocate(NCOL, NLAY + 1, lk.n_bnd, np.float64)
        f = solve(as_, bcs, lk, band_flux=bf, metric_scaling=metric)
        for n in ("flux_up", "flux_dn", "flux_net"):
            np.testing.assert_allclose(getattr(bf, n).sum(axis=2), getattr(f, n), rtol=0, atol=1e-11)
        np.testing.assert_array_equal(bf.flux_net, bf.flux_up - bf.flux_dn)
        if sw:   # night columns stay zero (shortwave_2stream.jl:41-46: no accumulate when mu0 <= 0)
            night = sb.cos_zenith <= 0
            assert night.any() and not bf.flux_up[:, night].any() and not bf.flux_dn[:, night].any()
    # independent restatement: one column, per-g-point numpy solver summed into bands
    icol = 2
    lk = t["lw"]
    bf = FluxBand.allocate(NCOL, NLAY + 1, lk.n_bnd, np.float64)
    oracle.solve_lw(as_, lb, lk, band_flux=bf)
    up, dn = np_oracle.solve_lw_2stream(lk, as_, lb, per_gpoint=True)
    for b in range(lk.n_bnd):
        sel = lk.major_gpt2bnd == b + 1
        np.testing.assert_allclose(bf.flux_up[:, icol, b], up[:, icol, sel].sum(axis=1), rtol=1e-11, atol=1e-11)
        np.testing.assert_allclose(bf.flux_dn[:, icol, b], dn[:, icol, sel].sum(axis=1), rtol=1e-11, atol=1e-11)


def test_oracle_rejects_band_fluxes_for_noscat(small_tables64):
    t = small_tables64
    as_, lb, sb, _ = _case(t, clouds=False)
    bf = FluxBand.allocate(NCOL, NLAY + 1, t["lw"].n_bnd, np.float64)
    with pytest.raises(Exception):
        oracle.solve_lw(as_, lb, t["lw"], twostream=False, band_flux=bf)


@pytest.mark.gpu
@pytest.mark.parametrize("FT,tol_lw,tol_sw", [(np.float64, 1e-10, 1e-10), (np.float32, 2e-3, 3e-2)])
def test_hip_band_fluxes_match_oracle(tables64, FT, tol_lw, tol_sw):
    """Float32 tolerances are the F32-vs-F64 budgets of test_gpu_parity (W/m2 per band <= broadband)."""
    from rrtmgp_jl_amd import rte
    t64 = tables64
    t = {k: v.astype(FT) for k, v in t64.items()}
    as64, lb64, sb64, m64 = _case(t64, aerosols=True)
    as_, lb, sb, metric = _case(t, FT, aerosols=True)
    for sw in (False, True):
        lk, cld, aero = (t["sw"], t["cld_sw"], t["aero_sw"]) if sw else (t["lw"], t["cld_lw"], t["aero_lw"])
        lk64, cld64, aero64 = (t64["sw"], t64["cld_sw"], t64["aero_sw"]) if sw else (t64["lw"], t64["cld_lw"], t64["aero_lw"])
        ref_b = FluxBand.allocate(NCOL, NLAY + 1, lk.n_bnd, np.float64)
        ref = (oracle.solve_sw if sw else oracle.solve_lw)(as64, sb64 if sw else lb64, lk64, cld64, aero64,
                                                           band_flux=ref_b, metric_scaling=m64, seed=5)
        cls = rte.TwoStreamSWRTE if sw else rte.TwoStreamLWRTE
        slv = cls(NCOL, NLAY, FT, sb if sw else lb, n_bnd_band_flux=lk.n_bnd)
        f = (rte.solve_sw if sw else rte.solve_lw)(slv, as_, lk, cld, aero, metric_scaling=metric, seed=5)
        tol = tol_sw if sw else tol_lw
        for n in ("flux_up", "flux_dn", "flux_net"):
            got = np.asarray(getattr(slv.band_flux, n), dtype=np.float64)
            assert np.abs(got - getattr(ref_b, n)).max() <= tol, (sw, n)
            # broadband results are unchanged by the band accumulators and equal the band sum
            assert np.abs(np.asarray(getattr(f, n), dtype=np.float64) - getattr(ref, n)).max() <= tol
            assert np.abs(got.sum(axis=2) - np.asarray(getattr(f, n), dtype=np.float64)).max() <= (1e-10 if FT is np.float64 else 2e-3)
        if sw:
            night = sb.cos_zenith <= 0
            assert night.any() and not np.asarray(slv.band_flux.flux_up)[:, night].any()


def _band_parity(t, FT, tol_lw, tol_sw, **case_kw):
    from rrtmgp_jl_amd import rte
    t64 = t
    tF = {k: v.astype(FT) for k, v in t64.items()}
    as64, lb64, sb64, m64 = _case(t64, **case_kw)
    as_, lb, sb, metric = _case(tF, FT, **case_kw)
    for sw in (False, True):
        r = "sw" if sw else "lw"
        aero = case_kw.get("aerosols", False)
        cl = case_kw.get("clouds", True)
        lk, cld, ae = tF[r], tF["cld_" + r] if cl else None, tF["aero_" + r] if aero else None
        lk64, cld64, ae64 = t64[r], t64["cld_" + r] if cl else None, t64["aero_" + r] if aero else None
        ref_b = FluxBand.allocate(NCOL, NLAY + 1, lk.n_bnd, np.float64)
        ref = (oracle.solve_sw if sw else oracle.solve_lw)(as64, sb64 if sw else lb64, lk64, cld64, ae64,
                                                           band_flux=ref_b, metric_scaling=m64, seed=5)
        cls = rte.TwoStreamSWRTE if sw else rte.TwoStreamLWRTE
        slv = cls(NCOL, NLAY, FT, sb if sw else lb, n_bnd_band_flux=lk.n_bnd)
        f = (rte.solve_sw if sw else rte.solve_lw)(slv, as_, lk, cld, ae, metric_scaling=metric, seed=5)
        tol = tol_sw if sw else tol_lw
        for n in ("flux_up", "flux_dn", "flux_net"):
            got = np.asarray(getattr(slv.band_flux, n), dtype=np.float64)
            assert np.abs(got - getattr(ref_b, n)).max() <= tol, (sw, n)
            assert np.abs(np.asarray(getattr(f, n), dtype=np.float64) - getattr(ref, n)).max() <= tol, (sw, n)


@pytest.mark.gpu
@pytest.mark.parametrize("FT,tol_lw,tol_sw", [(np.float64, 1e-10, 1e-10), (np.float32, 2e-3, 3e-2)])
def test_hip_band_fluxes_ragged_bands(small_tables64, FT, tol_lw, tol_sw):
    """Bands that are not whole 16-g-point groups (8/4/12 and 6/10/4): the per-band variants lay the lanes out band
    by band on 16-lane rows, padding lanes idle."""
    _band_parity(small_tables64, FT, tol_lw, tol_sw, aerosols=True)


@pytest.mark.gpu
def test_hip_band_fluxes_reduced_gpoint_sets():
    """The shape of rrtmgp-data's reduced sets (rrtmgp-gas-lw-g128 / sw-g112: 16 / 14 bands of 4-12 g-points)."""
    rng = np.random.default_rng(11)
    n_lw, n_sw = 16, 14
    g_lw = rng.integers(4, 13, n_lw)
    g_sw = rng.integers(4, 13, n_sw)
    lw = S.make_gas_lookup("lw", np.float64, seed=3, gpt_per_bnd=[int(x) for x in g_lw])
    sw = S.make_gas_lookup("sw", np.float64, seed=3, gpt_per_bnd=[int(x) for x in g_sw])
    t = dict(lw=lw, sw=sw, cld_lw=S.make_cloud_lookup("lw", n_lw, seed=3), cld_sw=S.make_cloud_lookup("sw", n_sw, seed=3))
    _band_parity(t, np.float64, 1e-10, 1e-10)


@pytest.mark.gpu
def test_hip_band_fluxes_limits_and_device_memory(tables64, small_tables64):
    import torch
    from rrtmgp_jl_amd import rte, _lib
    # the one remaining limit: the bands, each padded to whole 16-lane rows, must fit the 256 lanes of a workgroup
    # (9 bands of 17 g-points need 18 rows); loud error, and the broadband solve of the same lookup is fine
    lw = S.make_gas_lookup("lw", np.float64, seed=5, n_bnd=9, gpt_per_bnd=17)
    as_, lb, sb = S.make_columns(NCOL, NLAY, np.float64, seed=23, n_bnd_lw=9, n_bnd_sw=3, clouds=False)
    slv = rte.TwoStreamLWRTE(NCOL, NLAY, np.float64, lb, n_bnd_band_flux=9)
    with pytest.raises(_lib.RRTMGPHipError, match="256 lanes"):
        rte.solve_lw(slv, as_, lw)
    f = rte.solve_lw(rte.TwoStreamLWRTE(NCOL, NLAY, np.float64, lb), as_, lw)
    ref = oracle.solve_lw(as_, lb, lw)
    assert np.abs(np.asarray(f.flux_up) - ref.flux_up).max() <= 1e-10
    with pytest.raises(ValueError, match="two-stream"):
        rte.NoScatLWRTE(NCOL, NLAY, np.float64, lb, n_bnd_band_flux=3)
    # device-resident band buffers (torch tensors), same numbers as host-staged
    t = tables64
    as_, lb, sb, _ = _case(t)
    host = rte.TwoStreamSWRTE(NCOL, NLAY, np.float64, sb, n_bnd_band_flux=t["sw"].n_bnd)
    rte.solve_sw(host, as_, t["sw"], t["cld_sw"], seed=9)
    dev = rte.TwoStreamSWRTE(NCOL, NLAY, np.float64, sb.to_device("cuda:0"), flux_device="cuda:0",
                             n_bnd_band_flux=t["sw"].n_bnd)
    rte.solve_sw(dev, as_.to_device("cuda:0"), t["sw"], t["cld_sw"], seed=9)
    dev.ws.synchronize()
    torch.cuda.synchronize()
    got = dev.band_flux.to_host()
    for n in ("flux_up", "flux_dn", "flux_net"):
        np.testing.assert_array_equal(getattr(got, n), getattr(host.band_flux, n))


@pytest.mark.gpu
def test_solver_spectral_getters(tables64):
    from rrtmgp_jl_amd import solver as L2
    from rrtmgp_jl_amd.states import TEST_PARAMETERS
    t = tables64
    as_, lb, sb, _ = _case(t)
    lookups = L2.LookupBundle(t["lw"], t["sw"], t["cld_lw"], t["cld_sw"], t["aero_lw"], t["aero_sw"])
    s = L2.RRTMGPSolver(L2.AllSkyRadiation(), TEST_PARAMETERS, lb, sb, as_, lookups=lookups, spectral_fluxes=True)
    L2.update_fluxes(s)
    assert L2.spectral_lw_flux_up(s).shape == (NLAY + 1, NCOL, t["lw"].n_bnd)
    assert L2.spectral_sw_flux_dn(s).shape == (NLAY + 1, NCOL, t["sw"].n_bnd)
    np.testing.assert_allclose(L2.spectral_lw_flux_net(s).sum(axis=2), L2.lw_flux_net(s), rtol=0, atol=1e-10)
    np.testing.assert_allclose(L2.spectral_sw_flux_up(s).sum(axis=2), L2.sw_flux_up(s), rtol=0, atol=1e-10)
    assert L2.lw_band_bounds(s).shape == (2, t["lw"].n_bnd)
    plain = L2.RRTMGPSolver(L2.AllSkyRadiation(), TEST_PARAMETERS, lb, sb, as_, lookups=lookups)
    with pytest.raises(ValueError, match="spectral_fluxes = true"):
        L2.spectral_lw_flux_up(plain)
    with pytest.raises(ValueError, match="two-stream optics"):
        L2.RRTMGPSolver(L2.ClearSkyRadiation(), TEST_PARAMETERS, lb, sb, as_, lookups=lookups, op_lw="onescalar",
                        spectral_fluxes=True)
