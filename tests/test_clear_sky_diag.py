"""N3: AllSkyRadiationWithClearSkyDiagnostics in ONE launch per band (the reference solves twice,
src/api/update_fluxes.jl:39-65,101-128).  The oracle implements the reference's two solves; the
HIP kernels carry the clear-sky recurrences next to the all-sky ones.  Both must agree, and the
one-pass result must equal two separate HIP solves."""
import numpy as np
import pytest

from rrtmgp_jl_amd import synthetic as S
from rrtmgp_jl_amd.states import Flux, FluxBand
from oracle import oracle

NCOL, NLAY = 13, 33
NAMES_LW = ("flux_up", "flux_dn", "flux_net")
NAMES_SW = NAMES_LW + ("flux_dn_dir",)


def _case(FT, aerosols, **kw):
    as_, lb, sb = S.make_columns(NCOL, NLAY, FT, seed=41, aerosols=aerosols, night_fraction=0.25,
                                 random_cld_frac=True, **kw)
    metric = np.asfortranarray(np.random.default_rng(2).uniform(0.97, 1.03, (NLAY + 1, NCOL)).astype(FT))
    return as_, lb, sb, metric


def test_oracle_clear_flux_is_the_cloudless_solve(small_tables64):
    t = small_tables64
    as_, lb, sb = S.make_columns(6, 12, seed=4, n_bnd_lw=3, n_bnd_sw=3, random_cld_frac=True, night_fraction=0.3)
    for sw in (False, True):
        solve, lk, cld, bcs = (oracle.solve_sw, t["sw"], t["cld_sw"], sb) if sw else (oracle.solve_lw, t["lw"], t["cld_lw"], lb)
        clear = Flux.allocate(6, 13, np.float64, sw=sw)
        both = solve(as_, bcs, lk, cld, seed=3, clear_flux=clear)
        want_clear, want_all = solve(as_, bcs, lk, None, seed=3), solve(as_, bcs, lk, cld, seed=3)
        for n in (NAMES_SW if sw else NAMES_LW):
            np.testing.assert_array_equal(getattr(clear, n), getattr(want_clear, n))
            np.testing.assert_array_equal(getattr(both, n), getattr(want_all, n))
        if not sw:   # clear-sky OLR >= all-sky OLR (all_sky_with_aerosols_utils.jl:190-197)
            assert np.all(clear.flux_up[-1] >= both.flux_up[-1] - 1e-9)
    with pytest.raises(Exception):   # needs a cloud lookup
        oracle.solve_lw(as_, lb, t["lw"], None, clear_flux=Flux.allocate(6, 13, np.float64))


@pytest.mark.gpu
@pytest.mark.parametrize("FT,tol_lw,tol_sw", [(np.float64, 1e-9, 1e-9), (np.float32, 2e-3, 1.2e-1)])
@pytest.mark.parametrize("aerosols", [False, True])
def test_hip_one_pass_matches_oracle_and_two_solves(tables64, FT, tol_lw, tol_sw, aerosols):
    """Float32 tolerances: the reference's own F32-vs-F64 ratchet (test/float32_consistency.jl:53-62)."""
    from rrtmgp_jl_amd import rte
    t64 = tables64
    t = {k: v.astype(FT) for k, v in t64.items()}
    as64, lb64, sb64, m64 = _case(np.float64, aerosols)
    as_, lb, sb, metric = _case(FT, aerosols)
    for sw in (False, True):
        sfx = "sw" if sw else "lw"
        lk, cld, aero = t[sfx], t["cld_" + sfx], t["aero_" + sfx] if aerosols else None
        names, tol = (NAMES_SW, tol_sw) if sw else (NAMES_LW, tol_lw)
        ref_clear = Flux.allocate(NCOL, NLAY + 1, np.float64, sw=sw)
        ref = (oracle.solve_sw if sw else oracle.solve_lw)(as64, sb64 if sw else lb64, t64[sfx], t64["cld_" + sfx],
                                                           t64["aero_" + sfx] if aerosols else None, seed=8,
                                                           metric_scaling=m64, clear_flux=ref_clear)
        cls, solve, bcs = (rte.TwoStreamSWRTE, rte.solve_sw, sb) if sw else (rte.TwoStreamLWRTE, rte.solve_lw, lb)
        one = cls(NCOL, NLAY, FT, bcs)
        clear = Flux.allocate(NCOL, NLAY + 1, FT, sw=sw)
        solve(one, as_, lk, cld, aero, metric_scaling=metric, seed=8, clear_flux=clear)
        for n in names:
            assert np.abs(np.float64(getattr(one.flux, n)) - getattr(ref, n)).max() <= tol, (sfx, "all-sky", n)
            assert np.abs(np.float64(getattr(clear, n)) - getattr(ref_clear, n)).max() <= tol, (sfx, "clear", n)
        # against two separate launches of the same library: identical arithmetic per stream
        two_all, two_clear = cls(NCOL, NLAY, FT, bcs), cls(NCOL, NLAY, FT, bcs)
        solve(two_all, as_, lk, cld, aero, metric_scaling=metric, seed=8)
        solve(two_clear, as_, lk, None, aero, metric_scaling=metric, seed=8)
        eps = 1e-11 if FT is np.float64 else 2e-4
        for n in names:
            assert np.abs(np.float64(getattr(one.flux, n)) - np.float64(getattr(two_all.flux, n))).max() <= eps
            assert np.abs(np.float64(getattr(clear, n)) - np.float64(getattr(two_clear.flux, n))).max() <= eps
        if sw:
            night = sb.cos_zenith <= 0
            assert night.any() and not clear.flux_dn[:, night].any() and not clear.flux_up[:, night].any()
        # cloud cover is still written by the same launch
        cov = as_.cloud_state.cld_cover_sw if sw else as_.cloud_state.cld_cover_lw
        assert np.all((cov >= 0) & (cov <= 1)) and cov.max() > 0


@pytest.mark.gpu
@pytest.mark.parametrize("aerosols", [False, True])
def test_float32_instances_of_four_waves_per_simd(tables64, aerosols, monkeypatch, capfd):
    """Float32 launches with more columns than the 168-VGPR instances keep resident take the 128-VGPR / 8-layer-chunk
    instances of the one-pass diagnostic (csrc/solve_lw.hip, solve_sw.hip: launch_*).  The other tests of this file are too
    small to reach them: 1 100 columns against the Float32 oracle, budgets of the reference's Float32 ratchet."""
    from rrtmgp_jl_amd import rte
    FT, ncol, nlay = np.float32, 1100, 20
    t = {k: v.astype(FT) for k, v in tables64.items()}
    as_, lb, sb = S.make_columns(ncol, nlay, FT, seed=77, aerosols=aerosols, night_fraction=0.2, random_cld_frac=True)
    monkeypatch.setenv("RRTMGP_HIP_TRACE_LAUNCH", "1")
    for sw in (False, True):
        sfx = "sw" if sw else "lw"
        lk, cld, aero = t[sfx], t["cld_" + sfx], t["aero_" + sfx] if aerosols else None
        names, tol = (NAMES_SW, 1.2e-1) if sw else (NAMES_LW, 1e-3)
        ref_clear = Flux.allocate(ncol, nlay + 1, FT, sw=sw)
        ref = (oracle.solve_sw if sw else oracle.solve_lw)(as_, sb if sw else lb, lk, cld, aero, seed=8, clear_flux=ref_clear)
        cls, solve, bcs = (rte.TwoStreamSWRTE, rte.solve_sw, sb) if sw else (rte.TwoStreamLWRTE, rte.solve_lw, lb)
        one = cls(ncol, nlay, FT, bcs)
        clear = Flux.allocate(ncol, nlay + 1, FT, sw=sw)
        capfd.readouterr()
        solve(one, as_, lk, cld, aero, seed=8, clear_flux=clear)
        trace = capfd.readouterr().err
        assert "-> 4 workgroups per CU" in trace and "-> 3 workgroups per CU" in trace, trace   # both asked, the larger taken
        for n in names:
            assert np.abs(np.float64(getattr(one.flux, n)) - np.float64(getattr(ref, n))).max() <= tol, (sfx, "all-sky", n)
            assert np.abs(np.float64(getattr(clear, n)) - np.float64(getattr(ref_clear, n))).max() <= tol, (sfx, "clear", n)
        # the first 700 columns alone are few enough for the 168-VGPR instances: the same arithmetic per (layer, g-point)
        from rrtmgp_jl_amd.sharding import shard_container
        as_p, bcs_p = shard_container(as_, 0, 700, ncol), shard_container(bcs, 0, 700, ncol)
        part, pclear = cls(700, nlay, FT, bcs_p), Flux.allocate(700, nlay + 1, FT, sw=sw)
        solve(part, as_p, lk, cld, aero, seed=8, clear_flux=pclear)
        for n in names:
            np.testing.assert_array_equal(getattr(part.flux, n), getattr(one.flux, n)[:, :700])
            np.testing.assert_array_equal(getattr(pclear, n), getattr(clear, n)[:, :700])


@pytest.mark.gpu
def test_hip_one_pass_argument_errors(tables64):
    from rrtmgp_jl_amd import rte, _lib
    t = tables64
    as_, lb, sb, _ = _case(np.float64, False)
    clear = Flux.allocate(NCOL, NLAY + 1, np.float64)
    with pytest.raises(_lib.RRTMGPHipError, match="cloud lookup"):      # clear-sky solve has nothing to add
        rte.solve_lw(rte.TwoStreamLWRTE(NCOL, NLAY, np.float64, lb), as_, t["lw"], None, clear_flux=clear)
    with pytest.raises(_lib.RRTMGPHipError, match="two-stream"):
        rte.solve_lw(rte.NoScatLWRTE(NCOL, NLAY, np.float64, lb), as_, t["lw"], t["cld_lw"], clear_flux=clear)
    banded = rte.TwoStreamLWRTE(NCOL, NLAY, np.float64, lb, n_bnd_band_flux=t["lw"].n_bnd)
    with pytest.raises(_lib.RRTMGPHipError, match="cannot be combined"):
        rte.solve_lw(banded, as_, t["lw"], t["cld_lw"], clear_flux=clear)
