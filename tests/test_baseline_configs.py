"""The five configurations of BASELINE.json, one test each, through the C ABI on the GPU.

configs[0] gray LW no-scattering, ncol = 1, nlay = 60, Float64 (test/gray_atm.jl plumbing)
configs[1] RFMIP-shaped clear sky, 100 columns x 60 layers, LW + SW no-scattering, Float64
configs[2] cloudy two-stream LW + SW with McICA, 128 columns x 64 layers, Float32
configs[3] all-sky + MERRA aerosols, 4096 columns x 73 layers, two-stream, sharded 1 -> 8 ways
configs[4] GCM scale, 1 048 576 columns x 64 layers, Float32, column-sharded over 8 GPUs
           (here: one GPU's 131 072-column shard of that global problem)
Tolerances: Float64 1e-8 W/m2 against the oracle; Float32 inside the reference's F32 budget
(test/float32_consistency.jl:53-62: LW 1e-3, SW 3e-2 clear / 1.2e-1 cloudy)."""
import numpy as np
import pytest

from oracle import oracle as O
from rrtmgp_jl_amd import rte, sharding, synthetic as S
from rrtmgp_jl_amd.states import GrayOpticalThicknessSchneider2004, LwBCs, RRTMGPParameters

pytestmark = pytest.mark.gpu
LWN, SWN = ("flux_up", "flux_dn", "flux_net"), ("flux_up", "flux_dn", "flux_net", "flux_dn_dir")


def _maxdiff(a, b, names):
    return max(float(np.abs(np.float64(getattr(a, n)) - np.float64(getattr(b, n))).max()) for n in names)


def test_config0_gray_lw_noscat_single_column():
    params = RRTMGPParameters()
    gs = O.setup_gray_as_pr_grid(60, np.array([30.0]), 100000.0, 9000.0, GrayOpticalThicknessSchneider2004(), params,
                                 np.float64)
    lb = LwBCs(np.ones((1, 1), order="F"), None)
    out = rte.solve_lw(rte.NoScatLWRTE(1, 60, np.float64, lb), gs)
    assert _maxdiff(out, O.solve_lw_gray(gs, lb, twostream=False), LWN) < 1e-9


def test_config1_clear_sky_100x60_noscat_f64(tables64):
    t = tables64
    as_, lb, sb = S.make_columns(100, 60, np.float64, seed=1, clouds=False, night_fraction=0.1)
    lw = rte.solve_lw(rte.NoScatLWRTE(100, 60, np.float64, lb), as_, t["lw"])
    sw = rte.solve_sw(rte.NoScatSWRTE(100, 60, np.float64, sb), as_, t["sw"])
    assert _maxdiff(lw, O.solve_lw(as_, lb, t["lw"], twostream=False), LWN) < 1e-8
    assert _maxdiff(sw, O.solve_sw(as_, sb, t["sw"], twostream=False), SWN) < 1e-8


def test_config2_cloudy_mcica_128x64_f32(tables32):
    """Against the Float32 oracle on the same Float32 inputs: with fractional cloudiness a Float64 run of
    the same case draws different McICA masks wherever a draw falls between the two roundings of 1 - cld_frac,
    so the F32-vs-F64 comparison is only meaningful for overcast / clear columns (tests/test_gpu_parity.py)."""
    t = tables32
    as_, lb, sb = S.make_columns(128, 64, np.float32, seed=2, random_cld_frac=True, cos_zenith=0.86)
    ref_as = S.make_columns(128, 64, np.float32, seed=2, random_cld_frac=True, cos_zenith=0.86)[0]
    lw = rte.solve_lw(rte.TwoStreamLWRTE(128, 64, np.float32, lb), as_, t["lw"], t["cld_lw"], seed=7)
    sw = rte.solve_sw(rte.TwoStreamSWRTE(128, 64, np.float32, sb), as_, t["sw"], t["cld_sw"], seed=7)
    assert _maxdiff(lw, O.solve_lw(ref_as, lb, t["lw"], t["cld_lw"], seed=7), LWN) < 1e-3
    assert _maxdiff(sw, O.solve_sw(ref_as, sb, t["sw"], t["cld_sw"], seed=7), SWN) < 2e-2
    # the McICA sample is the oracle's: identical cloud cover (counts of cloudy g-points)
    np.testing.assert_array_equal(as_.cloud_state.cld_cover_lw, ref_as.cloud_state.cld_cover_lw)
    np.testing.assert_array_equal(as_.cloud_state.cld_cover_sw, ref_as.cloud_state.cld_cover_sw)
    assert 0 < as_.cloud_state.cld_cover_lw.max() <= 1


def test_config2_full_vmr_overcast_vs_float64_oracle(tables32, tables64):
    """SURVEY §8(d) config 3 as the reference test builds it (test/read_cloudy_sky.jl:70-84): the full `Vmr`
    storage (19 gases x nlay x ncol) and cld_frac = 1, so that the Float32 run and a Float64 run draw the same
    McICA masks and the reference's own F32 <-> F64 ratchet applies (test/float32_consistency.jl:53-62:
    LW 1e-3, cloudy SW 1.2e-1 W/m2)."""
    kw = dict(seed=2, vmr_kind="full", cld_frac=1.0, cos_zenith=0.86)
    as32, lb32, sb32 = S.make_columns(128, 64, np.float32, **kw)
    assert as32.vmr.vmr.shape[0] == 19
    as64, lb64, sb64 = S.make_columns(128, 64, np.float64, **kw)
    lw = rte.solve_lw(rte.TwoStreamLWRTE(128, 64, np.float32, lb32), as32, tables32["lw"], tables32["cld_lw"], seed=7)
    sw = rte.solve_sw(rte.TwoStreamSWRTE(128, 64, np.float32, sb32), as32, tables32["sw"], tables32["cld_sw"], seed=7)
    ref_lw = O.solve_lw(as64, lb64, tables64["lw"], tables64["cld_lw"], seed=7)
    ref_sw = O.solve_sw(as64, sb64, tables64["sw"], tables64["cld_sw"], seed=7)
    assert _maxdiff(lw, ref_lw, LWN) < 1e-3
    # Cloudy SW: the ratchet (1.2e-1) is what the reference measured on ITS test column.  On these 128 columns the
    # reference algorithm itself, run in Float32 on the CPU, is 0.33 W/m2 away from its Float64 run in one nearly
    # conservative cloud layer (k_min = sqrt(eps) is of the working precision, src/Numerics.jl:24).  The device must
    # be as close to the Float64 reference as the reference's own Float32 arithmetic is, and within the F32 <-> F32
    # budget of that Float32 run.
    cpu32 = O.solve_sw(S.make_columns(128, 64, np.float32, **kw)[0], sb32, tables32["sw"], tables32["cld_sw"], seed=7)
    assert _maxdiff(sw, cpu32, SWN) < 2e-2                       # F32 <-> F32: the device against the reference arithmetic
    # F32 <-> F64, column by column: the reference's fixed 1.2e-1 FIRST.  A column may exceed it only where the reference
    # algorithm in Float32 on the CPU exceeds it too (its inherent Float32 error there), and then by no more than that
    # error + the F32 <-> F32 budget; such columns are named, so a regression cannot hide behind the oracle's own error.
    def per_column(a, b):
        return np.max([np.abs(np.float64(getattr(a, n)) - np.float64(getattr(b, n))).max(axis=0) for n in SWN], axis=0)
    dev, inherent = per_column(sw, ref_sw), per_column(cpu32, ref_sw)
    over = np.flatnonzero(dev >= 1.2e-1)
    assert np.all(inherent[over] >= 1.2e-1), ("columns beyond the reference ratchet where the CPU Float32 run is inside it",
                                             over[inherent[over] < 1.2e-1], dev[over], inherent[over])
    assert np.all(dev[over] <= inherent[over] + 2e-2), (over, dev[over], inherent[over])
    assert len(over) <= 2, (over, dev[over], inherent[over])      # today: one nearly conservative cloud layer in one column
    if len(over):
        print(f"note: column(s) {over.tolist()} exceed the 1.2e-1 ratchet in Float32 on the CPU as well: "
              f"device {dev[over]}, CPU Float32 {inherent[over]} W/m2 from Float64")
    np.testing.assert_array_equal(as32.cloud_state.cld_cover_lw, as64.cloud_state.cld_cover_lw.astype(np.float32))
    np.testing.assert_array_equal(as32.cloud_state.cld_cover_sw, as64.cloud_state.cld_cover_sw.astype(np.float32))


@pytest.mark.parametrize("nlay", [72, 73])
def test_config3_oracle_parity_on_strided_columns(tables32, nlay):
    """The 4096-column all-sky + MERRA-aerosol case (72 layers is what the reference data set has,
    all_sky_with_aerosols_highres_gpu_benchmark.jl:285; BASELINE.json says 73): every flux component of LW and SW,
    the 550 nm AODs and both cloud covers of 32 strided columns against the Float32 oracle at the F32 budget."""
    t = dict(tables32)
    t["aero_sw"] = __import__("dataclasses").replace(t["aero_sw"], iband_550nm=10)
    ncol = 4096
    kw = dict(seed=3, aerosols=True, night_fraction=0.1, random_cld_frac=True)
    as_, lb, sb = S.make_columns(ncol, nlay, np.float32, **kw)
    lw = rte.solve_lw(rte.TwoStreamLWRTE(ncol, nlay, np.float32, lb), as_, t["lw"], t["cld_lw"], t["aero_lw"], seed=5)
    sw = rte.solve_sw(rte.TwoStreamSWRTE(ncol, nlay, np.float32, sb), as_, t["sw"], t["cld_sw"], t["aero_sw"], seed=5)
    worst_lw = worst_sw = 0.0
    n_day = 0
    for g in range(5, ncol, 128):                      # 32 columns; the synthetic generator is keyed by the global column
        a1, l1, s1 = S.make_columns(1, nlay, np.float32, col_offset=g, **kw)
        r_lw = O.solve_lw(a1, l1, t["lw"], t["cld_lw"], t["aero_lw"], seed=5, col_offset=g)
        r_sw = O.solve_sw(a1, s1, t["sw"], t["cld_sw"], t["aero_sw"], seed=5, col_offset=g)
        for n in LWN:
            worst_lw = max(worst_lw, float(np.abs(np.float64(getattr(r_lw, n)[:, 0]) - np.float64(getattr(lw, n)[:, g])).max()))
        for n in SWN:
            worst_sw = max(worst_sw, float(np.abs(np.float64(getattr(r_sw, n)[:, 0]) - np.float64(getattr(sw, n)[:, g])).max()))
        n_day += int(s1.cos_zenith[0] > 0)
        assert as_.cloud_state.cld_cover_lw[g] == a1.cloud_state.cld_cover_lw[0]
        assert as_.cloud_state.cld_cover_sw[g] == a1.cloud_state.cld_cover_sw[0]
        np.testing.assert_allclose(as_.aerosol_state.aod_sw_ext[g], a1.aerosol_state.aod_sw_ext[0], rtol=2e-5)
        np.testing.assert_allclose(as_.aerosol_state.aod_sw_sca[g], a1.aerosol_state.aod_sw_sca[0], rtol=2e-5)
    assert 0 < n_day < 32 or n_day == 32
    assert worst_lw < 1e-3 and worst_sw < 2e-2, (worst_lw, worst_sw)


@pytest.mark.parametrize("mu0", [0.5, 0.0, 1e-10, -0.5])
def test_cos_zenith_edge_set_on_the_gpu(tables64, mu0):
    """test/cos_zenith_edge_cases.jl:199-226 on the device: mu0 <= 0 gives exactly 0 everywhere, mu0 = 1e-10 stays
    finite and non-negative, mu0 = 0.5 matches the oracle; both SW solvers."""
    t = tables64
    as_, _, sb = S.make_columns(6, 40, np.float64, seed=12, cos_zenith=mu0)
    for twostream in (True, False):
        cls = rte.TwoStreamSWRTE if twostream else rte.NoScatSWRTE
        cld = t["cld_sw"] if twostream else None
        out = rte.solve_sw(cls(6, 40, np.float64, sb), as_, t["sw"], cld)
        ref = O.solve_sw(as_, sb, t["sw"], cld, twostream=twostream)
        for n in SWN:
            a = getattr(out, n)
            assert np.all(np.isfinite(a)) and np.all(a >= 0 if n != "flux_net" else True)
            if mu0 <= 0:
                assert np.all(a == 0.0)
        assert _maxdiff(out, ref, SWN) < 1e-8


def test_config3_allsky_aerosols_4096x73_sharded(tables32):
    """Eight contiguous shards (what 8 ranks would own) reproduce the single-launch result bit for bit."""
    t = tables32
    ncol, nlay = 4096, 73
    as_, lb, sb = S.make_columns(ncol, nlay, np.float32, seed=3, aerosols=True, night_fraction=0.1)
    whole_lw = rte.solve_lw(rte.TwoStreamLWRTE(ncol, nlay, np.float32, lb), as_, t["lw"], t["cld_lw"], t["aero_lw"], seed=5)
    whole_sw = rte.solve_sw(rte.TwoStreamSWRTE(ncol, nlay, np.float32, sb), as_, t["sw"], t["cld_sw"], t["aero_sw"], seed=5)
    for rank in (0, 3, 7):
        lo, hi = sharding.shard_range(ncol, rank, 8)
        a, b_lw, b_sw = (sharding.shard_container(x, lo, hi, ncol) for x in (as_, lb, sb))
        f = rte.solve_lw(rte.TwoStreamLWRTE(hi - lo, nlay, np.float32, b_lw), a, t["lw"], t["cld_lw"], t["aero_lw"], seed=5,
                         col_offset=lo)
        np.testing.assert_array_equal(f.flux_net, whole_lw.flux_net[:, lo:hi])
        f = rte.solve_sw(rte.TwoStreamSWRTE(hi - lo, nlay, np.float32, b_sw), a, t["sw"], t["cld_sw"], t["aero_sw"], seed=5,
                         col_offset=lo)
        np.testing.assert_array_equal(f.flux_dn_dir, whole_sw.flux_dn_dir[:, lo:hi])
    # oracle on a strided sample: every flux of 16 columns (the generator is keyed by the global column, so a one-column
    # state with col_offset = g is column g of the big one)
    kw = dict(seed=3, aerosols=True, night_fraction=0.1)
    worst_lw = worst_sw = 0.0
    for g in range(11, ncol, 256):
        a1, l1, s1 = S.make_columns(1, nlay, np.float32, col_offset=g, **kw)
        r_lw = O.solve_lw(a1, l1, t["lw"], t["cld_lw"], t["aero_lw"], seed=5, col_offset=g)
        r_sw = O.solve_sw(a1, s1, t["sw"], t["cld_sw"], t["aero_sw"], seed=5, col_offset=g)
        for n in LWN:
            worst_lw = max(worst_lw, float(np.abs(np.float64(getattr(r_lw, n)[:, 0]) - np.float64(getattr(whole_lw, n)[:, g])).max()))
        for n in SWN:
            worst_sw = max(worst_sw, float(np.abs(np.float64(getattr(r_sw, n)[:, 0]) - np.float64(getattr(whole_sw, n)[:, g])).max()))
    assert worst_lw < 1e-3 and worst_sw < 2e-2, (worst_lw, worst_sw)
    for f, names in ((whole_lw, LWN), (whole_sw, SWN)):
        for n in names:
            assert np.all(np.isfinite(getattr(f, n)))


def test_config4_gcm_scale_shard_of_1m_columns(tables32):
    """Rank 5's 131 072 columns of the 1 048 576-column problem: size-independent properties plus
    oracle parity on a sample (the oracle at this size would take minutes)."""
    t = tables32
    total, world, rank, nlay = 1_048_576, 8, 5, 64
    lo, hi = sharding.shard_range(total, rank, world)
    assert hi - lo == 131_072
    as_, lb, sb = S.make_columns(hi - lo, nlay, np.float32, seed=2026, col_offset=lo, cos_zenith=0.86)
    lw = rte.solve_lw(rte.TwoStreamLWRTE(hi - lo, nlay, np.float32, lb), as_, t["lw"], t["cld_lw"], seed=11, col_offset=lo)
    sw = rte.solve_sw(rte.TwoStreamSWRTE(hi - lo, nlay, np.float32, sb), as_, t["sw"], t["cld_sw"], seed=11, col_offset=lo)
    for f, names in ((lw, LWN), (sw, SWN)):
        for n in names:
            assert np.all(np.isfinite(getattr(f, n)))
        np.testing.assert_array_equal(f.flux_net, f.flux_up - f.flux_dn)
    np.testing.assert_allclose(sw.flux_dn[-1], sb.toa_flux * sb.cos_zenith, rtol=2e-5)   # TOA incoming = S0 mu0
    assert np.all(lw.flux_dn[-1] == 0) and np.all(lw.flux_up[0] > 0)
    assert np.all(sw.flux_dn_dir <= sw.flux_dn * (1 + 1e-5) + 1e-4)
    # the same global columns generated and solved on their own (another shard width) give the same bits,
    # and the oracle agrees on them
    g0 = lo + 77_777
    sa, slb, ssb = S.make_columns(48, nlay, np.float32, seed=2026, col_offset=g0, cos_zenith=0.86)
    f2 = rte.solve_lw(rte.TwoStreamLWRTE(48, nlay, np.float32, slb), sa, t["lw"], t["cld_lw"], seed=11, col_offset=g0)
    np.testing.assert_array_equal(f2.flux_up, lw.flux_up[:, 77_777:77_777 + 48])
    ref = O.solve_lw(sa, slb, t["lw"], t["cld_lw"], seed=11, col_offset=g0)
    assert _maxdiff(f2, ref, LWN) < 1e-3
    refs = O.solve_sw(sa, ssb, t["sw"], t["cld_sw"], seed=11, col_offset=g0)
    f3 = rte.solve_sw(rte.TwoStreamSWRTE(48, nlay, np.float32, ssb), sa, t["sw"], t["cld_sw"], seed=11, col_offset=g0)
    np.testing.assert_array_equal(f3.flux_dn, sw.flux_dn[:, 77_777:77_777 + 48])
    assert _maxdiff(f3, refs, SWN) < 2e-2


def test_fast_float32_build_passes_the_same_cases():
    """`libhip_rrtmgp_fast.so` (-DRR_FAST_F32: raw v_rcp / v_sqrt / __expf forms in Float32, `make fast`; bench.py times it as
    `variants.fast_f32`) is held to the same budgets as the shipped (IEEE-accurate) library: this file's BASELINE config 1-5
    cases, in a child process that loads it through RRTMGP_HIP_LIBRARY."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    lib = os.path.join(root, "rrtmgp.jl_amd", "libhip_rrtmgp_fast.so")
    assert os.path.exists(lib), "make -C rrtmgp.jl_amd/csrc fast"
    env = dict(os.environ, RRTMGP_HIP_LIBRARY=lib)
    code = ("import sys, pytest; from rrtmgp_jl_amd import _lib; "
            "assert _lib.lib().rrtmgp_hip_build_flags() == b'RR_FAST_F32'; "
            "sys.exit(pytest.main([%r, '-q', '-x', '-m', 'gpu', '-k', 'not fast_float32_build', '-p', 'no:cacheprovider']))"
            % os.path.abspath(__file__))
    r = subprocess.run([sys.executable, "-c", code], cwd=root, env=env, capture_output=True, text=True, timeout=1200)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
