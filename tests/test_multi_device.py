"""Multi-shard workspaces behind the C ABI (`rrtmgp_hip_workspace_create_multi`, include/rrtmgp_hip.h):
columns sharded in contiguous ranges inside the library, one host thread + one stream per shard, lookups
replicated per distinct device.  A 1-GPU box exercises it with `device_ids = [0, 0]` / `[0, 0, 0]`: several
shards on the one GPU.  The bits must be those of the single launch (columns are independent and the McICA
stream is keyed by the global column)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from oracle import oracle as O  # noqa: E402
from rrtmgp_jl_amd import _abi, _lib, rte, synthetic as S  # noqa: E402
from rrtmgp_jl_amd.states import (GrayOpticalThicknessSchneider2004, LwBCs, RRTMGPParameters, SwBCs)  # noqa: E402

LWN = ("flux_up", "flux_dn", "flux_net")
SWN = ("flux_up", "flux_dn", "flux_net", "flux_dn_dir")


def _solve_pair(t, as_, lb, sb, device, seed=11, **kw):
    nlay, ncol = as_.dims
    ws = rte.Workspace(ncol, nlay, as_.dtype, device)
    lw = rte.TwoStreamLWRTE(ncol, nlay, as_.dtype, lb, workspace=ws)
    sw = rte.TwoStreamSWRTE(ncol, nlay, as_.dtype, sb, workspace=ws)
    dl = {k: rte.DeviceLookup(t[k], device) for k in ("lw", "sw", "cld_lw", "cld_sw", "aero_lw", "aero_sw")}
    f_lw = rte.solve_lw(lw, as_, dl["lw"], dl["cld_lw"], dl["aero_lw"], seed=seed, **kw)
    cov_lw = as_.cloud_state.cld_cover_lw.copy()
    f_sw = rte.solve_sw(sw, as_, dl["sw"], dl["cld_sw"], dl["aero_sw"], seed=seed, **kw)
    return ws, f_lw, f_sw, cov_lw, as_.cloud_state.cld_cover_sw.copy()


@pytest.mark.parametrize("ids", [[0], [0, 0], [0, 0, 0]])   # [0] = the one-shard workspace of HIPDevice(0): runs on the caller's thread
@pytest.mark.parametrize("ft", [np.float64, np.float32])
def test_two_stream_shards_are_bit_equal_to_the_single_launch(tables64, tables32, ids, ft):
    t = tables64 if ft == np.float64 else tables32
    ncol = 50 if len(ids) == 3 else 37          # ragged shards: 18 + 19, 16 + 17 + 17
    as_, lb, sb = S.make_columns(ncol, 33, ft, seed=3, aerosols=True, night_fraction=0.25, random_cld_frac=True)
    ws1, a_lw, a_sw, c_lw, c_sw = _solve_pair(t, as_, lb, sb, 0)
    a = {n: a_lw.as_nlev_ncol(n).copy() for n in LWN}, {n: a_sw.as_nlev_ncol(n).copy() for n in SWN}
    wsn, b_lw, b_sw, d_lw, d_sw = _solve_pair(t, as_, lb, sb, ids)
    assert ws1.n_shards == 1 and wsn.n_shards == len(ids)
    for n in LWN:
        np.testing.assert_array_equal(b_lw.as_nlev_ncol(n), a[0][n])
    for n in SWN:
        np.testing.assert_array_equal(b_sw.as_nlev_ncol(n), a[1][n])
    np.testing.assert_array_equal(c_lw, d_lw)
    np.testing.assert_array_equal(c_sw, d_sw)


def test_shards_match_the_oracle_and_respect_col_offset(tables64):
    t = tables64
    as_, lb, sb = S.make_columns(21, 24, np.float64, seed=9, random_cld_frac=True, night_fraction=0.2)
    nlay, ncol = as_.dims
    ws = rte.Workspace(ncol, nlay, np.float64, [0, 0, 0])
    lw = rte.TwoStreamLWRTE(ncol, nlay, np.float64, lb, workspace=ws)
    sw = rte.TwoStreamSWRTE(ncol, nlay, np.float64, sb, workspace=ws)
    out_lw = rte.solve_lw(lw, as_, rte.DeviceLookup(t["lw"], [0, 0, 0]), rte.DeviceLookup(t["cld_lw"], [0, 0, 0]), seed=5, col_offset=1000)
    out_sw = rte.solve_sw(sw, as_, rte.DeviceLookup(t["sw"], [0, 0, 0]), rte.DeviceLookup(t["cld_sw"], [0, 0, 0]), seed=5, col_offset=1000)
    ref_lw = O.solve_lw(as_, lb, t["lw"], t["cld_lw"], seed=5, col_offset=1000)
    ref_sw = O.solve_sw(as_, sb, t["sw"], t["cld_sw"], seed=5, col_offset=1000)
    for n in LWN:
        assert np.abs(out_lw.as_nlev_ncol(n) - ref_lw.as_nlev_ncol(n)).max() < 1e-8
    for n in SWN:
        assert np.abs(out_sw.as_nlev_ncol(n) - ref_sw.as_nlev_ncol(n)).max() < 1e-8


def test_noscat_gray_and_preparation_steps_on_shards(tables64):
    t = tables64
    as_, lb, sb = S.make_columns(13, 20, np.float64, seed=4)
    nlay, ncol = as_.dims
    ids = [0, 0]
    ws = rte.Workspace(ncol, nlay, np.float64, ids)
    # no-scattering LW (3 angles) and SW
    nl = rte.NoScatLWRTE(ncol, nlay, np.float64, lb, n_gauss_angles=3, workspace=ws)
    out = rte.solve_lw(nl, as_, rte.DeviceLookup(t["lw"], ids))
    ref = O.solve_lw(as_, lb, t["lw"], twostream=False, n_gauss_angles=3)
    for n in LWN:
        assert np.abs(out.as_nlev_ncol(n) - ref.as_nlev_ncol(n)).max() < 1e-8
    ns = rte.NoScatSWRTE(ncol, nlay, np.float64, sb, workspace=ws)
    out = rte.solve_sw(ns, as_, rte.DeviceLookup(t["sw"], ids))
    ref = O.solve_sw(as_, sb, t["sw"], twostream=False)
    for n in SWN:
        assert np.abs(out.as_nlev_ncol(n) - ref.as_nlev_ncol(n)).max() < 1e-8
    # col_dry / relative humidity
    params = RRTMGPParameters()
    h2o = as_.vmr.vmr_h2o
    cd = rte.compute_col_gas(ws, as_.p_lev, params, h2o, as_.lat)
    np.testing.assert_allclose(cd, O.compute_col_gas(as_.p_lev, params, h2o, as_.lat), rtol=1e-13)
    p_lay, t_lay = np.asfortranarray(as_.layerdata[1]), np.asfortranarray(as_.layerdata[2])
    rh = rte.compute_relative_humidity(ws, p_lay, t_lay, params, h2o)
    np.testing.assert_allclose(rh, O.compute_relative_humidity(p_lay, t_lay, params, h2o), rtol=1e-12)
    # gray two-stream LW on shards == single
    gs = O.setup_gray_as_pr_grid(nlay, np.linspace(-60.0, 60.0, ncol), 100000.0, 9000.0, GrayOpticalThicknessSchneider2004(),
                                 params, np.float64)
    gb = LwBCs(sfc_emis=np.full((ncol,), 0.98), inc_flux=None)
    one = rte.solve_lw(rte.TwoStreamLWRTE(ncol, nlay, np.float64, gb), gs)
    a = {n: one.as_nlev_ncol(n).copy() for n in LWN}
    many = rte.solve_lw(rte.TwoStreamLWRTE(ncol, nlay, np.float64, gb, workspace=ws), gs)
    for n in LWN:
        np.testing.assert_array_equal(many.as_nlev_ncol(n), a[n])


@pytest.mark.parametrize("twostream", [True, False])
def test_incident_flux_is_sharded_as_a_2d_block(tables64, twostream):
    """LwBCs.inc_flux is (ncol, ngpt): ncol is its FASTEST dimension, so a shard's column range is `ngpt` rows of a
    wider array.  The library hands such blocks around with a leading dimension (`inc_flux_ld`) and packs them while
    staging; same bits as the single launch."""
    t = tables64
    as_, lb, sb = S.make_columns(11, 16, np.float64, seed=2, inc_flux_ngpt=t["lw"].n_gpt)
    nlay, ncol = as_.dims
    cls = rte.TwoStreamLWRTE if twostream else rte.NoScatLWRTE
    one = rte.solve_lw(cls(ncol, nlay, np.float64, lb), as_, t["lw"], t["cld_lw"], seed=3)
    ws = rte.Workspace(ncol, nlay, np.float64, [0, 0, 0])
    dl, dc = rte.DeviceLookup(t["lw"], [0, 0, 0]), rte.DeviceLookup(t["cld_lw"], [0, 0, 0])
    many = rte.solve_lw(cls(ncol, nlay, np.float64, lb, workspace=ws), as_, dl, dc, seed=3)
    for n in LWN:
        np.testing.assert_array_equal(many.as_nlev_ncol(n), one.as_nlev_ncol(n))
    ref = O.solve_lw(as_, lb, t["lw"], t["cld_lw"], twostream=twostream, seed=3)
    assert np.abs(many.as_nlev_ncol("flux_dn") - ref.flux_dn).max() < 1e-8
    assert np.abs(many.as_nlev_ncol("flux_dn")[-1] - lb.inc_flux.sum(axis=1)).max() < 1e-9   # TOA dn == incident flux


def test_incident_flux_block_of_a_wider_array_host_and_device(tables64):
    """`inc_flux_ld` is public: a caller may pass columns [c0, c0 + ncol) of a wider (ncol_total, ngpt) array."""
    import torch
    t = tables64
    as_, lb, sb = S.make_columns(9, 16, np.float64, seed=4, inc_flux_ngpt=t["lw"].n_gpt)
    nlay, ncol = as_.dims
    ref = rte.solve_lw(rte.TwoStreamLWRTE(ncol, nlay, np.float64, lb), as_, t["lw"])
    wide = np.asfortranarray(np.full((ncol + 7, t["lw"].n_gpt), np.nan))
    wide[3:3 + ncol] = lb.inc_flux
    # host block
    blk = LwBCs(sfc_emis=lb.sfc_emis, inc_flux=wide[3:3 + ncol], inc_flux_ld=ncol + 7)
    out = rte.solve_lw(rte.TwoStreamLWRTE(ncol, nlay, np.float64, blk), as_, t["lw"])
    for n in LWN:
        np.testing.assert_array_equal(out.as_nlev_ncol(n), ref.as_nlev_ncol(n))
    # device block: read in place with the stride
    wd = torch.as_tensor(np.ascontiguousarray(wide.T), device="cuda:0")   # torch (ngpt, ncol_total) == Fortran (ncol_total, ngpt)
    dblk = LwBCs(sfc_emis=torch.as_tensor(np.ascontiguousarray(lb.sfc_emis.T), device="cuda:0"),
                 inc_flux=wd[:, 3:3 + ncol], inc_flux_ld=ncol + 7)
    slv = rte.TwoStreamLWRTE(ncol, nlay, np.float64, dblk, flux_device="cuda:0")
    rte.solve_lw(slv, as_.to_device("cuda:0"), t["lw"])
    slv.ws.synchronize()
    got = slv.flux.to_host()
    for n in LWN:
        np.testing.assert_array_equal(got.as_nlev_ncol(n), ref.as_nlev_ncol(n))


def test_band_fluxes_are_sharded_as_blocks_of_columns(tables64):
    """FluxBand is (nlev, ncol, nbnd): a shard's columns are nbnd blocks inside the caller's arrays.  The library hands
    them around with `band_flux_ncol` and writes them home with strided copies: the bits of the single launch."""
    t = tables64
    as_, lb, sb = S.make_columns(13, 24, np.float64, seed=6, random_cld_frac=True, night_fraction=0.2)
    nlay, ncol = as_.dims
    for sw in (False, True):
        r = "sw" if sw else "lw"
        cls, solve, bcs = (rte.TwoStreamSWRTE, rte.solve_sw, sb) if sw else (rte.TwoStreamLWRTE, rte.solve_lw, lb)
        nb = t[r].n_bnd
        one = cls(ncol, nlay, np.float64, bcs, n_bnd_band_flux=nb)
        solve(one, as_, t[r], t["cld_" + r], seed=8)
        ids = [0, 0, 0]
        many = cls(ncol, nlay, np.float64, bcs, n_bnd_band_flux=nb, workspace=rte.Workspace(ncol, nlay, np.float64, ids))
        solve(many, as_, rte.DeviceLookup(t[r], ids), rte.DeviceLookup(t["cld_" + r], ids), seed=8)
        for n in ("flux_up", "flux_dn", "flux_net"):
            np.testing.assert_array_equal(getattr(many.band_flux, n), getattr(one.band_flux, n))
            np.testing.assert_array_equal(many.flux.as_nlev_ncol(n), one.flux.as_nlev_ncol(n))


def test_ncol_fastest_fluxes_are_sharded_as_strided_blocks(tables64):
    """Fluxes in the (ncol, nlev) layout (what FluxLW holds on a device array type): a shard's columns are nlev rows of
    the caller's arrays (`flux_ncol`); same numbers as the single launch, and as the (nlev, ncol) layout transposed."""
    t = tables64
    as_, lb, sb = S.make_columns(11, 16, np.float64, seed=9, random_cld_frac=True, night_fraction=0.2)
    nlay, ncol = as_.dims
    for sw in (False, True):
        r = "sw" if sw else "lw"
        cls, solve, bcs, names = (rte.TwoStreamSWRTE, rte.solve_sw, sb, SWN) if sw else (rte.TwoStreamLWRTE, rte.solve_lw, lb, LWN)
        ref = solve(cls(ncol, nlay, np.float64, bcs), as_, t[r], t["cld_" + r], seed=8)
        ids = [0, 0, 0]
        many = solve(cls(ncol, nlay, np.float64, bcs, workspace=rte.Workspace(ncol, nlay, np.float64, ids),
                         layout=_abi.LAYOUT_NCOL_NLEV),
                     as_, rte.DeviceLookup(t[r], ids), rte.DeviceLookup(t["cld_" + r], ids), seed=8)
        for n in names:
            assert getattr(many, n).shape == (ncol, nlay + 1)
            np.testing.assert_array_equal(many.as_nlev_ncol(n), ref.as_nlev_ncol(n))


def test_what_cannot_be_sharded_is_rejected_loudly(tables64):
    t = tables64
    as_, lb, sb = S.make_columns(8, 16, np.float64, seed=2, inc_flux_ngpt=t["lw"].n_gpt)
    nlay, ncol = as_.dims
    ws = rte.Workspace(ncol, nlay, np.float64, [0, 0])
    lb2 = LwBCs(sfc_emis=lb.sfc_emis, inc_flux=None)
    with pytest.raises(_lib.RRTMGPHipError):   # more shards than columns
        rte.Workspace(1, nlay, np.float64, [0, 0])
    with pytest.raises(_lib.RRTMGPHipError):   # no such device
        rte.Workspace(ncol, nlay, np.float64, [0, 99])
    # a single-device lookup has a replica on device 0, so it serves a [0, 0] workspace too
    out = rte.solve_lw(rte.TwoStreamLWRTE(ncol, nlay, np.float64, lb2, workspace=ws), as_, rte.DeviceLookup(t["lw"], 0))
    assert np.isfinite(out.flux_up).all()


def test_distinct_devices_or_the_documented_error(tables64):
    """device_ids = [0, 1]: on a box with ONE GPU the library must refuse (no such device) before touching anything; on a box
    with two or more it is the real multi-GPU path — lookups replicated per device, one worker thread per shard bound next
    to its GPU — and must give the bits of the single launch.  (The GPU box of this build has one GPU: the second branch
    runs wherever the driver has more.)"""
    t = tables64
    as_, lb, sb = S.make_columns(23, 20, np.float64, seed=12, aerosols=True, night_fraction=0.25, random_cld_frac=True)
    n = _lib.lib().rrtmgp_hip_device_count()
    if n < 2:
        with pytest.raises(_lib.RRTMGPHipError) as e:
            rte.Workspace(23, 20, np.float64, [0, 1])
        assert "device" in str(e.value).lower()
        with pytest.raises(_lib.RRTMGPHipError):
            rte.DeviceLookup(t["lw"], [0, 1])
        return
    ws1, a_lw, a_sw, c_lw, c_sw = _solve_pair(t, as_, lb, sb, 0)
    a = {n_: a_lw.as_nlev_ncol(n_).copy() for n_ in LWN}, {n_: a_sw.as_nlev_ncol(n_).copy() for n_ in SWN}
    ids = list(range(min(n, 8)))
    wsn, b_lw, b_sw, d_lw, d_sw = _solve_pair(t, as_, lb, sb, ids)
    assert wsn.n_shards == len(ids)
    for n_ in LWN:
        np.testing.assert_array_equal(b_lw.as_nlev_ncol(n_), a[0][n_])
    for n_ in SWN:
        np.testing.assert_array_equal(b_sw.as_nlev_ncol(n_), a[1][n_])
    np.testing.assert_array_equal(c_lw, d_lw)
    np.testing.assert_array_equal(c_sw, d_sw)

