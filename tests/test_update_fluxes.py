"""The whole radiation step behind ONE entry of the C ABI (`rrtmgp_hip_update_fluxes`): update_fluxes!(s) of
src/api/update_fluxes.jl:223-233 = prepare_atmosphere! (:252-281) -> update_lw_fluxes! (:12-65) -> update_sw_fluxes!
(:74-128) -> update_net_fluxes! (:165-194), with the state staged once.

Checked here: against the oracle (preparation cascade + the four solves of the clear-sky-diagnostic method + the net
sums), bit for bit against the separate calls of the same library (every radiation method, both LW solvers, per-band
fluxes, interpolation + isothermal boundary layer), sharded and pipelined, on device-resident arrays in both flux
layouts, and by the bytes it moves over PCIe."""
import copy
import ctypes as C

import numpy as np
import pytest

from rrtmgp_jl_amd import _abi, _lib, rte, synthetic as S
from rrtmgp_jl_amd import grid_adaptation as GA
from rrtmgp_jl_amd import solver as L2
from rrtmgp_jl_amd.states import TEST_PARAMETERS, Flux
from oracle import oracle as O

GETTERS = ("lw_flux_up", "lw_flux_dn", "lw_flux_net", "sw_flux_up", "sw_flux_dn", "sw_flux_net", "sw_direct_flux_dn", "net_flux")
CLEAR_GETTERS = ("clear_lw_flux_up", "clear_lw_flux_dn", "clear_lw_flux_net", "clear_sw_flux_up", "clear_sw_flux_dn",
                 "clear_sw_flux_net", "clear_sw_direct_flux_dn", "clear_net_flux")


def test_null_arguments_are_refused_without_a_gpu():
    """The argument checks come before any device work: callable on a box without a GPU."""
    L = _lib.lib()
    assert L.rrtmgp_hip_update_fluxes(None, None) == -1
    a = _abi.UpdateFluxesArgs()
    assert L.rrtmgp_hip_update_fluxes(None, C.byref(a)) == -1
    h, d = C.c_uint64(), C.c_uint64()
    assert L.rrtmgp_hip_workspace_transfer_bytes(None, C.byref(h), C.byref(d)) == -1
    assert C.sizeof(_abi.UpdateFluxesArgs) == L.rrtmgp_hip_abi_sizeof(13) == 17 * 8


def _lookups(t):
    return L2.LookupBundle(t["lw"], t["sw"], t["cld_lw"], t["cld_sw"], t["aero_lw"], t["aero_sw"])


def _method(name, aerosols):
    return {"clear": L2.ClearSkyRadiation, "allsky": L2.AllSkyRadiation,
            "diag": L2.AllSkyRadiationWithClearSkyDiagnostics}[name](aerosol_radiation=aerosols)


def _columns(FT, ncol, nlay, seed=23, **kw):
    as_, lb, sb = S.make_columns(ncol, nlay, FT, seed=seed, aerosols=True, night_fraction=0.2, random_cld_frac=True, **kw)
    # something for clip! to do (grid_adaptation.jl:232-258)
    as_.vmr.vmr_h2o[1, 0] = -1e-4
    as_.layerdata[2][2, ncol - 1] = 400.0
    return as_, lb, sb


def _state_arrays(as_):
    out = {"layerdata": as_.layerdata, "p_lev": as_.p_lev, "t_lev": as_.t_lev, "vmr_h2o": as_.vmr.vmr_h2o, "vmr_o3": as_.vmr.vmr_o3}
    for n in ("cld_r_eff_liq", "cld_r_eff_ice", "cld_path_liq", "cld_path_ice", "cld_frac"):
        out[n] = getattr(as_.cloud_state, n)
    if as_.aerosol_state is not None:
        out["aero_size"], out["aero_mass"] = as_.aerosol_state.aero_size, as_.aerosol_state.aero_mass
    return out


def _pair(t, FT, method, aerosols, ncol=11, nlay=24, device=0, iso=False, interpolation=GA.NoInterpolation, **kw):
    """Two solvers on equal copies of one state: the fused step and the reference's four calls."""
    as_, lb, sb = _columns(FT, ncol, nlay)
    metric = np.asfortranarray(np.random.default_rng(5).uniform(0.97, 1.03, (nlay + 1, ncol)).astype(FT))
    if interpolation != GA.NoInterpolation:   # the levels are outputs then: poison them
        as_.p_lev[1:] = np.nan
        as_.t_lev[1:] = np.nan
    tt = {k: v.astype(FT) for k, v in t.items()}
    out = []
    for fused in (True, False):
        s = L2.RRTMGPSolver(_method(method, aerosols), TEST_PARAMETERS, copy.deepcopy(lb), copy.deepcopy(sb), copy.deepcopy(as_),
                            lookups=_lookups(tt), deep_atmosphere_inverse_scaling=metric, device=device, fused=fused,
                            isothermal_boundary_layer=iso, interpolation=interpolation, **kw)
        out.append(s)
    return out


def _assert_same(a, b, method, exact=True):
    names = GETTERS + (CLEAR_GETTERS if method == "diag" else ())
    for n in names:
        x, y = getattr(L2, n)(a), getattr(L2, n)(b)
        assert np.isfinite(x).all(), n
        if exact:
            np.testing.assert_array_equal(x, y, err_msg=n)
        else:
            np.testing.assert_allclose(x, y, rtol=0, atol=1e-11 * max(1.0, np.abs(y).max()), err_msg=n)
    for n, x in _state_arrays(a.as_).items():
        np.testing.assert_array_equal(x, _state_arrays(b.as_)[n], err_msg=n)
    if method != "clear":
        np.testing.assert_array_equal(L2.lw_cloud_cover(a), L2.lw_cloud_cover(b))
        np.testing.assert_array_equal(L2.sw_cloud_cover(a), L2.sw_cloud_cover(b))


@pytest.mark.gpu
@pytest.mark.parametrize("FT", [np.float64, np.float32])
@pytest.mark.parametrize("method,aerosols", [("clear", False), ("clear", True), ("allsky", False), ("allsky", True),
                                             ("diag", False), ("diag", True)])
def test_fused_step_equals_the_four_calls(tables64, FT, method, aerosols):
    fused, split = _pair(tables64, FT, method, aerosols)
    L2.update_fluxes(fused, 9)
    L2.update_fluxes(split, 9)
    # the one-pass clear-sky diagnostic of the split path is the same kernel; everything else is the same launches
    _assert_same(fused, split, method)
    if aerosols:
        np.testing.assert_array_equal(L2.aod_sw_extinction(fused), L2.aod_sw_extinction(split))
    np.testing.assert_array_equal(L2.net_flux(fused), L2.lw_flux_net(fused) + L2.sw_flux_net(fused))


@pytest.mark.gpu
@pytest.mark.parametrize("ncol,method", [(4500, "allsky"), (4200, "diag"), (600, "diag")])
def test_short_step_on_two_lanes_equals_the_four_calls(tables64, ncol, method):
    """A step of a few columns per resident workgroup (4 096-12 288 on 256 CUs; so is every step of up to 512 columns, which
    the other tests cover) runs its SW kernels on the workspace's second lane, forked after the staged uploads and joined
    before the net sums: same bits as the four calls on one stream.  600 columns: the one-lane range in between."""
    fused, split = _pair(tables64, np.float32, method, True, ncol=ncol, nlay=10)
    for seed in (3, 4):   # the second call reuses both lanes' scratch
        L2.update_fluxes(fused, seed)
        L2.update_fluxes(split, seed)
    _assert_same(fused, split, method)
    np.testing.assert_array_equal(L2.net_flux(fused), L2.lw_flux_net(fused) + L2.sw_flux_net(fused))


@pytest.mark.gpu
@pytest.mark.parametrize("interpolation,bottom", [(GA.ArithmeticMean, GA.SameAsInterpolation), (GA.UniformZ, GA.UseSurfaceTempAtBottom),
                                                  (GA.BestFit, GA.HydrostaticBottom)])
@pytest.mark.parametrize("iso", [False, True])
def test_fused_step_with_preparation_schemes(tables64, interpolation, bottom, iso):
    """Level interpolation + isothermal boundary layer + clipping + col_dry run as a kernel in front of the solves; the
    prepared state comes back to the caller exactly as prepare_atmosphere! leaves it."""
    ncol, nlay = 7, 20
    rng = np.random.default_rng(3)
    cz = np.asfortranarray(np.cumsum(rng.uniform(300, 900, (nlay, ncol)), axis=0))
    fz = np.asfortranarray(np.vstack([cz[:1] - 200.0, 0.5 * (cz[1:] + cz[:-1]), cz[-1:] + 300.0]))
    kw = dict(center_z=cz, face_z=fz, bottom_extrapolation=bottom)
    fused, split = _pair(tables64, np.float64, "diag", True, ncol=ncol, nlay=nlay, iso=iso, interpolation=interpolation, **kw)
    L2.update_fluxes(fused, 4)
    L2.update_fluxes(split, 4)
    _assert_same(fused, split, "diag")
    assert np.isfinite(fused.as_.p_lev).all() and np.isfinite(fused.as_.t_lev).all()
    assert L2.net_flux(fused).shape == (nlay + 1 - int(iso), ncol)


@pytest.mark.gpu
@pytest.mark.parametrize("method,aerosols", [("clear", False), ("allsky", False), ("clear", True)])
def test_fused_step_fills_the_boundary_layer_of_arrays_the_method_does_not_read(tables64, method, aerosols):
    """with the isothermal boundary layer the preparation fills the extra layer of EVERY cloud / aerosol array the
    state carries (grid_adaptation.jl), whatever the radiation method reads.  ClearSkyRadiation on a state with a CloudState,
    AllSkyRadiation(aerosol_radiation = false) on one with an AerosolState: the state after the fused step must equal the
    state after the reference's four separate calls, array by array, extra layer included."""
    fused, split = _pair(tables64, np.float64, method, aerosols, iso=True, interpolation=GA.ArithmeticMean)
    for s in (fused, split):    # poison the extra layer of everything the preparation has to define
        for arr in _state_arrays(s.as_).values():
            if arr.ndim == 2 and arr.shape[0] == s.nlay:
                arr[-1, :] = np.nan
            elif arr.ndim == 3 and arr.shape[1] == s.nlay:
                arr[:, -1, :] = np.nan
    L2.update_fluxes(fused, 3)
    L2.update_fluxes(split, 3)
    a, b = _state_arrays(fused.as_), _state_arrays(split.as_)
    for name in a:
        np.testing.assert_array_equal(a[name], b[name], err_msg=name)
        assert np.isfinite(a[name]).all(), name
    _assert_same(fused, split, method)


@pytest.mark.gpu
def test_clear_sky_step_ignores_cloud_inputs_no_lookup_reads(tables64):
    """ClearSkyRadiation with the isothermal boundary layer stages the cloud arrays for the preparation only; an
    `ice_rgh` that no cloud lookup will ever read (0 here) is nobody's business, and the step must not refuse the state for
    it (round 5 did: "ice_rgh must be in 1..nrghice of the cloud lookup")."""
    fused, split = _pair(tables64, np.float64, "clear", False, iso=True, interpolation=GA.ArithmeticMean)
    for s in (fused, split):
        s.as_.cloud_state.ice_rgh = 0
    L2.update_fluxes(fused, 3)
    L2.update_fluxes(split, 3)
    _assert_same(fused, split, "clear")
    # ... while a method that does read the clouds still refuses it
    bad, _ = _pair(tables64, np.float64, "allsky", False)
    bad.as_.cloud_state.ice_rgh = 0
    with pytest.raises(Exception, match="ice_rgh"):
        L2.update_fluxes(bad, 3)


@pytest.mark.gpu
def test_fused_step_matches_the_oracle(tables64):
    """prepare (interpolation, clip, col_dry) + LW + SW + clear-sky pair + net sums against the CPU restatement."""
    t = tables64
    ncol, nlay = 9, 28
    as_, lb, sb = _columns(np.float64, ncol, nlay)
    as_.p_lev[1:] = np.nan
    as_.t_lev[1:] = np.nan
    ref_as = copy.deepcopy(as_)
    s = L2.RRTMGPSolver(L2.AllSkyRadiationWithClearSkyDiagnostics(aerosol_radiation=True, reset_rng_seed=True), TEST_PARAMETERS,
                        lb, sb, as_, lookups=_lookups(t), interpolation=GA.GeometricMean)
    assert s.fused
    L2.update_fluxes(s, 77)
    lw = t["lw"]
    O.prepare_atmosphere(ref_as, TEST_PARAMETERS, _abi.PREP_ALL, interpolation=GA.GeometricMean, p_min=lw.p_ref_min,
                         t_min=lw.t_ref_min, t_max=lw.t_ref_max, idx_h2o=lw.idx_h2o)
    for n, x in _state_arrays(as_).items():
        np.testing.assert_allclose(x, _state_arrays(ref_as)[n], rtol=1e-13, atol=0, err_msg=n)
    key = s._seed
    r_lw = O.solve_lw(ref_as, lb, t["lw"], t["cld_lw"], t["aero_lw"], seed=key)
    r_sw = O.solve_sw(ref_as, sb, t["sw"], t["cld_sw"], t["aero_sw"], seed=key)
    c_lw = O.solve_lw(ref_as, lb, t["lw"], None, t["aero_lw"], seed=key)
    c_sw = O.solve_sw(ref_as, sb, t["sw"], None, t["aero_sw"], seed=key)
    tol = 1e-8
    for g, ref in (("lw_flux_up", r_lw.flux_up), ("lw_flux_dn", r_lw.flux_dn), ("lw_flux_net", r_lw.flux_net),
                   ("sw_flux_up", r_sw.flux_up), ("sw_flux_dn", r_sw.flux_dn), ("sw_flux_net", r_sw.flux_net),
                   ("sw_direct_flux_dn", r_sw.flux_dn_dir), ("clear_lw_flux_up", c_lw.flux_up), ("clear_lw_flux_dn", c_lw.flux_dn),
                   ("clear_sw_flux_up", c_sw.flux_up), ("clear_sw_flux_dn", c_sw.flux_dn),
                   ("clear_sw_direct_flux_dn", c_sw.flux_dn_dir), ("net_flux", r_lw.flux_net + r_sw.flux_net),
                   ("clear_net_flux", c_lw.flux_net + c_sw.flux_net)):
        assert np.abs(getattr(L2, g)(s) - ref).max() < tol, g


@pytest.mark.gpu
@pytest.mark.parametrize("what", ["noscat", "noscat3", "bands"])
def test_fused_step_where_the_one_pass_diagnostic_does_not_apply(tables64, what):
    """A no-scattering LW solver, or per-band fluxes, next to the clear-sky diagnostic: the cloudless solve runs first on
    the staged state, as the reference does (update_fluxes.jl:39-65)."""
    kw = {"noscat": dict(op_lw="onescalar"), "noscat3": dict(op_lw="onescalar", n_gauss_angles=3),
          "bands": dict(spectral_fluxes=True)}[what]
    fused, split = _pair(tables64, np.float64, "diag", True, **kw)
    L2.update_fluxes(fused, 3)
    L2.update_fluxes(split, 3)
    # per-band fluxes: the split path's cloudless solve is the per-band kernel instance too (g-point sums per 16-lane row,
    # rows added afterwards), the fused step's the plain one (sums per wavefront): same fluxes to rounding, not to the bit
    _assert_same(fused, split, "diag", exact=what != "bands")
    if what == "bands":
        for n in ("spectral_lw_flux_up", "spectral_lw_flux_dn", "spectral_sw_flux_up", "spectral_sw_flux_dn"):
            np.testing.assert_array_equal(getattr(L2, n)(fused), getattr(L2, n)(split), err_msg=n)
        np.testing.assert_array_equal(L2.spectral_lw_flux_net(fused), L2.spectral_lw_flux_up(fused) - L2.spectral_lw_flux_dn(fused))
        np.testing.assert_allclose(L2.spectral_sw_flux_up(fused).sum(axis=2), L2.sw_flux_up(fused), rtol=1e-12, atol=1e-9)


@pytest.mark.gpu
@pytest.mark.parametrize("device", [[0], [0, 0], [0, 0, 0]])   # [0]: what HIPDevice(0) creates (one shard, no worker pool)
def test_fused_step_on_a_sharded_workspace(tables64, device):
    one, _ = _pair(tables64, np.float64, "diag", True, ncol=13)
    many, _ = _pair(tables64, np.float64, "diag", True, ncol=13, device=device, interpolation=GA.NoInterpolation)
    assert many.lws.ws.n_shards == len(device)
    L2.update_fluxes(one, 5)
    L2.update_fluxes(many, 5)
    _assert_same(many, one, "diag")


@pytest.mark.gpu
def test_fused_step_through_the_column_pipeline(tables32):
    """From 16 384 columns on a host-array step runs as a pipeline of column chunks (uploads of chunk c + 1 and downloads
    of chunk c - 1 overlap the kernels of chunk c): same bits as the four separate (also pipelined) calls, and the state
    crosses PCIe once instead of once per solve."""
    ncol, nlay = 16384 + 37, 12
    as_, lb, sb = S.make_columns(ncol, nlay, np.float32, seed=8, night_fraction=0.1, random_cld_frac=True)
    t = tables32
    lookups = L2.LookupBundle(t["lw"], t["sw"], t["cld_lw"], t["cld_sw"])
    ss = []
    for fused in (True, False):
        s = L2.RRTMGPSolver(L2.AllSkyRadiationWithClearSkyDiagnostics(), TEST_PARAMETERS, copy.deepcopy(lb), copy.deepcopy(sb),
                            copy.deepcopy(as_), lookups=lookups, interpolation=GA.ArithmeticMean, fused=fused)
        L2.update_fluxes(s, 1)   # warm (staging buffers, lookups)
        b0 = s.lws.ws.transfer_bytes()
        L2.update_fluxes(s, 2)
        b1 = s.lws.ws.transfer_bytes()
        ss.append((s, b1[0] - b0[0], b1[1] - b0[1]))
    (f, f_up, f_dn), (u, u_up, u_dn) = ss
    _assert_same(f, u, "diag")
    # uploads: every input array exactly once
    a = f.as_
    state = sum(x.nbytes for x in (a.layerdata, a.p_lev, a.t_lev, a.t_sfc, a.lat, a.vmr.vmr_h2o, a.vmr.vmr_o3)) + \
        sum(getattr(a.cloud_state, n).nbytes for n in ("cld_r_eff_liq", "cld_r_eff_ice", "cld_path_liq", "cld_path_ice", "cld_frac"))
    bcs = f.lws.bcs.sfc_emis.nbytes + sum(getattr(f.sws.bcs, n).nbytes for n in ("cos_zenith", "toa_flux", "sfc_alb_direct", "sfc_alb_diffuse"))
    assert state + bcs <= f_up <= state + bcs + 64 * a.vmr.vmr.nbytes   # (+ the well-mixed vector, once per staging set)
    assert u_up > 1.9 * f_up, (u_up, f_up)   # prepare, LW and SW each stage (their part of) the state
    # downloads: the same fluxes and prepared state, plus the two net sums that the split path forms on the host
    assert f_dn <= u_dn + 2 * f.net_flux_buffer.nbytes


@pytest.mark.gpu
@pytest.mark.parametrize("layout", [_abi.LAYOUT_NLEV_NCOL, _abi.LAYOUT_NCOL_NLEV])
def test_fused_step_on_device_resident_arrays(tables64, layout):
    """State, boundary conditions and fluxes in HBM (torch tensors): used in place; net_flux is (nlev, ncol) whatever the
    layout of the flux arrays (transpose_sum_into!, Fluxes.jl:407-424)."""
    import torch
    from rrtmgp_jl_amd import rte
    from rrtmgp_jl_amd.states import to_host
    t = tables64
    ncol, nlay = 10, 18
    as_h, lb_h, sb_h = S.make_columns(ncol, nlay, np.float64, seed=12, night_fraction=0.2, random_cld_frac=True)
    dev = torch.device("cuda", 0)
    as_d, lb_d, sb_d = as_h.to_device(dev), lb_h.to_device(dev), sb_h.to_device(dev)
    ws = rte.Workspace(ncol, nlay, np.float64, 0)
    lws = rte.TwoStreamLWRTE(ncol, nlay, np.float64, lb_d, flux_device=dev, layout=layout, workspace=ws)
    sws = rte.TwoStreamSWRTE(ncol, nlay, np.float64, sb_d, flux_device=dev, layout=layout, workspace=ws)
    net = torch.full((ncol, nlay + 1), float("nan"), dtype=torch.float64, device=dev)   # Julia (nlev, ncol)
    prep = GA.prepare_atmosphere_opts(as_d, t["lw"])
    rte.update_fluxes(lws, sws, as_d, t["lw"], t["sw"], t["cld_lw"], t["cld_sw"], seed=6, net_flux=net, params=TEST_PARAMETERS,
                      prepare=prep)
    ws.synchronize()
    b0 = ws.transfer_bytes()
    assert b0 == (0, 0)   # nothing was staged
    # reference: the host path of the same library
    ws_h = rte.Workspace(ncol, nlay, np.float64, 0)
    lh = rte.TwoStreamLWRTE(ncol, nlay, np.float64, lb_h, workspace=ws_h)
    sh = rte.TwoStreamSWRTE(ncol, nlay, np.float64, sb_h, workspace=ws_h)
    net_h = np.zeros((nlay + 1, ncol), order="F")
    rte.update_fluxes(lh, sh, as_h, t["lw"], t["sw"], t["cld_lw"], t["cld_sw"], seed=6, net_flux=net_h, params=TEST_PARAMETERS,
                      prepare=GA.prepare_atmosphere_opts(as_h, t["lw"]))
    np.testing.assert_array_equal(to_host(net), net_h)
    for n in ("flux_up", "flux_dn", "flux_net"):
        np.testing.assert_array_equal(lws.flux.as_nlev_ncol(n), getattr(lh.flux, n))
        np.testing.assert_array_equal(sws.flux.as_nlev_ncol(n), getattr(sh.flux, n))
    np.testing.assert_array_equal(to_host(as_d.layerdata), as_h.layerdata)   # col_dry written in place on the device too


@pytest.mark.gpu
def test_fused_step_argument_errors(tables64):
    from rrtmgp_jl_amd import rte
    t = tables64
    as_, lb, sb = S.make_columns(4, 10, np.float64, seed=1)
    ws = rte.Workspace(4, 10, np.float64, 0)
    lws, sws = rte.TwoStreamLWRTE(4, 10, np.float64, lb, workspace=ws), rte.TwoStreamSWRTE(4, 10, np.float64, sb, workspace=ws)
    with pytest.raises(_lib.RRTMGPHipError, match="mismatch"):
        rte.update_fluxes(lws, sws, as_, t["sw"], t["sw"])
    with pytest.raises(_lib.RRTMGPHipError, match="cloud lookups"):
        rte.update_fluxes(lws, sws, as_, t["lw"], t["sw"], clear_flux_lw=Flux.allocate(4, 11, np.float64),
                          clear_flux_sw=Flux.allocate(4, 11, np.float64, sw=True))
    with pytest.raises(ValueError, match="share one Workspace"):
        rte.update_fluxes(lws, rte.TwoStreamSWRTE(4, 10, np.float64, sb), as_, t["lw"], t["sw"])
    with pytest.raises(ValueError, match="params"):
        rte.update_fluxes(lws, sws, as_, t["lw"], t["sw"], prepare=GA.prepare_atmosphere_opts(as_, t["lw"]))


@pytest.mark.gpu
def test_lane_orders_of_a_short_step_give_the_same_bits():
    """the two-lane form of a short fused step (LW on the main lane, SW on the second) against the one-lane form, and
    the two alternative queueing orders kept behind RRTMGP_HIP_STEP_ORDER — with the clear-sky diagnostic, aerosols, the
    preparation cascade and the isothermal layer in the step.  The switches are read once per process: one child each, a
    digest of every output and of the prepared state."""
    import hashlib
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = r'''
import hashlib, sys, numpy as np
sys.path.insert(0, %r); sys.path.insert(0, %r)
import rrtmgp_jl_amd
from rrtmgp_jl_amd import grid_adaptation as GA, solver as L2, synthetic as S
from rrtmgp_jl_amd.states import TEST_PARAMETERS
import test_update_fluxes as T
t = {k: None for k in ()}
lw, sw = S.make_gas_lookup("lw", np.float64), S.make_gas_lookup("sw", np.float64)
tabs = dict(lw=lw, sw=sw, cld_lw=S.make_cloud_lookup("lw", lw.n_bnd), cld_sw=S.make_cloud_lookup("sw", sw.n_bnd),
            aero_lw=S.make_aerosol_lookup("lw", lw.bnd_lims_wn), aero_sw=S.make_aerosol_lookup("sw", sw.bnd_lims_wn))
h = hashlib.sha256()
for method, ncol in (("diag", 300), ("allsky", 37)):
    fused, _ = T._pair(tabs, np.float64, method, True, ncol=ncol, nlay=20, iso=True, interpolation=GA.UniformZ)
    for seed in (3, 4):
        L2.update_fluxes(fused, seed)
    for g in T.GETTERS + (T.CLEAR_GETTERS if method == "diag" else ()):
        h.update(np.ascontiguousarray(getattr(L2, g)(fused)).tobytes())
    for a in T._state_arrays(fused.as_).values():
        h.update(np.ascontiguousarray(a).tobytes())
    h.update(np.ascontiguousarray(L2.sw_cloud_cover(fused)).tobytes())
print("DIGEST", h.hexdigest())
''' % (root, os.path.join(root, "tests"))
    digests = {}
    for name, env in (("one lane", {"RRTMGP_HIP_STEP_OVERLAP": "0"}), ("two lanes", {"RRTMGP_HIP_STEP_OVERLAP": "1"}),
                      ("order 1", {"RRTMGP_HIP_STEP_OVERLAP": "1", "RRTMGP_HIP_STEP_ORDER": "1"}),
                      ("order 2", {"RRTMGP_HIP_STEP_OVERLAP": "1", "RRTMGP_HIP_STEP_ORDER": "2"})):
        r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=dict(os.environ, **env), timeout=600)
        assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-1500:]
        digests[name] = [ln.split()[1] for ln in r.stdout.splitlines() if ln.startswith("DIGEST")][0]
    assert len(set(digests.values())) == 1, digests
    _ = hashlib


# ---- the gray step (rrtmgp_hip_update_fluxes_gray) --------------------------------------------------------------------
def _gray_solver(fused, op_lw="twostream", op_sw="twostream", iso=False, interpolation=GA.NoInterpolation, resident=False,
                 device=0, FT=np.float64, ncol=9, nlay=30, metric=True):
    from oracle import oracle as O
    from rrtmgp_jl_amd.states import GrayOpticalThicknessOGorman2008, LwBCs, RRTMGPParameters, SwBCs
    params = RRTMGPParameters()
    gs = O.setup_gray_as_pr_grid(nlay, np.linspace(-70.0, 70.0, ncol), 100000.0, 9000.0, GrayOpticalThicknessOGorman2008(), params, FT)
    if interpolation != GA.NoInterpolation:   # the levels are outputs then
        gs.p_lev[1:] = np.nan
        gs.t_lev[1:] = np.nan
    lb = LwBCs(np.asfortranarray(np.full((1, ncol), 0.97, FT)), np.asfortranarray(np.linspace(0.0, 3.0, ncol).astype(FT)))
    mu0 = np.full(ncol, 0.6, FT); mu0[2] = -0.1
    sb = SwBCs(mu0, np.full(ncol, 1407.679, FT), np.asfortranarray(np.full((1, ncol), 0.12, FT)),
               np.asfortranarray(np.full((1, ncol), 0.1, FT)))
    ms = np.asfortranarray(np.random.default_rng(4).uniform(0.97, 1.03, (nlay + 1, ncol)).astype(FT)) if metric else None
    return L2.RRTMGPSolver(L2.GrayRadiation(), params, lb, sb, gs, op_lw=op_lw, op_sw=op_sw, fused=fused, resident=resident,
                           device=device, isothermal_boundary_layer=iso, interpolation=interpolation,
                           deep_atmosphere_inverse_scaling=ms)


@pytest.mark.gpu
@pytest.mark.parametrize("FT", [np.float64, np.float32])
@pytest.mark.parametrize("op_lw,op_sw", [("twostream", "twostream"), ("onescalar", "twostream"), ("twostream", "onescalar"),
                                         ("onescalar", "onescalar")])
def test_gray_step_equals_the_separate_calls(FT, op_lw, op_sw):
    """update_fluxes! for GrayRadiation as ONE library call (prepare + gray LW + gray SW + net sum) against the reference's
    four steps as separate calls: the same bits in every getter, with incident flux, a night column and the metric factors."""
    fused, split = _gray_solver(True, op_lw, op_sw, FT=FT), _gray_solver(False, op_lw, op_sw, FT=FT)
    L2.update_fluxes(fused)
    L2.update_fluxes(split)
    for g in GETTERS:
        x, y = getattr(L2, g)(fused), getattr(L2, g)(split)
        assert np.isfinite(x).all() and (np.abs(x).max() > 0 or (g == "sw_flux_up" and op_sw == "onescalar")), g   # (no-scattering SW: nothing goes up)
        np.testing.assert_array_equal(x, y, err_msg=g)
    np.testing.assert_array_equal(L2.net_flux(fused), L2.lw_flux_net(fused) + L2.sw_flux_net(fused))
    assert (L2.sw_flux_dn(fused)[:, 2] == 0).all()


@pytest.mark.gpu
@pytest.mark.parametrize("interpolation,iso", [(GA.ArithmeticMean, True), (GA.GeometricMean, False), (GA.NoInterpolation, True)])
def test_gray_step_with_preparation_and_on_resident_and_sharded_arrays(interpolation, iso):
    """The preparation cascade of a gray state (level interpolation, isothermal boundary layer, pressure clip) runs as a kernel
    in front of the two gray solves; host arrays, device-resident arrays (nothing crosses PCIe) and a three-shard workspace
    give the bits of the separate calls, prepared state included."""
    from rrtmgp_jl_amd.states import to_host
    ref = _gray_solver(False, iso=iso, interpolation=interpolation)
    L2.update_fluxes(ref)
    for kw in (dict(), dict(resident=True), dict(device=[0, 0, 0])):
        s = _gray_solver(True, iso=iso, interpolation=interpolation, **kw)
        L2.update_fluxes(s)
        for g in GETTERS:
            np.testing.assert_array_equal(to_host(getattr(L2, g)(s)), getattr(L2, g)(ref), err_msg=f"{g} {kw}")
        for n in ("p_lay", "p_lev", "t_lay", "t_lev"):
            np.testing.assert_array_equal(to_host(getattr(s.as_, n)), getattr(ref.as_, n), err_msg=f"{n} {kw}")
        if kw.get("resident"):
            assert s.lws.ws.transfer_bytes() == (0, 0)


@pytest.mark.gpu
def test_gray_step_argument_errors():
    import ctypes as C
    from rrtmgp_jl_amd import _abi, _lib
    s = _gray_solver(True)
    a = _abi.UpdateFluxesGrayArgs()
    assert _lib.lib().rrtmgp_hip_update_fluxes_gray(s.lws.ws.handle, C.byref(a)) != 0 and "required" in _lib.last_error()
    assert _lib.lib().rrtmgp_hip_update_fluxes_gray(None, C.byref(a)) != 0
    other = rte.Workspace(3, 30, np.float64)
    with pytest.raises(Exception, match="share one Workspace|dimensions"):
        rte.update_fluxes_gray(s.lws, rte.TwoStreamSWRTE(9, 30, np.float64, s.sws.bcs, workspace=other), s.as_)
