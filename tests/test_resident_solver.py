"""N4 in the form this environment can run: a Layer-2 solver whose EVERY array lives in HBM (`RRTMGPSolver(..., resident=True)`),
what the reference gets from `array_type(::ClimaComms.CUDADevice) = CuArray` (ext/RRTMGPCUDAExt.jl:1-66).

Checked: bits equal to the host-array solver for all four radiation methods (gray, clear sky, all sky, all sky with clear-sky
diagnostics) and both LW solvers; `update_fluxes` stages nothing over PCIe (`rrtmgp_hip_workspace_transfer_bytes == (0, 0)`);
getters return device VIEWS that alias the solver's buffers (also with the isothermal boundary layer, where they are proper
sub-views); the Adapt-style round trip `to_host()` / `to_device()` gives fresh arrays, keeps that aliasing, carries the
current fluxes and RNG state, and the restored solver continues with the same bits (test/standalone.jl:294-335)."""
import copy

import numpy as np
import pytest

from rrtmgp_jl_amd import grid_adaptation as GA
from rrtmgp_jl_amd import solver as L2, synthetic as S
from rrtmgp_jl_amd.states import (GrayOpticalThicknessOGorman2008, LwBCs, RRTMGPParameters, SwBCs, TEST_PARAMETERS, to_host)
from oracle import oracle as O

pytestmark = pytest.mark.gpu

GETTERS = ("lw_flux_up", "lw_flux_dn", "lw_flux_net", "sw_flux_up", "sw_flux_dn", "sw_flux_net", "sw_direct_flux_dn", "net_flux")
CLEAR = ("clear_lw_flux_up", "clear_lw_flux_dn", "clear_lw_flux_net", "clear_sw_flux_up", "clear_sw_flux_dn", "clear_sw_flux_net",
         "clear_sw_direct_flux_dn", "clear_net_flux")


def _spectral(t, FT, method, resident, ncol=9, nlay=22, **kw):
    as_, lb, sb = S.make_columns(ncol, nlay, FT, seed=31, aerosols=True, night_fraction=0.25, random_cld_frac=True)
    as_.vmr.vmr_h2o[1, 0] = -1e-4     # something for clip! to do
    metric = np.asfortranarray(np.random.default_rng(2).uniform(0.97, 1.03, (nlay + 1, ncol)).astype(FT))
    tt = {k: v.astype(FT) for k, v in t.items()}
    lk = L2.LookupBundle(tt["lw"], tt["sw"], tt["cld_lw"], tt["cld_sw"], tt["aero_lw"], tt["aero_sw"])
    return L2.RRTMGPSolver(method, TEST_PARAMETERS, lb, sb, as_, lookups=lk, deep_atmosphere_inverse_scaling=metric,
                           resident=resident, **kw)


def _is_device(x):
    return hasattr(x, "is_cuda") and x.is_cuda


@pytest.mark.parametrize("FT", [np.float64, np.float32])
@pytest.mark.parametrize("method,op_lw", [("clear", "twostream"), ("allsky", "twostream"), ("allsky", "onescalar"),
                                          ("diag", "twostream"), ("diag", "onescalar")])
def test_resident_solver_equals_the_host_array_solver(tables64, FT, method, op_lw):
    m = {"clear": L2.ClearSkyRadiation, "allsky": L2.AllSkyRadiation, "diag": L2.AllSkyRadiationWithClearSkyDiagnostics}[method]
    host = _spectral(tables64, FT, m(aerosol_radiation=True), False, op_lw=op_lw)
    dev = _spectral(tables64, FT, m(aerosol_radiation=True), True, op_lw=op_lw)
    for _ in range(2):                      # two steps: the McICA key advances the same way on both
        L2.update_fluxes(host)
        L2.update_fluxes(dev)
    assert dev.lws.ws.transfer_bytes() == (0, 0)          # nothing crossed PCIe inside update_fluxes
    assert host.lws.ws.transfer_bytes()[0] > 0
    for g in GETTERS + (CLEAR if method == "diag" else ()):
        a, b = getattr(L2, g)(dev), getattr(L2, g)(host)
        assert _is_device(a), g
        np.testing.assert_array_equal(to_host(a), b, err_msg=g)
    # diagnostics and the prepared state live on the device too, with the host solver's values
    if method != "clear":
        assert _is_device(L2.sw_cloud_cover(dev))
        np.testing.assert_array_equal(to_host(L2.sw_cloud_cover(dev)), L2.sw_cloud_cover(host))
    np.testing.assert_array_equal(to_host(L2.aod_sw_extinction(dev)), L2.aod_sw_extinction(host))
    np.testing.assert_array_equal(to_host(dev.as_.layerdata), host.as_.layerdata)       # clip! + col_dry ran in place
    np.testing.assert_array_equal(to_host(L2.heating_rate(dev)), L2.heating_rate(host))


def test_resident_gray_solver(tables64):
    params = RRTMGPParameters()
    ncol, nlay = 7, 40
    lat = np.linspace(-70.0, 70.0, ncol)

    def make(resident):
        gs = O.setup_gray_as_pr_grid(nlay, lat, 100000.0, 9000.0, GrayOpticalThicknessOGorman2008(), params, np.float64)
        lb = LwBCs(np.full((1, ncol), 0.98, order="F"), None)
        mu0 = np.full(ncol, 0.6); mu0[2] = -0.1
        sb = SwBCs(mu0, np.full(ncol, 1407.679), np.full((1, ncol), 0.1, order="F"), np.full((1, ncol), 0.1, order="F"))
        return L2.RRTMGPSolver(L2.GrayRadiation(), params, lb, sb, gs, resident=resident)
    host, dev = make(False), make(True)
    L2.update_fluxes(host)
    L2.update_fluxes(dev)
    for g in GETTERS:
        assert _is_device(getattr(L2, g)(dev))
        np.testing.assert_array_equal(to_host(getattr(L2, g)(dev)), getattr(L2, g)(host), err_msg=g)
    np.testing.assert_array_equal(to_host(L2.heating_rate(dev)), L2.heating_rate(host))


def test_getters_are_views_of_the_resident_buffers_with_the_boundary_layer(tables64):
    s = _spectral(tables64, np.float64, L2.AllSkyRadiation(), True, isothermal_boundary_layer=True)
    L2.update_fluxes(s)
    v = L2.lw_flux_up(s)
    nlev_total = s.nlay + 1
    assert tuple(v.shape) == (s.ncol, nlev_total - 1)                 # Julia (nlev - 1, ncol): without the extra level
    assert v.data_ptr() == s.lws.flux.flux_up.data_ptr()               # a view, not a copy
    v[0, 0] = -7.0
    assert float(s.lws.flux.flux_up[0, 0]) == -7.0
    assert L2.net_flux(s).data_ptr() == s.net_flux_buffer.data_ptr()
    p = L2.layer_pressure(s)
    lo, hi = s.as_.layerdata.data_ptr(), s.as_.layerdata.data_ptr() + s.as_.layerdata.numel() * 8
    assert lo <= p.data_ptr() < hi                                     # the row of `layerdata`, in place


def test_adapt_round_trip(tables64):
    dev = _spectral(tables64, np.float64, L2.AllSkyRadiationWithClearSkyDiagnostics(aerosol_radiation=True), True)
    L2.update_fluxes(dev)
    before = {g: to_host(getattr(L2, g)(dev)).copy() for g in GETTERS + CLEAR}
    host = dev.to_host()                                               # "checkpoint": everything on the host
    assert not host.resident and isinstance(L2.net_flux(host), np.ndarray)
    for g in GETTERS + CLEAR:
        np.testing.assert_array_equal(getattr(L2, g)(host), before[g])
    assert np.shares_memory(L2.lw_flux_up(host), host.lws.flux.flux_up)            # views stay views of ITS buffers
    back = host.to_device()                                            # "restore"
    assert back.resident and back.lws.flux.flux_up.data_ptr() != dev.lws.flux.flux_up.data_ptr()   # fresh arrays
    assert back.as_.layerdata.data_ptr() != dev.as_.layerdata.data_ptr()
    assert L2.lw_flux_up(back).data_ptr() == back.lws.flux.flux_up.data_ptr()                      # aliasing its OWN buffers
    for g in GETTERS + CLEAR:
        np.testing.assert_array_equal(to_host(getattr(L2, g)(back)), before[g])
    # the restored solver continues exactly where the original would
    L2.update_fluxes(dev)
    L2.update_fluxes(back)
    L2.update_fluxes(host)
    for g in GETTERS + CLEAR:
        np.testing.assert_array_equal(to_host(getattr(L2, g)(back)), to_host(getattr(L2, g)(dev)), err_msg=g)
        np.testing.assert_array_equal(getattr(L2, g)(host), to_host(getattr(L2, g)(dev)), err_msg=g)
    assert back.lws.ws.transfer_bytes() == (0, 0)


def test_a_resident_solver_refuses_a_sharded_workspace(tables64):
    with pytest.raises(ValueError, match="ONE device"):
        _spectral(tables64, np.float64, L2.AllSkyRadiation(), True, device=[0, 0])
    _ = copy, GA


def test_device_array_entry_points_round_trip():
    """rrtmgp_hip_device_malloc / _memcpy / _memset / _device_free: what the Julia glue's HIPArray is made of."""
    import ctypes as C
    from rrtmgp_jl_amd import _lib
    L = _lib.lib()
    a = np.arange(1000, dtype=np.float64)
    p = C.c_void_p()
    _lib.check(L.rrtmgp_hip_device_malloc(0, a.nbytes, C.byref(p)), "malloc")
    q = C.c_void_p()
    _lib.check(L.rrtmgp_hip_device_malloc(0, a.nbytes, C.byref(q)), "malloc")
    _lib.check(L.rrtmgp_hip_memcpy(0, p, a.ctypes.data_as(C.c_void_p), a.nbytes, 1), "h2d")
    _lib.check(L.rrtmgp_hip_memcpy(0, q, p, a.nbytes, 3), "d2d")
    b = np.empty_like(a)
    _lib.check(L.rrtmgp_hip_memcpy(0, b.ctypes.data_as(C.c_void_p), q, a.nbytes, 2), "d2h")
    np.testing.assert_array_equal(a, b)
    _lib.check(L.rrtmgp_hip_memset(0, q, 0, a.nbytes), "memset")
    _lib.check(L.rrtmgp_hip_memcpy(0, b.ctypes.data_as(C.c_void_p), q, a.nbytes, 2), "d2h")
    assert (b == 0).all()
    assert L.rrtmgp_hip_memcpy(0, q, p, a.nbytes, 7) != 0 and "kind" in _lib.last_error()
    before = _lib.allocation_counts()
    _lib.check(L.rrtmgp_hip_device_free(0, p), "free")
    _lib.check(L.rrtmgp_hip_device_free(0, q), "free")
    _lib.check(L.rrtmgp_hip_device_free(0, None), "free(NULL)")
    assert _lib.allocation_counts() == before      # caller-owned arrays are outside the library's accounting
