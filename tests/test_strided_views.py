"""compute_col_gas! / compute_relative_humidity! / compute_gray_heating_rate! with the arguments the REFERENCE passes:
strided views, not dense copies.

  * `update_concentrations!` (src/api/grid_adaptation.jl:278-292) hands over `getview_col_dry(as)` =
    `view(as.layerdata, 1, :, :)` (element stride 4, AtmosphericStates.jl:96-97) and, for a full `Vmr`,
    `view(vmr.vmr, idx_h2o, :, :)` (element stride ngas, grid_adaptation.jl:204);
  * the reference's drivers call `compute_relative_humidity!` on rows 4, 2, 3 of layerdata (test/read_clear_sky.jl:162-169);
  * `heating_rate` (src/api/standalone.jl:106-122) passes the getters' domain views `view(x, 1:nlev-1, :)` when the grid
    has an isothermal boundary layer (src/api/getters.jl:42-43).

The C ABI takes these as rrtmgp_view2d (pointer + element strides): nothing is copied on the caller's side and whatever
else lives in the parent arrays must come back bit-identical."""
import numpy as np
import pytest

import rrtmgp_jl_amd  # noqa: F401
from oracle import oracle as O
from rrtmgp_jl_amd import _lib, rte, synthetic as S
from rrtmgp_jl_amd.states import RRTMGPParameters

pytestmark = pytest.mark.gpu


def _state(ft, ncol=37, nlay=26, full_vmr=True, seed=5):
    as_, _, _ = S.make_columns(ncol, nlay, ft, seed=seed, vmr_kind="full" if full_vmr else "gm")
    return as_


@pytest.mark.parametrize("ft,rtol", [(np.float64, 1e-13), (np.float32, 1e-5)])
@pytest.mark.parametrize("device", [0, [0, 0, 0]])
def test_col_gas_writes_row_1_of_layerdata_from_a_row_of_vmr(ft, rtol, device):
    params = RRTMGPParameters()
    as_ = _state(ft)
    nlay, ncol = as_.dims
    ws = rte.Workspace(ncol, nlay, ft, device)
    ld = as_.layerdata                      # (4, nlay, ncol), Fortran order
    vmr = as_.vmr.vmr                       # (ngas, nlay, ncol)
    assert ld.flags.f_contiguous and vmr.flags.f_contiguous and vmr.ndim == 3
    before = ld.copy(order="F")
    vmr_before = vmr.copy(order="F")
    h2o_view, cd_view = vmr[0], ld[0]       # what _vmr_h2o / getview_col_dry return: element strides ngas and 4
    assert cd_view.strides[0] == 4 * ld.itemsize and h2o_view.strides[0] == vmr.shape[0] * vmr.itemsize
    want = O.compute_col_gas(as_.p_lev, params, np.asfortranarray(h2o_view), as_.lat)
    ld[0] = -1.0
    out = rte.compute_col_gas(ws, as_.p_lev, params, h2o_view, as_.lat, out=cd_view)
    assert out is cd_view
    np.testing.assert_allclose(ld[0], want, rtol=rtol)
    np.testing.assert_array_equal(ld[1:], before[1:])       # p_lay, t_lay, rel_hum untouched
    np.testing.assert_array_equal(vmr, vmr_before)


@pytest.mark.parametrize("ft,rtol", [(np.float64, 1e-12), (np.float32, 2e-5)])
@pytest.mark.parametrize("device", [0, [0, 0]])
def test_relative_humidity_on_the_rows_of_layerdata(ft, rtol, device):
    params = RRTMGPParameters()
    as_ = _state(ft, full_vmr=False)
    nlay, ncol = as_.dims
    ws = rte.Workspace(ncol, nlay, ft, device)
    ld = as_.layerdata
    before = ld.copy(order="F")
    h2o = as_.vmr.vmr_h2o
    want = O.compute_relative_humidity(np.asfortranarray(ld[1]), np.asfortranarray(ld[2]), params, h2o)
    ld[3] = -1.0
    rte.compute_relative_humidity(ws, ld[1], ld[2], params, h2o, out=ld[3])   # three views of ONE parent + a dense array
    np.testing.assert_allclose(ld[3], want, rtol=rtol)
    np.testing.assert_array_equal(ld[:3], before[:3])


def test_views_in_device_memory():
    import torch
    params = RRTMGPParameters()
    ft = np.float64
    as_ = _state(ft)
    nlay, ncol = as_.dims
    ws = rte.Workspace(ncol, nlay, ft)
    dev = torch.device("cuda", 0)
    ld = torch.from_numpy(np.ascontiguousarray(as_.layerdata.T)).to(dev)      # torch shape (ncol, nlay, 4): same bytes
    vmr = torch.from_numpy(np.ascontiguousarray(as_.vmr.vmr.T)).to(dev)       # (ncol, nlay, ngas)
    p_lev = torch.from_numpy(np.ascontiguousarray(as_.p_lev.T)).to(dev)
    lat = torch.from_numpy(as_.lat).to(dev)
    want = O.compute_col_gas(as_.p_lev, params, np.asfortranarray(as_.vmr.vmr[0]), as_.lat)
    keep = ld.clone()
    rte.compute_col_gas(ws, p_lev, params, vmr[:, :, 0], lat, out=ld[:, :, 0])
    ws.synchronize()
    np.testing.assert_allclose(ld[:, :, 0].cpu().numpy().T, want, rtol=1e-13)
    assert torch.equal(ld[:, :, 1:], keep[:, :, 1:])
    want_rh = O.compute_relative_humidity(np.asfortranarray(as_.layerdata[1]), np.asfortranarray(as_.layerdata[2]), params,
                                          np.asfortranarray(as_.vmr.vmr[0]))
    rte.compute_relative_humidity(ws, ld[:, :, 1], ld[:, :, 2], params, vmr[:, :, 0], out=ld[:, :, 3])
    ws.synchronize()
    np.testing.assert_allclose(ld[:, :, 3].cpu().numpy().T, want_rh, rtol=1e-12)


@pytest.mark.parametrize("device", [0, [0, 0]])
def test_heating_rate_on_domain_views(device):
    """The solver's arrays have nlev + 1 levels when there is an isothermal boundary layer; `heating_rate` passes the
    first nlev of them as views and the domain's layer count: one less than the workspace's."""
    params = RRTMGPParameters()
    ft = np.float64
    ncol, nlay = 19, 31                       # workspace (solver) layers; the domain has nlay - 1
    as_ = _state(ft, ncol=ncol, nlay=nlay, full_vmr=False)
    ws = rte.Workspace(ncol, nlay, ft, device)
    rng = np.random.default_rng(11)
    fnet = np.asfortranarray(rng.normal(0.0, 40.0, (nlay + 1, ncol)))
    p_dom, f_dom = as_.p_lev[:nlay], fnet[:nlay]      # _domain_view: rows 1:nlev-1, column stride nlev
    assert p_dom.strides[1] == (nlay + 1) * 8 and not p_dom.flags.f_contiguous
    want = O.gray_heating_rate(np.asfortranarray(f_dom), np.asfortranarray(p_dom), params.grav, params.cp_d)
    hr = rte.compute_gray_heating_rate(ws, p_dom, f_dom, params.cp_d, params.grav)
    assert hr.shape == (nlay - 1, ncol)
    np.testing.assert_allclose(hr, want, rtol=1e-13)
    # into a view as well (a host model's (nlay, ncol) buffer, first nlay - 1 rows)
    buf = np.full((nlay, ncol), 7.0, order="F")
    rte.compute_gray_heating_rate(ws, p_dom, f_dom, params.cp_d, params.grav, out=buf[:nlay - 1])
    np.testing.assert_allclose(buf[:nlay - 1], want, rtol=1e-13)
    assert np.all(buf[nlay - 1] == 7.0)


def test_extents_are_checked():
    """Round 1's entry points took their sizes from the workspace: a smaller caller array was an out-of-bounds read."""
    params = RRTMGPParameters()
    as_ = _state(np.float64, ncol=8, nlay=12, full_vmr=False)
    small = rte.Workspace(4, 12, np.float64)
    with pytest.raises(_lib.RRTMGPHipError, match="ncol exceeds"):
        rte.compute_col_gas(small, as_.p_lev, params)
    shallow = rte.Workspace(8, 10, np.float64)
    with pytest.raises(_lib.RRTMGPHipError, match="nlay exceeds"):
        rte.compute_col_gas(shallow, as_.p_lev, params)
    sharded = rte.Workspace(16, 12, np.float64, [0, 0])
    with pytest.raises(_lib.RRTMGPHipError, match="multi-device"):
        rte.compute_col_gas(sharded, as_.p_lev, params)
    # fewer columns / layers than the workspace is fine on one device
    big = rte.Workspace(64, 20, np.float64)
    np.testing.assert_allclose(rte.compute_col_gas(big, as_.p_lev, params), O.compute_col_gas(as_.p_lev, params), rtol=1e-13)


def test_malformed_views_are_refused():
    """Status codes, not crashes: null pointer, non-positive strides, null view where one is required."""
    import ctypes as C
    from rrtmgp_jl_amd import _abi
    L = _lib.lib()
    ws = rte.Workspace(8, 12, np.float64)
    p_lev = np.asfortranarray(np.linspace(1e5, 1e3, 13)[:, None] * np.ones((1, 8)))
    out = np.zeros((12, 8), order="F")
    pd = RRTMGPParameters().desc()

    def view(a, s0=None, s1=None, null=False):
        v = _abi.View2D()
        v.ptr = None if null else a.ctypes.data
        v.stride0 = a.strides[0] // 8 if s0 is None else s0
        v.stride1 = a.strides[1] // 8 if s1 is None else s1
        return v

    def call(pl, cd, h2o=None):
        return L.rrtmgp_hip_compute_col_gas(ws.handle, _abi.MEM_HOST, 8, 12, C.byref(pl), C.byref(cd), C.byref(pd),
                                            None if h2o is None else C.byref(h2o), None)

    assert call(view(p_lev), view(out)) == 0
    assert call(view(p_lev, null=True), view(out)) != 0 and "null pointer or non-positive stride" in _lib.last_error()
    assert call(view(p_lev, s0=0), view(out)) != 0
    assert call(view(p_lev), view(out, s1=-12)) != 0
    # a WRITTEN view whose elements overlap (stride1 < n0 * stride0 and stride0 < n1 * stride1): the scatter back would race
    big = np.zeros(12 * 8 * 4)
    assert call(view(p_lev), view(big, s0=2, s1=3)) != 0 and "overlap" in _lib.last_error()
    assert call(view(p_lev), view(big, s0=4, s1=48)) == 0        # a row of a (4, nlay, ncol) array: fine
    assert call(view(p_lev), view(big, s0=8, s1=1)) == 0         # a C-ordered (nlay, ncol) block: fine
    # `view(A, 1:2:23, :)` of a 23-row parent: 12 rows, stride 2, columns 23 apart - the last row of a column sits
    # at 22, the next column starts at 23: injective although s1 < n0 * s0 (round 5 refused it)
    odd = np.zeros(23 * 8)
    assert call(view(p_lev), view(odd, s0=2, s1=23)) == 0, _lib.last_error()
    got = odd.reshape(8, 23).T[0:23:2]
    want = np.zeros((12, 8), order="F")
    assert call(view(p_lev), view(want)) == 0
    np.testing.assert_array_equal(got, want)
    assert (odd.reshape(8, 23).T[1:23:2] == 0).all()          # the rows between stay untouched
    assert call(view(p_lev), view(odd, s0=2, s1=22)) != 0 and "overlap" in _lib.last_error()   # one short: (11, j) == (0, j + 1)
    # an absent optional array may be a NULL view or a view with a null pointer (what a binding that always passes a struct does)
    assert call(view(p_lev), view(out), view(out, null=True)) == 0
    assert L.rrtmgp_hip_compute_col_gas(ws.handle, _abi.MEM_HOST, 8, 12, None, C.byref(view(out)), C.byref(pd), None, None) != 0
    assert L.rrtmgp_hip_compute_col_gas(ws.handle, _abi.MEM_HOST, 0, 12, C.byref(view(p_lev)), C.byref(view(out)), C.byref(pd),
                                        None, None) != 0


def test_solver_getters_are_domain_views_and_heating_rate_uses_them(tables64):
    """Layer 2 with an isothermal boundary layer (solver.jl:136-331): the solver's arrays carry one extra layer / level on
    top, every getter with a vertical dimension returns a VIEW without it (getters.jl:40-47) and `heating_rate` hands those
    views and the domain layer count to compute_gray_heating_rate! (standalone.jl:106-122) — one layer fewer than the
    workspace, uncopied."""
    from rrtmgp_jl_amd import solver as L2
    from rrtmgp_jl_amd.states import TEST_PARAMETERS
    t = tables64
    ncol, nlay = 7, 21                        # 20 domain layers + the boundary layer
    as_, lb, sb = S.make_columns(ncol, nlay, np.float64, seed=23, night_fraction=0.2)
    lookups = L2.LookupBundle(t["lw"], t["sw"], t["cld_lw"], t["cld_sw"])
    s = L2.RRTMGPSolver(L2.AllSkyRadiation(reset_rng_seed=True), TEST_PARAMETERS, lb, sb, as_, lookups=lookups,
                        isothermal_boundary_layer=True)
    L2.update_fluxes(s, 7)
    nf, p = L2.net_flux(s), L2.level_pressure(s)
    assert nf.shape == (nlay, ncol) and p.shape == (nlay, ncol)               # nlev - 1 levels
    assert np.shares_memory(nf, s.net_flux_buffer) and np.shares_memory(p, as_.p_lev)
    assert L2.lw_flux_up(s).shape == (nlay, ncol) and L2.layer_temperature(s).shape == (nlay - 1, ncol)
    assert not nf.flags.f_contiguous                                          # a strided view: column stride nlev
    hr = L2.heating_rate(s)
    assert hr.shape == (nlay - 1, ncol)
    np.testing.assert_allclose(hr, TEST_PARAMETERS.grav * (nf[1:] - nf[:-1]) / (p[1:] - p[:-1]) / TEST_PARAMETERS.cp_d,
                               rtol=1e-12)


def test_a_strided_view_moves_its_own_elements_once():
    """a written view with stride0 > 1 used to be staged as the memory span it covers, in AND out — for
    `compute_col_gas!` on `view(layerdata, 1, :, :)` the whole (4, nlay, ncol) array both ways.  Now every host view moves
    exactly its elements: counted by the stager, <= 1.1x the algorithmic bytes (the 256-byte alignment of the pieces)."""
    params = RRTMGPParameters()
    ft = np.float32
    as_ = _state(ft, ncol=4096, nlay=64)
    nlay, ncol = as_.dims
    ws = rte.Workspace(ncol, nlay, ft)
    ld, vmr = as_.layerdata, as_.vmr.vmr
    before = ld.copy(order="F")
    want = O.compute_col_gas(as_.p_lev, params, np.asfortranarray(vmr[0]), as_.lat)
    rte.compute_col_gas(ws, as_.p_lev, params, vmr[0], as_.lat, out=ld[0])          # warm: buffers
    u0, d0 = ws.transfer_bytes()
    ld[0] = -1.0
    rte.compute_col_gas(ws, as_.p_lev, params, vmr[0], as_.lat, out=ld[0])
    u1, d1 = ws.transfer_bytes()
    np.testing.assert_allclose(ld[0], want, rtol=1e-5)
    np.testing.assert_array_equal(ld[1:], before[1:])

    E = 4
    up_alg = (nlay + 1) * ncol * E + nlay * ncol * E + ncol * E        # p_lev + vmr_h2o + lat
    dn_alg = nlay * ncol * E                                           # col_dry
    assert up_alg <= u1 - u0 <= 1.1 * up_alg, (u1 - u0, up_alg)
    assert dn_alg <= d1 - d0 <= 1.1 * dn_alg, (d1 - d0, dn_alg)
    # relative humidity on three rows of ONE parent: each row once, not the parent three times
    h2o = np.asfortranarray(vmr[0])
    rte.compute_relative_humidity(ws, ld[1], ld[2], params, h2o, out=ld[3])
    u2, d2 = ws.transfer_bytes()
    rte.compute_relative_humidity(ws, ld[1], ld[2], params, h2o, out=ld[3])
    u3, d3 = ws.transfer_bytes()
    assert u3 - u2 <= 1.1 * 3 * nlay * ncol * E and d3 - d2 <= 1.1 * nlay * ncol * E


@pytest.mark.parametrize("device", [0, [0, 0, 0]])
def test_column_fastest_views_on_a_sharded_workspace(device):
    """C-order numpy (nlay, ncol) arrays are views with stride1 < stride0.  On a multi-shard workspace
    their column ranges' memory spans overlap almost entirely; round 3 staged each shard's whole span in and out, and
    the concurrent write-backs overwrote other shards' fresh columns with stale data.  Every shard now moves only its own
    elements."""
    params = RRTMGPParameters()
    ft = np.float64
    as_ = _state(ft, ncol=41, nlay=23, full_vmr=False)
    nlay, ncol = as_.dims
    ws = rte.Workspace(ncol, nlay, ft, device)
    p_lev_c = np.ascontiguousarray(as_.p_lev)                  # C order: stride0 = ncol, stride1 = 1
    h2o_c = np.ascontiguousarray(as_.vmr.vmr_h2o)
    out_c = np.full((nlay, ncol), -1.0, order="C")
    assert p_lev_c.strides == (ncol * 8, 8)
    want = O.compute_col_gas(as_.p_lev, params, as_.vmr.vmr_h2o, as_.lat)
    for _ in range(5):                                         # the old failure was a race: give it chances
        out_c[:] = -1.0
        rte.compute_col_gas(ws, p_lev_c, params, h2o_c, as_.lat, out=out_c)
        np.testing.assert_allclose(out_c, want, rtol=1e-13)
    rh_c = np.full((nlay, ncol), -1.0, order="C")
    p_c, t_c = np.ascontiguousarray(as_.layerdata[1]), np.ascontiguousarray(as_.layerdata[2])
    rte.compute_relative_humidity(ws, p_c, t_c, params, h2o_c, out=rh_c)
    np.testing.assert_allclose(rh_c, O.compute_relative_humidity(np.asfortranarray(p_c), np.asfortranarray(t_c), params, as_.vmr.vmr_h2o),
                               rtol=1e-12)


def test_large_strided_views_are_gathered_by_several_threads():
    """From 2^20 elements on the CPU gather / scatter of a strided view is split over a few threads (disjoint column ranges)."""
    params = RRTMGPParameters()
    ft = np.float32
    ncol, nlay = 20000, 64
    rng = np.random.default_rng(0)
    ld = np.asfortranarray(rng.uniform(1.0, 2.0, (4, nlay, ncol)).astype(ft))
    p_lev = np.asfortranarray(np.linspace(1.0e5, 1.0e3, nlay + 1, dtype=ft)[:, None] * np.ones((1, ncol), dtype=ft))
    h2o3 = np.asfortranarray(rng.uniform(0.0, 0.02, (3, nlay, ncol)).astype(ft))
    before = ld.copy(order="F")
    ws = rte.Workspace(ncol, nlay, ft)
    rte.compute_col_gas(ws, p_lev, params, h2o3[1], None, out=ld[0])
    want = O.compute_col_gas(p_lev, params, np.asfortranarray(h2o3[1]), None)
    np.testing.assert_allclose(ld[0], want, rtol=1e-5)
    np.testing.assert_array_equal(ld[1:], before[1:])
    # the helper threads are persistent (three std::threads were spawned and joined per call): many calls, and
    # two workspaces gathering at the same time from two Python threads (the second finds the helpers taken and copies alone)
    import threading
    n_before = threading.active_count()
    for _ in range(5):
        rte.compute_col_gas(ws, p_lev, params, h2o3[1], None, out=ld[0])
    ws2 = rte.Workspace(ncol, nlay, ft)
    ld2 = np.asfortranarray(rng.uniform(1.0, 2.0, (4, nlay, ncol)).astype(ft))
    res = []

    def work(w, dst):
        for _ in range(4):
            rte.compute_col_gas(w, p_lev, params, h2o3[1], None, out=dst[0])
        res.append(np.array_equal(dst[0], ld[0]) or np.allclose(dst[0], want, rtol=1e-5))
    ts = [threading.Thread(target=work, args=(ws, ld)), threading.Thread(target=work, args=(ws2, ld2))]
    [t.start() for t in ts]; [t.join() for t in ts]
    assert res == [True, True] and threading.active_count() == n_before
    np.testing.assert_allclose(ld2[0], want, rtol=1e-5)
