"""Layer-1 solver workspaces and `solve_lw` / `solve_sw`, bound to libhip_rrtmgp.so.

Host-side mirror of the reference's `NoScatLWRTE`, `TwoStreamLWRTE`, `NoScatSWRTE`,
`TwoStreamSWRTE` (src/rte/RTE.jl:53,111,177,229) and `solve_lw!` / `solve_sw!`
(src/rte/RTESolver.jl:33-247): same argument order and meaning; the device dispatch
on `context.device` becomes a call through the C ABI.  Julia's `solve_lw!` is
`solve_lw` here.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional

import numpy as np

from . import _abi, _lib
from .lookups import GasLookup, LookUpAerosolMerra, LookUpCld
from .states import (AtmosphericState, Flux, FluxBand, GrayAtmosphericState, LwBCs, SwBCs, array_dtype, array_ptr,
                     julia_shape, view2d)


def _device_key(device):
    """A device is an int (one GPU) or a sequence of ints (one shard per entry; ids may repeat)."""
    return device if isinstance(device, int) else tuple(int(d) for d in device)


class DeviceLookup:
    """A lookup table uploaded to HBM (handle owned by this object).  `device` may be a sequence of
    device ids: one replica per distinct device behind one handle (`*_lookup_create_multi`)."""

    def __init__(self, host, device=0):
        self.host = host
        self.device = _device_key(device)
        self.handle = C.c_void_p()
        d = host.desc()
        L = _lib.lib()
        kind = ("gas" if isinstance(host, GasLookup) else "cloud" if isinstance(host, LookUpCld)
                else "aerosol" if isinstance(host, LookUpAerosolMerra) else None)
        if kind is None:
            raise TypeError(type(host))
        if isinstance(self.device, int):
            rc = getattr(L, f"rrtmgp_hip_{kind}_lookup_create")(C.byref(d), self.device, C.byref(self.handle))
        else:
            ids = (C.c_int32 * len(self.device))(*self.device)
            rc = getattr(L, f"rrtmgp_hip_{kind}_lookup_create_multi")(C.byref(d), ids, len(self.device), C.byref(self.handle))
        _lib.check(rc, "lookup upload")

    def __del__(self):
        try:
            if self.handle:
                _lib.lib().rrtmgp_hip_lookup_destroy(self.handle)
                self.handle = C.c_void_p()
        except Exception:
            pass


def _dev(lk, device):
    if lk is None:
        return None
    if isinstance(lk, DeviceLookup):
        return lk
    cache = getattr(lk, "_device_cache", None)
    if cache is None:
        cache = {}
        object.__setattr__(lk, "_device_cache", cache)
    if device not in cache:
        cache[device] = DeviceLookup(lk, device)
    return cache[device]


class Workspace:
    """Library-owned scratch for one (ncol, nlay, FT): what the reference keeps in
    `op`, `src`, `fluxb`, `state_cache` and the masks (src/rte/RTE.jl)."""

    def __init__(self, ncol: int, nlay: int, dtype, device=0):
        """`device`: an int, or a sequence of device ids = one column shard per entry (ids may repeat), solved
        concurrently inside the library (`rrtmgp_hip_workspace_create_multi`; host arrays, (nlev, ncol) fluxes)."""
        self.ncol, self.nlay, self.dtype, self.device = ncol, nlay, np.dtype(dtype), _device_key(device)
        self.handle = C.c_void_p()
        if isinstance(self.device, int):
            _lib.check(_lib.lib().rrtmgp_hip_workspace_create(self.device, ncol, nlay, _abi.ftype_of(dtype),
                                                              C.byref(self.handle)), "workspace_create")
        else:
            ids = (C.c_int32 * len(self.device))(*self.device)
            _lib.check(_lib.lib().rrtmgp_hip_workspace_create_multi(ids, len(self.device), ncol, nlay, _abi.ftype_of(dtype),
                                                                    C.byref(self.handle)), "workspace_create_multi")

    @property
    def n_shards(self) -> int:
        return _lib.lib().rrtmgp_hip_workspace_shards(self.handle)

    def set_stream(self, stream_ptr: Optional[int]):
        _lib.check(_lib.lib().rrtmgp_hip_workspace_set_stream(self.handle, C.c_void_p(stream_ptr or 0)), "set_stream")

    def use_torch_stream(self):
        import torch
        self.set_stream(torch.cuda.current_stream(self.device).cuda_stream)  # single-device workspaces only

    def synchronize(self):
        _lib.check(_lib.lib().rrtmgp_hip_workspace_synchronize(self.handle), "synchronize")

    def last_kernel_ms(self) -> float:
        ms = C.c_double()
        _lib.check(_lib.lib().rrtmgp_hip_workspace_last_kernel_ms(self.handle, C.byref(ms)), "last_kernel_ms")
        return ms.value

    def transfer_bytes(self):
        """(host -> device, device -> host) bytes this workspace has staged since it was created."""
        a, b = C.c_uint64(), C.c_uint64()
        _lib.check(_lib.lib().rrtmgp_hip_workspace_transfer_bytes(self.handle, C.byref(a), C.byref(b)), "transfer_bytes")
        return a.value, b.value

    def __del__(self):
        try:
            if self.handle:
                _lib.lib().rrtmgp_hip_workspace_destroy(self.handle)
                self.handle = C.c_void_p()
        except Exception:
            pass


def _opts(n_gauss_angles, metric_scaling, seed, col_offset):
    o = _abi.SolveOpts()
    o.n_gauss_angles = n_gauss_angles
    p, m = array_ptr(metric_scaling)
    o.metric_scaling = p
    o.metric_mem = m if m is not None else _abi.MEM_HOST
    o.seed = seed
    o.col_offset = col_offset
    return o


class _RTE:
    """Common part of the four workspaces: boundary conditions, flux buffers, library scratch."""
    twostream = True
    sw = False

    def __init__(self, ncol, nlay, dtype, bcs, device=0, flux_device=None, layout=_abi.LAYOUT_NLEV_NCOL,
                 n_gauss_angles=1, workspace: Optional[Workspace] = None, n_bnd_band_flux: int = 0):
        """`n_bnd_band_flux` > 0 allocates the optional FluxBand (RTE.jl:106,127,224,245);
        only the two-stream workspaces carry one."""
        self.bcs = bcs
        self.n_gauss_angles = n_gauss_angles
        self.ws = workspace or Workspace(ncol, nlay, dtype, device)
        self.flux = Flux.allocate(ncol, nlay + 1, dtype, sw=self.sw, layout=layout, device=flux_device)
        self.band_flux = None
        if n_bnd_band_flux:
            if not self.twostream:
                raise ValueError("spectral fluxes require a two-stream, non-gray solver (getters.jl:404)")
            self.band_flux = FluxBand.allocate(ncol, nlay + 1, n_bnd_band_flux, dtype, device=flux_device)

    @property
    def device(self):
        return self.ws.device


class NoScatLWRTE(_RTE):
    twostream, sw = False, False


class TwoStreamLWRTE(_RTE):
    twostream, sw = True, False


class NoScatSWRTE(_RTE):
    twostream, sw = False, True


class TwoStreamSWRTE(_RTE):
    twostream, sw = True, True


def _null(h):
    return None if h is None else h.handle


def _check_precision(slv: "_RTE", as_, *lookups):
    """The C structs carry raw pointers: element type and extents are the caller's contract.  The
    reference gets the same guarantees from its type parameters (`RTE{FT}`, `AtmosphericState{FT}`)."""
    want = slv.ws.dtype
    got = np.dtype(as_.dtype)
    if got != want:
        raise TypeError(f"state arrays are {got}, the workspace was created for {want}")
    for f in ("flux_up",):
        if np.dtype(array_dtype(getattr(slv.flux, f))) != want:
            raise TypeError(f"flux buffers are not {want}")
    for lk in lookups:
        host = getattr(lk, "host", lk)
        if host is not None and np.dtype(host.dtype) != want:
            raise TypeError(f"lookup tables are {np.dtype(host.dtype)}, the workspace was created for {want}")
    nlay, ncol = as_.dims
    if nlay != slv.ws.nlay or ncol != slv.ws.ncol:  # the rule the library applies (check_common, staging.hip)
        raise ValueError(f"state is (nlay={nlay}, ncol={ncol}); the workspace was created for "
                         f"(nlay={slv.ws.nlay}, ncol={slv.ws.ncol})")


def solve_lw(slv: _RTE, as_, lookup_lw=None, lookup_lw_cld=None, lookup_lw_aero=None, metric_scaling=None,
             seed: int = 0, col_offset: int = 0, clear_flux: Optional[Flux] = None) -> Flux:
    """solve_lw! (RTESolver.jl:33,54,77,117).  Gray when `as_` is a GrayAtmosphericState.
    `clear_flux`: also produce the clear-sky fluxes in the same launch (two-stream + cloud lookup)."""
    L = _lib.lib()
    _check_precision(slv, as_, lookup_lw, lookup_lw_cld, lookup_lw_aero)
    o = _opts(slv.n_gauss_angles, metric_scaling, seed, col_offset)
    db, df = slv.bcs.desc(), slv.flux.desc(slv.band_flux, clear_flux)
    if isinstance(as_, GrayAtmosphericState):
        if slv.n_gauss_angles != 1:
            raise ValueError("gray radiation is solved with a single quadrature angle")
        dg = as_.desc()
        fn = L.rrtmgp_hip_rte_lw_2stream_solve_gray if slv.twostream else L.rrtmgp_hip_rte_lw_noscat_solve_gray
        _lib.check(fn(slv.ws.handle, C.byref(dg), C.byref(db), C.byref(df), C.byref(o)), "solve_lw (gray)")
        return slv.flux
    lw, cld, aero = _dev(lookup_lw, slv.device), _dev(lookup_lw_cld, slv.device), _dev(lookup_lw_aero, slv.device)
    ds = as_.desc(cld is not None, aero is not None)
    fn = L.rrtmgp_hip_rte_lw_2stream_solve if slv.twostream else L.rrtmgp_hip_rte_lw_noscat_solve
    _lib.check(fn(slv.ws.handle, lw.handle, _null(cld), _null(aero), C.byref(ds), C.byref(db), C.byref(df), C.byref(o)),
               "solve_lw")
    return slv.flux


def solve_sw(slv: _RTE, as_, lookup_sw=None, lookup_sw_cld=None, lookup_sw_aero=None, metric_scaling=None,
             seed: int = 0, col_offset: int = 0, clear_flux: Optional[Flux] = None) -> Flux:
    """solve_sw! (RTESolver.jl:151,167,188,222)."""
    L = _lib.lib()
    _check_precision(slv, as_, lookup_sw, lookup_sw_cld, lookup_sw_aero)
    o = _opts(1, metric_scaling, seed, col_offset)
    db, df = slv.bcs.desc(), slv.flux.desc(slv.band_flux, clear_flux)
    if isinstance(as_, GrayAtmosphericState):
        dg = as_.desc()
        fn = L.rrtmgp_hip_rte_sw_2stream_solve_gray if slv.twostream else L.rrtmgp_hip_rte_sw_noscat_solve_gray
        _lib.check(fn(slv.ws.handle, C.byref(dg), C.byref(db), C.byref(df), C.byref(o)), "solve_sw (gray)")
        return slv.flux
    sw = _dev(lookup_sw, slv.device)
    if slv.twostream:
        cld, aero = _dev(lookup_sw_cld, slv.device), _dev(lookup_sw_aero, slv.device)
        ds = as_.desc(cld is not None, aero is not None)
        _lib.check(L.rrtmgp_hip_rte_sw_2stream_solve(slv.ws.handle, sw.handle, _null(cld), _null(aero), C.byref(ds),
                                                     C.byref(db), C.byref(df), C.byref(o)), "solve_sw")
    else:
        if lookup_sw_cld is not None or lookup_sw_aero is not None:
            raise ValueError("NoScatSWRTE takes no cloud / aerosol lookups (RTESolver.jl:188-209)")
        ds = as_.desc(False, False)
        _lib.check(L.rrtmgp_hip_rte_sw_noscat_solve(slv.ws.handle, sw.handle, C.byref(ds), C.byref(db), C.byref(df),
                                                    C.byref(o)), "solve_sw")
    return slv.flux


def update_fluxes(lws: _RTE, sws: _RTE, as_, lookup_lw, lookup_sw, lookup_lw_cld=None, lookup_sw_cld=None, lookup_lw_aero=None,
                  lookup_sw_aero=None, metric_scaling=None, seed: int = 0, col_offset: int = 0, net_flux=None,
                  clear_flux_lw: Optional[Flux] = None, clear_flux_sw: Optional[Flux] = None, clear_net_flux=None,
                  params=None, prepare: Optional[_abi.PrepareOpts] = None):
    """The whole radiation step in ONE call of the library (`rrtmgp_hip_update_fluxes`): the state crosses to the device
    once, then [`prepare` kernel] -> LW solve -> SW solve -> `net_flux = lw net + sw net` (update_fluxes!,
    src/api/update_fluxes.jl:223-233).  `lws` / `sws` are the two solver workspaces (they must share one library
    workspace); `clear_flux_*` receive the clear-sky fluxes of AllSkyRadiationWithClearSkyDiagnostics; `prepare` = the
    options of prepare_atmosphere! (grid_adaptation.make_prepare_opts), None when the state is already prepared."""
    if lws.ws is not sws.ws:
        raise ValueError("update_fluxes: the longwave and the shortwave solver must share one Workspace")
    if isinstance(as_, GrayAtmosphericState):
        raise TypeError("update_fluxes is the spectral step; gray radiation goes through solve_lw / solve_sw")
    if not sws.twostream:
        raise ValueError("spectral shortwave radiation requires the two-stream solver (solver.jl:176-182)")
    _check_precision(lws, as_, lookup_lw, lookup_lw_cld, lookup_lw_aero)
    _check_precision(sws, as_, lookup_sw, lookup_sw_cld, lookup_sw_aero)
    dev = lws.device
    lk = [_dev(x, dev) for x in (lookup_lw, lookup_sw, lookup_lw_cld, lookup_sw_cld, lookup_lw_aero, lookup_sw_aero)]
    a = _abi.UpdateFluxesArgs()
    (a.lookup_lw, a.lookup_sw, a.lookup_lw_cld, a.lookup_sw_cld, a.lookup_lw_aero, a.lookup_sw_aero) = [_null(x) for x in lk]
    # with a preparation step the library gets the WHOLE state: the isothermal boundary layer fills the extra layer of every
    # cloud / aerosol array the state carries, whatever the radiation method reads (grid_adaptation.jl)
    full = prepare is not None
    ds = as_.desc(full or lk[2] is not None or lk[3] is not None, full or lk[4] is not None or lk[5] is not None)
    bl, bs = lws.bcs.desc(), sws.bcs.desc()
    fl, fs = lws.flux.desc(lws.band_flux, clear_flux_lw), sws.flux.desc(sws.band_flux, clear_flux_sw)
    o = _opts(lws.n_gauss_angles, metric_scaling, seed, col_offset)
    a.as_, a.bcs_lw, a.bcs_sw, a.flux_lw, a.flux_sw, a.opts = (C.pointer(ds), C.pointer(bl), C.pointer(bs), C.pointer(fl),
                                                               C.pointer(fs), C.pointer(o))
    for name, arr in (("net_flux", net_flux), ("clear_net_flux", clear_net_flux)):
        if arr is None:
            continue
        ptr, mem = array_ptr(arr)
        if mem != fl.mem:
            raise ValueError(f"update_fluxes: {name} must live where the flux buffers live")
        if tuple(julia_shape(arr)) != (lws.ws.nlay + 1, lws.ws.ncol):
            raise ValueError(f"update_fluxes: {name} must be (nlev, ncol)")
        setattr(a, name, ptr)
    keep = None
    if prepare is not None:
        if params is None:
            raise ValueError("update_fluxes: `params` is required with `prepare`")
        keep = params.desc()
        a.params, a.prepare = C.pointer(keep), C.pointer(prepare)
    a.lw_solver = _abi.LW_TWOSTREAM if lws.twostream else _abi.LW_NOSCAT
    _lib.check(_lib.lib().rrtmgp_hip_update_fluxes(lws.ws.handle, C.byref(a)), "update_fluxes")
    return lws.flux, sws.flux


def update_fluxes_gray(lws: _RTE, sws: _RTE, as_, metric_scaling=None, net_flux=None, params=None,
                       prepare: Optional[_abi.PrepareOpts] = None):
    """update_fluxes! for GrayRadiation in ONE call of the library (`rrtmgp_hip_update_fluxes_gray`): the gray state crosses
    once, then [`prepare` kernel] -> gray LW -> gray SW -> `net_flux = lw net + sw net` (update_fluxes.jl:223-233 with the gray
    methods :19-23 / :81-85).  What a resident solver needs: the reference's generic path forms the presentation copies and
    the net sum with array broadcasts."""
    if lws.ws is not sws.ws:
        raise ValueError("update_fluxes_gray: the longwave and the shortwave solver must share one Workspace")
    if not isinstance(as_, GrayAtmosphericState):
        raise TypeError("update_fluxes_gray takes a GrayAtmosphericState")
    _check_precision(lws, as_)
    _check_precision(sws, as_)
    a = _abi.UpdateFluxesGrayArgs()
    dg, bl, bs = as_.desc(), lws.bcs.desc(), sws.bcs.desc()
    fl, fs = lws.flux.desc(), sws.flux.desc()
    o = _opts(1, metric_scaling, 0, 0)
    a.as_, a.bcs_lw, a.bcs_sw, a.flux_lw, a.flux_sw, a.opts = (C.pointer(dg), C.pointer(bl), C.pointer(bs), C.pointer(fl),
                                                               C.pointer(fs), C.pointer(o))
    if net_flux is not None:
        ptr, mem = array_ptr(net_flux)
        if mem != fl.mem:
            raise ValueError("update_fluxes_gray: net_flux must live where the flux buffers live")
        if tuple(julia_shape(net_flux)) != (lws.ws.nlay + 1, lws.ws.ncol):
            raise ValueError("update_fluxes_gray: net_flux must be (nlev, ncol)")
        a.net_flux = ptr
    keep = None
    if prepare is not None:
        if params is None:
            raise ValueError("update_fluxes_gray: `params` is required with `prepare`")
        keep = params.desc()
        a.params, a.prepare = C.pointer(keep), C.pointer(prepare)
    a.lw_solver = _abi.LW_TWOSTREAM if lws.twostream else _abi.LW_NOSCAT
    a.sw_twostream = 1 if sws.twostream else 0
    _lib.check(_lib.lib().rrtmgp_hip_update_fluxes_gray(lws.ws.handle, C.byref(a)), "update_fluxes_gray")
    return lws.flux, sws.flux


def _check_extents(ws: Workspace, what: str, **arrays):
    for name, (a, want) in arrays.items():
        if a is None:
            continue
        got = tuple(julia_shape(a))
        if got != tuple(want):
            raise ValueError(f"{what}: {name} has shape {got}, expected {tuple(want)}")
        if np.dtype(array_dtype(a)) != ws.dtype:
            raise TypeError(f"{what}: {name} is {np.dtype(array_dtype(a))}, the workspace was created for {ws.dtype}")


def _views(*arrays):
    """rrtmgp_view2d descriptors (by reference, None -> NULL) of arrays that share one memory kind."""
    vs, mems = [], set()
    for a in arrays:
        v, m = view2d(a)
        vs.append(v)
        if m is not None:
            mems.add(m)
    if len(mems) != 1:
        raise ValueError("arrays must all be host or all be device memory")
    return [None if v is None else C.byref(v) for v in vs], mems.pop(), vs


def compute_col_gas(ws: Workspace, p_lev, params, vmr_h2o=None, lat=None, out=None):
    """compute_col_gas! (src/optics/column_amounts.jl:14-43) on the device.  Like the reference's method the arrays
    may be strided views: `out = layerdata[0]` of a `(4, nlay, ncol)` array (`getview_col_dry`,
    AtmosphericStates.jl:96-97), `vmr_h2o = vmr[idx_h2o - 1]` of a full Vmr (grid_adaptation.jl:204)."""
    nlev, ncol = julia_shape(p_lev)
    if out is None:
        out = np.empty((nlev - 1, ncol), dtype=array_dtype(p_lev), order="F")
    _check_extents(ws, "compute_col_gas", col_dry=(out, (nlev - 1, ncol)), vmr_h2o=(vmr_h2o, (nlev - 1, ncol)), lat=(lat, (ncol,)),
                   p_lev=(p_lev, (nlev, ncol)))
    (pl, cd, h2o), mem, keep = _views(p_lev, out, vmr_h2o)
    pd = params.desc()
    _lib.check(_lib.lib().rrtmgp_hip_compute_col_gas(ws.handle, mem, ncol, nlev - 1, pl, cd, C.byref(pd), h2o,
                                                     array_ptr(lat)[0]), "compute_col_gas")
    return out


def compute_relative_humidity(ws: Workspace, p_lay, t_lay, params, vmr_h2o, out=None):
    """compute_relative_humidity! (src/optics/column_amounts.jl:52-76) on the device; strided views as above
    (the reference's drivers pass rows 4, 2, 3 of layerdata, test/read_clear_sky.jl:162-169)."""
    nlay, ncol = julia_shape(p_lay)
    if out is None:
        out = np.empty((nlay, ncol), dtype=array_dtype(p_lay), order="F")
    full = (nlay, ncol)
    _check_extents(ws, "compute_relative_humidity", rh=(out, full), p_lay=(p_lay, full), t_lay=(t_lay, full),
                   vmr_h2o=(vmr_h2o, full))
    (r, p, t, h), mem, keep = _views(out, p_lay, t_lay, vmr_h2o)
    pd = params.desc()
    _lib.check(_lib.lib().rrtmgp_hip_compute_relative_humidity(ws.handle, mem, ncol, nlay, r, p, t, C.byref(pd), h),
               "compute_relative_humidity")
    return out


def compute_gray_heating_rate(ws: Workspace, p_lev, flux_net, cp_d: float, grav: float, out=None):
    """compute_gray_heating_rate! (src/optics/GrayAtmosphere.jl:133-167) on the device:
    hr(nlay, ncol) = grav (F_net[k+1] - F_net[k]) / (p_lev[k+1] - p_lev[k]) / cp_d.  `p_lev` / `flux_net` may be the
    getters' domain views (`x[:nlev_domain]`, src/api/getters.jl:42-43): nlay is then one less than the workspace's."""
    nlev, ncol = julia_shape(p_lev)
    if out is None:   # a fresh array where the inputs live (the reference's `similar`)
        if isinstance(p_lev, np.ndarray):
            out = np.empty((nlev - 1, ncol), dtype=array_dtype(p_lev), order="F")
        else:
            import torch
            out = torch.empty((ncol, nlev - 1), dtype=p_lev.dtype, device=p_lev.device)   # reversed-shape convention
    lev = (nlev, ncol)
    _check_extents(ws, "compute_gray_heating_rate", p_lev=(p_lev, lev), flux_net=(flux_net, lev), hr_lay=(out, (nlev - 1, ncol)))
    (hr, pl, fn), mem, keep = _views(out, p_lev, flux_net)
    _lib.check(_lib.lib().rrtmgp_hip_compute_gray_heating_rate(ws.handle, mem, ncol, nlev - 1, hr, pl, fn,
                                                               float(cp_d), float(grav)), "compute_gray_heating_rate")
    return out
