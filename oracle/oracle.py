"""TEST INFRASTRUCTURE: ctypes driver for the plain-C parity oracle (oracle/rrtmgp_oracle.c).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this
module.  It reuses the product's struct definitions (rrtmgp_jl_amd._abi) because the
oracle implements the same C ABI with host pointers; the product never imports it.

Parity status: see the header of oracle/rrtmgp_oracle.h (real-data parity unpinned).
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

import rrtmgp_jl_amd  # noqa: F401  (registers the package)
from rrtmgp_jl_amd import _abi
from rrtmgp_jl_amd.states import Flux

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "build", "librrtmgp_oracle.so")
_lib = None


def build(force: bool = False) -> str:
    srcs = [os.path.join(_HERE, f) for f in ("rrtmgp_oracle.c", "rrtmgp_oracle_impl.inc", "rrtmgp_oracle.h")]
    srcs.append(os.path.join(_HERE, "..", "include", "rrtmgp_hip.h"))
    stale = force or not os.path.exists(_SO) or any(os.path.getmtime(s) > os.path.getmtime(_SO) for s in srcs)
    if stale:
        subprocess.run(["make", "-C", _HERE], check=True, capture_output=True)
    return _SO


def n_threads() -> int:
    """OpenMP threads the oracle uses: the cores this process may run on (not every core
    of the host), unless OMP_NUM_THREADS is set."""
    if "OMP_NUM_THREADS" in os.environ:
        return int(os.environ["OMP_NUM_THREADS"])
    try:
        return max(1, len(os.sched_getaffinity(0)))
    except AttributeError:
        return max(1, os.cpu_count() or 1)


def load_library(path):
    """A build of the oracle at `path` with its entry points typed (tests/test_oracle_mutations.py loads deliberately
    broken builds through this)."""
    os.environ.setdefault("OMP_NUM_THREADS", str(min(n_threads(), 32)))
    os.environ.setdefault("OMP_WAIT_POLICY", "PASSIVE")
    L = C.CDLL(path)
    L.rrtmgp_oracle_mcica_uniform.restype = C.c_double
    L.rrtmgp_oracle_mcica_uniform.argtypes = [C.c_uint64, C.c_int64, C.c_int64, C.c_int32, C.c_int32]
    L.rrtmgp_oracle_interp1d_equispaced.restype = C.c_double
    L.rrtmgp_oracle_loc_lower_eq.restype = C.c_int64
    L.rrtmgp_oracle_loc_lower.restype = C.c_int64
    return L


def lib():
    global _lib
    if _lib is None:
        _lib = load_library(build())
    return _lib


class using:
    """`with using(other_build): ...` — every oracle call inside runs on `other_build` (a load_library handle)."""

    def __init__(self, other):
        self.other = other

    def __enter__(self):
        global _lib
        lib()
        self.saved, _lib = _lib, self.other
        return self.other

    def __exit__(self, *exc):
        global _lib
        _lib = self.saved
        return False


def _check(rc, what):
    if rc != 0:
        raise RuntimeError(f"oracle {what} failed: {_abi.ERRORS.get(rc, rc)}")


def _opts(n_gauss_angles=1, metric_scaling=None, seed=0, col_offset=0):
    o = _abi.SolveOpts()
    o.n_gauss_angles = n_gauss_angles
    o.metric_mem = _abi.MEM_HOST
    o.metric_scaling = _abi.fptr(metric_scaling)
    o.seed = seed
    o.col_offset = col_offset
    return o


def _byref_or_none(d):
    return None if d is None else C.byref(d)


def solve_lw(as_, bcs, lookup_lw, lookup_cld=None, lookup_aero=None, twostream=True, n_gauss_angles=1,
             metric_scaling=None, seed=0, col_offset=0, layout=_abi.LAYOUT_NLEV_NCOL, band_flux=None, clear_flux=None) -> Flux:
    """`band_flux`: a states.FluxBand to fill (two-stream only), or None.  `clear_flux`: a Flux that
    receives the clear-sky diagnostic (the reference's first solve of the pair), or None."""
    nlay, ncol = as_.dims
    flux = Flux.allocate(ncol, nlay + 1, as_.dtype, sw=False, layout=layout)
    dl = lookup_lw.desc()
    dc = None if lookup_cld is None else lookup_cld.desc()
    da = None if lookup_aero is None else lookup_aero.desc()
    ds, db, df = as_.desc(lookup_cld is not None, lookup_aero is not None), bcs.desc(), flux.desc(band_flux, clear_flux)
    o = _opts(n_gauss_angles, metric_scaling, seed, col_offset)
    fn = lib().rrtmgp_oracle_rte_lw_2stream_solve if twostream else lib().rrtmgp_oracle_rte_lw_noscat_solve
    _check(fn(C.byref(dl), _byref_or_none(dc), _byref_or_none(da), C.byref(ds), C.byref(db), C.byref(df), C.byref(o)),
           "solve_lw")
    return flux


def solve_sw(as_, bcs, lookup_sw, lookup_cld=None, lookup_aero=None, twostream=True, metric_scaling=None, seed=0,
             col_offset=0, layout=_abi.LAYOUT_NLEV_NCOL, band_flux=None, clear_flux=None) -> Flux:
    nlay, ncol = as_.dims
    flux = Flux.allocate(ncol, nlay + 1, as_.dtype, sw=True, layout=layout)
    dl = lookup_sw.desc()
    dc = None if lookup_cld is None else lookup_cld.desc()
    da = None if lookup_aero is None else lookup_aero.desc()
    ds, db, df = as_.desc(lookup_cld is not None, lookup_aero is not None), bcs.desc(), flux.desc(band_flux, clear_flux)
    o = _opts(1, metric_scaling, seed, col_offset)
    if twostream:
        rc = lib().rrtmgp_oracle_rte_sw_2stream_solve(C.byref(dl), _byref_or_none(dc), _byref_or_none(da), C.byref(ds),
                                                      C.byref(db), C.byref(df), C.byref(o))
    else:
        rc = lib().rrtmgp_oracle_rte_sw_noscat_solve(C.byref(dl), C.byref(ds), C.byref(db), C.byref(df), C.byref(o))
    _check(rc, "solve_sw")
    return flux


def solve_lw_gray(gs, bcs, twostream=True, metric_scaling=None, layout=_abi.LAYOUT_NLEV_NCOL) -> Flux:
    nlay, ncol = gs.dims
    flux = Flux.allocate(ncol, nlay + 1, gs.dtype, sw=False, layout=layout)
    dg, db, df, o = gs.desc(), bcs.desc(), flux.desc(), _opts(1, metric_scaling)
    fn = lib().rrtmgp_oracle_rte_lw_2stream_solve_gray if twostream else lib().rrtmgp_oracle_rte_lw_noscat_solve_gray
    _check(fn(C.byref(dg), C.byref(db), C.byref(df), C.byref(o), _abi.ftype_of(gs.dtype)), "solve_lw_gray")
    return flux


def solve_sw_gray(gs, bcs, twostream=True, metric_scaling=None, layout=_abi.LAYOUT_NLEV_NCOL) -> Flux:
    nlay, ncol = gs.dims
    flux = Flux.allocate(ncol, nlay + 1, gs.dtype, sw=True, layout=layout)
    dg, db, df, o = gs.desc(), bcs.desc(), flux.desc(), _opts(1, metric_scaling)
    fn = lib().rrtmgp_oracle_rte_sw_2stream_solve_gray if twostream else lib().rrtmgp_oracle_rte_sw_noscat_solve_gray
    _check(fn(C.byref(dg), C.byref(db), C.byref(df), C.byref(o), _abi.ftype_of(gs.dtype)), "solve_sw_gray")
    return flux


def prepare_atmosphere(as_, params, steps=_abi.PREP_ALL, **kw):
    """The prepare_atmosphere! cascade on host arrays, in place; `kw` as
    rrtmgp_jl_amd.grid_adaptation.make_prepare_opts."""
    from rrtmgp_jl_amd.grid_adaptation import make_prepare_opts
    from rrtmgp_jl_amd.states import GrayAtmosphericState
    o, pd = make_prepare_opts(steps, **kw), params.desc()
    if isinstance(as_, GrayAtmosphericState):
        d = as_.desc()
        _check(lib().rrtmgp_oracle_prepare_atmosphere_gray(_abi.ftype_of(as_.dtype), C.byref(d), C.byref(pd),
                                                           C.byref(o)), "prepare_atmosphere_gray")
    else:
        d = as_.desc()
        _check(lib().rrtmgp_oracle_prepare_atmosphere(_abi.ftype_of(as_.dtype), C.byref(d), C.byref(pd), C.byref(o)),
               "prepare_atmosphere")
    return as_


def compute_col_gas(p_lev, params, vmr_h2o=None, lat=None):
    nlev, ncol = p_lev.shape
    col_dry = np.empty((nlev - 1, ncol), dtype=p_lev.dtype, order="F")
    pd = params.desc()
    _check(lib().rrtmgp_oracle_compute_col_gas(_abi.ftype_of(p_lev.dtype), C.c_int64(ncol), C.c_int64(nlev - 1),
                                               C.c_void_p(_abi.fptr(p_lev)), C.c_void_p(_abi.fptr(col_dry)),
                                               C.byref(pd), C.c_void_p(_abi.fptr(vmr_h2o)), C.c_void_p(_abi.fptr(lat))),
           "compute_col_gas")
    return col_dry


def compute_relative_humidity(p_lay, t_lay, params, vmr_h2o):
    nlay, ncol = p_lay.shape
    rh = np.empty((nlay, ncol), dtype=p_lay.dtype, order="F")
    pd = params.desc()
    _check(lib().rrtmgp_oracle_compute_relative_humidity(_abi.ftype_of(p_lay.dtype), C.c_int64(ncol), C.c_int64(nlay),
                                                         C.c_void_p(_abi.fptr(rh)), C.c_void_p(_abi.fptr(p_lay)),
                                                         C.c_void_p(_abi.fptr(t_lay)), C.byref(pd),
                                                         C.c_void_p(_abi.fptr(vmr_h2o))), "compute_relative_humidity")
    return rh


def mcica_uniform(seed, gcol, igpt, is_sw, draw) -> float:
    return lib().rrtmgp_oracle_mcica_uniform(seed, gcol, igpt, int(is_sw), draw)


def build_cloud_mask(cld_frac, seed, gcol, igpt, is_sw):
    cld_frac = np.ascontiguousarray(cld_frac)
    mask = np.zeros(cld_frac.shape[0], dtype=np.uint8)
    any_ = lib().rrtmgp_oracle_build_cloud_mask(_abi.ftype_of(cld_frac.dtype), C.c_void_p(mask.ctypes.data),
                                                C.c_void_p(cld_frac.ctypes.data), C.c_int64(cld_frac.shape[0]),
                                                C.c_uint64(seed), C.c_int64(gcol), C.c_int64(igpt), C.c_int32(int(is_sw)))
    return mask.astype(bool), bool(any_)


def angular_discretization(n, dtype=np.float64):
    Ds, wts = np.zeros(n, dtype=dtype), np.zeros(n, dtype=dtype)
    lib().rrtmgp_oracle_angular_discretization(C.c_int(n), C.c_size_t(np.dtype(dtype).itemsize),
                                               C.c_void_p(Ds.ctypes.data), C.c_void_p(wts.ctypes.data))
    return Ds, wts


def rte_lw_noscat_one_angle(tau, lay_source, lev_source, sfc_source, sfc_emis, inc_flux, Ds, w_mu):
    dt = tau.dtype
    nlay = tau.shape[0]
    up, dn = np.zeros(nlay + 1, dtype=dt), np.zeros(nlay + 1, dtype=dt)
    f = lib().rrtmgp_oracle_rte_lw_noscat_one_angle
    f(C.c_int32(_abi.ftype_of(dt)), C.c_int64(nlay), C.c_void_p(tau.ctypes.data), C.c_void_p(lay_source.ctypes.data),
      C.c_void_p(lev_source.ctypes.data), C.c_double(sfc_source), C.c_double(sfc_emis),
      C.c_int(inc_flux is not None), C.c_double(0.0 if inc_flux is None else inc_flux), C.c_double(Ds),
      C.c_double(w_mu), C.c_void_p(up.ctypes.data), C.c_void_p(dn.ctypes.data))
    return up, dn


def lw_2stream_coeffs(tau, ssa, g, bot, top, dtype=np.float64):
    out = (C.c_double * 4)()
    lib().rrtmgp_oracle_lw_2stream_coeffs(C.c_int32(_abi.ftype_of(dtype)), C.c_double(tau), C.c_double(ssa),
                                          C.c_double(g), C.c_double(bot), C.c_double(top), out)
    return tuple(out)


def sw_2stream_coeffs(tau, ssa, g, mu0, dtype=np.float64):
    out = (C.c_double * 5)()
    lib().rrtmgp_oracle_sw_2stream_coeffs(C.c_int32(_abi.ftype_of(dtype)), C.c_double(tau), C.c_double(ssa),
                                          C.c_double(g), C.c_double(mu0), out)
    return tuple(out)


def setup_gray_as_pr_grid(nlay, lat, p0, pe, otp, params, dtype=np.float64):
    """setup_gray_as_pr_grid (gray_atmospheric_states.jl:152-223): returns GrayAtmosphericState."""
    from rrtmgp_jl_amd.states import GrayAtmosphericState
    lat = np.ascontiguousarray(lat, dtype=dtype)
    ncol = lat.shape[0]
    mk = lambda n: np.zeros((n, ncol), dtype=dtype, order="F")
    p_lev, p_lay, t_lev, t_lay, z_lev = mk(nlay + 1), mk(nlay), mk(nlay + 1), mk(nlay), mk(nlay + 1)
    t_sfc = np.zeros(ncol, dtype=dtype)
    v = lambda a: C.c_void_p(a.ctypes.data)
    _check(lib().rrtmgp_oracle_setup_gray_as_pr_grid(
        C.c_int32(_abi.ftype_of(dtype)), C.c_int64(ncol), C.c_int64(nlay), v(lat), C.c_double(p0), C.c_double(pe),
        C.c_double(300.0), C.c_double(200.0), C.c_double(60.0), C.c_double(3.5), C.c_double(params.R_d),
        C.c_double(params.grav), v(p_lev), v(p_lay), v(t_lev), v(t_lay), v(z_lev), v(t_sfc)), "setup_gray")
    return GrayAtmosphericState(lat, p_lay, p_lev, t_lay, t_lev, z_lev, t_sfc, otp, params.Stefan)


def gray_heating_rate(flux_net, p_lev, grav, cp_d):
    nlev, ncol = p_lev.shape
    hr = np.zeros((nlev - 1, ncol), dtype=p_lev.dtype, order="F")
    v = lambda a: C.c_void_p(a.ctypes.data)
    _check(lib().rrtmgp_oracle_gray_heating_rate(C.c_int32(_abi.ftype_of(p_lev.dtype)), C.c_int64(ncol),
                                                 C.c_int64(nlev - 1), v(hr), v(flux_net), v(p_lev), C.c_double(grav),
                                                 C.c_double(cp_d)), "gray_heating_rate")
    return hr


def update_profile_lw(sbc, t_lay, t_lev, hr_lay, flux_dn, flux_net, dt):
    nlay, ncol = t_lay.shape
    flux_grad = np.zeros((nlay, ncol), dtype=t_lay.dtype, order="F")
    T_ex = np.zeros((nlay + 1, ncol), dtype=t_lay.dtype, order="F")
    v = lambda a: C.c_void_p(a.ctypes.data)
    _check(lib().rrtmgp_oracle_update_profile_lw(C.c_int32(_abi.ftype_of(t_lay.dtype)), C.c_int64(ncol),
                                                 C.c_int64(nlay), C.c_double(sbc), v(t_lay), v(t_lev), v(hr_lay),
                                                 v(flux_dn), v(flux_net), v(flux_grad), v(T_ex), C.c_double(dt)),
           "update_profile_lw")
    return flux_grad, T_ex
