"""Caller-owned state, boundary-condition and flux containers.

Mirrors the reference structs `AtmosphericState`, `CloudState`, `AerosolState`
(src/optics/AtmosphericStates.jl:70-82,236-248,292-298), `VmrGM`/`Vmr`
(src/optics/VolumeMixingRatios.jl:34-80), `LwBCs`/`SwBCs` (src/optics/BCs.jl),
`FluxLW`/`FluxSW` (src/optics/Fluxes.jl:93-149) and `GrayAtmosphericState`
(src/optics/gray_atmospheric_states.jl:100-128).

An array field is either
  * a numpy array in the reference's column-major layout (shape as in Julia,
    order="F")  -> host memory, or
  * a torch tensor on a HIP device whose C-contiguous shape is the REVERSED
    Julia shape (same bytes in memory) -> device memory, used in place.
"""
from __future__ import annotations

import os
import weakref
from dataclasses import dataclass, fields, replace
from typing import Optional

import numpy as np

from . import _abi


def _is_torch(x) -> bool:
    return type(x).__module__.startswith("torch")


# ---- page-locked host arrays: explicit lifetime (include/rrtmgp_hip.h, rrtmgp_hip_host_register) -----------------
# The binding is the owner's agent: a large numpy array is registered the first time it is handed to the library and
# unregistered by a finalizer on the object that owns its memory (the ultimate `.base`), so a registration can never
# outlive its pages.  Arrays below 32 MB stay pageable (they share heap pages with other objects: see the header).
PIN_MIN_BYTES = int(os.environ.get("RRTMGP_HIP_HOST_REGISTER_MIN_BYTES", 32 << 20))
_PIN_OFF = bool(os.environ.get("RRTMGP_HIP_NO_HOST_REGISTER"))
_PINNED = {}   # (address, nbytes) of an owner array -> its finalizer


def _unpin(key):
    from . import _lib
    try:
        if _PINNED.pop(key, None) is not None:
            _lib.lib().rrtmgp_hip_host_unregister(key[0])
    except Exception:   # interpreter shutdown: the library may be gone
        pass


def pin_host_array(a: np.ndarray, min_bytes: Optional[int] = None) -> bool:
    """Page-lock the memory of `a` (its owner array's whole range) until that owner is garbage-collected.  Idempotent;
    returns True when the range is (now) registered."""
    owner = a
    while isinstance(owner.base, np.ndarray):
        owner = owner.base
    if owner.base is not None or owner.nbytes < (PIN_MIN_BYTES if min_bytes is None else min_bytes):
        return False          # memory owned by something else (a buffer, an mmap), or too small
    key = (owner.ctypes.data, owner.nbytes)
    if key in _PINNED:
        return True
    from . import _lib
    if _lib.lib().rrtmgp_hip_host_register(key[0], key[1]) != 0:
        return False
    _PINNED[key] = weakref.finalize(owner, _unpin, key)
    return True


def array_ptr(x):
    """(pointer, mem kind) of a field, (None, None) for None."""
    if x is None:
        return None, None
    if isinstance(x, np.ndarray):
        if not _PIN_OFF and x.nbytes >= PIN_MIN_BYTES:
            pin_host_array(x)
        return _abi.fptr(x), _abi.MEM_HOST
    if _is_torch(x):
        if not x.is_contiguous():
            raise ValueError("device tensors must be contiguous")
        return x.data_ptr(), (_abi.MEM_DEVICE if x.is_cuda else _abi.MEM_HOST)
    raise TypeError(f"unsupported array type {type(x)}")


def view2d(x):
    """(rrtmgp_view2d, mem kind) of a 2-D array in Julia index order, dense or strided — a numpy view such as
    `layerdata[1]` of a (4, nlay, ncol) Fortran array (the reference's `view(as.layerdata, 2, :, :)`), a row of a
    full Vmr, a domain view `x[:n]`; torch tensors (reversed shape) likewise.  (None, None) for None."""
    if x is None:
        return None, None
    v = _abi.View2D()
    if isinstance(x, np.ndarray):
        if x.ndim != 2 or any(s % x.itemsize or s <= 0 for s in x.strides):
            raise ValueError("expected a 2-D array with positive element strides")
        v.ptr, v.stride0, v.stride1 = x.ctypes.data, x.strides[0] // x.itemsize, x.strides[1] // x.itemsize
        return v, _abi.MEM_HOST
    if _is_torch(x):
        if x.dim() != 2 or any(s <= 0 for s in x.stride()):
            raise ValueError("expected a 2-D tensor with positive strides")
        v.ptr, v.stride0, v.stride1 = x.data_ptr(), x.stride(1), x.stride(0)   # torch shape is the Julia shape reversed
        return v, (_abi.MEM_DEVICE if x.is_cuda else _abi.MEM_HOST)
    raise TypeError(f"unsupported array type {type(x)}")


def array_dtype(x):
    if isinstance(x, np.ndarray):
        return x.dtype
    import torch
    return {torch.float32: np.dtype(np.float32), torch.float64: np.dtype(np.float64)}[x.dtype]


def julia_shape(x):
    """Shape in the reference's (Julia) index order."""
    return tuple(x.shape) if isinstance(x, np.ndarray) else tuple(reversed(x.shape))


def to_device(x, device):
    """numpy (Julia layout) -> torch tensor on `device` with reversed shape; same bytes."""
    if x is None or _is_torch(x):
        return x if x is None else x.to(device)
    import torch
    return torch.from_numpy(np.ascontiguousarray(np.asfortranarray(x).T)).to(device)


def to_host(x):
    """torch tensor (reversed shape) -> numpy array in Julia layout."""
    if x is None or isinstance(x, np.ndarray):
        return x
    return np.asfortranarray(x.detach().cpu().numpy().T)


class _Container:
    def _map(self, fn):
        kw = {}
        for f in fields(self):
            v = getattr(self, f.name)
            if isinstance(v, _Container):
                v = v._map(fn)
            elif isinstance(v, np.ndarray) or _is_torch(v):
                v = fn(v)
            kw[f.name] = v
        return replace(self, **kw)

    def to_device(self, device):
        return self._map(lambda a: to_device(a, device))

    def to_host(self):
        return self._map(to_host)

    def _set_ptrs(self, d, names, mems):
        for n in names:
            p, m = array_ptr(getattr(self, n))
            setattr(d, n, p)
            if m is not None:
                mems.add(m)


@dataclass
class VmrGM(_Container):
    vmr_h2o: object  # (nlay, ncol)
    vmr_o3: object   # (nlay, ncol)
    vmr: object      # (ngas)


@dataclass
class Vmr(_Container):
    vmr: object  # (ngas, nlay, ncol)


@dataclass
class CloudState(_Container):
    cld_r_eff_liq: object  # (nlay, ncol) [um]
    cld_r_eff_ice: object
    cld_path_liq: object   # [g/m2]
    cld_path_ice: object
    cld_frac: object
    cld_cover_sw: object = None  # (ncol) out
    cld_cover_lw: object = None  # (ncol) out
    ice_rgh: int = 2


@dataclass
class AerosolState(_Container):
    aero_size: object  # (15, nlay, ncol)
    aero_mass: object  # (15, nlay, ncol)
    aod_sw_ext: object = None  # (ncol) out
    aod_sw_sca: object = None  # (ncol) out


@dataclass
class AtmosphericState(_Container):
    layerdata: object  # (4, nlay, ncol): col_dry, p_lay, t_lay, rel_hum
    p_lev: object      # (nlev, ncol)
    t_lev: object      # (nlev, ncol)
    t_sfc: object      # (ncol)
    vmr: object        # VmrGM | Vmr
    lat: object = None  # (ncol) or None
    cloud_state: Optional[CloudState] = None
    aerosol_state: Optional[AerosolState] = None

    @property
    def dims(self):
        _, nlay, ncol = julia_shape(self.layerdata)
        return nlay, ncol

    @property
    def dtype(self):
        return array_dtype(self.layerdata)

    def desc(self, use_clouds=True, use_aerosols=True) -> _abi.AtmosState:
        d = _abi.AtmosState()
        mems = set()
        nlay, ncol = self.dims
        d.ncol, d.nlay = ncol, nlay
        self._set_ptrs(d, ("layerdata", "p_lev", "t_lev", "t_sfc", "lat"), mems)
        if isinstance(self.vmr, VmrGM):
            d.vmr_kind = _abi.VMR_GM
            d.ngas = julia_shape(self.vmr.vmr)[0]
            self.vmr._set_ptrs(d, ("vmr_h2o", "vmr_o3", "vmr"), mems)
        else:
            d.vmr_kind = _abi.VMR_FULL
            d.ngas = julia_shape(self.vmr.vmr)[0]
            self.vmr._set_ptrs(d, ("vmr",), mems)
        cs = self.cloud_state if use_clouds else None
        if cs is not None:
            cs._set_ptrs(d, ("cld_r_eff_liq", "cld_r_eff_ice", "cld_path_liq", "cld_path_ice", "cld_frac",
                             "cld_cover_lw", "cld_cover_sw"), mems)
            d.ice_rgh = cs.ice_rgh
        aes = self.aerosol_state if use_aerosols else None
        if aes is not None:
            aes._set_ptrs(d, ("aero_size", "aero_mass", "aod_sw_ext", "aod_sw_sca"), mems)
        if len(mems) != 1:
            raise ValueError("all state arrays must live in the same memory space (all host or all device)")
        d.mem = mems.pop()
        return d


@dataclass
class LwBCs(_Container):
    sfc_emis: object          # (nbnd, ncol)
    inc_flux: object = None   # (ncol, ngpt) or None
    inc_flux_ld: int = 0      # > 0: `inc_flux` is a block of columns of a wider (inc_flux_ld, ngpt) array (a view)

    def desc(self) -> _abi.LwBcs:
        d = _abi.LwBcs()
        mems = set()
        if self.inc_flux_ld and self.inc_flux is not None:
            self._set_ptrs(d, ("sfc_emis",), mems)
            x, ld = self.inc_flux, int(self.inc_flux_ld)
            if isinstance(x, np.ndarray):   # (ncol, ngpt) view: unit stride along columns, ld elements between g-points
                if x.strides != (x.itemsize, ld * x.itemsize):
                    raise ValueError("inc_flux must be a column block of a column-major (inc_flux_ld, ngpt) array")
                d.inc_flux, m = x.ctypes.data, _abi.MEM_HOST
            else:                           # torch sees the reversed shape: (ngpt, ncol) with strides (ld, 1)
                if tuple(x.stride()) != (ld, 1):
                    raise ValueError("inc_flux must be a column block of a (ngpt, inc_flux_ld) tensor")
                d.inc_flux, m = x.data_ptr(), (_abi.MEM_DEVICE if x.is_cuda else _abi.MEM_HOST)
            mems.add(m)
            d.inc_flux_ld = ld
        else:
            self._set_ptrs(d, ("sfc_emis", "inc_flux"), mems)
        if len(mems) != 1:
            raise ValueError("sfc_emis and inc_flux must live in the same memory space")
        d.mem = mems.pop()
        return d


@dataclass
class SwBCs(_Container):
    cos_zenith: object       # (ncol)
    toa_flux: object         # (ncol)
    sfc_alb_direct: object   # (nbnd, ncol)
    sfc_alb_diffuse: object  # (nbnd, ncol)
    inc_flux_diffuse: object = None  # stored, never read (shortwave_2stream.jl:331)

    def desc(self) -> _abi.SwBcs:
        d = _abi.SwBcs()
        mems = set()
        self._set_ptrs(d, ("cos_zenith", "toa_flux", "sfc_alb_direct", "sfc_alb_diffuse"), mems)
        d.mem = mems.pop()
        return d


@dataclass
class Flux(_Container):
    """FluxLW (flux_dn_dir is None) or FluxSW."""
    flux_up: object
    flux_dn: object
    flux_net: object
    flux_dn_dir: object = None
    layout: int = _abi.LAYOUT_NLEV_NCOL

    @staticmethod
    def allocate(ncol, nlev, dtype, sw=False, layout=_abi.LAYOUT_NLEV_NCOL, device=None):
        shape = (nlev, ncol) if layout == _abi.LAYOUT_NLEV_NCOL else (ncol, nlev)
        n = 4 if sw else 3
        if device is None:
            arrs = [np.full(shape, np.nan, dtype=dtype, order="F") for _ in range(n)]
        else:
            import torch
            tdt = {np.dtype(np.float32): torch.float32, np.dtype(np.float64): torch.float64}[np.dtype(dtype)]
            arrs = [torch.full(tuple(reversed(shape)), float("nan"), dtype=tdt, device=device) for _ in range(n)]
        return Flux(arrs[0], arrs[1], arrs[2], arrs[3] if sw else None, layout)

    def desc(self, band: "FluxBand" = None, clear: "Flux" = None) -> _abi.FluxOut:
        """`clear`: a second Flux of the same shape that receives the one-pass clear-sky diagnostic."""
        d = _abi.FluxOut()
        mems = set()
        self._set_ptrs(d, ("flux_up", "flux_dn", "flux_net", "flux_dn_dir"), mems)
        if clear is not None:
            if clear.layout != self.layout:
                raise ValueError("clear-sky flux buffers must use the layout of the all-sky ones")
            for n in ("flux_up", "flux_dn", "flux_net", "flux_dn_dir"):
                p, m = array_ptr(getattr(clear, n))
                setattr(d, "clear_" + n, p)
                if m is not None:
                    mems.add(m)
        if band is not None:
            for n in ("flux_up", "flux_dn", "flux_net"):
                p, m = array_ptr(getattr(band, n))
                setattr(d, "band_" + n, p)
                mems.add(m)
        if len(mems) != 1:
            raise ValueError("flux buffers must all live in the same memory space")
        d.mem = mems.pop()
        d.layout = self.layout
        return d

    def as_nlev_ncol(self, name):
        """Host copy of one flux array indexed [ilev, icol] whatever the storage layout."""
        a = to_host(getattr(self, name))
        return a if self.layout == _abi.LAYOUT_NLEV_NCOL else np.asfortranarray(a.T)


@dataclass
class FluxBand(_Container):
    """Optional per-band fluxes (nlev, ncol, n_bnd), src/optics/Fluxes.jl:170-186."""
    flux_up: object
    flux_dn: object
    flux_net: object

    @staticmethod
    def allocate(ncol, nlev, n_bnd, dtype, device=None):
        shape = (nlev, ncol, n_bnd)
        if device is None:
            arrs = [np.full(shape, np.nan, dtype=dtype, order="F") for _ in range(3)]
        else:
            import torch
            tdt = {np.dtype(np.float32): torch.float32, np.dtype(np.float64): torch.float64}[np.dtype(dtype)]
            arrs = [torch.full(tuple(reversed(shape)), float("nan"), dtype=tdt, device=device) for _ in range(3)]
        return FluxBand(*arrs)


@dataclass
class GrayOpticalThicknessSchneider2004:
    """src/optics/gray_atmospheric_states.jl:37-44"""
    alpha: float = 3.5
    te: float = 300.0
    tt: float = 200.0
    dt: float = 60.0
    kind = 0

    def as_array(self):
        return [self.alpha, self.te, self.tt, self.dt, 0.0]


@dataclass
class GrayOpticalThicknessOGorman2008:
    """src/optics/gray_atmospheric_states.jl:75-83"""
    alpha: float = 1.0
    fl: float = 0.2
    tau_e: float = 7.2
    tau_p: float = 1.8
    tau_0: float = 0.22
    kind = 1

    def as_array(self):
        return [self.alpha, self.fl, self.tau_e, self.tau_p, self.tau_0]


@dataclass
class GrayAtmosphericState(_Container):
    lat: object
    p_lay: object
    p_lev: object
    t_lay: object
    t_lev: object
    z_lev: object
    t_sfc: object
    otp: object = None
    stefan: float = 5.670374419e-8

    @property
    def dims(self):
        return julia_shape(self.p_lay)

    @property
    def dtype(self):
        return array_dtype(self.p_lay)

    def desc(self) -> _abi.GrayState:
        d = _abi.GrayState()
        mems = set()
        d.nlay, d.ncol = self.dims
        self._set_ptrs(d, ("lat", "p_lay", "p_lev", "t_lay", "t_lev", "t_sfc"), mems)
        d.mem = mems.pop()
        d.otp_kind = self.otp.kind
        for i, v in enumerate(self.otp.as_array()):
            d.otp[i] = v
        d.stefan = self.stefan
        return d


@dataclass
class RRTMGPParameters:
    """src/Parameters.jl:6-14; defaults from src/api/standalone.jl:87-97."""
    grav: float = 9.81
    molmass_dryair: float = 0.02897
    molmass_water: float = 0.018015
    gas_constant: float = 8.314462618
    kappa_d: float = 2.0 / 7.0
    Stefan: float = 5.670374419e-8
    avogad: float = 6.02214076e23

    @property
    def R_d(self):
        return self.gas_constant / self.molmass_dryair

    @property
    def cp_d(self):
        return self.R_d / self.kappa_d

    def desc(self) -> _abi.Params:
        return _abi.Params(self.grav, self.molmass_dryair, self.molmass_water, self.gas_constant, self.kappa_d,
                           self.Stefan, self.avogad)


# parameter overrides the reference's spectral tests use to match the Fortran code
# (test/clear_sky_utils.jl:40-42)
TEST_PARAMETERS = RRTMGPParameters(grav=9.80665, molmass_dryair=0.028964, molmass_water=0.018016)
