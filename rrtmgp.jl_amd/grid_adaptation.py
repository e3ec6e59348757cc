"""Grid adaptation on the device: the reference's `prepare_atmosphere!` cascade
(src/api/update_fluxes.jl:252-281 over src/api/grid_adaptation.jl:73-292 and
src/api/interpolation.jl:39-252) as ONE HIP launch, in place on the state arrays.

The separable steps keep the reference's names; each is the same kernel with a
different `steps` mask, so a host model can run them one at a time or all at once.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _abi, _lib
from .states import GrayAtmosphericState, Vmr, array_ptr

# AbstractInterpolation / AbstractBottomExtrapolation singletons (interpolation.jl:39-136)
NoInterpolation, ArithmeticMean, GeometricMean = "none", "arithmetic_mean", "geometric_mean"
UniformZ, UniformP, BestFit = "uniform_z", "uniform_p", "best_fit"
SameAsInterpolation, UseSurfaceTempAtBottom, HydrostaticBottom = (
    "same_as_interpolation", "use_surface_temp_at_bottom", "hydrostatic_bottom")


def requires_z(scheme) -> bool:
    """interpolation.jl:145-146"""
    return scheme in (BestFit, HydrostaticBottom)


def make_prepare_opts(steps, interpolation=NoInterpolation, bottom_extrapolation=SameAsInterpolation,
                      isothermal_boundary_layer=False, center_z=None, face_z=None, p_min=0.0, t_min=None, t_max=None,
                      idx_h2o=1) -> _abi.PrepareOpts:
    if interpolation not in _abi.INTERP:
        raise ValueError(f"unknown interpolation scheme {interpolation!r}")
    if bottom_extrapolation not in _abi.BOTTOM:
        raise ValueError(f"unknown bottom extrapolation scheme {bottom_extrapolation!r}")
    if (steps & _abi.PREP_INTERPOLATE) and interpolation != NoInterpolation and \
            (requires_z(interpolation) or requires_z(bottom_extrapolation)) and (center_z is None or face_z is None):
        raise ValueError("BestFit / HydrostaticBottom need `center_z` and `face_z`")
    o = _abi.PrepareOpts()
    o.steps = steps
    o.interpolation = _abi.INTERP[interpolation]
    o.bottom_extrapolation = _abi.BOTTOM[bottom_extrapolation]
    o.isothermal_boundary_layer = int(bool(isothermal_boundary_layer))
    pc, mc = array_ptr(center_z)
    pf, mf = array_ptr(face_z)
    if mc is not None and mf is not None and mc != mf:
        raise ValueError("center_z and face_z must live in the same memory space")
    o.center_z, o.face_z = pc, pf
    o.z_mem = mc if mc is not None else _abi.MEM_HOST
    o.idx_h2o = idx_h2o
    o.p_min = float(p_min)
    # `nothing` bounds skip the clamp (grid_adaptation.jl:248-250): encoded as t_min > t_max
    o.t_min, o.t_max = (1.0, 0.0) if t_min is None or t_max is None else (float(t_min), float(t_max))
    return o


def _run(ws, as_, params, opts: _abi.PrepareOpts):
    pd = params.desc()
    L = _lib.lib()
    if isinstance(as_, GrayAtmosphericState):
        dg = as_.desc()
        _lib.check(L.rrtmgp_hip_prepare_atmosphere_gray(ws.handle, C.byref(dg), C.byref(pd), C.byref(opts)),
                   "prepare_atmosphere (gray)")
    else:
        ds = as_.desc()
        _lib.check(L.rrtmgp_hip_prepare_atmosphere(ws.handle, C.byref(ds), C.byref(pd), C.byref(opts)),
                   "prepare_atmosphere")
    return as_


def interpolate_levels(ws, as_, interpolation, bottom_extrapolation, params, center_z=None, face_z=None,
                       isothermal_boundary_layer=False):
    """interpolate_levels! (grid_adaptation.jl:73-113); a no-op for NoInterpolation."""
    if interpolation == NoInterpolation:
        return as_
    return _run(ws, as_, params, make_prepare_opts(_abi.PREP_INTERPOLATE, interpolation, bottom_extrapolation,
                                                   isothermal_boundary_layer, center_z, face_z))


def add_isothermal_boundary_layer(ws, as_, p_min, params):
    """add_isothermal_boundary_layer! (grid_adaptation.jl:137-173)."""
    return _run(ws, as_, params, make_prepare_opts(_abi.PREP_ISOTHERMAL, isothermal_boundary_layer=True, p_min=p_min))


def clip(ws, as_, p_min, params, idx_h2o=1, t_min=None, t_max=None):
    """clip! (grid_adaptation.jl:215-258)."""
    return _run(ws, as_, params, make_prepare_opts(_abi.PREP_CLIP, p_min=p_min, t_min=t_min, t_max=t_max,
                                                   idx_h2o=idx_h2o))


def update_concentrations(ws, as_, params, idx_h2o=1):
    """update_concentrations! (grid_adaptation.jl:262-292): col_dry only, never relative humidity."""
    if isinstance(as_, GrayAtmosphericState):
        return as_
    return _run(ws, as_, params, make_prepare_opts(_abi.PREP_COL_DRY, idx_h2o=idx_h2o))


def prepare_atmosphere_opts(as_, lookup_lw=None, interpolation=NoInterpolation, bottom_extrapolation=SameAsInterpolation,
                            isothermal_boundary_layer=False, center_z=None, face_z=None,
                            relative_humidity=False) -> _abi.PrepareOpts:
    """The options of the whole cascade (update_fluxes.jl:252-281).  Bounds come from the longwave lookup (get_p_min /
    get_t_min / get_t_max, grid_adaptation.jl:22-53); gray states get p_min = 0 and no temperature clamp."""
    gray = isinstance(as_, GrayAtmosphericState)
    if not gray and lookup_lw is None:
        raise ValueError("a spectral state needs `lookup_lw` for its pressure / temperature bounds")
    p_min = 0.0 if gray else lookup_lw.p_ref_min
    t_min, t_max = (None, None) if gray else (lookup_lw.t_ref_min, lookup_lw.t_ref_max)
    idx_h2o = 1 if gray else lookup_lw.idx_h2o
    steps = _abi.PREP_ALL if interpolation != NoInterpolation else _abi.PREP_ALL & ~_abi.PREP_INTERPOLATE
    if relative_humidity and not gray:
        steps |= _abi.PREP_REL_HUM
    return make_prepare_opts(steps, interpolation, bottom_extrapolation, isothermal_boundary_layer, center_z, face_z, p_min,
                             t_min, t_max, idx_h2o)


def prepare_atmosphere(ws, as_, params, lookup_lw=None, interpolation=NoInterpolation,
                       bottom_extrapolation=SameAsInterpolation, isothermal_boundary_layer=False, center_z=None,
                       face_z=None, relative_humidity=False):
    """The whole cascade in one launch (update_fluxes.jl:252-281).  `relative_humidity=True` also refreshes layerdata
    row 4 from the clipped state in the same launch (compute_relative_humidity!, which the reference's drivers call
    separately)."""
    return _run(ws, as_, params, prepare_atmosphere_opts(as_, lookup_lw, interpolation, bottom_extrapolation,
                                                         isothermal_boundary_layer, center_z, face_z, relative_humidity))
