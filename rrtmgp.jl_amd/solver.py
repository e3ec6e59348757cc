"""Layer-2 solver aggregate: `RRTMGPSolver`, `update_fluxes`, flux getters.

Host-side mirror of src/api/solver.jl:136-331, src/api/update_fluxes.jl:12-281 and the
flux/diagnostic getters of src/api/getters.jl (public.jl:64-123): the call order of the
reference (`prepare_atmosphere!` -> `update_lw_fluxes!` -> `update_sw_fluxes!` ->
`update_net_fluxes!`), the clear-sky-diagnostic double solve, and the `(nlev, ncol)`
presentation of every flux.  The device work behind it is the C ABI of
libhip_rrtmgp.so; nothing here computes fluxes.

`prepare_atmosphere!` (level interpolation, isothermal boundary layer, clipping, col_dry:
src/api/grid_adaptation.jl, interpolation.jl) is one device launch, grid_adaptation.py.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Optional

import numpy as np

from . import _abi, grid_adaptation, rte
from .states import (AtmosphericState, Flux, GrayAtmosphericState, LwBCs, RRTMGPParameters, SwBCs, VmrGM, _Container,
                     array_dtype, to_device, to_host)


# radiation methods, src/api/radiation_methods.jl:19-70
class GrayRadiation:
    pass


@dataclass
class ClearSkyRadiation:
    aerosol_radiation: bool = False


@dataclass
class AllSkyRadiation:
    aerosol_radiation: bool = False
    reset_rng_seed: bool = False


@dataclass
class AllSkyRadiationWithClearSkyDiagnostics:
    aerosol_radiation: bool = False
    reset_rng_seed: bool = False


@dataclass
class LookupBundle:
    """Typed bundle of the lookup tables a method needs (src/api/lookup_bundle.jl)."""
    lookup_lw: object = None
    lookup_sw: object = None
    lookup_lw_cld: object = None
    lookup_sw_cld: object = None
    lookup_lw_aero: object = None
    lookup_sw_aero: object = None


class RRTMGPSolver:
    """RRTMGPSolver(grid, method, params, bcs_lw, bcs_sw, as; ...) of src/api/solver.jl:136.

    `op_lw` / `op_sw` are "twostream" (TwoStream) or "onescalar" (OneScalar) and pick the
    solver exactly as solver.jl:273-310 does.
    """

    def __init__(self, radiation_method, params: RRTMGPParameters, bcs_lw: LwBCs, bcs_sw: SwBCs, as_,
                 op_lw: str = "twostream", op_sw: str = "twostream", deep_atmosphere_inverse_scaling=None,
                 lookups: Optional[LookupBundle] = None, n_gauss_angles: int = 1, device: int = 0,
                 spectral_fluxes: bool = False, interpolation: str = grid_adaptation.NoInterpolation,
                 bottom_extrapolation: str = grid_adaptation.SameAsInterpolation,
                 isothermal_boundary_layer: bool = False, center_z=None, face_z=None, fused: bool = True,
                 resident: bool = False):
        """`fused` (default): `update_fluxes` of a spectral method is ONE call of the library (`rrtmgp_hip_update_fluxes`:
        state staged once, prepare -> LW -> SW -> net on the device); False runs the reference's four steps as four calls.

        `resident=True`: EVERY array of the solver lives in HBM — state, boundary conditions, metric factors, flux / net /
        clear-sky / per-band buffers (torch tensors, the reversed-shape convention of states.to_device) — what the reference
        gets from `array_type(device) = CuArray` (ext/RRTMGPCUDAExt.jl:1-66): host arrays handed to the constructor are moved
        once, `update_fluxes` stages nothing (`Workspace.transfer_bytes() == (0, 0)`), the getters return device VIEWS of the
        solver's buffers, and `to_host()` / `to_device()` are the Adapt round trip (test/standalone.jl:294-335)."""
        self.resident = bool(resident)
        self._torch_device = None
        if self.resident:
            import torch
            if not isinstance(device, int):
                raise ValueError("a resident solver lives on ONE device: pass its ordinal (sharded workspaces take host arrays)")
            dev = self._torch_device = torch.device("cuda", device)
            # (numpy 2 arrays have a `to_device` of their own — the array-API one, "cpu" only: ask for OUR containers by type)
            mv = lambda x: None if x is None else (x.to_device(dev) if isinstance(x, _Container) else to_device(x, dev))  # noqa: E731
            as_, bcs_lw, bcs_sw = mv(as_), mv(bcs_lw), mv(bcs_sw)
            deep_atmosphere_inverse_scaling, center_z, face_z = mv(deep_atmosphere_inverse_scaling), mv(center_z), mv(face_z)
        self.radiation_method, self.params, self.as_ = radiation_method, params, as_
        self.deep_atmosphere_inverse_scaling = deep_atmosphere_inverse_scaling
        self.interpolation, self.bottom_extrapolation = interpolation, bottom_extrapolation
        self.isothermal_boundary_layer, self.center_z, self.face_z = isothermal_boundary_layer, center_z, face_z
        self._ctor = dict(op_lw=op_lw, op_sw=op_sw, n_gauss_angles=n_gauss_angles, device=device, spectral_fluxes=spectral_fluxes,
                          fused=fused)
        if interpolation != grid_adaptation.NoInterpolation and (
                grid_adaptation.requires_z(interpolation) or grid_adaptation.requires_z(bottom_extrapolation)) and (
                center_z is None or face_z is None):
            raise ValueError("BestFit / HydrostaticBottom need `center_z` and `face_z` (solver.jl:183-190)")
        gray = isinstance(radiation_method, GrayRadiation)
        # constructor-time errors of solver.jl:159-181
        if n_gauss_angles != 1:
            if gray:
                raise ValueError("`n_gauss_angles` applies only to spectral radiation; gray radiation uses the "
                                 "single diffusivity angle.")
            if op_lw != "onescalar":
                raise ValueError("`n_gauss_angles` applies only to the non-scattering longwave solver; pass "
                                 "op_lw='onescalar'.")
        if op_sw == "onescalar" and not gray:
            raise ValueError("non-scattering shortwave optics are only supported with GrayRadiation; spectral "
                             "shortwave radiation requires scattering.")
        if not gray and lookups is None:
            raise ValueError("spectral radiation needs `lookups` (the NetCDF loader is outside this back end)")
        if spectral_fluxes:   # solver.jl:252-258
            if gray:
                raise ValueError("spectral_fluxes = true is not supported for GrayRadiation (a single band).")
            if op_lw == "onescalar" or op_sw == "onescalar":
                raise ValueError("spectral_fluxes = true requires two-stream optics for both bands.")
        self.lookups = lookups or LookupBundle()
        nlay, ncol = as_.dims
        dtype = as_.dtype
        self.nlay, self.ncol, self.dtype = nlay, ncol, dtype
        ws = rte.Workspace(ncol, nlay, dtype, device)  # LW and SW share the library scratch
        if resident:
            # Device arrays are used in place and the library's work is ordered on the WORKSPACE stream: a resident solver runs
            # it on torch's current stream, behind the fills / copies that made its arrays and in front of whatever the caller
            # does with the getters' views next.  (On its private non-blocking stream nothing ordered the first launch behind the
            # NaN fill of the flux buffers: a gray step, which uploads no lookup in between, could lose that race - round 6.)
            ws.use_torch_stream()
        lw_cls = rte.TwoStreamLWRTE if op_lw == "twostream" else rte.NoScatLWRTE
        sw_cls = rte.TwoStreamSWRTE if op_sw == "twostream" else rte.NoScatSWRTE
        # compute buffers ARE the (nlev, ncol) presentation: update_presentation! is a no-op here
        nb_lw = self.lookups.lookup_lw.n_bnd if spectral_fluxes else 0
        nb_sw = self.lookups.lookup_sw.n_bnd if spectral_fluxes else 0
        fdev = self._torch_device   # None: host buffers
        self.lws = lw_cls(ncol, nlay, dtype, bcs_lw, n_gauss_angles=n_gauss_angles, workspace=ws,
                          n_bnd_band_flux=nb_lw, flux_device=fdev)
        self.sws = sw_cls(ncol, nlay, dtype, bcs_sw, workspace=ws, n_bnd_band_flux=nb_sw, flux_device=fdev)

        def zeros():   # (nlev, ncol) in the reference's index order, where the solver's arrays live
            a = np.zeros((nlay + 1, ncol), dtype=dtype, order="F")
            return a if fdev is None else to_device(a, fdev)
        self.net_flux_buffer = zeros()
        diag = isinstance(radiation_method, AllSkyRadiationWithClearSkyDiagnostics)
        self.clear_flux_lw = Flux.allocate(ncol, nlay + 1, dtype, sw=False, device=fdev) if diag else None
        self.clear_flux_sw = Flux.allocate(ncol, nlay + 1, dtype, sw=True, device=fdev) if diag else None
        self.clear_net_flux_buffer = zeros() if diag else None
        self.fused = bool(fused)   # gray radiation too (round 6: rrtmgp_hip_update_fluxes_gray)
        self._seed = 0       # key of the counter-based McICA stream of the current update_fluxes call
        self._rng_state = 0  # host generator the per-call keys are drawn from (the reference's global `Random` state)

    # ---- prepare_atmosphere!, update_fluxes.jl:252-281: one device launch -------------------------
    def prepare_atmosphere(self):
        grid_adaptation.prepare_atmosphere(self.lws.ws, self.as_, self.params, self.lookups.lookup_lw,
                                           self.interpolation, self.bottom_extrapolation,
                                           self.isothermal_boundary_layer, self.center_z, self.face_z)

    # ---- update_lw_fluxes!, update_fluxes.jl:12-65 ------------------------------------------------
    def update_lw_fluxes(self):
        m, lk, ms = self.radiation_method, self.lookups, self.deep_atmosphere_inverse_scaling
        if isinstance(m, GrayRadiation):
            rte.solve_lw(self.lws, self.as_, metric_scaling=ms)
            return
        aero = lk.lookup_lw_aero if m.aerosol_radiation else None
        if isinstance(m, ClearSkyRadiation):
            rte.solve_lw(self.lws, self.as_, lk.lookup_lw, None, aero, ms, seed=self._seed)
        elif isinstance(m, AllSkyRadiation):
            rte.solve_lw(self.lws, self.as_, lk.lookup_lw, lk.lookup_lw_cld, aero, ms, seed=self._seed)
        elif self.lws.twostream and self.lws.band_flux is None:
            # one launch carries both recurrences (the reference solves twice, update_fluxes.jl:39-65)
            rte.solve_lw(self.lws, self.as_, lk.lookup_lw, lk.lookup_lw_cld, aero, ms, seed=self._seed,
                         clear_flux=self.clear_flux_lw)
        else:
            rte.solve_lw(self.lws, self.as_, lk.lookup_lw, None, aero, ms, seed=self._seed)
            for n in ("flux_up", "flux_dn", "flux_net"):
                getattr(self.clear_flux_lw, n)[...] = getattr(self.lws.flux, n)
            rte.solve_lw(self.lws, self.as_, lk.lookup_lw, lk.lookup_lw_cld, aero, ms, seed=self._seed)

    # ---- update_sw_fluxes!, update_fluxes.jl:74-128 -----------------------------------------------
    def update_sw_fluxes(self):
        m, lk, ms = self.radiation_method, self.lookups, self.deep_atmosphere_inverse_scaling
        if isinstance(m, GrayRadiation):
            rte.solve_sw(self.sws, self.as_, metric_scaling=ms)
            return
        aero = lk.lookup_sw_aero if m.aerosol_radiation else None
        if isinstance(m, ClearSkyRadiation):
            rte.solve_sw(self.sws, self.as_, lk.lookup_sw, None, aero, ms, seed=self._seed)
        elif isinstance(m, AllSkyRadiation):
            rte.solve_sw(self.sws, self.as_, lk.lookup_sw, lk.lookup_sw_cld, aero, ms, seed=self._seed)
        elif self.sws.twostream and self.sws.band_flux is None:
            rte.solve_sw(self.sws, self.as_, lk.lookup_sw, lk.lookup_sw_cld, aero, ms, seed=self._seed,
                         clear_flux=self.clear_flux_sw)
        else:
            rte.solve_sw(self.sws, self.as_, lk.lookup_sw, None, aero, ms, seed=self._seed)
            for n in ("flux_up", "flux_dn", "flux_net", "flux_dn_dir"):
                getattr(self.clear_flux_sw, n)[...] = getattr(self.sws.flux, n)
            rte.solve_sw(self.sws, self.as_, lk.lookup_sw, lk.lookup_sw_cld, aero, ms, seed=self._seed)

    # ---- update_net_fluxes!, update_fluxes.jl:165-194 ----------------------------------------------
    def update_net_fluxes(self):
        # The UNFUSED path only (`fused=False`, the reference's four separate steps; the fused step forms both sums inside
        # rrtmgp_hip_update_fluxes).  A mirror of the reference's host broadcast `net .= lw .+ sw` (Fluxes.jl:407-424): numpy
        # for host arrays, torch for resident ones - torch is plumbing for the caller's device arrays here, not the product path.
        if self.resident:
            import torch
            add = torch.add
        else:
            add = np.add
        add(self.lws.flux.flux_net, self.sws.flux.flux_net, out=self.net_flux_buffer)
        if self.clear_net_flux_buffer is not None:
            add(self.clear_flux_lw.flux_net, self.clear_flux_sw.flux_net, out=self.clear_net_flux_buffer)

    # ---- Adapt round trip (test/standalone.jl:294-335; ext/RRTMGPCUDAExt.jl Adapt rules) -----------------------------
    _ARRAY_SLOTS = ("net_flux_buffer", "clear_net_flux_buffer", "deep_atmosphere_inverse_scaling", "center_z", "face_z")

    def _adapted(self, resident: bool) -> "RRTMGPSolver":
        """A solver with fresh copies of every array in the other memory: same method, lookups, options, RNG state and
        CURRENT contents (state, boundary conditions, fluxes), so that a checkpoint restored on the other side continues
        where this one stood.  The getters of the result return views of ITS buffers."""
        if resident and self._torch_device is None:
            import torch
            dev = torch.device("cuda", self._ctor["device"] if isinstance(self._ctor["device"], int) else 0)
        else:
            dev = self._torch_device
        mv = (lambda x: None if x is None else (x.to_device(dev) if isinstance(x, _Container) else to_device(x, dev))) if resident else \
             (lambda x: None if x is None else (x.to_host() if isinstance(x, _Container) else to_host(x)))
        kw = dict(self._ctor)
        if resident and not isinstance(kw["device"], int):
            raise ValueError("a resident solver lives on ONE device")
        t = RRTMGPSolver(self.radiation_method, self.params, mv(self.lws.bcs), mv(self.sws.bcs), mv(self.as_),
                         deep_atmosphere_inverse_scaling=mv(self.deep_atmosphere_inverse_scaling), lookups=self.lookups,
                         interpolation=self.interpolation, bottom_extrapolation=self.bottom_extrapolation,
                         isothermal_boundary_layer=self.isothermal_boundary_layer, center_z=mv(self.center_z),
                         face_z=mv(self.face_z), resident=resident, **kw)
        if resident == self.resident:   # a plain copy: arrays must still be fresh
            cp = (lambda x: x.clone()) if resident else (lambda x: x.copy(order="F"))
            t.as_, t.lws.bcs, t.sws.bcs = t.as_._map(cp), t.lws.bcs._map(cp), t.sws.bcs._map(cp)
        for a, b in ((self.lws.flux, t.lws.flux), (self.sws.flux, t.sws.flux), (self.clear_flux_lw, t.clear_flux_lw),
                     (self.clear_flux_sw, t.clear_flux_sw), (self.lws.band_flux, t.lws.band_flux),
                     (self.sws.band_flux, t.sws.band_flux)):
            if a is not None:
                for n in ("flux_up", "flux_dn", "flux_net", "flux_dn_dir"):
                    if getattr(a, n, None) is not None:
                        setattr(b, n, mv(getattr(a, n)) if resident != self.resident else
                                (getattr(a, n).clone() if resident else getattr(a, n).copy(order="F")))
        for n in ("net_flux_buffer", "clear_net_flux_buffer"):
            v = getattr(self, n)
            if v is not None:
                setattr(t, n, mv(v) if resident != self.resident else (v.clone() if resident else v.copy(order="F")))
        t._seed, t._rng_state = self._seed, self._rng_state
        return t

    def to_host(self) -> "RRTMGPSolver":
        return self._adapted(False)

    def to_device(self) -> "RRTMGPSolver":
        return self._adapted(True)

    # ---- update_fluxes!, update_fluxes.jl:223-233 ----------------------------------------------------
    def update_fluxes(self, seedval=None):
        m = self.radiation_method
        # _maybe_reset_rng_seed! (update_fluxes.jl:149-156): `Random.seed!(seedval)` only when the method asks for it
        # AND a seed is given; otherwise the generator keeps advancing, so successive radiation steps draw
        # independent McICA samples (frozen sampling would leave a persistent per-column bias).  Here one draw
        # of a host splitmix64 generator keys the counter-based device stream of this call.
        if getattr(m, "reset_rng_seed", False) and seedval is not None:
            self._rng_state = int(seedval) & 0xFFFFFFFFFFFFFFFF
        self._rng_state = (self._rng_state + 0x9E3779B97F4A7C15) & 0xFFFFFFFFFFFFFFFF
        z = self._rng_state
        z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & 0xFFFFFFFFFFFFFFFF
        z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & 0xFFFFFFFFFFFFFFFF
        self._seed = z ^ (z >> 31)
        if self.fused:
            self._update_fluxes_fused()
            return
        self.prepare_atmosphere()
        self.update_lw_fluxes()
        self.update_sw_fluxes()
        self.update_net_fluxes()

    def _update_fluxes_fused(self):
        """The four steps above as one library call: the state is staged once, `prepare_atmosphere!` runs as a kernel in
        front of the two solves, the clear-sky diagnostic rides in the all-sky launches, and `net_flux` / `clear_net_flux`
        are summed on the device (update_fluxes.jl:223-233)."""
        m, lk = self.radiation_method, self.lookups
        if isinstance(m, GrayRadiation):
            prep = grid_adaptation.prepare_atmosphere_opts(self.as_, None, self.interpolation, self.bottom_extrapolation,
                                                           self.isothermal_boundary_layer, self.center_z, self.face_z)
            rte.update_fluxes_gray(self.lws, self.sws, self.as_, metric_scaling=self.deep_atmosphere_inverse_scaling,
                                   net_flux=self.net_flux_buffer, params=self.params, prepare=prep)
            return
        clouds = not isinstance(m, ClearSkyRadiation)
        aero = m.aerosol_radiation
        prep = grid_adaptation.prepare_atmosphere_opts(self.as_, lk.lookup_lw, self.interpolation, self.bottom_extrapolation,
                                                       self.isothermal_boundary_layer, self.center_z, self.face_z)
        rte.update_fluxes(self.lws, self.sws, self.as_, lk.lookup_lw, lk.lookup_sw,
                          lk.lookup_lw_cld if clouds else None, lk.lookup_sw_cld if clouds else None,
                          lk.lookup_lw_aero if aero else None, lk.lookup_sw_aero if aero else None,
                          metric_scaling=self.deep_atmosphere_inverse_scaling, seed=self._seed,
                          net_flux=self.net_flux_buffer, clear_flux_lw=self.clear_flux_lw, clear_flux_sw=self.clear_flux_sw,
                          clear_net_flux=self.clear_net_flux_buffer, params=self.params, prepare=prep)


def update_fluxes(s: RRTMGPSolver, seedval=None):
    s.update_fluxes(seedval)


# ---- getters (src/api/getters.jl; names of public.jl:96-118) ----------------------------------------
def _domain_view(s, x):
    """_domain_view (src/api/getters.jl:40-47): every getter with a vertical dimension returns a VIEW of the solver's
    buffer without the extra isothermal boundary layer / level (`view(x, 1:size(x, 1) - Int(bl), :)`), the whole array
    when there is none.  numpy arrays are in the reference's index order (vertical first); device tensors carry the
    reversed shape (vertical last)."""
    if x is None:
        return None
    bl = int(bool(s.isothermal_boundary_layer))
    if isinstance(x, np.ndarray):
        return x[:x.shape[0] - bl]
    return x[..., :x.shape[-1] - bl]


def lw_flux_up(s): return _domain_view(s, s.lws.flux.flux_up)
def lw_flux_dn(s): return _domain_view(s, s.lws.flux.flux_dn)
def lw_flux_net(s): return _domain_view(s, s.lws.flux.flux_net)
def sw_flux_up(s): return _domain_view(s, s.sws.flux.flux_up)
def sw_flux_dn(s): return _domain_view(s, s.sws.flux.flux_dn)
def sw_flux_net(s): return _domain_view(s, s.sws.flux.flux_net)
def sw_direct_flux_dn(s): return _domain_view(s, s.sws.flux.flux_dn_dir)
def net_flux(s): return _domain_view(s, s.net_flux_buffer)


def _solver_band_flux(ws):   # getters.jl:398-404
    if not ws.twostream:
        raise ValueError("spectral fluxes require a two-stream, non-gray solver.")
    if ws.band_flux is None:
        raise ValueError("spectral fluxes were not retained; construct the `RRTMGPSolver` with "
                         "`spectral_fluxes = true`.")
    return ws.band_flux


def spectral_lw_flux_up(s): return _domain_view(s, _solver_band_flux(s.lws).flux_up)
def spectral_lw_flux_dn(s): return _domain_view(s, _solver_band_flux(s.lws).flux_dn)
def spectral_lw_flux_net(s): return _domain_view(s, _solver_band_flux(s.lws).flux_net)
def spectral_sw_flux_up(s): return _domain_view(s, _solver_band_flux(s.sws).flux_up)
def spectral_sw_flux_dn(s): return _domain_view(s, _solver_band_flux(s.sws).flux_dn)
def spectral_sw_flux_net(s): return _domain_view(s, _solver_band_flux(s.sws).flux_net)
def lw_band_bounds(s): return s.lookups.lookup_lw.bnd_lims_wn
def sw_band_bounds(s): return s.lookups.lookup_sw.bnd_lims_wn
def clear_lw_flux_up(s): return _domain_view(s, s.clear_flux_lw.flux_up)
def clear_lw_flux_dn(s): return _domain_view(s, s.clear_flux_lw.flux_dn)
def clear_lw_flux_net(s): return _domain_view(s, s.clear_flux_lw.flux_net)
def clear_sw_flux_up(s): return _domain_view(s, s.clear_flux_sw.flux_up)
def clear_sw_flux_dn(s): return _domain_view(s, s.clear_flux_sw.flux_dn)
def clear_sw_direct_flux_dn(s): return _domain_view(s, s.clear_flux_sw.flux_dn_dir)
def clear_sw_flux_net(s): return _domain_view(s, s.clear_flux_sw.flux_net)
def clear_net_flux(s): return _domain_view(s, s.clear_net_flux_buffer)
def lw_cloud_cover(s): return s.as_.cloud_state.cld_cover_lw
def sw_cloud_cover(s): return s.as_.cloud_state.cld_cover_sw
def aod_sw_extinction(s): return s.as_.aerosol_state.aod_sw_ext
def aod_sw_scattering(s): return s.as_.aerosol_state.aod_sw_sca
def level_pressure(s): return _domain_view(s, s.as_.p_lev)
def layer_pressure(s): return _domain_view(s, s.as_.p_lay if isinstance(s.as_, GrayAtmosphericState) else s.as_.layerdata[1])
def layer_temperature(s): return _domain_view(s, s.as_.t_lay if isinstance(s.as_, GrayAtmosphericState) else s.as_.layerdata[2])
def level_temperature(s): return _domain_view(s, s.as_.t_lev)
def surface_temperature(s): return s.as_.t_sfc


def heating_rate(s: RRTMGPSolver):
    """heating_rate (src/api/standalone.jl:100-122): (g / cp) dF_net/dp per layer [K/s]; fresh array.  Like the reference it
    hands `level_pressure(s)` and `net_flux(s)` — domain VIEWS — and the domain layer count to the device method; nothing
    is copied on the way (rrtmgp_view2d)."""
    p_lev, f_net = level_pressure(s), net_flux(s)
    if isinstance(f_net, np.ndarray) and not isinstance(p_lev, np.ndarray):
        p_lev = to_host(p_lev)   # a device-resident state next to the host flux buffers: one memory kind per call
    return rte.compute_gray_heating_rate(s.lws.ws, p_lev, f_net, s.params.cp_d, s.params.grav)
