"""Lookup-table containers in the reference's in-memory form.

Mirrors src/optics/LookUpTables.jl of the reference: `LookUpMinor` (:36-41),
`LookUpLW` (:130-143), `LookUpSW` (:185-201), `LookUpCld` (:239-284),
`LookUpAerosolMerra` (:312-325).  Arrays are numpy, column-major (order="F"), with
the same shapes and 1-based integer tables (int64) as the Julia structs, so
`.desc()` can hand their pointers to the C ABI unchanged.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass, fields
from typing import Optional

import numpy as np

from . import _abi


def _f(a, dt):
    return None if a is None else np.asfortranarray(a, dtype=dt)


def _i(a):
    return None if a is None else np.asfortranarray(a, dtype=np.int64)


@dataclass
class LookUpMinor:
    bnd_st: np.ndarray   # (n_bnd+1) int64, 1-based
    gpt_st: np.ndarray   # (n_gpt+1) int64, 1-based
    gasdata: np.ndarray  # (4, n_min_absrb) int64
    kminor: np.ndarray   # (n_eta, n_t_ref, n_contrib) FT

    def desc(self) -> _abi.MinorDesc:
        d = _abi.MinorDesc()
        d.n_min_absrb = self.gasdata.shape[1]
        d.n_contrib = self.kminor.shape[2]
        d.bnd_st = _abi.fptr(self.bnd_st, np.int64)
        d.gpt_st = _abi.fptr(self.gpt_st, np.int64)
        d.gasdata = _abi.fptr(self.gasdata, np.int64)
        d.kminor = _abi.fptr(self.kminor)
        return d


@dataclass
class GasLookup:
    """LookUpLW (is_sw=False) or LookUpSW (is_sw=True)."""
    is_sw: bool
    idx_h2o: int
    p_ref_tropo: float
    p_ref_min: float
    t_ref_min: float
    t_ref_max: float
    key_species: np.ndarray    # (2, 2, n_bnd) int64
    kmajor: np.ndarray         # (n_eta, n_p_ref+1, n_t_ref, n_gpt)
    major_gpt2bnd: np.ndarray  # (n_gpt) int64
    bnd_lims_wn: np.ndarray    # (2, n_bnd)
    ln_p_ref: np.ndarray       # (n_p_ref)
    t_ref: np.ndarray          # (n_t_ref)
    vmr_ref: np.ndarray        # (2, n_gases, n_t_ref)
    minor_lower: LookUpMinor
    minor_upper: LookUpMinor
    # LW only
    planck_fraction: Optional[np.ndarray] = None
    t_planck: Optional[np.ndarray] = None
    tot_planck: Optional[np.ndarray] = None
    # SW only
    solar_src_tot: float = 0.0
    rayl_lower: Optional[np.ndarray] = None
    rayl_upper: Optional[np.ndarray] = None
    solar_src_scaled: Optional[np.ndarray] = None

    @property
    def dtype(self):
        return self.kmajor.dtype

    @property
    def n_gpt(self):
        return self.kmajor.shape[3]

    @property
    def n_bnd(self):
        return self.key_species.shape[2]

    @property
    def n_gases(self):
        return self.vmr_ref.shape[1]

    def astype(self, dt) -> "GasLookup":
        kw = {}
        for f in fields(self):
            v = getattr(self, f.name)
            if isinstance(v, np.ndarray) and v.dtype.kind == "f":
                v = np.asfortranarray(v, dtype=dt)
            elif isinstance(v, LookUpMinor):
                v = LookUpMinor(v.bnd_st, v.gpt_st, v.gasdata, np.asfortranarray(v.kminor, dtype=dt))
            kw[f.name] = v
        return GasLookup(**kw)

    def desc(self) -> _abi.GasLookupDesc:
        dt = self.dtype
        d = _abi.GasLookupDesc()
        d.ftype = _abi.ftype_of(dt)
        d.is_sw = int(self.is_sw)
        n_eta, n_p1, n_t, n_gpt = self.kmajor.shape
        d.n_gpt, d.n_bnd, d.n_eta, d.n_p_ref, d.n_t_ref = n_gpt, self.n_bnd, n_eta, n_p1 - 1, n_t
        assert self.ln_p_ref.shape[0] == n_p1 - 1 and self.t_ref.shape[0] == n_t
        d.n_gases = self.n_gases
        d.n_t_plnk = 0 if self.t_planck is None else self.t_planck.shape[0]
        d.idx_h2o = self.idx_h2o
        d.p_ref_tropo, d.p_ref_min = self.p_ref_tropo, self.p_ref_min
        d.t_ref_min, d.t_ref_max = self.t_ref_min, self.t_ref_max
        d.solar_src_tot = self.solar_src_tot
        d.key_species = _abi.fptr(self.key_species, np.int64)
        d.major_gpt2bnd = _abi.fptr(self.major_gpt2bnd, np.int64)
        d.kmajor = _abi.fptr(self.kmajor, dt)
        d.planck_fraction = _abi.fptr(self.planck_fraction, dt)
        d.t_planck = _abi.fptr(self.t_planck, dt)
        d.tot_planck = _abi.fptr(self.tot_planck, dt)
        d.ln_p_ref = _abi.fptr(self.ln_p_ref, dt)
        d.t_ref = _abi.fptr(self.t_ref, dt)
        d.vmr_ref = _abi.fptr(self.vmr_ref, dt)
        d.minor_lower = self.minor_lower.desc()
        d.minor_upper = self.minor_upper.desc()
        d.rayl_lower = _abi.fptr(self.rayl_lower, dt)
        d.rayl_upper = _abi.fptr(self.rayl_upper, dt)
        d.solar_src_scaled = _abi.fptr(self.solar_src_scaled, dt)
        return d


@dataclass
class LookUpCld:
    dims: np.ndarray     # int64 (5): nband, nrghice, nsize_liq, nsize_ice, pair
    bounds: np.ndarray   # FT (4)
    liqdata: np.ndarray  # FT (3*nsize_liq, nband)
    icedata: np.ndarray  # FT (3*nsize_ice, nband, nrghice)

    @property
    def dtype(self):
        return self.liqdata.dtype

    def astype(self, dt) -> "LookUpCld":
        return LookUpCld(self.dims, _f(self.bounds, dt), _f(self.liqdata, dt), _f(self.icedata, dt))

    def desc(self) -> _abi.CloudLookupDesc:
        dt = self.dtype
        d = _abi.CloudLookupDesc()
        d.ftype = _abi.ftype_of(dt)
        d.nband, d.nrghice, d.nsize_liq, d.nsize_ice = (int(x) for x in self.dims[:4])
        assert self.liqdata.shape == (3 * d.nsize_liq, d.nband)
        assert self.icedata.shape == (3 * d.nsize_ice, d.nband, d.nrghice)
        d.bounds = _abi.fptr(self.bounds, dt)
        d.liqdata = _abi.fptr(self.liqdata, dt)
        d.icedata = _abi.fptr(self.icedata, dt)
        return d


@dataclass
class LookUpAerosolMerra:
    size_bin_limits: np.ndarray    # (2, nbin)
    rh_levels: np.ndarray          # (nrh)
    dust: np.ndarray               # (3, nbin, nband)
    sea_salt: np.ndarray           # (3, nrh, nbin, nband)
    sulfate: np.ndarray            # (3, nrh, nband)
    black_carbon_rh: np.ndarray    # (3, nrh, nband)
    black_carbon: np.ndarray       # (3, nband)
    organic_carbon_rh: np.ndarray  # (3, nrh, nband)
    organic_carbon: np.ndarray     # (3, nband)
    iband_550nm: int = 0

    @property
    def dtype(self):
        return self.dust.dtype

    def astype(self, dt) -> "LookUpAerosolMerra":
        kw = {f.name: (_f(getattr(self, f.name), dt) if isinstance(getattr(self, f.name), np.ndarray)
                       else getattr(self, f.name)) for f in fields(self)}
        return LookUpAerosolMerra(**kw)

    def desc(self) -> _abi.AerosolLookupDesc:
        dt = self.dtype
        d = _abi.AerosolLookupDesc()
        d.ftype = _abi.ftype_of(dt)
        d.nbin = self.size_bin_limits.shape[1]
        d.nrh = self.rh_levels.shape[0]
        d.nband = self.dust.shape[2]
        d.iband_550nm = self.iband_550nm
        for name in ("size_bin_limits", "rh_levels", "dust", "sea_salt", "sulfate", "black_carbon_rh",
                     "black_carbon", "organic_carbon_rh", "organic_carbon"):
            setattr(d, name, _abi.fptr(getattr(self, name), dt))
        return d


def build_minor_index(bnd_lims_gpt: np.ndarray, minor_limits_gpt: np.ndarray):
    """Derive `bnd_st`, `gpt_st` and the kminor re-ordering from the NetCDF-form
    `minor_limits_gpt_{lower,upper}`, as ext/lookup_constructors.jl:220-311 does.

    `bnd_lims_gpt` is (2, n_bnd) and `minor_limits_gpt` (2, n_min_absrb), both 1-based
    inclusive.  Returns (bnd_st, gpt_st, reorder) with `reorder` 1-based indices
    into the file-order contributor axis.
    """
    n_bnd = bnd_lims_gpt.shape[1]
    n_gpt = int(bnd_lims_gpt[1, -1])
    gpt2bnd = np.zeros(n_gpt, dtype=np.int64)
    for ib in range(n_bnd):
        gpt2bnd[bnd_lims_gpt[0, ib] - 1:bnd_lims_gpt[1, ib]] = ib + 1
    n_min = minor_limits_gpt.shape[1]
    minor_bnd = np.zeros(n_min, dtype=np.int64)
    gpt_sh = np.zeros(n_min, dtype=np.int64)
    for i in range(n_min):
        minor_bnd[i] = gpt2bnd[minor_limits_gpt[0, i] - 1]
        if i > 0:
            gpt_sh[i] = gpt_sh[i - 1] + minor_limits_gpt[1, i - 1] - minor_limits_gpt[0, i - 1] + 1
    bnd_st = np.zeros(n_bnd + 1, dtype=np.int64)
    bnd_st[0] = 1
    for ibnd in range(2, n_bnd + 2):
        locs = np.nonzero(minor_bnd == ibnd - 1)[0]
        bnd_st[ibnd - 1] = bnd_st[ibnd - 2] if locs.size == 0 else locs[-1] + 2
    gpt_st = np.ones(n_gpt + 1, dtype=np.int64)
    reorder = []
    for ibnd in range(1, n_bnd + 1):
        nminor = bnd_st[ibnd] - bnd_st[ibnd - 1]
        for loc_in_bnd, igpt in enumerate(range(bnd_lims_gpt[0, ibnd - 1], bnd_lims_gpt[1, ibnd - 1] + 1), start=1):
            gpt_st[igpt] = gpt_st[igpt - 1] + nminor
            for i in range(bnd_st[ibnd - 1], bnd_st[ibnd]):
                reorder.append(gpt_sh[i - 1] + loc_in_bnd)
    return bnd_st, gpt_st, np.asarray(reorder, dtype=np.int64)
