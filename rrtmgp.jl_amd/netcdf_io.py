"""rrtmgp-data ingestion: NetCDF lookup files -> the flat host containers of lookups.py,
and a flat `.npz` container so that the run-time path needs no NetCDF library.

Mirrors the constructors of the reference's `ext/lookup_constructors.jl`
(`LookUpLW` :83-405, `LookUpSW` :407-725, `LookUpCld` :727-751, `LookUpAerosolMerra`
:18-81).  NCDatasets presents a NetCDF variable declared `v(d1, d2, d3)` as a Julia
array of shape (d3, d2, d1); `Dataset.jl()` returns exactly that (the transposed view,
Fortran-ordered), so every `permutedims` below carries the reference's axis numbers.

Back ends: NetCDF-3 classic / 64-bit offset files go to `scipy.io.netcdf_file`; NetCDF-4 files
(HDF5, what rrtmgp-data v1.9 ships) go to `netCDF4` or `h5py` when one is importable and otherwise
to the built-in `hdf5_lite` reader (numpy + zlib only), so ingestion needs nothing beyond the data.
The tests write classic files in the rrtmgp-data schema and NetCDF-4 files produced by the real HDF5
library (tools/nc4_fixture_writer.py).
"""
from __future__ import annotations

import os
from typing import Dict, Tuple

import numpy as np

from .lookups import GasLookup, LookUpAerosolMerra, LookUpCld, LookUpMinor, build_minor_index


class Dataset:
    """Read-only view of one NetCDF file with NCDatasets-like accessors."""

    def __init__(self, path: str):
        self.path = path
        self._kind, self._h = _open(path)

    def close(self):
        if self._kind in ("netcdf4", "scipy"):
            self._h.close()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def dim(self, name: str) -> int:
        if self._kind == "scipy":
            return int(self._h.dimensions[name])
        if self._kind == "netcdf4":
            return int(len(self._h.dimensions[name]))
        return int(self._h[name].shape[0])

    def has(self, name: str) -> bool:
        return name in (self._h.variables if self._kind in ("scipy", "netcdf4") else self._h)

    def raw(self, name: str) -> np.ndarray:
        """The variable in file (C) order, masked values filled, as a fresh array."""
        if self._kind in ("scipy", "netcdf4"):
            v = self._h.variables[name]
            a = v[...] if v.shape else v.getValue() if self._kind == "scipy" else v[...]
        else:
            a = self._h[name][()]
        if np.ma.isMaskedArray(a):
            a = a.filled()
        return np.array(a)

    def jl(self, name: str) -> np.ndarray:
        """The variable with NCDatasets' (reversed) axis order."""
        return np.asfortranarray(self.raw(name).T)

    def scalar(self, name: str) -> float:
        return float(np.asarray(self.raw(name)).reshape(-1)[0])

    def attr(self, var: str, name: str):
        if self._kind in ("scipy", "netcdf4"):
            a = getattr(self._h.variables[var], name)
        else:
            a = self._h[var].attrs[name]
        return a.decode() if isinstance(a, bytes) else a

    def strings(self, name: str):
        """Char array (n, string_len) -> list of stripped str."""
        a = self.raw(name)
        out = []
        for row in a.reshape(a.shape[0], -1):
            s = b"".join(bytes(c) if isinstance(c, (bytes, np.bytes_)) else bytes([int(c)]) for c in row)
            out.append(s.decode("ascii", "ignore").replace("\x00", " ").strip())
        return out


def _open(path):
    if not os.path.exists(path):
        raise FileNotFoundError(path)
    with open(path, "rb") as f:
        magic = f.read(8)
    if magic[:3] == b"CDF":
        from scipy.io import netcdf_file
        return "scipy", netcdf_file(path, "r", mmap=False)
    if magic == b"\x89HDF\r\n\x1a\n":
        try:
            import netCDF4
            return "netcdf4", netCDF4.Dataset(path, "r")
        except ImportError:
            pass
        try:
            import h5py
            return "h5py", h5py.File(path, "r")
        except ImportError:
            pass
        # dependency-free reader (numpy + zlib) for the HDF5 subset NetCDF-4 files use; same mini-interface as h5py
        from . import hdf5_lite
        return "h5py", hdf5_lite.File(path)
    raise RuntimeError(f"{path}: not a NetCDF file")


# ---- gas optics ------------------------------------------------------------------
def _gas_common(ds: Dataset, FT):
    """The part shared by LookUpLW and LookUpSW (lookup_constructors.jl:97-361 / :421-651)."""
    n_bnd, n_gpt = ds.dim("bnd"), ds.dim("gpt")
    gases_major = ds.strings("gas_names")
    idx_gases: Dict[str, int] = {g: i + 1 for i, g in enumerate(gases_major)}
    idx_h2o = idx_gases["h2o"]
    idx_gases["h2o_frgn"] = idx_h2o
    idx_gases["h2o_self"] = idx_h2o
    # VmrGM hard-wires h2o = 1 and o3 = 3 (lookup_constructors.jl:9-16)
    if idx_gases["h2o"] != 1 or idx_gases.get("o3") != 3:
        raise ValueError("lookup file does not keep h2o / o3 in gas slots 1 / 3")

    def minor_idx(gname, sname):
        g = ds.strings(gname)
        s = ds.strings(sname)
        ig = np.array([idx_gases[x] if x else 0 for x in g], dtype=np.int64)
        isc = np.array([idx_gases[x] if x else 0 for x in s], dtype=np.int64)
        return ig, isc

    key_species = ds.jl("key_species").astype(np.int64)       # (2, 2, n_bnd)
    zero = (key_species[0] == 0) & (key_species[1] == 0)       # :175-182
    key_species[0][zero] = 2
    key_species[1][zero] = 2

    kmajor = np.transpose(ds.jl("kmajor"), (1, 2, 3, 0))       # :186
    bnd_lims_gpt = ds.jl("bnd_limits_gpt").astype(np.int64)    # (2, n_bnd)
    bnd_lims_wn = ds.jl("bnd_limits_wavenumber")
    gpt2bnd = np.zeros(n_gpt, dtype=np.int64)
    for ib in range(n_bnd):
        gpt2bnd[bnd_lims_gpt[0, ib] - 1:bnd_lims_gpt[1, ib]] = ib + 1

    minors = []
    for reg in ("lower", "upper"):
        lims = ds.jl(f"minor_limits_gpt_{reg}").astype(np.int64)
        bnd_st, gpt_st, reorder = build_minor_index(bnd_lims_gpt, lims)
        kminor = np.transpose(ds.jl(f"kminor_{reg}"), (1, 2, 0))[:, :, reorder - 1]   # :298-311
        ig, isc = minor_idx(f"minor_gases_{reg}", f"scaling_gas_{reg}")
        gasdata = np.stack([ig, isc,
                            ds.raw(f"minor_scales_with_density_{reg}").astype(np.int64).reshape(-1),
                            ds.raw(f"scale_by_complement_{reg}").astype(np.int64).reshape(-1)])
        minors.append(LookUpMinor(bnd_st, gpt_st, np.asfortranarray(gasdata),
                                  np.asfortranarray(kminor, dtype=FT)))

    p_ref = ds.raw("press_ref").astype(np.float64)
    t_ref = ds.raw("temp_ref").astype(np.float64)
    common = dict(
        idx_h2o=idx_h2o,
        p_ref_tropo=float(FT(ds.scalar("press_ref_trop"))),
        p_ref_min=float(FT(p_ref.min())),
        t_ref_min=float(FT(t_ref.min())), t_ref_max=float(FT(t_ref.max())),
        key_species=np.asfortranarray(key_species),
        kmajor=np.asfortranarray(kmajor, dtype=FT),
        major_gpt2bnd=gpt2bnd,
        bnd_lims_wn=np.asfortranarray(bnd_lims_wn, dtype=FT),
        ln_p_ref=np.log(p_ref.astype(FT)).astype(FT),           # :346 (log in working precision)
        t_ref=t_ref.astype(FT),
        vmr_ref=np.asfortranarray(ds.jl("vmr_ref"), dtype=FT),
        minor_lower=minors[0], minor_upper=minors[1])
    return common, idx_gases


def lookup_lw(ds: Dataset, FT=np.float64) -> Tuple[GasLookup, Dict[str, int]]:
    """LookUpLW(ds, FT, DA) — lookup_constructors.jl:83-405."""
    FT = np.dtype(FT).type
    common, idx_gases = _gas_common(ds, FT)
    t_planck = ds.raw("temperature_Planck").astype(np.float64)
    if not (100 <= t_planck[0] and t_planck[-1] <= 500):        # :196-200
        raise ValueError(f"`temperature_Planck` does not look like Kelvin ({t_planck[0]}…{t_planck[-1]}); "
                         "this file is not usable with the Planck interpolation")
    planck_fraction = np.transpose(ds.jl("plank_fraction"), (1, 2, 3, 0))   # [sic], :189
    lk = GasLookup(is_sw=False, **common,
                   planck_fraction=np.asfortranarray(planck_fraction, dtype=FT),
                   t_planck=t_planck.astype(FT),
                   tot_planck=np.asfortranarray(ds.jl("totplnk"), dtype=FT))
    return lk, idx_gases


def lookup_sw(ds: Dataset, FT=np.float64) -> Tuple[GasLookup, Dict[str, int]]:
    """LookUpSW(ds, FT, DA) — lookup_constructors.jl:407-725."""
    FT = np.dtype(FT).type
    common, idx_gases = _gas_common(ds, FT)
    a_offset, b_offset = FT(0.1495954), FT(0.00066696)          # :656-665
    mg = FT(max(ds.scalar("mg_default"), 0))
    sb = FT(max(ds.scalar("sb_default"), 0))
    solar_src = (ds.raw("solar_source_quiet") + (mg - a_offset) * ds.raw("solar_source_facular")
                 + (sb - b_offset) * ds.raw("solar_source_sunspot"))
    solar_src_tot = FT(solar_src.sum())
    lk = GasLookup(is_sw=True, **common,
                   solar_src_tot=float(solar_src_tot),
                   rayl_lower=np.asfortranarray(np.transpose(ds.jl("rayl_lower"), (1, 2, 0)), dtype=FT),
                   rayl_upper=np.asfortranarray(np.transpose(ds.jl("rayl_upper"), (1, 2, 0)), dtype=FT),
                   solar_src_scaled=(solar_src / solar_src_tot).astype(FT))
    return lk, idx_gases


# ---- clouds / aerosols ---------------------------------------------------------------
def lookup_cld(ds: Dataset, FT=np.float64) -> LookUpCld:
    """LookUpCld(ds, FT, DA) — lookup_constructors.jl:727-751 (ice diameters halved to radii)."""
    FT = np.dtype(FT).type
    dims = np.array([ds.dim("nband"), ds.dim("nrghice"), ds.dim("nsize_liq"), ds.dim("nsize_ice"),
                     ds.dim("pair")], dtype=np.int64)
    bounds = np.array([ds.scalar("radliq_lwr"), ds.scalar("radliq_upr"),
                       ds.scalar("diamice_lwr") / 2, ds.scalar("diamice_upr") / 2], dtype=FT)
    liq = np.concatenate([ds.jl("extliq"), ds.jl("ssaliq"), ds.jl("asyliq")], axis=0)
    ice = np.concatenate([ds.jl("extice"), ds.jl("ssaice"), ds.jl("asyice")], axis=0)
    return LookUpCld(dims, bounds, np.asfortranarray(liq, dtype=FT), np.asfortranarray(ice, dtype=FT))


AEROSOL_INDEX = {"dust1": 1, "sea_salt1": 2, "sulfate": 3, "black_carbon_rh": 4, "black_carbon": 5,
                 "organic_carbon_rh": 6, "organic_carbon": 7,
                 **{f"dust{i}": i + 6 for i in range(2, 6)}, **{f"sea_salt{i}": i + 10 for i in range(2, 6)}}
AEROSIZE_INDEX = {v: v for k, v in AEROSOL_INDEX.items() if "dust" in k or "sea_salt" in k}


def lookup_aerosol(ds: Dataset, FT=np.float64):
    """LookUpAerosolMerra(ds, FT, DA) — lookup_constructors.jl:18-81.
    Returns (lookup, idx_aerosol, idx_aerosize)."""
    FT = np.dtype(FT).type
    wn = ds.jl("bnd_limits_wavenumber")
    i550 = 0
    for i in range(wn.shape[1]):                                 # first band holding 550 nm, :41-45
        if 1.0 / (wn[1, i] * 100.0) <= 550e-9 <= 1.0 / (wn[0, i] * 100.0):
            i550 = i + 1
            break

    def t(name):
        return np.asfortranarray(ds.jl(name), dtype=FT)
    lk = LookUpAerosolMerra(t("merra_aero_bin_lims"), ds.raw("aero_rh").astype(FT), t("aero_dust_tbl"),
                            t("aero_salt_tbl"), t("aero_sulf_tbl"), t("aero_bcar_rh_tbl"), t("aero_bcar_tbl"),
                            t("aero_ocar_rh_tbl"), t("aero_ocar_tbl"), i550)
    return lk, dict(AEROSOL_INDEX), dict(AEROSIZE_INDEX)


# ---- flat container ------------------------------------------------------------------
def _flatten(prefix, obj, out):
    from dataclasses import fields, is_dataclass
    for f in fields(obj):
        v = getattr(obj, f.name)
        key = f"{prefix}{f.name}"
        if v is None:
            continue
        if is_dataclass(v):
            _flatten(key + ".", v, out)
        else:
            out[key] = np.asarray(v)


def save_lookups(path: str, **lookups):
    """Write named lookups (`lw=…, sw=…, lw_cld=…, sw_aero=…`) plus optional `idx_gases`
    into one `.npz`.  Arrays keep their Julia shape; order is restored on load."""
    out = {}
    for name, lk in lookups.items():
        if lk is None:
            continue
        if isinstance(lk, dict):
            out[f"{name}#keys"] = np.array(sorted(lk), dtype="U32")
            out[f"{name}#vals"] = np.array([lk[k] for k in sorted(lk)], dtype=np.int64)
            continue
        out[f"{name}#type"] = np.array(type(lk).__name__)
        _flatten(f"{name}/", lk, out)
    np.savez_compressed(path, **out)


def load_lookups(path: str) -> dict:
    """Inverse of `save_lookups`."""
    z = np.load(path, allow_pickle=False)
    names = sorted({k.split("#")[0] for k in z.files if "#" in k})
    res = {}
    for name in names:
        if f"{name}#keys" in z.files:
            res[name] = {str(k): int(v) for k, v in zip(z[f"{name}#keys"], z[f"{name}#vals"])}
            continue
        typ = str(z[f"{name}#type"])
        items = {k[len(name) + 1:]: z[k] for k in z.files if k.startswith(name + "/")}

        def val(a):
            if a.ndim == 0:
                return a.item()
            return np.asfortranarray(a)
        top = {k: val(v) for k, v in items.items() if "." not in k}
        if typ == "GasLookup":
            for reg in ("minor_lower", "minor_upper"):
                top[reg] = LookUpMinor(**{k.split(".", 1)[1]: val(v) for k, v in items.items()
                                          if k.startswith(reg + ".")})
            top["is_sw"] = bool(top["is_sw"])
            res[name] = GasLookup(**top)
        elif typ == "LookUpCld":
            res[name] = LookUpCld(**top)
        elif typ == "LookUpAerosolMerra":
            res[name] = LookUpAerosolMerra(**top)
        else:
            raise ValueError(f"{path}: unknown lookup type {typ}")
    return res


RRTMGP_DATA_FILES = {  # artifact layout of rrtmgp-data v1.9 (src/ArtifactPaths.jl:28-46)
    "lw": "rrtmgp-gas-lw-g256.nc", "sw": "rrtmgp-gas-sw-g224.nc",
    "lw_cld": "rrtmgp-clouds-lw-bnd.nc", "sw_cld": "rrtmgp-clouds-sw-bnd.nc",
    "lw_aero": "rrtmgp-aerosols-merra-lw.nc", "sw_aero": "rrtmgp-aerosols-merra-sw.nc",
}


def convert_rrtmgp_data(data_dir: str, out_path: str, FT=np.float64, files=None):
    """Read whichever of the six lookup files exist under `data_dir` and write one flat
    container.  Returns the dict that `load_lookups(out_path)` would give."""
    files = dict(RRTMGP_DATA_FILES, **(files or {}))
    got = {}
    for key, fname in files.items():
        p = os.path.join(data_dir, fname)
        if not os.path.exists(p):
            continue
        with Dataset(p) as ds:
            if key in ("lw", "sw"):
                got[key], idx = (lookup_lw if key == "lw" else lookup_sw)(ds, FT)
                got["idx_gases"] = idx
            elif key.endswith("_cld"):
                got[key] = lookup_cld(ds, FT)
            else:
                got[key], got["idx_aerosol"], got["idx_aerosize"] = lookup_aerosol(ds, FT)
    if not got:
        raise FileNotFoundError(f"no rrtmgp-data lookup file found under {data_dir}")
    save_lookups(out_path, **got)
    return got
