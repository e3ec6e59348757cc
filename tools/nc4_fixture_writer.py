#!/usr/bin/env python3
"""Writes NetCDF-4-style HDF5 files with the REAL HDF5 library (ctypes over libhdf5 / libhdf5_hl,
present in the build image under /opt/conda/lib), using the creation properties netCDF-C uses
(libhdf5/hdf5create.c, nc4hdf.c): link and attribute creation order tracked + indexed, hence new-style
groups whose links live in a fractal heap + v2 B-tree once there are more than 8 variables; one
dimension-scale dataset per dimension (H5DSset_scale / H5DSattach_scale: `CLASS`, `NAME`,
`DIMENSION_LIST`, `REFERENCE_LIST`, `_Netcdf4Dimid` attributes); chunked + shuffle + deflate variables;
fixed-length character arrays for strings; `_NCProperties` on the root group.

Used ONLY to produce test fixtures for rrtmgp_jl_amd/hdf5_lite.py (the dependency-free reader):

    python tools/nc4_fixture_writer.py tests/golden        # writes tests/golden/nc4_features_*.nc

and by tests/test_hdf5_lite.py to convert the schema-faithful classic files of tests/nc_fixture.py when
the library is available.  Nothing in the product imports this file.
"""
from __future__ import annotations

import ctypes as C
import ctypes.util
import os
import sys

import numpy as np

hid_t = C.c_int64
hsize_t = C.c_uint64


def _find(name):
    for d in ("/opt/conda/lib", "/usr/lib/x86_64-linux-gnu", "/usr/lib", "/usr/local/lib"):
        for cand in (f"lib{name}.so", f"lib{name}_serial.so"):
            p = os.path.join(d, cand)
            if os.path.exists(p):
                return p
    return ctypes.util.find_library(name)


def available() -> bool:
    return bool(_find("hdf5")) and bool(_find("hdf5_hl"))


class H5:
    def __init__(self):
        p, ph = _find("hdf5"), _find("hdf5_hl")
        if not p or not ph:
            raise RuntimeError("libhdf5 / libhdf5_hl not found")
        self.L = C.CDLL(p, mode=C.RTLD_GLOBAL)
        self.HL = C.CDLL(ph, mode=C.RTLD_GLOBAL)
        L = self.L
        L.H5open()
        for fn, res, args in [
            ("H5Pcreate", hid_t, [hid_t]), ("H5Pclose", C.c_int, [hid_t]),
            ("H5Pset_libver_bounds", C.c_int, [hid_t, C.c_int, C.c_int]),
            ("H5Pset_link_creation_order", C.c_int, [hid_t, C.c_uint]),
            ("H5Pset_attr_creation_order", C.c_int, [hid_t, C.c_uint]),
            ("H5Pset_chunk", C.c_int, [hid_t, C.c_int, C.POINTER(hsize_t)]),
            ("H5Pset_deflate", C.c_int, [hid_t, C.c_uint]), ("H5Pset_shuffle", C.c_int, [hid_t]),
            ("H5Pset_fletcher32", C.c_int, [hid_t]), ("H5Pset_layout", C.c_int, [hid_t, C.c_int]),
            ("H5Fcreate", hid_t, [C.c_char_p, C.c_uint, hid_t, hid_t]), ("H5Fclose", C.c_int, [hid_t]),
            ("H5Screate_simple", hid_t, [C.c_int, C.POINTER(hsize_t), C.POINTER(hsize_t)]),
            ("H5Screate", hid_t, [C.c_int]), ("H5Sclose", C.c_int, [hid_t]),
            ("H5Tcopy", hid_t, [hid_t]), ("H5Tset_size", C.c_int, [hid_t, C.c_size_t]), ("H5Tclose", C.c_int, [hid_t]),
            ("H5Tset_strpad", C.c_int, [hid_t, C.c_int]),
            ("H5Dcreate2", hid_t, [hid_t, C.c_char_p, hid_t, hid_t, hid_t, hid_t, hid_t]),
            ("H5Dwrite", C.c_int, [hid_t, hid_t, hid_t, hid_t, hid_t, C.c_void_p]), ("H5Dclose", C.c_int, [hid_t]),
            ("H5Acreate2", hid_t, [hid_t, C.c_char_p, hid_t, hid_t, hid_t, hid_t]),
            ("H5Awrite", C.c_int, [hid_t, hid_t, C.c_void_p]), ("H5Aclose", C.c_int, [hid_t]),
            ("H5Gopen2", hid_t, [hid_t, C.c_char_p, hid_t]), ("H5Gclose", C.c_int, [hid_t]),
        ]:
            f = getattr(L, fn)
            f.restype, f.argtypes = res, args
        for fn, res, args in [("H5DSset_scale", C.c_int, [hid_t, C.c_char_p]),
                              ("H5DSattach_scale", C.c_int, [hid_t, hid_t, C.c_uint])]:
            f = getattr(self.HL, fn)
            f.restype, f.argtypes = res, args

        def g(name):
            return hid_t.in_dll(L, name).value
        self.P_FILE_CREATE, self.P_FILE_ACCESS = g("H5P_CLS_FILE_CREATE_ID_g"), g("H5P_CLS_FILE_ACCESS_ID_g")
        self.P_DATASET_CREATE = g("H5P_CLS_DATASET_CREATE_ID_g")
        self.types = {np.dtype("f8"): g("H5T_NATIVE_DOUBLE_g"), np.dtype("f4"): g("H5T_NATIVE_FLOAT_g"),
                      np.dtype("i4"): g("H5T_NATIVE_INT32_g"), np.dtype("i8"): g("H5T_NATIVE_INT64_g"),
                      np.dtype("i2"): g("H5T_NATIVE_INT16_g"), np.dtype("u1"): g("H5T_NATIVE_UINT8_g"),
                      np.dtype("i1"): g("H5T_NATIVE_INT8_g")}
        self.C_S1 = g("H5T_C_S1_g")

    def ok(self, rc, what):
        if rc < 0:
            raise RuntimeError(f"HDF5 call failed: {what}")
        return rc


CRT = 0x0001 | 0x0002        # H5P_CRT_ORDER_TRACKED | H5P_CRT_ORDER_INDEXED
NOT_A_VAR = b"This is a netCDF dimension but not a netCDF variable."


class NC4Writer:
    """createDimension / createVariable in the style of netCDF-C's HDF5 layer."""

    def __init__(self, path, libver=("earliest", "v18"), h5=None, track_order=True):
        """`track_order=False` leaves HDF5's defaults (what h5py and plain HDF5 writers produce): old-style groups
        (symbol table: v1 B-tree + local heap) and no creation-order indexes."""
        self.h5 = h5 or H5()
        L = self.h5.L
        bounds = {"earliest": 0, "v18": 1, "latest": 2 if not hasattr(L, "H5F_LIBVER_V110") else 2}
        fapl = L.H5Pcreate(self.h5.P_FILE_ACCESS)
        self.h5.ok(L.H5Pset_libver_bounds(fapl, bounds[libver[0]], bounds[libver[1]]), "libver bounds")
        fcpl = L.H5Pcreate(self.h5.P_FILE_CREATE)
        self.track_order = track_order
        if track_order:
            self.h5.ok(L.H5Pset_link_creation_order(fcpl, CRT), "link creation order")
            self.h5.ok(L.H5Pset_attr_creation_order(fcpl, CRT), "attr creation order")
        self.fid = self.h5.ok(L.H5Fcreate(path.encode(), 2, fcpl, fapl), "H5Fcreate")
        L.H5Pclose(fapl); L.H5Pclose(fcpl)
        self.dims = {}       # name -> (size, dataset id or None, dimid)
        self.pending = []    # variables whose scales are attached at close
        self.root = self.fid
        self._attr_str(self.fid, "_NCProperties", b"version=2,netcdf=4.9.2,hdf5=1.10.6")

    # -- attributes ---------------------------------------------------------------------------------
    def _attr_str(self, loc, name, val: bytes):
        L = self.h5.L
        t = L.H5Tcopy(self.h5.C_S1)
        L.H5Tset_size(t, max(len(val), 1))
        s = L.H5Screate(0)
        a = self.h5.ok(L.H5Acreate2(loc, name.encode(), t, s, 0, 0), "H5Acreate2")
        buf = C.create_string_buffer(val, max(len(val), 1))
        L.H5Awrite(a, t, buf)
        L.H5Aclose(a); L.H5Sclose(s); L.H5Tclose(t)

    def _attr_num(self, loc, name, arr):
        L = self.h5.L
        arr = np.require(arr, requirements="C")
        t = self.h5.types[arr.dtype]
        if arr.ndim == 0:
            s = L.H5Screate(0)
        else:
            d = (hsize_t * 1)(arr.size)
            s = L.H5Screate_simple(1, d, None)
        a = self.h5.ok(L.H5Acreate2(loc, name.encode(), t, s, 0, 0), "H5Acreate2")
        L.H5Awrite(a, t, arr.ctypes.data_as(C.c_void_p))
        L.H5Aclose(a); L.H5Sclose(s)

    # -- dimensions / variables -----------------------------------------------------------------------
    def createDimension(self, name, size):
        self.dims[name] = [int(size), None, len(self.dims)]

    def _dcpl(self, shape, chunks, deflate, shuffle, fletcher):
        L = self.h5.L
        dcpl = L.H5Pcreate(self.h5.P_DATASET_CREATE)
        if self.track_order:
            L.H5Pset_attr_creation_order(dcpl, CRT)
        if chunks is not None and len(shape):
            c = (hsize_t * len(shape))(*chunks)
            self.h5.ok(L.H5Pset_chunk(dcpl, len(shape), c), "H5Pset_chunk")
            if shuffle:
                L.H5Pset_shuffle(dcpl)
            if deflate:
                L.H5Pset_deflate(dcpl, deflate)
            if fletcher:
                L.H5Pset_fletcher32(dcpl)
        return dcpl

    def _create(self, name, arr, dcpl, h5type=None):
        L = self.h5.L
        arr = np.require(arr, requirements="C")
        t = h5type if h5type is not None else self.h5.types[arr.dtype]
        if arr.ndim == 0:
            s = L.H5Screate(0)
        else:
            d = (hsize_t * arr.ndim)(*arr.shape)
            s = L.H5Screate_simple(arr.ndim, d, None)
        ds = self.h5.ok(L.H5Dcreate2(self.fid, name.encode(), t, s, 0, dcpl, 0), f"H5Dcreate2 {name}")
        self.h5.ok(L.H5Dwrite(ds, t, 0, 0, 0, arr.ctypes.data_as(C.c_void_p)), f"H5Dwrite {name}")
        L.H5Sclose(s)
        return ds

    def createVariable(self, name, arr, dims, chunks=None, deflate=0, shuffle=False, fletcher=False, attrs=None,
                       compact=False):
        """`arr` in file (C) order; `dims` names its axes.  A variable named like its only dimension becomes the
        coordinate variable (= the dimension scale) of that dimension."""
        L = self.h5.L
        arr = np.require(arr, requirements="C")     # (ascontiguousarray would turn a scalar into a 1-d array)
        assert arr.ndim == len(dims), (name, arr.shape, dims)
        for d, n in zip(dims, arr.shape):
            if d not in self.dims:
                self.createDimension(d, n)
            assert self.dims[d][0] == n, (name, d, n, self.dims[d][0])
        h5type = None
        if arr.dtype.kind == "S":                      # NC_CHAR: 1-byte fixed strings
            h5type = L.H5Tcopy(self.h5.C_S1)
            L.H5Tset_size(h5type, arr.dtype.itemsize)
        dcpl = self._dcpl(arr.shape, chunks, deflate, shuffle, fletcher)
        if compact:
            L.H5Pset_layout(dcpl, 0)
        ds = self._create(name, arr, dcpl, h5type)
        L.H5Pclose(dcpl)
        for k, v in (attrs or {}).items():
            if isinstance(v, (bytes, str)):
                self._attr_str(ds, k, v if isinstance(v, bytes) else v.encode())
            else:
                self._attr_num(ds, k, np.asarray(v))
        if len(dims) == 1 and dims[0] == name:
            self.dims[name][1] = ds                     # coordinate variable: becomes the scale at close
        else:
            self.pending.append((ds, dims))
        return ds

    def close(self):
        L, HL = self.h5.L, self.h5.HL
        # dimensions without a coordinate variable get a placeholder dataset, as netCDF-C writes them
        for name, rec in self.dims.items():
            if rec[1] is None:
                dcpl = self._dcpl((rec[0],), None, 0, False, False)
                rec[1] = self._create(name, np.zeros(rec[0], dtype="f4"), dcpl)
                L.H5Pclose(dcpl)
                self.h5.ok(HL.H5DSset_scale(rec[1], NOT_A_VAR + b"%10d" % rec[0]), "H5DSset_scale")
            else:
                self.h5.ok(HL.H5DSset_scale(rec[1], name.encode()), "H5DSset_scale")
            self._attr_num(rec[1], "_Netcdf4Dimid", np.int32(rec[2]))
        for ds, dims in self.pending:
            for i, d in enumerate(dims):
                self.h5.ok(HL.H5DSattach_scale(ds, self.dims[d][1], i), "H5DSattach_scale")
            L.H5Dclose(ds)
        for rec in self.dims.values():
            L.H5Dclose(rec[1])
        L.H5Fclose(self.fid)


def convert_classic(src, dst, libver=("earliest", "v18"), deflate=4, h5=None):
    """NetCDF-3 classic file (scipy) -> NetCDF-4-style HDF5 with the same variables, dimensions and attributes.
    Multi-dimensional numeric variables are chunked (one chunk = the last two axes, or ragged 3 x ... chunks for
    the small ones), shuffled and deflated, like an `nccopy -d4 -s`."""
    from scipy.io import netcdf_file
    nc = netcdf_file(src, "r", mmap=False)
    w = NC4Writer(dst, libver, h5)
    for d, n in nc.dimensions.items():
        w.createDimension(d, n)
    for name, v in nc.variables.items():
        a = np.array(v[...] if v.shape else v.getValue())
        if a.dtype.byteorder == ">" or (a.dtype.byteorder == "=" and sys.byteorder == "big"):
            a = a.astype(a.dtype.newbyteorder("<"))
        a = np.require(a, requirements="C")
        attrs = {k: (val if isinstance(val, (bytes, str)) else np.asarray(val)) for k, val in v._attributes.items()}
        chunks = None
        if a.ndim >= 2 and a.dtype.kind in "fi" and a.size >= 64:
            chunks = tuple(1 if i < a.ndim - 2 else max(1, (n + 1) // 2 + 1) for i, n in enumerate(a.shape))
        w.createVariable(name, a, v.dimensions, chunks=chunks, deflate=deflate if chunks else 0, shuffle=bool(chunks),
                         attrs=attrs)
    nc.close()
    w.close()


def write_feature_fixture(path, libver, seed=0, h5=None, track_order=True):
    """One small file carrying every on-disk feature the reader claims (see rrtmgp_jl_amd/hdf5_lite.py); returns the
    arrays written, by variable name."""
    rng = np.random.default_rng(seed)
    w = NC4Writer(path, libver, h5, track_order)
    out = {}

    def put(name, arr, dims, **kw):
        out[name] = np.array(arr)
        w.createVariable(name, arr, dims, **kw)
    put("temperature", np.linspace(160.0, 355.0, 14), ("temperature",), attrs={"units": "K"})           # coordinate variable
    # values on a coarse grid so that shuffle + deflate really compress (the committed files stay small)
    put("kmajor", np.round(rng.random((14, 3, 9, 12)), 2), ("temperature", "pressure", "mixing_fraction", "gpt"),
        chunks=(3, 3, 4, 12), deflate=4, shuffle=True, attrs={"units": "cm2 mol-1", "scale": np.float64(1.5)})
    put("kminor_lower", np.round(rng.random((14, 9, 20)), 1).astype("f4"), ("temperature", "mixing_fraction", "contributors_lower"),
        chunks=(5, 9, 16), deflate=1, shuffle=False, fletcher=True)
    put("bnd_limits_gpt", np.arange(1, 7, dtype="i4").reshape(3, 2), ("bnd", "pair"))                     # contiguous int32
    put("key_species", rng.integers(0, 8, (3, 2, 2)).astype("i4"), ("bnd", "atmos_layer", "pair"), chunks=(2, 2, 2),
        deflate=9, shuffle=True)
    names = np.array([b"h2o", b"co2", b"o3", b"n2o", b"co", b"ch4", b"o2", b"n2"]).astype("S1")
    gas = np.zeros((8, 32), dtype="S1")
    for i, s in enumerate([b"h2o", b"co2", b"o3", b"n2o", b"co", b"ch4", b"o2", b"n2"]):
        gas[i, :len(s)] = np.frombuffer(s, dtype="S1")
        gas[i, len(s):] = b" "
    put("gas_names", gas, ("absorber", "string_len"))
    put("press_ref_trop", np.array(9948.4316), ())                                                         # scalar
    put("absorption_coefficient_ref_T", np.array(296.0), (), attrs={"units": "K"})
    put("tiny", np.arange(6, dtype="i2"), ("six",), compact=True)                                          # compact layout
    put("big_endian_free", rng.integers(-100, 100, (5, 3)).astype("i8"), ("five", "three"))
    put("bytes", np.arange(10, dtype="u1"), ("ten",))
    put("totplnk", np.round(rng.random((3, 196)), 3), ("bnd", "temperature_Planck"), chunks=(1, 50), deflate=4, shuffle=True)
    put("solar_source_quiet", rng.random(12), ("gpt",))
    put("many_attrs", np.arange(4.0), ("four",),
        attrs={f"a{i}": np.float64(i) for i in range(12)})                                                # > 8 attributes: dense storage
    w.close()
    return out


def main(outdir):
    os.makedirs(outdir, exist_ok=True)
    h5 = H5()
    # superblock 0 + layout v3 (what netCDF-C writes by default), superblock 2, and superblock 3 + layout v4
    for tag, lv in (("v0", ("earliest", "v18")), ("v2", ("v18", "latest")), ("v3", ("latest", "latest"))):
        p = os.path.join(outdir, f"nc4_features_{tag}.nc")
        arrs = write_feature_fixture(p, lv, seed=1, h5=h5)
        print(p, os.path.getsize(p), "bytes,", len(arrs), "variables")
    # HDF5 defaults instead of netCDF-C's creation-order tracking: old-style groups (symbol table + local heap)
    p = os.path.join(outdir, "nc4_features_oldstyle.nc")
    write_feature_fixture(p, ("earliest", "v18"), seed=1, h5=h5, track_order=False)
    print(p, os.path.getsize(p), "bytes (old-style groups)")
    np.savez_compressed(os.path.join(outdir, "nc4_features_expected.npz"), **arrs)   # the same arrays in all four


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "tests/golden")
