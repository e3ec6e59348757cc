"""The dependency-free HDF5 / NetCDF-4 reader (rrtmgp_jl_amd/hdf5_lite.py) against files written by the REAL HDF5
library with netCDF-C's creation properties (tools/nc4_fixture_writer.py):

  * committed fixtures tests/golden/nc4_features_{v0,v2,v3,oldstyle}.nc — superblock 0 / 2 / 3, object headers v1 / v2,
    old-style groups (symbol table + local heap: HDF5's defaults, e.g. h5py) and dense
    link storage (29 objects: fractal heap + v2 B-tree), dense attribute storage (12 attributes on one variable),
    chunked + shuffle + deflate (+ fletcher32) with ragged edge chunks, layout v3 (v1 B-tree index) and v4 (fixed
    array), contiguous, compact and scalar datasets, fixed-length strings, dimension scales with the reference-typed
    `DIMENSION_LIST` / `REFERENCE_LIST` attributes — every variable must come back bit for bit;
  * when libhdf5 is available (it is in the build image), the schema-faithful rrtmgp-data files of tests/nc_fixture.py
    are converted to NetCDF-4 form and must give exactly the lookups the classic files give, through
    `netcdf_io` with no netCDF4 / h5py installed: this is the path `tools/run_reference_parity.py <rrtmgp-data>` takes.
"""
import dataclasses
import os
import sys

import numpy as np
import pytest

from rrtmgp_jl_amd import hdf5_lite, netcdf_io, synthetic
from rrtmgp_jl_amd.lookups import LookUpMinor

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import nc4_fixture_writer as W  # noqa: E402
import nc_fixture  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")
EXPECTED = np.load(os.path.join(GOLD, "nc4_features_expected.npz"))


@pytest.mark.parametrize("tag,superblock", [("v0", 0), ("v2", 2), ("v3", 3), ("oldstyle", 0)])
def test_committed_fixtures_read_back_bit_for_bit(tag, superblock):
    f = hdf5_lite.File(os.path.join(GOLD, f"nc4_features_{tag}.nc"))
    assert f.buf[8] == superblock
    assert len(f.keys()) == 29                       # 14 variables + 15 dimension-only scales: dense link storage
    for name in EXPECTED.files:
        assert name in f
        a, e = f[name][()], EXPECTED[name]
        assert a.shape == e.shape and a.dtype == e.dtype, name
        assert np.array_equal(a, e), name
        assert f[name].shape == e.shape
    # netCDF-C tracks creation order -> new-style groups (link info 0x02); plain HDF5 defaults -> symbol table 0x11
    assert (0x11 in [m for m, _ in f._root_msgs]) == (tag == "oldstyle")
    assert f["kmajor"]._layout[0] == (4 if tag == "v3" else 3) and f["kmajor"]._filters == [(2, (8,)), (1, (4,))]
    assert f["kminor_lower"]._filters[-1][0] == 3     # fletcher32
    # attributes: header-resident, dense (> 8 on one object), strings, numbers; reference-typed ones are skipped
    assert f["kmajor"].attrs["units"] == b"cm2 mol-1" and f["kmajor"].attrs["scale"] == 1.5
    many = f["many_attrs"].attrs
    assert [many[f"a{i}"] for i in range(12)] == list(map(float, range(12)))
    assert many["DIMENSION_LIST"] is None
    assert f["bnd"].attrs["CLASS"] == b"DIMENSION_SCALE" and f["bnd"].attrs["_Netcdf4Dimid"] == 5
    assert f["temperature"].attrs["NAME"] == b"temperature"          # coordinate variable
    assert f.attrs["_NCProperties"].startswith(b"version=2,netcdf=")
    with pytest.raises(KeyError):
        f["nope"]


def test_netcdf_io_accessors_on_hdf5(tmp_path):
    """The NCDatasets-like accessors of netcdf_io.Dataset work on the HDF5 back end: dims, strings, scalars, attrs."""
    with netcdf_io.Dataset(os.path.join(GOLD, "nc4_features_v0.nc")) as ds:
        assert ds._kind == "h5py" and isinstance(ds._h, hdf5_lite.File)
        assert ds.dim("bnd") == 3 and ds.dim("gpt") == 12 and ds.dim("temperature") == 14
        assert ds.has("kmajor") and not ds.has("kmajor_nope")
        assert ds.strings("gas_names") == ["h2o", "co2", "o3", "n2o", "co", "ch4", "o2", "n2"]
        assert ds.scalar("press_ref_trop") == 9948.4316
        assert ds.attr("absorption_coefficient_ref_T", "units") == "K"
        assert ds.jl("kmajor").shape == (12, 9, 3, 14) and ds.jl("kmajor").flags.f_contiguous


def _same(a, b, path=""):
    for f in dataclasses.fields(a):
        x, y = getattr(a, f.name), getattr(b, f.name)
        if isinstance(x, LookUpMinor):
            _same(x, y, path + f.name + ".")
        elif isinstance(x, np.ndarray):
            assert x.shape == y.shape and np.array_equal(x, y), path + f.name
        else:
            assert x == y, path + f.name


needs_libhdf5 = pytest.mark.skipif(not W.available(), reason="libhdf5 not found (fixtures are committed; conversion needs it)")


@needs_libhdf5
@pytest.mark.parametrize("libver", [("earliest", "v18"), ("latest", "latest")])
def test_rrtmgp_data_schema_files_give_identical_lookups_from_classic_and_netcdf4(tmp_path, libver):
    h5 = W.H5()
    lw = synthetic.make_gas_lookup("lw", n_bnd=5, gpt_per_bnd=[16, 8, 16, 4, 16], seed=7)
    sw = synthetic.make_gas_lookup("sw", n_bnd=4, gpt_per_bnd=[16, 16, 8, 16], seed=8)
    files = {}
    for kind, lk in (("lw", lw), ("sw", sw)):
        p3 = str(tmp_path / f"gas-{kind}.nc")
        nc_fixture.write_gas_file(p3, lk)
        files[kind] = p3
    p3 = str(tmp_path / "cld.nc")
    cl = synthetic.make_cloud_lookup("lw", 5)
    nc_fixture.write_cloud_file(p3, cl, lw.bnd_lims_wn)
    files["cld"] = p3
    p3 = str(tmp_path / "aer.nc")
    ae = synthetic.make_aerosol_lookup("lw", lw.bnd_lims_wn)
    nc_fixture.write_aerosol_file(p3, ae, lw.bnd_lims_wn)
    files["aer"] = p3
    readers = {"lw": lambda d: netcdf_io.lookup_lw(d)[0], "sw": lambda d: netcdf_io.lookup_sw(d)[0],
               "cld": netcdf_io.lookup_cld, "aer": lambda d: netcdf_io.lookup_aerosol(d)}
    for kind, p3 in files.items():
        p4 = p3.replace(".nc", ".nc4")
        W.convert_classic(p3, p4, libver=libver, h5=h5)
        assert open(p4, "rb").read(8) == hdf5_lite.SIGNATURE
        with netcdf_io.Dataset(p3) as d3, netcdf_io.Dataset(p4) as d4:
            assert d3._kind == "scipy" and d4._kind == "h5py"
            a, b = readers[kind](d3), readers[kind](d4)
            if isinstance(a, tuple):
                for x, y in zip(a, b):
                    _same(x, y) if dataclasses.is_dataclass(x) else None
            else:
                _same(a, b)


@needs_libhdf5
def test_convert_rrtmgp_data_directory_of_netcdf4_files(tmp_path):
    """tools/convert_rrtmgp_data.py / run_reference_parity.py entry: a directory of NetCDF-4 files -> the flat container."""
    h5 = W.H5()
    d = tmp_path / "rrtmgp-data"
    d.mkdir()
    lw = synthetic.make_gas_lookup("lw", n_bnd=3, gpt_per_bnd=[8, 4, 12], seed=7)
    sw = synthetic.make_gas_lookup("sw", n_bnd=3, gpt_per_bnd=[6, 10, 4], seed=7)
    for name, writer, args in (("rrtmgp-gas-lw-g256.nc", nc_fixture.write_gas_file, (lw,)),
                               ("rrtmgp-gas-sw-g224.nc", nc_fixture.write_gas_file, (sw,))):
        tmp = str(tmp_path / ("c_" + name))
        writer(tmp, *args)
        W.convert_classic(tmp, str(d / name), h5=h5)
    out = str(tmp_path / "lookups.npz")
    netcdf_io.convert_rrtmgp_data(str(d), out)
    got = netcdf_io.load_lookups(out)
    assert np.array_equal(got["lw"].kmajor, lw.kmajor) and np.array_equal(got["sw"].kmajor, sw.kmajor)


@pytest.mark.skipif(not W.available(), reason="libhdf5 not available")
@pytest.mark.parametrize("libver", [("earliest", "v18"), ("v18", "latest")])
def test_unwritten_elements_read_as_the_fill_value(tmp_path, libver):
    """Partially written and never-written variables: netCDF4 / h5py return the dataset's fill value for what was not
    written (netCDF-C stores `_FillValue`, or the type's default fill, in the HDF5 fill-value message); the reader used to
    return 0.  Written with the real libhdf5: a chunked variable with only its first chunk written, a
    contiguous variable that was never written, and one without a defined fill value (reads 0)."""
    import ctypes as C
    h5 = W.H5()
    L = h5.L
    hs = W.hsize_t
    L.H5Pset_fill_value.restype, L.H5Pset_fill_value.argtypes = C.c_int, [W.hid_t, W.hid_t, C.c_void_p]
    L.H5Sselect_hyperslab.restype = C.c_int
    L.H5Sselect_hyperslab.argtypes = [W.hid_t, C.c_int, C.POINTER(hs), C.POINTER(hs), C.POINTER(hs), C.POINTER(hs)]
    path = str(tmp_path / "fill.nc")
    w = W.NC4Writer(path, libver=libver, h5=h5)
    f4 = h5.types[np.dtype("f4")]
    fill = np.array([9.96921e36], dtype="f4")                     # NC_FILL_FLOAT

    def create(name, shape, chunks, with_fill):
        dcpl = w._dcpl(shape, chunks, 0, False, False)
        if with_fill:
            h5.ok(L.H5Pset_fill_value(dcpl, f4, fill.ctypes.data_as(C.c_void_p)), "H5Pset_fill_value")
        d = (hs * len(shape))(*shape)
        s = L.H5Screate_simple(len(shape), d, None)
        ds = h5.ok(L.H5Dcreate2(w.fid, name.encode(), f4, s, 0, dcpl, 0), name)
        L.H5Pclose(dcpl)
        return ds, s

    ds, fs = create("partial", (6, 8), (3, 4), True)
    block = np.arange(12, dtype="f4").reshape(3, 4) + 1
    start, count = (hs * 2)(0, 0), (hs * 2)(3, 4)
    h5.ok(L.H5Sselect_hyperslab(fs, 0, start, None, count, None), "hyperslab")
    ms = L.H5Screate_simple(2, (hs * 2)(3, 4), None)
    h5.ok(L.H5Dwrite(ds, f4, ms, fs, 0, block.ctypes.data_as(C.c_void_p)), "partial write")
    for h in (ms, fs):
        L.H5Sclose(h)
    L.H5Dclose(ds)
    for name, shape, chunks, with_fill in (("never", (5,), None, True), ("never_chunked", (4, 4), (2, 2), True),
                                           ("no_fill", (3,), None, False)):
        ds, fs = create(name, shape, chunks, with_fill)
        L.H5Sclose(fs)
        L.H5Dclose(ds)
    L.H5Fclose(w.fid)

    f = hdf5_lite.File(path)
    got = f["partial"][()]
    want = np.full((6, 8), fill[0], dtype="f4")
    want[:3, :4] = block
    np.testing.assert_array_equal(got, want)
    np.testing.assert_array_equal(f["never"][()], np.full(5, fill[0], dtype="f4"))
    np.testing.assert_array_equal(f["never_chunked"][()], np.full((4, 4), fill[0], dtype="f4"))
    np.testing.assert_array_equal(f["no_fill"][()], np.zeros(3, dtype="f4"))


def test_truncated_chunk_is_reported(tmp_path):
    """A chunk that decodes to the wrong number of bytes must raise HDF5Error, not an opaque numpy error."""
    src = os.path.join(GOLD, "nc4_features_v0.nc")
    f = hdf5_lite.File(src)
    name = next(n for n in EXPECTED.files if f[n]._filters and f[n].shape)
    ds = f[name]
    orig = ds._decode_chunk
    ds._decode_chunk = lambda raw, mask, dt, nbytes: orig(raw, mask, dt, nbytes)[:-3]
    with pytest.raises(hdf5_lite.HDF5Error, match="truncated or corrupt"):
        ds[()]
