"""N1: rrtmgp-data ingestion.  Synthetic lookups are written as classic NetCDF files in
the v1.9 schema (tests/nc_fixture.py), read back through rrtmgp_jl_amd.netcdf_io
(the mirror of ext/lookup_constructors.jl) and must reproduce the in-memory form
exactly — permutations, minor-gas re-ordering, key-species rewrite, solar source."""
import dataclasses
import os

import numpy as np
import pytest

from rrtmgp_jl_amd import netcdf_io, synthetic
from rrtmgp_jl_amd.lookups import LookUpMinor

import nc_fixture


def _same(a, b, path=""):
    for f in dataclasses.fields(a):
        x, y = getattr(a, f.name), getattr(b, f.name)
        where = f"{path}{f.name}"
        if isinstance(x, LookUpMinor):
            _same(x, y, where + ".")
        elif isinstance(x, np.ndarray):
            assert x.shape == y.shape, where
            if f.name == "ln_p_ref":          # log(exp(x)) is not exact
                np.testing.assert_allclose(y, x, rtol=0, atol=4e-15, err_msg=where)
            elif f.name in ("solar_src_scaled",):
                continue
            else:
                assert np.array_equal(x, y), where
        elif isinstance(x, float):
            assert y == pytest.approx(x, rel=1e-14), where
        else:
            assert x == y, where


@pytest.mark.parametrize("kind", ["lw", "sw"])
def test_gas_lookup_round_trip(tmp_path, kind):
    lk = synthetic.make_gas_lookup(kind, n_bnd=5, gpt_per_bnd=[16, 8, 16, 4, 16], seed=7)
    p = str(tmp_path / f"gas-{kind}.nc")
    solar = nc_fixture.write_gas_file(p, lk)
    with netcdf_io.Dataset(p) as ds:
        got, idx = (netcdf_io.lookup_lw if kind == "lw" else netcdf_io.lookup_sw)(ds, np.float64)
    assert idx["h2o"] == 1 and idx["o3"] == 3 and idx["h2o_self"] == 1 and idx["h2o_frgn"] == 1
    if kind == "sw":
        quiet, fac, spot, mg, sb = solar
        src = quiet + (mg - 0.1495954) * fac + (sb - 0.00066696) * spot     # Appendix A.6
        assert got.solar_src_tot == pytest.approx(src.sum(), rel=1e-14)
        np.testing.assert_allclose(got.solar_src_scaled, src / src.sum(), rtol=1e-14)
        lk = dataclasses.replace(lk, solar_src_tot=got.solar_src_tot)
    _same(lk, got)
    # the (0,0) -> (2,2) rewrite happened
    assert not np.any((got.key_species[0] == 0) & (got.key_species[1] == 0))


def test_gas_lookup_float32_and_planck_guard(tmp_path):
    lk = synthetic.make_gas_lookup("lw", n_bnd=2, seed=3)
    p = str(tmp_path / "lw.nc")
    nc_fixture.write_gas_file(p, lk)
    with netcdf_io.Dataset(p) as ds:
        got, _ = netcdf_io.lookup_lw(ds, np.float32)
    assert got.kmajor.dtype == np.float32 and got.minor_lower.kminor.dtype == np.float32
    np.testing.assert_array_equal(got.kmajor, lk.kmajor.astype(np.float32))
    # index-valued temperature_Planck (the g128 files) must be rejected, lookup_constructors.jl:196-200
    bad = dataclasses.replace(lk, t_planck=np.arange(196.0))
    p2 = str(tmp_path / "bad.nc")
    nc_fixture.write_gas_file(p2, bad)
    with netcdf_io.Dataset(p2) as ds, pytest.raises(ValueError, match="Kelvin"):
        netcdf_io.lookup_lw(ds, np.float64)


def test_cloud_and_aerosol_round_trip(tmp_path):
    wn = synthetic.SW_BAND_WN
    cld = synthetic.make_cloud_lookup("sw", wn.shape[1])
    aer = synthetic.make_aerosol_lookup("sw", wn)
    pc, pa = str(tmp_path / "cld.nc"), str(tmp_path / "aer.nc")
    nc_fixture.write_cloud_file(pc, cld, wn)
    nc_fixture.write_aerosol_file(pa, aer, wn)
    with netcdf_io.Dataset(pc) as ds:
        _same(cld, netcdf_io.lookup_cld(ds))
    with netcdf_io.Dataset(pa) as ds:
        got, idx_aero, idx_size = netcdf_io.lookup_aerosol(ds)
    _same(aer, got)
    assert got.iband_550nm > 0
    assert idx_aero["dust5"] == 11 and idx_aero["sea_salt5"] == 15 and len(idx_aero) == 15
    assert sorted(idx_size) == [1, 2, 8, 9, 10, 11, 12, 13, 14, 15]


def test_convert_and_flat_container(tmp_path):
    d = tmp_path / "rrtmgp-data"
    d.mkdir()
    lw = synthetic.make_gas_lookup("lw", n_bnd=3, seed=1)
    sw = synthetic.make_gas_lookup("sw", n_bnd=3, seed=1)
    nc_fixture.write_gas_file(str(d / netcdf_io.RRTMGP_DATA_FILES["lw"]), lw)
    nc_fixture.write_gas_file(str(d / netcdf_io.RRTMGP_DATA_FILES["sw"]), sw)
    nc_fixture.write_cloud_file(str(d / netcdf_io.RRTMGP_DATA_FILES["lw_cld"]),
                                synthetic.make_cloud_lookup("lw", 3), synthetic.LW_BAND_WN[:, :3])
    nc_fixture.write_aerosol_file(str(d / netcdf_io.RRTMGP_DATA_FILES["sw_aero"]),
                                  synthetic.make_aerosol_lookup("sw", synthetic.SW_BAND_WN[:, :3]),
                                  synthetic.SW_BAND_WN[:, :3])
    out = str(tmp_path / "lookups.npz")
    got = netcdf_io.convert_rrtmgp_data(str(d), out, np.float32)
    assert set(got) == {"lw", "sw", "lw_cld", "sw_aero", "idx_gases", "idx_aerosol", "idx_aerosize"}
    back = netcdf_io.load_lookups(out)
    assert set(back) == set(got)
    for k in ("lw", "sw", "lw_cld", "sw_aero"):
        _same(got[k], back[k])
        assert back[k].dtype == np.float32
    assert back["idx_gases"] == got["idx_gases"]
    # arrays come back Fortran-ordered so the C ABI can take their pointers directly
    assert back["lw"].kmajor.flags.f_contiguous and back["lw"].minor_lower.kminor.flags.f_contiguous
    with pytest.raises(FileNotFoundError):
        netcdf_io.convert_rrtmgp_data(str(tmp_path / "nowhere"), out)


def test_broken_hdf5_is_a_clear_error(tmp_path):
    """NetCDF-4 files go to the built-in HDF5 reader when neither netCDF4 nor h5py is importable
    (tests/test_hdf5_lite.py); a file that only has the signature fails loudly."""
    p = tmp_path / "x.nc"
    p.write_bytes(b"\x89HDF\r\n\x1a\n" + b"\xff" * 64)
    with pytest.raises(Exception):
        netcdf_io.Dataset(str(p))
