"""N1: the reference's three data-driven parity cases (test/read_*.jl + *_utils.jl), driven
from schema-faithful synthetic example files.  CPU tests check the readers (orientation
flip, experiment slicing, units, column replication, cloud pattern, aerosol scatter); the
GPU test runs the whole case through the C ABI and applies the reference's pass criteria
against "reference flux files" produced by the oracle."""
import os

import numpy as np
import pytest

from rrtmgp_jl_amd import netcdf_io, reference_cases as rc, synthetic
from rrtmgp_jl_amd.states import TEST_PARAMETERS, VmrGM, Vmr
from oracle import oracle

import nc_fixture

NSITE, NLAY, NEXPT = 6, 12, 3
GM_UNITS = {"carbon_dioxide_GM": "1e-6", "nitrous_oxide_GM": "1e-9", "carbon_monoxide_GM": "1e-9",
            "methane_GM": "1e-9", "oxygen_GM": "1", "nitrogen_GM": "1"}


def _gm():
    gm = {}
    for gas, var in rc.RFMIP_GM.items():
        units = GM_UNITS.get(var, "1e-12")
        val = {"carbon_dioxide_GM": 348.0, "nitrous_oxide_GM": 306.0, "methane_GM": 1650.0, "oxygen_GM": 0.2095,
               "nitrogen_GM": 0.7808, "carbon_monoxide_GM": 100.0}.get(var, 20.0 + len(var))
        gm[var] = (val, units)
    return gm


def _tables(st, FT=np.float64):
    """conftest's table set under the key names netcdf_io.load_lookups() uses."""
    m = dict(lw=st["lw"], sw=st["sw"], lw_cld=st["cld_lw"], sw_cld=st["cld_sw"], lw_aero=st["aero_lw"],
             sw_aero=st["aero_sw"])
    return {k: v.astype(FT) for k, v in m.items()}


def _oracle_routines(params):
    return (lambda p_lev, h2o, lat=None: oracle.compute_col_gas(p_lev, params, h2o, lat),
            lambda p_lay, t_lay, h2o: oracle.compute_relative_humidity(p_lay, t_lay, params, h2o))


@pytest.fixture(scope="module")
def rfmip(tmp_path_factory, small_tables64):
    d = tmp_path_factory.mktemp("rfmip")
    as_, lw_bcs, sw_bcs = synthetic.make_columns(NSITE, NLAY, clouds=False, n_bnd_lw=small_tables64["lw"].n_bnd,
                                                 n_bnd_sw=small_tables64["sw"].n_bnd, night_fraction=0.34, seed=11)
    ld = as_.layerdata
    rng = np.random.default_rng(5)
    emis, alb = rng.uniform(0.9, 1.0, NSITE), rng.uniform(0.05, 0.4, NSITE)
    sza = np.degrees(np.arccos(np.clip(sw_bcs.cos_zenith, -1, 1)))
    sza[1] = 95.0   # a night site
    tsi = rng.uniform(1355, 1365, NSITE)
    p = str(d / "in.nc")
    nc_fixture.write_rfmip_input(p, as_.p_lev, ld[1], as_.t_lev, ld[2], as_.t_sfc, as_.vmr.vmr_h2o, as_.vmr.vmr_o3,
                                 _gm(), emis, alb, sza, tsi, np.zeros(NSITE), np.zeros(NSITE), n_expt=NEXPT)
    return dict(path=p, as_=as_, emis=emis, alb=alb, sza=sza, tsi=tsi, dir=d)


@pytest.mark.parametrize("vmr_type", [VmrGM, Vmr])
def test_clear_sky_reader(rfmip, small_tables64, vmr_type):
    lw = small_tables64["lw"]
    col_gas, rh = _oracle_routines(TEST_PARAMETERS)
    ncol, expt = 15, 2     # 2.5 replications of the 6 sites, second experiment
    with netcdf_io.Dataset(rfmip["path"]) as ds:
        case = rc.setup_clear_sky_as(ds, synthetic.IDX_GASES, expt, lw, ncol, np.float64, col_gas, rh, vmr_type)
    src, a = rfmip["as_"], case.as_
    rep = np.arange(ncol) % NSITE
    assert case.bot_at_1 is False                      # file is top-first; arrays come back bottom-first
    assert np.all(a.p_lev[0] > a.p_lev[-1])
    np.testing.assert_array_equal(a.p_lev[:-1], src.p_lev[:-1][:, rep])
    assert np.all(a.p_lev[-1] == lw.p_ref_min)         # read_clear_sky.jl:66
    np.testing.assert_allclose(a.t_lev, src.t_lev[:, rep] + (expt - 1), rtol=0, atol=1e-12)
    np.testing.assert_allclose(a.layerdata[2], src.layerdata[2][:, rep] + (expt - 1), rtol=0, atol=1e-12)
    np.testing.assert_array_equal(a.layerdata[1], src.layerdata[1][:, rep])
    np.testing.assert_allclose(a.t_sfc, src.t_sfc[rep] + (expt - 1), rtol=0, atol=1e-12)
    h2o = src.vmr.vmr_h2o[:, rep] * (1 + 0.1 * (expt - 1))
    if vmr_type is VmrGM:
        np.testing.assert_allclose(a.vmr.vmr_h2o, h2o, rtol=1e-15)
        wm = a.vmr.vmr
    else:
        np.testing.assert_allclose(a.vmr.vmr[synthetic.IDX_GASES["h2o"] - 1], h2o, rtol=1e-15)
        wm = a.vmr.vmr[:, 3, 4]
    assert wm[synthetic.IDX_GASES["co2"] - 1] == pytest.approx(348.0e-6 * 1.01, rel=1e-14)
    assert wm[synthetic.IDX_GASES["cf4"] - 1] == pytest.approx((20.0 + len("cf4_GM")) * 1e-12 * 1.01, rel=1e-14)
    # col_dry / rel_hum were filled by the supplied routines from the final arrays
    np.testing.assert_array_equal(a.layerdata[0], col_gas(a.p_lev, h2o if vmr_type is Vmr else a.vmr.vmr_h2o))
    # boundary conditions: one value per site broadcast over bands, replicated over columns
    assert case.bcs_lw.sfc_emis.shape == (lw.n_bnd, ncol)
    np.testing.assert_array_equal(case.bcs_lw.sfc_emis[2], rfmip["emis"][rep])
    np.testing.assert_allclose(case.bcs_sw.cos_zenith, np.cos(np.radians(rfmip["sza"]))[rep], rtol=1e-15)
    assert case.bcs_sw.cos_zenith[1] < 0
    np.testing.assert_array_equal(case.bcs_sw.toa_flux, rfmip["tsi"][rep])


def _allsky_files(d, aerosols, ncol_in=4):
    as_, _, _ = synthetic.make_columns(ncol_in, NLAY, clouds=False, seed=3)
    ld = as_.layerdata
    aero = None
    if aerosols:
        rng = np.random.default_rng(9)
        typ = rng.integers(0, 16, (NLAY, ncol_in))
        aero = (typ, rng.uniform(0.2, 8.0, (NLAY, ncol_in)), rng.uniform(1e-6, 1e-4, (NLAY, ncol_in)))
    p = str(d / ("in_aero.nc" if aerosols else "in.nc"))
    nc_fixture.write_allsky_input(p, as_.p_lev, ld[1], as_.t_lev, ld[2], as_.vmr.vmr_h2o, as_.vmr.vmr_o3, aero)
    return p, as_, aero


def test_all_sky_readers(tmp_path, small_tables64):
    t = _tables(small_tables64)
    col_gas, rh = _oracle_routines(TEST_PARAMETERS)
    p, src, _ = _allsky_files(tmp_path, False)
    ncol, ncol_ds = 10, 4
    with netcdf_io.Dataset(p) as ds:
        case = rc.setup_cloudy_sky_as(ds, synthetic.IDX_GASES, t["lw"], t["sw"], t["lw_cld"], 1.0, ncol, ncol_ds,
                                      np.float64, col_gas, rh)
    a = case.as_
    assert case.bot_at_1 is False
    # only the first input column is used, replicated (read_cloudy_sky.jl:52-66)
    np.testing.assert_array_equal(a.p_lev, np.repeat(src.p_lev[:, :1], ncol, 1))
    np.testing.assert_array_equal(a.t_sfc, np.full(ncol, src.t_lev[0, 0]))
    assert a.vmr.vmr[synthetic.IDX_GASES["co2"] - 1, 2, 7] == 348e-6
    # cloud pattern: two columns in three, 100-900 hPa, liquid above 263 K and ice below 273 K
    cs = a.cloud_state
    icol_ds = np.arange(ncol) % ncol_ds + 1
    assert np.all(cs.cld_frac[:, icol_ds % 3 == 0] == 0)
    p_lay, t_lay = a.layerdata[1], a.layerdata[2]
    inb = (p_lay > 1e4) & (p_lay < 9e4) & (icol_ds % 3 != 0)[None]
    assert inb.any()
    np.testing.assert_array_equal(cs.cld_frac, np.where(inb, 1.0, 0.0))
    np.testing.assert_array_equal(cs.cld_path_liq, np.where(inb & (t_lay > 263), 10.0, 0.0))
    np.testing.assert_array_equal(cs.cld_r_eff_ice, np.where(inb & (t_lay < 273), t["lw_cld"].bounds[2:].mean(), 0.0))
    assert np.all(case.bcs_sw.toa_flux == t["sw"].solar_src_tot) and np.all(case.bcs_sw.cos_zenith == 0.86)

    p, src, aero = _allsky_files(tmp_path, True)
    with netcdf_io.Dataset(p) as ds:
        case = rc.setup_allsky_with_aerosols_as(ds, synthetic.IDX_GASES, netcdf_io.AEROSOL_INDEX,
                                                netcdf_io.AEROSIZE_INDEX, t["lw"], t["sw"], t["lw_cld"], 1.0, ncol,
                                                ncol_ds, np.float64, col_gas, rh)
    ae = case.as_.aerosol_state
    assert ae.aero_mass.shape == (15, NLAY, ncol) and ae.aero_size.shape == (15, NLAY, ncol)
    typ, size, mass = aero
    for icol in (0, 3, 5, 9):
        for ilay in (0, 4, NLAY - 1):
            ty = typ[ilay, icol % 4]
            col_mass = ae.aero_mass[:, ilay, icol]
            if ty == 0:
                assert not col_mass.any()
            else:
                assert col_mass[ty - 1] == mass[ilay, icol % 4] and np.count_nonzero(col_mass) == 1
                want = size[ilay, icol % 4] if ty in netcdf_io.AEROSIZE_INDEX else 0.0
                assert ae.aero_size[ty - 1, ilay, icol] == want


def test_comparison_metrics_and_tolerances():
    assert rc.TOLERANCES["clear_sky"]["lw_noscat"][np.float64] == 1e-4      # test/clear_sky.jl:7
    assert rc.TOLERANCES["cloudy_sky"]["sw"][np.float32] == 0.06            # test/cloudy_sky.jl:8
    up = np.array([[10.0, 20.0], [11.0, 19.0]])
    dn = np.array([[10.0, 5.0], [1.0, 19.0]])
    e = rc.compare_fluxes(up + 0.5, dn, up, dn, np.float64)
    assert e["up"] == 0.5 and e["dn"] == 0.0 and e["net"] == 0.5
    assert e["rel_net"] == pytest.approx(0.5)          # net = 0 entries stay absolute (0.5), 0.5/10, 0.5/15
    assert rc.night_columns_are_dark(np.zeros((3, 2)), np.zeros((3, 2)), np.array([-0.1, 0.0]))
    assert not rc.night_columns_are_dark(np.ones((3, 2)), np.zeros((3, 2)), np.array([-0.1, 0.5]))


@pytest.mark.gpu
@pytest.mark.parametrize("FT,lw_twostream", [(np.float64, True), (np.float64, False), (np.float32, True)])
def test_clear_sky_case_end_to_end(rfmip, small_tables64, tmp_path, FT, lw_twostream):
    """The whole clear-sky driver on the GPU.  The "RFMIP reference flux files" are written
    from the Float64 oracle, so the reference's own criteria (per-FT tolerances, dark night
    columns) are applied to HIP-vs-oracle differences."""
    from rrtmgp_jl_amd import rte
    t = _tables(small_tables64, FT)
    expt, ncol = 2, NSITE
    col_gas64, rh64 = _oracle_routines(TEST_PARAMETERS)
    with netcdf_io.Dataset(rfmip["path"]) as ds:
        ref_case = rc.setup_clear_sky_as(ds, synthetic.IDX_GASES, expt, small_tables64["lw"], ncol, np.float64,
                                         col_gas64, rh64)
    f_lw = oracle.solve_lw(ref_case.as_, ref_case.bcs_lw, small_tables64["lw"], twostream=lw_twostream)
    f_sw = oracle.solve_sw(ref_case.as_, ref_case.bcs_sw, small_tables64["sw"])
    root = tmp_path / "data"
    for key, var, f in ((("gas", "lw", "up"), "rlu", f_lw.flux_up), (("gas", "lw", "dn"), "rld", f_lw.flux_dn),
                        (("gas", "sw", "up"), "rsu", f_sw.flux_up), (("gas", "sw", "dn"), "rsd", f_sw.flux_dn)):
        path = root / rc.REFERENCE_FILES[key]
        os.makedirs(path.parent, exist_ok=True)
        flux3 = np.repeat(np.asarray(f)[:, :, None], NEXPT, 2) * np.array([7.0, 1.0, 3.0])   # only expt 2 is right
        nc_fixture.write_rfmip_flux(str(path), var, flux3)
    ws = rte.Workspace(ncol, NLAY, FT)
    col_gas, rh = rc.hip_column_routines(ws, TEST_PARAMETERS)
    with netcdf_io.Dataset(rfmip["path"]) as ds:
        case = rc.setup_clear_sky_as(ds, synthetic.IDX_GASES, expt, t["lw"], ncol, FT, col_gas, rh)
    got_lw, got_sw = rc.solve_case(case, t, FT, lw_twostream, clouds=False, aerosols=False)
    comp = rc.load_clear_sky_comparison(str(root), expt, case.bot_at_1, ncol)
    rep = rc.check_against_reference("clear_sky", got_lw, got_sw, comp, FT, lw_twostream, case.bcs_sw.cos_zenith)
    assert rep["passed"], rep
    if FT is np.float64:
        assert rep["lw"]["net"] < 1e-9 and rep["sw"]["net"] < 1e-9, rep


@pytest.mark.gpu
def test_all_sky_with_aerosols_case_end_to_end(small_tables64, tmp_path):
    t = _tables(small_tables64)
    p, _, _ = _allsky_files(tmp_path, True)
    ncol, ncol_ds = 9, 4
    col_gas64, rh64 = _oracle_routines(TEST_PARAMETERS)
    args = (synthetic.IDX_GASES, netcdf_io.AEROSOL_INDEX, netcdf_io.AEROSIZE_INDEX, t["lw"], t["sw"], t["lw_cld"], 1.0)
    with netcdf_io.Dataset(p) as ds:
        ref_case = rc.setup_allsky_with_aerosols_as(ds, *args, ncol_ds, ncol_ds, np.float64, col_gas64, rh64)
    f_lw = oracle.solve_lw(ref_case.as_, ref_case.bcs_lw, t["lw"], t["lw_cld"], t["lw_aero"])
    f_sw = oracle.solve_sw(ref_case.as_, ref_case.bcs_sw, t["sw"], t["sw_cld"], t["sw_aero"])
    root = tmp_path / "data"
    for lam, f in (("lw", f_lw), ("sw", f_sw)):
        path = root / rc.REFERENCE_FILES[("gas_clouds_aerosols", lam)]
        os.makedirs(path.parent, exist_ok=True)
        nc_fixture.write_allsky_flux(str(path), lam, np.asarray(f.flux_up), np.asarray(f.flux_dn))
    assert rc.ncol_ds_all_sky(str(root), "gas_clouds_aerosols") == ncol_ds
    from rrtmgp_jl_amd import rte
    ws = rte.Workspace(ncol, NLAY, np.float64)
    col_gas, rh = rc.hip_column_routines(ws, TEST_PARAMETERS)
    with netcdf_io.Dataset(p) as ds:
        case = rc.setup_allsky_with_aerosols_as(ds, *args, ncol, ncol_ds, np.float64, col_gas, rh)
    got_lw, got_sw = rc.solve_case(case, t, np.float64, True, clouds=True, aerosols=True)
    comp = rc.load_all_sky_comparison(str(root), "gas_clouds_aerosols", case.bot_at_1, ncol)
    rep = rc.check_against_reference("all_sky_with_aerosols", got_lw, got_sw, comp, np.float64, True)
    # overcast (cld_frac = 1) columns carry no McICA randomness, so HIP == oracle to round-off
    assert rep["passed"] and rep["lw"]["net"] < 1e-9 and rep["sw"]["net"] < 1e-9, rep
