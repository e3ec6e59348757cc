#!/usr/bin/env python
"""Run the reference's three data-driven parity cases on the GPU against rrtmgp-data v1.9.

    python tools/run_reference_parity.py /path/to/rrtmgp-data [--ncol-clear 250] [--ncol-all 128]

This is the step that pins parity on real tables: the same inputs, solvers and tolerances
as test/clear_sky.jl, test/cloudy_sky.jl and test/all_sky_with_aerosols.jl (file names from
src/ArtifactPaths.jl:28-92).  rrtmgp-data is not available in the build image, so this
script is exercised there only on synthetic stand-in files (tests/test_reference_cases.py).
NetCDF-4 files need netCDF4 or h5py; `nccopy -k classic` copies work with scipy alone.
Prints one JSON report per case and exits non-zero if any reference criterion fails.
"""
import argparse
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from rrtmgp_jl_amd import netcdf_io, reference_cases as rc, rte   # noqa: E402
from rrtmgp_jl_amd.states import RRTMGPParameters                 # noqa: E402

RFMIP_INPUT = "examples/rfmip-clear-sky/inputs/multiple_input4MIPs_radiation_RFMIP_UColorado-RFMIP-1-2_none.nc"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("data_root")
    ap.add_argument("--ncol-clear", type=int, default=250)    # test/clear_sky.jl:11
    ap.add_argument("--ncol-all", type=int, default=128)      # test/cloudy_sky_utils.jl:104
    ap.add_argument("--container", default=None, help="write/read the flat lookup container here")
    a = ap.parse_args()
    params = RRTMGPParameters()
    ok = True
    for FT in (np.float64, np.float32):
        path = a.container or os.path.join("/tmp", f"rrtmgp_lookups_{np.dtype(FT).name}.npz")
        lk = netcdf_io.convert_rrtmgp_data(a.data_root, path, FT)
        idx = lk["idx_gases"]
        for lw_twostream in (False, True):
            # clear sky ---------------------------------------------------------------
            with netcdf_io.Dataset(os.path.join(a.data_root, RFMIP_INPUT)) as ds:
                nlay = ds.dim("layer")
                ws = rte.Workspace(a.ncol_clear, nlay, FT)
                case = rc.setup_clear_sky_as(ds, idx, 1, lk["lw"], a.ncol_clear, FT,
                                             *rc.hip_column_routines(ws, params))
            f_lw, f_sw = rc.solve_case(case, lk, FT, lw_twostream, clouds=False, aerosols=False)
            comp = rc.load_clear_sky_comparison(a.data_root, 1, case.bot_at_1, a.ncol_clear)
            reports = [rc.check_against_reference("clear_sky", f_lw, f_sw, comp, FT, lw_twostream,
                                                  case.bcs_sw.cos_zenith)]
            # all sky, without and with aerosols ---------------------------------------------
            for problem, name in (("gas_clouds", "cloudy_sky"), ("gas_clouds_aerosols", "all_sky_with_aerosols")):
                ncol_ds = rc.ncol_ds_all_sky(a.data_root, problem)
                with netcdf_io.Dataset(os.path.join(a.data_root, rc.REFERENCE_FILES[(problem, "lw")])) as ds:
                    nlay = ds.dim("lay")
                    ws = rte.Workspace(a.ncol_all, nlay, FT)
                    routines = rc.hip_column_routines(ws, params)
                    if problem == "gas_clouds":
                        case = rc.setup_cloudy_sky_as(ds, idx, lk["lw"], lk["sw"], lk["lw_cld"], FT(1), a.ncol_all,
                                                      ncol_ds, FT, *routines)
                    else:
                        case = rc.setup_allsky_with_aerosols_as(ds, idx, lk["idx_aerosol"], lk["idx_aerosize"],
                                                                lk["lw"], lk["sw"], lk["lw_cld"], FT(1), a.ncol_all,
                                                                ncol_ds, FT, *routines)
                f_lw, f_sw = rc.solve_case(case, lk, FT, lw_twostream, clouds=True,
                                           aerosols=problem == "gas_clouds_aerosols")
                comp = rc.load_all_sky_comparison(a.data_root, problem, case.bot_at_1, a.ncol_all)
                reports.append(rc.check_against_reference(name, f_lw, f_sw, comp, FT, lw_twostream))
            for r in reports:
                r.update(FT=np.dtype(FT).name, lw_solver="TwoStreamLWRTE" if lw_twostream else "NoScatLWRTE")
                print(json.dumps(r))
                ok = ok and r["passed"]
    sys.exit(0 if ok else 1)


if __name__ == "__main__":
    main()
