#!/usr/bin/env python
"""rrtmgp-data NetCDF lookups -> one flat .npz container (no NetCDF needed at run time).

    python tools/convert_rrtmgp_data.py /path/to/rrtmgp-data lookups_f32.npz --dtype f32
"""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from rrtmgp_jl_amd import netcdf_io   # noqa: E402

if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("data_dir")
    ap.add_argument("out")
    ap.add_argument("--dtype", choices=["f32", "f64"], default="f64")
    a = ap.parse_args()
    got = netcdf_io.convert_rrtmgp_data(a.data_dir, a.out, np.float32 if a.dtype == "f32" else np.float64)
    for k, v in got.items():
        print(k, type(v).__name__, getattr(v, "n_gpt", ""))
