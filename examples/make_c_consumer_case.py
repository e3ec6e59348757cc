#!/usr/bin/env python3
"""Writes the raw case file that examples/c_consumer.c solves on the GPU with nothing but include/rrtmgp_hip.h:

    python examples/make_c_consumer_case.py case.bin [ncol nlay]

Contents: synthetic Float64 lookup tables of the rrtmgp-data v1.9 dimensionality (LW / SW gas optics, LW / SW cloud optics)
in the reference's in-memory form, an all-sky AtmosphericState with boundary conditions, and the fluxes the CPU oracle
(oracle/rrtmgp_oracle.c, test infrastructure) computes for them - the numbers the C program compares its own solve with.

Format: the 8 bytes "RRCASE1\\0", then records until end of file:
    char name[32] (NUL padded) | int32 dtype (1 = float64, 2 = int64) | int32 ndim | int64 dims[4] | int64 nbytes | data,
data in column-major order (the first dimension fastest, as the C ABI takes every array), padded to a multiple of 8 bytes."""
import os
import struct
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import rrtmgp_jl_amd  # noqa: E402,F401
from oracle import oracle as O  # noqa: E402
from rrtmgp_jl_amd import synthetic as S  # noqa: E402

SEED = 2026


def main(path, ncol=24, nlay=40):
    ft = np.float64
    lw, sw = S.make_gas_lookup("lw", ft), S.make_gas_lookup("sw", ft)
    cl, cs = S.make_cloud_lookup("lw", lw.n_bnd, ft), S.make_cloud_lookup("sw", sw.n_bnd, ft)
    as_, lb, sb = S.make_columns(ncol, nlay, ft, seed=SEED, clouds=True, cld_frac=0.6, night_fraction=0.2)
    f_lw = O.solve_lw(as_, lb, lw, cl, seed=SEED)
    f_sw = O.solve_sw(as_, sb, sw, cs, seed=SEED)
    rec = {}

    def put(name, a, dt=None):
        a = np.asarray(a)
        if dt is None:
            dt = np.int64 if a.dtype.kind in "iu" else np.float64
        rec[name] = np.asfortranarray(a, dtype=dt)

    for tag, lk in (("lw", lw), ("sw", sw)):
        put(tag + ".scalars", [lk.idx_h2o, lk.p_ref_tropo, lk.p_ref_min, lk.t_ref_min, lk.t_ref_max, lk.solar_src_tot], np.float64)
        for n in ("key_species", "major_gpt2bnd", "kmajor", "ln_p_ref", "t_ref", "vmr_ref"):
            put(f"{tag}.{n}", getattr(lk, n))
        for reg, m in (("lower", lk.minor_lower), ("upper", lk.minor_upper)):
            for n in ("bnd_st", "gpt_st", "gasdata", "kminor"):
                put(f"{tag}.minor_{reg}.{n}", getattr(m, n))
    for n in ("planck_fraction", "t_planck", "tot_planck"):
        put("lw." + n, getattr(lw, n))
    for n in ("rayl_lower", "rayl_upper", "solar_src_scaled"):
        put("sw." + n, getattr(sw, n))
    for tag, c in (("cld_lw", cl), ("cld_sw", cs)):
        put(tag + ".dims", c.dims)
        put(tag + ".bounds", c.bounds)
        put(tag + ".liqdata", c.liqdata)
        put(tag + ".icedata", c.icedata)
    put("as.layerdata", as_.layerdata); put("as.p_lev", as_.p_lev); put("as.t_lev", as_.t_lev); put("as.t_sfc", as_.t_sfc)
    put("as.vmr_h2o", as_.vmr.vmr_h2o); put("as.vmr_o3", as_.vmr.vmr_o3); put("as.vmr", as_.vmr.vmr)
    c = as_.cloud_state
    for n in ("cld_r_eff_liq", "cld_r_eff_ice", "cld_path_liq", "cld_path_ice", "cld_frac"):
        put("as." + n, getattr(c, n))
    put("as.ice_rgh", [c.ice_rgh])
    put("lw_bcs.sfc_emis", lb.sfc_emis)
    for n in ("cos_zenith", "toa_flux", "sfc_alb_direct", "sfc_alb_diffuse"):
        put("sw_bcs." + n, getattr(sb, n))
    put("seed", [SEED])
    for n in ("flux_up", "flux_dn", "flux_net"):
        put("expect.lw." + n, getattr(f_lw, n))
    for n in ("flux_up", "flux_dn", "flux_net", "flux_dn_dir"):
        put("expect.sw." + n, getattr(f_sw, n))
    put("expect.cld_cover_lw", c.cld_cover_lw); put("expect.cld_cover_sw", c.cld_cover_sw)
    with open(path, "wb") as fh:
        fh.write(b"RRCASE1\0")
        for name, a in rec.items():
            assert len(name) < 32 and a.ndim <= 4, name
            dims = list(a.shape) + [1] * (4 - a.ndim)
            raw = a.tobytes(order="F")
            fh.write(struct.pack("<32sii4qq", name.encode(), 1 if a.dtype == np.float64 else 2, a.ndim, *dims, len(raw)))
            fh.write(raw + b"\0" * (-len(raw) % 8))
    print(f"{path}: {len(rec)} arrays, {os.path.getsize(path) / 1e6:.1f} MB, {ncol} columns x {nlay} layers")


if __name__ == "__main__":
    main(sys.argv[1], *(int(x) for x in sys.argv[2:4]))
