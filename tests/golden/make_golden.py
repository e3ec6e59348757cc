#!/usr/bin/env python3
"""Generates tests/golden/*.npz: expected broadband fluxes of small seeded cases.

The reference's own golden vectors (rte-rrtmgp fluxes inside the rrtmgp-data v1.9
artifact, test/reference_files.jl:15-62) are not available offline and Julia is not
installed, so these fixtures come from this repository's Float64 C oracle
(oracle/rrtmgp_oracle.c) on synthetic tables; the oracle itself is pinned by
tests/test_oracle_known_answers.py and cross-checked by oracle/np_oracle.py.
Inputs are NOT stored: they are regenerated from (seed, sizes) by
rrtmgp_jl_amd.synthetic, which is deterministic.  Run from the repo root:

    python tests/golden/make_golden.py
"""
import dataclasses
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import rrtmgp_jl_amd  # noqa: E402,F401
from rrtmgp_jl_amd import synthetic as S  # noqa: E402

CASES = {
    # name: (ncol, nlay, kwargs for make_columns, solver options)
    "clear_gm": dict(ncol=6, nlay=20, cols=dict(clouds=False, night_fraction=0.3), aero=False),
    "cloudy_full": dict(ncol=6, nlay=24, cols=dict(vmr_kind="full", night_fraction=0.2, inc_flux_ngpt=24), aero=False),
    "aerosol_mcica": dict(ncol=5, nlay=17, cols=dict(aerosols=True, random_cld_frac=True, cos_zenith=0.7), aero=True),
}
SEED = 2026


def tables():
    lw = S.make_gas_lookup("lw", np.float64, seed=7, n_bnd=3, gpt_per_bnd=[8, 4, 12])
    sw = S.make_gas_lookup("sw", np.float64, seed=7, n_bnd=3, gpt_per_bnd=[6, 10, 4])
    return dict(lw=lw, sw=sw, cld_lw=S.make_cloud_lookup("lw", 3, seed=7), cld_sw=S.make_cloud_lookup("sw", 3, seed=7),
                aero_lw=S.make_aerosol_lookup("lw", lw.bnd_lims_wn, seed=7),
                aero_sw=dataclasses.replace(S.make_aerosol_lookup("sw", sw.bnd_lims_wn, seed=7), iband_550nm=2))


def inputs(case):
    c = CASES[case]
    return S.make_columns(c["ncol"], c["nlay"], np.float64, seed=SEED, n_bnd_lw=3, n_bnd_sw=3, **c["cols"])


def run(case, solve_lw, solve_sw):
    """Expected arrays of one case given solver callables with the oracle's signature."""
    t = tables()
    c = CASES[case]
    as_, lb, sb = inputs(case)
    clouds = as_.cloud_state is not None
    out = {}
    for two in (True, False):
        f = solve_lw(as_, lb, t["lw"], t["cld_lw"] if clouds else None, t["aero_lw"] if c["aero"] else None,
                     twostream=two, seed=11)
        tag = "lw2s" if two else "lwns"
        out[f"{tag}_up"], out[f"{tag}_dn"] = f.as_nlev_ncol("flux_up"), f.as_nlev_ncol("flux_dn")
    # no-scattering with 3 Gauss-Jacobi angles (AngularDiscretizations.jl:47-49; longwave_noscat.jl:45-96)
    f = solve_lw(as_, lb, t["lw"], t["cld_lw"] if clouds else None, t["aero_lw"] if c["aero"] else None,
                 twostream=False, seed=11, n_gauss_angles=3)
    out["lwns3_up"], out["lwns3_dn"] = f.as_nlev_ncol("flux_up"), f.as_nlev_ncol("flux_dn")
    f = solve_sw(as_, sb, t["sw"], t["cld_sw"] if clouds else None, t["aero_sw"] if c["aero"] else None, seed=11)
    out["sw_up"], out["sw_dn"], out["sw_dir"] = (f.as_nlev_ncol(n) for n in ("flux_up", "flux_dn", "flux_dn_dir"))
    if clouds:
        out["cover_sw"] = np.asarray(as_.cloud_state.cld_cover_sw).copy()
    if c["aero"]:
        out["aod_ext"] = np.asarray(as_.aerosol_state.aod_sw_ext).copy()
    return out


if __name__ == "__main__":
    from oracle import oracle as O
    here = os.path.dirname(os.path.abspath(__file__))
    for case in CASES:
        np.savez_compressed(os.path.join(here, f"{case}.npz"), **run(case, O.solve_lw, O.solve_sw))
        print("wrote", case)
