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
    # the other two ice roughness classes of LookUpCld (cloud_optics.jl:207-244, LookUpTables.jl:260-284); round 5
    "ice_rgh1": dict(ncol=4, nlay=14, cols=dict(random_cld_frac=True, night_fraction=0.25), aero=False, ice_rgh=1),
    "ice_rgh3": dict(ncol=4, nlay=14, cols=dict(random_cld_frac=True, aerosols=True), aero=True, ice_rgh=3),
    # every MERRA species in play in most layers, particle sizes inside, between and outside the size bins (the "else the
    # last bin" rule of locate_merra_size_bin, aerosol_optics.jl:438-451), relative humidities beyond both table ends: the
    # synthetic recipe of make_columns activates ONE species per (layer, column) and only low down — the mutation check of
    # round 5 (tests/test_oracle_mutations.py) showed that a wrong species list passed `aerosol_mcica` unnoticed
    "aerosol_dense": dict(ncol=3, nlay=12, cols=dict(aerosols=True, random_cld_frac=True, night_fraction=0.3), aero=True,
                          dense_aerosols=True),
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
    as_, lb, sb = S.make_columns(c["ncol"], c["nlay"], np.float64, seed=SEED, n_bnd_lw=3, n_bnd_sw=3, **c["cols"])
    if "ice_rgh" in c:
        as_.cloud_state.ice_rgh = c["ice_rgh"]
    if c.get("dense_aerosols"):
        rng = np.random.default_rng(SEED + 1)
        ae = as_.aerosol_state
        shape = ae.aero_mass.shape                      # (15, nlay, ncol)
        on = rng.uniform(size=shape) < 0.6
        on[:, 0, :] = True                              # a layer with all fifteen
        on[:, 1, :] = False                             # and one with none (aero_mask false)
        ae.aero_mass[...] = np.where(on, 10.0 ** rng.uniform(-7.0, -4.0, shape), 0.0)
        ae.aero_size[...] = rng.choice([0.05, 0.1, 0.5, 1.0, 1.4, 1.8, 2.9, 3.0, 5.0, 6.0, 9.9, 10.0, 12.0, 30.0], shape)
        as_.layerdata[3] = rng.uniform(-0.1, 1.2, as_.layerdata[3].shape)   # relative humidity, beyond both ends of rh_levels
    return as_, lb, sb


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
    for case in (sys.argv[1:] or CASES):   # (name the new cases to leave the committed fixtures of the others untouched)
        np.savez_compressed(os.path.join(here, f"{case}.npz"), **run(case, O.solve_lw, O.solve_sw))
        print("wrote", case)
