"""Mutation check of the two-transcription argument ("What's weak" 1): the GPU parity tests compare the
HIP kernels with the C oracle, and the C oracle is cross-checked by an independent numpy restatement written from the Julia
sources (tests/test_golden.py).  That cross-check is only worth something if it FAILS when one of the two is wrong.  Here a
literal of the C oracle is flipped — in a copy, compiled into a temporary library — and the numpy twin must disagree with the
mutant on a golden case, for every piece of the path that used to rest on a single transcription: the MERRA species lists,
the size-bin fallback, the maximum-random overlap scaling, the ice roughness row, delta scaling, the gray optics."""
import os
import shutil
import subprocess
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "golden"))
import make_golden as G  # noqa: E402
from oracle import np_oracle as NP  # noqa: E402
from oracle import oracle as O  # noqa: E402

ORACLE_DIR = os.path.dirname(os.path.abspath(O.__file__))

MUTATIONS = {
    # name: (text in rrtmgp_oracle_impl.inc, replacement, golden case that must notice)
    "dust_species_list": ("dust_ids[5] = {1, 8, 9, 10, 11}", "dust_ids[5] = {1, 8, 9, 10, 12}", "aerosol_dense"),
    "sea_salt_species_list": ("salt_ids[5] = {2, 12, 13, 14, 15}", "salt_ids[5] = {2, 11, 13, 14, 15}", "aerosol_dense"),
    "size_bin_fallback_is_the_last_bin": ("        bin = nbins;\n", "        bin = 1;\n", "aerosol_dense"),
    "overlap_rescales_the_draw": ("rrtmgp_oracle_mcica_uniform(seed, gcol_global, igpt, is_sw, draw++) *\n"
                                  "                              (double)((FT)1 - cld_frac_ilayplus1)",
                                  "rrtmgp_oracle_mcica_uniform(seed, gcol_global, igpt, is_sw, draw++)", "aerosol_mcica"),
    "ice_roughness_row": ("(size_t)lk->nband * (size_t)(ice_rgh - 1)", "(size_t)lk->nband * (size_t)(2 - 1)", "ice_rgh3"),
    "hydrophobic_black_carbon_is_species_5": ("if ((m = aero_mass[IX2(5, glay, NA)]) > (FT)0) {", "if ((m = aero_mass[IX2(4, glay, NA)]) > (FT)0) {",
                                              "aerosol_dense"),
    "organic_carbon_uses_its_own_table": ("FN(aero_rh_props)(lk->organic_carbon_rh, lk->nrh", "FN(aero_rh_props)(lk->black_carbon_rh, lk->nrh",
                                          "aerosol_dense"),
    "aerosols_are_delta_scaled_in_the_shortwave": ("if (delta_scaling) FN(delta_scale)(&ta, &ssa_aero, &g_aero);", ";", "aerosol_dense"),
    "delta_scaled_asymmetry": ("FT g_s = (*g) / FMAX(EPS, (FT)1 + (*g));", "FT g_s = (*g) / FMAX(EPS, (FT)1 - (*g));", "ice_rgh1"),
}


def _twin(case):
    t = G.tables()
    as_, lb, sb = G.inputs(case)
    aero = G.CASES[case]["aero"]
    out = {}
    out["lw2s_up"], out["lw2s_dn"] = NP.solve_lw_2stream(t["lw"], as_, lb, t["cld_lw"], lka=t["aero_lw"] if aero else None, seed=11)
    out["sw_up"], out["sw_dn"], out["sw_dir"] = NP.solve_sw_2stream(t["sw"], as_, sb, t["cld_sw"],
                                                                   lka=t["aero_sw"] if aero else None, seed=11)
    return out


def _agree(twin, got):
    return all(np.allclose(twin[k], got[k], rtol=1e-11, atol=1e-10) for k in twin)


@pytest.fixture(scope="module")
def twins():
    return {case: _twin(case) for case in sorted({m[2] for m in MUTATIONS.values()})}


def _build_mutant(tmp, name, old, new):
    d = os.path.join(tmp, name)
    os.makedirs(os.path.join(d, "oracle"))
    for f in ("rrtmgp_oracle.c", "rrtmgp_oracle_impl.inc", "rrtmgp_oracle.h"):
        shutil.copy(os.path.join(ORACLE_DIR, f), os.path.join(d, "oracle", f))
    shutil.copytree(os.path.join(ORACLE_DIR, "..", "include"), os.path.join(d, "include"))
    inc = os.path.join(d, "oracle", "rrtmgp_oracle_impl.inc")
    src = open(inc).read()
    assert src.count(old) == 1, f"{name}: the literal to flip must occur exactly once ({src.count(old)})"
    open(inc, "w").write(src.replace(old, new))
    so = os.path.join(d, "mutant.so")
    subprocess.run(["gcc", "-O1", "-fPIC", "-std=c11", "-ffp-contract=off", "-fno-fast-math", "-fopenmp", "-shared", "-o", so,
                    os.path.join(d, "oracle", "rrtmgp_oracle.c"), "-lm"], check=True, capture_output=True)
    return so


def test_the_unmutated_oracle_agrees_with_the_twin(twins):
    for case, twin in twins.items():
        assert _agree(twin, G.run(case, O.solve_lw, O.solve_sw)), case


@pytest.mark.parametrize("name", list(MUTATIONS))
def test_twin_catches_a_flipped_literal(tmp_path, twins, name):
    old, new, case = MUTATIONS[name]
    mutant = O.load_library(_build_mutant(str(tmp_path), name, old, new))
    with O.using(mutant):
        got = G.run(case, O.solve_lw, O.solve_sw)
    assert not _agree(twins[case], got), f"the numpy twin did not notice mutation `{name}`"
    # and the committed fixture (made by the unmutated oracle) notices it too
    exp = np.load(os.path.join(os.path.dirname(__file__), "golden", f"{case}.npz"))
    assert not all(np.allclose(got[k], exp[k], rtol=1e-12, atol=1e-12) for k in exp.files)
