#!/usr/bin/env python3
"""For fuzz seeds whose Float32 SW comparison exceeds its budget: who is off?  HIP-Float32 and the oracle in Float32 are both
compared with the oracle in Float64 on the SAME (Float32-valued) inputs — the quantity the reference's own ratchet bounds
(test/float32_consistency.jl:53-62: |F32 - F64| <= 3e-2 clear / 1.2e-1 cloudy for SW).

    python tools/f32_fuzz_diagnose.py 633 1053 1319
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import rrtmgp_jl_amd  # noqa: E402,F401
from oracle import oracle as O  # noqa: E402
from rrtmgp_jl_amd import rte, synthetic as S  # noqa: E402

SWN = ("flux_up", "flux_dn", "flux_net", "flux_dn_dir")


def case(seed, FT):
    """The inputs tests/test_gpu_fuzz.py draws for `seed` (same generator calls in the same order)."""
    rng = np.random.default_rng(1000 + seed)
    n_bnd = int(rng.integers(1, 6))
    gpb_lw = [int(x) for x in rng.choice([1, 3, 4, 8, 16, 20], n_bnd)]
    gpb_sw = [int(x) for x in rng.choice([2, 5, 8, 16, 24], n_bnd)]
    sw = S.make_gas_lookup("sw", FT, seed=seed, n_bnd=n_bnd, gpt_per_bnd=gpb_sw, n_minor_lower=(0, 8), n_minor_upper=(0, 5))
    cs = S.make_cloud_lookup("sw", n_bnd, FT, seed=seed)
    asw = S.make_aerosol_lookup("sw", sw.bnd_lims_wn, FT, seed=seed)
    ncol = int(rng.choice([1, 2, 7, 33, 130]))
    nlay = int(rng.choice([2, 3, 15, 16, 17, 31, 47, 63, 64, 65, 80, 127, 128, 129, 143, 192, 193]))
    clouds, aerosols = bool(rng.integers(0, 2)), bool(rng.integers(0, 2))
    vmr_kind = str(rng.choice(["gm", "full"]))
    lw_ngpt = sum(gpb_lw)
    as_, lb, sb = S.make_columns(ncol, nlay, FT, seed=seed, vmr_kind=vmr_kind, clouds=clouds, aerosols=aerosols,
                                 n_bnd_lw=n_bnd, n_bnd_sw=n_bnd, night_fraction=0.3, random_cld_frac=True,
                                 inc_flux_ngpt=lw_ngpt if rng.integers(0, 2) else 0)
    rng.integers(0, 2)   # (the metric draw is skipped: no metric scaling here)
    return as_, sb, sw, (cs if clouds else None), (asw if aerosols else None), dict(ncol=ncol, nlay=nlay, clouds=clouds, aerosols=aerosols)


def promote(x):
    return x._map(lambda a: a.astype(np.float64) if a.dtype == np.float32 else a)


def main(seeds):
    for seed in seeds:
        as32, sb32, sw32, cs32, a32, info = case(seed, np.float32)
        hip = rte.solve_sw(rte.TwoStreamSWRTE(info["ncol"], info["nlay"], np.float32, sb32), as32, sw32, cs32, a32, seed=7)
        o32 = O.solve_sw(as32, sb32, sw32, cs32, a32, seed=7)
        as64, sb64 = promote(as32), promote(sb32)
        sw64, cs64, a64 = sw32.astype(np.float64), cs32.astype(np.float64) if cs32 else None, a32.astype(np.float64) if a32 else None
        o64 = O.solve_sw(as64, sb64, sw64, cs64, a64, seed=7)
        d = lambda a, b: max(float(np.abs(np.float64(getattr(a, n)) - np.float64(getattr(b, n))).max()) for n in SWN)  # noqa: E731
        print(f"seed {seed} {info}: |HIP32 - O64| {d(hip, o64):.3e}   |O32 - O64| {d(o32, o64):.3e}   |HIP32 - O32| {d(hip, o32):.3e}"
              f"   max flux {float(np.abs(o64.flux_dn).max()):.0f}")


if __name__ == "__main__":
    main([int(x) for x in sys.argv[1:]] or [633, 1053, 1319])
