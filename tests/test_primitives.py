"""Accuracy of the device math forms of the per-g-point loops (rrtmgp.jl_amd/csrc/device.h), measured on the GPU against
Float64 through `rrtmgp_hip_eval_primitive` — the numbers behind DESIGN.md "Float32 numerics".

The reference computes with Julia's `exp` / `expm1` (< 1 ulp), IEEE `/` and `sqrt` (correctly rounded, 0.5 ulp).  Two builds:
  * libhip_rrtmgp.so (the shipped default since round 6): hand-written forms of THAT accuracy class — quotient, reciprocal and
    square root correctly rounded (<= 0.5 ulp up to ties closer than 2^-23 ulp), e^-x <= 1.2 ulp (measured 1.14: v_exp_f32's
    own error, ~0.65 ulp, plus the one rounding of the argument correction), 1 - e^-x <= 1.7 ulp;
  * libhip_rrtmgp_fast.so (-DRR_FAST_F32, `make fast`, opt-in): v_rcp_f32 / v_sqrt_f32 raw (1 ulp each, a quotient < 2 ulp),
    e^-x by __expf whose error grows with the argument (~4 ulp up to x = 4, ~x ulp beyond: only where e^-x itself is small —
    never more than 1.5 ulp OF ONE in absolute terms, which is what a flux sees), the quotients of increment_2stream
    correctly rounded in both builds.
Bounds asserted below are the measured maxima over 2e5 log-uniform samples plus a margin; the measured values are printed
(`pytest -s`) and recorded in profiles/r05_primitives_ulp.txt."""
import ctypes as C
import os

import numpy as np
import pytest

from rrtmgp_jl_amd import _lib

pytestmark = pytest.mark.gpu
N = 200_000


def _ulp_err(got, ref64):
    """|got - ref| in units of the Float32 spacing at ref."""
    ref32 = ref64.astype(np.float32)
    return np.abs(got.astype(np.float64) - ref64) / np.spacing(np.abs(ref32)).astype(np.float64)


def _samples(lo, hi, seed):
    r = np.random.default_rng(seed)
    return np.exp(r.uniform(np.log(lo), np.log(hi), N)).astype(np.float32)


@pytest.fixture(scope="module")
def fast():
    path = os.path.join(os.path.dirname(_lib.SO_PATH), "libhip_rrtmgp_fast.so")
    assert _lib.lib().rrtmgp_hip_build_flags() == b""   # (also loads torch's HIP runtime first, as for the shipped library)
    L = C.CDLL(path)
    L.rrtmgp_hip_build_flags.restype = C.c_char_p
    assert b"RR_FAST_F32" in L.rrtmgp_hip_build_flags()
    return L


CASES = {
    # name: (x range, y range or None, Float64 reference, max ulp shipped (IEEE-accurate) build, max ulp fast build)
    "exp_neg": ((1e-6, 80.0), None, lambda x, y: np.exp(-x), 1.25, None),
    "exp_pair_e1": ((1e-6, 80.0), None, lambda x, y: np.exp(-x), 1.25, None),
    "exp_pair_om1": ((1e-7, 60.0), None, lambda x, y: -np.expm1(-x), 1.7, 3.0),
    "rcp": ((1e-5, 1e5), None, lambda x, y: 1.0 / x, 0.501, 1.01),
    "div": ((1e-5, 1e5), (1e-5, 1e5), lambda x, y: x / y, 0.501, 2.01),
    "sqrt_pos": ((3e-4, 50.0), None, lambda x, y: np.sqrt(x), 0.501, 1.01),
    "ieee_div": ((1e-5, 1e5), (1e-5, 1e5), lambda x, y: x / y, 0.501, 0.501),
}


@pytest.mark.parametrize("name", list(CASES))
def test_shipped_build_forms_are_ieee_accurate(name):
    xr, yr, ref, bound, _ = CASES[name]
    x = _samples(*xr, seed=1)
    y = _samples(*yr, seed=2) if yr else None
    got = _lib.eval_primitive(name, x, y)
    err = _ulp_err(got, ref(x.astype(np.float64), None if y is None else y.astype(np.float64)))
    print(f"shipped {name}: max {err.max():.3f} ulp, mean {err.mean():.3f}")
    assert err.max() <= bound, (name, err.max(), x[err.argmax()])


@pytest.mark.parametrize("name", [n for n in CASES if CASES[n][4] is not None])
def test_fast_build_forms(fast, name):
    xr, yr, ref, _, bound = CASES[name]
    x = _samples(*xr, seed=1)
    y = _samples(*yr, seed=2) if yr else None
    got = _lib.eval_primitive(name, x, y, library=fast)
    err = _ulp_err(got, ref(x.astype(np.float64), None if y is None else y.astype(np.float64)))
    print(f"fast {name}: max {err.max():.3f} ulp, mean {err.mean():.3f}")
    assert err.max() <= bound, (name, err.max(), x[err.argmax()])


def test_fast_exp_error_grows_with_the_argument_only(fast):
    """__expf: the rounding of x log2(e) costs ~|x| ulp, i.e. nothing where e^-x matters."""
    for hi, bound in ((4.0, 5.0), (16.0, 18.0), (80.0, 90.0)):
        x = _samples(1e-6, hi, seed=3)
        err = _ulp_err(_lib.eval_primitive("exp_neg", x, library=fast), np.exp(-x.astype(np.float64)))
        print(f"fast exp_neg up to {hi}: max {err.max():.2f} ulp")
        assert err.max() <= bound
    # in absolute terms (what a flux sees): never more than 1.5 ulp of 1
    x = _samples(1e-6, 80.0, seed=4)
    d = np.abs(_lib.eval_primitive("exp_neg", x, library=fast).astype(np.float64) - np.exp(-x.astype(np.float64)))
    assert d.max() <= 1.5 * 2.0 ** -24


def test_nan_and_saturation(fast):
    x = np.array([np.nan, 250.0, 1e30, np.inf, 0.0], np.float32)
    for L in (None, fast):
        e = _lib.eval_primitive("exp_neg", x, library=L)
        assert np.isnan(e[0]) and (e[1:4] == 0).all() and e[4] == 1.0
    x64 = np.array([np.nan, 800.0, 1e300, np.inf, 0.0])
    e = _lib.eval_primitive("exp_neg", x64)
    assert np.isnan(e[0]) and (e[1:4] == 0).all() and e[4] == 1.0


def test_float64_forms():
    """Float64: exp_pair / m_exp_neg from one argument reduction (<= 2 ulp), Newton quotients (<= 1 ulp)."""
    r = np.random.default_rng(5)
    x = np.exp(r.uniform(np.log(1e-9), np.log(700.0), N))
    y = np.exp(r.uniform(np.log(1e-6), np.log(1e6), N))
    import math
    def ulp64(got, ref):   # ref in extended precision via numpy longdouble where available
        return np.abs(got - ref.astype(np.float64)) / np.spacing(np.abs(ref.astype(np.float64)))
    xl, yl = x.astype(np.longdouble), y.astype(np.longdouble)
    assert ulp64(_lib.eval_primitive("exp_neg", x), np.exp(-xl)).max() <= 2.0
    assert ulp64(_lib.eval_primitive("exp_pair_om1", x), -np.expm1(-xl)).max() <= 2.5
    assert ulp64(_lib.eval_primitive("div", x, y), xl / yl).max() <= 1.0
    assert ulp64(_lib.eval_primitive("rcp", y), 1 / yl).max() <= 1.0
    _ = math
