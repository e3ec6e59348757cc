"""Golden fixtures (tests/golden/*.npz, made by tests/golden/make_golden.py) and the
cross-check of the two independent CPU restatements (C oracle vs numpy)."""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "golden"))
import make_golden as G  # noqa: E402
from oracle import np_oracle as NP  # noqa: E402
from oracle import oracle as O  # noqa: E402

HERE = os.path.join(os.path.dirname(__file__), "golden")


@pytest.mark.parametrize("case", list(G.CASES))
def test_c_oracle_reproduces_golden(case):
    exp = np.load(os.path.join(HERE, f"{case}.npz"))
    got = G.run(case, O.solve_lw, O.solve_sw)
    assert set(exp.files) == set(got)
    for k in exp.files:
        np.testing.assert_allclose(got[k], exp[k], rtol=1e-12, atol=1e-12, err_msg=k)


@pytest.mark.parametrize("case", ["clear_gm", "cloudy_full"])
def test_numpy_restatement_agrees_with_golden(case):
    """Two transcriptions of the reference that share no code must agree to rounding."""
    exp = np.load(os.path.join(HERE, f"{case}.npz"))
    t = G.tables()
    as_, lb, sb = G.inputs(case)
    cl = t["cld_lw"] if as_.cloud_state is not None else None
    cs = t["cld_sw"] if as_.cloud_state is not None else None
    up, dn = NP.solve_lw_2stream(t["lw"], as_, lb, cl)
    np.testing.assert_allclose(up, exp["lw2s_up"], rtol=1e-11, atol=1e-10)
    np.testing.assert_allclose(dn, exp["lw2s_dn"], rtol=1e-11, atol=1e-10)
    up, dn = NP.solve_lw_noscat(t["lw"], as_, lb, cl)
    np.testing.assert_allclose(up, exp["lwns_up"], rtol=1e-11, atol=1e-10)
    np.testing.assert_allclose(dn, exp["lwns_dn"], rtol=1e-11, atol=1e-10)
    # three Gauss-Jacobi-5 angles (literal table of src/optics/AngularDiscretizations.jl:47-49): the flux is the
    # weighted sum of the one-angle solves (longwave_noscat.jl:45-96)
    mu3 = (0.1024922169, 0.4417960320, 0.8633751621)
    w3 = (0.0437820218, 0.3875796738, 0.5686383044)
    up = dn = 0.0
    for mu, w in zip(mu3, w3):
        u, d = NP.solve_lw_noscat(t["lw"], as_, lb, cl, Ds=1.0 / mu, w=w)
        up, dn = up + u, dn + d
    np.testing.assert_allclose(up, exp["lwns3_up"], rtol=1e-11, atol=1e-10)
    np.testing.assert_allclose(dn, exp["lwns3_dn"], rtol=1e-11, atol=1e-10)
    up, dn, dr = NP.solve_sw_2stream(t["sw"], as_, sb, cs)
    np.testing.assert_allclose(up, exp["sw_up"], rtol=1e-11, atol=1e-10)
    np.testing.assert_allclose(dn, exp["sw_dn"], rtol=1e-11, atol=1e-10)
    np.testing.assert_allclose(dr, exp["sw_dir"], rtol=1e-11, atol=1e-10)


@pytest.mark.gpu
@pytest.mark.parametrize("case", list(G.CASES))
def test_hip_reproduces_golden(case):
    from rrtmgp_jl_amd import rte

    def lw(as_, bcs, lk, cld, aero, twostream, seed, n_gauss_angles=1):
        nlay, ncol = as_.dims
        cls = rte.TwoStreamLWRTE if twostream else rte.NoScatLWRTE
        return rte.solve_lw(cls(ncol, nlay, as_.dtype, bcs, n_gauss_angles=n_gauss_angles), as_, lk, cld, aero, seed=seed)

    def sw(as_, bcs, lk, cld, aero, seed):
        nlay, ncol = as_.dims
        return rte.solve_sw(rte.TwoStreamSWRTE(ncol, nlay, as_.dtype, bcs), as_, lk, cld, aero, seed=seed)
    exp = np.load(os.path.join(HERE, f"{case}.npz"))
    got = G.run(case, lw, sw)
    for k in exp.files:
        np.testing.assert_allclose(got[k], exp[k], rtol=1e-10, atol=1e-8, err_msg=k)
