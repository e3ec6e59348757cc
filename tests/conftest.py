import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import rrtmgp_jl_amd  # noqa: E402,F401  (registers the `rrtmgp.jl_amd/` package)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `-m gpu` on the GPU box)")


@pytest.fixture(scope="session")
def tables64():
    """Full-size synthetic lookup tables (Float64), shared by the session."""
    import numpy as np
    from rrtmgp_jl_amd import synthetic as S
    lw, sw = S.make_gas_lookup("lw", np.float64), S.make_gas_lookup("sw", np.float64)
    return dict(lw=lw, sw=sw, cld_lw=S.make_cloud_lookup("lw", lw.n_bnd), cld_sw=S.make_cloud_lookup("sw", sw.n_bnd),
                aero_lw=S.make_aerosol_lookup("lw", lw.bnd_lims_wn), aero_sw=S.make_aerosol_lookup("sw", sw.bnd_lims_wn))


@pytest.fixture(scope="session")
def tables32(tables64):
    import numpy as np
    return {k: v.astype(np.float32) for k, v in tables64.items()}


@pytest.fixture(scope="session")
def small_tables64():
    """Reduced tables (3 bands of 8/4/12 g-points) for fast, exhaustive cases."""
    import numpy as np
    from rrtmgp_jl_amd import synthetic as S
    lw = S.make_gas_lookup("lw", np.float64, seed=7, n_bnd=3, gpt_per_bnd=[8, 4, 12])
    sw = S.make_gas_lookup("sw", np.float64, seed=7, n_bnd=3, gpt_per_bnd=[6, 10, 4])
    return dict(lw=lw, sw=sw, cld_lw=S.make_cloud_lookup("lw", 3, seed=7), cld_sw=S.make_cloud_lookup("sw", 3, seed=7),
                aero_lw=S.make_aerosol_lookup("lw", lw.bnd_lims_wn, seed=7),
                aero_sw=S.make_aerosol_lookup("sw", sw.bnd_lims_wn, seed=7))
