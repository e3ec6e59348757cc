"""Error behaviour of the C ABI (include/rrtmgp_hip.h): bad arguments give a negative status and a
message through rrtmgp_hip_last_error, never a crash or a silent wrong answer; limits documented in
INTEGRATION.md §4 are enforced.  The reference raises Julia errors at the same places
(constructor checks of src/api/solver.jl:159-190, dimension asserts of src/rte/RTE.jl)."""
import ctypes as C

import numpy as np
import pytest

from rrtmgp_jl_amd import _abi, synthetic as S

pytestmark = pytest.mark.gpu


def test_dimension_and_lookup_mismatches(tables64, small_tables64):
    from rrtmgp_jl_amd import _lib, rte
    t = tables64
    as_, lb, sb = S.make_columns(5, 16, seed=1)
    # band count of the boundary conditions must match the lookup: cloud lookup of another band count
    with pytest.raises(_lib.RRTMGPHipError, match="band count"):
        rte.solve_lw(rte.TwoStreamLWRTE(5, 16, np.float64, lb), as_, t["lw"], small_tables64["cld_lw"])
    # a longwave solve given a shortwave table
    with pytest.raises(_lib.RRTMGPHipError):
        rte.solve_lw(rte.TwoStreamLWRTE(5, 16, np.float64, lb), as_, t["sw"])
    # state precision differs from the workspace
    as32, lb32, _ = S.make_columns(5, 16, np.float32, seed=1)
    with pytest.raises(TypeError, match="float32"):
        rte.solve_lw(rte.TwoStreamLWRTE(5, 16, np.float64, lb), as32, t["lw"])
    with pytest.raises(TypeError, match="lookup tables"):
        rte.solve_lw(rte.TwoStreamLWRTE(5, 16, np.float64, lb), as_, t["lw"].astype(np.float32))
    # workspace smaller than the state
    with pytest.raises(ValueError, match="workspace was created"):
        rte.solve_lw(rte.TwoStreamLWRTE(4, 16, np.float64, lb), as_, t["lw"])
    # n_gauss_angles outside 1..4 (AngularDiscretizations.jl:34-63)
    with pytest.raises(_lib.RRTMGPHipError, match="n_gauss_angles"):
        rte.solve_lw(rte.NoScatLWRTE(5, 16, np.float64, lb, n_gauss_angles=5), as_, t["lw"])
    # the one size limit left: a column's records must fit the 160 KB LDS of a CU (Float64: ~250 layers)
    big, lbb, _ = S.make_columns(2, 400, seed=1)
    with pytest.raises(_lib.RRTMGPHipError, match="160 KB LDS"):
        rte.solve_lw(rte.TwoStreamLWRTE(2, 400, np.float64, lbb), big, t["lw"], t["cld_lw"])
    # the error text is available through the C entry point as well
    buf = C.create_string_buffer(256)
    assert _lib.lib().rrtmgp_hip_last_error(buf, 256) == 0


def test_null_and_missing_arrays():
    from rrtmgp_jl_amd import _lib
    L = _lib.lib()
    assert L.rrtmgp_hip_workspace_destroy(None) in (0, _abi.EINVAL if hasattr(_abi, "EINVAL") else -1)
    ws = C.c_void_p()
    assert L.rrtmgp_hip_workspace_create(0, 4, 8, _abi.F64, C.byref(ws)) == 0
    d = _abi.AtmosState()          # all pointers NULL
    d.ncol, d.nlay, d.mem = 4, 8, _abi.MEM_HOST
    b, f, o = _abi.LwBcs(), _abi.FluxOut(), _abi.SolveOpts()
    rc = L.rrtmgp_hip_rte_lw_2stream_solve(ws, None, None, None, C.byref(d), C.byref(b), C.byref(f), C.byref(o))
    assert rc < 0
    assert L.rrtmgp_hip_workspace_create(0, 4, 8, 3, C.byref(C.c_void_p())) < 0      # unknown precision
    assert L.rrtmgp_hip_workspace_create(99, 4, 8, _abi.F64, C.byref(C.c_void_p())) < 0   # no such device
    assert L.rrtmgp_hip_workspace_destroy(ws) == 0


def test_single_column_and_single_layer_edge_sizes(small_tables64):
    """ncol = 1 (BASELINE configs[0]) and the smallest grids the solvers accept."""
    from rrtmgp_jl_amd import rte
    from oracle import oracle
    t = small_tables64
    from rrtmgp_jl_amd import _lib
    with pytest.raises(_lib.RRTMGPHipError, match="dimensions"):   # a single layer is not a column (documented limit)
        rte.Workspace(3, 1, np.float64)
    for ncol, nlay in ((1, 60), (1, 2), (100, 3)):
        as_, lb, sb = S.make_columns(ncol, nlay, seed=9, n_bnd_lw=3, n_bnd_sw=3, night_fraction=0.0)
        f = rte.solve_lw(rte.TwoStreamLWRTE(ncol, nlay, np.float64, lb), as_, t["lw"], t["cld_lw"], seed=1)
        r = oracle.solve_lw(as_, lb, t["lw"], t["cld_lw"], seed=1)
        assert np.abs(f.flux_up - r.flux_up).max() < 1e-9 and np.abs(f.flux_dn - r.flux_dn).max() < 1e-9
        f = rte.solve_sw(rte.TwoStreamSWRTE(ncol, nlay, np.float64, sb), as_, t["sw"], t["cld_sw"], seed=1)
        r = oracle.solve_sw(as_, sb, t["sw"], t["cld_sw"], seed=1)
        assert np.abs(f.flux_up - r.flux_up).max() < 1e-9 and np.abs(f.flux_dn_dir - r.flux_dn_dir).max() < 1e-9
