"""Contracts of the C ABI beyond numerics:
  * zero allocation per call once warm (update_fluxes.jl:215-218, test/standalone.jl:361-383): the library counts
    every hipMalloc / hipFree / hipHostRegister it makes (`rrtmgp_hip_allocation_counts`); 100 warm solves must not
    move the counters, for device-resident and for host-resident callers;
  * different workspaces may be driven from different host threads at the same time (include/rrtmgp_hip.h,
    SURVEY.md §8(b) "Threading") and give the single-thread bits."""
import threading

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from rrtmgp_jl_amd import _lib, rte, synthetic as S  # noqa: E402

LWN = ("flux_up", "flux_dn", "flux_net")
SWN = ("flux_up", "flux_dn", "flux_net", "flux_dn_dir")


@pytest.mark.parametrize("where", ["host", "host-pipelined", "device", "shards"])
def test_no_allocation_after_warm_up(tables32, where):
    t = tables32
    ncol = 16384 if where == "host-pipelined" else 96      # >= 16384 host columns take the chunked pipeline (host.h run_column_pipeline)
    nlay = 24
    as_, lb, sb = S.make_columns(ncol, nlay, np.float32, seed=1, aerosols=True, night_fraction=0.1)
    dev = [0, 0] if where == "shards" else 0
    fdev = None
    if where == "device":
        import torch
        d = torch.device("cuda", 0)
        as_, lb, sb, fdev = as_.to_device(d), lb.to_device(d), sb.to_device(d), d
    ws = rte.Workspace(ncol, nlay, np.float32, dev)
    lw = rte.TwoStreamLWRTE(ncol, nlay, np.float32, lb, workspace=ws, flux_device=fdev)
    sw = rte.TwoStreamSWRTE(ncol, nlay, np.float32, sb, workspace=ws, flux_device=fdev)
    dl = {k: rte.DeviceLookup(t[k], dev) for k in ("lw", "sw", "cld_lw", "cld_sw", "aero_lw", "aero_sw")}

    def step(seed):
        rte.solve_lw(lw, as_, dl["lw"], dl["cld_lw"], dl["aero_lw"], seed=seed)
        rte.solve_sw(sw, as_, dl["sw"], dl["cld_sw"], dl["aero_sw"], seed=seed)

    for i in range(2):
        step(i)
    ws.synchronize()
    before = _lib.allocation_counts()
    for i in range(100 if ncol < 1000 else 6):
        step(i)
    ws.synchronize()
    assert _lib.allocation_counts() == before, (before, _lib.allocation_counts())


def test_two_workspaces_from_two_host_threads(tables64):
    t = tables64
    cases = []
    for seed, ncol in ((1, 40), (2, 29)):
        as_, lb, sb = S.make_columns(ncol, 30, np.float64, seed=seed, random_cld_frac=True, night_fraction=0.2)
        cases.append((as_, lb, sb))
    dl = {k: rte.DeviceLookup(t[k], 0) for k in ("lw", "sw", "cld_lw", "cld_sw")}

    def run(case, reps, out):
        as_, lb, sb = case
        nlay, ncol = as_.dims
        ws = rte.Workspace(ncol, nlay, np.float64, 0)
        lw = rte.TwoStreamLWRTE(ncol, nlay, np.float64, lb, workspace=ws)
        sw = rte.TwoStreamSWRTE(ncol, nlay, np.float64, sb, workspace=ws)
        try:
            for _ in range(reps):
                f_lw = rte.solve_lw(lw, as_, dl["lw"], dl["cld_lw"], seed=7)
                f_sw = rte.solve_sw(sw, as_, dl["sw"], dl["cld_sw"], seed=7)
            out.append(({n: f_lw.as_nlev_ncol(n).copy() for n in LWN}, {n: f_sw.as_nlev_ncol(n).copy() for n in SWN}))
        except Exception as e:  # noqa: BLE001
            out.append(e)

    serial = []
    for c in cases:
        run(c, 1, serial)
    results = [[], []]
    threads = [threading.Thread(target=run, args=(cases[i], 25, results[i])) for i in range(2)]
    for th in threads:
        th.start()
    for th in threads:
        th.join()
    for i in range(2):
        assert len(results[i]) == 1 and not isinstance(results[i][0], Exception), results[i]
        for n in LWN:
            np.testing.assert_array_equal(results[i][0][0][n], serial[i][0][n])
        for n in SWN:
            np.testing.assert_array_equal(results[i][0][1][n], serial[i][1][n])


def test_only_large_host_arrays_are_page_locked(tables32):
    """hipHostRegister locks whole pages: only arrays of >= 32 MB (always mmapped: pages of their own) are registered —
    by the BINDING, through rrtmgp_hip_host_register, for the lifetime of the array (include/rrtmgp_hip.h); smaller ones
    share heap pages with other objects and stay pageable (the rule that ended the intermittent GPU memory faults of
    round 2).  The registration goes away with the array, not with the workspace."""
    import gc
    t = tables32
    nlay = 16
    regs = lambda: _lib.allocation_counts()[2]  # noqa: E731
    live = _lib.lib().rrtmgp_hip_host_registered_count
    # 96 columns: nothing is large enough
    as_, lb, sb = S.make_columns(96, nlay, np.float32, seed=1)
    r0 = regs()
    rte.solve_lw(rte.TwoStreamLWRTE(96, nlay, np.float32, lb), as_, t["lw"], t["cld_lw"])
    small = regs() - r0
    # 140 000 columns: layerdata (4, nlay, ncol) is 35.8 MB, every other array is below the floor
    ncol = 140_000
    as_, lb, sb = S.make_columns(ncol, nlay, np.float32, seed=1)
    assert as_.layerdata.nbytes >= 32 << 20 > as_.t_lev.nbytes
    slv = rte.TwoStreamLWRTE(ncol, nlay, np.float32, lb)
    r1, n1 = regs(), live()
    rte.solve_lw(slv, as_, t["lw"], t["cld_lw"])
    big = regs() - r1
    assert live() == n1 + 1
    rte.solve_lw(slv, as_, t["lw"], t["cld_lw"])
    again = regs() - r1 - big
    # (the first small solve of a workspace page-locks its own bounce buffer: counted with the registrations)
    assert small == 1 and big == 1 and again == 0, (small, big, again)
    del as_
    gc.collect()
    assert live() == n1      # released with the array (the workspace is still alive)


def test_registered_array_freed_and_reallocated_at_the_same_address(tables32):
    """A host model that rebinds a fresh same-size state array every step while keeping its workspace: glibc hands the
    same mmap hole out again, so the new array sits at the address of the old one.  With registrations tied to the
    array's lifetime the old one is gone before the new one is seen; every solve must read the NEW pages (a stale
    registration would silently keep the old physical pages mapped for the GPU)."""
    import gc
    t = tables32
    nlay, ncol = 16, 140_000
    base, lb, _ = S.make_columns(ncol, nlay, np.float32, seed=3)
    slv = rte.TwoStreamLWRTE(ncol, nlay, np.float32, lb)
    regs0 = _lib.allocation_counts()[2]
    addrs, olr = set(), []
    want = {}
    for it in range(200):
        import copy
        as_ = copy.copy(base)
        ld = np.array(base.layerdata, order="F", copy=True)       # a fresh 35.8 MB array (mmapped), registered on first use
        shift = float(it % 4)                                     # four different temperature profiles, cycling
        ld[2] += shift
        as_.layerdata = ld
        addrs.add(ld.ctypes.data)
        out = rte.solve_lw(slv, as_, t["lw"], t["cld_lw"], seed=5)
        up = out.as_nlev_ncol("flux_up")[-1, ::4096].copy()
        if it < 4:
            want[shift] = up
        else:
            np.testing.assert_array_equal(up, want[shift])        # the bits of the first solve with this profile
        del as_, ld, out
        gc.collect()
    assert not np.array_equal(want[0.0], want[1.0])
    assert len(addrs) < 200, "the allocator never reused an address: the test did not exercise the hazard"
    assert _lib.allocation_counts()[2] - regs0 >= 100             # re-registered with (nearly) every new array
    assert _lib.lib().rrtmgp_hip_host_registered_count() <= 2


def test_two_workspaces_share_large_registered_arrays_from_two_threads(tables32):
    """Two workspaces (LW and SW of one model) on two host threads, both reading the SAME >= 32 MB state arrays through the
    pipelined host path: neither may release or re-register what the other is copying from (the registry used
    to keep one `verified_by` slot per entry)."""
    t = tables32
    nlay, ncol = 16, 140_000
    as_, lb, sb = S.make_columns(ncol, nlay, np.float32, seed=2)
    dl = {k: rte.DeviceLookup(t[k], 0) for k in ("lw", "sw", "cld_lw", "cld_sw")}
    lw = rte.TwoStreamLWRTE(ncol, nlay, np.float32, lb)
    sw = rte.TwoStreamSWRTE(ncol, nlay, np.float32, sb)
    ref_lw = {n: rte.solve_lw(lw, as_, dl["lw"], dl["cld_lw"], seed=7).as_nlev_ncol(n).copy() for n in LWN}
    ref_sw = {n: rte.solve_sw(sw, as_, dl["sw"], dl["cld_sw"], seed=7).as_nlev_ncol(n).copy() for n in SWN}
    regs0 = _lib.allocation_counts()[2]
    errors = []

    def run_lw():
        try:
            for _ in range(12):
                out = rte.solve_lw(lw, as_, dl["lw"], dl["cld_lw"], seed=7)
                for n in LWN:
                    np.testing.assert_array_equal(out.as_nlev_ncol(n), ref_lw[n])
        except Exception as e:  # noqa: BLE001
            errors.append(e)

    def run_sw():
        try:
            for _ in range(12):
                out = rte.solve_sw(sw, as_, dl["sw"], dl["cld_sw"], seed=7)
                for n in SWN:
                    np.testing.assert_array_equal(out.as_nlev_ncol(n), ref_sw[n])
        except Exception as e:  # noqa: BLE001
            errors.append(e)

    threads = [threading.Thread(target=run_lw), threading.Thread(target=run_sw)]
    for th in threads:
        th.start()
    for th in threads:
        th.join()
    assert not errors, errors
    assert _lib.allocation_counts()[2] == regs0        # nothing was re-registered along the way


def test_host_register_contract():
    """rrtmgp_hip_host_register / _unregister: reference counted per exact range, overlapping ranges of other extents and
    unknown pointers are refused."""
    L = _lib.lib()
    a = np.zeros(40 << 20, dtype=np.uint8)
    p, n0 = a.ctypes.data, L.rrtmgp_hip_host_registered_count()
    assert L.rrtmgp_hip_host_register(p, a.nbytes) == 0
    assert L.rrtmgp_hip_host_register(p, a.nbytes) == 0            # second owner of the same range
    assert L.rrtmgp_hip_host_registered_count() == n0 + 1
    assert L.rrtmgp_hip_host_register(p + 4096, a.nbytes - 8192) != 0 and "overlaps" in _lib.last_error()
    assert L.rrtmgp_hip_host_unregister(p) == 0
    assert L.rrtmgp_hip_host_registered_count() == n0 + 1
    assert L.rrtmgp_hip_host_unregister(p) == 0
    assert L.rrtmgp_hip_host_registered_count() == n0
    assert L.rrtmgp_hip_host_unregister(p) != 0 and "not a registered range" in _lib.last_error()
    assert L.rrtmgp_hip_host_register(None, 16) != 0
