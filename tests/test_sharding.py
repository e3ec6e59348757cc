"""Multi-GPU path on CPU: one process per rank over `gloo`, world_size 2.

Columns shard embarrassingly; what has to be right is the sharding and
gather plumbing and that a column's result does not depend on how columns are split
(the McICA stream is keyed by the GLOBAL column index).  The compute function injected
here is the CPU oracle (allowed in tests only); on GPUs it is the HIP solve.
"""
import os
import socket

import numpy as np
import pytest

from rrtmgp_jl_amd import sharding, synthetic as S


def test_shard_ranges_cover_and_balance():
    for ncol, world in [(10, 3), (7, 8), (4096, 8), (1, 2), (131072 * 8, 8)]:
        rs = [sharding.shard_range(ncol, r, world) for r in range(world)]
        assert rs[0][0] == 0 and rs[-1][1] == ncol
        assert all(rs[i][1] == rs[i + 1][0] for i in range(world - 1))
        w = [hi - lo for lo, hi in rs]
        assert max(w) - min(w) <= 1


def test_shard_container_slices_every_column_array():
    as_, lb, sb = S.make_columns(19, 6, np.float64, seed=3, aerosols=True, inc_flux_ngpt=24, n_bnd_lw=3, n_bnd_sw=3)
    sh = sharding.shard_container(as_, 5, 12, 19)
    assert sh.dims == (6, 7)
    np.testing.assert_array_equal(sh.layerdata, as_.layerdata[:, :, 5:12])
    np.testing.assert_array_equal(sh.vmr.vmr_h2o, as_.vmr.vmr_h2o[:, 5:12])
    assert sh.vmr.vmr.shape == as_.vmr.vmr.shape  # well-mixed vector (length 19 == ncol here) is NOT sliced
    np.testing.assert_array_equal(sh.cloud_state.cld_frac, as_.cloud_state.cld_frac[:, 5:12])
    np.testing.assert_array_equal(sh.aerosol_state.aero_mass, as_.aerosol_state.aero_mass[:, :, 5:12])
    assert sh.t_sfc.shape == (7,) and sh.lat.shape == (7,)
    lbs = sharding.shard_container(lb, 5, 12, 19)
    np.testing.assert_array_equal(lbs.inc_flux, lb.inc_flux[5:12])  # (ncol, ngpt): column index first
    np.testing.assert_array_equal(lbs.sfc_emis, lb.sfc_emis[:, 5:12])
    sbs = sharding.shard_container(sb, 5, 12, 19)
    assert sbs.cos_zenith.shape == (7,) and sbs.sfc_alb_direct.shape == (3, 7)


def _worker(rank, world, port, q):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle import oracle as O
        ncol, nlay = 11, 12
        lw = S.make_gas_lookup("lw", np.float64, seed=7, n_bnd=3, gpt_per_bnd=[8, 4, 12])
        sw = S.make_gas_lookup("sw", np.float64, seed=7, n_bnd=3, gpt_per_bnd=[6, 10, 4])
        cl, cs = S.make_cloud_lookup("lw", 3, seed=7), S.make_cloud_lookup("sw", 3, seed=7)
        as_, lb, sb = S.make_columns(ncol, nlay, np.float64, seed=5, random_cld_frac=True, n_bnd_lw=3, n_bnd_sw=3,
                                     night_fraction=0.2)
        f_lw, (lo, hi) = sharding.solve_sharded(lambda a, b, col_offset: O.solve_lw(a, b, lw, cl, seed=9, col_offset=col_offset),
                                                as_, lb, ncol, rank, world)
        f_sw, _ = sharding.solve_sharded(lambda a, b, col_offset: O.solve_sw(a, b, sw, cs, seed=9, col_offset=col_offset),
                                         as_, sb, ncol, rank, world)
        assert f_lw.flux_up.shape == (nlay + 1, hi - lo)
        g_lw = sharding.gather_columns(f_lw.flux_up, ncol)
        g_sw = sharding.gather_columns(f_sw.flux_dn, ncol)
        ref_lw = O.solve_lw(as_, lb, lw, cl, seed=9).flux_up
        ref_sw = O.solve_sw(as_, sb, sw, cs, seed=9).flux_dn
        ok = bool(np.array_equal(g_lw, ref_lw) and np.array_equal(g_sw, ref_sw))
        # timing agreement used by bench.py: MAX over ranks
        import torch
        t = torch.tensor([float(rank + 1)], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        q.put((rank, ok, float(t.item())))
    finally:
        dist.destroy_process_group()


def test_world_size_2_gloo_matches_unsharded():
    import torch.multiprocessing as mp
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=240) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert res == [(0, True, 2.0), (1, True, 2.0)]


@pytest.mark.gpu
def test_bench_multi_process_path_on_one_gpu():
    """bench.py as the driver launches it for N > 1 (torch.distributed.run, one rank per GPU), exercised with two
    ranks SHARING the only GPU of the test box over gloo (bench.py test hook): barrier, MAX over ranks, one JSON
    line from rank 0 with the whole-job column count."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, RRTMGP_BENCH_BACKEND="gloo", RRTMGP_BENCH_SHARE_GPU="1")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                        "--master-addr", "127.0.0.1", "--master-port", "29533", os.path.join(root, "bench.py"),
                        "--gpus", "2", "--steps", "2", "--warmup", "1", "--ncol", "4096", "--cpu-sample", "0"],
                       capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["unit"] == "columns/s" and d["steps"] == 2
    assert d["config"]["ncol_per_gpu"] == 4096
    assert abs(d["value"] - 2 * 4096 * 2 / (d["ms_per_step"] * 2e-3)) / d["value"] < 1e-6
    # the audit record of a scaling run: one entry per rank with its device ordinal, name, PCI address and pid (two ranks
    # that shared a GPU, as here, show the SAME pci address and different pids: exactly what a reader must be able to see)
    rd = d["rank_devices"]
    assert [x["rank"] for x in rd] == [0, 1] and len({x["pid"] for x in rd}) == 2
    assert all(x["name"] and len(x["pci"].split(":")) == 3 for x in rd) and rd[0]["pci"] == rd[1]["pci"]
    assert d["ranks_seen"] == 2 and len(d["rank_ms_per_step"]) == 2


@pytest.mark.gpu
def test_bench_bare_command_starts_its_own_ranks():
    """`python bench.py --gpus 2 ...` with NO WORLD_SIZE in the environment (how the driver starts the N = 1 run): the
    script launches its own two ranks under torch.distributed.run on 127.0.0.1 and rank 0 prints the one JSON line, with
    the world size the process group reported and every rank's own time."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(RRTMGP_BENCH_BACKEND="gloo", RRTMGP_BENCH_SHARE_GPU="1")
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
                        "--ncol", "4096", "--cpu-sample", "0"], capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["ranks_seen"] == 2 and len(d["rank_ms_per_step"]) == 2 and d["rank_columns"] == [4096, 4096]
    assert max(d["rank_ms_per_step"]) <= d["ms_per_step"] * (1 + 1e-9)
    assert abs(d["value"] - 2 * 4096 * 2 / (d["ms_per_step"] * 2e-3)) / d["value"] < 1e-6


def _worker8(rank, world, port, q, ncol):
    """One rank of a world-size-8 job on BASELINE config 4's column counts: its contiguous range through the oracle, the host
    gather, bench.py's agreement on the step time (MAX over ranks) and its per-rank audit record."""
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle import oracle as O
        nlay = 6
        sw = S.make_gas_lookup("sw", np.float64, seed=7, n_bnd=2, gpt_per_bnd=[4, 4])
        cs = S.make_cloud_lookup("sw", 2, seed=7)
        as_, _, sb = S.make_columns(ncol, nlay, np.float64, seed=5, random_cld_frac=True, n_bnd_lw=2, n_bnd_sw=2, night_fraction=0.2)
        f_sw, (lo, hi) = sharding.solve_sharded(lambda a, b, col_offset: O.solve_sw(a, b, sw, cs, seed=9, col_offset=col_offset),
                                                as_, sb, ncol, rank, world)
        g = sharding.gather_columns(f_sw.flux_dn, ncol)
        ok = True
        if rank == 0:   # the unsharded solve, once
            ok = bool(np.array_equal(g, O.solve_sw(as_, sb, sw, cs, seed=9).flux_dn))
        every = [None] * world
        dist.all_gather_object(every, {"rank": rank, "columns": hi - lo, "lo": lo})
        t = torch.tensor([1.0 + rank], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        q.put((rank, ok, float(t.item()), (lo, hi), [e["columns"] for e in every]))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("ncol", [4096, 4097])
def test_world_size_8_gloo_on_config4_column_counts(ncol):
    """BASELINE config 4 strong-scales 4096 columns over 8 GPUs: 512-column ranges; 4097 gives the ragged case (513 + 7 x 512).
    Eight gloo ranks on CPU: ranges cover the job, the gathered fluxes equal the unsharded solve bit for bit (the McICA stream
    is keyed by the global column), every rank sees every rank's column count, MAX over ranks is the last rank's time."""
    import torch.multiprocessing as mp
    world = 8
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker8, args=(r, world, port, q, ncol)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=600) for _ in procs)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    widths = [512 + (1 if r < ncol - 4096 else 0) for r in range(world)]
    lo = 0
    for r, (rank, ok, tmax, rng, counts) in enumerate(res):
        assert rank == r and ok and tmax == 8.0 and counts == widths
        assert rng == (lo, lo + widths[r])
        lo += widths[r]
    assert lo == ncol


def test_bench_strong_scaling_ranges_are_the_sharding_ranges():
    """bench.py --scaling strong hands rank r the range of rrtmgp.jl_amd/sharding.py (what the library's multi-device
    workspace uses too): 8 ranks on config 4's 4096 columns, ragged 4097, fewer columns than ranks."""
    import importlib.util
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("bench_mod2", os.path.join(root, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    assert [bench.rank_columns(4096, r, 8) for r in range(8)] == [(512 * r, 512 * (r + 1)) for r in range(8)]
    rs = [bench.rank_columns(4097, r, 8) for r in range(8)]
    assert rs[0] == (0, 513) and rs[1] == (513, 1025) and rs[-1] == (3585, 4097)
    assert [hi - lo for lo, hi in (bench.rank_columns(5, r, 8) for r in range(8))] == [1, 1, 1, 1, 1, 0, 0, 0]   # bench.py refuses the empty ranks


def test_bench_self_launch_command(monkeypatch):
    """The re-exec of a bare `--gpus N` command: torch.distributed.run, one node, N ranks, loopback rendezvous, the
    original arguments unchanged (no GPU needed: the child process is not started)."""
    import importlib.util
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(root, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    seen = {}

    def fake_run(cmd, env=None, **kw):
        seen["cmd"], seen["env"] = cmd, env
        return subprocess.CompletedProcess(cmd, 7)
    monkeypatch.setattr(bench.subprocess, "run", fake_run)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "5", "--warmup", "2"])
    assert bench.self_launch(4) == 7
    cmd = seen["cmd"]
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"] and "--nproc-per-node=4" in cmd and "--nnodes=1" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and int(cmd[cmd.index("--master-port") + 1]) > 0
    assert cmd[-6:] == ["--gpus", "4", "--steps", "5", "--warmup", "2"] and cmd[-7].endswith("bench.py")
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"


@pytest.mark.gpu
def test_bench_single_process_fan_out_on_one_gpu():
    """`bench.py --gpus 2 --single-process`: ONE process, host arrays of 2 x ncol columns, the library's multi-device
    workspace fans out (device ids wrap onto the only GPU of the test box).  What a Julia host gets from
    HIPDevice([0, 1])."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--single-process", "--steps", "2",
                        "--warmup", "1", "--ncol", "4096", "--cpu-sample", "0", "--no-legs"],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    d = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert d["n_gpus"] == 2 and d["config"]["ncol_per_gpu"] == 4096 and "ONE host process" in d["config"]["parallelism"]
    assert abs(d["value"] - 2 * 4096 * 2 / (d["ms_per_step"] * 2e-3)) / d["value"] < 1e-6


@pytest.mark.gpu
def test_bench_single_process_eight_shards_on_one_gpu():
    """`bench.py --gpus 8 --single-process`: the library's fan-out with EIGHT shards (eight parked host threads, eight
    streams, replicated lookups), all wrapped onto the one GPU of the test box: the path `HIPDevice(0:7)` takes."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "8", "--single-process", "--steps", "2",
                        "--warmup", "1", "--ncol", "512", "--nlay", "72", "--aerosols", "--cpu-sample", "0", "--no-legs"],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    d = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert d["n_gpus"] == 8 and d["config"]["ncol_per_gpu"] == 512 and "8 shards in one process" in d["config"]["workload"]
    assert abs(d["value"] - 8 * 512 * 2 / (d["ms_per_step"] * 2e-3)) / d["value"] < 1e-6


@pytest.mark.gpu
def test_bench_strong_scaling_mode_on_one_gpu():
    """BASELINE config 4 is a STRONG-scaling case (4096 columns on 1 -> 8 GPUs): `--scaling strong --ncol-total N` splits the
    job's columns over the ranks in contiguous ranges; value = total columns per second.  Two ranks share the one GPU of
    the test box over gloo (bench.py test hook); 4097 columns so that the ranges are ragged (2048 + 2049)."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, RRTMGP_BENCH_BACKEND="gloo", RRTMGP_BENCH_SHARE_GPU="1")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                        "--master-addr", "127.0.0.1", "--master-port", "29541", os.path.join(root, "bench.py"),
                        "--gpus", "2", "--steps", "3", "--warmup", "1", "--scaling", "strong", "--ncol-total", "4097",
                        "--nlay", "72", "--aerosols", "--cpu-sample", "0"],
                       capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "strong" and "4097 columns split over 2 GPU(s)" in d["config"]["workload"]
    assert abs(d["value"] - 4097 * 3 / (d["ms_per_step"] * 3e-3)) / d["value"] < 1e-6


@pytest.mark.gpu
def test_bench_layer2_host_leg():
    """`bench.py --l2 fused|split`: the Layer-2 step of a host model on host arrays through the Python mirror; the record
    carries the bytes that crossed PCIe per column, and the fused step uploads the state once."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    rec = {}
    for mode in ("fused", "split"):
        r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--l2", mode, "--leg", "x", "--ncol", "20480", "--nlay", "32",
                            "--steps", "2", "--warmup", "1"], capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-1500:]
        rec[mode] = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
        assert rec[mode]["unit"] == "columns/s" and rec[mode]["value"] > 0 and rec[mode]["ncol"] == 20480
    assert rec["fused"]["calls_per_step"] == 1 and rec["split"]["calls_per_step"] == 3
    assert rec["fused"]["h2d_bytes_per_column"] < 0.55 * rec["split"]["h2d_bytes_per_column"]
    # state (4 + 2 + 5 layer arrays, 2 level arrays, t_sfc, lat) + boundary conditions, Float32, 32 layers, 16 / 14 bands
    want = 4 * ((4 + 2 + 5) * 32 + 2 * 33 + 2 + 16 + 2 + 2 * 14)
    assert want <= rec["fused"]["h2d_bytes_per_column"] <= want + 8


def test_local_cpus_of_a_gpu_from_a_fake_sysfs_tree(tmp_path, monkeypatch):
    """The NUMA binding of the shard workers (csrc/multi.hip) is host logic: which CPUs the kernel lists next to a GPU's
    PCIe root.  A one-GPU box never shows the multi-socket case, so it is driven here against a fake sysfs tree:
    two sockets, range lists with several pieces, single CPUs, upper-case bus ids as hipDeviceGetPCIBusId prints them,
    a missing device, an empty and a malformed list."""
    import ctypes as C
    from rrtmgp_jl_amd import _lib
    L = _lib.lib()
    root = tmp_path / "sys"
    lists = {"0000:05:00.0": "0-3,64-67\n", "0000:c5:00.0": "32-35,96,98-99\n", "0000:e5:00.0": "\n",
             "0000:f5:00.0": "7,x,9-8,12-13\n"}
    for dev, text in lists.items():
        d = root / "bus" / "pci" / "devices" / dev
        d.mkdir(parents=True)
        (d / "local_cpulist").write_text(text)
    monkeypatch.setenv("RRTMGP_HIP_SYSFS_ROOT", str(root))

    def cpus(bus):
        buf = (C.c_int32 * 64)()
        n = L.rrtmgp_hip_local_cpus(bus.encode(), buf, 64)
        return n, list(buf[:max(n, 0)])
    assert cpus("0000:05:00.0") == (8, [0, 1, 2, 3, 64, 65, 66, 67])
    assert cpus("0000:C5:00.0") == (7, [32, 33, 34, 35, 96, 98, 99])     # upper-case id, as the HIP runtime prints it
    assert cpus("0000:e5:00.0")[0] == 0                                    # an empty list binds nothing
    assert cpus("0000:f5:00.0") == (3, [7, 12, 13])                        # junk and inverted ranges are skipped
    assert cpus("0000:aa:00.0")[0] < 0 and "local_cpulist" in _lib.last_error()
    n = L.rrtmgp_hip_local_cpus(b"0000:05:00.0", None, 0)                  # counting only
    assert n == 8
    assert L.rrtmgp_hip_local_cpus(None, None, 0) < 0
