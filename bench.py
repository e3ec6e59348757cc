#!/usr/bin/env python3
"""Benchmark of the hot path: all-sky LW + SW two-stream radiative transfer, columns/sec.

    python bench.py --gpus N --steps K --warmup W

One process per GPU (for N > 1 the driver launches this file under
torch.distributed.run; RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* come from the env).
Columns shard embarrassingly: every rank owns a contiguous range of global columns and
runs the same per-GPU workload (weak scaling); there is no collective on the data path.
A "step" is one `solve_lw` + `solve_sw` over the rank's batch with the state already
resident in HBM (torch tensors handed to the C ABI as device pointers).

Prints ONE JSON line on rank 0.  Besides the contract's keys:
  * `step_ms`: min / median / mean of the K timed steps (HIP events on the launch stream, read after the
    timed region; the reference quotes BenchmarkTools' min for its ratchet and the median in its tables);
  * `roofline`: the ceiling that binds — the FP32 vector rate (the path has no contraction and runs HBM at < 1 % of peak,
    SURVEY.md F8): the DOMINANT kernel's algorithmic flops over its live event time against 157.3 TFLOP/s, per-kernel
    fractions under `per_kernel`; the HBM view of the task contract (algorithmic bytes against 8 TB/s, PMC traffic and its
    ratio to the algorithmic bytes) is nested under `hbm`.  `traffic` comes from the committed rocprofv3 PMC summary and
    carries the git SHA it was taken at (`traffic_source`; null when the summary is missing or was taken on other kernel
    sources — compared by a hash of csrc/{common.h, device.h, solve_lw.hip, solve_sw.hip, Makefile}); `valu`: both kernels;
  * `strong_emulated`: BASELINE config 4's shards (4096 / 2048 / 1024 / 512 columns x 72, aerosols) each solved alone on
    this GPU: the one-GPU prediction of the 1 -> 8 strong-scaling curve;
  * `rank_devices`: ordinal / name / PCI address of the device every rank ran on;
  * `host_end_to_end`: the same workload handed over as HOST arrays (library stages H2D / D2H over PCIe
    every step): never `value`;
  * `variants`: Float64, MERRA aerosols, one-pass clear-sky diagnostic, 1 048 576 columns on one GPU, `gcm_mix` (half the
    columns at night, fractional cloud fractions), `fast_f32` (the opt-in raw-instruction Float32 build, `make fast`; the
    headline library is the IEEE-accurate one: correctly rounded div / sqrt, exp <= 1.2 ulp); a leg whose kernels have a
    committed rocprofv3 summary on the current kernel sources (profiles/latest_<leg>.json) carries a `profiled` block;
  * `a100_shape`: the reference's own benchmark grid (86 400 columns x 63 layers, docs/src/howto/gpu.md:116-150,
    perf/benchmark_baselines/nvidia_a100_sxm4_40gb.txt) for Float32 / Float64 x clear / all-sky / all-sky + aerosols, LW and
    SW kernel times separately, next to the published A100 numbers (context, never `value`);
  * `cpu_baseline`: the plain-C oracle on a bounded sample of the same workload on this box's host cores.
The extra legs run as short child processes after the timed region (rank 0, N = 1 only; `--no-legs` skips them).
"""
import argparse
import json
import os
import statistics
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

NLAY = 64
NCOL_PER_GPU = 131072          # BASELINE.json configs[4]: 1,048,576 columns over 8 GPUs
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: 8 TB/s spec
VALU_PEAK_TFLOPS = 157.3       # FP32 vector peak
# algorithmic flops per (layer, g-point) cell, n_m = 3 minor contributors
LW_FLOPS_PER_CELL, SW_FLOPS_PER_CELL = 301.0, 345.0
# no-scattering LW (rte_lw_noscat_one_angle!, src/rte/longwave_noscat.jl:171-301), same counting convention: the optics and
# sources of the two-stream figure without lw_2stream_coeffs (2 x 53) and the adding step (22) = 173, plus per quadrature
# angle and direction tau*Ds, the `fact` branch, the source and the transport + accumulation: 14 flops (+ 1 exp, 1 div)
LW_NOSCAT_FLOPS_PER_CELL, LW_NOSCAT_FLOPS_PER_ANGLE = 173.0, 28.0
FP64_VALU_PEAK_TFLOPS = 78.6   # MI355X FP64 vector peak (datasheet; half the FP32 vector rate of the guide's table)


def algorithmic_bytes(nlay, nbnd_lw, nbnd_sw, ft_bytes):
    """Compulsory HBM bytes per column per kernel launch (VmrGM, clouds, no aerosols)."""
    nlev = nlay + 1
    lw = ((4 + 2 + 5) * nlay + nlev + 1 + nbnd_lw) + (3 * nlev + 1)
    sw = ((4 + 2 + 5) * nlay + 2 * nbnd_sw + 2) + (4 * nlev + 1)
    # one step with the state read once by both solves (1275 elements at nlay = 64)
    step = ((4 + 2 + 5) * nlay + nlev + 1 + nbnd_lw + 2 * nbnd_sw + 2) + (7 * nlev + 4)
    return lw * ft_bytes, sw * ft_bytes, step * ft_bytes


def kernel_valu(kernel, kernel_ms, ncol, nlay, n_gpt, peak=VALU_PEAK_TFLOPS, lw_cell=LW_FLOPS_PER_CELL):
    """A column kernel's algorithmic FP32 rate and its fraction of the vector peak: SURVEY.md section 8(d) flops per
    (layer, g-point) cell x the cells one launch processes / its duration.  The same function prices the live event time
    (`roofline.frac`) and the committed profile's average (`roofline.profiled`; tests/test_profiles.py recomputes it)."""
    cell = lw_cell if kernel.startswith("lw") else SW_FLOPS_PER_CELL
    tflops = cell * n_gpt * nlay * ncol / (kernel_ms * 1e-3) / 1e12
    return {"kernel_ms": kernel_ms, "achieved": tflops, "frac": tflops / peak}


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--ncol", type=int, default=NCOL_PER_GPU, help="columns per GPU")
    ap.add_argument("--nlay", type=int, default=NLAY)
    ap.add_argument("--dtype", default="f32", choices=["f32", "f64"])
    ap.add_argument("--aerosols", action="store_true")
    ap.add_argument("--cld-frac", type=float, default=1.0, help="cloud fraction of cloudy layers (reference benchmark: 1)")
    ap.add_argument("--gcm-mix", action="store_true",
                    help="a GCM-like column mix instead of the all-day / overcast benchmark columns: half the columns at night "
                         "(mu0 <= 0), the others with their own zenith angle, a random cloud fraction per cloudy layer")
    ap.add_argument("--cpu-sample", type=int, default=None, help="columns for the CPU baseline (0 disables)")
    ap.add_argument("--host", action="store_true",
                    help="hand HOST arrays to the C ABI (library stages H2D/D2H every step): the PCIe-inclusive rate, "
                         "never the headline value")
    ap.add_argument("--shards", type=int, default=0,
                    help="with --host: ONE process drives this many column shards through a multi-device workspace "
                         "(rrtmgp_hip_workspace_create_multi): device ids 0..n-1, wrapped onto the visible GPUs")
    ap.add_argument("--single-process", action="store_true",
                    help="with --gpus N: ONE host process (no torch.distributed) hands host arrays of N x --ncol columns "
                         "to a multi-device workspace over GPUs 0..N-1; the library fans out (one host thread + stream "
                         "per GPU) and returns when every slab is home.  What a Julia host gets from HIPDevice(ids)")
    ap.add_argument("--streams", type=int, default=1, choices=[1, 2],
                    help="1: LW and SW kernels on torch's current stream; 2: each on its own stream (concurrent)")
    ap.add_argument("--clear-sky-diag", choices=["off", "one-pass", "two-solves"], default="off",
                    help="also produce clear-sky fluxes (AllSkyRadiationWithClearSkyDiagnostics): in the same launch, "
                         "or as the reference does with a second, cloudless solve.  Not the default workload.")
    ap.add_argument("--tile", type=int, default=1, help="repeat the generated columns this many times on the device "
                                                        "(large single-GPU batches without the host-side generation cost)")
    ap.add_argument("--lw-solver", choices=["2stream", "noscat"], default="2stream",
                    help="noscat: NoScatLWRTE (rte_lw_noscat_solve!) with --angles quadrature angles; the SW solver stays "
                         "two-stream: the reference's clear-sky pairing (test/runtests.jl:54-62)")
    ap.add_argument("--angles", type=int, default=1, help="Gauss-Jacobi angles of the no-scattering LW solver (1..4)")
    ap.add_argument("--no-clouds", action="store_true", help="clear sky: no cloud state, no cloud lookups")
    ap.add_argument("--lw-only", action="store_true", help="a step is the LW solve alone")
    ap.add_argument("--scaling", choices=["weak", "strong"], default="weak",
                    help="strong: --ncol-total columns are split over the ranks in contiguous ranges (BASELINE config 4: "
                         "4096 columns on 1 -> 8 GPUs); weak (default): --ncol columns per rank")
    ap.add_argument("--ncol-total", type=int, default=4096, help="with --scaling strong: columns of the whole job")
    ap.add_argument("--l2", choices=["off", "fused", "split"], default="off",
                    help="time the Layer-2 step of a host model (update_fluxes! on HOST arrays: prepare_atmosphere! + LW + SW + net, "
                         "AllSkyRadiation): `fused` = ONE call of rrtmgp_hip_update_fluxes (state staged once), `split` = the "
                         "reference's four steps as separate calls (prepare, LW, SW staged separately; net sum on the host)")
    ap.add_argument("--resident", action="store_true",
                    help="with --l2 fused: the Layer-2 solver with EVERY array in HBM (RRTMGPSolver(resident=True), what "
                         "array_type = a device array gives the reference): update_fluxes! stages nothing over PCIe")
    ap.add_argument("--fused-step", action="store_true",
                    help="a step is ONE rrtmgp_hip_update_fluxes call (LW + SW + net sum on one workspace; short steps "
                         "overlap the two solvers) instead of two solve calls")
    ap.add_argument("--no-legs", action="store_true", help="only the headline measurement (no variant / host / CPU legs)")
    ap.add_argument("--leg", default=None, help=argparse.SUPPRESS)   # child-process mode: compact JSON, no legs
    return ap.parse_args(argv)


def run_leg(name, extra, env=None, timeout=600):
    """One variant of the workload in a child process (its own library instance); returns its compact record."""
    cmd = [sys.executable, os.path.abspath(__file__), "--leg", name, "--steps", "8", "--warmup", "2"] + extra
    e = dict(os.environ)
    e.update(env or {})
    try:
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout, env=e)
        line = [x for x in r.stdout.splitlines() if x.startswith("{")]
        if r.returncode != 0 or not line:
            return {"error": (r.stderr or r.stdout)[-300:]}
        return json.loads(line[-1])
    except Exception as ex:  # noqa: BLE001
        return {"error": repr(ex)[:300]}


def rank_columns(ncol_total, rank, world):
    """Strong scaling: the contiguous range [lo, hi) of the job's columns that `rank` of `world` solves — the ranges of
    rrtmgp.jl_amd/sharding.py (balanced, the first ranks take the remainder), which is also how the library shards a
    multi-device workspace.  A pure function (tests/test_sharding.py checks it at world size 8 without a GPU)."""
    from rrtmgp_jl_amd import sharding
    return sharding.shard_range(ncol_total, rank, world)


def self_launch(n):
    """Re-executes this command line under torch.distributed.run with `n` ranks on 127.0.0.1 (a free port)."""
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # RCCL needs dmabuf IPC on this platform
    return subprocess.run(cmd, env=env).returncode


def vmem_pipeline(k):
    """Share of the dominant kernel's time that the CUs' vector memory pipelines are busy, from the profile's counters and
    the per-instruction cost measured by tools/ubench/gather_l1.hip (profiles/r04_gather_l1_ubench.txt): a wave-wide
    access of 4 bytes per lane holds a CU's pipeline for 8 cycles, one of 8-16 bytes for 16.5.  The 4-byte instructions
    of these kernels are the sweep-scratch accesses: every store and as many loads; every other load is 16 bytes wide."""
    need = ("SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR", "GRBM_GUI_ACTIVE", "avg_us")
    if not all(k.get(c) for c in need):
        return None
    narrow = 2.0 * k["SQ_INSTS_VMEM_WR"]
    wide = k["SQ_INSTS_VMEM_RD"] - k["SQ_INSTS_VMEM_WR"]
    busy = 8.0 * narrow + 16.5 * wide                      # pipeline cycles, all CUs together
    cycles = k["GRBM_GUI_ACTIVE"] / 8.0                    # the counter sums the 8 XCDs
    n_cu = 256
    return {"frac": busy / (n_cu * cycles), "busy_cycles_per_cu": busy / n_cu, "kernel_cycles": cycles,
            "instructions_4_byte": narrow, "instructions_16_byte": wide, "profiled_kernel_ms": k["avg_us"] / 1e3,
            "model": "8 cycles per 4-byte, 16.5 per 8..16-byte wave instruction (tools/ubench/gather_l1.hip); "
                     "counters SQ_INSTS_VMEM_RD/WR, GRBM_GUI_ACTIVE of the profile named in roofline.traffic_source"}


def leg_profile(leg, ncol, nlay, ngpt_lw, ngpt_sw, peak, lw_cell):
    """`profiled` block of a variant leg: the same flops-per-cell pricing as `roofline.profiled`, on the rocprofv3
    --kernel-trace --stats averages committed as profiles/latest_<leg>.json (tools/profile2.sh with BENCH_ARGS = the leg's
    flags) — only when that summary was taken on the kernel sources being timed."""
    path = os.path.join(ROOT, "profiles", f"latest_{leg}.json")
    if not os.path.exists(path):
        return None
    with open(path) as fh:
        pj = json.load(fh)
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    from rocprof_summary import kernel_source_sha256
    if pj.get("kernel_source_sha256") != kernel_source_sha256(ROOT):
        return {"source": pj.get("source"), "note": "profile was taken on other kernel sources: withheld"}
    out = {"source": f"rocprofv3 --kernel-trace --stats averages of profiles/latest_{leg}.json ({pj.get('source')})",
           "git_sha": pj.get("git_sha"), "kernels": {}}
    tot_ms = tot_flops = 0.0
    for name, k in pj.get("kernels", {}).items():
        if not k.get("avg_us"):
            continue
        r = kernel_valu(name, k["avg_us"] / 1e3, ncol, nlay, ngpt_lw if name.startswith("lw") else ngpt_sw, peak, lw_cell)
        if k.get("FETCH_SIZE") is not None and k.get("WRITE_SIZE") is not None:
            r["traffic"] = (2.0 * k["FETCH_SIZE"] + k["WRITE_SIZE"]) * 1024.0
        out["kernels"][name] = r
        tot_ms += r["kernel_ms"]
        tot_flops += r["achieved"] * r["kernel_ms"]
    if tot_ms > 0:
        out["step"] = {"kernel_ms": tot_ms, "achieved": tot_flops / tot_ms, "frac": tot_flops / tot_ms / peak, "peak": peak}
    return out


# The reference's published numbers for its own benchmark grid, 86 400 columns x 63 layers (ns, minimum;
# /root/reference/perf/benchmark_baselines/nvidia_a100_sxm4_40gb.txt:6-17) and the wall times of docs/src/howto/gpu.md:138-150
A100_MIN_NS = {("all_sky", "f32"): (7.26150458e8, 6.71148384e8), ("all_sky", "f64"): (9.92342112e8, 9.40083761e8),
               ("all_sky_with_aerosols", "f32"): (7.97787842e8, 7.28343007e8), ("all_sky_with_aerosols", "f64"): (1.09698575e9, 9.66275699e8),
               ("clear_sky", "f32"): (4.89919056e8, 4.55409919e8), ("clear_sky", "f64"): (6.82068605e8, 6.39967283e8)}
A100_DOC_S = {("clear_sky", "f32"): 0.95, ("all_sky", "f32"): 1.40, ("all_sky_with_aerosols", "f32"): 1.56,
              ("clear_sky", "f64"): 1.30, ("all_sky", "f64"): 1.92, ("all_sky_with_aerosols", "f64"): 2.07}


def a100_shape_legs():
    """Same-shape context for the reference's published tables: every cell of docs/src/howto/gpu.md:138-150 and of the A100
    ratchet file on the reference's own grid (86 400 x 63, two-stream LW + SW), LW and SW kernel times separately."""
    cells = {}
    for case, flags in (("clear_sky", ["--no-clouds"]), ("all_sky", []), ("all_sky_with_aerosols", ["--aerosols"])):
        for dt in ("f32", "f64"):
            leg = run_leg(f"a100_shape_{case}_{dt}", ["--ncol", "86400", "--nlay", "63", "--dtype", dt] + flags)
            if "error" not in leg:
                lw_ns, sw_ns = A100_MIN_NS[(case, dt)]
                step_s = leg["min_ms"] * 1e-3 if "min_ms" in leg else leg["ms_per_step"] * 1e-3
                leg = {"lw_kernel_ms": leg["lw_kernel_ms"], "sw_kernel_ms": leg["sw_kernel_ms"], "step_min_ms": leg.get("min_ms"),
                       "step_median_ms": leg.get("median_ms"), "columns_per_s": leg["value"],
                       "a100_lw_ms": lw_ns / 1e6, "a100_sw_ms": sw_ns / 1e6, "a100_doc_step_s": A100_DOC_S[(case, dt)],
                       "lw_vs_a100": lw_ns / 1e6 / leg["lw_kernel_ms"], "sw_vs_a100": sw_ns / 1e6 / leg["sw_kernel_ms"],
                       "step_vs_a100_ratchet_min": (lw_ns + sw_ns) / 1e9 / step_s, "step_vs_a100_doc": A100_DOC_S[(case, dt)] / step_s}
            cells[f"{case}_{dt}"] = leg
    return {"grid": "86 400 columns x 63 layers, 256 + 224 g-points, two-stream LW + SW (perf/benchmark_ratchet.jl:49-50)",
            "cells": cells,
            "note": "context only, never `value`: synthetic tables and columns of the reference's dimensions on ONE MI355X next to the "
                    "reference's published A100-SXM4-40GB minimum times (benchmark_baselines/nvidia_a100_sxm4_40gb.txt) and "
                    "documented wall times (docs/src/howto/gpu.md:138-150); *_vs_a100 = A100 time / MI355X time"}


def run_l2(args):
    """The Layer-2 step on host arrays through the Python mirror of RRTMGPSolver (the Julia glue's update_fluxes! override
    makes the same library call): columns/s and the bytes that crossed PCIe per column, from the workspace's counters."""
    import rrtmgp_jl_amd  # noqa: F401
    from rrtmgp_jl_amd import _lib, solver as L2, synthetic as S
    from rrtmgp_jl_amd.states import TEST_PARAMETERS
    _lib.require_gpu()
    ft = np.float32 if args.dtype == "f32" else np.float64
    ncol, nlay = args.ncol, args.nlay
    lw, sw = S.make_gas_lookup("lw", ft), S.make_gas_lookup("sw", ft)
    cl, cs = S.make_cloud_lookup("lw", lw.n_bnd, ft), S.make_cloud_lookup("sw", sw.n_bnd, ft)
    al = S.make_aerosol_lookup("lw", lw.bnd_lims_wn, ft) if args.aerosols else None
    asw = S.make_aerosol_lookup("sw", sw.bnd_lims_wn, ft) if args.aerosols else None
    as_, lb, sb = S.make_columns(ncol, nlay, ft, seed=2026, clouds=True, cld_frac=args.cld_frac, aerosols=args.aerosols,
                                 cos_zenith=0.86)
    diag = args.clear_sky_diag != "off"
    method = (L2.AllSkyRadiationWithClearSkyDiagnostics if diag else L2.AllSkyRadiation)(aerosol_radiation=args.aerosols)
    s = L2.RRTMGPSolver(method, TEST_PARAMETERS, lb, sb, as_, lookups=L2.LookupBundle(lw, sw, cl, cs, al, asw),
                        fused=args.l2 == "fused", resident=args.resident)
    for _ in range(args.warmup):
        L2.update_fluxes(s)
    s.lws.ws.synchronize()
    b0 = s.lws.ws.transfer_bytes()
    ms = []
    t0 = time.perf_counter()
    for _ in range(args.steps):
        ts = time.perf_counter()
        L2.update_fluxes(s)
        if args.resident:
            s.lws.ws.synchronize()   # device arrays: the call is stream-ordered; a host model reads the fluxes after this
        ms.append(1e3 * (time.perf_counter() - ts))
    elapsed = time.perf_counter() - t0
    b1 = s.lws.ws.transfer_bytes()
    from rrtmgp_jl_amd.states import to_host
    assert np.isfinite(to_host(L2.net_flux(s))).all() and (to_host(L2.lw_flux_up(s))[0] > 0).all()
    rec = {"value": ncol * args.steps / elapsed, "unit": "columns/s", "ms_per_step": 1e3 * elapsed / args.steps,
           "min_ms": min(ms), "median_ms": statistics.median(ms), "ncol": ncol, "dtype": args.dtype,
           "h2d_bytes_per_column": (b1[0] - b0[0]) / (ncol * args.steps), "d2h_bytes_per_column": (b1[1] - b0[1]) / (ncol * args.steps),
           "calls_per_step": 1 if args.l2 == "fused" else 3,
           "workload": f"update_fluxes! on {'DEVICE-RESIDENT' if args.resident else 'HOST'} arrays ({args.l2}): prepare_atmosphere! (clip + col_dry) + LW + SW two-stream + net, "
                       f"{'AllSkyRadiationWithClearSkyDiagnostics' if diag else 'AllSkyRadiation'}, {ncol} x {nlay}"}
    print(json.dumps(rec))


def main():
    args = parse_args()
    if args.l2 != "off":
        return run_l2(args)
    import torch
    import rrtmgp_jl_amd  # noqa: F401
    from rrtmgp_jl_amd import _lib, rte, synthetic as S

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    single = args.single_process
    if single:
        if world != 1:
            raise SystemExit("--single-process is one process: do not launch it through torch.distributed.run")
        args.host, args.shards = True, args.gpus
    elif args.gpus != world:
        if "WORLD_SIZE" not in os.environ and args.gpus > 1:
            # a bare `python bench.py --gpus N`: start the N ranks ourselves (what the driver's torch.distributed.run
            # command does), forward the one JSON line of rank 0 and its exit status
            raise SystemExit(self_launch(args.gpus))
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run "
                         f"--nproc-per-node {args.gpus} (a bare `python bench.py --gpus N` starts its own ranks)")
    _lib.require_gpu()
    # RRTMGP_BENCH_BACKEND=gloo + RRTMGP_BENCH_SHARE_GPU=1 is a test hook: several ranks on ONE GPU, to exercise the
    # multi-process path (barrier, MAX over ranks, rank-0 report) where only one device exists.  Never for numbers.
    backend = os.environ.get("RRTMGP_BENCH_BACKEND", "nccl")
    if os.environ.get("RRTMGP_BENCH_SHARE_GPU"):
        local_rank = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    ft = np.float32 if args.dtype == "f32" else np.float64
    strong = args.scaling == "strong"
    if strong:
        if single or args.tile != 1:
            raise SystemExit("--scaling strong is the one-process-per-GPU path without --tile")
        c_lo, c_hi = rank_columns(args.ncol_total, rank, world)
        args.ncol = c_hi - c_lo
        if args.ncol < 1:
            raise SystemExit("--ncol-total is smaller than the number of ranks")
    clouds = not args.no_clouds
    ncol0, nlay = args.ncol * (args.gpus if single else 1), args.nlay
    ncol = ncol0 * args.tile
    lw, sw = S.make_gas_lookup("lw", ft), S.make_gas_lookup("sw", ft)
    cl, cs = (S.make_cloud_lookup("lw", lw.n_bnd, ft), S.make_cloud_lookup("sw", sw.n_bnd, ft)) if clouds else (None, None)
    al = S.make_aerosol_lookup("lw", lw.bnd_lims_wn, ft) if args.aerosols else None
    asw = S.make_aerosol_lookup("sw", sw.bnd_lims_wn, ft) if args.aerosols else None
    col_offset = c_lo if strong else rank * ncol
    mix = dict(cos_zenith=None, night_fraction=0.5, random_cld_frac=True) if args.gcm_mix else dict(cos_zenith=0.86)
    as_h, lb_h, sb_h = S.make_columns(ncol0, nlay, ft, seed=2026, col_offset=col_offset, clouds=clouds,
                                      cld_frac=args.cld_frac, aerosols=args.aerosols, **mix)
    shards = None
    if args.host:
        if args.tile != 1:
            raise SystemExit("--tile is for device-resident runs")
        as_d, lb_d, sb_d = as_h, lb_h, sb_h
        if args.shards > 0:
            ndev = torch.cuda.device_count()
            shards = [i % ndev for i in range(args.shards)]
    else:
        as_d, lb_d, sb_d = as_h.to_device(dev), lb_h.to_device(dev), sb_h.to_device(dev)
        if args.tile > 1:   # ncol is the slowest (first torch) dimension of every per-column array
            rep = lambda t: t if t is None or t.dim() == 0 or t.shape[0] != ncol0 else t.repeat(  # noqa: E731
                (args.tile,) + (1,) * (t.dim() - 1)).contiguous()
            as_d, lb_d, sb_d = as_d._map(rep), lb_d._map(rep), sb_d._map(rep)
    wdev = shards if shards else local_rank
    ws_lw = rte.Workspace(ncol, nlay, ft, wdev)
    ws_sw = ws_lw if args.fused_step else rte.Workspace(ncol, nlay, ft, wdev)
    noscat = args.lw_solver == "noscat"
    slv_lw = (rte.NoScatLWRTE(ncol, nlay, ft, lb_d, device=local_rank, flux_device=None if args.host else dev, workspace=ws_lw,
                              n_gauss_angles=args.angles) if noscat else
              rte.TwoStreamLWRTE(ncol, nlay, ft, lb_d, device=local_rank, flux_device=None if args.host else dev, workspace=ws_lw))
    slv_sw = rte.TwoStreamSWRTE(ncol, nlay, ft, sb_d, device=local_rank, flux_device=None if args.host else dev, workspace=ws_sw)
    if args.streams == 1 and not shards:
        slv_lw.ws.use_torch_stream()
        slv_sw.ws.use_torch_stream()
    d_lw, d_lw_cld, d_lw_aero = (rte.DeviceLookup(x, wdev) if x is not None else None for x in (lw, cl, al))
    d_sw, d_sw_cld, d_sw_aero = (rte.DeviceLookup(x, wdev) if x is not None else None for x in (sw, cs, asw))

    clr_lw = clr_sw = None
    if args.clear_sky_diag != "off":
        from rrtmgp_jl_amd.states import Flux
        fdev = None if args.host else dev
        clr_lw = Flux.allocate(ncol, nlay + 1, ft, sw=False, device=fdev)
        clr_sw = Flux.allocate(ncol, nlay + 1, ft, sw=True, device=fdev)
        if args.clear_sky_diag == "two-solves":
            slv_lw_c = rte.TwoStreamLWRTE(ncol, nlay, ft, lb_d, device=local_rank, flux_device=fdev, workspace=slv_lw.ws)
            slv_sw_c = rte.TwoStreamSWRTE(ncol, nlay, ft, sb_d, device=local_rank, flux_device=fdev, workspace=slv_sw.ws)
            slv_lw_c.flux, slv_sw_c.flux = clr_lw, clr_sw

    net_d = None
    if args.fused_step:
        if args.clear_sky_diag == "two-solves" or args.lw_only or noscat:
            raise SystemExit("--fused-step is the two-stream LW + SW step (optionally with --clear-sky-diag one-pass)")
        net_d = np.empty((nlay + 1, ncol), ft, order="F") if args.host else torch.empty((ncol, nlay + 1), dtype=lb_d.sfc_emis.dtype, device=dev)

    def step():
        if args.fused_step:
            one = args.clear_sky_diag == "one-pass"
            rte.update_fluxes(slv_lw, slv_sw, as_d, d_lw, d_sw, d_lw_cld, d_sw_cld, d_lw_aero, d_sw_aero, seed=2026,
                              col_offset=col_offset, net_flux=net_d, clear_flux_lw=clr_lw if one else None,
                              clear_flux_sw=clr_sw if one else None)
            return
        if args.clear_sky_diag == "two-solves":
            rte.solve_lw(slv_lw_c, as_d, d_lw, None, d_lw_aero, seed=2026, col_offset=col_offset)
            rte.solve_sw(slv_sw_c, as_d, d_sw, None, d_sw_aero, seed=2026, col_offset=col_offset)
        one = args.clear_sky_diag == "one-pass"
        rte.solve_lw(slv_lw, as_d, d_lw, d_lw_cld, d_lw_aero, seed=2026, col_offset=col_offset,
                     clear_flux=clr_lw if one else None)
        if not args.lw_only:
            rte.solve_sw(slv_sw, as_d, d_sw, d_sw_cld, d_sw_aero, seed=2026, col_offset=col_offset,
                         clear_flux=clr_sw if one else None)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    barrier()
    # Per-step times: device-resident runs are stream-ordered, so events on the launch stream bracket each step and
    # are read AFTER the timed region; host-array runs block inside the C ABI call, so the host clock is the step time.
    on_torch_stream = args.streams == 1 and not args.host
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)] \
        if on_torch_stream else None
    host_ms = []
    t0 = time.perf_counter()
    for i in range(args.steps):
        if ev:
            ev[i][0].record()
            step()
            ev[i][1].record()
        else:
            ts = time.perf_counter()
            step()
            if args.host:
                host_ms.append(1e3 * (time.perf_counter() - ts))
    barrier()
    elapsed = time.perf_counter() - t0
    ranks_seen, rank_ms, rank_ncol = 1, [1e3 * elapsed / args.steps], [ncol]
    # which physical device each rank ran on (ordinal, name, PCI address): lets a scaling record be audited for ranks that
    # shared a GPU or never showed up
    pr = torch.cuda.get_device_properties(local_rank)
    me = {"rank": rank, "local_rank": local_rank, "device": torch.cuda.current_device(), "name": pr.name,
          "pci": "%04x:%02x:%02x" % (getattr(pr, "pci_domain_id", 0), getattr(pr, "pci_bus_id", 0), getattr(pr, "pci_device_id", 0)),
          "uuid": str(getattr(pr, "uuid", "")), "hip_visible_devices": os.environ.get("HIP_VISIBLE_DEVICES"),
          "pid": os.getpid()}
    rank_devices = [me]
    if world > 1:
        cdev = dev if backend == "nccl" else "cpu"
        ranks_seen = dist.get_world_size()
        rank_devices = [None] * ranks_seen
        dist.all_gather_object(rank_devices, me)
        mine = torch.tensor([elapsed, float(ncol)], dtype=torch.float64, device=cdev)
        every = [torch.zeros_like(mine) for _ in range(ranks_seen)]
        dist.all_gather(every, mine)     # per-rank wall time of the timed region and columns per step
        rank_ms = [1e3 * float(e[0].item()) / args.steps for e in every]
        rank_ncol = [int(e[1].item()) for e in every]
        t = torch.tensor([elapsed], dtype=torch.float64, device=cdev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    per_step = [a.elapsed_time(b) for a, b in ev] if ev else host_ms

    # kernel durations: HIP events recorded by the library around each launch on the launch stream, over three
    # further steps OUTSIDE the timed region (reading them synchronises)
    # (median of 5: a single sample is exposed to whatever the box does between two synchronised launches)
    s_lw, s_sw = [], []
    for _ in range(5):
        step()
        s_lw.append(slv_lw.ws.last_kernel_ms())
        if not args.lw_only:
            s_sw.append(slv_sw.ws.last_kernel_ms())
    k_lw = statistics.median(s_lw)
    k_sw = statistics.median(s_sw) if s_sw else 0.0

    # sanity: results are finite and physical (never timed)
    up, sdn = torch.as_tensor(slv_lw.flux.flux_up), torch.as_tensor(slv_sw.flux.flux_dn)
    if args.host:
        up, sdn = up.T, sdn.T  # numpy (nlev, ncol) -> (ncol, nlev) like the device tensors
    if not os.environ.get("RRTMGP_BENCH_NO_CHECK"):   # (timing-only experiment libraries give wrong results by construction)
        assert bool(torch.isfinite(up).all()) and bool((up[:, 0] > 0).all())
        assert args.lw_only or bool(torch.isfinite(sdn).all())

    lw_cell = (LW_NOSCAT_FLOPS_PER_CELL + LW_NOSCAT_FLOPS_PER_ANGLE * args.angles) if noscat else LW_FLOPS_PER_CELL
    flops_col = nlay * (lw.n_gpt * lw_cell + (0.0 if args.lw_only else sw.n_gpt * SW_FLOPS_PER_CELL))
    valu_peak = VALU_PEAK_TFLOPS if args.dtype == "f32" else FP64_VALU_PEAK_TFLOPS
    if rank == 0:
        # weak: every rank solved ncol columns per step; strong: the ranks' ranges add up to --ncol-total
        value = (args.ncol_total if strong else world * ncol) * args.steps / elapsed   # single-process: ncol already is the whole job
        n_gpus = args.gpus if single else world
        step_ms = None
        if per_step:
            step_ms = {"min": min(per_step), "median": statistics.median(per_step), "mean": statistics.fmean(per_step),
                       "n": len(per_step), "clock": "hip events on the launch stream" if ev else "host clock around the blocking call"}
        if args.leg:   # child-process mode: a compact record for the parent's JSON line
            rec = {"value": value, "unit": "columns/s", "ms_per_step": 1e3 * elapsed / args.steps,
                   "lw_kernel_ms": k_lw, "sw_kernel_ms": k_sw, "ncol": ncol, "dtype": args.dtype,
                   "library": os.path.basename(_lib.SO_PATH)}
            if not args.host:   # (a pipelined host solve is many chunk launches: the library's kernel timer holds the last one only)
                rec["valu"] = {"achieved": flops_col * ncol / ((k_lw + k_sw) * 1e-3) / 1e12, "peak": valu_peak, "unit": "TFLOP/s",
                               "frac": flops_col * ncol / ((k_lw + k_sw) * 1e-3) / 1e12 / valu_peak,
                               "algorithmic_flops_per_column": flops_col}
            else:
                rec.pop("lw_kernel_ms"), rec.pop("sw_kernel_ms")
            if noscat or not clouds or args.lw_only:
                rec["workload"] = (f"{'NoScatLWRTE x ' + str(args.angles) + ' angle(s)' if noscat else 'TwoStreamLWRTE'}"
                                   f"{'' if args.lw_only else ' + TwoStreamSWRTE'}, {'McICA clouds' if clouds else 'clear sky'}, "
                                   f"{ncol} x {nlay}")
            if step_ms:
                rec["min_ms"], rec["median_ms"] = step_ms["min"], step_ms["median"]
                rec["value_at_min"] = ncol / (step_ms["min"] * 1e-3)
            if args.gcm_mix:
                rec["workload"] = (f"all-sky LW+SW two-stream, {ncol} x {nlay}: {float((sb_h.cos_zenith <= 0).mean()):.2f} of the columns "
                                   "at night, per-column zenith angle, random cloud fraction per cloudy layer")
            prof_leg = leg_profile(args.leg, ncol, nlay, lw.n_gpt, sw.n_gpt, valu_peak, lw_cell)
            if prof_leg:
                rec["profiled"] = prof_leg
            print(json.dumps(rec))
            return
        ms_lw, ms_sw = k_lw, k_sw
        ft_bytes = np.dtype(ft).itemsize
        b_lw, b_sw, b_step = algorithmic_bytes(nlay, lw.n_bnd, sw.n_bnd, ft_bytes)
        if args.aerosols:
            b_lw += 30 * nlay * ft_bytes
            b_sw += (30 * nlay + 2) * ft_bytes
            b_step += (30 * nlay + 2) * ft_bytes
        dom = "lw_solve_kernel" if ms_lw >= ms_sw else "sw_solve_kernel"
        dom_ms, dom_bytes = (ms_lw, b_lw) if ms_lw >= ms_sw else (ms_sw, b_sw)
        achieved = dom_bytes * ncol / (dom_ms * 1e-3) / 1e9
        flops = flops_col * ncol
        valu_tflops = flops / ((ms_lw + ms_sw) * 1e-3) / 1e12
        lw_flops_col = nlay * lw.n_gpt * lw_cell
        sw_flops_col = 0.0 if args.lw_only else nlay * sw.n_gpt * SW_FLOPS_PER_CELL
        lw_tflops = lw_flops_col * ncol / (ms_lw * 1e-3) / 1e12
        sw_tflops = sw_flops_col * ncol / (ms_sw * 1e-3) / 1e12 if ms_sw > 0 else 0.0
        dom_flops_col, dom_tflops = (lw_flops_col, lw_tflops) if ms_lw >= ms_sw else (sw_flops_col, sw_tflops)
        traffic = traffic_source = profiled = None
        prof = os.path.join(ROOT, "profiles", "latest.json")
        default_workload = (not args.aerosols and ncol == NCOL_PER_GPU and nlay == NLAY and args.dtype == "f32"
                            and not args.host and args.clear_sky_diag == "off" and not noscat and clouds and not args.lw_only
                            and not strong and not args.gcm_mix)
        vmem = None
        if os.path.exists(prof) and default_workload:
            # HBM bytes per launch of the dominant kernel from the rocprofv3 PMC passes of this same command
            # (tools/profile2.sh -> tools/rocprof_summary.py): (2 * FETCH_SIZE + WRITE_SIZE) KB, the factor 2
            # being the gfx950 FETCH_SIZE correction of MI355X_MICROARCH.md
            with open(prof) as fh:
                pj = json.load(fh)
            k = pj.get("kernels", {}).get(dom)
            # the counters belong to this line only if they were taken on the kernel sources that are being timed
            sys.path.insert(0, os.path.join(ROOT, "tools"))
            from rocprof_summary import kernel_source_sha256
            same_sources = pj.get("kernel_source_sha256") == kernel_source_sha256(ROOT)
            if not same_sources:
                traffic_source = {"profile": pj.get("source"), "git_sha": pj.get("git_sha"),
                                  "note": "profile was taken on other kernel sources: traffic withheld"}
            elif k and k.get("FETCH_SIZE") is not None and k.get("WRITE_SIZE") is not None:
                traffic = (2.0 * k["FETCH_SIZE"] + k["WRITE_SIZE"]) * 1024.0
                traffic_source = {"profile": pj.get("source"), "git_sha": pj.get("git_sha"),
                                  "profiled_kernel_ms": k.get("avg_us", 0.0) / 1e3}
                vmem = vmem_pipeline(k)
                profiled = kernel_valu(dom, k["avg_us"] / 1e3, ncol, nlay, lw.n_gpt if dom.startswith("lw") else sw.n_gpt)
                profiled["source"] = "rocprofv3 --kernel-trace --stats average of profiles/latest.json (" + str(pj.get("source")) + ")"
        out = {
            "metric": "columns/sec, all-sky LW+SW 2-stream (nlay=64, 256+224 gpt)",
            "value": value,
            "unit": "columns/s",
            "n_gpus": n_gpus, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None,
            "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": (f"all-sky (McICA clouds, cld_frac={args.cld_frac:g})" if clouds else "clear-sky") +
                                   (f" LW no-scattering ({args.angles} angle(s))" + ("" if args.lw_only else " + SW two-stream")
                                    if noscat else " LW" + ("" if args.lw_only else "+SW") + " two-stream") + ", " +
                                   (f"{args.ncol_total} columns split over {world} GPU(s) (strong scaling, BASELINE config 4)"
                                    if strong else f"{ncol // (args.gpus if single else 1)} columns/GPU") +
                                   f" x {nlay} layers, {lw.n_gpt}+{sw.n_gpt} g-points, "
                                   f"VmrGM{', MERRA aerosols' if args.aerosols else ''}, "
                                   f"{'HOST arrays staged over PCIe every step' if args.host else 'state resident in HBM'}"
                                   + (f", {len(shards)} shards in one process" if shards else "")
                                   + ("" if args.clear_sky_diag == "off" else f", + clear-sky diagnostic ({args.clear_sky_diag})"),
                       "ncol_per_gpu": ncol // (args.gpus if single else 1), "nlay": nlay, "ngpt_lw": lw.n_gpt, "ngpt_sw": sw.n_gpt,
                       "parallelism": (f"ONE host process, columns sharded over {n_gpus} GPU(s) inside the library "
                                       "(rrtmgp_hip_workspace_create_multi), no collective" if single else
                                       f"columns sharded over {world} GPU(s), no collective")},
            "step_ms": step_ms,
            # what the process group reported after init_process_group, every rank's own wall time per step (the job's
            # ms_per_step is their MAX) and the columns each rank solved per step
            "ranks_seen": ranks_seen, "rank_ms_per_step": rank_ms, "rank_columns": rank_ncol, "rank_devices": rank_devices,
            # The ceiling that binds is the FP32 vector rate (no contraction, HBM at < 1 %): the
            # contract object prices the DOMINANT kernel's own algorithmic flops against it, with its live event time; the
            # HBM view the task statement asks for (algorithmic bytes against 8 TB/s, PMC traffic) is nested under `hbm`
            "roofline": {"bound": "valu", "kernel": dom, "achieved": dom_tflops, "peak": valu_peak, "unit": "TFLOP/s",
                         "frac": dom_tflops / valu_peak, "kernel_ms": dom_ms, "algorithmic_flops_per_column": dom_flops_col,
                         "traffic": traffic, "profiled": profiled,
                         "hbm": {"achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                                 "algorithmic_bytes_per_column": dom_bytes, "algorithmic_bytes": dom_bytes * ncol,
                                 "traffic": traffic, "traffic_ratio": (traffic / (dom_bytes * ncol)) if traffic else None,
                                 "traffic_source": traffic_source},
                         "per_kernel": {"lw_solve_kernel": {"ms": ms_lw, "achieved": lw_tflops, "frac": lw_tflops / valu_peak},
                                        "sw_solve_kernel": {"ms": ms_sw, "achieved": sw_tflops, "frac": (sw_tflops / valu_peak)}},
                         "note": "FP32 vector peak 157.3 TFLOP/s (MI355X_MICROARCH.md); algorithmic flops per (layer, g-point) cell: "
                                 "LW 301, SW 345 (SURVEY.md section 8(d)); `traffic` = 2*FETCH_SIZE + WRITE_SIZE of the committed "
                                 "rocprofv3 PMC summary (profiles/latest.json), per launch of the dominant kernel"},
            "valu": {"achieved": valu_tflops, "peak": valu_peak, "unit": "TFLOP/s",
                     "frac": valu_tflops / valu_peak,
                     "algorithmic_flops_per_column": flops / ncol,
                     "note": "both kernels of the step together"},
            # the resource that does bind, from the same profile as `traffic`; null without it
            "vmem_pipeline": vmem,
            "kernels": {"lw_solve_kernel_ms": ms_lw, "sw_solve_kernel_ms": ms_sw,
                        "lw_bytes_per_column": b_lw, "sw_bytes_per_column": b_sw, "step_bytes_per_column": b_step,
                        "timing": "HIP events by the library around each launch, median of 5 steps after the timed region"},
        }
        legs = world == 1 and not args.no_legs and default_workload
        if legs:
            # free this process's device memory first: the children run on the same GPU
            del as_d, lb_d, sb_d, slv_lw, slv_sw, ws_lw, ws_sw, d_lw, d_lw_cld, d_sw, d_sw_cld
            torch.cuda.empty_cache()
            out["host_end_to_end"] = run_leg("host", ["--host"])
            out["host_end_to_end"]["note"] = ("same workload, HOST arrays staged H2D/D2H inside each solve (page-locked once, "
                                              "chunked pipeline); never `value`")
            # what a HOST model (the Julia drop-in with array_type = Array) pays per radiation step: the Layer-2 step as one
            # library call, next to the reference's four steps as separate calls
            out["l2_update_fluxes_host"] = {"fused": run_leg("l2_fused", ["--l2", "fused"]),
                                            "split": run_leg("l2_split", ["--l2", "split"]),
                                            "fused_clear_sky_diag": run_leg("l2_fused_diag", ["--l2", "fused", "--clear-sky-diag", "one-pass"]),
                                            "note": "update_fluxes!(solver) from host arrays, PCIe inclusive; never `value`"}
            # the same Layer-2 step with every array of the solver resident in HBM (RRTMGPSolver(resident=True)): what the
            # reference's device array type buys — nothing crosses PCIe inside update_fluxes!
            out["l2_update_fluxes_device"] = run_leg("l2_device", ["--l2", "fused", "--resident"])
            out["l2_update_fluxes_device"]["vs_value"] = (out["l2_update_fluxes_device"].get("value", 0.0) / value) if value else None
            fast = os.path.join(ROOT, "rrtmgp.jl_amd", "libhip_rrtmgp_fast.so")
            out["variants"] = {
                # the opt-in raw-instruction Float32 build (-DRR_FAST_F32: __expf, v_rcp_f32, v_sqrt_f32 as they come); `value`
                # above is the shipped library, whose Float32 forms are IEEE-accurate (rrtmgp_hip_build_flags() == "")
                "fast_f32": (run_leg("fast_f32", [], env={"RRTMGP_HIP_LIBRARY": fast}) if os.path.exists(fast)
                             else {"error": "libhip_rrtmgp_fast.so not built (make -C rrtmgp.jl_amd/csrc fast)"}),
                # the headline's columns are all sunlit and overcast where cloudy (the reference benchmark's choice); a GCM's
                # are not: half of them at night (the SW solve is skipped there), partial cloud fractions (McICA draws)
                "gcm_mix": run_leg("gcm_mix", ["--gcm-mix"]),
                "f64": run_leg("f64", ["--dtype", "f64"]),
                "aerosols": run_leg("aerosols", ["--aerosols"]),
                "clear_sky_diag": run_leg("clear_sky_diag", ["--clear-sky-diag", "one-pass"]),
                # ... with MERRA aerosols: AllSkyRadiationWithClearSkyDiagnostics(aerosol_radiation = true), what ClimaAtmos runs
                "clear_sky_diag_aerosols": run_leg("clear_sky_diag_aerosols", ["--clear-sky-diag", "one-pass", "--aerosols"]),
                "ncol_1048576": run_leg("ncol_1048576", ["--tile", "8", "--steps", "3", "--warmup", "1"]),
                # BASELINE config 2's solvers and precision at bench size: the reference's clear-sky pairing, Float64
                "noscat_clear_f64": run_leg("noscat_clear_f64", ["--lw-solver", "noscat", "--angles", "1", "--no-clouds",
                                                                 "--dtype", "f64", "--nlay", "60"]),
                "noscat_lw_f32_3angles": run_leg("noscat_lw_f32_3angles", ["--lw-solver", "noscat", "--angles", "3",
                                                                           "--no-clouds", "--lw-only"]),
                # BASELINE config 4 at its own size (test/all_sky_with_aerosols_highres_gpu_benchmark.jl:285,321-326):
                # 4096 columns = 4 per resident workgroup, so the launch is one round of the persistent grid + its tail
                "config4_4096x72_aerosols": run_leg("config4_4096x72_aerosols", ["--ncol", "4096", "--nlay", "72", "--aerosols",
                                                                                 "--steps", "50", "--warmup", "5"]),
                # the same batch as ONE rrtmgp_hip_update_fluxes call on device arrays (LW + SW + net sum; a short step runs the
                # two solvers on the workspace's two lanes)
                "config4_fused_step_device": run_leg("config4_fused_step_device", ["--ncol", "4096", "--nlay", "72", "--aerosols",
                                                                                   "--fused-step", "--steps", "50", "--warmup", "5"]),
            }
            out["a100_shape"] = a100_shape_legs()
            # BASELINE config 4 is a STRONG-scaling case: 4096 columns over 1 -> 8 GPUs, i.e. 4096 / 2048 / 1024 / 512 columns
            # per GPU.  The driver measures the curve when it has an 8-GPU node; this is its one-GPU prediction: the shard each
            # rank would own, solved alone on this GPU (columns are independent and there is no collective, so N ranks take
            # the time of one shard).  min / median of 50 steps.
            se = {}
            for n_g in (1, 2, 4, 8):
                # (ONE rrtmgp_hip_update_fluxes call per step on device arrays — what update_fluxes! of a resident solver
                # makes: short steps run the LW and SW grids side by side on the workspace's two lanes)
                leg = run_leg(f"strong_emulated_{4096 // n_g}", ["--ncol", str(4096 // n_g), "--nlay", "72", "--aerosols",
                                                                  "--fused-step", "--steps", "50", "--warmup", "5"])
                if "error" not in leg:
                    leg = {"shard_columns": 4096 // n_g, "min_ms": leg["min_ms"], "median_ms": leg["median_ms"],
                           "per_gpu_columns_per_s_at_median": (4096 // n_g) / (leg["median_ms"] * 1e-3),
                           "predicted_job_columns_per_s": 4096 / (leg["median_ms"] * 1e-3)}
                se[str(n_g)] = leg
            ok = all("error" not in v for v in se.values())
            out["strong_emulated"] = {
                "workload": "BASELINE config 4: 4096 columns x 72 layers, MERRA aerosols, Float32, split over N GPUs",
                "gpus": se,
                "predicted_efficiency_vs_1": ({k: v["predicted_job_columns_per_s"] / (int(k) * se["1"]["predicted_job_columns_per_s"])
                                               for k, v in se.items()} if ok else None),
                "per_column_rate_512_vs_4096": (se["8"]["per_gpu_columns_per_s_at_median"] / se["1"]["per_gpu_columns_per_s_at_median"]
                                                if ok else None),
                "note": "emulated on ONE GPU (no multi-GPU hardware was touched): each entry is the shard of one rank solved alone"}
        # CPU baseline: the plain-C oracle (a port, not the Julia reference) on a bounded sample of
        # the same workload, on this box's host cores.  Rank 0, N = 1 only.
        sample = args.cpu_sample if args.cpu_sample is not None else (512 if world == 1 else 0)
        if sample > 0 and world == 1 and not noscat and not args.lw_only and not strong:
            from oracle import oracle as O
            O.lib()

            def cpu_run(n):
                cas, clb, csb = S.make_columns(n, nlay, ft, seed=2026, col_offset=0, clouds=clouds,
                                               cld_frac=args.cld_frac, aerosols=args.aerosols, **mix)
                tc = time.perf_counter()
                O.solve_lw(cas, clb, lw, cl, al, seed=2026)
                O.solve_sw(cas, csb, sw, cs, asw, seed=2026)
                return time.perf_counter() - tc
            cpu_run(256)                                         # thread pool and page faults of the first call
            probe = cpu_run(1024)                                # sizes the sample to ~15 s of CPU work
            n = int(min(max(1024, sample if args.cpu_sample else 15.0 * 1024 / probe), 65536, ncol))
            tc = cpu_run(n)
            if not args.cpu_sample and tc < 10.0 and n < min(65536, ncol):   # the probe over-estimated: once more, larger
                n = int(min(n * min(6.0, 15.0 / tc), 65536, ncol))
                tc = cpu_run(n)
            cpu_model = None
            try:
                with open("/proc/cpuinfo") as fh:
                    cpu_model = next((ln.split(":", 1)[1].strip() for ln in fh if ln.startswith("model name")), None)
            except OSError:
                pass
            out["cpu_baseline"] = {"value": n / tc, "unit": "columns/s", "cores": min(O.n_threads(), 32), "kind": "port",
                                   "threads": min(O.n_threads(), 32), "cpu_model": cpu_model, "host_logical_cpus": os.cpu_count(),
                                   "sample": f"{n} columns of the same workload, oracle/rrtmgp_oracle.c "
                                             f"(gcc -O2, OpenMP over columns), {tc:.1f} s"}
        if args.host:
            # a pipelined / sharded host solve is many launches: the library's kernel timer holds the last chunk only, so
            # per-kernel rates would be fractions of nothing
            for key in ("roofline", "valu", "kernels"):
                out[key] = None
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
