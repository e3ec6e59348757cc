"""The committed rocprofv3 summary that bench.py folds into its line (`profiles/latest.json`): it must belong to the kernel
sources in the tree — bench.py withholds `roofline.traffic` and `vmem_pipeline` otherwise — and carry the counters both
need.  CPU only: nothing here touches a GPU."""
import importlib.util
import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    spec = importlib.util.spec_from_file_location("bench_module", os.path.join(ROOT, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def _profile():
    with open(os.path.join(ROOT, "profiles", "latest.json")) as fh:
        return json.load(fh)


def test_latest_profile_was_taken_on_the_kernel_sources_in_the_tree():
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    from rocprof_summary import kernel_source_sha256
    pj = _profile()
    if pj["kernel_source_sha256"] != kernel_source_sha256(ROOT):
        # not a failure of the product: bench.py then reports `traffic` and `vmem_pipeline` as null, by design
        pytest.skip("kernel sources changed after profiles/latest.json was taken: bench.py withholds its counters until "
                    "tools/profile2.sh has been re-run and its summary committed")
    # the human-readable summary of the same run is committed next to it
    assert os.path.exists(os.path.join(ROOT, "profiles", pj["source"].replace("prof_", "") + "_kernels.txt"))


def test_profile_carries_the_counters_of_traffic_and_vmem_pipeline():
    pj, bench = _profile(), _bench()
    for name in ("lw_solve_kernel", "sw_solve_kernel"):
        k = pj["kernels"][name]
        for c in ("FETCH_SIZE", "WRITE_SIZE", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR", "GRBM_GUI_ACTIVE", "avg_us"):
            assert k.get(c), (name, c)
        v = bench.vmem_pipeline(k)
        # a share of the kernel's time: positive, below 1, and the two instruction classes add up to the counters
        assert 0.3 < v["frac"] < 1.0, v
        assert v["instructions_4_byte"] + v["instructions_16_byte"] == k["SQ_INSTS_VMEM_RD"] + k["SQ_INSTS_VMEM_WR"]
        # HBM-side bytes per launch stay far above the algorithmic 0.5 GB (the sweep scratch) and far below HBM's reach
        traffic = (2.0 * k["FETCH_SIZE"] + k["WRITE_SIZE"]) * 1024.0
        assert 2e10 < traffic < 2e11, traffic
    assert bench.vmem_pipeline({"SQ_INSTS_VMEM_RD": 1.0}) is None   # an incomplete profile gives no number


def test_roofline_fraction_is_reproducible_from_the_committed_profile():
    """`roofline.profiled` of the bench line = the dominant kernel's SURVEY section 8(d) flops over the profile's average
    duration, against the FP32 vector peak: recomputed here from profiles/latest.json alone, and compared with the committed
    bench line of the round when that line was produced from this profile."""
    pj, bench = _profile(), _bench()
    ncol, nlay = bench.NCOL_PER_GPU, bench.NLAY
    for name, ngpt, cell in (("lw_solve_kernel", 256, bench.LW_FLOPS_PER_CELL), ("sw_solve_kernel", 224, bench.SW_FLOPS_PER_CELL)):
        ms = pj["kernels"][name]["avg_us"] / 1e3
        v = bench.kernel_valu(name, ms, ncol, nlay, ngpt)
        assert abs(v["achieved"] - cell * ngpt * nlay * ncol / (ms * 1e-3) / 1e12) < 1e-9
        assert abs(v["frac"] - v["achieved"] / bench.VALU_PEAK_TFLOPS) < 1e-12 and 0.15 < v["frac"] < 0.6
    lines = sorted(f for f in os.listdir(os.path.join(ROOT, "profiles")) if f.endswith("_bench_line.json"))
    with open(os.path.join(ROOT, "profiles", lines[-1])) as fh:
        line = json.load(fh)
    r = line.get("roofline") or {}
    prof = r.get("profiled")
    if not prof or (r.get("hbm", {}).get("traffic_source") or {}).get("profile") != pj["source"]:
        pytest.skip("the newest committed bench line was not produced from profiles/latest.json")
    dom = r["kernel"]
    v = bench.kernel_valu(dom, pj["kernels"][dom]["avg_us"] / 1e3, ncol, nlay, 256 if dom.startswith("lw") else 224)
    assert abs(v["frac"] - prof["frac"]) < 1e-9 and r["bound"] == "valu"



LEG_KERNELS = {"clear_sky_diag": ("lw_solve_kernel", "sw_solve_kernel"), "clear_sky_diag_aerosols": ("lw_solve_kernel", "sw_solve_kernel"),
               "f64": ("lw_solve_kernel", "sw_solve_kernel"), "noscat_clear_f64": ("lw_noscat_kernel", "sw_solve_kernel"),
               "aerosols": ("lw_solve_kernel", "sw_solve_kernel")}


@pytest.mark.parametrize("leg", sorted(LEG_KERNELS))
def test_variant_legs_have_a_committed_profile_that_prices_their_kernels(leg):
    """Round 6: the production (one-pass diagnostic, with and without aerosols), Float64 and no-scattering instances have
    their own rocprofv3 summaries (tools/profile_legs.sh -> profiles/latest_<leg>.json); bench.py's `leg_profile` turns the
    kernel-trace averages into the leg's `profiled` block with the formula of `roofline.profiled`."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    from rocprof_summary import kernel_source_sha256
    path = os.path.join(ROOT, "profiles", f"latest_{leg}.json")
    assert os.path.exists(path)
    pj = json.load(open(path))
    assert os.path.exists(os.path.join(ROOT, "profiles", pj["source"].replace("prof_", "") + "_kernels.txt"))
    for k in LEG_KERNELS[leg]:
        assert pj["kernels"][k]["avg_us"] > 1000 and pj["kernels"][k]["FETCH_SIZE"] > 0 and pj["kernels"][k]["WRITE_SIZE"] > 0
    bench = _bench()
    f64 = "f64" in leg
    nlay = 60 if leg == "noscat_clear_f64" else 64
    peak = bench.FP64_VALU_PEAK_TFLOPS if f64 else bench.VALU_PEAK_TFLOPS
    lw_cell = (bench.LW_NOSCAT_FLOPS_PER_CELL + bench.LW_NOSCAT_FLOPS_PER_ANGLE) if leg == "noscat_clear_f64" else bench.LW_FLOPS_PER_CELL
    rec = bench.leg_profile(leg, bench.NCOL_PER_GPU, nlay, 256, 224, peak, lw_cell)
    if pj["kernel_source_sha256"] != kernel_source_sha256(ROOT):
        assert "withheld" in rec["note"]
        pytest.skip("kernel sources changed after this leg was profiled: bench.py withholds the block")
    assert set(rec["kernels"]) == set(LEG_KERNELS[leg])
    for name, r in rec["kernels"].items():
        assert 0.05 < r["frac"] < 0.6 and r["traffic"] > 1e10, (name, r)
    assert abs(rec["step"]["kernel_ms"] - sum(r["kernel_ms"] for r in rec["kernels"].values())) < 1e-9
