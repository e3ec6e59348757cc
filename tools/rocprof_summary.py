#!/usr/bin/env python3
"""Summarise rocprofv3 rocpd databases (kernel trace + PMC passes) written by tools/profile.sh
into a small text file that can be committed under profiles/.

    python tools/rocprof_summary.py gpurun_out/prof_<tag> > profiles/<name>.txt
"""
import glob
import os
import sqlite3
import sys


def q(dbp, sql):
    db = sqlite3.connect(dbp)
    try:
        return db.execute(sql).fetchall()
    finally:
        db.close()


def kernel_source_sha256(repo):
    """Hash of the sources the two column kernels are compiled from: bench.py only folds a profile's HBM counters into
    its line when this matches the sources of the library it is timing."""
    import hashlib
    h = hashlib.sha256()
    for name in ("common.h", "device.h", "solve_lw.hip", "solve_sw.hip", "Makefile"):
        with open(os.path.join(repo, "rrtmgp.jl_amd", "csrc", name), "rb") as fh:
            h.update(name.encode() + b"\0" + fh.read())
    return h.hexdigest()


def to_json(root, path, sha=None):
    """Per-kernel averages (duration in us, PMC counters per launch) as JSON for bench.py's `traffic`."""
    import json
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = {"source": os.path.basename(root.rstrip("/")), "git_sha": sha or None,
           "kernel_source_sha256": kernel_source_sha256(repo), "kernels": {}}
    tr = os.path.join(root, "trace", "bench_results.db")
    if os.path.exists(tr):
        for name, calls, tot, avg, pct in q(tr, "select name, total_calls, total_duration, average, percentage from top_kernels"):
            for key in ("lw_solve_kernel", "sw_solve_kernel", "lw_noscat_kernel"):
                if key in name:
                    out["kernels"].setdefault(key, {})["avg_us"] = avg
                    out["kernels"][key]["calls"] = calls
    for d in sorted(glob.glob(os.path.join(root, "pmc_*", "bench_results.db"))):
        for k, c, s, n in q(d, "select kernel_name, counter_name, sum(value), count(*) from counters_collection "
                               "where (kernel_name like '%solve_kernel%' or kernel_name like '%noscat_kernel%') group by kernel_name, counter_name"):
            for key in ("lw_solve_kernel", "sw_solve_kernel", "lw_noscat_kernel"):
                if key in k:
                    out["kernels"].setdefault(key, {})[c] = s / n
    with open(path, "w") as fh:
        json.dump(out, fh, indent=1, sort_keys=True)


def main(root):
    print(f"# rocprofv3 summary of {os.path.basename(root.rstrip('/'))}" + (f" at git {sys.argv[3]}" if len(sys.argv) > 3 else ""))
    tr = os.path.join(root, "trace", "bench_results.db")
    if os.path.exists(tr):
        print("\n## kernel trace (--kernel-trace --stats): name, calls, total_us, avg_us, pct")
        for name, calls, tot, avg, pct in q(tr, "select name, total_calls, total_duration, average, percentage from top_kernels limit 8"):
            print(f"{name[:90]:90s} {calls:6d} {tot:14.1f} {avg:12.1f} {pct:7.2f}")
        print("\n## dispatch geometry: name, grid, workgroup, lds_size, vgpr, accum_vgpr, sgpr, scratch")
        for r in q(tr, "select name, grid_x, workgroup_x, lds_size, vgpr_count, accum_vgpr_count, sgpr_count, scratch_size "
                       "from kernels where (name like '%solve_kernel%' or name like '%noscat_kernel%') group by name"):
            print("  ", tuple(x if not isinstance(x, str) else x[:60] for x in r))
    print("\n## PMC passes (--pmc ...): kernel, counter, per-launch average, launches")
    for d in sorted(glob.glob(os.path.join(root, "pmc_*", "bench_results.db"))):
        for k, c, s, n in q(d, "select kernel_name, counter_name, sum(value), count(*) from counters_collection "
                               "where (kernel_name like '%solve_kernel%' or kernel_name like '%noscat_kernel%') group by kernel_name, counter_name"):
            print(f"{k.replace('void rrtmgp::', '')[:56]:56s} {c:28s} {s / n:20.1f} {n:4d}")


if __name__ == "__main__":
    main(sys.argv[1])
    if len(sys.argv) > 2:
        to_json(sys.argv[1], sys.argv[2], sys.argv[3] if len(sys.argv) > 3 else None)
