#!/usr/bin/env python3
"""Latency of one small call on a 2-shard workspace (both shards on GPU 0): persistent shard workers (default) against a
thread spawned and joined per shard and call (RRTMGP_HIP_SPAWN_SHARD_THREADS=1, the round-2 fan-out).

    python tools/experiments/shard_call_latency.py            # prints both (each in a child process)
"""
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def child():
    import numpy as np
    import rrtmgp_jl_amd  # noqa: F401
    from rrtmgp_jl_amd import rte, synthetic as S
    from rrtmgp_jl_amd.states import RRTMGPParameters
    ncol, nlay = 2, 60
    as_, lb, sb = S.make_columns(ncol, nlay, np.float64, seed=1)
    lw = S.make_gas_lookup("lw", np.float64)
    ws = rte.Workspace(ncol, nlay, np.float64, [0, 0])
    dl = rte.DeviceLookup(lw, [0, 0])
    slv = rte.NoScatLWRTE(ncol, nlay, np.float64, lb, workspace=ws)
    params = RRTMGPParameters()
    out = np.empty((nlay, ncol), order="F")
    for name, fn, n in (("compute_col_gas (1 column per shard)", lambda: rte.compute_col_gas(ws, as_.p_lev, params, out=out), 3000),
                        ("NoScatLW solve (1 column per shard)", lambda: rte.solve_lw(slv, as_, dl), 1000)):
        for _ in range(50):
            fn()
        t = time.perf_counter()
        for _ in range(n):
            fn()
        print(f"  {name}: {1e6 * (time.perf_counter() - t) / n:.1f} us per call")


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "child":
        child()
    else:
        for label, env in (("persistent workers", {}), ("spawn + join per call", {"RRTMGP_HIP_SPAWN_SHARD_THREADS": "1"})):
            print(label)
            sys.stdout.flush()
            subprocess.run([sys.executable, os.path.abspath(__file__), "child"], env=dict(os.environ, **env), check=True)
