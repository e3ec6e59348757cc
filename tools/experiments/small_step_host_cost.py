#!/usr/bin/env python3
"""Round 5: is a short fused step (BASELINE config 4's strong-scaling shards: 512 ... 4096 columns x 72, aerosols) bound by the
GPU or by the HOST side of the call?  Per step: time the Python mirror spends inside rte.update_fluxes (descriptor building +
the library's launches, no synchronisation) against the wall time per step of a back-to-back loop."""
import sys, time, os
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
import rrtmgp_jl_amd  # noqa: F401
from rrtmgp_jl_amd import rte, synthetic as S

ft = np.float32
lw, sw = S.make_gas_lookup("lw", ft), S.make_gas_lookup("sw", ft)
cl, cs = S.make_cloud_lookup("lw", lw.n_bnd, ft), S.make_cloud_lookup("sw", sw.n_bnd, ft)
AER = os.environ.get("AEROSOLS", "1") == "1"
NLAY = int(os.environ.get("NLAY", "72"))
al, asw = (S.make_aerosol_lookup("lw", lw.bnd_lims_wn, ft), S.make_aerosol_lookup("sw", sw.bnd_lims_wn, ft)) if AER else (None, None)
dev = torch.device("cuda", 0)
for ncol in [int(x) for x in os.environ.get('NCOLS', '512,1024,2048,4096').split(',')]:
    as_h, lb_h, sb_h = S.make_columns(ncol, NLAY, ft, seed=2026, aerosols=AER, cos_zenith=0.86)
    as_d, lb_d, sb_d = as_h.to_device(dev), lb_h.to_device(dev), sb_h.to_device(dev)
    ws = rte.Workspace(ncol, NLAY, ft, 0)
    lws = rte.TwoStreamLWRTE(ncol, NLAY, ft, lb_d, flux_device=dev, workspace=ws)
    sws = rte.TwoStreamSWRTE(ncol, NLAY, ft, sb_d, flux_device=dev, workspace=ws)
    net = torch.empty((ncol, NLAY + 1), dtype=torch.float32, device=dev)
    d = [rte.DeviceLookup(x, 0) if x is not None else None for x in (lw, sw, cl, cs, al, asw)]

    def step():
        rte.update_fluxes(lws, sws, as_d, d[0], d[1], d[2], d[3], d[4], d[5], seed=1, net_flux=net)
    for _ in range(10):
        step()
    ws.synchronize()
    n = 200
    host = 0.0
    t0 = time.perf_counter()
    for _ in range(n):
        t = time.perf_counter(); step(); host += time.perf_counter() - t
    ws.synchronize()
    wall = time.perf_counter() - t0
    # GPU-side time of one step alone: synchronise around each call
    g = []
    for _ in range(50):
        ws.synchronize(); t = time.perf_counter(); step(); ws.synchronize(); g.append(time.perf_counter() - t)
    print(f"ncol {ncol:5d}: wall {1e6 * wall / n:7.1f} us/step, host inside the call {1e6 * host / n:7.1f} us, "
          f"one step alone (call + sync) min {1e6 * min(g):7.1f} us")
