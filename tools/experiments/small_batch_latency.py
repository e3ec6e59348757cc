"""Runs on the GPU box: wall time of one LW + SW all-sky solve for small batches, host arrays vs device-resident."""
import sys, time
import numpy as np
sys.path.insert(0, ".")
import torch
import rrtmgp_jl_amd  # noqa: F401
from rrtmgp_jl_amd import rte, synthetic as S

ft = np.float32
lw, sw = S.make_gas_lookup("lw", ft), S.make_gas_lookup("sw", ft)
cl, cs = S.make_cloud_lookup("lw", lw.n_bnd, ft), S.make_cloud_lookup("sw", sw.n_bnd, ft)
dl, ds, dcl, dcs = (rte.DeviceLookup(x, 0) for x in (lw, sw, cl, cs))
for ncol in (1, 16, 128, 1024, 4096):
    as_, lb, sb = S.make_columns(ncol, 64, ft, seed=1, clouds=True)
    for where in ("host", "device"):
        if where == "device":
            dev = torch.device("cuda", 0)
            a, l, s = as_.to_device(dev), lb.to_device(dev), sb.to_device(dev)
            fd = dev
        else:
            a, l, s, fd = as_, lb, sb, None
        slw = rte.TwoStreamLWRTE(ncol, 64, ft, l, flux_device=fd)
        ssw = rte.TwoStreamSWRTE(ncol, 64, ft, s, flux_device=fd)
        def step():
            rte.solve_lw(slw, a, dl, dcl, seed=1)
            rte.solve_sw(ssw, a, ds, dcs, seed=1)
            if where == "device":
                slw.ws.synchronize(); ssw.ws.synchronize()
        for _ in range(5): step()
        t = time.perf_counter()
        n = 50
        for _ in range(n): step()
        dt = (time.perf_counter() - t) / n
        print(f"ncol {ncol:5d} {where:6s}: {dt * 1e6:8.1f} us per LW+SW step")
