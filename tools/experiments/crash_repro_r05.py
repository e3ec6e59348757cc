import sys, numpy as np
sys.path.insert(0, '/root/repo')
import rrtmgp_jl_amd
from rrtmgp_jl_amd import rte, synthetic as S
lw = S.make_gas_lookup("lw", np.float64); cl = S.make_cloud_lookup("lw", lw.n_bnd)
as_, lb, _ = S.make_columns(4, 37, np.float64, seed=8)
slv = rte.NoScatLWRTE(4, 37, np.float64, lb, n_gauss_angles=3)
out = rte.solve_lw(slv, as_, lw, cl)
print("ok", float(out.flux_up[0, 0]))
