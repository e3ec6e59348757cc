# deep columns with clouds: HIP vs oracle (nlay 150 / 200 / 130), LW + SW, two-stream + noscat LW
import numpy as np, sys
sys.path.insert(0, '.')
import rrtmgp_jl_amd
from rrtmgp_jl_amd import rte, synthetic as S
from oracle import oracle as O
lw, sw = S.make_gas_lookup("lw", np.float64), S.make_gas_lookup("sw", np.float64)
cl, cs = S.make_cloud_lookup("lw", lw.n_bnd), S.make_cloud_lookup("sw", sw.n_bnd)
for nlay in (130, 150, 200, 257):
    as_, lb, sb = S.make_columns(6, nlay, np.float64, seed=nlay, clouds=True, random_cld_frac=True, night_fraction=0.2)
    for ts in (True, False):
        f = rte.solve_lw((rte.TwoStreamLWRTE if ts else rte.NoScatLWRTE)(6, nlay, np.float64, lb), as_, lw, cl, seed=4)
        r = O.solve_lw(as_, lb, lw, cl, twostream=ts, seed=4)
        print(nlay, 'lw', ts, np.abs(f.as_nlev_ncol('flux_up') - r.flux_up).max(), np.abs(f.as_nlev_ncol('flux_dn') - r.flux_dn).max())
    f = rte.solve_sw(rte.TwoStreamSWRTE(6, nlay, np.float64, sb), as_, sw, cs, seed=4)
    r = O.solve_sw(as_, sb, sw, cs, seed=4)
    print(nlay, 'sw', np.abs(f.as_nlev_ncol('flux_up') - r.flux_up).max(), np.abs(f.as_nlev_ncol('flux_dn') - r.flux_dn).max(), np.abs(np.asarray(as_.cloud_state.cld_cover_sw) - 0).max())
