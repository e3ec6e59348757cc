import numpy as np, sys
sys.path.insert(0,'.')
import rrtmgp_jl_amd
from rrtmgp_jl_amd import rte, synthetic as S
from oracle import oracle as O
t64 = dict(lw=S.make_gas_lookup("lw"), sw=S.make_gas_lookup("sw")); t64["cl"]=S.make_cloud_lookup("lw",16); t64["cs"]=S.make_cloud_lookup("sw",14)
t32 = {k:v.astype(np.float32) for k,v in t64.items()}
a64,l64,s64 = S.make_columns(96,64,np.float64,seed=3,cos_zenith=0.86)
a32,l32,s32 = S.make_columns(96,64,np.float32,seed=3,cos_zenith=0.86)
def md(a,b,names): return max(float(np.abs(getattr(a,n).astype(np.float64)-getattr(b,n)).max()) for n in names)
f=rte.solve_lw(rte.TwoStreamLWRTE(96,64,np.float32,l32),a32,t32["lw"],t32["cl"])
print("LW F32 HIP vs F64 oracle:", md(f,O.solve_lw(a64,l64,t64["lw"],t64["cl"]),("flux_up","flux_dn","flux_net")), " | oracle F32 vs F64:", md(O.solve_lw(a32,l32,t32["lw"],t32["cl"]),O.solve_lw(a64,l64,t64["lw"],t64["cl"]),("flux_up","flux_dn","flux_net")))
f=rte.solve_sw(rte.TwoStreamSWRTE(96,64,np.float32,s32),a32,t32["sw"],t32["cs"])
print("SW F32 HIP vs F64 oracle:", md(f,O.solve_sw(a64,s64,t64["sw"],t64["cs"]),("flux_up","flux_dn","flux_net","flux_dn_dir")), " | oracle F32 vs F64:", md(O.solve_sw(a32,s32,t32["sw"],t32["cs"]),O.solve_sw(a64,s64,t64["sw"],t64["cs"]),("flux_up","flux_dn","flux_net","flux_dn_dir")))
