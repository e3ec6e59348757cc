# One radiation step of RRTMGP.jl on an MI355X through libhip_rrtmgp.so (julia/README.md: how to install the extension).
# Gray radiation needs no lookup tables; the spectral methods work the same way once `using NCDatasets` has loaded them
# (`lookups = RRTMGP.lookup_tables(grid_params, method)`).
using RRTMGP, HIPRRTMGP
import ClimaComms

HIP = HIPRRTMGP.extension()                       # the RRTMGPHIPExt module: `HIPDevice` lives inside the extension
device = HIP.HIPDevice(0)                         # HIP.HIPDevice([0, 1, 2, 3]): columns sharded over four GPUs, one process
context = ClimaComms.SingletonCommsContext(device)
FT = Float32
grid_params = RRTMGP.RRTMGPGridParams(FT; context, domain_nlay = 64, ncol = 4096)

# the standalone front door of the reference builds state, boundary conditions and solver for a gray atmosphere
# (src/api/standalone.jl); a host model constructs `RRTMGP.RRTMGPSolver(grid_params, method, params, bcs_lw, bcs_sw, as; ...)`
# itself (src/api/solver.jl:136) with arrays from `ClimaComms.array_type(device)` (= Array: host memory the library stages)
solver = RRTMGP.solve_gray(FT; nlay = 64, ncol = 4096, context).solver

HIP.pin!(solver)                                  # page-lock the solver's large arrays once: uploads become real DMA
RRTMGP.update_fluxes!(solver)                     # prepare -> LW -> SW -> net
println("net flux at the top of the first column: ", RRTMGP.net_flux(solver)[end, 1], " W/m^2")
println("heating rate of its lowest layer: ", RRTMGP.heating_rate(solver)[1, 1], " K/s")
HIP.release_all!(device)                          # library handles (also released by finalizers and at exit)
