"""MI355X-native device back end for RRTMGP.jl's radiative-transfer hot path.

The directory is named `rrtmgp.jl_amd`; import it as `rrtmgp_jl_amd` (the
top-level `rrtmgp_jl_amd.py` shim registers this directory under that name).

Layout:
  csrc/        HIP kernels (gfx950) and the C-ABI implementation -> libhip_rrtmgp.so
  _abi.py      ctypes mirror of include/rrtmgp_hip.h
  _lib.py      loader for libhip_rrtmgp.so (fails loudly when it is missing)
  lookups.py   LookUpLW / LookUpSW / LookUpCld / LookUpAerosolMerra containers
  states.py    AtmosphericState, CloudState, AerosolState, Vmr(GM), BCs, Flux, gray state
  rte.py       NoScatLWRTE / TwoStreamLWRTE / NoScatSWRTE / TwoStreamSWRTE + solve_lw / solve_sw
  solver.py    RRTMGPSolver / update_fluxes / getters (Layer-2 API surface)
  sharding.py  contiguous column sharding over ranks (one process per GPU)
  synthetic.py seeded synthetic tables and columns
"""
__version__ = "0.1.0"
