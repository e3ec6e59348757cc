"""
    HIPRRTMGP

Trigger package of RRTMGP.jl's `RRTMGPHIPExt` extension (the MI355X back end): loading it next to RRTMGP makes Julia load
`ext/RRTMGPHIPExt.jl`, exactly as `using CUDA` loads `RRTMGPCUDAExt` (reference `Project.toml:14-22`).  Its only content is
where the shared library lives.

    julia> using RRTMGP, HIPRRTMGP
    julia> HIP = HIPRRTMGP.extension()          # the extension module (HIPDevice, pin!, release_all!)
    julia> dev = HIP.HIPDevice(0)

`libpath` is, in this order: `ENV["RRTMGP_HIP_LIBRARY"]`, `<this package>/../../rrtmgp.jl_amd/libhip_rrtmgp.so` (the
in-tree build of this repository: `make -C rrtmgp.jl_amd/csrc`), or plain `"libhip_rrtmgp.so"` for the dynamic loader's
search path.
"""
module HIPRRTMGP

const intree = normpath(joinpath(@__DIR__, "..", "..", "..", "rrtmgp.jl_amd", "libhip_rrtmgp.so"))
const libpath = get(ENV, "RRTMGP_HIP_LIBRARY", isfile(intree) ? intree : "libhip_rrtmgp.so")

"""
    extension()

The loaded `RRTMGPHIPExt` module (`Base.get_extension`), or an error saying what is missing.  `HIPDevice` lives inside
the extension: a package extension cannot add names to its parent, so this is how user code reaches it.
"""
function extension()
    rr = get(Base.loaded_modules, Base.PkgId(Base.UUID("a01a1ee8-cea4-48fc-987c-fc7878d79da1"), "RRTMGP"), nothing)
    rr === nothing && error("HIPRRTMGP.extension(): load RRTMGP first (`using RRTMGP`)")
    ext = Base.get_extension(rr, :RRTMGPHIPExt)
    ext === nothing && error("HIPRRTMGP.extension(): RRTMGP has no RRTMGPHIPExt extension; apply julia/Project.toml.patch " *
                             "to RRTMGP.jl's Project.toml and copy ext/RRTMGPHIPExt.jl into its ext/ directory")
    return ext
end

function __init__()
    # the extension reads this variable when it is loaded (after this package): one source of truth for the path
    haskey(ENV, "RRTMGP_HIP_LIBRARY") || (ENV["RRTMGP_HIP_LIBRARY"] = libpath)
    return nothing
end

end # module
