# Run with a Julia that has RRTMGP.jl (patched as julia/README.md says), HIPRRTMGP and an MI355X.  These are the checks of the
# reference's own suite that need a running Julia (test/standalone.jl:294-335 Adapt topology is CPU-side and unchanged;
# :361-383 zero allocation and type stability are what the HIP path has to keep).
using Test, RRTMGP, HIPRRTMGP
import ClimaComms

HIP = HIPRRTMGP.extension()
device = HIP.HIPDevice(0)
context = ClimaComms.SingletonCommsContext(device)

@testset "gray Layer 2 on a HIPDevice" begin
    for FT in (Float32, Float64)
        hip = RRTMGP.solve_gray(FT; nlay = 60, ncol = 10, context).solver
        cpu = RRTMGP.solve_gray(FT; nlay = 60, ncol = 10, context = ClimaComms.SingletonCommsContext(ClimaComms.CPUSingleThreaded())).solver
        RRTMGP.update_fluxes!(hip)
        RRTMGP.update_fluxes!(cpu)
        tol = FT === Float64 ? 1e-8 : 1e-2
        @test maximum(abs, RRTMGP.net_flux(hip) .- RRTMGP.net_flux(cpu)) < tol
        @test (@inferred RRTMGP.net_flux(hip)) isa SubArray
        @allocated RRTMGP.update_fluxes!(hip)
        @test (@allocated RRTMGP.update_fluxes!(hip)) == 0
        @test (@allocated RRTMGP.update_fluxes!(hip, 1234)) == 0
        @test (@allocated RRTMGP.heating_rate(hip)) <= sizeof(FT) * 60 * 10 + 256   # its result array, nothing else
    end
end

@testset "page-locking has the array's lifetime" begin
    a = zeros(Float32, 64, 1 << 18)               # 64 MB
    @test HIP.pin!(a) == 1
    @test HIP.pin!(a) == 1                        # reference counted per exact range
    a = nothing
    GC.gc(); GC.gc()
    HIP.retry_parked_slow()
    @test isempty(HIP.PARKED)
end
HIP.release_all!()
