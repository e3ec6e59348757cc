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
@testset "device-resident arrays (HIPDevice(0; resident = true))" begin
    import Adapt
    rdev = HIP.HIPDevice(0; resident = true)
    @test ClimaComms.array_type(rdev) === HIP.HIPArray
    @test_throws ErrorException HIP.HIPDevice([0, 0]; resident = true)        # one device only
    # the array type itself: construction, copies both ways, fill, scalar broadcast, views, Adapt
    h = rand(Float32, 7, 5)
    d = HIP.HIPArray(h)
    @test d isa DenseArray{Float32, 2} && size(d) == (7, 5) && HIP.mem(d) == 1 && HIP.mem(h) == 0
    @test Array(d) == h
    @test Array(copy(d)) == h
    @test all(==(0), Array(fill!(similar(d), 0)))
    @test all(==(2.5f0), Array(fill!(HIP.HIPArray{Float32}(undef, 3, 4), 2.5)))
    e = HIP.HIPArray{Float64, 3}(undef, 2, 3, 4); e .= 0
    @test all(==(0), Array(e))
    v = view(d, 2:4, :)
    @test HIP.mem(v) == 1 && pointer(v) == pointer(d) + sizeof(Float32) && strides(v) == (1, 7)
    @test_throws ErrorException d[1, 1]                                      # no element access over PCIe
    @test_throws ErrorException (d .= d .+ 1)
    @test Adapt.adapt(Array, d) == h && Adapt.adapt(HIP.HIPArray, h) isa HIP.HIPArray
    # The spectral solver (needs the lookup tables: `using NCDatasets` + the rrtmgp-data artifact): same bits as the
    # host-array solver on the same device, views from the getters, the whole-solver Adapt round trip of
    # test/standalone.jl:294-335, zero allocation.  update_fluxes! of a spectral solver on a HIPDevice is ONE library call.
    have_tables = try
        @eval using NCDatasets
        true
    catch
        false
    end
    if have_tables
        for FT in (Float32, Float64)
            prof = RRTMGP.standard_atmosphere(FT; kind = :tropical, nlay = 40, ncol = 10)
            rctx = ClimaComms.SingletonCommsContext(rdev)
            res = RRTMGP.solve(prof; context = rctx).solver
            hst = RRTMGP.solve(prof; context).solver
            RRTMGP.update_fluxes!(res, 7); RRTMGP.update_fluxes!(hst, 7)
            @test parent(RRTMGP.net_flux(res)) isa HIP.HIPArray
            @test Array(parent(RRTMGP.net_flux(res))) == parent(RRTMGP.net_flux(hst))
            back = Adapt.adapt(HIP.HIPArray, Adapt.adapt(Array, res))
            RRTMGP.update_fluxes!(back, 7)
            @test Array(parent(RRTMGP.net_flux(back))) == parent(RRTMGP.net_flux(hst))
            @test (@allocated RRTMGP.update_fluxes!(res)) == 0
        end
    else
        @info "spectral lookups unavailable (NCDatasets / rrtmgp-data): the resident-solver lane was skipped"
    end
    # Gray radiation keeps the reference's generic update_fluxes! (documented in the extension): its presentation copies and
    # net sums are array broadcasts (Fluxes.jl:408,424), which a bare HIPArray refuses with a message instead of crawling
    gray = RRTMGP.solve_gray(Float64; nlay = 20, ncol = 4, context = ClimaComms.SingletonCommsContext(rdev)).solver
    @test_throws ErrorException RRTMGP.update_fluxes!(gray)
end
HIP.release_all!()
