// solve_sw.hip — shortwave column kernels (two-stream and no-scattering) for gfx950.
//
// Replaces rte_sw_2stream_solve! / rte_sw_noscat_solve! of the reference
// (ext/cuda/rte_shortwave_2stream.jl:58-175, rte_shortwave_noscat.jl:54-112; bodies
// src/rte/shortwave_2stream.jl, shortwave_noscat.jl, src/optics/compute_optical_props.jl:263-388).
//
// One workgroup per column, one lane per g-point, layers in chunks of CH whose band-level
// records are prepared cooperatively in LDS (device.h).  Two-stream runs two sweeps:
//   1. top-down: gas/cloud/aerosol optics, cumulative direct beam, layer coefficients and
//      adding from the top -> (A, B, beta) per level in the scratch;
//   2. bottom-up: U_{k+1} = A_k U_k + B_k, D_{k+1} = beta_{k+1} U_{k+1} + delta_{k+1}.
// Night columns (mu0 <= 0) skip the solve and are zeroed, but still produce the
// cloud-cover and 550 nm AOD diagnostics as the reference does.
#include "device.h"
#include <climits>


namespace rrtmgp {

// sw_2stream_coeffs, src/rte/shortwave_2stream.jl:189-279
// inv_mu0 = 1 / max(mu0, mu0_min), formed once per column
template <typename FT>
__device__ __forceinline__ void sw_2stream_coeffs(FT tau, FT ssa, FT g, FT mu0, FT inv_mu0, FT &Rdir, FT &Tdir, FT &Rdif,
                                                  FT &Tdif) {
    const FT gamma1 = (FT(8) - ssa * (FT(5) + FT(3) * g)) * FT(0.25);
    const FT gamma2 = FT(3) * (ssa * (FT(1) - g)) * FT(0.25);
    const FT gamma3 = (FT(2) - (FT(3) * mu0) * g) * FT(0.25);
    const FT gamma4 = FT(1) - gamma3;
    const FT alpha1 = gamma1 * gamma4 + gamma2 * gamma3;
    const FT alpha2 = gamma1 * gamma3 + gamma2 * gamma4;
    const FT k = m_sqrt_pos(m_max(FT(2) * (FT(1) - ssa) * (gamma1 + gamma2), k_min<FT>()));
    FT exp_minusktau, om1;
    exp_pair(tau * k, exp_minusktau, om1);
    const FT exp_minus2ktau = exp_minusktau * exp_minusktau;
    const FT one_minus_e2kt = om1 * (FT(1) + exp_minusktau);
    FT RT_term = m_rcp(k * (FT(1) + exp_minus2ktau) + gamma1 * one_minus_e2kt);
    Rdif = RT_term * gamma2 * one_minus_e2kt;
    Tdif = RT_term * FT(2) * k * exp_minusktau;
    const FT T0 = m_exp_neg(tau * inv_mu0);
    FT k_mu = k * mu0;
    FT k_mu2 = k_mu * k_mu;
    const FT diff = FT(1) - k_mu2;
    if (m_abs(diff) < resonance_window<FT>()) {
        k_mu2 = diff >= FT(0) ? FT(1) - resonance_window<FT>() : FT(1) + resonance_window<FT>();
        k_mu = m_sqrt_pos(k_mu2);
    }
    const FT k_gamma3 = k * gamma3;
    const FT k_gamma4 = k * gamma4;
    RT_term = m_div(ssa * RT_term, FT(1) - k_mu2);
    const FT Rdir_u = RT_term * ((FT(1) - k_mu) * (alpha2 + k_gamma3) -
                                 (FT(1) + k_mu) * (alpha2 - k_gamma3) * exp_minus2ktau -
                                 FT(2) * (k_gamma3 - alpha2 * k_mu) * exp_minusktau * T0);
    const FT Tdir_u = -RT_term * ((FT(1) + k_mu) * (alpha1 + k_gamma4) * T0 -
                                  (FT(1) - k_mu) * (alpha1 - k_gamma4) * exp_minus2ktau * T0 -
                                  FT(2) * (k_gamma4 + alpha1 * k_mu) * exp_minusktau);
    Rdir = m_max(FT(0), Rdir_u);
    Tdir = m_max(FT(0), Tdir_u);
    const FT av_energy = m_max(FT(0), FT(1) - T0);
    const FT tot_dir = Rdir + Tdir;
    if (tot_dir > av_energy) {
        const FT scale = m_div(av_energy, m_max(Num<FT>::eps(), tot_dir));
        Rdir *= scale;
        Tdir *= scale;
    }
}

template <typename FT>
struct SwArgs {
    DevGas<FT> lk;
    DevCld<FT> cld;
    DevAero<FT> aero;
    DevState<FT> as;
    DevFlux<FT> fl;
    const FT *cos_zenith, *toa_flux, *alb_dir, *alb_dif;
    FT *scratch;
    int *queue;  // {next column of the persistent grid, workgroups done}: device counters, zero between launches (queue_release)
    ColDims dims;
    uint64_t seed;
    int64_t col_offset;
};

constexpr int DB = 16;  // levels per batch of the light sweeps

// DIAG: clear-sky recurrences carried next to the all-sky ones (see lw_solve_kernel)
// CA: see lw_solve_kernel
// HALF: the main (no-aerosol) instances once more with 8-layer chunks, for columns whose 16-layer records would push a
// workgroup past a quarter of the CU's LDS (Float32, 71-80 layers): 4 resident workgroups instead of 3.  Float32 DIAG
// instances: once more with 8-layer chunks AND at 128 VGPRs (4 waves per SIMD), taken when that admits one more workgroup.
template <typename FT, bool TWOSTREAM, bool BAND, bool DIAG, int CA = -1, bool HALF = false>
__global__ void __launch_bounds__(256, (sizeof(FT) == 4 ? (DIAG ? (HALF ? 4 : RR_DIAG_MIN_WAVES) : (HALF ? RR_SW_HALF_WAVES : RR_MIN_WAVES)) : HALF ? RR_F64_HALF_WAVES : 2)) sw_solve_kernel(const SwArgs<FT> a) {
    extern __shared__ __align__(16) char smem[];
    constexpr int CHK = HALF ? half_chunk_layers<FT>() : chunk_layers(CA, DIAG);  // layers per chunk of LDS records
    ColShared<FT, CHK> sh;
    ColDims dd = a.dims;
    if (CA >= 0) { dd.has_cld = CA & 1; dd.has_aero = (CA >> 1) & 1; }
    dd.diag = DIAG;  // what the host set, as a constant: the other flux set's pointers are never loaded
    dd.lw = 0; dd.twostream = TWOSTREAM;   // likewise (the preparation steps branch on them)
    const ColDims &d = dd;
    carve_shared(sh, smem, d);
    const int nlay = d.nlay, nlev = d.nlev, ncol = a.as.ncol, nb = d.nbnd;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    // lane -> g-point: whole bands dealt to the wavefronts by minor-gas slot count (build_gas: lane_gpt; the identity for
    // ragged bands), or (per-band fluxes) the band-by-band layout on 16-lane rows
    const int gl = BAND ? a.lk.band_lane_gpt[tid] : a.lk.lane_gpt[tid];
    const bool active = gl >= 0;
    const int g = active ? gl : a.lk.n_gpt - 1;
    const LaneBand lb = lane_band(a.lk, g);
    constexpr int NV = DIAG ? 6 : 3;  // sweep values per level
    constexpr int NA = DIAG ? 6 : 3;  // accumulated components per level: up, dn, dir (+ the clear-sky three)
    const Sweep<FT, NV> sw(a.scratch + (size_t)blockIdx.x * (size_t)nlev * NV * SWEEP_LANES, (size_t)nlev * NV * SWEEP_LANES * sizeof(FT), (unsigned)tid);
    const FT amask = active ? FT(1) : FT(0);
    const FT solar_frac = a.lk.solar_src_scaled[g];
    const int nchunk = (nlay + CHK - 1) / CHK;
    cache_small_tables(sh, d, a.lk, a.as);
    const bool want_aod = d.has_aero && a.aero.iband_550nm > 0 && a.as.aod_sw_ext != nullptr;
    // Everything below that is not an operand of the layer loops is read from the kernarg segment where it is used
    // (device.h, "cold kernel arguments"): offsets of the members of the one kernel argument
#define RR_ARG(member) ((unsigned)offsetof(SwArgs<FT>, member))
    // (one instance keeps the by-value form: clouds + aerosols without the diagnostic, whose layer loop the register
    // allocator serves worse with the cold form - 13 SGPR reloads per layer, SW 17.75 -> 18.4 ms; tools/experiments/README.md)
#ifndef RR_SW3_COLD
#define RR_SW3_COLD 0
#endif
    constexpr bool COLD = RR_SW3_COLD || !(CA == 3 && !DIAG && !BAND);
    auto arg_state = [&]() -> DevState<FT> { if constexpr (COLD) return cold_state<FT>(RR_ARG(as)); else return a.as; };
    auto arg_cld = [&]() -> DevCld<FT> { if constexpr (COLD) return cold_cld<FT>(RR_ARG(cld)); else return a.cld; };
    auto arg_aero = [&]() -> DevAero<FT> { if constexpr (COLD) return cold_aero<FT>(RR_ARG(aero)); else return a.aero; };
    auto arg_lk = [&]() -> DevGas<FT> { if constexpr (COLD) return cold_gas_view<FT>(RR_ARG(lk), sh, d); else return gas_view(a.lk, sh, d); };
    auto store = [&](int col, bool zero) {
        DevFlux<FT> fl_out;
        if constexpr (COLD) fl_out = cold_flux<FT>(RR_ARG(fl)); else fl_out = a.fl;
        if (!BAND) fl_out.band_up = fl_out.band_dn = fl_out.band_net = nullptr;
        store_column(fl_out, sh, d, col, ncol, zero, arg_lk());
    };
    auto chunk = [&](int col, int k0, int kn) {   // prepare_chunk with its arguments read where they are used
        const DevState<FT> as = arg_state();
        const DevCld<FT> cld = arg_cld();
        const DevAero<FT> aero = arg_aero();
        prepare_chunk(sh, d, arg_lk(), &cld, &aero, as, col, k0, kn, true);
    };

    for (int col = blockIdx.x; col < ncol; col = next_column(sh, d, a.queue)) {
        const FT mu0 = kernarg_ptr<const FT>(RR_ARG(cos_zenith))[col];
        const bool day = mu0 > FT(0);
        if (!TWOSTREAM && !day) {  // shortwave_noscat.jl:86-99: nothing runs for night columns
            store(col, true);
            continue;
        }
        {
            const DevState<FT> as = arg_state();
            const DevCld<FT> cld = arg_cld();
            const DevAero<FT> aero = arg_aero();
            prepare_column(sh, d, arg_lk(), &cld, &aero, as, col);
        }
        const FT toa_flux = kernarg_ptr<const FT>(RR_ARG(toa_flux))[col];
        FT *acc = sh.acc + (size_t)(BAND ? tid >> 4 : wave) * nlev * NA;
        const bool writer = BAND ? (lane & 15) == 15 : lane == 63;

        if (!TWOSTREAM) {
            // rte_sw_noscat!, shortwave_noscat.jl:120-148 (multiplicative Beer-Lambert, flux_up = 0)
            FT dir = toa_flux * solar_frac * mu0;
            {
                const FT s = seg_sum<BAND>(dir * amask);
                if (writer) { acc[nlay * 3] = FT(0); acc[nlay * 3 + 1] = s; acc[nlay * 3 + 2] = s; }
            }
            for (int c = nchunk - 1; c >= 0; c--) {
                const int k0 = c * CHK, kn = min(CHK, nlay - k0);
                __syncthreads();
                chunk(col, k0, kn);
                __syncthreads();
                for (int kk = kn - 1; kk >= 0; kk--) {
                    const int k = k0 + kk;
                    FT tau, ssa, pf;
                    gas_optics<FT, true>(a.lk, sh, lb, k, kk, nb, tau, ssa, pf);
                    dir = dir * m_exp_neg(tau / m_max(mu0, mu0_min<FT>()));
                    const FT s = seg_sum<BAND>(dir * amask);
                    if (writer) { acc[k * 3] = FT(0); acc[k * 3 + 1] = s; acc[k * 3 + 2] = s; }
                }
            }
            __syncthreads();
            store(col, false);
            __syncthreads();
            continue;
        }

        uint64_t m0 = 0, m1 = 0;
        if (d.has_cld) {
            const uint64_t key = mcica_key(kernarg_load<uint64_t>(RR_ARG(seed)), kernarg_load<int64_t>(RR_ARG(col_offset)) + col + 1, g + 1, 1);
            const bool cloudy = build_cloud_mask(sh, d, key, m0, m1) && active;
            const unsigned long long b = __ballot(cloudy);
            if (lane == 0) sh.misc[wave] = __popcll(b);
        }

        if (day) {
            // ---- sweep 1, top-down: optics, direct beam, layer coefficients and ADDING FROM THE TOP.
            // The reference adds from the surface up (albedo/src of everything below a level,
            // shortwave_2stream.jl:340-361) and then sweeps down; the direct beam, however, is only
            // known top-down.  The same two-stream layer relations
            //     U_{k+1} = T U_k + R D_{k+1} + S_up ,   D_k = T D_{k+1} + R U_k + S_dn
            // are closed here from the top instead: D_k = beta_k U_k + delta_k with beta = delta = 0 at
            // the top of the domain (no incident diffuse flux, :331),
            //     den = 1 / (1 - beta_{k+1} R),  beta_k = R + T^2 beta_{k+1} den,
            //     delta_k = S_dn + T den (delta_{k+1} + beta_{k+1} S_up),
            // which is the mirror image of Eqs 9-11 and gives the same fluxes.  One sweep fewer, and
            // 3 stored values per level; delta only enters D additively, so it is summed on the fly. ----
            const FT dir_top = toa_flux * solar_frac * mu0;
            const FT inv_mu0 = FT(1) / m_max(mu0, mu0_min<FT>());
            // one stream of the top-down adding: all-sky, and (DIAG) its clear-sky twin
            struct Stream { FT tau_cum, dir_above, beta, delta; };
            Stream S{FT(0), dir_top, FT(0), FT(0)}, C{FT(0), dir_top, FT(0), FT(0)};
            MaskWalk<false> mw(m0, m1, nlay, sh.mask);  // top-down
            if (RR_ACC_ATOMIC && !BAND)   // the layer loop ADDS into this wave's accumulators (wave_add_to)
                for (int i = lane; i < nlev * NA; i += 64) acc[i] = FT(0);
            {
                const FT s = seg_sum<BAND>(dir_top * amask);
                if (writer) {
                    acc[nlay * NA + 2] = s; acc[nlay * NA + 1] = FT(0);
                    if (DIAG) { acc[nlay * NA + 5] = s; acc[nlay * NA + 4] = FT(0); }
                }
            }
            // DIAG: above the highest cloudy layer of the COLUMN (wave-uniform; every layer of a cloudless column) the clear-sky
            // stream IS the all-sky stream — same optics, same beam, nothing reflected from above: no arithmetic of its own,
            // no scratch rows (the second sweep reads the all-sky rows for both); `also_twin`: its sums go to both accumulators
            const int twin_upto = DIAG ? sh.misc[d.nwaves + 2] : -1;   // last layer with own clear-sky rows, -1: none
            auto layer = [&](Stream &t, FT tau, FT ssa, FT gg, int k, int voff, int aoff, FT &Rdir, FT &Tdir, FT &Rdif,
                             FT &Tdif, bool recompute, bool also_twin) {
                t.tau_cum += tau;
                const FT dir_k = dir_top * m_exp_neg(t.tau_cum * inv_mu0);  // shortwave_2stream.jl:318-327
                if (recompute) sw_2stream_coeffs(tau, ssa, gg, mu0, inv_mu0, Rdir, Tdir, Rdif, Tdif);
                const FT s_up = Rdir * t.dir_above, s_dn = Tdir * t.dir_above;
                const FT den = m_rcp(FT(1) - t.beta * Rdif);
                // U_{k+1} = A U_k + B,  D_{k+1} = beta_{k+1} U_{k+1} + delta_{k+1}
                sw.put3(k, voff, Tdif * den, (Rdif * t.delta + s_up) * den, t.beta);
                const FT beta_n = Rdif + Tdif * Tdif * t.beta * den;
                t.delta = s_dn + Tdif * den * (t.delta + t.beta * s_up);
                t.beta = beta_n;
                if (RR_ACC_ATOMIC && !BAND) {
                    const FT rdir = row_sum(dir_k * amask), rdel = row_sum(t.delta * amask);
                    if ((threadIdx.x & 15) == 15) {   // wave_add_to: the four rows' lanes add their row sums
                        (void)__hip_atomic_fetch_add(&acc[k * NA + aoff + 2], rdir, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                        (void)__hip_atomic_fetch_add(&acc[k * NA + aoff + 1], rdel, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                        if (DIAG && also_twin) {
                            (void)__hip_atomic_fetch_add(&acc[k * NA + 5], rdir, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                            (void)__hip_atomic_fetch_add(&acc[k * NA + 4], rdel, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                        }
                    }
                } else {
                    const FT sdir = seg_sum<BAND>(dir_k * amask), sdel = seg_sum<BAND>(t.delta * amask);
                    if (writer) {
                        acc[k * NA + aoff + 2] = sdir; acc[k * NA + aoff + 1] = sdel;
                        if (DIAG && also_twin) { acc[k * NA + 5] = sdir; acc[k * NA + 4] = sdel; }
                    }
                }
                t.dir_above = dir_k;
            };
            for (int c = nchunk - 1; c >= 0; c--) {
                const int k0 = c * CHK, kn = min(CHK, nlay - k0);
                mw.refill(k0 + kn - 1);
                __syncthreads();
                chunk(col, k0, kn);
                __syncthreads();
                for (int kk = kn - 1; kk >= 0; kk--) {
                    const int k = k0 + kk, r = kk * NBMAX + lb.ibnd;
                    FT tau, ssa, pf, gg = FT(0);
                    gas_optics<FT, true>(a.lk, sh, lb, k, kk, nb, tau, ssa, pf);
                    FT tau_c = tau, ssa_c = ssa, g_c = FT(0);
                    const bool cld_k = d.has_cld && mw.next(k);
                    if (cld_k) { const V4<FT> cr = sh.ch->cld[r]; increment_2stream(tau, ssa, gg, cr.x, cr.y, cr.z); }
                    if (d.has_aero && sh.lay[k].aero_mask) {
                        const V4<FT> cr = sh.ch->aer[r];
                        increment_2stream(tau, ssa, gg, cr.x, cr.y, cr.z);
                        if (DIAG && cld_k) increment_2stream(tau_c, ssa_c, g_c, cr.x, cr.y, cr.z);
                    }
                    FT Rdir, Tdir, Rdif, Tdif;
                    const bool twin_same = DIAG && k > twin_upto;
                    layer(S, tau, ssa, gg, k, 0, 0, Rdir, Tdir, Rdif, Tdif, true, twin_same);
                    if (DIAG) {
                        if (twin_same) C = S;
                        else {
                            // without a cloud in this lane's sample the clear layer IS the all-sky layer: reuse its coefficients
                            if (!cld_k) { tau_c = tau; ssa_c = ssa; g_c = gg; }
                            layer(C, tau_c, ssa_c, g_c, k, 3, 3, Rdir, Tdir, Rdif, Tdif, cld_k, false);
                        }
                    }
                }
            }
            // ---- surface: U_1 = alb_dif D_1 + dir_sfc alb_dir, D_1 = beta_1 U_1 + delta_1 ----
            const FT alb = kernarg_ptr<const FT>(RR_ARG(alb_dif))[(size_t)lb.ibnd + (size_t)nb * col];
            const FT alb_d = kernarg_ptr<const FT>(RR_ARG(alb_dir))[(size_t)lb.ibnd + (size_t)nb * col];
            FT U = m_div(alb * S.delta + S.dir_above * alb_d, FT(1) - alb * S.beta);
            FT Uc = DIAG ? m_div(alb * C.delta + C.dir_above * alb_d, FT(1) - alb * C.beta) : FT(0);
            {
                const FT su = seg_sum<BAND>(U * amask), sb = seg_sum<BAND>(S.beta * U * amask);
                if (writer) { acc[0] = su; acc[1] = (acc[1] + sb) + acc[2]; }
                if (DIAG) {
                    const FT suc = seg_sum<BAND>(Uc * amask), sbc = seg_sum<BAND>(C.beta * Uc * amask);
                    if (writer) { acc[3] = suc; acc[4] = (acc[4] + sbc) + acc[5]; }
                }
            }
            // ---- sweep 2, bottom-up: fluxes ----
            constexpr int DBT = DIAG ? DB / 2 : DB;
            // DIAG: layers [0, twin_upto] carry their own clear-sky rows, the layers above them share the all-sky rows
            for (int phase = 0; phase < (DIAG ? 2 : 1); phase++) {
            const int k_lo = DIAG && phase == 1 ? twin_upto + 1 : 0, k_end = DIAG && phase == 0 ? twin_upto + 1 : nlay;
            const bool own = phase == 0;
            for (int kl = k_lo; kl < k_end; kl += DBT) {
                FT A[DBT], B[DBT], BE[DBT], Ac[DIAG ? DBT : 1], Bc[DIAG ? DBT : 1], BEc[DIAG ? DBT : 1];
#pragma unroll
                for (int j = 0; j < DBT; j++) {
                    const int k = kl + j < k_end ? kl + j : k_end - 1;
                    sw.get3(k, 0, A[j], B[j], BE[j]);
                    if (DIAG) {
                        if (own) sw.get3(k, 3, Ac[j], Bc[j], BEc[j]);
                        else { Ac[j] = A[j]; Bc[j] = B[j]; BEc[j] = BE[j]; }
                    }
                }
                if (!BAND && !DIAG && DBT == 16) {
                    FT pu[16], pb[16];  // the 2 x 16 g-point sums of the batch in two 16-value reductions
                    if (kl + 16 <= k_end) {   // a whole batch (wave-uniform): no per-level selects
#pragma unroll
                        for (int j = 0; j < 16; j++) {
                            U = A[j] * U + B[j];
                            pu[j] = U * amask;
                            pb[j] = BE[j] * U * amask;
                        }
                    } else {
#pragma unroll
                        for (int j = 0; j < 16; j++) {
                            const bool in = kl + j < k_end;
                            if (in) U = A[j] * U + B[j];
                            pu[j] = in ? U * amask : FT(0);
                            pb[j] = in ? BE[j] * U * amask : FT(0);
                        }
                    }
                    FT wu[4], wb[4];
                    wave_sum16(pu, wu);
                    wave_sum16(pb, wb);
                    if ((lane & 15) == 15) {  // row r holds batch entries j = i + 4 r
#pragma unroll
                        for (int i = 0; i < 4; i++) {
                            const int j = i + 4 * (lane >> 4), lev = kl + j + 1;
                            if (kl + j < k_end) {
                                acc[lev * NA] = wu[i];
                                acc[lev * NA + 1] = (acc[lev * NA + 1] + wb[i]) + acc[lev * NA + 2];
                            }
                        }
                    }
                } else if (!BAND && DIAG && DBT == 8) {
                    FT pa[16], pc[16];  // two streams x (U, beta U) x 8 levels: one 16-value reduction per stream
#pragma unroll
                    for (int j = 0; j < 8; j++) {
                        const bool in = kl + j < k_end;
                        if (in) { U = A[j] * U + B[j]; Uc = Ac[j] * Uc + Bc[j]; }
                        pa[j] = in ? U * amask : FT(0);
                        pa[j + 8] = in ? BE[j] * U * amask : FT(0);
                        pc[j] = in ? Uc * amask : FT(0);
                        pc[j + 8] = in ? BEc[j] * Uc * amask : FT(0);
                    }
                    FT wa[4], wc[4];
                    wave_sum16(pa, wa);
                    wave_sum16(pc, wc);
                    if ((lane & 15) == 15) {  // rows 0, 1: U sums of entries j = i + 4 r; rows 2, 3: beta U sums
                        const int r = lane >> 4;
#pragma unroll
                        for (int i = 0; i < 4; i++) {
                            const int j = i + 4 * (r & 1), lev = kl + j + 1;
                            if (kl + j < k_end) {
                                if (r < 2) { acc[lev * NA] = wa[i]; acc[lev * NA + 3] = wc[i]; }
                                else {
                                    acc[lev * NA + 1] = (acc[lev * NA + 1] + wa[i]) + acc[lev * NA + 2];
                                    acc[lev * NA + 4] = (acc[lev * NA + 4] + wc[i]) + acc[lev * NA + 5];
                                }
                            }
                        }
                    }
                } else {
#pragma unroll
                    for (int j = 0; j < DBT; j++) {
                        if (kl + j < k_end) {
                            const int lev = kl + j + 1;
                            U = A[j] * U + B[j];
                            const FT su = seg_sum<BAND>(U * amask), sb = seg_sum<BAND>(BE[j] * U * amask);
                            if (writer) { acc[lev * NA] = su; acc[lev * NA + 1] = (acc[lev * NA + 1] + sb) + acc[lev * NA + 2]; }
                            if (DIAG) {
                                Uc = Ac[j] * Uc + Bc[j];
                                const FT suc = seg_sum<BAND>(Uc * amask), sbc = seg_sum<BAND>(BEc[j] * Uc * amask);
                                if (writer) { acc[lev * NA + 3] = suc; acc[lev * NA + 4] = (acc[lev * NA + 4] + sbc) + acc[lev * NA + 5]; }
                            }
                        }
                    }
                }
            }
            }
        } else if (want_aod) {
            // night column: the reference still runs the optics, so the AOD diagnostic is defined
            for (int c = 0; c < nchunk; c++) {
                const int k0 = c * CHK, kn = min(CHK, nlay - k0);
                __syncthreads();
                chunk(col, k0, kn);
            }
        }
        __syncthreads();
        if (want_aod && tid == 0) {
            // the 550 nm AOD: sum over masked layers in layer order (aerosol_optics.jl:96-116)
            FT e = FT(0), s = FT(0);
            for (int k = 0; k < nlay; k++)
                if (sh.lay[k].aero_mask) { e += sh.lay[k].aod_t; s += sh.lay[k].aod_ts; }
            kernarg_ptr<FT>(RR_ARG(as.aod_sw_ext))[col] = e;
            kernarg_ptr<FT>(RR_ARG(as.aod_sw_sca))[col] = s;
        }
        store(col, !day);
        if (d.has_cld && tid == 0) {
            FT *cover = kernarg_ptr<FT>(RR_ARG(as.cld_cover));
            if (cover) {
                int n = 0;
                for (int w = 0; w < d.nwaves; w++) n += sh.misc[w];
                cover[col] = (FT)((double)n / (double)a.lk.n_gpt);  // exact quotient
            }
        }
    }
    queue_release(a.queue);
#undef RR_ARG
}

int column_grid(rrtmgp_workspace *ws, int ncol, int threads, size_t lds_bytes, const void *kernel);

template <typename FT>
int launch_sw(rrtmgp_workspace *ws, int twostream, const DevGas<FT> &lk, const DevCld<FT> *cld, const DevAero<FT> *aero,
              const DevState<FT> &as, const FT *cos_zenith, const FT *toa_flux, const FT *alb_dir, const FT *alb_dif,
              const DevFlux<FT> &fl, uint64_t seed, int64_t col_offset, int max_int, const Lane *lane) {
    const Lane main_lane{ws->stream, &ws->scratch, 0};
    const Lane &ln = lane ? *lane : main_lane;   // where this launch goes: the workspace's main lane, or the one it was handed
    SwArgs<FT> a{};
    a.lk = lk;
    if (cld) a.cld = *cld;
    if (aero) a.aero = *aero;
    a.as = as; a.fl = fl;
    a.cos_zenith = cos_zenith; a.toa_flux = toa_flux; a.alb_dir = alb_dir; a.alb_dif = alb_dif;
    a.seed = seed; a.col_offset = col_offset;
    const int threads = ((fl.band_up ? lk.band_rows * 16 : lk.n_gpt) + 63) / 64 * 64;
    RR_CHECK(threads <= 256, "n_gpt > 256 is not supported");
    ColDims d{};
    d.nlay = as.nlay; d.nlev = as.nlay + 1;
    d.ngas1 = (as.ngas + 1 > lk.n_gases) ? as.ngas + 1 : lk.n_gases;
    RR_CHECK(lk.n_eta <= 255 && lk.n_pp <= 255 && lk.n_t_ref <= 255, "lookup axes longer than 255 are not supported");
    RR_CHECK(lk.n_bnd <= NBMAX, "more than 16 bands per lookup are not supported");
    if (fl.band_up) {
        RR_CHECK(twostream && fl.band_dn, "per-band fluxes need a two-stream solver and both up/dn buffers");
        if (!lk.band_rows) return rrtmgp::set_error(RRTMGP_EUNSUPPORTED, "per-band fluxes: the bands, each padded to 16 g-points, must fit 256 lanes");
    }
    d.nwaves = threads / 64; d.nseg = fl.band_up ? threads / 16 : d.nwaves;
    d.n_t_ref = lk.n_t_ref; d.n_p_ref = lk.n_pp - 1; d.n_t_plnk = lk.n_t_plnk; d.n_gases_ref = lk.n_gases;
    d.nint0 = lk.m_nint[0]; d.nint1 = lk.m_nint[1]; d.nslot0 = lk.m_nslot[0]; d.nslot1 = lk.m_nslot[1]; d.nbnd = lk.n_bnd; d.lw = 0; d.twostream = twostream;
    const bool diag = fl.clear_up != nullptr;
    d.has_cld = cld != nullptr; d.has_aero = aero != nullptr; d.n_acc = diag ? 6 : 3; d.diag = diag; d.max_int = max_int;
    a.dims = d;
    // the variants instantiated with aerosols known at compile time (CA >= 2 below) prepare chunk_layers(CA) layers at a time
    const bool ca_aero = twostream && ((aero && (diag || !fl.band_up)) || (diag && chunk_layers(1, true) != CH));
    ColShared<FT, chunk_layers(0)> dummy;
    ColShared<FT, chunk_layers(2)> dummy_aero;
    size_t lds = ca_aero ? carve_shared(dummy_aero, (char *)nullptr, d) : carve_shared(dummy, (char *)nullptr, d);
    // main Float32 instances: 8-layer chunks when that is what keeps 4 workgroups resident per CU (160 KB / 4).  Measured:
    // 72 layers LW 21.4 -> 20.2 ms; at 96 layers 3 workgroups with 16-layer chunks are faster (28.0 vs 29.5 ms), hence <= 80 (round 2 kernels; re-measured in round 5, below)
    static const bool no_half = getenv("RRTMGP_HIP_NO_HALF_CHUNKS") != nullptr;  // A/B switch
    // Float64: the 8-layer instances are compiled for RR_F64_HALF_WAVES waves per SIMD (device.h) and taken when their
    // records let that many workgroups share the CU's LDS
    static const size_t lds_cap_f32 = getenv("RRTMGP_HIP_SW_LDS_CAP") ? (size_t)atol(getenv("RRTMGP_HIP_SW_LDS_CAP")) : 40960;   // A/B switch
    const size_t lds_cap = sizeof(FT) == 4 ? lds_cap_f32 : (160 * 1024) / RR_F64_HALF_WAVES;
    ColShared<FT, half_chunk_layers<FT>()> dummy_half;
    // (round 5 sweep, tools/experiments/half_rule_sweep.sh, profiles/r05_half_rule_sweep_ab.txt: 80 layers +6.9 %, 84 +6.1 %, 88 +4.5 %,
    // 96 layers LW 11.49 -> 10.98 ms but SW 13.04 -> 13.12: the limit was 80 for both kernels since round 2; now 88 here)
    static const bool force_half_main = getenv("RRTMGP_HIP_FORCE_HALF_CHUNKS") != nullptr;  // A/B switch: whatever the depth
    const bool half = !no_half && (sizeof(FT) == 4 ? (d.nlay <= 88 || force_half_main) : RR_F64_HALF_WAVES > 2) && twostream && !diag && !fl.band_up && !aero &&
                      lds > lds_cap && carve_shared(dummy_half, (char *)nullptr, d) <= lds_cap;
    if (half) lds = carve_shared(dummy_half, (char *)nullptr, d);
    if (diag) {
        RR_CHECK(twostream && cld, "the one-pass clear-sky diagnostic needs the two-stream solver and a cloud lookup");
        RR_CHECK(!fl.band_up, "per-band fluxes and the one-pass clear-sky diagnostic cannot be combined in one launch");
    }
    auto kern = !twostream ? sw_solve_kernel<FT, false, false, false>
                : diag     ? (aero ? sw_solve_kernel<FT, true, false, true, 3> : sw_solve_kernel<FT, true, false, true, 1>)
                : fl.band_up ? sw_solve_kernel<FT, true, true, false>
                : (cld && aero) ? sw_solve_kernel<FT, true, false, false, 3>
                : half ? (cld ? sw_solve_kernel<FT, true, false, false, 1, true> : sw_solve_kernel<FT, true, false, false, 0, true>)
                : cld  ? sw_solve_kernel<FT, true, false, false, 1>
                : aero ? sw_solve_kernel<FT, true, false, false, 2> : sw_solve_kernel<FT, true, false, false, 0>;
    int grid = column_grid(ws, as.ncol, threads, lds, (const void *)kern);
    if constexpr (sizeof(FT) == 4) {
        // One-pass diagnostic, Float32: the same instances exist once more at 128 VGPRs (4 waves per SIMD instead of 3) with
        // 8-layer chunks (HALF).  They spill (51 / 32 registers, LW / SW) and pay off when they put one more workgroup on a CU
        // than the 168-VGPR instances AND the sweep scratch of that workgroup still finds room behind the L2: measured
        // (tools/experiments/diag4_sweep.sh, profiles/r04_diag4_sweep_ab.txt; both flux sets, M columns/s)
        //   40 layers 4.89 -> 5.39 (4 vs 3 per CU), 64 layers 3.20 -> 3.26, with aerosols 2.76 -> 3.00, 128 layers 1.16 -> 1.29 (3 vs 2),
        //   but 72 layers 2.81 -> 2.74 and 96 layers 2.00 -> 1.96 (4 vs 3 per CU, 450 -> 600 MB of scratch).
        // So: when they admit one more workgroup, up to 64 layers or when the others would leave a CU with two.
        static const bool no_diag_half = getenv("RRTMGP_HIP_NO_DIAG_HALF") != nullptr;  // A/B switch
        if (diag && !no_diag_half && grid > 0) {
            auto k4 = aero ? sw_solve_kernel<FT, true, false, true, 3, true> : sw_solve_kernel<FT, true, false, true, 1, true>;
            const size_t lds4 = carve_shared(dummy_half, (char *)nullptr, d);
            const int cap3 = column_grid(ws, INT_MAX, threads, lds, (const void *)kern);
            const int cap4 = lds4 <= 160 * 1024 ? column_grid(ws, INT_MAX, threads, lds4, (const void *)k4) : -1;
            static const bool force_half = getenv("RRTMGP_HIP_FORCE_DIAG_HALF") != nullptr;  // A/B switch: whenever they admit one more
            // Round 5 swept the rule at 60 / 64 / 72 / 73 / 80 layers with and without MERRA aerosols (tools/experiments/
            // diag_rule_sweep.sh, profiles/r05_diag_rule_sweep_ab.txt): without aerosols it stands (65-80 layers: the fourth
            // workgroup costs 0.3-1.6 percent); WITH aerosols this kernel gains from it up to 80 layers
            // (SW 15.56 -> 14.55 ms at 72 layers, 17.49 -> 16.55 at 80; LW 11.65 -> 11.52 at 72, but 12.83 -> 13.07 at 80).
            if (cap4 > cap3 && as.ncol > cap3 && (force_half || d.nlay <= 64 || cap3 <= 2 * ws->n_cu || (aero && d.nlay <= 80))) {
                kern = k4; lds = lds4; grid = std::min(as.ncol, cap4);
            }
        } else if (diag && !no_diag_half) {   // the 16-layer records do not fit the LDS at all: the 8-layer ones may
            auto k4 = aero ? sw_solve_kernel<FT, true, false, true, 3, true> : sw_solve_kernel<FT, true, false, true, 1, true>;
            const size_t lds4 = carve_shared(dummy_half, (char *)nullptr, d);
            const int g4 = lds4 <= 160 * 1024 ? column_grid(ws, as.ncol, threads, lds4, (const void *)k4) : -1;
            if (g4 > 0) { kern = k4; lds = lds4; grid = g4; }
        }
    }
    if (grid < 0) return grid;
    const size_t sweep_bytes = (size_t)grid * d.nlev * (diag ? 6 : 3) * SWEEP_LANES * sizeof(FT);
    int rc = scratch_ensure(ws, sweep_bytes, &ln);
    if (rc) return rc;
    a.scratch = (FT *)ln.scratch->ptr;
    rc = queue_ensure(ws, ln.idx, &a.queue);   // {next column, workgroups done}: zero between launches (queue_release)
    if (rc) return rc;
    // (last_kernel_ms = the workspace's LAST solver launch, whichever lane it ran on: include/rrtmgp_hip.h)
    if (ws->timed) RR_HIP(hipEventRecord(ws->ev_start, ln.stream));
    hipLaunchKernelGGL(kern, dim3(grid), dim3(threads), lds, ln.stream, a);
    RR_HIP(hipGetLastError());
    if (ws->timed) RR_HIP(hipEventRecord(ws->ev_stop, ln.stream));
    return RRTMGP_OK;
}

template int launch_sw<float>(rrtmgp_workspace *, int, const DevGas<float> &, const DevCld<float> *,
                              const DevAero<float> *, const DevState<float> &, const float *, const float *,
                              const float *, const float *, const DevFlux<float> &, uint64_t, int64_t, int, const Lane *);
template int launch_sw<double>(rrtmgp_workspace *, int, const DevGas<double> &, const DevCld<double> *,
                               const DevAero<double> *, const DevState<double> &, const double *, const double *,
                               const double *, const double *, const DevFlux<double> &, uint64_t, int64_t, int, const Lane *);

}  // namespace rrtmgp
