// solve_lw.hip — longwave column kernels (two-stream and no-scattering) for gfx950.
//
// Replaces rte_lw_2stream_solve! / rte_lw_noscat_solve! of the reference
// (ext/cuda/rte_longwave_2stream.jl:48-141, rte_longwave_noscat.jl:54-150; bodies
// src/rte/longwave_2stream.jl, longwave_noscat.jl, src/optics/compute_optical_props.jl:18-245).
//
// One workgroup per column, one lane per g-point, layers in chunks of CH whose
// band-level records are prepared cooperatively in LDS (device.h).  Two-stream: a
// single TOP-DOWN sweep fuses gas/cloud/aerosol optics, Planck sources, the layer
// reflectance/transmittance and the adding step closed from the top of the domain
// (D_k = beta_k U_k + delta_k; the mirror image of the reference's Eqs 9-11), and leaves
// 3 numbers per level (A, B, beta; delta is summed over g-points on the fly) - 2 above the
// column's highest cloud / aerosol layer, where nothing scatters and beta stays 0 - so that the
// bottom-up sweep is two FMAs per level:
//     U_{k+1} = A_k U_k + B_k ,   D_{k+1} = beta_{k+1} U_{k+1} + delta_{k+1} .
// Broadband fluxes are wavefront sums over g-points (fixed DPP order).
#include "device.h"
#include <climits>


namespace rrtmgp {

// lw_2stream_coeffs (src/rte/longwave_2stream.jl:149-222) of a purely absorbing layer (gas only: every layer above the
// column's highest cloud / aerosol layer, 3 of 4 cells of an all-sky column).  The general expressions with ssa = 0
// substituted: gamma1 = D, gamma2 = 0, k = D, RT_term = 1/(2D), hence Rdif = 0, Tdif = e^{-D tau}, emis_fac = 1 - Tdif and
// dBz = dB (1 - Tdif) / (D tau).  No sqrt, one reciprocal instead of three.
template <typename FT>
__device__ __forceinline__ void lw_absorbing_coeffs(FT tau, FT lev_src_bot, FT lev_src_top, FT &Tdif, FT &src_up, FT &src_dn) {
    const FT lw_diff_sec = FT(1.66);
    FT e1, om1;
    exp_pair(tau * lw_diff_sec, e1, om1);
    Tdif = e1;
    const FT dB = lev_src_bot - lev_src_top;
    const FT dBz = m_div(dB * om1, lw_diff_sec * tau);
    const bool pos = tau > FT(0);
    src_up = pos ? Num<FT>::pi() * (lev_src_top * om1 - e1 * dB + dBz) : FT(0);
    src_dn = pos ? Num<FT>::pi() * (lev_src_bot * om1 + e1 * dB - dBz) : FT(0);
}

// lw_2stream_coeffs, src/rte/longwave_2stream.jl:149-222
template <typename FT>
__device__ __forceinline__ void lw_2stream_coeffs(FT tau, FT ssa, FT g, FT lev_src_bot, FT lev_src_top, FT &Rdif,
                                                  FT &Tdif, FT &src_up, FT &src_dn) {
    const FT lw_diff_sec = FT(1.66);
    if (__all(ssa == FT(0))) {   // every lane of the wavefront is a purely absorbing layer
        Rdif = FT(0);
        lw_absorbing_coeffs(tau, lev_src_bot, lev_src_top, Tdif, src_up, src_dn);
        return;
    }
    const FT gamma1 = lw_diff_sec * (FT(1) - FT(0.5) * ssa * (FT(1) + g));
    const FT gamma2 = lw_diff_sec * FT(0.5) * ssa * (FT(1) - g);
    const FT k = m_sqrt_pos(m_max(lw_diff_sec * (FT(1) - ssa) * (gamma1 + gamma2), k_min<FT>()));
    FT e1, om1;
    exp_pair(tau * k, e1, om1);
    const FT coeff = e1 * e1;
    const FT one_minus_e2kt = om1 * (FT(1) + e1);
    const FT RT_term = m_rcp(k * (FT(1) + coeff) + gamma1 * one_minus_e2kt);
    Rdif = RT_term * gamma2 * one_minus_e2kt;
    Tdif = RT_term * FT(2) * k * e1;
    if (tau > FT(0)) {
        const FT dB = lev_src_bot - lev_src_top;
        const FT gamma_sum = gamma1 + gamma2;
        const FT one_p_e1 = FT(1) + e1;
        const FT emis_fac = om1 * (k * om1 + lw_diff_sec * (FT(1) - ssa) * one_p_e1) * RT_term;
        const FT dBz = m_div(dB * m_div(om1, tau) * (k * om1 + gamma_sum * one_p_e1) * RT_term, m_max(gamma_sum, Num<FT>::eps()));
        src_up = Num<FT>::pi() * (lev_src_top * emis_fac - Tdif * dB + dBz);
        src_dn = Num<FT>::pi() * (lev_src_bot * emis_fac + Tdif * dB - dBz);
    } else {
        src_up = FT(0);
        src_dn = FT(0);
    }
}

template <typename FT>
struct LwArgs {
    DevGas<FT> lk;
    DevCld<FT> cld;
    DevAero<FT> aero;
    DevState<FT> as;
    DevFlux<FT> fl;
    const FT *sfc_emis;  // (nbnd, ncol)
    const FT *inc_flux;  // (inc_ld, ngpt) or nullptr: column col of g-point g at col + inc_ld * g
    int inc_ld;
    FT *scratch;
    int *queue;  // {next column of the persistent grid, workgroups done}: device counters, zero between launches (queue_release)
    ColDims dims;
    int n_angles;
    FT Ds[4], wts[4];
    uint64_t seed;
    int64_t col_offset;
};

// cloud + aerosol increments of one layer for this lane (TwoStream), or their absorption only (OneScalar)
template <typename FT, bool TWOSTREAM, int CHK>
__device__ __forceinline__ void lw_layer_increments(const ColDims &d, const ColShared<FT, CHK> &sh, const LaneBand &lb, int k,
                                                    int kk, bool cloudy, FT &tau, FT &ssa, FT &g) {
    g = FT(0);
    const int r = kk * NBMAX + lb.ibnd;
    if (d.has_cld && cloudy) {
        const V4<FT> c = sh.ch->cld[r];
        if (TWOSTREAM) increment_2stream(tau, ssa, g, c.x, c.y, c.z);
        else tau += c.x;
    }
    if (d.has_aero && sh.lay[k].aero_mask) {
        const V4<FT> c = sh.ch->aer[r];
        if (TWOSTREAM) increment_2stream(tau, ssa, g, c.x, c.y, c.z);
        else tau += c.x;
    }
}

// optics of one layer for this lane: gas, then the increments
template <typename FT, bool TWOSTREAM, int CHK>
__device__ __forceinline__ void lw_layer_optics(const LwArgs<FT> &a, const ColDims &d, const ColShared<FT, CHK> &sh,
                                                const LaneBand &lb, int k, int kk, bool cloudy, FT &tau, FT &ssa,
                                                FT &g, FT &pfrac) {
    gas_optics<FT, false>(a.lk, sh, lb, k, kk, d.nbnd, tau, ssa, pfrac);
    lw_layer_increments<FT, TWOSTREAM>(d, sh, lb, k, kk, cloudy, tau, ssa, g);
}

constexpr int DB = 16;  // levels per batch of the top-down sweeps

// DIAG: the clear-sky recurrences (no cloud increment) are carried next to the all-sky ones in
// the same launch, sharing the gas optics, sources and aerosol record: the one-pass form of
// AllSkyRadiationWithClearSkyDiagnostics (update_fluxes.jl:39-65), which the reference solves twice.
// CA: -1 = clouds / aerosols are run-time flags; 0..3 = (clouds | aerosols << 1) known at compile time (the main
// two-stream instance: absent optics leave no code, no kernel arguments in registers and no lane masks behind).
// HALF: the main (no-aerosol) instances once more with 8-layer chunks, for columns whose 16-layer records would push a
// workgroup past a quarter of the CU's LDS (Float32, 71-80 layers): 4 resident workgroups instead of 3.  Float32 DIAG
// instances: once more with 8-layer chunks AND at 128 VGPRs (4 waves per SIMD), taken when that admits one more workgroup.
template <typename FT, bool TWOSTREAM, bool BAND, bool DIAG, int CA = -1, bool HALF = false>
__global__ void __launch_bounds__(256, (sizeof(FT) == 4 ? (DIAG ? (HALF ? 4 : RR_DIAG_MIN_WAVES) : RR_MIN_WAVES) : HALF ? RR_F64_HALF_WAVES : 2)) lw_solve_kernel(const LwArgs<FT> a) {
    static_assert(TWOSTREAM, "the no-scattering solver is lw_noscat_kernel");
    extern __shared__ __align__(16) char smem[];
    constexpr int CHK = HALF ? half_chunk_layers<FT>() : chunk_layers(CA, DIAG);  // layers per chunk of LDS records
    ColShared<FT, CHK> sh;
    ColDims dd = a.dims;
    if (CA >= 0) { dd.has_cld = CA & 1; dd.has_aero = (CA >> 1) & 1; }
    dd.diag = DIAG;  // what the host set, as a constant: the other flux set's pointers are never loaded
    dd.lw = 1; dd.twostream = 1;   // likewise (the preparation steps branch on them)
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
    constexpr int NA = DIAG ? 4 : 2;  // accumulated components per level: up, dn (+ clear up, dn)
    const Sweep<FT, NV> sw(a.scratch + (size_t)blockIdx.x * (size_t)nlev * NV * SWEEP_LANES, (size_t)nlev * NV * SWEEP_LANES * sizeof(FT), (unsigned)tid);
    const FT amask = active ? FT(1) : FT(0);
    const int nchunk = (nlay + CHK - 1) / CHK;
    cache_small_tables(sh, d, a.lk, a.as);
    // Everything below that is not an operand of the layer loops is read from the kernarg segment where it is used
    // (device.h, "cold kernel arguments"): offsets of the members of the one kernel argument
#define RR_ARG(member) ((unsigned)offsetof(LwArgs<FT>, member))

    for (int col = blockIdx.x; col < ncol; col = next_column(sh, d, a.queue)) {
        {
            const DevState<FT> as = cold_state<FT>(RR_ARG(as));
            const DevCld<FT> cld = cold_cld<FT>(RR_ARG(cld));
            const DevAero<FT> aero = cold_aero<FT>(RR_ARG(aero));
            prepare_column(sh, d, cold_gas_view<FT>(RR_ARG(lk), sh, d), &cld, &aero, as, col);
        }

        uint64_t m0 = 0, m1 = 0;
        if (d.has_cld) {
            const uint64_t key = mcica_key(kernarg_load<uint64_t>(RR_ARG(seed)), kernarg_load<int64_t>(RR_ARG(col_offset)) + col + 1, g + 1, 0);
            const bool cloudy = build_cloud_mask(sh, d, key, m0, m1) && active;
            const unsigned long long b = __ballot(cloudy);
            if (lane == 0) sh.misc[wave] = __popcll(b);
        }
        const FT emis = kernarg_ptr<const FT>(RR_ARG(sfc_emis))[(size_t)lb.ibnd + (size_t)nb * col];
        const FT *inc_flux = kernarg_ptr<const FT>(RR_ARG(inc_flux));
        const FT inc = inc_flux ? inc_flux[(size_t)col + (size_t)kernarg_load<int>(RR_ARG(inc_ld)) * g] : FT(0);
        FT *acc = sh.acc + (size_t)(BAND ? tid >> 4 : wave) * nlev * NA;
        const bool writer = BAND ? (lane & 15) == 15 : lane == 63;
        FT sfc_source = FT(0);

        // ---- sweep 1, TOP-DOWN: optics + sources, and one layer behind them coefficients + adding
        //      (compute_optical_props.jl:163-198, longwave_2stream.jl:273-333).  The reference adds from the surface up
        //      (albedo / source of everything below a level, Eqs 9-11) and then sweeps down.  The same layer relations
        //     U_{k+1} = T U_k + R D_{k+1} + S_up ,   D_k = T D_{k+1} + R U_k + S_dn
        // are closed here from the top instead, as sw_solve_kernel does: D_k = beta_k U_k + delta_k with beta = 0 and
        // delta = inc_flux at the top of the domain,
        //     den = 1 / (1 - beta_{k+1} R),  beta_k = R + T^2 beta_{k+1} den,
        //     delta_k = S_dn + T den (delta_{k+1} + beta_{k+1} S_up)
        // (the mirror image of Eqs 9-11; same fluxes), followed by the surface relation and one bottom-up sweep
        //     U_{k+1} = A U_k + B,  D_{k+1} = beta_{k+1} U_{k+1} + delta_{k+1};   A = T den, B = (R delta + S_up) den.
        // Per level the sweep keeps A, B, beta_{k+1}; delta only enters D additively, so its g-point sum is taken on the fly.
        // What the closure from the top buys is everything ABOVE the column's highest cloudy layer (k > k2):
        //   * gas-only longwave layers do not scatter (ssa = 0: R = 0 exactly), so beta stays 0 up there: D = delta is
        //     complete after this sweep, den = 1, and the second sweep needs only A = T and B = S_up - TWO stored values
        //     and one g-point sum per level instead of three and two (LW 15.1 -> 13.9 ms; aerosols scatter too: k2 is the
        //     highest layer with a cloud or an aerosol);
        //   * DIAG: above the highest cloudy layer the clear-sky twin IS the all-sky stream - same optics, same sources,
        //     the same boundary above: no coefficients, no adding step, no scratch rows of its own (the second sweep reads
        //     the all-sky rows for both).  Closed from the surface the two differ from the LOWEST cloudy layer up.
        // The optics run one layer ahead of the adding step, whose lower level source sqrt(inc[k-1] dec[k]) needs the Planck
        // fraction of the layer below. ----
        struct Stream { FT beta, delta; };
        Stream S{FT(0), inc}, C{FT(0), inc};
        MaskWalk<false> mw(m0, m1, nlay, sh.mask);
        {
            const FT sd = seg_sum<BAND>(inc * amask);
            if (writer) { acc[nlay * NA + 1] = sd; if (DIAG) acc[nlay * NA + 3] = sd; }
        }
        const int ctop = sh.misc[d.nwaves + 2];        // the column's highest cloudy layer, -1: none (prepare_column)
        const int k2 = sh.misc[d.nwaves + 4];          // highest layer with a cloud or an aerosol: the layers above it only absorb
        FT tau_p = FT(0), ssa_p = FT(0), g_p = FT(0);      // optics of layer k+1
        FT tau_pc = FT(0), ssa_pc = FT(0), g_pc = FT(0);   // ... without the cloud increment (DIAG)
        bool cld_p = false;
        FT lev_top = FT(0), dec_p = FT(0);                 // lev_source[k+2], B(t_lev[k+1]) * pfrac[k+1]
        FT pfrac = FT(0);                                  // Planck fraction of the current layer (after the loop: of layer 0)
        auto adding = [&](Stream &t, FT Rdif, FT Tdif, FT src_up, FT src_dn, int kl, int voff, int aoff, bool also_twin) {
            const FT den = m_rcp(FT(1) - t.beta * Rdif);
            sw.put3(kl, voff, Tdif * den /* A */, (Rdif * t.delta + src_up) * den /* B */, t.beta);
            const FT beta_n = Rdif + Tdif * Tdif * t.beta * den;
            t.delta = src_dn + Tdif * den * (t.delta + t.beta * src_up);
            t.beta = beta_n;
            const FT sdel = seg_sum<BAND>(t.delta * amask);
            if (writer) { acc[kl * NA + aoff + 1] = sdel; if (DIAG && also_twin) acc[kl * NA + 3] = sdel; }
        };
        auto add_layer = [&](int kl, FT lev_bot) {   // layer kl between levels kl and kl+1
            FT Rdif, Tdif, src_up, src_dn;
            const FT top = lev_top;   // level source at the top of this layer; its bottom is the next layer's top
            lev_top = lev_bot;
            if (kl > k2) {   // gas only in every lane: R = 0, beta stays 0 (wave-uniform: a property of the column)
                lw_absorbing_coeffs(tau_p, lev_bot, top, Tdif, src_up, src_dn);
                sw.put(kl, 0, Tdif); sw.put(kl, 1, src_up);
                S.delta = src_dn + Tdif * S.delta;
                const FT sdel = seg_sum<BAND>(S.delta * amask);
                if (writer) { acc[kl * NA + 1] = sdel; if (DIAG) acc[kl * NA + 3] = sdel; }
                if (DIAG) C = S;
                return;
            }
            lw_2stream_coeffs(tau_p, ssa_p, g_p, lev_bot, top, Rdif, Tdif, src_up, src_dn);
            const bool twin_same = DIAG && kl > ctop;
            adding(S, Rdif, Tdif, src_up, src_dn, kl, 0, 0, twin_same);
            if (DIAG) {
                if (twin_same) C = S;
                else {
                    // same coefficients unless this lane's McICA sample put a cloud in the layer
                    if (cld_p) lw_2stream_coeffs(tau_pc, ssa_pc, g_pc, lev_bot, top, Rdif, Tdif, src_up, src_dn);
                    adding(C, Rdif, Tdif, src_up, src_dn, kl, 3, 2, false);
                }
            }
        };
        for (int c = nchunk - 1; c >= 0; c--) {
            const int k0 = c * CHK, kn = min(CHK, nlay - k0);
            mw.refill(k0 + kn - 1);
            __syncthreads();
            {
                const DevState<FT> as = cold_state<FT>(RR_ARG(as));
                const DevCld<FT> cld = cold_cld<FT>(RR_ARG(cld));
                const DevAero<FT> aero = cold_aero<FT>(RR_ARG(aero));
                prepare_chunk(sh, d, cold_gas_view<FT>(RR_ARG(lk), sh, d), &cld, &aero, as, col, k0, kn, false);
            }
            __syncthreads();
            for (int kk = kn - 1; kk >= 0; kk--) {
                const int k = k0 + kk;
                FT tau, ssa, gg;
                FT tau_c = FT(0), ssa_c = FT(0), g_c = FT(0);
                bool cld_k = false;
                if (DIAG) {
                    gas_optics<FT, false>(a.lk, sh, lb, k, kk, nb, tau, ssa, pfrac);
                    tau_c = tau; ssa_c = ssa;
                    cld_k = d.has_cld && mw.next(k);
                    lw_layer_increments<FT, true>(d, sh, lb, k, kk, cld_k, tau, ssa, gg);
                    if (cld_k) lw_layer_increments<FT, true>(d, sh, lb, k, kk, false, tau_c, ssa_c, g_c);
                    else { tau_c = tau; ssa_c = ssa; g_c = gg; }
                } else {
                    lw_layer_optics<FT, true>(a, d, sh, lb, k, kk, d.has_cld && mw.next(k), tau, ssa, gg, pfrac);
                }
                const FT lev_src_dec = sh.ch->Blev[kk * NBMAX + lb.ibnd] * pfrac;
                const FT lev_src_inc = sh.ch->Blev[(kk + 1) * NBMAX + lb.ibnd] * pfrac;
                if (k == nlay - 1) lev_top = lev_src_inc;                    // lev_source[nlev] = lev_src_inc of the last layer
                else add_layer(k + 1, m_sqrt_pos(lev_src_inc * dec_p));      // compute_optical_props.jl:189
                dec_p = lev_src_dec;
                tau_p = tau; ssa_p = ssa; g_p = gg;
                if (DIAG) { tau_pc = tau_c; ssa_pc = ssa_c; g_pc = g_c; cld_p = cld_k; }
            }
        }
        {   // surface source: the Planck function at t_sfc times the Planck fraction of the lowest layer (the loop's last)
            const FT *tp = kernarg_ptr<const FT>(RR_ARG(lk.tot_planck)) + (size_t)a.lk.n_t_plnk * lb.ibnd + sh.misc[d.nwaves];
            sfc_source = (tp[0] * (FT(1) - sh.miscf[0]) + tp[1] * sh.miscf[0]) * pfrac;
        }
        add_layer(0, dec_p);  // lev_source[1] = lev_src_dec of the first layer
        // ---- surface: U_1 = (1 - emis) D_1 + pi emis B_sfc,  D_1 = beta_1 U_1 + delta_1 ----
        const FT alb = FT(1) - emis, src_sfc = Num<FT>::pi() * emis * sfc_source;
        FT U = m_div(src_sfc + alb * S.delta, FT(1) - alb * S.beta);
        FT Uc = DIAG ? m_div(src_sfc + alb * C.delta, FT(1) - alb * C.beta) : FT(0);
        {
            const FT su = seg_sum<BAND>(U * amask), sb = seg_sum<BAND>(S.beta * U * amask);
            if (writer) { acc[0] = su; acc[1] += sb; }
            if (DIAG) {
                const FT suc = seg_sum<BAND>(Uc * amask), sbc = seg_sum<BAND>(C.beta * Uc * amask);
                if (writer) { acc[2] = suc; acc[3] += sbc; }
            }
        }
        // ---- sweep 2, bottom-up.  Rows of three values up to layer k2 (DIAG: up to ctop with the twin's own rows behind them,
        //      above that the all-sky rows serve both), rows of two values above k2.  DBT levels per batch: all scratch
        //      loads are issued before the dependent FMA chain ----
        constexpr int DBT = DIAG ? DB / 2 : DB;  // the twin doubles the batch registers
        for (int phase = (DIAG ? 0 : 1); phase < 2; phase++) {
            const int k_lo = DIAG && phase == 1 ? ctop + 1 : 0, k_end = DIAG && phase == 0 ? ctop + 1 : k2 + 1;
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
                    FT pu[16], pb[16];  // the 2 x 16 g-point sums of the batch in two 16-value reductions (wave_sum16)
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
                            if (kl + j < k_end) { acc[lev * NA] = wu[i]; acc[lev * NA + 1] += wb[i]; }
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
                                if (r < 2) { acc[lev * NA] = wa[i]; acc[lev * NA + 2] = wc[i]; }
                                else { acc[lev * NA + 1] += wa[i]; acc[lev * NA + 3] += wc[i]; }
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
                            if (writer) { acc[lev * NA] = su; acc[lev * NA + 1] += sb; }
                            if (DIAG) {
                                Uc = Ac[j] * Uc + Bc[j];
                                const FT suc = seg_sum<BAND>(Uc * amask), sbc = seg_sum<BAND>(BEc[j] * Uc * amask);
                                if (writer) { acc[lev * NA + 2] = suc; acc[lev * NA + 3] += sbc; }
                            }
                        }
                    }
                }
            }
        }
        // rows of two values: U_{k+1} = T U_k + S_up; D_{k+1} = delta_{k+1} is already in the accumulators
        for (int kl = k2 + 1; kl < nlay; kl += DB) {
            FT A[DB], B[DB];
#pragma unroll
            for (int j = 0; j < DB; j++) {
                const int k = kl + j < nlay ? kl + j : nlay - 1;
                A[j] = sw.get(k, 0); B[j] = sw.get(k, 1);
            }
            if (!BAND) {
                FT pu[16], pc[16];   // 16 g-point sums per stream in one 16-value reduction
                if (kl + 16 <= nlay) {   // a whole batch (wave-uniform): no per-level selects
#pragma unroll
                    for (int j = 0; j < 16; j++) {
                        U = A[j] * U + B[j];
                        if (DIAG) Uc = A[j] * Uc + B[j];
                        pu[j] = U * amask;
                        pc[j] = DIAG ? Uc * amask : FT(0);
                    }
                } else {
#pragma unroll
                    for (int j = 0; j < 16; j++) {
                        const bool in = kl + j < nlay;
                        if (in) { U = A[j] * U + B[j]; if (DIAG) Uc = A[j] * Uc + B[j]; }
                        pu[j] = in ? U * amask : FT(0);
                        pc[j] = DIAG && in ? Uc * amask : FT(0);
                    }
                }
                FT wu[4], wc[4];
                wave_sum16(pu, wu);
                if constexpr (DIAG) wave_sum16(pc, wc);
                if ((lane & 15) == 15) {  // row r holds batch entries j = i + 4 r
#pragma unroll
                    for (int i = 0; i < 4; i++) {
                        const int j = i + 4 * (lane >> 4), lev = kl + j + 1;
                        if (kl + j < nlay) {
                            acc[lev * NA] = wu[i];
                            if constexpr (DIAG) acc[lev * NA + 2] = wc[i];
                        }
                    }
                }
            } else {
#pragma unroll
                for (int j = 0; j < DB; j++) {
                    if (kl + j < nlay) {
                        const int lev = kl + j + 1;
                        U = A[j] * U + B[j];
                        const FT su = seg_sum<BAND>(U * amask);
                        if (writer) acc[lev * NA] = su;
                        if (DIAG) {
                            Uc = A[j] * Uc + B[j];
                            const FT suc = seg_sum<BAND>(Uc * amask);
                            if (writer) acc[lev * NA + 2] = suc;
                        }
                    }
                }
            }
        }
        __syncthreads();
        {
            DevFlux<FT> fl_out = cold_flux<FT>(RR_ARG(fl));
            if (!BAND) fl_out.band_up = fl_out.band_dn = fl_out.band_net = nullptr;
            store_column(fl_out, sh, d, col, ncol, false, cold_gas_view<FT>(RR_ARG(lk), sh, d));
        }
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

// fact of rte_lw_noscat_one_angle! (longwave_noscat.jl:178-180): (1 - t) / tau - t, or its series below tau_thresh.
// Both forms are evaluated and SELECTED: as a branch (what the ternary on expressions compiles to) every angle of every
// level becomes its own exec-masked region, and the register allocator spills around them (183 registers at 3 angles).
template <typename FT>
__device__ __forceinline__ FT noscat_fact(FT tau_loc, FT trans, FT inv_tau_loc, FT tthresh) {
    const FT big = (FT(1) - trans) * inv_tau_loc - trans;
    const FT small = tau_loc * (FT(1.0 / 2.0) + tau_loc * (-FT(1.0 / 3.0) + tau_loc * FT(1.0 / 8.0)));
    return tau_loc > tthresh ? big : small;
}

// ---- no-scattering longwave: rte_lw_noscat_solve! (ext/cuda/rte_longwave_noscat.jl:54-150; bodies
// src/rte/longwave_noscat.jl:45-96 multi-angle driver, :171-205 rte_lw_noscat_one_angle!, :224-301 sources) ----------
// The reference runs, per quadrature angle, a down sweep and an up sweep over optics that a separate pass left in memory.
// Here ONE top-down sweep fuses gas / cloud / aerosol optics, the Planck sources and the downward transport of ALL
// angles (the layer's exp(-tau D_s) and source terms are formed once per angle and serve both directions).  What it leaves
// for the upward transport, I_s[lev] = trans_s I_s[lev-1] + src_up_s, depends on the number of angles:
//   * one angle: (trans, src_up) per level, and the second sweep is one FMA per level;
//   * more: (tau, B_lay, B_lev_top) per level, from which the second sweep re-forms exp(-tau D_s) and the source per angle.
//     Storing 2 values per (level, angle) instead made 3 angles cost twice the time of one: 6 values per level are twice
//     the sweep-scratch traffic of the two-stream kernel (105 GB per launch of 131 072 columns) and that, not the
//     arithmetic, set the pace (29.2 ms; tools/experiments/README.md).
// The flux of a level is sum_s pi w_s I_s, summed over the angles in the lane BEFORE the g-point reduction: one
// wavefront sum per level whatever the number of angles (the reference accumulates angle by angle).
// The level source sqrt(inc[k-1] dec[k]) (compute_optical_props.jl:189) couples neighbouring layers, so the downward
// step of a layer is taken one iteration later, when the layer below has its Planck fraction.
// NANG: quadrature angles (1..4, AngularDiscretizations.jl:34-63); CA: clouds | aerosols << 1, known at compile time.
template <typename FT, int NANG, int CA>
__global__ void __launch_bounds__(256, (sizeof(FT) == 4 ? RR_MIN_WAVES : 2)) lw_noscat_kernel(const LwArgs<FT> a) {
    extern __shared__ __align__(16) char smem[];
    constexpr int CHK = chunk_layers(CA);
    ColShared<FT, CHK> sh;
    ColDims dd = a.dims;
    dd.has_cld = CA & 1; dd.has_aero = (CA >> 1) & 1; dd.diag = 0;
    dd.lw = 1; dd.twostream = 0;
    const ColDims &d = dd;
    carve_shared(sh, smem, d);
    const int nlay = d.nlay, nlev = d.nlev, ncol = a.as.ncol, nb = d.nbnd;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int gl = a.lk.lane_gpt[tid];   // bands dealt to the wavefronts by minor-gas slot count (build_gas)
    const bool active = gl >= 0;
    const int g = active ? gl : a.lk.n_gpt - 1;
    const LaneBand lb = lane_band(a.lk, g);
    constexpr bool ONE = NANG == 1;
    constexpr int NV = ONE ? 2 : 3;  // per level: (trans, src_up), or (tau, B_lay, B_lev_top)
    const Sweep<FT, NV> sw(a.scratch + (size_t)blockIdx.x * (size_t)nlev * NV * SWEEP_LANES, (size_t)nlev * NV * SWEEP_LANES * sizeof(FT), (unsigned)tid);
    const FT amask = active ? FT(1) : FT(0);
    const int nchunk = (nlay + CHK - 1) / CHK;
    cache_small_tables(sh, d, a.lk, a.as);
#define RR_ARG(member) ((unsigned)offsetof(LwArgs<FT>, member))   // cold kernel arguments: device.h
    const FT tthresh = tau_thresh<FT>();
    FT Ds[NANG], rD[NANG], i2f[NANG];
#pragma unroll
    for (int s = 0; s < NANG; s++) { Ds[s] = a.Ds[s]; rD[s] = FT(1) / a.Ds[s]; i2f[s] = Num<FT>::pi() * a.wts[s]; }

    for (int col = blockIdx.x; col < ncol; col = next_column(sh, d, a.queue)) {
        {
            const DevState<FT> as = cold_state<FT>(RR_ARG(as));
            const DevCld<FT> cld = cold_cld<FT>(RR_ARG(cld));
            const DevAero<FT> aero = cold_aero<FT>(RR_ARG(aero));
            prepare_column(sh, d, cold_gas_view<FT>(RR_ARG(lk), sh, d), &cld, &aero, as, col);
        }
        uint64_t m0 = 0, m1 = 0;
        if (d.has_cld) {
            const uint64_t key = mcica_key(kernarg_load<uint64_t>(RR_ARG(seed)), kernarg_load<int64_t>(RR_ARG(col_offset)) + col + 1, g + 1, 0);
            const bool cloudy = build_cloud_mask(sh, d, key, m0, m1) && active;
            const unsigned long long b = __ballot(cloudy);
            if (lane == 0) sh.misc[wave] = __popcll(b);
        }
        const FT emis = kernarg_ptr<const FT>(RR_ARG(sfc_emis))[(size_t)lb.ibnd + (size_t)nb * col];
        const FT *inc_flux = kernarg_ptr<const FT>(RR_ARG(inc_flux));
        const FT inc = inc_flux ? inc_flux[(size_t)col + (size_t)kernarg_load<int>(RR_ARG(inc_ld)) * g] : FT(0);
        FT *acc = sh.acc + (size_t)wave * nlev * 2;
        const bool writer = lane == 63;
        MaskWalk<false> mw(m0, m1, nlay, sh.mask);  // top-down

        // incident intensity inc_flux / pi at every angle (longwave_noscat.jl:245-249)
        FT I[NANG];
        FT f = FT(0);
#pragma unroll
        for (int s = 0; s < NANG; s++) { I[s] = inc_flux ? inc / Num<FT>::pi() : FT(0); f += I[s] * i2f[s]; }
        {
            const FT sd = wave_sum_to_lane63(f * amask);
            if (writer) acc[nlay * 2 + 1] = sd;
        }
        // the layer above the current one, waiting for its lower level source
        FT tr_p[NANG], fa_p[NANG], lay_p = FT(0), dec_p = FT(0);
#pragma unroll
        for (int s = 0; s < NANG; s++) { tr_p[s] = FT(1); fa_p[s] = FT(0); }
        // downward step of layer kp with its lower level source: src = (1 - t) B_lev + 2 fact (B_lay - B_lev), :178-181
        auto step_down = [&](int kp, FT lev_src) {
            FT fs = FT(0);
#pragma unroll
            for (int s = 0; s < NANG; s++) {
                I[s] = tr_p[s] * I[s] + ((FT(1) - tr_p[s]) * lev_src + FT(2) * fa_p[s] * (lay_p - lev_src));
                fs += I[s] * i2f[s];
            }
            const FT sd = wave_sum_to_lane63(fs * amask);
            if (writer) acc[kp * 2 + 1] = sd;
        };
        FT sfc_source = FT(0), pfrac = FT(0);   // pfrac: Planck fraction of the current layer (after the loop: of layer 0)
        for (int c = nchunk - 1; c >= 0; c--) {
            const int k0 = c * CHK, kn = min(CHK, nlay - k0);
            mw.refill(k0 + kn - 1);
            __syncthreads();
            {
                const DevState<FT> as = cold_state<FT>(RR_ARG(as));
                const DevCld<FT> cld = cold_cld<FT>(RR_ARG(cld));
                const DevAero<FT> aero = cold_aero<FT>(RR_ARG(aero));
                prepare_chunk(sh, d, cold_gas_view<FT>(RR_ARG(lk), sh, d), &cld, &aero, as, col, k0, kn, false);
            }
            __syncthreads();
            for (int kk = kn - 1; kk >= 0; kk--) {
                const int k = k0 + kk, r = kk * NBMAX + lb.ibnd;
                FT tau, ssa;
                gas_optics<FT, false>(a.lk, sh, lb, k, kk, nb, tau, ssa, pfrac);
                // OneScalar: clouds and aerosols add their absorption optical depth (cloud_optics.jl:45, aerosol_optics.jl:45)
                if (d.has_cld && mw.next(k)) tau += sh.ch->cld[r].x;
                if (d.has_aero && sh.lay[k].aero_mask) tau += sh.ch->aer[r].x;
                const FT dec = sh.ch->Blev[r] * pfrac;             // B(t_lev[k]) pfrac[k]
                const FT inc_k = sh.ch->Blev[r + NBMAX] * pfrac;   // B(t_lev[k+1]) pfrac[k]
                const FT lay_src = sh.ch->Blay[r] * pfrac;
                // level source at the top of this layer: sqrt(inc[k] dec[k+1]); the top of the atmosphere keeps inc
                const FT lev_up = k == nlay - 1 ? inc_k : m_sqrt_pos(inc_k * dec_p);
                if (k < nlay - 1) step_down(k + 1, lev_up);
                // (1 - t) / (tau D_s): one reciprocal of tau serves every angle (rD[s] = 1 / D_s)
                const FT inv_tau = m_rcp(tau);
#pragma unroll
                for (int s = 0; s < NANG; s++) {
                    const FT tau_loc = tau * Ds[s];
                    const FT trans = m_exp_neg(tau_loc);
                    const FT fact = noscat_fact(tau_loc, trans, inv_tau * rD[s], tthresh);
                    tr_p[s] = trans; fa_p[s] = fact;
                    if (ONE) {
                        sw.put(k, 0, trans);
                        sw.put(k, 1, (FT(1) - trans) * lev_up + FT(2) * fact * (lay_src - lev_up));  // upward source
                    }
                }
                if (!ONE) { sw.put(k, 0, tau); sw.put(k, 1, lay_src); sw.put(k, 2, lev_up); }
                lay_p = lay_src; dec_p = dec;
            }
        }
        {   // compute_optical_props.jl:184-186: the Planck function at t_sfc times the Planck fraction of the lowest layer
            const FT *tp = kernarg_ptr<const FT>(RR_ARG(lk.tot_planck)) + (size_t)a.lk.n_t_plnk * lb.ibnd + sh.misc[d.nwaves];
            sfc_source = (tp[0] * (FT(1) - sh.miscf[0]) + tp[1] * sh.miscf[0]) * pfrac;
        }
        step_down(0, dec_p);  // lev_source[1] = lev_dec of the first layer
        // surface: I_up = I_dn (1 - emis) + emis B_sfc (:262-266)
        {
            FT fs = FT(0);
#pragma unroll
            for (int s = 0; s < NANG; s++) { I[s] = I[s] * (FT(1) - emis) + emis * sfc_source; fs += I[s] * i2f[s]; }
            const FT su = wave_sum_to_lane63(fs * amask);
            if (writer) acc[0] = su;
        }
        // ---- bottom-up: 16 levels per g-point reduction.  One angle: every scratch load of the batch is in flight before
        //      the FMA chain.  More: 4 levels at a time (the per-angle exponentials of 16 levels at once do not fit the
        //      registers: 179 spilled), the compiler barrier keeps the sub-batches from being merged again ----
        constexpr int SB = ONE ? 16 : 4;
        for (int kl = 0; kl < nlay; kl += 16) {
            FT pu[16];
#pragma unroll
            for (int j0 = 0; j0 < 16; j0 += SB) {
                FT X[SB][NV];
#pragma unroll
                for (int j = 0; j < SB; j++) {
                    const int k = kl + j0 + j < nlay ? kl + j0 + j : nlay - 1;
#pragma unroll
                    for (int v = 0; v < NV; v++) X[j][v] = sw.get(k, v);
                }
#pragma unroll
                for (int j = 0; j < SB; j++) {
                    const bool in = kl + j0 + j < nlay;
                    FT fs = FT(0);
                    if (ONE) {
                        if (in) I[0] = X[j][0] * I[0] + X[j][1];
                        fs = I[0] * i2f[0];
                    } else {
                        const FT tau = X[j][0], lay_src = X[j][1], lev_up = X[j][2];
                        const FT inv_tau = m_rcp(tau), dl = FT(2) * (lay_src - lev_up);
#pragma unroll
                        for (int s = 0; s < NANG; s++) {
                            const FT tau_loc = tau * Ds[s];
                            const FT trans = m_exp_neg(tau_loc);
                            const FT fact = noscat_fact(tau_loc, trans, inv_tau * rD[s], tthresh);
                            if (in) I[s] = trans * I[s] + ((FT(1) - trans) * lev_up + fact * dl);
                            fs += I[s] * i2f[s];
                        }
                    }
                    pu[j0 + j] = in ? fs * amask : FT(0);
                }
                if (!ONE) __builtin_amdgcn_sched_barrier(0);  // nothing moves across: the next sub-batch starts with empty hands
            }
            FT wu[4];
            wave_sum16(pu, wu);
            if ((lane & 15) == 15) {  // row r holds batch entries j = i + 4 r
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    const int j = i + 4 * (lane >> 4);
                    if (kl + j < nlay) acc[(kl + j + 1) * 2] = wu[i];
                }
            }
        }
        __syncthreads();
        {
            DevFlux<FT> fl_out = cold_flux<FT>(RR_ARG(fl));
            fl_out.band_up = fl_out.band_dn = fl_out.band_net = nullptr;
            store_column(fl_out, sh, d, col, ncol, false, cold_gas_view<FT>(RR_ARG(lk), sh, d));
        }
        if (d.has_cld && tid == 0) {
            FT *cover = kernarg_ptr<FT>(RR_ARG(as.cld_cover));
            if (cover) {
                int n = 0;
                for (int w = 0; w < d.nwaves; w++) n += sh.misc[w];
                cover[col] = (FT)((double)n / (double)a.lk.n_gpt);
            }
        }
    }
    queue_release(a.queue);
#undef RR_ARG
}

// Gauss-Jacobi-5 secants and weights, src/optics/AngularDiscretizations.jl:41-56
static void angular_discretization(int n, double *Ds, double *wts) {
    static const double mu[4][4] = {{0.6096748751, 0, 0, 0},
                                    {0.2509907356, 0.7908473988, 0, 0},
                                    {0.1024922169, 0.4417960320, 0.8633751621, 0},
                                    {0.0454586727, 0.2322334416, 0.5740198775, 0.9030775973}};
    static const double w[4][4] = {{1, 0, 0, 0},
                                   {0.2300253764, 0.7699746236, 0, 0},
                                   {0.0437820218, 0.3875796738, 0.5686383044, 0},
                                   {0.0092068785, 0.1285704278, 0.4323381850, 0.4298845087}};
    for (int i = 0; i < n; i++) { Ds[i] = 1.0 / mu[n - 1][i]; wts[i] = w[n - 1][i]; }
}

int column_grid(rrtmgp_workspace *ws, int ncol, int threads, size_t lds_bytes, const void *kernel);

template <typename FT>
int launch_lw(rrtmgp_workspace *ws, int twostream, const DevGas<FT> &lk, const DevCld<FT> *cld, const DevAero<FT> *aero,
              const DevState<FT> &as, const FT *sfc_emis, const FT *inc_flux, int inc_ld, const DevFlux<FT> &fl, int n_angles,
              uint64_t seed, int64_t col_offset, int max_int) {
    LwArgs<FT> a{};
    a.lk = lk;
    if (cld) a.cld = *cld;
    if (aero) a.aero = *aero;
    a.as = as; a.fl = fl; a.sfc_emis = sfc_emis; a.inc_flux = inc_flux; a.inc_ld = inc_ld;
    a.seed = seed; a.col_offset = col_offset;
    const int threads = ((fl.band_up ? lk.band_rows * 16 : lk.n_gpt) + 63) / 64 * 64;
    RR_CHECK(threads <= 256, "n_gpt > 256 is not supported");
    RR_CHECK(lk.n_eta <= 255 && lk.n_pp <= 255 && lk.n_t_ref <= 255, "lookup axes longer than 255 are not supported");
    RR_CHECK(lk.n_bnd <= NBMAX, "more than 16 bands per lookup are not supported");
    if (fl.band_up) {
        RR_CHECK(twostream && fl.band_dn, "per-band fluxes need a two-stream solver and both up/dn buffers");
        if (!lk.band_rows) return rrtmgp::set_error(RRTMGP_EUNSUPPORTED, "per-band fluxes: the bands, each padded to 16 g-points, must fit 256 lanes");
    }
    ColDims d{};
    d.nlay = as.nlay; d.nlev = as.nlay + 1;
    d.ngas1 = (as.ngas + 1 > lk.n_gases) ? as.ngas + 1 : lk.n_gases;
    d.nwaves = threads / 64; d.nseg = fl.band_up ? threads / 16 : d.nwaves;
    d.n_t_ref = lk.n_t_ref; d.n_p_ref = lk.n_pp - 1; d.n_t_plnk = lk.n_t_plnk; d.n_gases_ref = lk.n_gases;
    d.nint0 = lk.m_nint[0]; d.nint1 = lk.m_nint[1]; d.nslot0 = lk.m_nslot[0]; d.nslot1 = lk.m_nslot[1]; d.nbnd = lk.n_bnd; d.lw = 1; d.twostream = twostream;
    const bool diag = fl.clear_up != nullptr;
    d.has_cld = cld != nullptr; d.has_aero = aero != nullptr; d.n_acc = diag ? 4 : 2; d.diag = diag; d.max_int = max_int;
    a.dims = d;
    RR_CHECK(twostream || (n_angles >= 1 && n_angles <= 4), "n_gauss_angles must be 1..4");
    a.n_angles = twostream ? 1 : n_angles;
    double Ds[4], wts[4];
    angular_discretization(a.n_angles, Ds, wts);
    for (int i = 0; i < a.n_angles; i++) { a.Ds[i] = (FT)Ds[i]; a.wts[i] = (FT)wts[i]; }
    // the variants instantiated with aerosols known at compile time (CA >= 2 below) prepare chunk_layers(CA) layers at a time
    const bool ca_aero = (twostream && ((aero && (diag || !fl.band_up)) || (diag && chunk_layers(1, true) != CH))) || (!twostream && aero);
    ColShared<FT, chunk_layers(0)> dummy;
    ColShared<FT, chunk_layers(2)> dummy_aero;
    size_t lds = ca_aero ? carve_shared(dummy_aero, (char *)nullptr, d) : carve_shared(dummy, (char *)nullptr, d);
    // main Float32 instances: 8-layer chunks when that is what keeps 4 workgroups resident per CU (160 KB / 4).  Measured:
    // 72 layers LW 21.4 -> 20.2 ms; at 96 layers 3 workgroups with 16-layer chunks are faster (28.0 vs 29.5 ms), hence <= 80 (round 2 kernels; re-measured in round 5, below)
    static const bool no_half = getenv("RRTMGP_HIP_NO_HALF_CHUNKS") != nullptr;  // A/B switch
    // Float64: the 8-layer instances are compiled for RR_F64_HALF_WAVES waves per SIMD (device.h) and taken when their
    // records let that many workgroups share the CU's LDS
    constexpr size_t lds_cap = sizeof(FT) == 4 ? 40960 : (160 * 1024) / RR_F64_HALF_WAVES;
    ColShared<FT, half_chunk_layers<FT>()> dummy_half;
    // (round 5 sweep, tools/experiments/half_rule_sweep.sh, profiles/r05_half_rule_sweep_ab.txt: 80 layers +6.9 %, 84 +6.1 %, 88 +4.5 %,
    // 96 layers LW 11.49 -> 10.98 ms but SW 13.04 -> 13.12: the limit was 80 for both kernels since round 2; now 96 here)
    static const bool force_half_main = getenv("RRTMGP_HIP_FORCE_HALF_CHUNKS") != nullptr;  // A/B switch: whatever the depth
    const bool half = !no_half && (sizeof(FT) == 4 ? (d.nlay <= 96 || force_half_main) : RR_F64_HALF_WAVES > 2) && twostream && !diag && !fl.band_up && !aero &&
                      lds > lds_cap && carve_shared(dummy_half, (char *)nullptr, d) <= lds_cap;
    if (half) lds = carve_shared(dummy_half, (char *)nullptr, d);
    if (diag) {
        RR_CHECK(twostream && cld, "the one-pass clear-sky diagnostic needs the two-stream solver and a cloud lookup");
        RR_CHECK(!fl.band_up, "per-band fluxes and the one-pass clear-sky diagnostic cannot be combined in one launch");
    }
    using Kern = void (*)(const LwArgs<FT>);
    auto noscat_kernel = [&]() -> Kern {
        const int ca = (cld ? 1 : 0) | (aero ? 2 : 0);
#define RR_NOSCAT(N) (ca == 0 ? lw_noscat_kernel<FT, N, 0> : ca == 1 ? lw_noscat_kernel<FT, N, 1> : ca == 2 ? lw_noscat_kernel<FT, N, 2> : lw_noscat_kernel<FT, N, 3>)
        switch (a.n_angles) {
            case 1: return RR_NOSCAT(1);
            case 2: return RR_NOSCAT(2);
            case 3: return RR_NOSCAT(3);
            default: return RR_NOSCAT(4);
        }
#undef RR_NOSCAT
    };
    Kern kern = !twostream ? noscat_kernel()
                : diag     ? (aero ? lw_solve_kernel<FT, true, false, true, 3> : lw_solve_kernel<FT, true, false, true, 1>)
                : fl.band_up ? lw_solve_kernel<FT, true, true, false>
                : (cld && aero) ? lw_solve_kernel<FT, true, false, false, 3>
                : half ? (cld ? lw_solve_kernel<FT, true, false, false, 1, true> : lw_solve_kernel<FT, true, false, false, 0, true>)
                : cld  ? lw_solve_kernel<FT, true, false, false, 1>
                : aero ? lw_solve_kernel<FT, true, false, false, 2> : lw_solve_kernel<FT, true, false, false, 0>;
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
            auto k4 = aero ? lw_solve_kernel<FT, true, false, true, 3, true> : lw_solve_kernel<FT, true, false, true, 1, true>;
            const size_t lds4 = carve_shared(dummy_half, (char *)nullptr, d);
            const int cap3 = column_grid(ws, INT_MAX, threads, lds, (const void *)kern);
            const int cap4 = lds4 <= 160 * 1024 ? column_grid(ws, INT_MAX, threads, lds4, (const void *)k4) : -1;
            static const bool force_half = getenv("RRTMGP_HIP_FORCE_DIAG_HALF") != nullptr;  // A/B switch: whenever they admit one more
            // Round 5 swept the rule at 60 / 64 / 72 / 73 / 80 layers with and without MERRA aerosols (tools/experiments/
            // diag_rule_sweep.sh, profiles/r05_diag_rule_sweep_ab.txt): without aerosols it stands (65-80 layers: the fourth
            // workgroup costs 0.3-1.6 percent); WITH aerosols this kernel gains from it up to 73 layers
            // (SW 15.56 -> 14.55 ms at 72 layers, 17.49 -> 16.55 at 80; LW 11.65 -> 11.52 at 72, but 12.83 -> 13.07 at 80).
            if (cap4 > cap3 && as.ncol > cap3 && (force_half || d.nlay <= 64 || cap3 <= 2 * ws->n_cu || (aero && d.nlay <= 73))) {
                kern = k4; lds = lds4; grid = std::min(as.ncol, cap4);
            }
        } else if (diag && !no_diag_half) {   // the 16-layer records do not fit the LDS at all: the 8-layer ones may
            auto k4 = aero ? lw_solve_kernel<FT, true, false, true, 3, true> : lw_solve_kernel<FT, true, false, true, 1, true>;
            const size_t lds4 = carve_shared(dummy_half, (char *)nullptr, d);
            const int g4 = lds4 <= 160 * 1024 ? column_grid(ws, as.ncol, threads, lds4, (const void *)k4) : -1;
            if (g4 > 0) { kern = k4; lds = lds4; grid = g4; }
        }
    }
    if (grid < 0) return grid;
    const size_t sweep_bytes = (size_t)grid * d.nlev * (!twostream ? (a.n_angles == 1 ? 2 : 3) : diag ? 6 : 3) * SWEEP_LANES * sizeof(FT);
    int rc = scratch_ensure(ws, sweep_bytes);
    if (rc) return rc;
    a.scratch = (FT *)ws->scratch.ptr;
    rc = queue_ensure(ws, 0, &a.queue);   // {next column, workgroups done}: zero between launches (queue_release)
    if (rc) return rc;
    if (ws->timed) RR_HIP(hipEventRecord(ws->ev_start, ws->stream));
    hipLaunchKernelGGL(kern, dim3(grid), dim3(threads), lds, ws->stream, a);
    RR_HIP(hipGetLastError());
    if (ws->timed) RR_HIP(hipEventRecord(ws->ev_stop, ws->stream));
    return RRTMGP_OK;
}

template int launch_lw<float>(rrtmgp_workspace *, int, const DevGas<float> &, const DevCld<float> *,
                              const DevAero<float> *, const DevState<float> &, const float *, const float *, int,
                              const DevFlux<float> &, int, uint64_t, int64_t, int);
template int launch_lw<double>(rrtmgp_workspace *, int, const DevGas<double> &, const DevCld<double> *,
                               const DevAero<double> *, const DevState<double> &, const double *, const double *, int,
                               const DevFlux<double> &, int, uint64_t, int64_t, int);

}  // namespace rrtmgp
