// solve_lw.hip — longwave column kernels (two-stream and no-scattering) for gfx950.
//
// Replaces rte_lw_2stream_solve! / rte_lw_noscat_solve! of the reference
// (ext/cuda/rte_longwave_2stream.jl:48-141, rte_longwave_noscat.jl:54-150; bodies
// src/rte/longwave_2stream.jl, longwave_noscat.jl, src/optics/compute_optical_props.jl:18-245).
//
// One workgroup per column, one lane per g-point, layers in chunks of CH whose
// band-level records are prepared cooperatively in LDS (device.h).  Two-stream: a
// single bottom-up sweep fuses gas/cloud/aerosol optics, Planck sources, the layer
// reflectance/transmittance and the adding step, and leaves 3 numbers per level
// (A, B, albedo; the src term is summed over g-points on the fly) so that the
// top-down sweep is two FMAs per level:
//     F_k = A_k F_{k+1} + B_k ,   U_k = albedo_k F_k + src_k .
// Broadband fluxes are wavefront sums over g-points (fixed DPP order).
#include "device.h"

#ifdef RR_EXP_NO_CHUNK_BARRIER  // timing-only experiment: the chunk loop's barriers removed (races: results are wrong)
#define RR_CHUNK_SYNC() ((void)0)
#else
#define RR_CHUNK_SYNC() __syncthreads()
#endif

namespace rrtmgp {

// lw_2stream_coeffs, src/rte/longwave_2stream.jl:149-222
template <typename FT>
__device__ __forceinline__ void lw_2stream_coeffs(FT tau, FT ssa, FT g, FT lev_src_bot, FT lev_src_top, FT &Rdif,
                                                  FT &Tdif, FT &src_up, FT &src_dn) {
    const FT lw_diff_sec = FT(1.66);
    if (__all(ssa == FT(0))) {
        // Every lane of the wavefront is a purely absorbing layer (gas only: 3 of 4 cells of an all-sky
        // column).  The expressions below are the general ones with ssa = 0 substituted: gamma1 = D,
        // gamma2 = 0, k = D, RT_term = 1/(2D), hence Rdif = 0, Tdif = e^{-D tau}, emis_fac = 1 - Tdif and
        // dBz = dB (1 - Tdif) / (D tau).  No sqrt, one reciprocal instead of three.
        FT e1, om1;
        exp_pair(tau * lw_diff_sec, e1, om1);
        Rdif = FT(0);
        Tdif = e1;
        const FT dB = lev_src_bot - lev_src_top;
        const FT dBz = m_div(dB * om1, lw_diff_sec * tau);
        const bool pos = tau > FT(0);
        src_up = pos ? Num<FT>::pi() * (lev_src_top * om1 - e1 * dB + dBz) : FT(0);
        src_dn = pos ? Num<FT>::pi() * (lev_src_bot * om1 + e1 * dB - dBz) : FT(0);
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
    int *queue;  // next column of the persistent grid (device counter, zeroed before the launch)
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
// workgroup past a quarter of the CU's LDS (Float32, 71-80 layers): 4 resident workgroups instead of 3.
template <typename FT, bool TWOSTREAM, bool BAND, bool DIAG, int CA = -1, bool HALF = false>
__global__ void __launch_bounds__(256, (sizeof(FT) == 4 ? (DIAG ? RR_DIAG_MIN_WAVES : RR_MIN_WAVES) : 2)) lw_solve_kernel(const LwArgs<FT> a) {
    extern __shared__ __align__(16) char smem[];
    constexpr int CHK = HALF ? CH / 2 : chunk_layers(CA, DIAG);  // layers per chunk of LDS records
    ColShared<FT, CHK> sh;
    ColDims dd = a.dims;
    if (CA >= 0) { dd.has_cld = CA & 1; dd.has_aero = (CA >> 1) & 1; }
    dd.diag = DIAG;  // what the host set, as a constant: the other flux set's pointers are never loaded
    DevFlux<FT> fl_out = a.fl;
    if (!BAND) fl_out.band_up = fl_out.band_dn = fl_out.band_net = nullptr;
    const ColDims &d = dd;
    carve_shared(sh, smem, d);
    const int nlay = d.nlay, nlev = d.nlev, ncol = a.as.ncol, nb = d.nbnd;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    // lane -> g-point: the identity, or (per-band fluxes) the band-by-band layout on 16-lane rows
    const int gl = BAND ? a.lk.band_lane_gpt[tid] : (tid < a.lk.n_gpt ? tid : -1);
    const bool active = gl >= 0;
    const int g = active ? gl : a.lk.n_gpt - 1;
    const LaneBand lb = lane_band(a.lk, g);
    constexpr int NV = DIAG ? 6 : 3;  // sweep values per level
    constexpr int NA = DIAG ? 4 : 2;  // accumulated components per level: up, dn (+ clear up, dn)
    Sweep<FT, NV> sw{(char *)(a.scratch + (size_t)blockIdx.x * (size_t)nlev * NV * SWEEP_LANES), (unsigned)(tid * sizeof(FT))};
    const FT amask = active ? FT(1) : FT(0);
    const int nchunk = (nlay + CHK - 1) / CHK;
    const DevGas<FT> lkp = cache_small_tables(sh, d, a.lk, a.as);  // lookup view for the preparation steps

    for (int col = blockIdx.x; col < ncol; col = next_column(sh, d, a.queue)) {
        prepare_column(sh, d, lkp, &a.cld, &a.aero, a.as, col);

        uint64_t m0 = 0, m1 = 0;
        if (d.has_cld) {
            const uint64_t key = mcica_key(a.seed, a.col_offset + col + 1, g + 1, 0);
            const bool cloudy = build_cloud_mask(sh, d, key, m0, m1) && active;
            const unsigned long long b = __ballot(cloudy);
            if (lane == 0) sh.misc[wave] = __popcll(b);
        }
        const FT emis = a.sfc_emis[(size_t)lb.ibnd + (size_t)nb * col];
        const FT inc = a.inc_flux ? a.inc_flux[(size_t)col + (size_t)a.inc_ld * g] : FT(0);
        FT *acc = sh.acc + (size_t)(BAND ? tid >> 4 : wave) * nlev * NA;
        const bool writer = BAND ? (lane & 15) == 15 : lane == 63;
        FT sfc_source = FT(0);
        MaskWalk<true> mw(m0, m1, nlay, sh.mask);  // both LW solvers visit the layers bottom-up

        if (TWOSTREAM) {
            // ---- bottom-up: optics + sources, and one layer behind them coefficients + adding
            //      (compute_optical_props.jl:163-198, longwave_2stream.jl:273-302).  Per level the
            //      sweep keeps A, B, albedo; src_k only enters U_k = albedo_k F_k + src_k additively,
            //      so its g-point sum is taken here instead of being stored. ----
            FT tau_p = FT(0), ssa_p = FT(0), g_p = FT(0);   // optics of layer k-1
            FT lev_src_bot = FT(0), inc_prev = FT(0);         // lev_source[k-1], B(t_lev[k]) * pfrac[k-1]
            FT albedo = FT(1) - emis, src = FT(0);
            // clear-sky twin (DIAG): optics of layer k-1 without the cloud increment, and whether they differ
            FT tau_pc = FT(0), ssa_pc = FT(0), g_pc = FT(0), albedo_c = FT(1) - emis, src_c = FT(0);
            bool cld_p = false;
            // one adding step for one of the two streams: voff / aoff = its slots in the sweep record / accumulators
            auto adding = [&](FT &alb, FT &sr, FT Rdif, FT Tdif, FT src_up, FT src_dn, int kl, int voff, int aoff) {
                const FT denom = m_rcp(FT(1) - Rdif * alb);  // Eq 10
                sw.put(kl, voff, Tdif * denom);                        // A
                sw.put(kl, voff + 1, (Rdif * sr + src_dn) * denom);    // B
                sw.put(kl, voff + 2, alb);
#ifndef RR_EXP_NO_LAYER_SUMS  // timing-only experiment: no g-point sums inside the layer loop
                const FT ss = seg_sum<BAND>(sr * amask);
                if (writer) acc[kl * NA + aoff] = ss;
#endif
                const FT alb_n = Rdif + Tdif * Tdif * alb * denom;  // Eq 9
                sr = src_up + Tdif * denom * (sr + alb * src_dn);   // Eq 11
                alb = alb_n;
            };
            auto add_layer = [&](int kl, FT lev_src_top) {   // layer kl between levels kl and kl+1
                FT Rdif, Tdif, src_up, src_dn;
                lw_2stream_coeffs(tau_p, ssa_p, g_p, lev_src_bot, lev_src_top, Rdif, Tdif, src_up, src_dn);
                adding(albedo, src, Rdif, Tdif, src_up, src_dn, kl, 0, 0);
                if (DIAG) {
                    // same coefficients unless this lane's McICA sample put a cloud in the layer
                    if (cld_p) lw_2stream_coeffs(tau_pc, ssa_pc, g_pc, lev_src_bot, lev_src_top, Rdif, Tdif, src_up, src_dn);
                    adding(albedo_c, src_c, Rdif, Tdif, src_up, src_dn, kl, 3, 2);
                }
            };
            for (int c = 0; c < nchunk; c++) {
                const int k0 = c * CHK, kn = min(CHK, nlay - k0);
                mw.refill(k0);
                RR_CHUNK_SYNC();
#ifdef RR_EXP_PREP_ONCE  // timing-only experiment: chunk records prepared for the first chunk only (barriers kept)
                if (c == 0)
#endif
                prepare_chunk(sh, d, lkp, &a.cld, &a.aero, a.as, col, k0, kn, false);
                RR_CHUNK_SYNC();
                for (int kk = 0; kk < kn; kk++) {
                    const int k = k0 + kk;
                    FT tau, ssa, gg, pfrac;
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
                    FT lev_src_k;
                    if (k == 0) {
                        const FT *tp = a.lk.tot_planck + (size_t)a.lk.n_t_plnk * lb.ibnd + sh.misc[d.nwaves];
                        sfc_source = (tp[0] * (FT(1) - sh.miscf[0]) + tp[1] * sh.miscf[0]) * pfrac;
                        src = Num<FT>::pi() * emis * sfc_source;
                        src_c = src;
                        lev_src_k = lev_src_dec;
                    } else {
                        lev_src_k = m_sqrt_pos(inc_prev * lev_src_dec);  // compute_optical_props.jl:189
                        add_layer(k - 1, lev_src_k);
                    }
                    lev_src_bot = lev_src_k;
                    inc_prev = lev_src_inc;
                    tau_p = tau; ssa_p = ssa; g_p = gg;
                    if (DIAG) { tau_pc = tau_c; ssa_pc = ssa_c; g_pc = g_c; cld_p = cld_k; }
                }
            }
            add_layer(nlay - 1, inc_prev);  // lev_source[nlev] = lev_src_inc of the last layer
            // ---- top-down fluxes (longwave_2stream.jl:304-333) ----
            FT F = inc, Fc = inc;
            {
                const FT su = seg_sum<BAND>((F * albedo + src) * amask), sd = seg_sum<BAND>(F * amask);
                if (writer) { acc[nlay * NA] = su; acc[nlay * NA + 1] = sd; }
                if (DIAG) {
                    const FT suc = seg_sum<BAND>((Fc * albedo_c + src_c) * amask);
                    if (writer) { acc[nlay * NA + 2] = suc; acc[nlay * NA + 3] = sd; }
                }
            }
            constexpr int DBT = DIAG ? DB / 2 : DB;  // the twin doubles the batch registers
            for (int kh = nlay - 1; kh >= 0; kh -= DBT) {
                // DBT levels per batch: all scratch loads are issued before the dependent FMA chain
                FT A[DBT], B[DBT], AL[DBT], Ac[DIAG ? DBT : 1], Bc[DIAG ? DBT : 1], ALc[DIAG ? DBT : 1];
#pragma unroll
                for (int j = 0; j < DBT; j++) {
                    const int k = kh - j >= 0 ? kh - j : 0;
                    A[j] = sw.get(k, 0); B[j] = sw.get(k, 1); AL[j] = sw.get(k, 2);
                    if (DIAG) { Ac[j] = sw.get(k, 3); Bc[j] = sw.get(k, 4); ALc[j] = sw.get(k, 5); }
                }
                if (!BAND && !DIAG && DBT == 16) {
                    // the 2 x 16 g-point sums of the batch in two 16-value reductions (wave_sum16)
                    FT pu[16], pd[16];
#pragma unroll
                    for (int j = 0; j < 16; j++) {
                        const bool in = kh - j >= 0;
                        if (in) F = A[j] * F + B[j];
                        pu[j] = in ? F * AL[j] * amask : FT(0);
                        pd[j] = in ? F * amask : FT(0);
                    }
                    FT wu[4], wd[4];
                    wave_sum16(pu, wu);
                    wave_sum16(pd, wd);
                    if ((lane & 15) == 15) {  // row r holds batch entries j = i + 4 r
#pragma unroll
                        for (int i = 0; i < 4; i++) {
                            const int k = kh - (i + 4 * (lane >> 4));
                            if (k >= 0) { acc[k * NA] += wu[i]; acc[k * NA + 1] = wd[i]; }
                        }
                    }
                } else if (!BAND && DIAG && DBT == 8) {
                    // two streams x (up, dn) x 8 levels: one 16-value reduction per stream
                    FT pa[16], pc[16];
#pragma unroll
                    for (int j = 0; j < 8; j++) {
                        const bool in = kh - j >= 0;
                        if (in) { F = A[j] * F + B[j]; Fc = Ac[j] * Fc + Bc[j]; }
                        pa[j] = in ? F * AL[j] * amask : FT(0);
                        pa[j + 8] = in ? F * amask : FT(0);
                        pc[j] = in ? Fc * ALc[j] * amask : FT(0);
                        pc[j + 8] = in ? Fc * amask : FT(0);
                    }
                    FT wa[4], wc[4];
                    wave_sum16(pa, wa);
                    wave_sum16(pc, wc);
                    if ((lane & 15) == 15) {  // rows 0, 1: up of levels j = i + 4 r; rows 2, 3: dn of j = i + 4 (r - 2)
                        const int r = lane >> 4;
#pragma unroll
                        for (int i = 0; i < 4; i++) {
                            const int k = kh - (i + 4 * (r & 1));
                            if (k >= 0) {
                                if (r < 2) { acc[k * NA] += wa[i]; acc[k * NA + 2] += wc[i]; }
                                else { acc[k * NA + 1] = wa[i]; acc[k * NA + 3] = wc[i]; }
                            }
                        }
                    }
                } else {
#pragma unroll
                    for (int j = 0; j < DBT; j++) {
                        if (kh - j >= 0) {
                            const int k = kh - j;
                            F = A[j] * F + B[j];
                            const FT su = seg_sum<BAND>(F * AL[j] * amask), sd = seg_sum<BAND>(F * amask);
                            if (writer) { acc[k * NA] += su; acc[k * NA + 1] = sd; }
                            if (DIAG) {
                                Fc = Ac[j] * Fc + Bc[j];
                                const FT suc = seg_sum<BAND>(Fc * ALc[j] * amask), sdc = seg_sum<BAND>(Fc * amask);
                                if (writer) { acc[k * NA + 2] += suc; acc[k * NA + 3] = sdc; }
                            }
                        }
                    }
                }
            }
        } else {
            // ---- no-scattering: optics sweep, then one down + one up transport per angle
            //      (longwave_noscat.jl:45-96, 224-301) ----
            FT inc_prev = FT(0);
            for (int c = 0; c < nchunk; c++) {
                const int k0 = c * CHK, kn = min(CHK, nlay - k0);
                mw.refill(k0);
                __syncthreads();
                prepare_chunk(sh, d, lkp, &a.cld, &a.aero, a.as, col, k0, kn, false);
                RR_CHUNK_SYNC();
                for (int kk = 0; kk < kn; kk++) {
                    const int k = k0 + kk;
                    FT tau, ssa, gg, pfrac;
                    lw_layer_optics<FT, false>(a, d, sh, lb, k, kk, d.has_cld && mw.next(k), tau, ssa, gg, pfrac);
                    const FT lev_src_dec = sh.ch->Blev[kk * NBMAX + lb.ibnd] * pfrac;
                    const FT lev_src_inc = sh.ch->Blev[(kk + 1) * NBMAX + lb.ibnd] * pfrac;
                    const FT lay_src = sh.ch->Blay[kk * NBMAX + lb.ibnd] * pfrac;
                    FT lev_src;
                    if (k == 0) {
                        const FT *tp = a.lk.tot_planck + (size_t)a.lk.n_t_plnk * lb.ibnd + sh.misc[d.nwaves];
                        sfc_source = (tp[0] * (FT(1) - sh.miscf[0]) + tp[1] * sh.miscf[0]) * pfrac;
                        lev_src = lev_src_dec;
                    } else {
                        lev_src = m_sqrt_pos(inc_prev * lev_src_dec);
                    }
                    sw.put(k, 0, tau);
                    sw.put(k, 1, lay_src);
                    sw.put(k, 2, lev_src);
                    inc_prev = lev_src_inc;
                }
            }
            sw.put(nlay, 2, inc_prev);
            const FT tthresh = tau_thresh<FT>();
            for (int imu = 0; imu < a.n_angles; imu++) {
                const FT Ds = a.Ds[imu], w_mu = a.wts[imu];
                const FT i2f = Num<FT>::pi() * w_mu;
                FT I = a.inc_flux ? inc / Num<FT>::pi() : FT(0);
                const bool first = imu == 0;
                {
                    const FT sd = seg_sum<BAND>(I * i2f * amask);
                    if (writer) acc[nlay * 2 + 1] = first ? sd : acc[nlay * 2 + 1] + sd;
                }
                for (int k = nlay - 1; k >= 0; k--) {
                    const FT tau_loc = sw.get(k, 0) * Ds;
                    const FT trans = m_exp(-tau_loc);
                    const FT lay_src = sw.get(k, 1), lev_src = sw.get(k, 2);
                    const FT fact = (tau_loc > tthresh)
                                        ? ((FT(1) - trans) / tau_loc - trans)
                                        : tau_loc * (FT(1.0 / 2.0) + tau_loc * (-FT(1.0 / 3.0) + tau_loc * FT(1.0 / 8.0)));
                    I = trans * I + ((FT(1) - trans) * lev_src + FT(2) * fact * (lay_src - lev_src));
                    const FT sd = seg_sum<BAND>(I * i2f * amask);
                    if (writer) acc[k * 2 + 1] = first ? sd : acc[k * 2 + 1] + sd;
                }
                I = I * (FT(1) - emis) + emis * sfc_source;
                {
                    const FT su = seg_sum<BAND>(I * i2f * amask);
                    if (writer) acc[0] = first ? su : acc[0] + su;
                }
                for (int lev = 1; lev <= nlay; lev++) {
                    const FT tau_loc = sw.get(lev - 1, 0) * Ds;
                    const FT trans = m_exp(-tau_loc);
                    const FT lay_src = sw.get(lev - 1, 1), lev_src = sw.get(lev, 2);
                    const FT fact = (tau_loc > tthresh)
                                        ? ((FT(1) - trans) / tau_loc - trans)
                                        : tau_loc * (FT(1.0 / 2.0) + tau_loc * (-FT(1.0 / 3.0) + tau_loc * FT(1.0 / 8.0)));
                    I = trans * I + ((FT(1) - trans) * lev_src + FT(2) * fact * (lay_src - lev_src));
                    const FT su = seg_sum<BAND>(I * i2f * amask);
                    if (writer) acc[lev * 2] = first ? su : acc[lev * 2] + su;
                }
            }
        }
        __syncthreads();
        store_column(fl_out, sh, d, col, ncol, false, a.lk);
        if (d.has_cld && a.as.cld_cover && tid == 0) {
            int n = 0;
            for (int w = 0; w < d.nwaves; w++) n += sh.misc[w];
            a.as.cld_cover[col] = (FT)((double)n / (double)a.lk.n_gpt);  // exact quotient (the Float32 build divides in 2.5 ulp)
        }
    }
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
    a.n_angles = twostream ? 1 : n_angles;
    double Ds[4], wts[4];
    angular_discretization(a.n_angles, Ds, wts);
    for (int i = 0; i < a.n_angles; i++) { a.Ds[i] = (FT)Ds[i]; a.wts[i] = (FT)wts[i]; }
    // the variants instantiated with aerosols known at compile time (CA >= 2 below) prepare chunk_layers(CA) layers at a time
    const bool ca_aero = twostream && ((aero && (diag || !fl.band_up)) || (diag && chunk_layers(1, true) != CH));
    ColShared<FT, chunk_layers(0)> dummy;
    ColShared<FT, chunk_layers(2)> dummy_aero;
    size_t lds = ca_aero ? carve_shared(dummy_aero, (char *)nullptr, d) : carve_shared(dummy, (char *)nullptr, d);
    // main Float32 instances: 8-layer chunks when that is what keeps 4 workgroups resident per CU (160 KB / 4).  Measured:
    // 72 layers LW 21.4 -> 20.2 ms; at 96 layers 3 workgroups with 16-layer chunks are faster (28.0 vs 29.5 ms), hence <= 80
    static const bool no_half = getenv("RRTMGP_HIP_NO_HALF_CHUNKS") != nullptr;  // A/B switch
    const bool half = !no_half && sizeof(FT) == 4 && twostream && !diag && !fl.band_up && !aero && d.nlay <= 80 && lds > 40960 &&
                      carve_shared(dummy_aero, (char *)nullptr, d) <= 40960;
    if (half) lds = carve_shared(dummy_aero, (char *)nullptr, d);
    if (diag) {
        RR_CHECK(twostream && cld, "the one-pass clear-sky diagnostic needs the two-stream solver and a cloud lookup");
        RR_CHECK(!fl.band_up, "per-band fluxes and the one-pass clear-sky diagnostic cannot be combined in one launch");
    }
    auto kern = !twostream ? lw_solve_kernel<FT, false, false, false>
                : diag     ? (aero ? lw_solve_kernel<FT, true, false, true, 3> : lw_solve_kernel<FT, true, false, true, 1>)
                : fl.band_up ? lw_solve_kernel<FT, true, true, false>
                : (cld && aero) ? lw_solve_kernel<FT, true, false, false, 3>
                : half ? (cld ? lw_solve_kernel<FT, true, false, false, 1, true> : lw_solve_kernel<FT, true, false, false, 0, true>)
                : cld  ? lw_solve_kernel<FT, true, false, false, 1>
                : aero ? lw_solve_kernel<FT, true, false, false, 2> : lw_solve_kernel<FT, true, false, false, 0>;
    const int grid = column_grid(ws, as.ncol, threads, lds, (const void *)kern);
    if (grid < 0) return grid;
    const size_t sweep_bytes = (size_t)grid * d.nlev * (diag ? 6 : 3) * SWEEP_LANES * sizeof(FT);
    int rc = scratch_ensure(ws, sweep_bytes + 256);
    if (rc) return rc;
    a.scratch = (FT *)ws->scratch.ptr;
    a.queue = (int *)((char *)ws->scratch.ptr + sweep_bytes);
    RR_HIP(hipMemsetAsync(a.queue, 0, sizeof(int), ws->stream));
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
