// gray.hip — gray-atmosphere solvers and state-preparation kernels for gfx950.
//
// Gray radiation has a single g-point, so these follow the reference's own
// mapping (one lane per column; ext/cuda/rte_longwave_2stream.jl:1-39,
// rte_longwave_noscat.jl:1-46, rte_shortwave_2stream.jl:1-50, rte_shortwave_noscat.jl:1-45)
// with the per-level sweep state in a column-contiguous scratch so that
// neighbouring lanes touch neighbouring addresses.  compute_col_gas! /
// compute_relative_humidity! are ext/cuda/optics.jl:2-80.
#include "device.h"

namespace rrtmgp {

template <typename FT>
__device__ __forceinline__ FT gray_tau_lw(const GrayArgs &ga, FT p0, FT dp, FT p, FT lat) {
    // compute_gray_optical_thickness_lw, src/optics/gray_optics_kernels.jl:171-236
    if (ga.otp_kind == 0) {
        const FT alpha = (FT)ga.otp[0], te = (FT)ga.otp[1], tt = (FT)ga.otp[2], dt = (FT)ga.otp[3];
        const FT sl = m_sin(lat / FT(180) * Num<FT>::pi());
        const FT ts_by_tt = (te + dt * (FT(1) / FT(3) - sl * sl)) / tt;
        const FT p4 = ts_by_tt * ts_by_tt * ts_by_tt * ts_by_tt;
        const FT d0 = p4 - FT(1);
        return m_abs((alpha * d0 * m_exp(alpha * m_log(p / p0)) / p) * dp);  // pow_fast, Numerics.jl:72
    }
    const FT alpha = (FT)ga.otp[0], fl = (FT)ga.otp[1], te = (FT)ga.otp[2], tp = (FT)ga.otp[3];
    const FT sigma = p / p0;
    const FT sl = m_sin(lat / FT(180) * Num<FT>::pi());
    const FT s4 = (sigma * sigma) * (sigma * sigma);
    return m_abs((alpha * dp / p) * (fl * sigma + (FT(1) - fl) * FT(4) * s4) * (te + (tp - te) * (sl * sl)));
}

template <typename FT>
__device__ __forceinline__ FT gray_tau_sw(const GrayArgs &ga, FT p0, FT dp, FT p) {
    if (ga.otp_kind == 0) return FT(0);  // gray_optics_kernels.jl:205-208
    return m_abs(FT(2) * (FT)ga.otp[4] * (p / p0) * (dp / p0));  // :241-251
}

template <typename FT>
__device__ __forceinline__ void put(const DevFlux<FT> &fl, int col, int ncol, int lev, int nlev, FT up, FT dn, FT dir,
                                    bool has_dir) {
    FT net = up - dn;
    if (fl.metric) {
        const FT m = fl.metric[(size_t)nlev * col + lev];
        up *= m; dn *= m; net *= m; dir *= m;
    }
    const size_t o = fl.layout == RRTMGP_LAYOUT_NCOL_NLEV ? (size_t)col + (size_t)fl.ld * lev : (size_t)lev + (size_t)nlev * col;
    fl.up[o] = up; fl.dn[o] = dn; fl.net[o] = net;
    if (has_dir && fl.dir) fl.dir[o] = dir;
}

// lw_2stream_coeffs restated for the gray path (ssa = g = 0 always; gray_optics_kernels.jl:117-119)
template <typename FT>
__device__ __forceinline__ void gray_lw_coeffs(FT tau, FT bot, FT top, FT &Rdif, FT &Tdif, FT &src_up, FT &src_dn) {
    const FT D = FT(1.66), ssa = FT(0), g = FT(0);
    const FT gamma1 = D * (FT(1) - FT(0.5) * ssa * (FT(1) + g));
    const FT gamma2 = D * FT(0.5) * ssa * (FT(1) - g);
    const FT k = m_sqrt(m_max(D * (FT(1) - ssa) * (gamma1 + gamma2), k_min<FT>()));
    FT e1, om1;
    exp_pair(tau * k, e1, om1);
    const FT om2 = om1 * (FT(1) + e1);
    const FT RT = FT(1) / (k * (FT(1) + e1 * e1) + gamma1 * om2);
    Rdif = RT * gamma2 * om2;
    Tdif = RT * FT(2) * k * e1;
    if (tau > FT(0)) {
        const FT dB = bot - top, gs = gamma1 + gamma2, ope = FT(1) + e1;
        const FT emis_fac = om1 * (k * om1 + D * (FT(1) - ssa) * ope) * RT;
        const FT dBz = dB * (om1 / tau) * (k * om1 + gs * ope) * RT / m_max(gs, Num<FT>::eps());
        src_up = Num<FT>::pi() * (top * emis_fac - Tdif * dB + dBz);
        src_dn = Num<FT>::pi() * (bot * emis_fac + Tdif * dB - dBz);
    } else {
        src_up = FT(0); src_dn = FT(0);
    }
}

template <typename FT, bool TWOSTREAM>
__global__ void gray_lw_kernel(int ncol, int nlay, GrayArgs ga, const FT *lat, const FT *p_lay, const FT *p_lev,
                               const FT *t_lay, const FT *t_lev, const FT *t_sfc, const FT *sfc_emis, const FT *inc_flux,
                               DevFlux<FT> fl, FT *scratch) {
    const int col = blockIdx.x * blockDim.x + threadIdx.x;
    if (col >= ncol) return;
    const int nlev = nlay + 1;
    const FT sbc = (FT)ga.stefan, pi = Num<FT>::pi();
    auto S = [&](int a, int lev) -> FT & { return scratch[((size_t)a * nlev + lev) * ncol + col]; };
    const FT la = lat[col];
    const FT p0 = p_lev[(size_t)nlev * col];
    const FT ts = t_sfc[col];
    const FT sfc_source = sbc * (ts * ts * ts * ts) / pi;
    const FT emis = sfc_emis[col];
    const FT inc = inc_flux ? inc_flux[col] : FT(0);
    // optics + sources (gray_optics_kernels.jl:14-126): S0 = tau, S1 = lev_source, S2 = lay_source
    FT p_lo = p0, t_dec = t_lev[(size_t)nlev * col], inc_prev = FT(0);
    for (int k = 0; k < nlay; k++) {
        const FT p_hi = p_lev[(size_t)nlev * col + k + 1];
        S(0, k) = gray_tau_lw(ga, p0, p_hi - p_lo, p_lay[(size_t)nlay * col + k], la);
        p_lo = p_hi;
        const FT t_inc = t_lev[(size_t)nlev * col + k + 1], tl = t_lay[(size_t)nlay * col + k];
        S(2, k) = sbc * (tl * tl * tl * tl) / pi;
        const FT src_inc = sbc * (t_inc * t_inc * t_inc * t_inc) / pi;
        const FT src_dec = sbc * (t_dec * t_dec * t_dec * t_dec) / pi;
        S(1, k) = k == 0 ? src_dec : m_sqrt(inc_prev * src_dec);
        inc_prev = src_inc;
        t_dec = t_inc;
    }
    S(1, nlay) = inc_prev;
    if (TWOSTREAM) {
        // rte_lw_2stream!, longwave_2stream.jl:243-334: S3 = albedo, S4 = src
        FT albedo = FT(1) - emis, src = pi * emis * sfc_source;
        S(3, 0) = albedo; S(4, 0) = src;
        for (int k = 0; k < nlay; k++) {
            FT Rdif, Tdif, su, sd;
            gray_lw_coeffs(S(0, k), S(1, k), S(1, k + 1), Rdif, Tdif, su, sd);
            const FT denom = FT(1) / (FT(1) - Rdif * albedo);
            const FT albedo_n = Rdif + Tdif * Tdif * albedo * denom;
            src = su + Tdif * denom * (src + albedo * sd);
            albedo = albedo_n;
            S(3, k + 1) = albedo; S(4, k + 1) = src;
        }
        FT F = inc;
        put(fl, col, ncol, nlay, nlev, F * albedo + src, F, FT(0), false);
        for (int k = nlay - 1; k >= 0; k--) {
            FT Rdif, Tdif, su, sd;
            gray_lw_coeffs(S(0, k), S(1, k), S(1, k + 1), Rdif, Tdif, su, sd);
            const FT alb = S(3, k), sr = S(4, k);
            const FT denom = FT(1) / (FT(1) - Rdif * alb);
            F = (Tdif * F + Rdif * sr + sd) * denom;
            put(fl, col, ncol, k, nlev, F * alb + sr, F, FT(0), false);
        }
    } else {
        // rte_lw_noscat_one_angle!, longwave_noscat.jl:224-301, single angle (w = 1)
        const FT Ds = (FT)(1.0 / 0.6096748751), i2f = pi * FT(1);
        const FT tthresh = tau_thresh<FT>();
        FT I = inc_flux ? inc / pi : FT(0);
        S(3, nlay) = I * i2f;  // flux_dn
        for (int k = nlay - 1; k >= 0; k--) {
            const FT tau_loc = S(0, k) * Ds, trans = m_exp(-tau_loc);
            const FT fact = (tau_loc > tthresh) ? ((FT(1) - trans) / tau_loc - trans)
                                                : tau_loc * (FT(1.0 / 2.0) + tau_loc * (-FT(1.0 / 3.0) + tau_loc * FT(1.0 / 8.0)));
            I = trans * I + ((FT(1) - trans) * S(1, k) + FT(2) * fact * (S(2, k) - S(1, k)));
            S(3, k) = I * i2f;
        }
        I = I * (FT(1) - emis) + emis * sfc_source;
        put(fl, col, ncol, 0, nlev, I * i2f, S(3, 0), FT(0), false);
        for (int lev = 1; lev <= nlay; lev++) {
            const FT tau_loc = S(0, lev - 1) * Ds, trans = m_exp(-tau_loc);
            const FT fact = (tau_loc > tthresh) ? ((FT(1) - trans) / tau_loc - trans)
                                                : tau_loc * (FT(1.0 / 2.0) + tau_loc * (-FT(1.0 / 3.0) + tau_loc * FT(1.0 / 8.0)));
            I = trans * I + ((FT(1) - trans) * S(1, lev) + FT(2) * fact * (S(2, lev - 1) - S(1, lev)));
            put(fl, col, ncol, lev, nlev, I * i2f, S(3, lev), FT(0), false);
        }
    }
}

// gray two-stream SW coefficients: ssa = g = 0 (sw_2stream_coeffs with those arguments)
template <typename FT>
__device__ __forceinline__ void gray_sw_coeffs(FT tau, FT mu0, FT &Rdir, FT &Tdir, FT &Rdif, FT &Tdif) {
    const FT ssa = FT(0), g = FT(0);
    const FT gamma1 = (FT(8) - ssa * (FT(5) + FT(3) * g)) * FT(0.25);
    const FT gamma2 = FT(3) * (ssa * (FT(1) - g)) * FT(0.25);
    const FT gamma3 = (FT(2) - (FT(3) * mu0) * g) * FT(0.25);
    const FT gamma4 = FT(1) - gamma3;
    const FT alpha1 = gamma1 * gamma4 + gamma2 * gamma3, alpha2 = gamma1 * gamma3 + gamma2 * gamma4;
    const FT k = m_sqrt(m_max(FT(2) * (FT(1) - ssa) * (gamma1 + gamma2), k_min<FT>()));
    FT e1, om1;
    exp_pair(tau * k, e1, om1);
    const FT e2 = e1 * e1, om2 = om1 * (FT(1) + e1);
    FT RT = FT(1) / (k * (FT(1) + e2) + gamma1 * om2);
    Rdif = RT * gamma2 * om2;
    Tdif = RT * FT(2) * k * e1;
    const FT T0 = m_exp(-tau / m_max(mu0, mu0_min<FT>()));
    FT k_mu = k * mu0, k_mu2 = k_mu * k_mu;
    const FT diff = FT(1) - k_mu2;
    if (m_abs(diff) < resonance_window<FT>()) {
        k_mu2 = diff >= FT(0) ? FT(1) - resonance_window<FT>() : FT(1) + resonance_window<FT>();
        k_mu = m_sqrt(k_mu2);
    }
    const FT kg3 = k * gamma3, kg4 = k * gamma4;
    RT = ssa * RT / (FT(1) - k_mu2);
    const FT Ru = RT * ((FT(1) - k_mu) * (alpha2 + kg3) - (FT(1) + k_mu) * (alpha2 - kg3) * e2 -
                        FT(2) * (kg3 - alpha2 * k_mu) * e1 * T0);
    const FT Tu = -RT * ((FT(1) + k_mu) * (alpha1 + kg4) * T0 - (FT(1) - k_mu) * (alpha1 - kg4) * e2 * T0 -
                         FT(2) * (kg4 + alpha1 * k_mu) * e1);
    Rdir = m_max(FT(0), Ru);
    Tdir = m_max(FT(0), Tu);
    const FT av = m_max(FT(0), FT(1) - T0), tot = Rdir + Tdir;
    if (tot > av) {
        const FT sc = av / m_max(Num<FT>::eps(), tot);
        Rdir *= sc; Tdir *= sc;
    }
}

template <typename FT, bool TWOSTREAM>
__global__ void gray_sw_kernel(int ncol, int nlay, GrayArgs ga, const FT *p_lay, const FT *p_lev, const FT *cos_zenith,
                               const FT *toa_flux, const FT *alb_dir, const FT *alb_dif, DevFlux<FT> fl, FT *scratch) {
    const int col = blockIdx.x * blockDim.x + threadIdx.x;
    if (col >= ncol) return;
    const int nlev = nlay + 1;
    auto S = [&](int a, int lev) -> FT & { return scratch[((size_t)a * nlev + lev) * ncol + col]; };
    const FT mu0 = cos_zenith[col];
    if (!(mu0 > FT(0))) {  // set_flux_to_zero!, Fluxes.jl:268-283
        for (int lev = 0; lev < nlev; lev++) put(fl, col, ncol, lev, nlev, FT(0), FT(0), FT(0), true);
        return;
    }
    const FT p0 = p_lev[(size_t)nlev * col];
    FT p_lo = p0;
    for (int k = 0; k < nlay; k++) {
        const FT p_hi = p_lev[(size_t)nlev * col + k + 1];
        S(0, k) = gray_tau_sw(ga, p0, p_hi - p_lo, p_lay[(size_t)nlay * col + k]);
        p_lo = p_hi;
    }
    const FT dir_top = toa_flux[col] * FT(1) * mu0;
    if (!TWOSTREAM) {
        FT dir = dir_top;  // rte_sw_noscat!, shortwave_noscat.jl:120-148
        put(fl, col, ncol, nlay, nlev, FT(0), dir, dir, true);
        for (int k = nlay - 1; k >= 0; k--) {
            dir = dir * m_exp(-S(0, k) / m_max(mu0, mu0_min<FT>()));
            put(fl, col, ncol, k, nlev, FT(0), dir, dir, true);
        }
        return;
    }
    // rte_sw_2stream!, shortwave_2stream.jl:300-392: S1 = dir, S2 = albedo, S3 = src
    const FT inv_mu0 = FT(1) / m_max(mu0, mu0_min<FT>());
    S(1, nlay) = dir_top;
    FT tau_cum = FT(0);
    for (int k = nlay - 1; k >= 0; k--) {
        tau_cum += S(0, k);
        S(1, k) = dir_top * m_exp(-tau_cum * inv_mu0);
    }
    FT albedo = alb_dif[col], src = S(1, 0) * alb_dir[col];
    S(2, 0) = albedo; S(3, 0) = src;
    for (int k = 0; k < nlay; k++) {
        FT Rdir, Tdir, Rdif, Tdif;
        gray_sw_coeffs(S(0, k), mu0, Rdir, Tdir, Rdif, Tdif);
        const FT denom = FT(1) / (FT(1) - Rdif * albedo);
        const FT albedo_n = Rdif + Tdif * Tdif * albedo * denom;
        const FT d1 = S(1, k + 1);
        src = Rdir * d1 + Tdif * denom * (src + albedo * (Tdir * d1));
        albedo = albedo_n;
        S(2, k + 1) = albedo; S(3, k + 1) = src;
    }
    FT F = FT(0);
    put(fl, col, ncol, nlay, nlev, F * albedo + src, F + dir_top, dir_top, true);
    for (int k = nlay - 1; k >= 0; k--) {
        FT Rdir, Tdir, Rdif, Tdif;
        gray_sw_coeffs(S(0, k), mu0, Rdir, Tdir, Rdif, Tdif);
        const FT alb = S(2, k), sr = S(3, k);
        const FT denom = FT(1) / (FT(1) - Rdif * alb);
        F = (Tdif * F + Rdif * sr + Tdir * S(1, k + 1)) * denom;
        put(fl, col, ncol, k, nlev, F * alb + sr, F + S(1, k), S(1, k), true);
    }
}

// compute_col_gas_kernel!, src/optics/gas_optics.jl:16-47.  The arrays arrive as the reference passes them: dense or
// strided 2-D views (View2<FT>: element (k, col) at p[k * s0 + col * s1]).
template <typename FT>
__global__ void col_gas_kernel(int ncol, int nlay, View2<const FT> p_lev, View2<FT> col_dry, FT mol_m_dry, FT mol_m_h2o,
                               FT avogadro, FT helmert1, View2<const FT> vmr_h2o, const FT *lat) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)ncol * nlay) return;
    const int col = (int)(i / nlay), k = (int)(i - (size_t)col * nlay);
    const FT g0 = lat ? helmert1 - FT(0.02586) * m_cos(FT(2) * Num<FT>::pi() * lat[col] / FT(180)) : helmert1;
    const FT dp = p_lev(k, col) - p_lev(k + 1, col);
    const FT h2o = vmr_h2o.p ? vmr_h2o(k, col) : FT(0);
    const FT m_air = (mol_m_dry + mol_m_h2o * h2o);
    col_dry(k, col) = (dp * avogadro / (FT(100 * 100) * m_air * g0));
}

// compute_relative_humidity_kernel!, src/optics/gas_optics.jl:58-80
template <typename FT>
__global__ void rel_hum_kernel(int ncol, int nlay, View2<FT> rh, View2<const FT> p_lay, View2<const FT> t_lay, FT mwd,
                               View2<const FT> vmr_h2o) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)ncol * nlay) return;
    const int col = (int)(i / nlay), k = (int)(i - (size_t)col * nlay);
    const FT mmr = vmr_h2o(k, col) * mwd;
    const FT q = mmr / (FT(1) + mmr);
    const FT q_tmp = m_max(FT(1e-7), q);
    const FT t = t_lay(k, col);
    const FT es = m_exp((FT(17.67) * (t - FT(273.16))) / (t - FT(29.65)));
    rh(k, col) = m_max(FT(0.01) * (FT(0.263) * p_lay(k, col) * q_tmp) / es, FT(0));
}

// compute_gray_heating_rate_kernel!, src/optics/GrayAtmosphere.jl:152-167
template <typename FT>
__global__ void heating_rate_kernel(int ncol, int nlay, View2<FT> hr_lay, View2<const FT> flux_net, View2<const FT> p_lev,
                                    FT grav, FT cp_d) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)ncol * nlay) return;
    const int col = (int)(i / nlay), k = (int)(i - (size_t)col * nlay);
    hr_lay(k, col) = grav * (flux_net(k + 1, col) - flux_net(k, col)) / (p_lev(k + 1, col) - p_lev(k, col)) / cp_d;
}
template <typename FT>
int launch_heating_rate(rrtmgp_workspace *ws, int ncol, int nlay, View2<FT> hr_lay, View2<const FT> flux_net,
                        View2<const FT> p_lev, double grav, double cp_d) {
    const size_t n = (size_t)ncol * nlay;
    const int tx = 256;
    hipLaunchKernelGGL((heating_rate_kernel<FT>), dim3((unsigned)((n + tx - 1) / tx)), dim3(tx), 0, ws->stream, ncol, nlay, hr_lay,
                       flux_net, p_lev, (FT)grav, (FT)cp_d);
    RR_HIP(hipGetLastError());
    return RRTMGP_OK;
}
template int launch_heating_rate<float>(rrtmgp_workspace *, int, int, View2<float>, View2<const float>, View2<const float>, double, double);
template int launch_heating_rate<double>(rrtmgp_workspace *, int, int, View2<double>, View2<const double>, View2<const double>, double, double);

template <typename FT>
int launch_gray_lw(rrtmgp_workspace *ws, int twostream, int ncol, int nlay, const GrayArgs &ga, const FT *lat,
                   const FT *p_lay, const FT *p_lev, const FT *t_lay, const FT *t_lev, const FT *t_sfc,
                   const FT *sfc_emis, const FT *inc_flux, const DevFlux<FT> &fl) {
    int rc = scratch_ensure(ws, (size_t)5 * (nlay + 1) * ncol * sizeof(FT));
    if (rc) return rc;
    const int tx = 64, bx = (ncol + tx - 1) / tx;
    if (ws->timed) RR_HIP(hipEventRecord(ws->ev_start, ws->stream));
    if (twostream)
        hipLaunchKernelGGL((gray_lw_kernel<FT, true>), dim3(bx), dim3(tx), 0, ws->stream, ncol, nlay, ga, lat, p_lay, p_lev,
                           t_lay, t_lev, t_sfc, sfc_emis, inc_flux, fl, (FT *)ws->scratch.ptr);
    else
        hipLaunchKernelGGL((gray_lw_kernel<FT, false>), dim3(bx), dim3(tx), 0, ws->stream, ncol, nlay, ga, lat, p_lay, p_lev,
                           t_lay, t_lev, t_sfc, sfc_emis, inc_flux, fl, (FT *)ws->scratch.ptr);
    RR_HIP(hipGetLastError());
    if (ws->timed) RR_HIP(hipEventRecord(ws->ev_stop, ws->stream));
    return RRTMGP_OK;
}

template <typename FT>
int launch_gray_sw(rrtmgp_workspace *ws, int twostream, int ncol, int nlay, const GrayArgs &ga, const FT *p_lay,
                   const FT *p_lev, const FT *cos_zenith, const FT *toa_flux, const FT *alb_dir, const FT *alb_dif,
                   const DevFlux<FT> &fl) {
    int rc = scratch_ensure(ws, (size_t)4 * (nlay + 1) * ncol * sizeof(FT));
    if (rc) return rc;
    const int tx = 64, bx = (ncol + tx - 1) / tx;
    if (ws->timed) RR_HIP(hipEventRecord(ws->ev_start, ws->stream));
    if (twostream)
        hipLaunchKernelGGL((gray_sw_kernel<FT, true>), dim3(bx), dim3(tx), 0, ws->stream, ncol, nlay, ga, p_lay, p_lev,
                           cos_zenith, toa_flux, alb_dir, alb_dif, fl, (FT *)ws->scratch.ptr);
    else
        hipLaunchKernelGGL((gray_sw_kernel<FT, false>), dim3(bx), dim3(tx), 0, ws->stream, ncol, nlay, ga, p_lay, p_lev,
                           cos_zenith, toa_flux, alb_dir, alb_dif, fl, (FT *)ws->scratch.ptr);
    RR_HIP(hipGetLastError());
    if (ws->timed) RR_HIP(hipEventRecord(ws->ev_stop, ws->stream));
    return RRTMGP_OK;
}

template <typename FT>
int launch_col_gas(rrtmgp_workspace *ws, int ncol, int nlay, View2<const FT> p_lev, View2<FT> col_dry, const rrtmgp_params &ps,
                   View2<const FT> vmr_h2o, const FT *lat) {
    const size_t n = (size_t)ncol * nlay;
    const int tx = 256;
    hipLaunchKernelGGL((col_gas_kernel<FT>), dim3((unsigned)((n + tx - 1) / tx)), dim3(tx), 0, ws->stream, ncol, nlay, p_lev,
                       col_dry, (FT)ps.molmass_dryair, (FT)ps.molmass_water, (FT)ps.avogad, (FT)ps.grav, vmr_h2o, lat);
    RR_HIP(hipGetLastError());
    return RRTMGP_OK;
}

template <typename FT>
int launch_rel_hum(rrtmgp_workspace *ws, int ncol, int nlay, View2<FT> rh, View2<const FT> p_lay, View2<const FT> t_lay,
                   const rrtmgp_params &ps, View2<const FT> vmr_h2o) {
    const size_t n = (size_t)ncol * nlay;
    const int tx = 256;
    const FT mwd = (FT)ps.molmass_water / (FT)ps.molmass_dryair;
    hipLaunchKernelGGL((rel_hum_kernel<FT>), dim3((unsigned)((n + tx - 1) / tx)), dim3(tx), 0, ws->stream, ncol, nlay, rh, p_lay,
                       t_lay, mwd, vmr_h2o);
    RR_HIP(hipGetLastError());
    return RRTMGP_OK;
}

#define INST(FT)                                                                                                       \
    template int launch_gray_lw<FT>(rrtmgp_workspace *, int, int, int, const GrayArgs &, const FT *, const FT *,       \
                                    const FT *, const FT *, const FT *, const FT *, const FT *, const FT *,            \
                                    const DevFlux<FT> &);                                                              \
    template int launch_gray_sw<FT>(rrtmgp_workspace *, int, int, int, const GrayArgs &, const FT *, const FT *,       \
                                    const FT *, const FT *, const FT *, const FT *, const DevFlux<FT> &);              \
    template int launch_col_gas<FT>(rrtmgp_workspace *, int, int, View2<const FT>, View2<FT>, const rrtmgp_params &,   \
                                    View2<const FT>, const FT *);                                                      \
    template int launch_rel_hum<FT>(rrtmgp_workspace *, int, int, View2<FT>, View2<const FT>, View2<const FT>,         \
                                    const rrtmgp_params &, View2<const FT>);
INST(float)
INST(double)

}  // namespace rrtmgp
