// device.h — device-side building blocks shared by the LW and SW column kernels.
//
// Execution model (gfx950): one workgroup per column, one lane per g-point
// (wavefront w of the group owns g-points 64w..64w+63).  Column data that every
// g-point needs is staged once per column into LDS by `prepare_column` (lane =
// layer): the gas table of volume mixing ratios, the T / ln p interpolation
// indices and fractions (g-point independent), Planck-table positions, cloud and
// aerosol lookup positions.  The g-point lanes then run the vertical sweeps with
// all per-lane state in registers and only 4 values per level in the sweep
// scratch.
#pragma once

#include "common.h"

namespace rrtmgp {

// ---- numerics (src/Numerics.jl:24-63), all of the working precision --------------
template <typename FT> struct Num;
template <> struct Num<float> {
    static __device__ __forceinline__ float eps() { return 1.1920928955078125e-07f; }
    static __device__ __forceinline__ float pi() { return 3.14159265358979323846f; }
};
template <> struct Num<double> {
    static __device__ __forceinline__ double eps() { return 2.220446049250313e-16; }
    static __device__ __forceinline__ double pi() { return 3.14159265358979323846; }
};

__device__ __forceinline__ float m_exp(float x) { return expf(x); }
__device__ __forceinline__ double m_exp(double x) { return exp(x); }
__device__ __forceinline__ float m_expm1(float x) { return expm1f(x); }
__device__ __forceinline__ double m_expm1(double x) { return expm1(x); }
__device__ __forceinline__ float m_log(float x) { return logf(x); }
__device__ __forceinline__ double m_log(double x) { return log(x); }
__device__ __forceinline__ float m_sqrt(float x) { return sqrtf(x); }
__device__ __forceinline__ double m_sqrt(double x) { return sqrt(x); }
__device__ __forceinline__ float m_sin(float x) { return sinf(x); }
__device__ __forceinline__ double m_sin(double x) { return sin(x); }
__device__ __forceinline__ float m_cos(float x) { return cosf(x); }
__device__ __forceinline__ double m_cos(double x) { return cos(x); }
template <typename FT> __device__ __forceinline__ FT m_max(FT a, FT b) { return a > b ? a : b; }
template <typename FT> __device__ __forceinline__ FT m_min(FT a, FT b) { return a < b ? a : b; }
template <typename FT> __device__ __forceinline__ FT m_abs(FT a) { return a < FT(0) ? -a : a; }

template <typename FT> __device__ __forceinline__ FT k_min() { return m_sqrt(Num<FT>::eps()); }
template <typename FT> __device__ __forceinline__ FT tau_thresh() { return m_sqrt(m_sqrt(Num<FT>::eps())); }
template <typename FT> __device__ __forceinline__ FT resonance_window() { return m_sqrt(Num<FT>::eps()); }
template <typename FT> __device__ __forceinline__ FT mu0_min() { return Num<FT>::eps(); }

// ---- McICA counter-based stream (spec: include/rrtmgp_hip.h) -----------------------
__host__ __device__ __forceinline__ uint64_t mix64(uint64_t z) {
    z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ULL;
    z = (z ^ (z >> 27)) * 0x94d049bb133111ebULL;
    return z ^ (z >> 31);
}
__host__ __device__ __forceinline__ uint64_t mcica_key(uint64_t seed, int64_t gcol, int64_t igpt, int is_sw) {
    const uint64_t G = 0x9e3779b97f4a7c15ULL;
    uint64_t k = mix64(seed + G * (uint64_t)gcol);
    return mix64(k ^ ((uint64_t)igpt | ((uint64_t)(is_sw ? 1 : 0) << 32)));
}
__host__ __device__ __forceinline__ double mcica_draw(uint64_t key, int draw) {
    const uint64_t G = 0x9e3779b97f4a7c15ULL;
    uint64_t k = mix64(key + G * ((uint64_t)draw + 1));
    return (double)(k >> 11) * (1.0 / 9007199254740992.0);
}

// ---- wavefront (64-lane) sum, fixed butterfly order => deterministic ----------------
template <typename FT>
__device__ __forceinline__ FT wave_sum(FT v) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
    return v;
}

// ---- LDS carve ----------------------------------------------------------------------
struct ColDims {
    int nlay, nlev, ngas1 /* ngas + 1 */, nwaves;
    int lw, has_cld, has_aero, n_acc /* accumulated flux components per level */;
};

template <typename FT>
struct ColShared {
    FT *vmr;  // [ngas1][nlay]; row 0 is 1 (get_vmr ig == 0, VolumeMixingRatios.jl:97-99)
    FT *col_dry, *p_lay, *t_lay, *rel_hum, *t_lev;
    FT *fT, *fP, *dens_fact, *dry_fact;
    int *jT, *jP, *tropo;
    int *pl_lev_loc, *pl_lay_loc;  // Planck table positions (LW)
    FT *pl_lev_f, *pl_lay_f;
    FT *cld_frac, *path_liq, *path_ice, *liq_fac, *ice_fac;
    int *liq_loc, *ice_loc;
    FT *aero_mass, *aero_size;  // (15, nlay)
    FT *rh_f;
    int *rh_loc;
    unsigned char *aero_bin;  // [10][nlay] size bin (0-based) of the 5 dust + 5 sea-salt species
    unsigned char *aero_mask;
    FT *acc;    // [nwaves][nlev][n_acc]
    int *misc;  // [0..nwaves): cloudy g-point count per wave; [nwaves]: pl_sfc_loc; [nwaves+1]: cld start; [nwaves+2]: cld finish
    FT *miscf;  // [0]: pl_sfc_f; [1..2]: aod ext / sca
};

template <typename T>
__host__ __device__ inline T *carve(char *&p, size_t n) {
    T *r = reinterpret_cast<T *>(p);
    size_t b = (n * sizeof(T) + 15) & ~size_t(15);
    p += b;
    return r;
}

template <typename FT>
__host__ __device__ inline size_t carve_shared(ColShared<FT> &s, char *base, const ColDims &d) {
    char *p = base;
    const int nlay = d.nlay, nlev = d.nlev;
    s.vmr = carve<FT>(p, (size_t)d.ngas1 * nlay);
    s.col_dry = carve<FT>(p, nlay); s.p_lay = carve<FT>(p, nlay); s.t_lay = carve<FT>(p, nlay);
    s.rel_hum = carve<FT>(p, nlay); s.t_lev = carve<FT>(p, nlev);
    s.fT = carve<FT>(p, nlay); s.fP = carve<FT>(p, nlay); s.dens_fact = carve<FT>(p, nlay);
    s.dry_fact = carve<FT>(p, nlay);
    s.jT = carve<int>(p, nlay); s.jP = carve<int>(p, nlay); s.tropo = carve<int>(p, nlay);
    s.pl_lev_loc = carve<int>(p, nlev); s.pl_lay_loc = carve<int>(p, nlay);
    s.pl_lev_f = carve<FT>(p, nlev); s.pl_lay_f = carve<FT>(p, nlay);
    if (d.has_cld) {
        s.cld_frac = carve<FT>(p, nlay); s.path_liq = carve<FT>(p, nlay); s.path_ice = carve<FT>(p, nlay);
        s.liq_fac = carve<FT>(p, nlay); s.ice_fac = carve<FT>(p, nlay);
        s.liq_loc = carve<int>(p, nlay); s.ice_loc = carve<int>(p, nlay);
    }
    if (d.has_aero) {
        s.aero_mass = carve<FT>(p, (size_t)RRTMGP_N_AEROSOLS * nlay);
        s.aero_size = carve<FT>(p, (size_t)RRTMGP_N_AEROSOLS * nlay);
        s.rh_f = carve<FT>(p, nlay); s.rh_loc = carve<int>(p, nlay);
        s.aero_bin = carve<unsigned char>(p, (size_t)10 * nlay);
        s.aero_mask = carve<unsigned char>(p, nlay);
    }
    s.acc = carve<FT>(p, (size_t)d.nwaves * nlev * d.n_acc);
    s.misc = carve<int>(p, d.nwaves + 4);
    s.miscf = carve<FT>(p, 4);
    return (size_t)(p - base);
}

// ---- g-point independent column preparation (lane = layer) --------------------------
// loc_lower for uniform grids, src/optics/optics_utils.jl:7-14 (returns 0-based index)
template <typename FT>
__device__ __forceinline__ int loc_lower_eq0(FT xi, FT dx, int n, const FT *x) {
    if (xi <= x[0]) return 0;
    if (xi >= x[n - 1]) return n - 2;
    int j = (int)((xi - x[0]) / dx);
    return j < n - 2 ? j : n - 2;
}

// interp1d_equispaced position (optics_utils.jl:34-44) as (loc0, factor); the
// out-of-range returns y[1] / y[end] are encoded as (0, 0) and (n-2, 1).
template <typename FT>
__device__ __forceinline__ void planck_pos(FT xi, const FT *x, int n, int &loc, FT &f) {
    if (xi < x[0]) { loc = 0; f = FT(0); return; }
    if (xi > x[n - 1]) { loc = n - 2; f = FT(1); return; }
    FT dx = x[1] - x[0];
    loc = loc_lower_eq0(xi, dx, n, x);
    f = (xi - x[loc]) / dx;
}

// interp1d_loc_factor for non-uniform grids (optics_utils.jl:21-27,51-62), 0-based
template <typename FT>
__device__ __forceinline__ void loc_factor_gen(FT xi, const FT *x, int n, int &loc, FT &f) {
    if (xi < x[0]) { loc = 0; f = FT(0); return; }
    if (xi > x[n - 1]) { loc = n - 2; f = FT(1); return; }
    int l = n - 2;
    if (xi <= x[0]) l = 0;
    else
        for (int i = 0; i < n; i++)
            if (xi < x[i]) { l = i - 1; break; }
    loc = l;
    f = (xi - x[l]) / (x[l + 1] - x[l]);
}

// cloud size-table position, src/optics/cloud_optics.jl:170-182 (0-based loc)
template <typename FT>
__device__ __forceinline__ void cld_pos(FT re, FT r_lwr, FT r_upr, int nsize, int &loc, FT &fac) {
    FT dr = (r_upr - r_lwr) / FT(nsize - 1);
    re = m_max(m_min(re, r_upr), r_lwr);
    int l = (int)((re - r_lwr) / dr) + 1;
    l = l > nsize - 1 ? nsize - 1 : l;
    l = l < 1 ? 1 : l;
    fac = (re - r_lwr - FT(l - 1) * dr) / dr;
    loc = l - 1;
}

template <typename FT>
__device__ inline void prepare_column(const ColShared<FT> &sh, const ColDims &d, const DevGas<FT> &lk,
                                      const DevCld<FT> *cld, const DevAero<FT> *aero, const DevState<FT> &as, int col) {
    const int nlay = d.nlay, nlev = d.nlev, tid = threadIdx.x, nt = blockDim.x;
    const FT *ld = as.layerdata + (size_t)4 * nlay * col;
    for (int k = tid; k < nlay; k += nt) {
        const FT col_dry = ld[4 * k + 0], p = ld[4 * k + 1], t = ld[4 * k + 2];
        sh.col_dry[k] = col_dry; sh.p_lay[k] = p; sh.t_lay[k] = t; sh.rel_hum[k] = ld[4 * k + 3];
        const int tropo = p > lk.p_ref_tropo ? 0 : 1;  // gas_optics.jl:188 (0 = lower)
        sh.tropo[k] = tropo;
        // compute_interp_frac_temp, gas_optics.jl:87-93
        const FT dT = lk.t_ref[1] - lk.t_ref[0];
        const int jT = loc_lower_eq0(t, dT, lk.n_t_ref, lk.t_ref);
        sh.jT[k] = jT;
        sh.fT[k] = (t - lk.t_ref[jT]) / dT;
        // compute_interp_frac_press, gas_optics.jl:100-117; jP = 0-based lower pressure plane
        const FT dlp = lk.ln_p_ref[0] - lk.ln_p_ref[1];
        const FT logp = m_log(p);
        int j = (int)((lk.ln_p_ref[0] - logp) / dlp) + 1;
        const int n_p_ref = lk.n_pp - 1;
        j = j < 1 ? 1 : j;
        j = j > n_p_ref - 1 ? n_p_ref - 1 : j;
        j += 1;                                          // 1-based jpress
        sh.fP[k] = (lk.ln_p_ref[j - 2] - logp) / dlp;
        sh.jP[k] = (j + tropo) - 2;                      // (jpress + tropo1 - 1) - 1 -> 0-based lower plane
        sh.dens_fact[k] = FT(0.01) * p / t;              // gas_optics.jl:368-370
        if (d.lw) {
            planck_pos(t, lk.t_planck, lk.n_t_plnk, sh.pl_lay_loc[k], sh.pl_lay_f[k]);
        }
        if (d.has_cld) {
            const size_t o = (size_t)nlay * col + k;
            sh.cld_frac[k] = as.cld_frac[o];
            sh.path_liq[k] = as.cld_path_liq[o];
            sh.path_ice[k] = as.cld_path_ice[o];
            cld_pos(as.cld_r_eff_liq[o], cld->radliq_lwr, cld->radliq_upr, cld->nsize_liq, sh.liq_loc[k], sh.liq_fac[k]);
            cld_pos(as.cld_r_eff_ice[o], cld->radice_lwr, cld->radice_upr, cld->nsize_ice, sh.ice_loc[k], sh.ice_fac[k]);
        }
        if (d.has_aero) {
            const size_t o = (size_t)RRTMGP_N_AEROSOLS * ((size_t)nlay * col + k);
            unsigned char any = 0;
            for (int ia = 0; ia < RRTMGP_N_AEROSOLS; ia++) {
                const FT m = as.aero_mass[o + ia];
                sh.aero_mass[RRTMGP_N_AEROSOLS * k + ia] = m;
                sh.aero_size[RRTMGP_N_AEROSOLS * k + ia] = as.aero_size[o + ia];
                any |= (m > FT(0));  // compute_aero_mask!, aerosol_optics.jl:464-483
            }
            sh.aero_mask[k] = any;
            loc_factor_gen(sh.rel_hum[k], aero->rh_levels, aero->nrh, sh.rh_loc[k], sh.rh_f[k]);
            // locate_merra_size_bin (aerosol_optics.jl:438-451) for dust 1,8..11 and sea salt 2,12..15
            const int ids[10] = {0, 7, 8, 9, 10, 1, 11, 12, 13, 14};
            for (int s = 0; s < 10; s++) {
                const FT sz = as.aero_size[o + ids[s]];
                int bin = 0;
                for (int ib = 0; ib < aero->nbin; ib++) {
                    if (aero->size_bin_limits[2 * ib] <= sz && sz <= aero->size_bin_limits[2 * ib + 1]) { bin = ib; break; }
                    bin = aero->nbin - 1;
                }
                sh.aero_bin[s * nlay + k] = (unsigned char)bin;
            }
        }
    }
    // gas table: row ig (1-based gas index), row 0 = 1
    const int ngas = d.ngas1 - 1;
    for (int i = tid; i < d.ngas1 * nlay; i += nt) {
        const int ig = i / nlay, k = i - ig * nlay;
        FT v;
        if (ig == 0) v = FT(1);
        else if (as.vmr_kind == RRTMGP_VMR_GM) {
            if (ig == 1) v = as.vmr_h2o[(size_t)nlay * col + k];
            else if (ig == 3) v = as.vmr_o3[(size_t)nlay * col + k];
            else v = ig <= as.ngas ? as.vmr[ig - 1] : FT(0);
        } else {
            v = ig <= as.ngas ? as.vmr[(size_t)(ig - 1) + (size_t)as.ngas * ((size_t)k + (size_t)nlay * col)] : FT(0);
        }
        sh.vmr[i] = v;
    }
    (void)ngas;
    for (int k = tid; k < nlev; k += nt) {
        const FT t = as.t_lev[(size_t)nlev * col + k];
        sh.t_lev[k] = t;
        if (d.lw) planck_pos(t, lk.t_planck, lk.n_t_plnk, sh.pl_lev_loc[k], sh.pl_lev_f[k]);
    }
    if (tid == 0) {
        if (d.lw) planck_pos(as.t_sfc[col], lk.t_planck, lk.n_t_plnk, sh.misc[d.nwaves], sh.miscf[0]);
        for (int w = 0; w < d.nwaves; w++) sh.misc[w] = 0;
    }
    __syncthreads();
    for (int k = tid; k < nlay; k += nt)
        sh.dry_fact[k] = FT(1) / (FT(1) + sh.vmr[lk.idx_h2o * nlay + k]);  // gas_optics.jl:367
    if (d.has_cld && tid == 0) {
        // _get_start / _get_finish, cloud_optics.jl:310-322 (0-based, -1 when clear)
        int start = -1, finish = -1;
        for (int k = 0; k < nlay; k++) if (sh.cld_frac[k] > FT(0)) { start = k; break; }
        for (int k = nlay - 1; k >= 0; k--) if (sh.cld_frac[k] > FT(0)) { finish = k; break; }
        sh.misc[d.nwaves + 1] = start;
        sh.misc[d.nwaves + 2] = finish;
    }
    __syncthreads();
}

// ---- per-lane band constants ---------------------------------------------------------
struct LaneBand {
    int g, ibnd, gi, ngb;
    int ks[2][2];
    int m_st[2], m_n[2], m_koff[2];
};

template <typename FT>
__device__ __forceinline__ LaneBand lane_band(const DevGas<FT> &lk, int g) {
    LaneBand lb;
    lb.g = g;
    lb.ibnd = lk.gpt2bnd[g];
    lb.gi = g - lk.bnd_lo[lb.ibnd];
    lb.ngb = lk.bnd_ng[lb.ibnd];
    for (int tr = 0; tr < 2; tr++) {
        lb.ks[tr][0] = lk.key_species[0 + 2 * (tr + 2 * lb.ibnd)];
        lb.ks[tr][1] = lk.key_species[1 + 2 * (tr + 2 * lb.ibnd)];
        lb.m_st[tr] = lk.m_bnd_st[tr][lb.ibnd];
        lb.m_n[tr] = lk.m_bnd_st[tr][lb.ibnd + 1] - lb.m_st[tr];
        lb.m_koff[tr] = lk.m_koff[tr][lb.ibnd];
    }
    return lb;
}

// ---- gas optics for one (layer, g-point): src/optics/gas_optics.jl:176-320 -------------
template <typename FT, bool SW>
__device__ __forceinline__ void gas_optics(const DevGas<FT> &lk, const ColShared<FT> &sh, const LaneBand &lb, int k,
                                           int nlay, FT &tau, FT &ssa, FT &pfrac) {
    const int tropo = sh.tropo[k];
    const int jT = sh.jT[k], jP = sh.jP[k];
    const FT fT = sh.fT[k], fP = sh.fP[k];
    const FT col_dry = sh.col_dry[k];
    const int ig0 = lb.ks[tropo][0], ig1 = lb.ks[tropo][1];
    const FT vmr1 = sh.vmr[ig0 * nlay + k], vmr2 = sh.vmr[ig1 * nlay + k];
    // compute_interp_frac_eta, gas_optics.jl:129-170
    const int NE = lk.n_eta, NG = lk.n_gpt;
    int je[2];
    FT fe[2], cm[2];
#pragma unroll
    for (int it = 0; it < 2; it++) {
        const FT eta_half = lk.vmr_ref[tropo + 2 * (ig0 + lk.n_gases * (jT + it))] /
                            lk.vmr_ref[tropo + 2 * (ig1 + lk.n_gases * (jT + it))];
        const FT col_mix = vmr1 + eta_half * vmr2;
        FT eta = vmr1 * (FT(1) / col_mix);
        if (col_mix <= FT(0)) eta = FT(0.5);
        const FT loc_eta = eta * FT(NE - 1);
        int j = (int)loc_eta;
        j = j > NE - 2 ? NE - 2 : j;
        je[it] = j;
        fe[it] = loc_eta - FT(j);
        cm[it] = col_mix;
    }
    const FT omfT = FT(1) - fT, omfP = FT(1) - fP, omfe1 = FT(1) - fe[0], omfe2 = FT(1) - fe[1];
    // interp3d, optics_utils.jl:136-181, on the [t][p][eta][gpt] layout
    const size_t sE = (size_t)NG, sP = (size_t)NE * NG, sT = (size_t)lk.n_pp * NE * NG;
    const size_t b1 = (size_t)jT * sT + (size_t)jP * sP + (size_t)je[0] * sE + lb.g;
    const size_t b2 = (size_t)(jT + 1) * sT + (size_t)jP * sP + (size_t)je[1] * sE + lb.g;
    const FT *km = lk.kmajor;
    const FT tau_major =
        (cm[0] * (omfP * (omfT * (omfe1 * km[b1] + fe[0] * km[b1 + sE])) +
                  fP * (omfT * (omfe1 * km[b1 + sP] + fe[0] * km[b1 + sP + sE]))) +
         cm[1] * (omfP * (fT * (omfe2 * km[b2] + fe[1] * km[b2 + sE])) +
                  fP * (fT * (omfe2 * km[b2 + sP] + fe[1] * km[b2 + sP + sE])))) *
        col_dry;
    // compute_tau_minor, gas_optics.jl:344-412
    FT tau_minor = FT(0);
    const int n = lb.m_n[tropo];
    if (n > 0) {
        const FT vmr_h2o = sh.vmr[lk.idx_h2o * nlay + k];
        (void)vmr_h2o;
        const FT dry_fact = sh.dry_fact[k];
        const FT density_fact = sh.dens_fact[k];
        const int *gd = lk.m_gasdata[tropo] + 4 * lb.m_st[tropo];
        const FT *kmn = lk.m_kminor[tropo];
        const size_t NC = (size_t)lk.m_ncontrib[tropo];
        const size_t c0 = (size_t)lb.m_koff[tropo] + lb.gi;
        for (int i = 0; i < n; i++) {
            const int idx_gas = gd[4 * i + 0], idx_sgas = gd[4 * i + 1], w_dens = gd[4 * i + 2], by_comp = gd[4 * i + 3];
            const FT vmr_imnr = sh.vmr[idx_gas * nlay + k];
            if (vmr_imnr > FT(0)) {
                FT scaling = vmr_imnr * col_dry;
                if (w_dens == 1) {
                    scaling *= density_fact;
                    if (idx_sgas > 0) {
                        const FT vs = sh.vmr[idx_sgas * nlay + k];
                        if (by_comp == 1) scaling *= (FT(1) - vs * dry_fact);
                        else scaling *= vs * dry_fact;
                    }
                }
                const size_t c = c0 + (size_t)i * lb.ngb;
                const size_t a1 = ((size_t)jT * NE + je[0]) * NC + c;
                const size_t a2 = ((size_t)(jT + 1) * NE + je[1]) * NC + c;
                // interp2d, optics_utils.jl:85-98
                const FT kv = omfe1 * omfT * kmn[a1] + fe[0] * omfT * kmn[a1 + NC] + omfe2 * fT * kmn[a2] +
                              fe[1] * fT * kmn[a2 + NC];
                tau_minor += kv * scaling;
            }
        }
    }
    if (!SW) {
        const FT *pf = lk.pfrac;
        pfrac = (omfP * (omfT * (omfe1 * pf[b1] + fe[0] * pf[b1 + sE])) +
                 fP * (omfT * (omfe1 * pf[b1 + sP] + fe[0] * pf[b1 + sP + sE]))) +
                (omfP * (fT * (omfe2 * pf[b2] + fe[1] * pf[b2 + sE])) +
                 fP * (fT * (omfe2 * pf[b2 + sP] + fe[1] * pf[b2 + sP + sE])));
        tau = m_max(tau_major + tau_minor, FT(0));
        ssa = FT(0);
    } else {
        // compute_tau_rayleigh, gas_optics.jl:430-444
        const FT *rc = lk.rayl[tropo];
        const size_t r1 = ((size_t)jT * NE + je[0]) * NG + lb.g, r2 = ((size_t)(jT + 1) * NE + je[1]) * NG + lb.g;
        const FT kr = omfe1 * omfT * rc[r1] + fe[0] * omfT * rc[r1 + NG] + omfe2 * fT * rc[r2] + fe[1] * fT * rc[r2 + NG];
        const FT tau_ray = kr * (sh.vmr[lk.idx_h2o * nlay + k] + FT(1)) * col_dry;
        tau = m_max(tau_major + tau_minor + tau_ray, FT(0));
        ssa = tau_ray * (FT(1) / tau);
        if (tau <= FT(0)) ssa = FT(0);
        pfrac = FT(0);
    }
}

// ---- increment / delta-scale: optics_utils.jl:189-223 ------------------------------------
template <typename FT>
__device__ __forceinline__ void increment_2stream(FT &t1, FT &s1, FT &g1, FT t2, FT s2, FT g2) {
    const FT tau = t1 + t2;
    FT ssa = t1 * s1 + t2 * s2;
    const FT ssag = (t1 * s1 * g1 + t2 * s2 * g2) / m_max(Num<FT>::eps(), ssa);
    ssa /= m_max(Num<FT>::eps(), tau);
    t1 = tau; s1 = ssa; g1 = ssag;
}

template <typename FT>
__device__ __forceinline__ void delta_scale(FT &tau, FT &ssa, FT &g) {
    const FT ssa_one_minus_g2 = ssa * (FT(1) - g) * (FT(1) + g);
    const FT one_minus_wf = (FT(1) - ssa) + ssa_one_minus_g2;
    const FT tau_s = one_minus_wf * tau;
    const FT ssa_s = ssa_one_minus_g2 / m_max(Num<FT>::eps(), one_minus_wf);
    const FT g_s = g / m_max(Num<FT>::eps(), FT(1) + g);
    tau = tau_s; ssa = ssa_s; g = g_s;
}

// ---- cloud optics for one masked (layer, band): cloud_optics.jl:70-244 ---------------------
// returns (tau, tau*ssa, tau*ssa*g) of liquid + ice before combination
template <typename FT>
__device__ __forceinline__ void cloud_props(const DevCld<FT> &lc, const ColShared<FT> &sh, int ibnd, int ice_rgh, int k,
                                            FT &tl, FT &tls, FT &tlsg, FT &ti, FT &tis, FT &tisg) {
    tl = tls = tlsg = ti = tis = tisg = FT(0);
    const FT pl = sh.path_liq[k], pi = sh.path_ice[k];
    if (pl > Num<FT>::eps()) {
        const int nl = lc.nsize_liq;
        const FT *t = lc.liqdata + (size_t)(3 * nl) * ibnd;
        const int loc = sh.liq_loc[k];
        const FT fac = sh.liq_fac[k], fc1 = FT(1) - fac;
        tl = m_max((fc1 * t[loc] + fac * t[loc + 1]) * pl, FT(0));
        tls = (fc1 * t[nl + loc] + fac * t[nl + loc + 1]) * tl;
        tlsg = (fc1 * t[2 * nl + loc] + fac * t[2 * nl + loc + 1]) * tls;
    }
    if (pi > Num<FT>::eps()) {
        const int ni = lc.nsize_ice;
        const FT *t = lc.icedata + (size_t)(3 * ni) * ((size_t)ibnd + (size_t)lc.nband * (ice_rgh - 1));
        const int loc = sh.ice_loc[k];
        const FT fac = sh.ice_fac[k], fc1 = FT(1) - fac;
        ti = m_max((fc1 * t[loc] + fac * t[loc + 1]) * pi, FT(0));
        tis = (fc1 * t[ni + loc] + fac * t[ni + loc + 1]) * ti;
        tisg = (fc1 * t[2 * ni + loc] + fac * t[2 * ni + loc + 1]) * tis;
    }
}

template <typename FT>
__device__ __forceinline__ void add_cloud_2stream(const DevCld<FT> &lc, const ColShared<FT> &sh, int ibnd, int ice_rgh,
                                                  int k, bool delta, FT &tau, FT &ssa, FT &g) {
    FT tl, tls, tlsg, ti, tis, tisg;
    cloud_props(lc, sh, ibnd, ice_rgh, k, tl, tls, tlsg, ti, tis, tisg);
    FT tau_cl = tl + ti;
    FT ssa_cl = tls + tis;
    FT g_cl = (tlsg + tisg) / m_max(Num<FT>::eps(), ssa_cl);
    ssa_cl /= m_max(Num<FT>::eps(), tau_cl);
    if (delta) delta_scale(tau_cl, ssa_cl, g_cl);
    increment_2stream(tau, ssa, g, tau_cl, ssa_cl, g_cl);
}

template <typename FT>
__device__ __forceinline__ void add_cloud_1scalar(const DevCld<FT> &lc, const ColShared<FT> &sh, int ibnd, int ice_rgh,
                                                  int k, FT &tau) {
    FT tl, tls, tlsg, ti, tis, tisg;
    cloud_props(lc, sh, ibnd, ice_rgh, k, tl, tls, tlsg, ti, tis, tisg);
    tau += (tl - tls) + (ti - tis);  // cloud_optics.jl:45
}

// ---- McICA mask for this lane's g-point: cloud_optics.jl:264-334 ----------------------------
// Bits of (m0, m1) are layers 0..63 / 64..127.  Returns any(mask).
template <typename FT>
__device__ inline bool build_cloud_mask(const ColShared<FT> &sh, const ColDims &d, uint64_t key, uint64_t &m0,
                                        uint64_t &m1) {
    m0 = m1 = 0;
    const int start = sh.misc[d.nwaves + 1], finish = sh.misc[d.nwaves + 2];
    if (start < 0) return false;
    int draw = 0;
    FT cf_above = sh.cld_frac[finish];
    double r_above = mcica_draw(key, draw++);
    bool mask_above = r_above >= (double)(FT(1) - cf_above);
    auto setbit = [&](int k) { if (k < 64) m0 |= (1ULL << k); else m1 |= (1ULL << (k - 64)); };
    if (mask_above) setbit(finish);
    for (int k = finish - 1; k >= start; k--) {
        const FT cf = sh.cld_frac[k];
        bool mk;
        if (cf > FT(0)) {
            const double r = mask_above ? r_above : mcica_draw(key, draw++) * (double)(FT(1) - cf_above);
            mk = r >= (double)(FT(1) - cf);
            r_above = r;
        } else {
            mk = false;
        }
        if (mk) setbit(k);
        cf_above = cf;
        mask_above = mk;
    }
    return (m0 | m1) != 0;
}
__device__ __forceinline__ bool mask_bit(uint64_t m0, uint64_t m1, int k) {
    return k < 64 ? ((m0 >> k) & 1ULL) : ((m1 >> (k - 64)) & 1ULL);
}

// ---- aerosol optics for one masked (layer, band): aerosol_optics.jl:141-431 -----------------
template <typename FT>
__device__ inline void lookup_aerosol(const DevAero<FT> &la, const ColShared<FT> &sh, int ibnd, int k, int nlay, FT &tc,
                                      FT &tsc, FT &tsgc) {
    const int NA = RRTMGP_N_AEROSOLS;
    const FT *mass = sh.aero_mass + NA * k;
    const int nrh = la.nrh, nbin = la.nbin;
    const int loc = sh.rh_loc[k];
    const FT f = sh.rh_f[k], omf = FT(1) - f;
    FT t_cum = FT(0), ts_cum = FT(0), tsg_cum = FT(0);
    const int dust_ids[5] = {0, 7, 8, 9, 10}, salt_ids[5] = {1, 11, 12, 13, 14};
    for (int s = 0; s < 5; s++) {
        const FT m = mass[dust_ids[s]];
        if (m > FT(0)) {
            const int bin = sh.aero_bin[s * nlay + k];
            const FT *tb = la.dust + 3 * ((size_t)bin + (size_t)nbin * ibnd);
            const FT t = m * tb[0], ts = t * tb[1], tsg = ts * tb[2];
            t_cum += t; ts_cum += ts; tsg_cum += tsg;
        }
    }
    for (int s = 0; s < 5; s++) {
        const FT m = mass[salt_ids[s]];
        if (m > FT(0)) {
            const int bin = sh.aero_bin[(5 + s) * nlay + k];
            const FT *tb = la.sea_salt + 3 * ((size_t)loc + (size_t)nrh * ((size_t)bin + (size_t)nbin * ibnd));
            const FT t = m * (tb[0] * omf + tb[3] * f);
            const FT ts = t * (tb[1] * omf + tb[4] * f);
            const FT tsg = ts * (tb[2] * omf + tb[5] * f);
            t_cum += t; ts_cum += ts; tsg_cum += tsg;
        }
    }
    auto rh_species = [&](const FT *tab, FT m) {
        const FT *tb = tab + 3 * ((size_t)loc + (size_t)nrh * ibnd);
        const FT t = m * (tb[0] * omf + tb[3] * f);
        const FT ts = t * (tb[1] * omf + tb[4] * f);
        const FT tsg = ts * (tb[2] * omf + tb[5] * f);
        t_cum += t; ts_cum += ts; tsg_cum += tsg;
    };
    auto dry_species = [&](const FT *tab, FT m) {
        const FT *tb = tab + 3 * (size_t)ibnd;
        const FT t = m * tb[0], ts = t * tb[1], tsg = ts * tb[2];
        t_cum += t; ts_cum += ts; tsg_cum += tsg;
    };
    if (mass[2] > FT(0)) rh_species(la.sulfate, mass[2]);
    if (mass[3] > FT(0)) rh_species(la.black_carbon_rh, mass[3]);
    if (mass[4] > FT(0)) dry_species(la.black_carbon, mass[4]);
    if (mass[5] > FT(0)) rh_species(la.organic_carbon_rh, mass[5]);
    if (mass[6] > FT(0)) dry_species(la.organic_carbon, mass[6]);
    tc = t_cum; tsc = ts_cum; tsgc = tsg_cum;
}

// add_aerosol_optics_2stream! body for one layer, aerosol_optics.jl:104-130
template <typename FT>
__device__ __forceinline__ void add_aerosol_2stream(const DevAero<FT> &la, const ColShared<FT> &sh, int ibnd, int k,
                                                    int nlay, bool delta, FT &tau, FT &ssa, FT &g, FT &aod_ext,
                                                    FT &aod_sca) {
    FT ta, tsa, tsga;
    lookup_aerosol(la, sh, ibnd, k, nlay, ta, tsa, tsga);
    FT g_aero = tsga / m_max(Num<FT>::eps(), tsa);
    FT ssa_aero = tsa / m_max(Num<FT>::eps(), ta);
    aod_ext += ta;
    aod_sca += tsa;
    if (delta) delta_scale(ta, ssa_aero, g_aero);
    increment_2stream(tau, ssa, g, ta, ssa_aero, g_aero);
}

// ---- sweep scratch: 4 values per (level, lane), lane-contiguous ----------------------------
template <typename FT>
struct Sweep {
    FT *base;  // this workgroup's slab
    int nt;    // lanes in the workgroup
    __device__ __forceinline__ FT &at(int lev, int a) const { return base[((size_t)lev * 4 + a) * nt + threadIdx.x]; }
};

// ---- write one column's broadband fluxes ---------------------------------------------------
// compute_net_flux! (Fluxes.jl:225-237) then apply_metric_scaling! (:295-304): net = up - dn first,
// then up, dn, net (and dir) are each multiplied by the (nlev, ncol) factor.
template <typename FT>
__device__ inline void store_column(const DevFlux<FT> &fl, const ColShared<FT> &sh, const ColDims &d, int col, int ncol,
                                    bool zero) {
    const int nlev = d.nlev;
    for (int lev = threadIdx.x; lev < nlev; lev += blockDim.x) {
        FT c[3] = {FT(0), FT(0), FT(0)};
        if (!zero) {
            for (int w = 0; w < d.nwaves; w++)
                for (int a = 0; a < d.n_acc; a++) c[a] += sh.acc[((size_t)w * nlev + lev) * d.n_acc + a];
        }
        FT up = c[0], dn = c[1], dir = c[2];
        FT net = up - dn;
        if (fl.metric) {
            const FT m = fl.metric[(size_t)nlev * col + lev];
            up *= m; dn *= m; net *= m; dir *= m;
        }
        const size_t o = fl.layout == RRTMGP_LAYOUT_NCOL_NLEV ? (size_t)col + (size_t)ncol * lev
                                                               : (size_t)lev + (size_t)nlev * col;
        fl.up[o] = up; fl.dn[o] = dn; fl.net[o] = net;
        if (d.n_acc == 3 && fl.dir) fl.dir[o] = dir;
    }
}

}  // namespace rrtmgp
