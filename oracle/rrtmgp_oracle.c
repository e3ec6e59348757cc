/*
 * rrtmgp_oracle.c — TEST INFRASTRUCTURE ONLY (see rrtmgp_oracle.h).
 *
 * Instantiates rrtmgp_oracle_impl.inc for Float32 and Float64 and exports the
 * dispatching entry points.  Build: oracle/Makefile (gcc -O2 -ffp-contract=off
 * -fno-fast-math -fopenmp).
 */
#include "rrtmgp_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

/* ---- McICA counter-based stream (spec in include/rrtmgp_hip.h) -------------
 * splitmix64 finalizer chained over (seed, column, g-point | band-set, draw). */
static inline uint64_t mix64(uint64_t z) {
    z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ULL;
    z = (z ^ (z >> 27)) * 0x94d049bb133111ebULL;
    return z ^ (z >> 31);
}

double rrtmgp_oracle_mcica_uniform(uint64_t seed, int64_t gcol, int64_t igpt, int32_t is_sw, int32_t draw) {
    const uint64_t G = 0x9e3779b97f4a7c15ULL;
    uint64_t k = mix64(seed + G * (uint64_t)gcol);
    k = mix64(k ^ ((uint64_t)igpt | ((uint64_t)(is_sw ? 1 : 0) << 32)));
    k = mix64(k + G * ((uint64_t)draw + 1));
    return (double)(k >> 11) * (1.0 / 9007199254740992.0);
}

/* AngularDiscretizations.jl:41-56: FT(1) ./ [Float64 literals] divides in Float64, then converts */
void rrtmgp_oracle_angular_discretization(int n, size_t ft_size, void *Ds, void *wts) {
    static const double mu[4][4] = {{0.6096748751, 0, 0, 0},
                                    {0.2509907356, 0.7908473988, 0, 0},
                                    {0.1024922169, 0.4417960320, 0.8633751621, 0},
                                    {0.0454586727, 0.2322334416, 0.5740198775, 0.9030775973}};
    static const double w[4][4] = {{1, 0, 0, 0},
                                   {0.2300253764, 0.7699746236, 0, 0},
                                   {0.0437820218, 0.3875796738, 0.5686383044, 0},
                                   {0.0092068785, 0.1285704278, 0.4323381850, 0.4298845087}};
    for (int i = 0; i < n; i++) {
        double d = 1.0 / mu[n - 1][i];
        if (ft_size == 4) { ((float *)Ds)[i] = (float)d; ((float *)wts)[i] = (float)w[n - 1][i]; }
        else              { ((double *)Ds)[i] = d;       ((double *)wts)[i] = w[n - 1][i]; }
    }
}

#define CAT_(a, b) a##b
#define CAT(a, b) CAT_(a, b)
#define FN(name) CAT(name, SFX)

/* Float32 instance */
#define FT float
#define SFX _f32
#define EPS 1.1920928955078125e-07f
#define PI_FT 3.14159265358979323846f
#define SQRT sqrtf
#define EXP expf
#define EXPM1 expm1f
#define LOG logf
#define SIN sinf
#define COS cosf
#define POW powf
#define FABS fabsf
#define FMAX(a, b) ((a) > (b) ? (a) : (b))
#define FMIN(a, b) ((a) < (b) ? (a) : (b))
#include "rrtmgp_oracle_impl.inc"
#undef FT
#undef SFX
#undef EPS
#undef PI_FT
#undef SQRT
#undef EXP
#undef EXPM1
#undef LOG
#undef SIN
#undef COS
#undef POW
#undef FABS

/* Float64 instance */
#define FT double
#define SFX _f64
#define EPS 2.220446049250313e-16
#define PI_FT 3.14159265358979323846
#define SQRT sqrt
#define EXP exp
#define EXPM1 expm1
#define LOG log
#define SIN sin
#define COS cos
#define POW pow
#define FABS fabs
#include "rrtmgp_oracle_impl.inc"
#undef FT
#undef SFX

/* ---- dispatching exports --------------------------------------------------- */
#define DISPATCH(ftype, call32, call64) ((ftype) == RRTMGP_F32 ? (call32) : (ftype) == RRTMGP_F64 ? (call64) : RRTMGP_EINVAL)

int rrtmgp_oracle_rte_lw_2stream_solve(const rrtmgp_gas_lookup_desc *lw, const rrtmgp_cloud_lookup_desc *cld,
                                       const rrtmgp_aerosol_lookup_desc *aero, const rrtmgp_atmos_state *as,
                                       const rrtmgp_lw_bcs *bcs, const rrtmgp_flux_out *flux,
                                       const rrtmgp_solve_opts *opts) {
    return DISPATCH(lw->ftype, solve_lw_f32(1, lw, cld, aero, as, bcs, flux, opts),
                    solve_lw_f64(1, lw, cld, aero, as, bcs, flux, opts));
}
int rrtmgp_oracle_rte_lw_noscat_solve(const rrtmgp_gas_lookup_desc *lw, const rrtmgp_cloud_lookup_desc *cld,
                                      const rrtmgp_aerosol_lookup_desc *aero, const rrtmgp_atmos_state *as,
                                      const rrtmgp_lw_bcs *bcs, const rrtmgp_flux_out *flux,
                                      const rrtmgp_solve_opts *opts) {
    return DISPATCH(lw->ftype, solve_lw_f32(0, lw, cld, aero, as, bcs, flux, opts),
                    solve_lw_f64(0, lw, cld, aero, as, bcs, flux, opts));
}
int rrtmgp_oracle_rte_sw_2stream_solve(const rrtmgp_gas_lookup_desc *sw, const rrtmgp_cloud_lookup_desc *cld,
                                       const rrtmgp_aerosol_lookup_desc *aero, const rrtmgp_atmos_state *as,
                                       const rrtmgp_sw_bcs *bcs, const rrtmgp_flux_out *flux,
                                       const rrtmgp_solve_opts *opts) {
    return DISPATCH(sw->ftype, solve_sw_f32(1, sw, cld, aero, as, bcs, flux, opts),
                    solve_sw_f64(1, sw, cld, aero, as, bcs, flux, opts));
}
int rrtmgp_oracle_rte_sw_noscat_solve(const rrtmgp_gas_lookup_desc *sw, const rrtmgp_atmos_state *as,
                                      const rrtmgp_sw_bcs *bcs, const rrtmgp_flux_out *flux,
                                      const rrtmgp_solve_opts *opts) {
    return DISPATCH(sw->ftype, solve_sw_f32(0, sw, NULL, NULL, as, bcs, flux, opts),
                    solve_sw_f64(0, sw, NULL, NULL, as, bcs, flux, opts));
}

int rrtmgp_oracle_rte_lw_2stream_solve_gray(const rrtmgp_gray_state *as, const rrtmgp_lw_bcs *bcs,
                                            const rrtmgp_flux_out *flux, const rrtmgp_solve_opts *opts, int32_t ftype) {
    return DISPATCH(ftype, solve_lw_gray_f32(1, as, bcs, flux, opts), solve_lw_gray_f64(1, as, bcs, flux, opts));
}
int rrtmgp_oracle_rte_lw_noscat_solve_gray(const rrtmgp_gray_state *as, const rrtmgp_lw_bcs *bcs,
                                           const rrtmgp_flux_out *flux, const rrtmgp_solve_opts *opts, int32_t ftype) {
    return DISPATCH(ftype, solve_lw_gray_f32(0, as, bcs, flux, opts), solve_lw_gray_f64(0, as, bcs, flux, opts));
}
int rrtmgp_oracle_rte_sw_2stream_solve_gray(const rrtmgp_gray_state *as, const rrtmgp_sw_bcs *bcs,
                                            const rrtmgp_flux_out *flux, const rrtmgp_solve_opts *opts, int32_t ftype) {
    return DISPATCH(ftype, solve_sw_gray_f32(1, as, bcs, flux, opts), solve_sw_gray_f64(1, as, bcs, flux, opts));
}
int rrtmgp_oracle_rte_sw_noscat_solve_gray(const rrtmgp_gray_state *as, const rrtmgp_sw_bcs *bcs,
                                           const rrtmgp_flux_out *flux, const rrtmgp_solve_opts *opts, int32_t ftype) {
    return DISPATCH(ftype, solve_sw_gray_f32(0, as, bcs, flux, opts), solve_sw_gray_f64(0, as, bcs, flux, opts));
}

int rrtmgp_oracle_prepare_atmosphere(int32_t ftype, const rrtmgp_atmos_state *as, const rrtmgp_params *params,
                                     const rrtmgp_prepare_opts *opts) {
    return DISPATCH(ftype, prepare_atmosphere_as_f32(as, params, opts), prepare_atmosphere_as_f64(as, params, opts));
}
int rrtmgp_oracle_prepare_atmosphere_gray(int32_t ftype, const rrtmgp_gray_state *as, const rrtmgp_params *params,
                                          const rrtmgp_prepare_opts *opts) {
    return DISPATCH(ftype, prepare_atmosphere_gray_f32(as, params, opts),
                    prepare_atmosphere_gray_f64(as, params, opts));
}

int rrtmgp_oracle_compute_col_gas(int32_t ftype, int64_t ncol, int64_t nlay, const void *p_lev, void *col_dry,
                                  const rrtmgp_params *params, const void *vmr_h2o, const void *lat) {
    if (ftype == RRTMGP_F32)
        compute_col_gas_f32(ncol, nlay, (const float *)p_lev, (float *)col_dry, params, (const float *)vmr_h2o,
                            (const float *)lat);
    else if (ftype == RRTMGP_F64)
        compute_col_gas_f64(ncol, nlay, (const double *)p_lev, (double *)col_dry, params, (const double *)vmr_h2o,
                            (const double *)lat);
    else
        return RRTMGP_EINVAL;
    return RRTMGP_OK;
}

int rrtmgp_oracle_compute_relative_humidity(int32_t ftype, int64_t ncol, int64_t nlay, void *rh, const void *p_lay,
                                            const void *t_lay, const rrtmgp_params *params, const void *vmr_h2o) {
    if (ftype == RRTMGP_F32)
        compute_relative_humidity_f32(ncol, nlay, (float *)rh, (const float *)p_lay, (const float *)t_lay, params,
                                      (const float *)vmr_h2o);
    else if (ftype == RRTMGP_F64)
        compute_relative_humidity_f64(ncol, nlay, (double *)rh, (const double *)p_lay, (const double *)t_lay, params,
                                      (const double *)vmr_h2o);
    else
        return RRTMGP_EINVAL;
    return RRTMGP_OK;
}

int rrtmgp_oracle_gray_heating_rate(int32_t ftype, int64_t ncol, int64_t nlay, void *hr_lay, const void *flux_net,
                                    const void *p_lev, double grav, double cp_d) {
    if (ftype == RRTMGP_F32)
        gray_heating_rate_f32(ncol, nlay, (float *)hr_lay, (const float *)flux_net, (const float *)p_lev, (float)grav,
                              (float)cp_d);
    else if (ftype == RRTMGP_F64)
        gray_heating_rate_f64(ncol, nlay, (double *)hr_lay, (const double *)flux_net, (const double *)p_lev, grav, cp_d);
    else
        return RRTMGP_EINVAL;
    return RRTMGP_OK;
}

int rrtmgp_oracle_update_profile_lw(int32_t ftype, int64_t ncol, int64_t nlay, double sbc, void *t_lay, void *t_lev,
                                    const void *hr_lay, const void *flux_dn, const void *flux_net, void *flux_grad,
                                    void *T_ex_lev, double dt) {
    if (ftype == RRTMGP_F32)
        update_profile_lw_f32(ncol, nlay, (float)sbc, (float *)t_lay, (float *)t_lev, (const float *)hr_lay,
                              (const float *)flux_dn, (const float *)flux_net, (float *)flux_grad, (float *)T_ex_lev,
                              (float)dt);
    else if (ftype == RRTMGP_F64)
        update_profile_lw_f64(ncol, nlay, sbc, (double *)t_lay, (double *)t_lev, (const double *)hr_lay,
                              (const double *)flux_dn, (const double *)flux_net, (double *)flux_grad,
                              (double *)T_ex_lev, dt);
    else
        return RRTMGP_EINVAL;
    return RRTMGP_OK;
}

int rrtmgp_oracle_setup_gray_as_pr_grid(int32_t ftype, int64_t ncol, int64_t nlay, const void *lat, double p0,
                                        double pe, double te, double tt, double dt, double alpha, double r_d,
                                        double grav, void *p_lev, void *p_lay, void *t_lev, void *t_lay, void *z_lev,
                                        void *t_sfc) {
    if (ftype == RRTMGP_F32)
        setup_gray_as_pr_grid_f32(ncol, nlay, (const float *)lat, (float)p0, (float)pe, (float)te, (float)tt, (float)dt,
                                  (float)alpha, (float)r_d, (float)grav, (float *)p_lev, (float *)p_lay,
                                  (float *)t_lev, (float *)t_lay, (float *)z_lev, (float *)t_sfc);
    else if (ftype == RRTMGP_F64)
        setup_gray_as_pr_grid_f64(ncol, nlay, (const double *)lat, p0, pe, te, tt, dt, alpha, r_d, grav,
                                  (double *)p_lev, (double *)p_lay, (double *)t_lev, (double *)t_lay, (double *)z_lev,
                                  (double *)t_sfc);
    else
        return RRTMGP_EINVAL;
    return RRTMGP_OK;
}

int64_t rrtmgp_oracle_loc_lower_eq(double xi, double dx, int64_t n, const double *x) {
    return loc_lower_eq_f64(xi, dx, n, x);
}
int64_t rrtmgp_oracle_loc_lower(double xi, const double *x, int64_t n) { return loc_lower_gen_f64(xi, x, n); }
double rrtmgp_oracle_interp1d_equispaced(double xi, const double *x, const double *y, int64_t n) {
    return interp1d_equispaced_f64(xi, x, y, n);
}
void rrtmgp_oracle_interp1d_loc_factor(double xi, const double *x, int64_t n, int64_t *loc, double *factor) {
    interp1d_loc_factor_f64(xi, x, n, loc, factor);
}

void rrtmgp_oracle_rte_lw_noscat_one_angle(int32_t ftype, int64_t nlay, const void *tau, const void *lay_source,
                                           const void *lev_source, double sfc_source, double sfc_emis, int has_inc,
                                           double inc_flux, double Ds, double w_mu, void *flux_up, void *flux_dn) {
    if (ftype == RRTMGP_F32) {
        col_scratch_f32 s;
        memset(&s, 0, sizeof s);
        s.tau = (float *)tau; s.lay_source = (float *)lay_source; s.lev_source = (float *)lev_source;
        s.sfc_source = (float)sfc_source; s.fb_up = (float *)flux_up; s.fb_dn = (float *)flux_dn;
        rte_lw_noscat_one_angle_f32(&s, has_inc, (float)inc_flux, (float)sfc_emis, (float)Ds, (float)w_mu, nlay);
    } else {
        col_scratch_f64 s;
        memset(&s, 0, sizeof s);
        s.tau = (double *)tau; s.lay_source = (double *)lay_source; s.lev_source = (double *)lev_source;
        s.sfc_source = sfc_source; s.fb_up = (double *)flux_up; s.fb_dn = (double *)flux_dn;
        rte_lw_noscat_one_angle_f64(&s, has_inc, inc_flux, sfc_emis, Ds, w_mu, nlay);
    }
}

void rrtmgp_oracle_lw_2stream_coeffs(int32_t ftype, double tau, double ssa, double g, double lev_src_bot,
                                     double lev_src_top, double out[4]) {
    if (ftype == RRTMGP_F32) {
        float a, b, c, d;
        lw_2stream_coeffs_f32((float)tau, (float)ssa, (float)g, (float)lev_src_bot, (float)lev_src_top, &a, &b, &c, &d);
        out[0] = a; out[1] = b; out[2] = c; out[3] = d;
    } else {
        lw_2stream_coeffs_f64(tau, ssa, g, lev_src_bot, lev_src_top, &out[0], &out[1], &out[2], &out[3]);
    }
}

void rrtmgp_oracle_sw_2stream_coeffs(int32_t ftype, double tau, double ssa, double g, double mu0, double out[5]) {
    if (ftype == RRTMGP_F32) {
        float a, b, c, d, e;
        sw_2stream_coeffs_f32((float)tau, (float)ssa, (float)g, (float)mu0, &a, &b, &c, &d, &e);
        out[0] = a; out[1] = b; out[2] = c; out[3] = d; out[4] = e;
    } else {
        sw_2stream_coeffs_f64(tau, ssa, g, mu0, &out[0], &out[1], &out[2], &out[3], &out[4]);
    }
}

int rrtmgp_oracle_build_cloud_mask(int32_t ftype, uint8_t *mask, const void *cld_frac, int64_t nlay, uint64_t seed,
                                   int64_t gcol, int64_t igpt, int32_t is_sw) {
    if (ftype == RRTMGP_F32) return build_cloud_mask_f32(mask, (const float *)cld_frac, nlay, seed, gcol, igpt, is_sw);
    return build_cloud_mask_f64(mask, (const double *)cld_frac, nlay, seed, gcol, igpt, is_sw);
}
