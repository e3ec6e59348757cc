/*
 * rrtmgp_oracle.h — TEST INFRASTRUCTURE ONLY.
 *
 * CPU restatement (plain C) of RRTMGP.jl's per-column hot path, used as the
 * parity oracle for libhip_rrtmgp.so.  Only tests/, __graft_entry__.smoke() and
 * bench.py's `cpu_baseline` leg may load this library; the product path never
 * does.
 *
 * PARITY STATUS: pinned against every data-free known answer of the reference's
 * own test suite (see tests/test_oracle_*.py) and
 * cross-checked against an independent numpy restatement (oracle/np_oracle.py).
 * Real-data parity (rrtmgp-data v1.9 RFMIP / all-sky reference fluxes) is
 * UNPINNED: neither Julia nor the data artifact exists in the build
 * environment.
 *
 * The entry points take the same descriptor structs as the product C ABI
 * (include/rrtmgp_hip.h) with host pointers, so one Python harness drives both.
 */
#ifndef RRTMGP_ORACLE_H
#define RRTMGP_ORACLE_H

#include "../include/rrtmgp_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* spectral drivers: src/rte/longwave_2stream.jl:73, longwave_noscat.jl:98,
 * shortwave_2stream.jl:105, shortwave_noscat.jl:60 (+ RTESolver.jl metric scaling) */
int rrtmgp_oracle_rte_lw_2stream_solve(const rrtmgp_gas_lookup_desc *lw, const rrtmgp_cloud_lookup_desc *cld,
                                       const rrtmgp_aerosol_lookup_desc *aero, const rrtmgp_atmos_state *as,
                                       const rrtmgp_lw_bcs *bcs, const rrtmgp_flux_out *flux,
                                       const rrtmgp_solve_opts *opts);
int rrtmgp_oracle_rte_lw_noscat_solve(const rrtmgp_gas_lookup_desc *lw, const rrtmgp_cloud_lookup_desc *cld,
                                      const rrtmgp_aerosol_lookup_desc *aero, const rrtmgp_atmos_state *as,
                                      const rrtmgp_lw_bcs *bcs, const rrtmgp_flux_out *flux,
                                      const rrtmgp_solve_opts *opts);
int rrtmgp_oracle_rte_sw_2stream_solve(const rrtmgp_gas_lookup_desc *sw, const rrtmgp_cloud_lookup_desc *cld,
                                       const rrtmgp_aerosol_lookup_desc *aero, const rrtmgp_atmos_state *as,
                                       const rrtmgp_sw_bcs *bcs, const rrtmgp_flux_out *flux,
                                       const rrtmgp_solve_opts *opts);
int rrtmgp_oracle_rte_sw_noscat_solve(const rrtmgp_gas_lookup_desc *sw, const rrtmgp_atmos_state *as,
                                      const rrtmgp_sw_bcs *bcs, const rrtmgp_flux_out *flux,
                                      const rrtmgp_solve_opts *opts);

/* gray drivers */
int rrtmgp_oracle_rte_lw_2stream_solve_gray(const rrtmgp_gray_state *as, const rrtmgp_lw_bcs *bcs,
                                            const rrtmgp_flux_out *flux, const rrtmgp_solve_opts *opts, int32_t ftype);
int rrtmgp_oracle_rte_lw_noscat_solve_gray(const rrtmgp_gray_state *as, const rrtmgp_lw_bcs *bcs,
                                           const rrtmgp_flux_out *flux, const rrtmgp_solve_opts *opts, int32_t ftype);
int rrtmgp_oracle_rte_sw_2stream_solve_gray(const rrtmgp_gray_state *as, const rrtmgp_sw_bcs *bcs,
                                            const rrtmgp_flux_out *flux, const rrtmgp_solve_opts *opts, int32_t ftype);
int rrtmgp_oracle_rte_sw_noscat_solve_gray(const rrtmgp_gray_state *as, const rrtmgp_sw_bcs *bcs,
                                           const rrtmgp_flux_out *flux, const rrtmgp_solve_opts *opts, int32_t ftype);

/* prepare_atmosphere! cascade, src/api/grid_adaptation.jl:73-292 + interpolation.jl:148-252; in place.
 * `ftype` is needed because the state structs do not carry it. */
int rrtmgp_oracle_prepare_atmosphere(int32_t ftype, const rrtmgp_atmos_state *as, const rrtmgp_params *params,
                                     const rrtmgp_prepare_opts *opts);
int rrtmgp_oracle_prepare_atmosphere_gray(int32_t ftype, const rrtmgp_gray_state *as, const rrtmgp_params *params,
                                          const rrtmgp_prepare_opts *opts);

/* prep: src/optics/column_amounts.jl */
int rrtmgp_oracle_compute_col_gas(int32_t ftype, int64_t ncol, int64_t nlay, const void *p_lev, void *col_dry,
                                  const rrtmgp_params *params, const void *vmr_h2o, const void *lat);
int rrtmgp_oracle_compute_relative_humidity(int32_t ftype, int64_t ncol, int64_t nlay, void *rh, const void *p_lay,
                                            const void *t_lay, const rrtmgp_params *params, const void *vmr_h2o);

/* gray test helpers: src/optics/GrayAtmosphere.jl, gray_atmospheric_states.jl:243 */
int rrtmgp_oracle_gray_heating_rate(int32_t ftype, int64_t ncol, int64_t nlay, void *hr_lay, const void *flux_net,
                                    const void *p_lev, double grav, double cp_d);
int rrtmgp_oracle_update_profile_lw(int32_t ftype, int64_t ncol, int64_t nlay, double sbc, void *t_lay, void *t_lev,
                                    const void *hr_lay, const void *flux_dn, const void *flux_net, void *flux_grad,
                                    void *T_ex_lev, double dt);
int rrtmgp_oracle_setup_gray_as_pr_grid(int32_t ftype, int64_t ncol, int64_t nlay, const void *lat, double p0,
                                        double pe, double te, double tt, double dt, double alpha, double r_d,
                                        double grav, void *p_lev, void *p_lay, void *t_lev, void *t_lay, void *z_lev,
                                        void *t_sfc);

/* unit-level exports for the reference's known-answer tests (Float64) */
int64_t rrtmgp_oracle_loc_lower_eq(double xi, double dx, int64_t n, const double *x);
int64_t rrtmgp_oracle_loc_lower(double xi, const double *x, int64_t n);
double rrtmgp_oracle_interp1d_equispaced(double xi, const double *x, const double *y, int64_t n);
void rrtmgp_oracle_interp1d_loc_factor(double xi, const double *x, int64_t n, int64_t *loc, double *factor);
/* one transport angle on caller-supplied tau / sources (test/angular_discretization.jl:102-153);
 * arrays: tau, lay_source (nlay); lev_source (nlev); outputs flux_up, flux_dn (nlev) */
void rrtmgp_oracle_rte_lw_noscat_one_angle(int32_t ftype, int64_t nlay, const void *tau, const void *lay_source,
                                           const void *lev_source, double sfc_source, double sfc_emis, int has_inc,
                                           double inc_flux, double Ds, double w_mu, void *flux_up, void *flux_dn);
void rrtmgp_oracle_lw_2stream_coeffs(int32_t ftype, double tau, double ssa, double g, double lev_src_bot,
                                     double lev_src_top, double out[4]);
void rrtmgp_oracle_sw_2stream_coeffs(int32_t ftype, double tau, double ssa, double g, double mu0, double out[5]);
/* AngularDiscretization, src/optics/AngularDiscretizations.jl:34-63; ft_size = sizeof(FT) of Ds/wts */
void rrtmgp_oracle_angular_discretization(int n, size_t ft_size, void *Ds, void *wts);
/* McICA counter-based stream (spec: include/rrtmgp_hip.h) and the mask built from it
 * (src/optics/cloud_optics.jl:264-334); mask is uint8 (nlay); returns any(mask) */
double rrtmgp_oracle_mcica_uniform(uint64_t seed, int64_t gcol, int64_t igpt, int32_t is_sw, int32_t draw);
int rrtmgp_oracle_build_cloud_mask(int32_t ftype, uint8_t *mask, const void *cld_frac, int64_t nlay, uint64_t seed,
                                   int64_t gcol, int64_t igpt, int32_t is_sw);

#ifdef __cplusplus
}
#endif
#endif
