// solve.hip — the single solves: spectral LW / SW (device arrays in place, host arrays staged, packed or pipelined), gray
// LW / SW, prepare_atmosphere!; their multi-device dispatch (slice, pick the replicas, re-enter) and entry points.
#include "host.h"
#include "device.h"

namespace rrtmgp {

template <typename FT>
static int solve_lw_t(rrtmgp_workspace *ws, int twostream, const DevGas<FT> &lk, const DevCld<FT> *cld,
                      const DevAero<FT> *aero, int max_minor, const rrtmgp_atmos_state *as, const rrtmgp_lw_bcs *bcs,
                      const rrtmgp_flux_out *flux, const rrtmgp_solve_opts *opts, Stager *chunk = nullptr) {
    RR_CHECK(bcs && bcs->sfc_emis, "LwBCs: sfc_emis is required");
    RR_CHECK(!cld || cld->nband == lk.n_bnd, "cloud lookup band count differs from the gas lookup");
    RR_CHECK(!aero || aero->nband == lk.n_bnd, "aerosol lookup band count differs from the gas lookup");
    const int n_angles = opts ? opts->n_gauss_angles : 1;
    RR_CHECK(twostream || (n_angles >= 1 && n_angles <= 4), "n_gauss_angles must be 1..4");
    Stager own{ws, {}};
    Stager &st = chunk ? *chunk : own;
    DevState<FT> ds;
    TRY(stage_state(st, as, cld != nullptr, aero != nullptr, true, ds, cld ? cld->nrghice : 1));
    const FT *emis, *inc;
    TRY(st.in(bcs->mem, S_BC0, bcs->sfc_emis, (size_t)lk.n_bnd * as->ncol * sizeof(FT), (const void **)&emis));
    // inc_flux is the one array whose fastest dimension is ncol: a column range of it is a 2-D block
    const size_t inc_ld_in = bcs->inc_flux_ld > 0 ? (size_t)bcs->inc_flux_ld : (size_t)as->ncol;
    RR_CHECK(!bcs->inc_flux || inc_ld_in >= (size_t)as->ncol, "LwBCs.inc_flux_ld is smaller than ncol");
    int inc_ld = (int)as->ncol;  // what the kernel sees: host blocks are compacted while they are staged
    if (bcs->inc_flux && inc_ld_in != (size_t)as->ncol && bcs->mem == RRTMGP_MEM_HOST) {
        TRY(st.in2d(S_BC1, bcs->inc_flux, (size_t)as->ncol * sizeof(FT), (size_t)lk.n_gpt, inc_ld_in * sizeof(FT), (const void **)&inc));
    } else {
        if (bcs->inc_flux && bcs->mem == RRTMGP_MEM_DEVICE) inc_ld = (int)inc_ld_in;
        TRY(st.in(bcs->mem, S_BC1, bcs->inc_flux, (size_t)lk.n_gpt * as->ncol * sizeof(FT), (const void **)&inc));
    }
    DevFlux<FT> fl;
    TRY(stage_flux(st, flux, opts, as->ncol, as->nlay + 1, false, fl, twostream ? (size_t)lk.n_bnd : 0));
    if (st.pin_only) return RRTMGP_OK;
    if (chunk && !st.packed) {  // pipelined host path: the uploads ran on the copy stream
        RR_HIP(hipEventRecord(ws->ev_in[0], st.copy_stream()));
        RR_HIP(hipStreamWaitEvent(ws->stream, ws->ev_in[0], 0));
    }
    TRY(st.flush());  // packed small solve: the one upload
    TRY(launch_lw<FT>(ws, twostream, lk, cld, aero, ds, emis, inc, inc_ld, fl, n_angles, opts ? opts->seed : 0,
                      opts ? opts->col_offset : 0, max_minor));
    return chunk && !st.packed ? RRTMGP_OK : st.finish();
}

template <typename FT>
static int solve_sw_t(rrtmgp_workspace *ws, int twostream, const DevGas<FT> &lk, const DevCld<FT> *cld,
                      const DevAero<FT> *aero, int max_minor, const rrtmgp_atmos_state *as, const rrtmgp_sw_bcs *bcs,
                      const rrtmgp_flux_out *flux, const rrtmgp_solve_opts *opts, Stager *chunk = nullptr) {
    RR_CHECK(bcs && bcs->cos_zenith && bcs->toa_flux, "SwBCs: cos_zenith and toa_flux are required");
    RR_CHECK(!twostream || (bcs->sfc_alb_direct && bcs->sfc_alb_diffuse), "SwBCs: surface albedos are required");
    RR_CHECK(!cld || cld->nband == lk.n_bnd, "cloud lookup band count differs from the gas lookup");
    RR_CHECK(!aero || aero->nband == lk.n_bnd, "aerosol lookup band count differs from the gas lookup");
    RR_CHECK(flux && flux->flux_dn_dir, "FluxSW: flux_dn_dir is required");
    Stager own{ws, {}};
    Stager &st = chunk ? *chunk : own;
    DevState<FT> ds;
    TRY(stage_state(st, as, cld != nullptr, aero != nullptr, false, ds, cld ? cld->nrghice : 1));
    const FT *mu0, *toa, *adir, *adif;
    const size_t E = sizeof(FT), ncol = as->ncol;
    TRY(st.in(bcs->mem, S_BC0, bcs->cos_zenith, ncol * E, (const void **)&mu0));
    TRY(st.in(bcs->mem, S_BC1, bcs->toa_flux, ncol * E, (const void **)&toa));
    TRY(st.in(bcs->mem, S_BC2, bcs->sfc_alb_direct, (size_t)lk.n_bnd * ncol * E, (const void **)&adir));
    TRY(st.in(bcs->mem, S_BC3, bcs->sfc_alb_diffuse, (size_t)lk.n_bnd * ncol * E, (const void **)&adif));
    DevFlux<FT> fl;
    TRY(stage_flux(st, flux, opts, ncol, as->nlay + 1, true, fl, twostream ? (size_t)lk.n_bnd : 0));
    if (st.pin_only) return RRTMGP_OK;
    if (chunk && !st.packed) {  // pipelined host path: the uploads ran on the copy stream
        RR_HIP(hipEventRecord(ws->ev_in[0], st.copy_stream()));
        RR_HIP(hipStreamWaitEvent(ws->stream, ws->ev_in[0], 0));
    }
    TRY(st.flush());  // packed small solve: the one upload
    TRY(launch_sw<FT>(ws, twostream, lk, cld, aero, ds, mu0, toa, adir, adif, fl, opts ? opts->seed : 0,
                      opts ? opts->col_offset : 0, max_minor));
    return chunk && !st.packed ? RRTMGP_OK : st.finish();
}

template <typename FT>
static int pin_lw(rrtmgp_workspace *ws, int twostream, const DevGas<FT> &lk, const DevCld<FT> *cld, const DevAero<FT> *aero,
                  int max_minor, const rrtmgp_atmos_state *as, const rrtmgp_lw_bcs *bcs, const rrtmgp_flux_out *flux,
                  const rrtmgp_solve_opts *opts, size_t *need = nullptr) {
    host_pin_begin(ws);
    Stager pin{ws, {}};
    pin.pin_only = true;
    const int rc = solve_lw_t<FT>(ws, twostream, lk, cld, aero, max_minor, as, bcs, flux, opts, &pin);
    if (need) *need = pin.need;  // host bytes the solve stages (what the packed small-solve path sizes its bounce buffer from)
    return rc;
}
template <typename FT>
static int pin_sw(rrtmgp_workspace *ws, int twostream, const DevGas<FT> &lk, const DevCld<FT> *cld, const DevAero<FT> *aero,
                  int max_minor, const rrtmgp_atmos_state *as, const rrtmgp_sw_bcs *bcs, const rrtmgp_flux_out *flux,
                  const rrtmgp_solve_opts *opts, size_t *need = nullptr) {
    host_pin_begin(ws);
    Stager pin{ws, {}};
    pin.pin_only = true;
    const int rc = solve_sw_t<FT>(ws, twostream, lk, cld, aero, max_minor, as, bcs, flux, opts, &pin);
    if (need) *need = pin.need;  // host bytes the solve stages (what the packed small-solve path sizes its bounce buffer from)
    return rc;
}

template <typename FT>
static int solve_lw_host(rrtmgp_workspace *ws, int twostream, const DevGas<FT> &lk, const DevCld<FT> *cld,
                         const DevAero<FT> *aero, int max_minor, const rrtmgp_atmos_state *as, const rrtmgp_lw_bcs *bcs,
                         const rrtmgp_flux_out *flux, const rrtmgp_solve_opts *opts) {
    size_t need = 0;
    PinScope scope{ws};
    TRY(pin_lw<FT>(ws, twostream, lk, cld, aero, max_minor, as, bcs, flux, opts, &need));
    if (!bcs || !host_pipeline_applies(as, bcs->mem, flux, opts)) {
        if (need && need <= host_pack_max()) {   // small solve: one bounce buffer, one DMA each way
            TRY(bounce_ensure(ws, need));
            Stager st{ws, {}};
            st.packed = true;
            return solve_lw_t<FT>(ws, twostream, lk, cld, aero, max_minor, as, bcs, flux, opts, &st);
        }
        return solve_lw_t<FT>(ws, twostream, lk, cld, aero, max_minor, as, bcs, flux, opts);
    }
    return run_host_pipeline(ws, as, flux, opts, sizeof(FT),
                             [&](rrtmgp_atmos_state &a, rrtmgp_flux_out &f, rrtmgp_solve_opts &o, const ColumnSlice &sl, Stager &st) {
                                 rrtmgp_lw_bcs b = *bcs;
                                 slice_lw_bcs(b, sl, (size_t)lk.n_bnd, (size_t)as->ncol);
                                 return solve_lw_t<FT>(ws, twostream, lk, cld, aero, max_minor, &a, &b, &f, &o, &st);
                             });
}

template <typename FT>
static int solve_sw_host(rrtmgp_workspace *ws, int twostream, const DevGas<FT> &lk, const DevCld<FT> *cld,
                         const DevAero<FT> *aero, int max_minor, const rrtmgp_atmos_state *as, const rrtmgp_sw_bcs *bcs,
                         const rrtmgp_flux_out *flux, const rrtmgp_solve_opts *opts) {
    size_t need = 0;
    PinScope scope{ws};
    TRY(pin_sw<FT>(ws, twostream, lk, cld, aero, max_minor, as, bcs, flux, opts, &need));
    if (!bcs || !host_pipeline_applies(as, bcs->mem, flux, opts)) {
        if (need && need <= host_pack_max()) {
            TRY(bounce_ensure(ws, need));
            Stager st{ws, {}};
            st.packed = true;
            return solve_sw_t<FT>(ws, twostream, lk, cld, aero, max_minor, as, bcs, flux, opts, &st);
        }
        return solve_sw_t<FT>(ws, twostream, lk, cld, aero, max_minor, as, bcs, flux, opts);
    }
    return run_host_pipeline(ws, as, flux, opts, sizeof(FT),
                             [&](rrtmgp_atmos_state &a, rrtmgp_flux_out &f, rrtmgp_solve_opts &o, const ColumnSlice &sl, Stager &st) {
                                 rrtmgp_sw_bcs b = *bcs;
                                 b.cos_zenith = sl.adv(b.cos_zenith, 1); b.toa_flux = sl.adv(b.toa_flux, 1);
                                 b.sfc_alb_direct = sl.adv(b.sfc_alb_direct, (size_t)lk.n_bnd);
                                 b.sfc_alb_diffuse = sl.adv(b.sfc_alb_diffuse, (size_t)lk.n_bnd);
                                 return solve_sw_t<FT>(ws, twostream, lk, cld, aero, max_minor, &a, &b, &f, &o, &st);
                             });
}

// prepare_atmosphere! (update_fluxes.jl:252-281): stage every array the cascade touches as in/out
template <typename FT>
static int prepare_t(rrtmgp_workspace *ws, const rrtmgp_atmos_state *as, const rrtmgp_params *ps,
                     const rrtmgp_prepare_opts *o) {
    const size_t E = sizeof(FT), ncol = as->ncol, nlay = as->nlay, nlev = nlay + 1, n2 = nlay * ncol * E;
    RR_CHECK(as->layerdata && as->p_lev && as->t_lev && as->t_sfc, "prepare_atmosphere: missing array");
    Stager st{ws, {}};
    PrepView<FT> v{};
    v.ncol = (int)ncol; v.nlay = (int)nlay; v.ls = 4;
    const int mem = as->mem;
    FT *ld;
    TRY(st.inout(mem, S_LAYERDATA, as->layerdata, 4 * n2, (void **)&ld));
    v.col_dry = ld; v.p_lay = ld + 1; v.t_lay = ld + 2; v.rel_hum = ld + 3;
    TRY(st.inout(mem, S_PLEV, as->p_lev, nlev * ncol * E, (void **)&v.p_lev));
    TRY(st.inout(mem, S_TLEV, as->t_lev, nlev * ncol * E, (void **)&v.t_lev));
    TRY(st.in(mem, S_TSFC, as->t_sfc, ncol * E, (const void **)&v.t_sfc));
    TRY(st.in(mem, S_LAT, as->lat, ncol * E, (const void **)&v.lat));
    if (as->vmr_kind == RRTMGP_VMR_GM) {
        RR_CHECK(as->vmr_h2o, "VmrGM: vmr_h2o is required");
        TRY(st.inout(mem, S_VMR_H2O, as->vmr_h2o, n2, (void **)&v.vmr_h2o));
        TRY(st.inout(mem, S_VMR_O3, as->vmr_o3, n2, (void **)&v.vmr_o3));
        v.hs = 1;
    } else {
        RR_CHECK(as->vmr && o->idx_h2o >= 1 && o->idx_h2o <= as->ngas, "Vmr: idx_h2o out of range");
        TRY(st.inout(mem, S_VMR, as->vmr, (size_t)as->ngas * n2, (void **)&v.vmr_full));
        v.ngas = (int)as->ngas; v.hs = (int)as->ngas; v.vmr_h2o = v.vmr_full + (o->idx_h2o - 1);
    }
    if ((o->steps & RRTMGP_PREP_ISOTHERMAL) && o->isothermal_boundary_layer) {
        const void *cl[5] = {as->cld_r_eff_liq, as->cld_r_eff_ice, as->cld_path_liq, as->cld_path_ice, as->cld_frac};
        const int slot[5] = {S_CLD_RL, S_CLD_RI, S_CLD_PL, S_CLD_PI, S_CLD_F};
        for (int i = 0; i < 5; i++) TRY(st.inout(mem, slot[i], cl[i], n2, (void **)&v.cld[i]));
        TRY(st.inout(mem, S_AERO_SIZE, as->aero_size, RRTMGP_N_AEROSOLS * n2, (void **)&v.aero[0]));
        TRY(st.inout(mem, S_AERO_MASS, as->aero_mass, RRTMGP_N_AEROSOLS * n2, (void **)&v.aero[1]));
    }
    TRY(st.in(o->z_mem, S_ZC, o->center_z, n2, (const void **)&v.center_z));
    TRY(st.in(o->z_mem, S_ZF, o->face_z, nlev * ncol * E, (const void **)&v.face_z));
    TRY(launch_prepare<FT>(ws, v, *ps, *o, false));
    return st.finish();
}

template <typename FT>
static int prepare_gray_t(rrtmgp_workspace *ws, const rrtmgp_gray_state *gs, const rrtmgp_params *ps,
                          const rrtmgp_prepare_opts *o) {
    const size_t E = sizeof(FT), ncol = gs->ncol, nlay = gs->nlay, nlev = nlay + 1;
    RR_CHECK(gs->p_lay && gs->p_lev && gs->t_lay && gs->t_lev && gs->t_sfc, "prepare_atmosphere (gray): missing array");
    Stager st{ws, {}};
    PrepView<FT> v{};
    v.ncol = (int)ncol; v.nlay = (int)nlay; v.ls = 1;
    TRY(st.inout(gs->mem, S_PLAY, gs->p_lay, nlay * ncol * E, (void **)&v.p_lay));
    TRY(st.inout(gs->mem, S_TLAY, gs->t_lay, nlay * ncol * E, (void **)&v.t_lay));
    TRY(st.inout(gs->mem, S_PLEV, gs->p_lev, nlev * ncol * E, (void **)&v.p_lev));
    TRY(st.inout(gs->mem, S_TLEV, gs->t_lev, nlev * ncol * E, (void **)&v.t_lev));
    TRY(st.in(gs->mem, S_TSFC, gs->t_sfc, ncol * E, (const void **)&v.t_sfc));
    TRY(st.in(o->z_mem, S_ZC, o->center_z, nlay * ncol * E, (const void **)&v.center_z));
    TRY(st.in(o->z_mem, S_ZF, o->face_z, nlev * ncol * E, (const void **)&v.face_z));
    rrtmgp_prepare_opts og = *o;
    og.steps &= ~RRTMGP_PREP_COL_DRY;
    TRY(launch_prepare<FT>(ws, v, *ps, og, true));
    return st.finish();
}

template <typename FT>
static int solve_gray_lw_t(rrtmgp_workspace *ws, int twostream, const rrtmgp_gray_state *gs, const rrtmgp_lw_bcs *bcs,
                           const rrtmgp_flux_out *flux, const rrtmgp_solve_opts *opts) {
    const size_t E = sizeof(FT), ncol = gs->ncol, nlay = gs->nlay, nlev = nlay + 1;
    Stager st{ws, {}};
    const FT *lat, *p_lay, *p_lev, *t_lay, *t_lev, *t_sfc, *emis, *inc;
    TRY(st.in(gs->mem, S_LAT, gs->lat, ncol * E, (const void **)&lat));
    TRY(st.in(gs->mem, S_PLAY, gs->p_lay, nlay * ncol * E, (const void **)&p_lay));
    TRY(st.in(gs->mem, S_PLEV, gs->p_lev, nlev * ncol * E, (const void **)&p_lev));
    TRY(st.in(gs->mem, S_TLAY, gs->t_lay, nlay * ncol * E, (const void **)&t_lay));
    TRY(st.in(gs->mem, S_TLEV, gs->t_lev, nlev * ncol * E, (const void **)&t_lev));
    TRY(st.in(gs->mem, S_TSFC, gs->t_sfc, ncol * E, (const void **)&t_sfc));
    TRY(st.in(bcs->mem, S_BC0, bcs->sfc_emis, ncol * E, (const void **)&emis));
    TRY(st.in(bcs->mem, S_BC1, bcs->inc_flux, ncol * E, (const void **)&inc));
    DevFlux<FT> fl;
    TRY(stage_flux(st, flux, opts, ncol, nlev, false, fl));
    GrayArgs ga;
    ga.otp_kind = gs->otp_kind;
    for (int i = 0; i < 5; i++) ga.otp[i] = gs->otp[i];
    ga.stefan = gs->stefan;
    TRY(launch_gray_lw<FT>(ws, twostream, (int)ncol, (int)nlay, ga, lat, p_lay, p_lev, t_lay, t_lev, t_sfc, emis, inc, fl));
    return st.finish();
}

template <typename FT>
static int solve_gray_sw_t(rrtmgp_workspace *ws, int twostream, const rrtmgp_gray_state *gs, const rrtmgp_sw_bcs *bcs,
                           const rrtmgp_flux_out *flux, const rrtmgp_solve_opts *opts) {
    const size_t E = sizeof(FT), ncol = gs->ncol, nlay = gs->nlay, nlev = nlay + 1;
    Stager st{ws, {}};
    const FT *p_lay, *p_lev, *mu0, *toa, *adir, *adif;
    TRY(st.in(gs->mem, S_PLAY, gs->p_lay, nlay * ncol * E, (const void **)&p_lay));
    TRY(st.in(gs->mem, S_PLEV, gs->p_lev, nlev * ncol * E, (const void **)&p_lev));
    TRY(st.in(bcs->mem, S_BC0, bcs->cos_zenith, ncol * E, (const void **)&mu0));
    TRY(st.in(bcs->mem, S_BC1, bcs->toa_flux, ncol * E, (const void **)&toa));
    TRY(st.in(bcs->mem, S_BC2, bcs->sfc_alb_direct, ncol * E, (const void **)&adir));
    TRY(st.in(bcs->mem, S_BC3, bcs->sfc_alb_diffuse, ncol * E, (const void **)&adif));
    DevFlux<FT> fl;
    TRY(stage_flux(st, flux, opts, ncol, nlev, true, fl));
    GrayArgs ga;
    ga.otp_kind = gs->otp_kind;
    for (int i = 0; i < 5; i++) ga.otp[i] = gs->otp[i];
    ga.stefan = gs->stefan;
    TRY(launch_gray_sw<FT>(ws, twostream, (int)ncol, (int)nlay, ga, p_lay, p_lev, mu0, toa, adir, adif, fl));
    return st.finish();
}

}  // namespace rrtmgp

using namespace rrtmgp;

#define GAS_DISPATCH(ws, fn, lk, cld, aero, ...)                                                                      \
    ((ws)->ftype == RRTMGP_F32                                                                                        \
         ? fn<float>(ws, twostream, (lk)->gas32, (cld) ? &(cld)->cld32 : nullptr, (aero) ? &(aero)->aero32 : nullptr, \
                     (lk)->max_int, __VA_ARGS__)                                                                    \
         : fn<double>(ws, twostream, (lk)->gas64, (cld) ? &(cld)->cld64 : nullptr, (aero) ? &(aero)->aero64 : nullptr, \
                      (lk)->max_int, __VA_ARGS__))


// ---- multi-device dispatch of the spectral solvers: slice, pick the replicas, re-enter the single-device entry ----
template <typename BCS, typename SliceBcs, typename Call>
static int multi_spectral(rrtmgp_workspace *ws, const rrtmgp_lookup *gas, const rrtmgp_lookup *cld, const rrtmgp_lookup *aero,
                          const rrtmgp_atmos_state *as, const BCS *bcs, const rrtmgp_flux_out *flux,
                          const rrtmgp_solve_opts *opts, SliceBcs slice_bcs, Call call) {
    const size_t E = (size_t)ws->ftype, nlev = (size_t)as->nlay + 1;
    return multi_run(ws, [&](rrtmgp_workspace *sw, size_t c0, size_t nc) -> int {
        const ColumnSlice sl{E, c0};
        rrtmgp_atmos_state a = *as;
        BCS b = *bcs;
        rrtmgp_flux_out f = *flux;
        rrtmgp_solve_opts o{};
        if (opts) o = *opts; else o.n_gauss_angles = 1;
        slice_state(a, sl, nc);
        slice_flux(f, o, sl, nlev, (size_t)as->ncol);
        slice_bcs(b, sl);
        const rrtmgp_lookup *g = lookup_on(gas, sw->device), *c = lookup_on(cld, sw->device), *ae = lookup_on(aero, sw->device);
        if (!g || (cld && !c) || (aero && !ae))
            return set_error(RRTMGP_EINVAL, "a lookup has no replica on one of the workspace's devices (use *_lookup_create_multi)");
        return call(sw, g, c, ae, &a, &b, &f, &o);
    }, as->mem == RRTMGP_MEM_DEVICE || bcs->mem == RRTMGP_MEM_DEVICE || flux->mem == RRTMGP_MEM_DEVICE ||
       (opts && opts->metric_scaling && opts->metric_mem == RRTMGP_MEM_DEVICE));
}
static int check_multi_spectral(rrtmgp_workspace *ws, const rrtmgp_lookup *gas, const rrtmgp_atmos_state *as, const void *bcs,
                                int bcs_mem, const rrtmgp_flux_out *flux, const rrtmgp_solve_opts *opts, const void *inc_flux) {
    RR_CHECK(ws && gas && as && bcs && flux, "null argument");
    RR_CHECK(gas->kind == LK_GAS && gas->ftype == ws->ftype, "expected a gas lookup of the workspace's precision");
    RR_CHECK(as->ncol == ws->ncol && as->nlay == ws->nlay, "state dimensions differ from the workspace");
    return check_multi(ws, as->mem, bcs_mem, flux, opts, inc_flux);
}
static const rrtmgp_lookup *head_replica(const rrtmgp_workspace *ws, const rrtmgp_lookup *lk) {
    return lk ? lookup_on(lk, ws->shards[0]->device) : nullptr;
}

extern "C" {

int rrtmgp_hip_rte_lw_2stream_solve(rrtmgp_workspace *ws, const rrtmgp_lookup *lookup_lw, const rrtmgp_lookup *cld,
                                    const rrtmgp_lookup *aero, const rrtmgp_atmos_state *as, const rrtmgp_lw_bcs *bcs,
                                    const rrtmgp_flux_out *flux, const rrtmgp_solve_opts *opts) {
    const int twostream = 1;
    if (ws && !ws->shards.empty()) {
        TRY(check_multi_spectral(ws, lookup_lw, as, bcs, bcs ? bcs->mem : 0, flux, opts, bcs ? bcs->inc_flux : nullptr));
        const rrtmgp_lookup *g = head_replica(ws, lookup_lw), *c = head_replica(ws, cld), *ae = head_replica(ws, aero);
        RR_CHECK(g && (!cld || c) && (!aero || ae), "a lookup has no replica on the workspace's first device");
        PinScope scope{ws};
        TRY(GAS_DISPATCH(ws, pin_lw, g, c, ae, as, bcs, flux, opts));
        const size_t nb = (size_t)n_bnd_of(ws, lookup_lw);
        return multi_spectral(ws, lookup_lw, cld, aero, as, bcs, flux, opts,
                              [nb, as](rrtmgp_lw_bcs &b, const ColumnSlice &sl) { slice_lw_bcs(b, sl, nb, (size_t)as->ncol); },
                              rrtmgp_hip_rte_lw_2stream_solve);
    }
    TRY(check_common(ws, lookup_lw, 0, cld, aero, as));
    return GAS_DISPATCH(ws, solve_lw_host, lookup_lw, cld, aero, as, bcs, flux, opts);
}

int rrtmgp_hip_rte_lw_noscat_solve(rrtmgp_workspace *ws, const rrtmgp_lookup *lookup_lw, const rrtmgp_lookup *cld,
                                   const rrtmgp_lookup *aero, const rrtmgp_atmos_state *as, const rrtmgp_lw_bcs *bcs,
                                   const rrtmgp_flux_out *flux, const rrtmgp_solve_opts *opts) {
    const int twostream = 0;
    if (ws && !ws->shards.empty()) {
        TRY(check_multi_spectral(ws, lookup_lw, as, bcs, bcs ? bcs->mem : 0, flux, opts, bcs ? bcs->inc_flux : nullptr));
        const rrtmgp_lookup *g = head_replica(ws, lookup_lw), *c = head_replica(ws, cld), *ae = head_replica(ws, aero);
        RR_CHECK(g && (!cld || c) && (!aero || ae), "a lookup has no replica on the workspace's first device");
        PinScope scope{ws};
        TRY(GAS_DISPATCH(ws, pin_lw, g, c, ae, as, bcs, flux, opts));
        const size_t nb = (size_t)n_bnd_of(ws, lookup_lw);
        return multi_spectral(ws, lookup_lw, cld, aero, as, bcs, flux, opts,
                              [nb, as](rrtmgp_lw_bcs &b, const ColumnSlice &sl) { slice_lw_bcs(b, sl, nb, (size_t)as->ncol); },
                              rrtmgp_hip_rte_lw_noscat_solve);
    }
    TRY(check_common(ws, lookup_lw, 0, cld, aero, as));
    return GAS_DISPATCH(ws, solve_lw_host, lookup_lw, cld, aero, as, bcs, flux, opts);
}

int rrtmgp_hip_rte_sw_2stream_solve(rrtmgp_workspace *ws, const rrtmgp_lookup *lookup_sw, const rrtmgp_lookup *cld,
                                    const rrtmgp_lookup *aero, const rrtmgp_atmos_state *as, const rrtmgp_sw_bcs *bcs,
                                    const rrtmgp_flux_out *flux, const rrtmgp_solve_opts *opts) {
    const int twostream = 1;
    if (ws && !ws->shards.empty()) {
        TRY(check_multi_spectral(ws, lookup_sw, as, bcs, bcs ? bcs->mem : 0, flux, opts, nullptr));
        const rrtmgp_lookup *g = head_replica(ws, lookup_sw), *c = head_replica(ws, cld), *ae = head_replica(ws, aero);
        RR_CHECK(g && (!cld || c) && (!aero || ae), "a lookup has no replica on the workspace's first device");
        PinScope scope{ws};
        TRY(GAS_DISPATCH(ws, pin_sw, g, c, ae, as, bcs, flux, opts));
        const size_t nb = (size_t)n_bnd_of(ws, lookup_sw);
        return multi_spectral(ws, lookup_sw, cld, aero, as, bcs, flux, opts,
                              [nb](rrtmgp_sw_bcs &b, const ColumnSlice &sl) { slice_sw_bcs(b, sl, nb); },
                              rrtmgp_hip_rte_sw_2stream_solve);
    }
    TRY(check_common(ws, lookup_sw, 1, cld, aero, as));
    return GAS_DISPATCH(ws, solve_sw_host, lookup_sw, cld, aero, as, bcs, flux, opts);
}

int rrtmgp_hip_rte_sw_noscat_solve(rrtmgp_workspace *ws, const rrtmgp_lookup *lookup_sw, const rrtmgp_atmos_state *as,
                                   const rrtmgp_sw_bcs *bcs, const rrtmgp_flux_out *flux, const rrtmgp_solve_opts *opts) {
    const int twostream = 0;
    const rrtmgp_lookup *cld = nullptr, *aero = nullptr;
    if (ws && !ws->shards.empty()) {
        TRY(check_multi_spectral(ws, lookup_sw, as, bcs, bcs ? bcs->mem : 0, flux, opts, nullptr));
        const rrtmgp_lookup *g = head_replica(ws, lookup_sw);
        RR_CHECK(g, "the lookup has no replica on the workspace's first device");
        PinScope scope{ws};
        TRY(GAS_DISPATCH(ws, pin_sw, g, cld, aero, as, bcs, flux, opts));
        const size_t nb = (size_t)n_bnd_of(ws, lookup_sw);
        return multi_spectral(ws, lookup_sw, cld, aero, as, bcs, flux, opts,
                              [nb](rrtmgp_sw_bcs &b, const ColumnSlice &sl) { slice_sw_bcs(b, sl, nb); },
                              [](rrtmgp_workspace *sw, const rrtmgp_lookup *g2, const rrtmgp_lookup *, const rrtmgp_lookup *,
                                 const rrtmgp_atmos_state *a, const rrtmgp_sw_bcs *b, const rrtmgp_flux_out *f,
                                 const rrtmgp_solve_opts *o) { return rrtmgp_hip_rte_sw_noscat_solve(sw, g2, a, b, f, o); });
    }
    TRY(check_common(ws, lookup_sw, 1, nullptr, nullptr, as));
    return GAS_DISPATCH(ws, solve_sw_host, lookup_sw, cld, aero, as, bcs, flux, opts);
}

}  // extern "C"


// gray solves / preparation steps on a multi-device workspace: same slicing, no lookups
template <typename BCS, typename SliceBcs, typename Call>
static int multi_gray(rrtmgp_workspace *ws, const rrtmgp_gray_state *gs, const BCS *bcs, const rrtmgp_flux_out *flux,
                      const rrtmgp_solve_opts *opts, SliceBcs slice_bcs, Call call) {
    RR_CHECK(gs && bcs && flux, "null argument");
    RR_CHECK(gs->ncol == ws->ncol && gs->nlay == ws->nlay, "state dimensions differ from the workspace");
    TRY(check_multi(ws, gs->mem, bcs->mem, flux, opts, nullptr));
    const size_t E = (size_t)ws->ftype, nlev = (size_t)gs->nlay + 1;
    return multi_run(ws, [&](rrtmgp_workspace *sw, size_t c0, size_t nc) -> int {
        const ColumnSlice sl{E, c0};
        rrtmgp_gray_state g = *gs;
        BCS b = *bcs;
        rrtmgp_flux_out f = *flux;
        rrtmgp_solve_opts o{};
        if (opts) o = *opts; else o.n_gauss_angles = 1;
        slice_gray(g, sl, nc);
        slice_flux(f, o, sl, nlev, (size_t)gs->ncol);
        slice_bcs(b, sl);
        return call(sw, &g, &b, &f, &o);
    }, gs->mem == RRTMGP_MEM_DEVICE || bcs->mem == RRTMGP_MEM_DEVICE || flux->mem == RRTMGP_MEM_DEVICE ||
       (opts && opts->metric_scaling && opts->metric_mem == RRTMGP_MEM_DEVICE));
}
static void slice_gray_lw_bcs(rrtmgp_lw_bcs &b, const ColumnSlice &sl) { b.sfc_emis = sl.adv(b.sfc_emis, 1); b.inc_flux = sl.adv(b.inc_flux, 1); }
static void slice_gray_sw_bcs(rrtmgp_sw_bcs &b, const ColumnSlice &sl) { slice_sw_bcs(b, sl, 1); }

extern "C" {

static int check_gray(rrtmgp_workspace *ws, const rrtmgp_gray_state *gs, const void *bcs, const rrtmgp_flux_out *flux) {
    RR_CHECK(ws && gs && bcs && flux, "null argument");
    RR_CHECK(gs->ncol == ws->ncol && gs->nlay == ws->nlay, "state dimensions differ from the workspace");
    RR_CHECK(gs->otp_kind == 0 || gs->otp_kind == 1, "unknown gray optical-thickness kind");
    RR_CHECK(gs->p_lay && gs->p_lev, "gray state: missing pressure arrays");
    RR_HIP(hipSetDevice(ws->device));
    return RRTMGP_OK;
}

int rrtmgp_hip_rte_lw_2stream_solve_gray(rrtmgp_workspace *ws, const rrtmgp_gray_state *as, const rrtmgp_lw_bcs *bcs,
                                         const rrtmgp_flux_out *flux, const rrtmgp_solve_opts *opts) {
    if (ws && !ws->shards.empty()) return multi_gray(ws, as, bcs, flux, opts, slice_gray_lw_bcs, rrtmgp_hip_rte_lw_2stream_solve_gray);
    TRY(check_gray(ws, as, bcs, flux));
    RR_CHECK(as->lat && as->t_lay && as->t_lev && as->t_sfc && bcs->sfc_emis, "gray LW: missing array");
    return ws->ftype == RRTMGP_F32 ? solve_gray_lw_t<float>(ws, 1, as, bcs, flux, opts)
                                   : solve_gray_lw_t<double>(ws, 1, as, bcs, flux, opts);
}
int rrtmgp_hip_rte_lw_noscat_solve_gray(rrtmgp_workspace *ws, const rrtmgp_gray_state *as, const rrtmgp_lw_bcs *bcs,
                                        const rrtmgp_flux_out *flux, const rrtmgp_solve_opts *opts) {
    if (ws && !ws->shards.empty()) return multi_gray(ws, as, bcs, flux, opts, slice_gray_lw_bcs, rrtmgp_hip_rte_lw_noscat_solve_gray);
    TRY(check_gray(ws, as, bcs, flux));
    RR_CHECK(as->lat && as->t_lay && as->t_lev && as->t_sfc && bcs->sfc_emis, "gray LW: missing array");
    RR_CHECK(!opts || opts->n_gauss_angles <= 1, "gray radiation is solved with a single quadrature angle");
    return ws->ftype == RRTMGP_F32 ? solve_gray_lw_t<float>(ws, 0, as, bcs, flux, opts)
                                   : solve_gray_lw_t<double>(ws, 0, as, bcs, flux, opts);
}
int rrtmgp_hip_rte_sw_2stream_solve_gray(rrtmgp_workspace *ws, const rrtmgp_gray_state *as, const rrtmgp_sw_bcs *bcs,
                                         const rrtmgp_flux_out *flux, const rrtmgp_solve_opts *opts) {
    if (ws && !ws->shards.empty()) return multi_gray(ws, as, bcs, flux, opts, slice_gray_sw_bcs, rrtmgp_hip_rte_sw_2stream_solve_gray);
    TRY(check_gray(ws, as, bcs, flux));
    RR_CHECK(bcs->cos_zenith && bcs->toa_flux && bcs->sfc_alb_direct && bcs->sfc_alb_diffuse && flux->flux_dn_dir,
             "gray SW: missing array");
    return ws->ftype == RRTMGP_F32 ? solve_gray_sw_t<float>(ws, 1, as, bcs, flux, opts)
                                   : solve_gray_sw_t<double>(ws, 1, as, bcs, flux, opts);
}
int rrtmgp_hip_rte_sw_noscat_solve_gray(rrtmgp_workspace *ws, const rrtmgp_gray_state *as, const rrtmgp_sw_bcs *bcs,
                                        const rrtmgp_flux_out *flux, const rrtmgp_solve_opts *opts) {
    if (ws && !ws->shards.empty()) return multi_gray(ws, as, bcs, flux, opts, slice_gray_sw_bcs, rrtmgp_hip_rte_sw_noscat_solve_gray);
    TRY(check_gray(ws, as, bcs, flux));
    RR_CHECK(bcs->cos_zenith && bcs->toa_flux && flux->flux_dn_dir, "gray SW: missing array");
    return ws->ftype == RRTMGP_F32 ? solve_gray_sw_t<float>(ws, 0, as, bcs, flux, opts)
                                   : solve_gray_sw_t<double>(ws, 0, as, bcs, flux, opts);
}

int rrtmgp_hip_prepare_atmosphere(rrtmgp_workspace *ws, const rrtmgp_atmos_state *as, const rrtmgp_params *params,
                                  const rrtmgp_prepare_opts *opts) {
    RR_CHECK(ws && as && params && opts, "null argument");
    RR_CHECK(as->ncol >= 0 && as->ncol <= ws->ncol && as->nlay == ws->nlay, "state does not fit the workspace");
    if (!ws->shards.empty()) {
        RR_CHECK(as->ncol == ws->ncol, "state dimensions differ from the workspace");
        const size_t nlay = (size_t)as->nlay;
        return multi_run(ws, [&](rrtmgp_workspace *sw, size_t c0, size_t nc) -> int {
            const ColumnSlice sl{(size_t)ws->ftype, c0};
            rrtmgp_atmos_state a = *as;
            rrtmgp_prepare_opts o = *opts;
            slice_state(a, sl, nc);
            o.center_z = sl.adv(o.center_z, nlay); o.face_z = sl.adv(o.face_z, nlay + 1);
            return rrtmgp_hip_prepare_atmosphere(sw, &a, params, &o);
        }, as->mem == RRTMGP_MEM_DEVICE);
    }
    RR_HIP(hipSetDevice(ws->device));
    return ws->ftype == RRTMGP_F32 ? prepare_t<float>(ws, as, params, opts) : prepare_t<double>(ws, as, params, opts);
}

int rrtmgp_hip_prepare_atmosphere_gray(rrtmgp_workspace *ws, const rrtmgp_gray_state *as, const rrtmgp_params *params,
                                       const rrtmgp_prepare_opts *opts) {
    RR_CHECK(ws && as && params && opts, "null argument");
    RR_CHECK(as->ncol >= 0 && as->ncol <= ws->ncol && as->nlay == ws->nlay, "state does not fit the workspace");
    if (!ws->shards.empty()) {
        RR_CHECK(as->ncol == ws->ncol, "state dimensions differ from the workspace");
        const size_t nlay = (size_t)as->nlay;
        return multi_run(ws, [&](rrtmgp_workspace *sw, size_t c0, size_t nc) -> int {
            const ColumnSlice sl{(size_t)ws->ftype, c0};
            rrtmgp_gray_state g = *as;
            rrtmgp_prepare_opts o = *opts;
            slice_gray(g, sl, nc);
            o.center_z = sl.adv(o.center_z, nlay); o.face_z = sl.adv(o.face_z, nlay + 1);
            return rrtmgp_hip_prepare_atmosphere_gray(sw, &g, params, &o);
        }, as->mem == RRTMGP_MEM_DEVICE);
    }
    RR_HIP(hipSetDevice(ws->device));
    return ws->ftype == RRTMGP_F32 ? prepare_gray_t<float>(ws, as, params, opts)
                                   : prepare_gray_t<double>(ws, as, params, opts);
}

}  // extern "C"
