// staging.hip — which arrays of a state / flux descriptor cross to the device and back (Stager does the moving: host.h), the
// argument checks every solve shares, and the streams / events the pipelined and two-lane paths create on first use.
#include "host.h"
#include "device.h"

namespace rrtmgp {

// `nrghice`: roughness classes of the cloud lookup the solve uses; 0 = NO cloud lookup reads the cloud arrays (they are staged
// for the isothermal-layer preparation only: whichever of them the state carries, and `ice_rgh` is nobody's business).
// `aero_lookup` likewise for the aerosol arrays.
template <typename FT>
int stage_state(Stager &st, const rrtmgp_atmos_state *as, bool use_cld, bool use_aero, bool lw, DevState<FT> &d, int64_t nrghice,
                StateRW rw, bool aero_lookup) {
    const size_t E = sizeof(FT), ncol = as->ncol, nlay = as->nlay, nlev = nlay + 1;
    RR_CHECK(as->layerdata && as->t_sfc && as->vmr, "atmospheric state: missing array");
    RR_CHECK(!lw || as->t_lev, "atmospheric state: t_lev is required for longwave");
    d.ncol = (int)ncol; d.nlay = (int)nlay; d.ngas = (int)as->ngas; d.vmr_kind = as->vmr_kind;
    const int mem = as->mem;
    TRY(st.io(rw.core, mem, S_LAYERDATA, as->layerdata, 4 * nlay * ncol * E, (const void **)&d.layerdata));
    d.t_lev = nullptr;   // the shortwave kernels never read the level temperatures: not uploaded for them
    if (lw) TRY(st.io(rw.core, mem, S_TLEV, as->t_lev, nlev * ncol * E, (const void **)&d.t_lev));
    TRY(st.in(mem, S_TSFC, as->t_sfc, ncol * E, (const void **)&d.t_sfc));
    if (as->vmr_kind == RRTMGP_VMR_GM) {
        RR_CHECK(as->vmr_h2o && as->vmr_o3, "VmrGM: vmr_h2o and vmr_o3 are required");
        TRY(st.io(rw.core, mem, S_VMR_H2O, as->vmr_h2o, nlay * ncol * E, (const void **)&d.vmr_h2o));
        TRY(st.io(rw.particles, mem, S_VMR_O3, as->vmr_o3, nlay * ncol * E, (const void **)&d.vmr_o3));
        TRY(st.in(mem, S_VMR, as->vmr, as->ngas * E, (const void **)&d.vmr));
    } else {
        d.vmr_h2o = d.vmr_o3 = nullptr;
        TRY(st.io(rw.core, mem, S_VMR, as->vmr, (size_t)as->ngas * nlay * ncol * E, (const void **)&d.vmr));
    }
    d.cld_r_eff_liq = d.cld_r_eff_ice = d.cld_path_liq = d.cld_path_ice = d.cld_frac = nullptr;
    d.cld_cover = nullptr;
    d.ice_rgh = (int)as->ice_rgh;
    if (use_cld) {
        const bool lookup = nrghice > 0;
        if (lookup) {
            RR_CHECK(as->cld_frac && as->cld_r_eff_liq && as->cld_r_eff_ice && as->cld_path_liq && as->cld_path_ice,
                     "cloud lookup given but the state has no CloudState");
            RR_CHECK(as->ice_rgh >= 1 && as->ice_rgh <= nrghice, "ice_rgh must be in 1..nrghice of the cloud lookup");
        }
        // (st.io of a null array leaves the device pointer null: the preparation skips what the state does not carry)
        TRY(st.io(rw.particles, mem, S_CLD_RL, as->cld_r_eff_liq, nlay * ncol * E, (const void **)&d.cld_r_eff_liq));
        TRY(st.io(rw.particles, mem, S_CLD_RI, as->cld_r_eff_ice, nlay * ncol * E, (const void **)&d.cld_r_eff_ice));
        TRY(st.io(rw.particles, mem, S_CLD_PL, as->cld_path_liq, nlay * ncol * E, (const void **)&d.cld_path_liq));
        TRY(st.io(rw.particles, mem, S_CLD_PI, as->cld_path_ice, nlay * ncol * E, (const void **)&d.cld_path_ice));
        TRY(st.io(rw.particles, mem, S_CLD_F, as->cld_frac, nlay * ncol * E, (const void **)&d.cld_frac));
        if (lookup) TRY(st.out(mem, S_CLD_COVER, lw ? as->cld_cover_lw : as->cld_cover_sw, ncol * E, (void **)&d.cld_cover));
    }
    d.aero_size = d.aero_mass = nullptr;
    d.aod_sw_ext = d.aod_sw_sca = nullptr;
    if (use_aero) {
        RR_CHECK(!aero_lookup || (as->aero_size && as->aero_mass), "aerosol lookup given but the state has no AerosolState");
        const size_t n = (size_t)RRTMGP_N_AEROSOLS * nlay * ncol * E;
        TRY(st.io(rw.particles, mem, S_AERO_SIZE, as->aero_size, n, (const void **)&d.aero_size));
        TRY(st.io(rw.particles, mem, S_AERO_MASS, as->aero_mass, n, (const void **)&d.aero_mass));
        if (!lw && aero_lookup) {
            RR_CHECK((as->aod_sw_ext == nullptr) == (as->aod_sw_sca == nullptr), "aod_sw_ext and aod_sw_sca go together");
            TRY(st.out(mem, S_AOD_EXT, as->aod_sw_ext, ncol * E, (void **)&d.aod_sw_ext));
            TRY(st.out(mem, S_AOD_SCA, as->aod_sw_sca, ncol * E, (void **)&d.aod_sw_sca));
        }
    }
    return RRTMGP_OK;
}

template <typename FT>
int stage_flux(Stager &st, const rrtmgp_flux_out *f, const rrtmgp_solve_opts *opts, size_t ncol, size_t nlev, bool sw, DevFlux<FT> &d,
               size_t nbnd, int so, const FT *staged_metric) {
    RR_CHECK(f && f->flux_up && f->flux_dn && f->flux_net, "flux outputs: missing array");
    RR_CHECK(f->layout == RRTMGP_LAYOUT_NCOL_NLEV || f->layout == RRTMGP_LAYOUT_NLEV_NCOL, "bad flux layout");
    const size_t bytes = ncol * nlev * sizeof(FT);
    // (ncol, nlev) fluxes: a column range of wider arrays is nlev rows of ncol values, flux_ncol apart
    const size_t fcols = f->layout == RRTMGP_LAYOUT_NCOL_NLEV && f->flux_ncol > 0 ? (size_t)f->flux_ncol : ncol;
    RR_CHECK(fcols >= ncol, "flux_ncol is smaller than ncol");
    const bool strided = fcols != ncol && f->mem == RRTMGP_MEM_HOST;  // packed on the device, strided on the way home
    d.ld = f->mem == RRTMGP_MEM_DEVICE ? (int)fcols : (int)ncol;
    auto flux_out = [&](int slot, void *p, FT **dev) -> int {
        if (strided) return st.out2d(slot, p, ncol * sizeof(FT), nlev, fcols * sizeof(FT), (void **)dev);
        return st.out(f->mem, slot, p, bytes, (void **)dev);
    };
    TRY(flux_out(S_FLUX_UP + so, f->flux_up, &d.up));
    TRY(flux_out(S_FLUX_DN + so, f->flux_dn, &d.dn));
    TRY(flux_out(S_FLUX_NET + so, f->flux_net, &d.net));
    d.dir = nullptr;
    if (sw) TRY(flux_out(S_FLUX_DIR + so, f->flux_dn_dir, &d.dir));
    d.layout = f->layout;
    d.band_up = d.band_dn = d.band_net = nullptr;
    d.band_ncol = (int)ncol;
    if (f->band_flux_up || f->band_flux_dn || f->band_flux_net) {
        RR_CHECK(nbnd > 0, "per-band fluxes are only available from the two-stream, non-gray solvers");
        RR_CHECK(f->band_flux_up && f->band_flux_dn, "per-band fluxes: band_flux_up and band_flux_dn go together");
        // (nlev, ncol, nbnd): a column range of wider arrays is nbnd blocks of nlev * ncol values, band_flux_ncol * nlev apart
        const size_t bcols = f->band_flux_ncol > 0 ? (size_t)f->band_flux_ncol : ncol;
        RR_CHECK(bcols >= ncol, "band_flux_ncol is smaller than ncol");
        d.band_ncol = (int)ncol;  // host blocks are packed in the staging buffers and strided on the way home
        if (bcols != ncol && f->mem == RRTMGP_MEM_HOST) {
            const size_t w = bytes, pitch = bcols * nlev * sizeof(FT);
            TRY(st.out2d(S_BAND_UP + so, f->band_flux_up, w, nbnd, pitch, (void **)&d.band_up));
            TRY(st.out2d(S_BAND_DN + so, f->band_flux_dn, w, nbnd, pitch, (void **)&d.band_dn));
            if (f->band_flux_net) TRY(st.out2d(S_BAND_NET + so, f->band_flux_net, w, nbnd, pitch, (void **)&d.band_net));
        } else {
            if (f->mem == RRTMGP_MEM_DEVICE) d.band_ncol = (int)bcols;
            TRY(st.out(f->mem, S_BAND_UP + so, f->band_flux_up, bytes * nbnd, (void **)&d.band_up));
            TRY(st.out(f->mem, S_BAND_DN + so, f->band_flux_dn, bytes * nbnd, (void **)&d.band_dn));
            if (f->band_flux_net) TRY(st.out(f->mem, S_BAND_NET + so, f->band_flux_net, bytes * nbnd, (void **)&d.band_net));
        }
    }
    d.clear_up = d.clear_dn = d.clear_net = d.clear_dir = nullptr;
    if (f->clear_flux_up || f->clear_flux_dn || f->clear_flux_net || f->clear_flux_dn_dir) {
        RR_CHECK(nbnd > 0, "the clear-sky diagnostic is only available from the two-stream, non-gray solvers");
        RR_CHECK(f->clear_flux_up && f->clear_flux_dn && f->clear_flux_net && (!sw || f->clear_flux_dn_dir),
                 "clear-sky diagnostic: clear_flux_up / _dn / _net (and _dn_dir for SW) go together");
        TRY(flux_out(S_CLR_UP + so, f->clear_flux_up, &d.clear_up));
        TRY(flux_out(S_CLR_DN + so, f->clear_flux_dn, &d.clear_dn));
        TRY(flux_out(S_CLR_NET + so, f->clear_flux_net, &d.clear_net));
        if (sw) TRY(flux_out(S_CLR_DIR + so, f->clear_flux_dn_dir, &d.clear_dir));
    }
    d.metric = staged_metric;
    if (!staged_metric && opts && opts->metric_scaling)
        TRY(st.in(opts->metric_mem, S_METRIC, opts->metric_scaling, bytes, (const void **)&d.metric));
    return RRTMGP_OK;
}

template int stage_state<float>(Stager &, const rrtmgp_atmos_state *, bool, bool, bool, DevState<float> &, int64_t, StateRW, bool);
template int stage_state<double>(Stager &, const rrtmgp_atmos_state *, bool, bool, bool, DevState<double> &, int64_t, StateRW, bool);
template int stage_flux<float>(Stager &, const rrtmgp_flux_out *, const rrtmgp_solve_opts *, size_t, size_t, bool, DevFlux<float> &, size_t, int, const float *);
template int stage_flux<double>(Stager &, const rrtmgp_flux_out *, const rrtmgp_solve_opts *, size_t, size_t, bool, DevFlux<double> &, size_t, int, const double *);

int check_common(rrtmgp_workspace *ws, const rrtmgp_lookup *gas, int want_sw, const rrtmgp_lookup *cld,
                        const rrtmgp_lookup *aero, const rrtmgp_atmos_state *as) {
    RR_CHECK(ws && gas && as, "null argument");
    RR_CHECK(gas->kind == LK_GAS, "expected a gas lookup");
    RR_CHECK((want_sw ? gas->gas32.is_sw || gas->gas64.is_sw : !(gas->gas32.is_sw || gas->gas64.is_sw)),
             "longwave / shortwave lookup mismatch");
    RR_CHECK(gas->ftype == ws->ftype, "lookup and workspace precision differ");
    RR_CHECK(!cld || (cld->kind == LK_CLOUD && cld->ftype == ws->ftype), "bad cloud lookup");
    RR_CHECK(!aero || (aero->kind == LK_AEROSOL && aero->ftype == ws->ftype), "bad aerosol lookup");
    RR_CHECK(gas->device == ws->device && (!cld || cld->device == ws->device) && (!aero || aero->device == ws->device),
             "lookups and workspace live on different devices");
    RR_CHECK(as->ncol == ws->ncol && as->nlay == ws->nlay, "state dimensions differ from the workspace");
    RR_CHECK(as->ncol >= 1 && as->nlay >= 2, "need ncol >= 1 and nlay >= 2");
    RR_HIP(hipSetDevice(ws->device));
    return RRTMGP_OK;
}

// What a multi-device workspace can shard in one call (include/rrtmgp_hip.h): everything whose slowest dimension is ncol.
int check_multi(const rrtmgp_workspace *ws, int state_mem, int bcs_mem, const rrtmgp_flux_out *flux,
                       const rrtmgp_solve_opts *opts, const void *inc_flux) {
    if (ws->shards.size() <= 1) return RRTMGP_OK;
    bool one_device = true;
    for (auto *s : ws->shards) one_device = one_device && s->device == ws->shards[0]->device;
    const bool any_dev = state_mem == RRTMGP_MEM_DEVICE || bcs_mem == RRTMGP_MEM_DEVICE || (flux && flux->mem == RRTMGP_MEM_DEVICE) ||
                         (opts && opts->metric_scaling && opts->metric_mem == RRTMGP_MEM_DEVICE);
    if (any_dev && !one_device) return set_error(RRTMGP_EINVAL, "a workspace spanning several devices needs host arrays");
    return RRTMGP_OK;
}

bool host_pipeline_applies(const rrtmgp_atmos_state *as, int bcs_mem, const rrtmgp_flux_out *flux,
                                  const rrtmgp_solve_opts *opts) {
    static const bool off = getenv("RRTMGP_HIP_NO_HOST_PIPELINE") != nullptr;
    if (off || !as || !flux) return false;
    if (as->mem != RRTMGP_MEM_HOST || bcs_mem != RRTMGP_MEM_HOST || flux->mem != RRTMGP_MEM_HOST) return false;
    if (opts && opts->metric_scaling && opts->metric_mem != RRTMGP_MEM_HOST) return false;
    return as->ncol >= 16384;
}

// Streams are created only by the path that uses them: the runtime multiplexes a process's streams onto a few hardware
// queues (4 by default), and a stream that merely EXISTS can put a workspace's copy stream on the queue of another
// workspace's compute stream — the two-solve host leg (two workspaces) fell from 38 to 42-45 ms when every pipelined
// workspace also owned the second compute lane of the short Layer-2 step (tools/experiments/host_regress_ab.sh).
int fork_join_events(rrtmgp_workspace *ws) {
    for (int i = 0; i < 3; i++)
        if (!ws->ev_k[i]) RR_HIP(hipEventCreateWithFlags(&ws->ev_k[i], hipEventDisableTiming));
    return RRTMGP_OK;
}
int pipeline_resources(rrtmgp_workspace *ws) {
    if (ws->copy_stream) return RRTMGP_OK;
    RR_HIP(hipStreamCreateWithFlags(&ws->copy_stream, hipStreamNonBlocking));
    for (int i = 0; i < 2; i++) RR_HIP(hipEventCreateWithFlags(&ws->ev_in[i], hipEventDisableTiming));
    TRY(fork_join_events(ws));
    ws->stage_alt.resize(ws->stage.size());
    return RRTMGP_OK;
}
// second compute lane of a short Layer-2 step (step_t)
int lane_resources(rrtmgp_workspace *ws) {
    if (ws->alt_stream) return RRTMGP_OK;
    RR_HIP(hipStreamCreateWithFlags(&ws->alt_stream, hipStreamNonBlocking));
    return fork_join_events(ws);
}

// Below this many staged host bytes a solve goes through the bounce buffer (Stager::packed): per-array DMA commands
// cost ~15 us each whatever their size, a host memcpy ~0.1 us per KB (RRTMGP_HIP_HOST_PACK_BYTES overrides, 0 = never).
size_t host_pack_max() {
    static const size_t v = getenv("RRTMGP_HIP_HOST_PACK_BYTES") ? (size_t)atoll(getenv("RRTMGP_HIP_HOST_PACK_BYTES")) : (size_t)2 << 20;
    return v;
}

}  // namespace rrtmgp
