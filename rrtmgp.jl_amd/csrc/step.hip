// step.hip — the whole radiation step, update_fluxes!(s) (src/api/update_fluxes.jl:223-233), as ONE call: the state staged
// once, [prepare kernel] -> LW -> SW -> net sums on the workspace stream (short steps on two lanes), fluxes home.
#include <climits>

#include "host.h"
#include "device.h"

namespace rrtmgp {

// ---- the whole radiation step: update_fluxes!(s) (src/api/update_fluxes.jl:223-233) in one call ---------------------------
// net_flux (nlev, ncol) = lw_net + sw_net, whatever the layout of the two (transpose_sum_into!, Fluxes.jl:407-424)
template <typename FT>
__global__ void net_sum_kernel(const FT *a, const FT *b, FT *out, int ncol, int nlev, int layout, int lda, int ldb) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)ncol * nlev) return;
    const size_t col = i / nlev, lev = i - col * nlev;
    const size_t ia = layout == RRTMGP_LAYOUT_NCOL_NLEV ? col + (size_t)lda * lev : i;
    const size_t ib = layout == RRTMGP_LAYOUT_NCOL_NLEV ? col + (size_t)ldb * lev : i;
    out[i] = a[ia] + b[ib];
}
template <typename FT>
static int launch_net_sum(rrtmgp_workspace *ws, const FT *a, const FT *b, FT *out, size_t ncol, size_t nlev, int layout, int lda, int ldb,
                          hipStream_t stream = nullptr) {
    const size_t n = ncol * nlev;
    hipLaunchKernelGGL(net_sum_kernel<FT>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream ? stream : ws->stream, a, b, out,
                       (int)ncol, (int)nlev, layout, lda, ldb);
    RR_HIP(hipGetLastError());
    return RRTMGP_OK;
}

template <typename FT>
struct StepLookups {
    const DevGas<FT> *lw, *sw;
    const DevCld<FT> *lw_cld, *sw_cld;
    const DevAero<FT> *lw_aero, *sw_aero;
    int lw_max_int, sw_max_int;
};

// Everything one step (or one chunk of it) has on the device once its arguments are staged
template <typename FT>
struct StepStaged {
    DevState<FT> ds, ds_sw;          // the state as the LW solve sees it; the same with what only the SW solve writes
    PrepView<FT> pv{};               // ... and what only the preparation touches
    const FT *emis = nullptr, *inc = nullptr, *mu0 = nullptr, *toa = nullptr, *adir = nullptr, *adif = nullptr;
    int inc_ld = 0;
    DevFlux<FT> fl_lw, fl_sw;
    FT *net = nullptr, *clear_net = nullptr;
    bool prep = false, use_aero = false, twostream_lw = true, band_lw = false, band_sw = false, diag_lw = false, diag_sw = false;
    int n_angles = 1;
};

// ---- the state, once: as the LW solve stages it (t_lev included; cld_cover = the LW cover), plus the SW-only outputs and
//      the arrays only the preparation reads or writes
template <typename FT>
static int step_stage_state(Stager &st, const StepLookups<FT> &L, const rrtmgp_update_fluxes_args *a, StepStaged<FT> &S) {
    const rrtmgp_atmos_state *as = a->as;
    const rrtmgp_prepare_opts *po = a->prepare;
    const size_t E = sizeof(FT), ncol = as->ncol, nlay = as->nlay, nlev = nlay + 1;
    const bool use_cld = L.lw_cld || L.sw_cld, use_aero = L.lw_aero || L.sw_aero;
    const bool prep = po != nullptr;
    const bool iso = prep && (po->steps & RRTMGP_PREP_ISOTHERMAL) && po->isothermal_boundary_layer;
    const int mem = as->mem;
    S.prep = prep; S.use_aero = use_aero;
    DevState<FT> &ds = S.ds;
    const int64_t nrgh = std::min<int64_t>(L.lw_cld ? L.lw_cld->nrghice : INT32_MAX, L.sw_cld ? L.sw_cld->nrghice : INT32_MAX);
    rrtmgp_atmos_state as_lw = *as;
    if (!L.lw_cld) as_lw.cld_cover_lw = nullptr;   // a LW solve without clouds writes no cover: nothing to bring back
    // The isothermal boundary layer fills the extra layer of EVERY cloud / aerosol array the state carries, whatever the
    // radiation method reads (prepare_t does; AllSkyRadiation with aerosol_radiation = false on a state with an
    // AerosolState): stage them for the preparation even when no lookup asks for them.
    const bool st_cld = use_cld || (iso && as->cld_frac), st_aero = use_aero || (iso && as->aero_mass);
    TRY(stage_state(st, &as_lw, st_cld, st_aero, true, ds, use_cld ? nrgh : 0, StateRW{prep, iso}, use_aero));
    S.ds_sw = ds;
    S.ds_sw.cld_cover = nullptr;
    if (L.sw_cld) TRY(st.out(mem, S_CLD_COVER2, as->cld_cover_sw, ncol * E, (void **)&S.ds_sw.cld_cover));
    if (L.sw_aero) {
        RR_CHECK((as->aod_sw_ext == nullptr) == (as->aod_sw_sca == nullptr), "aod_sw_ext and aod_sw_sca go together");
        TRY(st.out(mem, S_AOD_EXT, as->aod_sw_ext, ncol * E, (void **)&S.ds_sw.aod_sw_ext));
        TRY(st.out(mem, S_AOD_SCA, as->aod_sw_sca, ncol * E, (void **)&S.ds_sw.aod_sw_sca));
    }
    if (!prep) return RRTMGP_OK;
    PrepView<FT> &pv = S.pv;
    RR_CHECK(a->params, "update_fluxes: `params` is required with `prepare`");
    RR_CHECK(as->p_lev, "prepare_atmosphere: p_lev is required");
    pv.ncol = (int)ncol; pv.nlay = (int)nlay; pv.ls = 4;
    FT *ld = const_cast<FT *>(ds.layerdata);
    pv.col_dry = ld; pv.p_lay = ld + 1; pv.t_lay = ld + 2; pv.rel_hum = ld + 3;
    TRY(st.inout(mem, S_PLEV, as->p_lev, nlev * ncol * E, (void **)&pv.p_lev));
    pv.t_lev = const_cast<FT *>(ds.t_lev);
    pv.t_sfc = ds.t_sfc;
    TRY(st.in(mem, S_LAT, as->lat, ncol * E, (const void **)&pv.lat));
    if (as->vmr_kind == RRTMGP_VMR_GM) {
        pv.vmr_h2o = const_cast<FT *>(ds.vmr_h2o); pv.vmr_o3 = const_cast<FT *>(ds.vmr_o3); pv.hs = 1;
    } else {
        RR_CHECK(po->idx_h2o >= 1 && po->idx_h2o <= as->ngas, "Vmr: idx_h2o out of range");
        pv.vmr_full = const_cast<FT *>(ds.vmr);
        pv.ngas = (int)as->ngas; pv.hs = (int)as->ngas; pv.vmr_h2o = pv.vmr_full + (po->idx_h2o - 1);
    }
    if (iso) {
        if (st_cld) {
            pv.cld[0] = const_cast<FT *>(ds.cld_r_eff_liq); pv.cld[1] = const_cast<FT *>(ds.cld_r_eff_ice);
            pv.cld[2] = const_cast<FT *>(ds.cld_path_liq); pv.cld[3] = const_cast<FT *>(ds.cld_path_ice);
            pv.cld[4] = const_cast<FT *>(ds.cld_frac);
        }
        if (st_aero) { pv.aero[0] = const_cast<FT *>(ds.aero_size); pv.aero[1] = const_cast<FT *>(ds.aero_mass); }
    }
    TRY(st.in(po->z_mem, S_ZC, po->center_z, nlay * ncol * E, (const void **)&pv.center_z));
    TRY(st.in(po->z_mem, S_ZF, po->face_z, nlev * ncol * E, (const void **)&pv.face_z));
    return RRTMGP_OK;
}

// ---- boundary conditions of both solvers and every output: the two flux sets side by side, the net sums
template <typename FT>
static int step_stage_bcs_and_fluxes(Stager &st, const StepLookups<FT> &L, const rrtmgp_update_fluxes_args *a, StepStaged<FT> &S) {
    const size_t E = sizeof(FT), ncol = a->as->ncol, nlev = (size_t)a->as->nlay + 1;
    const rrtmgp_lw_bcs *bl = a->bcs_lw;
    const rrtmgp_sw_bcs *bs = a->bcs_sw;
    TRY(st.in(bl->mem, S_LW_BC0, bl->sfc_emis, (size_t)L.lw->n_bnd * ncol * E, (const void **)&S.emis));
    const size_t inc_ld_in = bl->inc_flux_ld > 0 ? (size_t)bl->inc_flux_ld : ncol;
    RR_CHECK(!bl->inc_flux || inc_ld_in >= ncol, "LwBCs.inc_flux_ld is smaller than ncol");
    S.inc_ld = (int)ncol;
    if (bl->inc_flux && inc_ld_in != ncol && bl->mem == RRTMGP_MEM_HOST) {
        TRY(st.in2d(S_LW_BC1, bl->inc_flux, ncol * E, (size_t)L.lw->n_gpt, inc_ld_in * E, (const void **)&S.inc));
    } else {
        if (bl->inc_flux && bl->mem == RRTMGP_MEM_DEVICE) S.inc_ld = (int)inc_ld_in;
        TRY(st.in(bl->mem, S_LW_BC1, bl->inc_flux, (size_t)L.lw->n_gpt * ncol * E, (const void **)&S.inc));
    }
    TRY(st.in(bs->mem, S_BC0, bs->cos_zenith, ncol * E, (const void **)&S.mu0));
    TRY(st.in(bs->mem, S_BC1, bs->toa_flux, ncol * E, (const void **)&S.toa));
    TRY(st.in(bs->mem, S_BC2, bs->sfc_alb_direct, (size_t)L.sw->n_bnd * ncol * E, (const void **)&S.adir));
    TRY(st.in(bs->mem, S_BC3, bs->sfc_alb_diffuse, (size_t)L.sw->n_bnd * ncol * E, (const void **)&S.adif));

    TRY(stage_flux(st, a->flux_lw, a->opts, ncol, nlev, false, S.fl_lw, S.twostream_lw ? (size_t)L.lw->n_bnd : (a->flux_lw->clear_flux_up ? 1 : 0)));
    TRY(stage_flux(st, a->flux_sw, a->opts, ncol, nlev, true, S.fl_sw, (size_t)L.sw->n_bnd, S_X_FLUX_UP - S_FLUX_UP, S.fl_lw.metric));
    if (!S.fl_lw.metric) S.fl_sw.metric = nullptr;
    // (decided from the caller's descriptors: in the registration pass the staged pointers are all null)
    S.band_lw = a->flux_lw->band_flux_up != nullptr; S.band_sw = a->flux_sw->band_flux_up != nullptr;
    RR_CHECK(S.twostream_lw || !S.band_lw, "per-band fluxes are only available from the two-stream solvers");
    S.diag_lw = a->flux_lw->clear_flux_up != nullptr; S.diag_sw = a->flux_sw->clear_flux_up != nullptr;
    RR_CHECK(!(S.diag_lw || S.diag_sw) || (L.lw_cld && L.sw_cld), "the clear-sky diagnostic needs the cloud lookups (AllSkyRadiationWithClearSkyDiagnostics)");
    RR_CHECK(!a->clear_net_flux || (S.diag_lw && S.diag_sw), "clear_net_flux needs the clear-sky fluxes of both solvers");
    TRY(st.out(a->flux_lw->mem, S_NET, a->net_flux, ncol * nlev * E, (void **)&S.net));
    TRY(st.out(a->flux_lw->mem, S_CLR_NETSUM, a->clear_net_flux, ncol * nlev * E, (void **)&S.clear_net));
    return RRTMGP_OK;
}

// ---- [prepare] -> LW -> SW -> net on the workspace stream.  SHORT steps run their SW kernels on the workspace's second lane
// instead (own stream, own sweep scratch; forked after the preparation, joined before the net sums).  When that pays was swept
// in round 5 against f = columns / resident workgroup slots (4 per CU), one lane vs two, 72 layers with aerosols and 64 layers
// without (tools/experiments/overlap_range_sweep.sh, profiles/r05_overlap_range_sweep.txt):
//   f <= 0.65       the two grids fit the chip side by side:                         -22 ... -26 % in both configurations
//   0.7 <= f <= 1   the second grid only gets in the first one's way:                 +6 ... +11 %
//   1.05 < f < 1.5  the SW workgroups start in the slots the LW tail frees:           -8 ... -11 % (aerosols), -1 ... -2 %
//   2.1 < f <= 6    the same, smaller:  -1 ... -4 % with aerosols (BASELINE config 4 is f = 4), +0.5 ... +2 % without
//   beyond          nothing either way (+-0.2 %).
// (Round 4's rule, from five points: f <= 0.5 or 4 <= f <= 12.)  RRTMGP_HIP_STEP_OVERLAP=0/1 forces it off / on.
template <typename FT>
static int step_launch(rrtmgp_workspace *ws, const StepLookups<FT> &L, const rrtmgp_update_fluxes_args *a, StepStaged<FT> &S,
                       bool pipelined_chunk) {
    const size_t ncol = a->as->ncol, nlev = (size_t)a->as->nlay + 1;
    static const int force_overlap = getenv("RRTMGP_HIP_STEP_OVERLAP") ? atoi(getenv("RRTMGP_HIP_STEP_OVERLAP")) : -1;
    const double f_slots = (double)ncol / (4.0 * (double)ws->n_cu);
    const bool short_step = f_slots <= 0.65 || (f_slots > 1.05 && f_slots < 1.5) || (S.use_aero && f_slots > 2.1 && f_slots <= 6.0);
    const bool overlap = !pipelined_chunk && (force_overlap >= 0 ? force_overlap != 0 : short_step);
    if (overlap) TRY(lane_resources(ws));
    if (S.prep) TRY(launch_prepare<FT>(ws, S.pv, *a->params, *a->prepare, false));
    const uint64_t seed = a->opts ? a->opts->seed : 0;
    const int64_t coff = a->opts ? a->opts->col_offset : 0;
    // the clear-sky fluxes ride in the same launch when the solver is two-stream and no per-band fluxes are kept;
    // otherwise the cloudless solve runs first on the same staged state (update_fluxes.jl:39-65, :101-128)
    auto clear_first = [](DevFlux<FT> &f) {
        DevFlux<FT> c = f;
        c.up = f.clear_up; c.dn = f.clear_dn; c.net = f.clear_net; c.dir = f.clear_dir;
        c.band_up = c.band_dn = c.band_net = nullptr;
        c.clear_up = c.clear_dn = c.clear_net = c.clear_dir = nullptr;
        f.clear_up = f.clear_dn = f.clear_net = f.clear_dir = nullptr;
        return c;
    };
    DevFlux<FT> &fl_lw = S.fl_lw, &fl_sw = S.fl_sw;
    FT *lw_clear_net = fl_lw.clear_net, *sw_clear_net = fl_sw.clear_net;
    auto lw_lane = [&]() -> int {
        if (S.diag_lw && (!S.twostream_lw || S.band_lw)) {
            const DevFlux<FT> c = clear_first(fl_lw);
            TRY(launch_lw<FT>(ws, S.twostream_lw, *L.lw, nullptr, L.lw_aero, S.ds, S.emis, S.inc, S.inc_ld, c, S.n_angles, seed, coff, L.lw_max_int));
        }
        return launch_lw<FT>(ws, S.twostream_lw, *L.lw, L.lw_cld, L.lw_aero, S.ds, S.emis, S.inc, S.inc_ld, fl_lw, S.n_angles, seed, coff, L.lw_max_int);
    };
    auto sw_lane = [&](const Lane *lane) -> int {
        if (S.diag_sw && S.band_sw) {
            const DevFlux<FT> c = clear_first(fl_sw);
            DevState<FT> dc = S.ds_sw;
            dc.aod_sw_ext = dc.aod_sw_sca = nullptr;   // the all-sky solve writes the same values
            TRY(launch_sw<FT>(ws, 1, *L.sw, nullptr, L.sw_aero, dc, S.mu0, S.toa, S.adir, S.adif, c, seed, coff, L.sw_max_int, lane));
        }
        return launch_sw<FT>(ws, 1, *L.sw, L.sw_cld, L.sw_aero, S.ds_sw, S.mu0, S.toa, S.adir, S.adif, fl_sw, seed, coff, L.sw_max_int, lane);
    };
    auto net_sums = [&](hipStream_t stream) -> int {
        if (S.net) TRY(launch_net_sum<FT>(ws, fl_lw.net, fl_sw.net, S.net, ncol, nlev, fl_lw.layout, fl_lw.ld, fl_sw.ld, stream));
        if (S.clear_net) TRY(launch_net_sum<FT>(ws, lw_clear_net, sw_clear_net, S.clear_net, ncol, nlev, fl_lw.layout, fl_lw.ld, fl_sw.ld, stream));
        return RRTMGP_OK;
    };
    if (!overlap) {
        TRY(lw_lane());
        TRY(sw_lane(nullptr));
        return net_sums(nullptr);
    }
    // Two lanes:   main lane  : record fork, LW kernel(s) ................. [wait join] net sums
    //              second lane: [wait fork] SW kernel(s), record join
    // The fork sits behind everything queued so far (uploads, preparation).  No memset is queued in front of either kernel
    // (queue_release): 512 columns x 72 with aerosols 226 -> 220 us per step.  The lane is handed to the launches
    // explicitly: the workspace's fields stay what they are.  Two other orders were measured from the GPU-side timeline of
    // that step (tools/experiments/small_step_timeline.sh, small_step_ab.sh, profiles/r05_small_step_ab.txt) and are kept
    // behind RRTMGP_HIP_STEP_ORDER for A/B: 1 = net sums on the second lane behind SW, waiting only for an "LW done" event
    // (one exposed cross-stream wait instead of two: 230 us, worse), 2 = also SW, the longer kernel, queued first (226 us).
    const Lane second{ws->alt_stream, &ws->alt_scratch, 1};
    static const int order = getenv("RRTMGP_HIP_STEP_ORDER") ? atoi(getenv("RRTMGP_HIP_STEP_ORDER")) : 0;   // A/B switch
    RR_HIP(hipEventRecord(ws->ev_k[0], ws->stream));   // fork
    RR_HIP(hipStreamWaitEvent(second.stream, ws->ev_k[0], 0));
    // From here on the second lane may hold work of this step: whatever fails below, the main lane still waits for it (ADVICE
    // r5: an early return used to leave the lane un-joined, and the caller's next call or finish() raced with it).
    auto join = [&]() -> int {
        RR_HIP(hipEventRecord(ws->ev_k[1], second.stream));
        RR_HIP(hipStreamWaitEvent(ws->stream, ws->ev_k[1], 0));
        return RRTMGP_OK;
    };
    int rc = RRTMGP_OK;
    if (order == 0) {          // (shipped) LW queued first, net sums on the main lane behind the join
        rc = lw_lane();
        if (rc == RRTMGP_OK) rc = sw_lane(&second);
        const int rj = join();
        if (rc == RRTMGP_OK) rc = rj;
        return rc != RRTMGP_OK ? rc : net_sums(nullptr);
    }
    if (order == 1) { rc = lw_lane(); if (rc == RRTMGP_OK) rc = sw_lane(&second); }   // LW first, net sums on the second lane
    else { rc = sw_lane(&second); if (rc == RRTMGP_OK) rc = lw_lane(); }
    if (rc == RRTMGP_OK && hipEventRecord(ws->ev_k[2], ws->stream) != hipSuccess) rc = set_error(RRTMGP_EHIP, "hipEventRecord");   // LW done
    if (rc == RRTMGP_OK && hipStreamWaitEvent(second.stream, ws->ev_k[2], 0) != hipSuccess) rc = set_error(RRTMGP_EHIP, "hipStreamWaitEvent");
    if (rc == RRTMGP_OK) rc = net_sums(second.stream);
    const int rj = join();
    return rc != RRTMGP_OK ? rc : rj;
}

// One chunk (or the whole batch) of the step: stage everything once, [prepare] -> LW -> SW -> net sums, copy back.
template <typename FT>
static int step_t(rrtmgp_workspace *ws, const StepLookups<FT> &L, const rrtmgp_update_fluxes_args *a, Stager *chunk = nullptr) {
    const rrtmgp_solve_opts *opts = a->opts;
    StepStaged<FT> S;
    S.twostream_lw = a->lw_solver == RRTMGP_LW_TWOSTREAM;
    S.n_angles = opts ? opts->n_gauss_angles : 1;
    RR_CHECK(S.twostream_lw || (S.n_angles >= 1 && S.n_angles <= 4), "n_gauss_angles must be 1..4");
    RR_CHECK(a->bcs_lw->sfc_emis, "LwBCs: sfc_emis is required");
    RR_CHECK(a->bcs_sw->cos_zenith && a->bcs_sw->toa_flux && a->bcs_sw->sfc_alb_direct && a->bcs_sw->sfc_alb_diffuse,
             "SwBCs: cos_zenith, toa_flux and the surface albedos are required");
    RR_CHECK(a->flux_sw->flux_dn_dir, "FluxSW: flux_dn_dir is required");
    RR_CHECK((!L.lw_cld || L.lw_cld->nband == L.lw->n_bnd) && (!L.sw_cld || L.sw_cld->nband == L.sw->n_bnd),
             "cloud lookup band count differs from the gas lookup");
    RR_CHECK((!L.lw_aero || L.lw_aero->nband == L.lw->n_bnd) && (!L.sw_aero || L.sw_aero->nband == L.sw->n_bnd),
             "aerosol lookup band count differs from the gas lookup");
    RR_CHECK(a->flux_lw->layout == a->flux_sw->layout, "flux_lw and flux_sw must share one layout");
    Stager own{ws, {}};
    Stager &st = chunk ? *chunk : own;
    TRY(step_stage_state(st, L, a, S));
    TRY(step_stage_bcs_and_fluxes(st, L, a, S));
    if (st.pin_only) return RRTMGP_OK;
    const bool pipelined_chunk = chunk && !st.packed;
    if (pipelined_chunk) {  // pipelined host path: the uploads ran on the copy stream
        RR_HIP(hipEventRecord(ws->ev_in[0], st.copy_stream()));
        RR_HIP(hipStreamWaitEvent(ws->stream, ws->ev_in[0], 0));
    }
    TRY(st.flush());  // packed small step: the one upload
    TRY(step_launch(ws, L, a, S, pipelined_chunk));
    return pipelined_chunk ? RRTMGP_OK : st.finish();
}

// The column range [sl.c0, sl.c0 + nc) of every array of a step
struct StepSlice {
    rrtmgp_update_fluxes_args a;
    rrtmgp_atmos_state as;
    rrtmgp_lw_bcs bl;
    rrtmgp_sw_bcs bs;
    rrtmgp_flux_out fl, fs;
    rrtmgp_solve_opts o;
    rrtmgp_prepare_opts po;
    StepSlice(const rrtmgp_update_fluxes_args *src, const ColumnSlice &sl, size_t nc, size_t nb_lw, size_t nb_sw) {
        a = *src; as = *src->as; bl = *src->bcs_lw; bs = *src->bcs_sw; fl = *src->flux_lw; fs = *src->flux_sw;
        o = rrtmgp_solve_opts{};
        if (src->opts) o = *src->opts; else o.n_gauss_angles = 1;
        const size_t ncol = (size_t)src->as->ncol, nlay = (size_t)src->as->nlay, nlev = nlay + 1;
        slice_state(as, sl, nc);
        slice_lw_bcs(bl, sl, nb_lw, ncol);
        slice_sw_bcs(bs, sl, nb_sw);
        slice_flux_arrays(fl, sl, nlev, ncol);
        slice_flux_arrays(fs, sl, nlev, ncol);
        slice_opts(o, sl, nlev);
        a.net_flux = sl.adv(a.net_flux, nlev); a.clear_net_flux = sl.adv(a.clear_net_flux, nlev);
        if (src->prepare) {
            po = *src->prepare;
            po.center_z = sl.adv(po.center_z, nlay); po.face_z = sl.adv(po.face_z, nlev);
            a.prepare = &po;
        }
        a.as = &as; a.bcs_lw = &bl; a.bcs_sw = &bs; a.flux_lw = &fl; a.flux_sw = &fs; a.opts = &o;
    }
    StepSlice(const StepSlice &) = delete;
};

template <typename FT>
static int step_host(rrtmgp_workspace *ws, const StepLookups<FT> &L, const rrtmgp_update_fluxes_args *a) {
    size_t need = 0;
    PinScope scope{ws};
    {   // registration pass over the caller's whole host arrays
        host_pin_begin(ws);
        Stager pin{ws, {}};
        pin.pin_only = true;
        TRY(step_t<FT>(ws, L, a, &pin));
        need = pin.need;
    }
    const rrtmgp_solve_opts *o = a->opts;
    const bool all_host = a->as->mem == RRTMGP_MEM_HOST && a->bcs_lw->mem == RRTMGP_MEM_HOST && a->bcs_sw->mem == RRTMGP_MEM_HOST &&
                          a->flux_lw->mem == RRTMGP_MEM_HOST && a->flux_sw->mem == RRTMGP_MEM_HOST &&
                          (!o || !o->metric_scaling || o->metric_mem == RRTMGP_MEM_HOST) &&
                          (!a->prepare || (!a->prepare->center_z && !a->prepare->face_z) || a->prepare->z_mem == RRTMGP_MEM_HOST);
    static const bool no_pipe = getenv("RRTMGP_HIP_NO_HOST_PIPELINE") != nullptr;
    if (no_pipe || !all_host || a->as->ncol < 16384) {
        if (need && need <= host_pack_max()) {   // small step: one bounce buffer, one DMA each way
            TRY(bounce_ensure(ws, need));
            Stager st{ws, {}};
            st.packed = true;
            return step_t<FT>(ws, L, a, &st);
        }
        return step_t<FT>(ws, L, a);
    }
    return run_column_pipeline(ws, (size_t)a->as->ncol, sizeof(FT), a->as->vmr_kind == RRTMGP_VMR_GM,
                               [&](const ColumnSlice &sl, size_t nc, Stager &st) {
                                   StepSlice c(a, sl, nc, (size_t)L.lw->n_bnd, (size_t)L.sw->n_bnd);
                                   return step_t<FT>(ws, L, &c.a, &st);
                               });
}

// ---- the gray step: update_fluxes!(s) for GrayRadiation in one call (include/rrtmgp_hip.h) ------------------------------
// The gray state crosses once ([prepare] reads and writes p / T of layers and levels), then gray LW -> gray SW -> net sum on
// the workspace stream.  Same launches, in the same order, as the separate entry points: same bits.
template <typename FT>
static int step_gray_t(rrtmgp_workspace *ws, const rrtmgp_update_fluxes_gray_args *a) {
    const rrtmgp_gray_state *gs = a->as;
    const rrtmgp_lw_bcs *bl = a->bcs_lw;
    const rrtmgp_sw_bcs *bs = a->bcs_sw;
    const rrtmgp_prepare_opts *po = a->prepare;
    const size_t E = sizeof(FT), ncol = gs->ncol, nlay = gs->nlay, nlev = nlay + 1;
    const bool prep = po != nullptr;
    const int twostream_lw = a->lw_solver == RRTMGP_LW_TWOSTREAM, twostream_sw = a->sw_twostream != 0;
    RR_CHECK(!a->opts || a->opts->n_gauss_angles <= 1, "gray radiation is solved with a single quadrature angle");
    RR_CHECK(gs->lat && gs->p_lay && gs->p_lev && gs->t_lay && gs->t_lev && gs->t_sfc && bl->sfc_emis, "gray LW: missing array");
    RR_CHECK(bs->cos_zenith && bs->toa_flux && a->flux_sw->flux_dn_dir, "gray SW: missing array");
    RR_CHECK(!twostream_sw || (bs->sfc_alb_direct && bs->sfc_alb_diffuse), "gray SW two-stream: surface albedos are required");
    RR_CHECK(a->flux_lw->layout == a->flux_sw->layout, "flux_lw and flux_sw must share one layout");
    RR_CHECK(!prep || a->params, "update_fluxes (gray): `params` is required with `prepare`");
    Stager st{ws, {}};
    const int mem = gs->mem;
    const FT *lat, *p_lay, *p_lev, *t_lay, *t_lev, *t_sfc, *emis, *inc, *mu0, *toa, *adir, *adif;
    TRY(st.in(mem, S_LAT, gs->lat, ncol * E, (const void **)&lat));
    TRY(st.io(prep, mem, S_PLAY, gs->p_lay, nlay * ncol * E, (const void **)&p_lay));
    TRY(st.io(prep, mem, S_PLEV, gs->p_lev, nlev * ncol * E, (const void **)&p_lev));
    TRY(st.io(prep, mem, S_TLAY, gs->t_lay, nlay * ncol * E, (const void **)&t_lay));
    TRY(st.io(prep, mem, S_TLEV, gs->t_lev, nlev * ncol * E, (const void **)&t_lev));
    TRY(st.in(mem, S_TSFC, gs->t_sfc, ncol * E, (const void **)&t_sfc));
    PrepView<FT> pv{};
    if (prep) {
        pv.ncol = (int)ncol; pv.nlay = (int)nlay; pv.ls = 1;
        pv.p_lay = const_cast<FT *>(p_lay); pv.t_lay = const_cast<FT *>(t_lay);
        pv.p_lev = const_cast<FT *>(p_lev); pv.t_lev = const_cast<FT *>(t_lev);
        pv.t_sfc = t_sfc;
        TRY(st.in(po->z_mem, S_ZC, po->center_z, nlay * ncol * E, (const void **)&pv.center_z));
        TRY(st.in(po->z_mem, S_ZF, po->face_z, nlev * ncol * E, (const void **)&pv.face_z));
    }
    TRY(st.in(bl->mem, S_LW_BC0, bl->sfc_emis, ncol * E, (const void **)&emis));
    TRY(st.in(bl->mem, S_LW_BC1, bl->inc_flux, ncol * E, (const void **)&inc));
    TRY(st.in(bs->mem, S_BC0, bs->cos_zenith, ncol * E, (const void **)&mu0));
    TRY(st.in(bs->mem, S_BC1, bs->toa_flux, ncol * E, (const void **)&toa));
    TRY(st.in(bs->mem, S_BC2, bs->sfc_alb_direct, ncol * E, (const void **)&adir));
    TRY(st.in(bs->mem, S_BC3, bs->sfc_alb_diffuse, ncol * E, (const void **)&adif));
    DevFlux<FT> fl_lw, fl_sw;
    TRY(stage_flux(st, a->flux_lw, a->opts, ncol, nlev, false, fl_lw));
    TRY(stage_flux(st, a->flux_sw, a->opts, ncol, nlev, true, fl_sw, 0, S_X_FLUX_UP - S_FLUX_UP, fl_lw.metric));
    if (!fl_lw.metric) fl_sw.metric = nullptr;
    FT *net = nullptr;
    TRY(st.out(a->flux_lw->mem, S_NET, a->net_flux, ncol * nlev * E, (void **)&net));
    if (prep) {
        rrtmgp_prepare_opts og = *po;
        og.steps &= ~RRTMGP_PREP_COL_DRY;   // a gray state has no col_dry (rrtmgp_hip_prepare_atmosphere_gray)
        TRY(launch_prepare<FT>(ws, pv, *a->params, og, true));
    }
    GrayArgs ga;
    ga.otp_kind = gs->otp_kind;
    for (int i = 0; i < 5; i++) ga.otp[i] = gs->otp[i];
    ga.stefan = gs->stefan;
    TRY(launch_gray_lw<FT>(ws, twostream_lw, (int)ncol, (int)nlay, ga, lat, p_lay, p_lev, t_lay, t_lev, t_sfc, emis, inc, fl_lw));
    TRY(launch_gray_sw<FT>(ws, twostream_sw, (int)ncol, (int)nlay, ga, p_lay, p_lev, mu0, toa, adir, adif, fl_sw));
    if (net) TRY(launch_net_sum<FT>(ws, fl_lw.net, fl_sw.net, net, ncol, nlev, fl_lw.layout, fl_lw.ld, fl_sw.ld));
    return st.finish();
}

}  // namespace rrtmgp

using namespace rrtmgp;

// update_fluxes!(s::RRTMGPSolver, seedval)  src/api/update_fluxes.jl:223-233
static int step_dispatch(rrtmgp_workspace *ws, const rrtmgp_update_fluxes_args *a) {
    if (ws->ftype == RRTMGP_F32) {
        const StepLookups<float> L{&a->lookup_lw->gas32, &a->lookup_sw->gas32,
                                   a->lookup_lw_cld ? &a->lookup_lw_cld->cld32 : nullptr, a->lookup_sw_cld ? &a->lookup_sw_cld->cld32 : nullptr,
                                   a->lookup_lw_aero ? &a->lookup_lw_aero->aero32 : nullptr, a->lookup_sw_aero ? &a->lookup_sw_aero->aero32 : nullptr,
                                   a->lookup_lw->max_int, a->lookup_sw->max_int};
        return step_host<float>(ws, L, a);
    }
    const StepLookups<double> L{&a->lookup_lw->gas64, &a->lookup_sw->gas64,
                                a->lookup_lw_cld ? &a->lookup_lw_cld->cld64 : nullptr, a->lookup_sw_cld ? &a->lookup_sw_cld->cld64 : nullptr,
                                a->lookup_lw_aero ? &a->lookup_lw_aero->aero64 : nullptr, a->lookup_sw_aero ? &a->lookup_sw_aero->aero64 : nullptr,
                                a->lookup_lw->max_int, a->lookup_sw->max_int};
    return step_host<double>(ws, L, a);
}

extern "C" {

int rrtmgp_hip_update_fluxes(rrtmgp_workspace *ws, const rrtmgp_update_fluxes_args *a) {
    RR_CHECK(ws && a, "null argument");
    RR_CHECK(a->lookup_lw && a->lookup_sw && a->as && a->bcs_lw && a->bcs_sw && a->flux_lw && a->flux_sw,
             "update_fluxes: lookups, state, boundary conditions and flux outputs are required");
    RR_CHECK(a->lw_solver == RRTMGP_LW_TWOSTREAM || a->lw_solver == RRTMGP_LW_NOSCAT, "lw_solver must be RRTMGP_LW_TWOSTREAM or RRTMGP_LW_NOSCAT");
    const rrtmgp_solve_opts *o = a->opts;
    if (!ws->shards.empty()) {
        RR_CHECK(a->lookup_lw->kind == LK_GAS && a->lookup_lw->ftype == ws->ftype && a->lookup_sw->kind == LK_GAS && a->lookup_sw->ftype == ws->ftype,
                 "expected gas lookups of the workspace's precision");
        RR_CHECK(a->as->ncol == ws->ncol && a->as->nlay == ws->nlay, "state dimensions differ from the workspace");
        TRY(check_multi(ws, a->as->mem, a->bcs_lw->mem, a->flux_lw, o, nullptr));
        TRY(check_multi(ws, a->as->mem, a->bcs_sw->mem, a->flux_sw, o, nullptr));
        const bool dev_arrays = a->as->mem == RRTMGP_MEM_DEVICE || a->bcs_lw->mem == RRTMGP_MEM_DEVICE || a->bcs_sw->mem == RRTMGP_MEM_DEVICE ||
                                a->flux_lw->mem == RRTMGP_MEM_DEVICE || a->flux_sw->mem == RRTMGP_MEM_DEVICE ||
                                (o && o->metric_scaling && o->metric_mem == RRTMGP_MEM_DEVICE);
        const size_t E = (size_t)ws->ftype, nb_lw = (size_t)n_bnd_of(ws, a->lookup_lw), nb_sw = (size_t)n_bnd_of(ws, a->lookup_sw);
        return multi_run(ws, [&](rrtmgp_workspace *sw, size_t c0, size_t nc) -> int {
            StepSlice c(a, ColumnSlice{E, c0}, nc, nb_lw, nb_sw);
            const rrtmgp_lookup **lk[6] = {&c.a.lookup_lw, &c.a.lookup_sw, &c.a.lookup_lw_cld, &c.a.lookup_sw_cld, &c.a.lookup_lw_aero, &c.a.lookup_sw_aero};
            for (auto *p : lk) {
                if (!*p) continue;
                *p = lookup_on(*p, sw->device);
                if (!*p) return set_error(RRTMGP_EINVAL, "a lookup has no replica on one of the workspace's devices (use *_lookup_create_multi)");
            }
            return rrtmgp_hip_update_fluxes(sw, &c.a);
        }, dev_arrays);
    }
    TRY(check_common(ws, a->lookup_lw, 0, a->lookup_lw_cld, a->lookup_lw_aero, a->as));
    TRY(check_common(ws, a->lookup_sw, 1, a->lookup_sw_cld, a->lookup_sw_aero, a->as));
    return step_dispatch(ws, a);
}

int rrtmgp_hip_update_fluxes_gray(rrtmgp_workspace *ws, const rrtmgp_update_fluxes_gray_args *a) {
    RR_CHECK(ws && a, "null argument");
    RR_CHECK(a->as && a->bcs_lw && a->bcs_sw && a->flux_lw && a->flux_sw, "update_fluxes (gray): state, boundary conditions and flux outputs are required");
    RR_CHECK(a->lw_solver == RRTMGP_LW_TWOSTREAM || a->lw_solver == RRTMGP_LW_NOSCAT, "lw_solver must be RRTMGP_LW_TWOSTREAM or RRTMGP_LW_NOSCAT");
    RR_CHECK(a->as->ncol == ws->ncol && a->as->nlay == ws->nlay, "state dimensions differ from the workspace");
    RR_CHECK(a->as->otp_kind == 0 || a->as->otp_kind == 1, "unknown gray optical-thickness kind");
    const rrtmgp_solve_opts *o = a->opts;
    if (!ws->shards.empty()) {
        TRY(check_multi(ws, a->as->mem, a->bcs_lw->mem, a->flux_lw, o, nullptr));
        TRY(check_multi(ws, a->as->mem, a->bcs_sw->mem, a->flux_sw, o, nullptr));
        const bool dev_arrays = a->as->mem == RRTMGP_MEM_DEVICE || a->bcs_lw->mem == RRTMGP_MEM_DEVICE || a->bcs_sw->mem == RRTMGP_MEM_DEVICE ||
                                a->flux_lw->mem == RRTMGP_MEM_DEVICE || a->flux_sw->mem == RRTMGP_MEM_DEVICE ||
                                (o && o->metric_scaling && o->metric_mem == RRTMGP_MEM_DEVICE);
        const size_t E = (size_t)ws->ftype, ncol = (size_t)a->as->ncol, nlay = (size_t)a->as->nlay, nlev = nlay + 1;
        return multi_run(ws, [&](rrtmgp_workspace *sw, size_t c0, size_t nc) -> int {
            const ColumnSlice sl{E, c0};
            rrtmgp_update_fluxes_gray_args c = *a;
            rrtmgp_gray_state g = *a->as;
            rrtmgp_lw_bcs bl = *a->bcs_lw;
            rrtmgp_sw_bcs bs = *a->bcs_sw;
            rrtmgp_flux_out fl = *a->flux_lw, fs = *a->flux_sw;
            rrtmgp_solve_opts so{};
            if (o) so = *o; else so.n_gauss_angles = 1;
            rrtmgp_prepare_opts po{};
            slice_gray(g, sl, nc);
            bl.sfc_emis = sl.adv(bl.sfc_emis, 1); bl.inc_flux = sl.adv(bl.inc_flux, 1);
            slice_sw_bcs(bs, sl, 1);
            slice_flux_arrays(fl, sl, nlev, ncol);
            slice_flux_arrays(fs, sl, nlev, ncol);
            slice_opts(so, sl, nlev);
            c.net_flux = sl.adv(c.net_flux, nlev);
            if (a->prepare) {
                po = *a->prepare;
                po.center_z = sl.adv(po.center_z, nlay); po.face_z = sl.adv(po.face_z, nlev);
                c.prepare = &po;
            }
            c.as = &g; c.bcs_lw = &bl; c.bcs_sw = &bs; c.flux_lw = &fl; c.flux_sw = &fs; c.opts = &so;
            return rrtmgp_hip_update_fluxes_gray(sw, &c);
        }, dev_arrays);
    }
    RR_HIP(hipSetDevice(ws->device));
    return ws->ftype == RRTMGP_F32 ? step_gray_t<float>(ws, a) : step_gray_t<double>(ws, a);
}

}  // extern "C"
