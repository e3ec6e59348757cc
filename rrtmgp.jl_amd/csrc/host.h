// host.h — what the host-side translation units of libhip_rrtmgp.so share (round 6: api.hip, one 2 500-line file, became
// runtime.hip / lookups.hip / staging.hip / solve.hip / views.hip / step.hip):
//   * the staging of caller arrays (Stager: host <-> HBM mirrors, page-lock registration pass, packed small solves),
//   * column slices of every ABI struct (shards of a multi-device workspace, chunks of the pipelined host path),
//   * the column pipeline itself (a template over the chunk body: the single solves and the fused step both run on it).
// Internal: nothing here is part of the C ABI (include/rrtmgp_hip.h).
#pragma once

#include <execinfo.h>
#include <signal.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <thread>

#include "common.h"

#define TRY(x)            \
    do {                  \
        int _rc = (x);    \
        if (_rc) return _rc; \
    } while (0)

namespace rrtmgp {

// runtime.hip
int select_device(int device);
int bounce_ensure(rrtmgp_workspace *ws, size_t bytes);
int host_register_explicit(void *ptr, size_t bytes);
int host_unregister_explicit(void *ptr);
int host_registered_count();

// ---- staging of host-memory arguments -------------------------------------------------
enum Slot {
    S_LAYERDATA = 0, S_TLEV, S_TSFC, S_VMR_H2O, S_VMR_O3, S_VMR, S_CLD_RL, S_CLD_RI, S_CLD_PL, S_CLD_PI, S_CLD_F,
    S_CLD_COVER, S_AERO_SIZE, S_AERO_MASS, S_AOD_EXT, S_AOD_SCA, S_BC0, S_BC1, S_BC2, S_BC3, S_FLUX_UP, S_FLUX_DN,
    S_FLUX_NET, S_FLUX_DIR, S_BAND_UP, S_BAND_DN, S_BAND_NET, S_CLR_UP, S_CLR_DN, S_CLR_NET, S_CLR_DIR, S_METRIC, S_PLEV, S_LAT, S_TLAY, S_PLAY, S_AUX0, S_AUX1, S_ZC, S_ZF,
    // the whole-step entry (rrtmgp_hip_update_fluxes) keeps the LW and the SW arrays of one step side by side: a second
    // block of flux slots in the order of the first (S_FLUX_UP .. S_CLR_DIR), the LW boundary conditions, the second
    // cloud cover and the two net-flux sums
    S_X_FLUX_UP, S_X_FLUX_DN, S_X_FLUX_NET, S_X_FLUX_DIR, S_X_BAND_UP, S_X_BAND_DN, S_X_BAND_NET, S_X_CLR_UP, S_X_CLR_DN,
    S_X_CLR_NET, S_X_CLR_DIR, S_LW_BC0, S_LW_BC1, S_CLD_COVER2, S_NET, S_CLR_NETSUM, S_NSLOTS
};
static_assert(S_X_CLR_DIR - S_X_FLUX_UP == S_CLR_DIR - S_FLUX_UP, "the second block of flux slots mirrors the first");

struct Stager {
    rrtmgp_workspace *ws;
    struct Back { void *host; void *dev; size_t bytes; size_t rows = 0, hpitch = 0; };  // rows > 0: `rows` pieces of `bytes`, hpitch apart at the host
    std::vector<Back> backs;
    hipStream_t cs = nullptr;  // stream of the copies; the workspace stream unless the pipelined host path says otherwise
    bool pin_only = false;     // registration pass over the caller's WHOLE host arrays: no copies, no device memory
    uint64_t keep = 0;         // pipelined host path, bit per slot: the array does not depend on the column range and an
                               // earlier chunk has already put it into this staging set
    // small solves (packed): host arrays are copied into / out of the workspace's page-locked bounce buffer by the CPU
    // and cross PCIe in ONE transfer each way; `need` = what the registration pass found the solve to stage
    bool packed = false;
    size_t need = 0, off = 0, in_hi = 0, out_lo = ~size_t(0), out_hi = 0;
    static size_t al(size_t n) { return (n + 255) & ~size_t(255); }
    hipStream_t copy_stream() const { return cs ? cs : ws->stream; }
    bool pin(int mem, const void *p, size_t bytes, void **out) {
        if (!pin_only) return false;
        if (mem == RRTMGP_MEM_HOST) { host_pin(ws, p, bytes); need += al(bytes); }
        *out = nullptr;
        return true;
    }
    // reserves `bytes` of the bounce buffer (sized from the registration pass over the same arguments)
    int take(size_t bytes, bool input, bool output, size_t *o) {
        if (off + al(bytes) > ws->bounce_bytes) return set_error(RRTMGP_EINVAL, "internal: bounce buffer smaller than the staged arrays");
        *o = off;
        off += al(bytes);
        if (input) in_hi = off;
        if (output) { out_lo = std::min(out_lo, *o); out_hi = off; }
        return RRTMGP_OK;
    }
    // packed mode: the one upload, on the compute stream, right before the launch
    int flush() {
        if (packed && in_hi) {
            RR_HIP(hipMemcpyAsync(ws->bounce_d, ws->bounce_h, in_hi, hipMemcpyHostToDevice, ws->stream));
            ws->h2d_bytes += in_hi;
        }
        return RRTMGP_OK;
    }
    // read only, or (rw) read and written by the call: one upload either way, rw arrays also travel back
    int io(bool rw, int mem, int slot, const void *p, size_t bytes, const void **out) {
        return rw ? inout(mem, slot, p, bytes, (void **)out) : in(mem, slot, p, bytes, out);
    }

    // input: returns device pointer (copying H2D if mem == host)
    int in(int mem, int slot, const void *p, size_t bytes, const void **out) {
        if (!p) { *out = nullptr; return RRTMGP_OK; }
        if (pin(mem, p, bytes, const_cast<void **>(out))) return RRTMGP_OK;
        if (mem == RRTMGP_MEM_DEVICE) { *out = p; return RRTMGP_OK; }
        if (packed) {
            size_t o;
            TRY(take(bytes, true, false, &o));
            memcpy(ws->bounce_h + o, p, bytes);
            *out = ws->bounce_d + o;
            return RRTMGP_OK;
        }
        TRY(stage_ensure(ws, slot, bytes));
        host_range_check(ws, p, bytes);  // no stale page-lock registration under this buffer (host_pin)
        if (!((keep >> slot) & 1)) {
            RR_HIP(hipMemcpyAsync(ws->stage[slot].ptr, p, bytes, hipMemcpyHostToDevice, copy_stream()));
            ws->h2d_bytes += bytes;
        }
        *out = ws->stage[slot].ptr;
        return RRTMGP_OK;
    }
    // input, HOST memory only: `height` rows of `width` bytes, `spitch` bytes apart at the source, packed in the staging buffer
    int in2d(int slot, const void *p, size_t width, size_t height, size_t spitch, const void **out) {
        if (pin_only) { need += al(width * height); *out = nullptr; return RRTMGP_OK; }  // (registration: whole arrays, in())
        if (packed) {
            size_t o;
            TRY(take(width * height, true, false, &o));
            for (size_t r = 0; r < height; r++) memcpy(ws->bounce_h + o + r * width, (const char *)p + r * spitch, width);
            *out = ws->bounce_d + o;
            return RRTMGP_OK;
        }
        TRY(stage_ensure(ws, slot, width * height));
        host_range_check(ws, p, spitch * (height - 1) + width);
        RR_HIP(hipMemcpy2DAsync(ws->stage[slot].ptr, width, p, spitch, width, height, hipMemcpyHostToDevice, copy_stream()));
        ws->h2d_bytes += width * height;
        *out = ws->stage[slot].ptr;
        return RRTMGP_OK;
    }
    // output: returns device pointer; host copies are done by finish()
    int out(int mem, int slot, void *p, size_t bytes, void **outp) {
        if (!p) { *outp = nullptr; return RRTMGP_OK; }
        if (pin(mem, p, bytes, outp)) return RRTMGP_OK;
        if (mem == RRTMGP_MEM_DEVICE) { *outp = p; return RRTMGP_OK; }
        if (packed) {
            size_t o;
            TRY(take(bytes, false, true, &o));
            *outp = ws->bounce_d + o;
            backs.push_back({p, ws->bounce_d + o, bytes});
            return RRTMGP_OK;
        }
        TRY(stage_ensure(ws, slot, bytes));
        host_range_check(ws, p, bytes);
        *outp = ws->stage[slot].ptr;
        backs.push_back({p, ws->stage[slot].ptr, bytes});
        return RRTMGP_OK;
    }
    // output, HOST memory only: `height` pieces of `width` bytes, packed on the device, `dpitch` bytes apart at the host
    int out2d(int slot, void *p, size_t width, size_t height, size_t dpitch, void **outp) {
        if (!p) { *outp = nullptr; return RRTMGP_OK; }
        if (pin_only) { need += al(width * height); *outp = nullptr; return RRTMGP_OK; }
        if (packed) {
            size_t o;
            TRY(take(width * height, false, true, &o));
            *outp = ws->bounce_d + o;
            backs.push_back({p, ws->bounce_d + o, width, height, dpitch});
            return RRTMGP_OK;
        }
        TRY(stage_ensure(ws, slot, width * height));
        host_range_check(ws, p, dpitch * (height - 1) + width);
        *outp = ws->stage[slot].ptr;
        backs.push_back({p, ws->stage[slot].ptr, width, height, dpitch});
        return RRTMGP_OK;
    }
    // read AND written: staged in, copied back by finish()
    int inout(int mem, int slot, const void *p, size_t bytes, void **outp) {
        if (!p) { *outp = nullptr; return RRTMGP_OK; }
        if (pin(mem, p, bytes, outp)) return RRTMGP_OK;
        if (mem == RRTMGP_MEM_DEVICE) { *outp = const_cast<void *>(p); return RRTMGP_OK; }
        if (packed) {
            size_t o;
            TRY(take(bytes, true, true, &o));
            memcpy(ws->bounce_h + o, p, bytes);
            *outp = ws->bounce_d + o;
            backs.push_back({const_cast<void *>(p), ws->bounce_d + o, bytes});
            return RRTMGP_OK;
        }
        TRY(stage_ensure(ws, slot, bytes));
        host_range_check(ws, p, bytes);
        RR_HIP(hipMemcpyAsync(ws->stage[slot].ptr, p, bytes, hipMemcpyHostToDevice, copy_stream()));
        ws->h2d_bytes += bytes;
        *outp = ws->stage[slot].ptr;
        backs.push_back({const_cast<void *>(p), ws->stage[slot].ptr, bytes});
        return RRTMGP_OK;
    }
    int finish() {
        if (packed) {
            if (out_hi > out_lo) {
                RR_HIP(hipMemcpyAsync(ws->bounce_h + out_lo, ws->bounce_d + out_lo, out_hi - out_lo, hipMemcpyDeviceToHost, ws->stream));
                ws->d2h_bytes += out_hi - out_lo;
            }
            RR_HIP(hipStreamSynchronize(ws->stream));
            for (auto &b : backs) {
                const char *src = ws->bounce_h + ((char *)b.dev - ws->bounce_d);
                if (!b.rows) memcpy(b.host, src, b.bytes);
                else for (size_t r = 0; r < b.rows; r++) memcpy((char *)b.host + r * b.hpitch, src + r * b.bytes, b.bytes);
            }
            return RRTMGP_OK;
        }
        if (backs.empty()) return RRTMGP_OK;
        TRY(issue_backs(ws->stream));
        RR_HIP(hipStreamSynchronize(ws->stream));
        return RRTMGP_OK;
    }
    int issue_backs(hipStream_t s) {
        for (auto &b : backs) {
            if (!b.rows) RR_HIP(hipMemcpyAsync(b.host, b.dev, b.bytes, hipMemcpyDeviceToHost, s));
            else RR_HIP(hipMemcpy2DAsync(b.host, b.hpitch, b.dev, b.bytes, b.bytes, b.rows, hipMemcpyDeviceToHost, s));
            ws->d2h_bytes += b.bytes * (b.rows ? b.rows : 1);
        }
        return RRTMGP_OK;
    }
    // copies back on the copy stream, no synchronisation (pipelined host path)
    int copy_back() {
        TRY(issue_backs(copy_stream()));
        backs.clear();
        return RRTMGP_OK;
    }
};

// What a preparation step in the same call writes (rrtmgp_hip_update_fluxes with `prepare`): `core` = layerdata, t_lev,
// vmr_h2o / the full vmr (interpolation, clipping, col_dry); `particles` = vmr_o3 and the cloud / aerosol inputs too (the
// isothermal boundary layer fills their last layer).  Written arrays are staged in AND copied back.
struct StateRW {
    bool core = false, particles = false;
};

// staging.hip
template <typename FT>
int stage_state(Stager &st, const rrtmgp_atmos_state *as, bool use_cld, bool use_aero, bool lw, DevState<FT> &d,
                int64_t nrghice = 1, StateRW rw = StateRW(), bool aero_lookup = true);
template <typename FT>
int stage_flux(Stager &st, const rrtmgp_flux_out *f, const rrtmgp_solve_opts *opts, size_t ncol, size_t nlev, bool sw,
               DevFlux<FT> &d, size_t nbnd = 0, int so = 0 /* S_X_FLUX_UP - S_FLUX_UP: the second block of slots */,
               const FT *staged_metric = nullptr /* the metric factors are on the device already */);
int check_common(rrtmgp_workspace *ws, const rrtmgp_lookup *gas, int want_sw, const rrtmgp_lookup *cld, const rrtmgp_lookup *aero,
                 const rrtmgp_atmos_state *as);
int check_multi(const rrtmgp_workspace *ws, int state_mem, int bcs_mem, const rrtmgp_flux_out *flux, const rrtmgp_solve_opts *opts,
                const void *inc_flux);
bool host_pipeline_applies(const rrtmgp_atmos_state *as, int bcs_mem, const rrtmgp_flux_out *flux, const rrtmgp_solve_opts *opts);
int fork_join_events(rrtmgp_workspace *ws);
int pipeline_resources(rrtmgp_workspace *ws);
int lane_resources(rrtmgp_workspace *ws);
size_t host_pack_max();
inline int n_bnd_of(const rrtmgp_workspace *ws, const rrtmgp_lookup *gas) {
    return ws->ftype == RRTMGP_F32 ? gas->gas32.n_bnd : gas->gas64.n_bnd;
}

// ---- column ranges of the ABI structs ---------------------------------------------------------------------------------
// Host-resident callers (the Julia glue with `array_type = Array`) pay 5 KB per column of uploads
// and 1-2 KB of downloads per solve.  Every per-column array is one contiguous slab per column range
// (ncol is the slowest dimension), so a large solve is cut into column chunks: chunk c + 1 is uploaded
// on a copy stream (into the other staging set) while chunk c is being solved, and chunk c - 1's
// fluxes travel back at the same time.  Results are identical to the single-launch path: columns are
// independent and the McICA stream is keyed by the global column (col_offset).
struct ColumnSlice {
    size_t E, c0;
    template <typename T>
    T *adv(T *p, size_t elems_per_col) const {
        return p ? (T *)((char *)const_cast<typename std::remove_const<T>::type *>(p) + elems_per_col * c0 * E) : nullptr;
    }
};

inline void slice_state(rrtmgp_atmos_state &a, const ColumnSlice &s, size_t nc) {
    const size_t nlay = a.nlay, nlev = nlay + 1;
    a.ncol = (int64_t)nc;
    a.layerdata = s.adv(a.layerdata, 4 * nlay); a.p_lev = s.adv(a.p_lev, nlev); a.t_lev = s.adv(a.t_lev, nlev);
    a.t_sfc = s.adv(a.t_sfc, 1); a.lat = s.adv(a.lat, 1);
    a.vmr_h2o = s.adv(a.vmr_h2o, nlay); a.vmr_o3 = s.adv(a.vmr_o3, nlay);
    if (a.vmr_kind == RRTMGP_VMR_FULL) a.vmr = s.adv(a.vmr, (size_t)a.ngas * nlay);
    a.cld_r_eff_liq = s.adv(a.cld_r_eff_liq, nlay); a.cld_r_eff_ice = s.adv(a.cld_r_eff_ice, nlay);
    a.cld_path_liq = s.adv(a.cld_path_liq, nlay); a.cld_path_ice = s.adv(a.cld_path_ice, nlay);
    a.cld_frac = s.adv(a.cld_frac, nlay);
    a.cld_cover_lw = s.adv(a.cld_cover_lw, 1); a.cld_cover_sw = s.adv(a.cld_cover_sw, 1);
    a.aero_size = s.adv(a.aero_size, RRTMGP_N_AEROSOLS * nlay); a.aero_mass = s.adv(a.aero_mass, RRTMGP_N_AEROSOLS * nlay);
    a.aod_sw_ext = s.adv(a.aod_sw_ext, 1); a.aod_sw_sca = s.adv(a.aod_sw_sca, 1);
}
inline void slice_opts(rrtmgp_solve_opts &o, const ColumnSlice &s, size_t nlev) {
    o.metric_scaling = s.adv(o.metric_scaling, nlev);
    o.col_offset += (int64_t)s.c0;
}
inline void slice_flux_arrays(rrtmgp_flux_out &f, const ColumnSlice &s, size_t nlev, size_t ncol_total) {
    // (nlev, ncol): a contiguous slab, nlev values per column; (ncol, nlev): the block starts c0 elements in and keeps
    // the row length of the whole array
    size_t per_col = nlev;
    if (f.layout == RRTMGP_LAYOUT_NCOL_NLEV) {
        per_col = 1;
        if (f.flux_ncol <= 0) f.flux_ncol = (int64_t)ncol_total;
    }
    f.flux_up = s.adv(f.flux_up, per_col); f.flux_dn = s.adv(f.flux_dn, per_col); f.flux_net = s.adv(f.flux_net, per_col);
    f.flux_dn_dir = s.adv(f.flux_dn_dir, per_col);
    if (f.band_flux_up || f.band_flux_dn || f.band_flux_net) {   // (nlev, ncol, nbnd): the block starts c0 columns in
        if (f.band_flux_ncol <= 0) f.band_flux_ncol = (int64_t)ncol_total;
        f.band_flux_up = s.adv(f.band_flux_up, nlev); f.band_flux_dn = s.adv(f.band_flux_dn, nlev);
        f.band_flux_net = s.adv(f.band_flux_net, nlev);
    }
    f.clear_flux_up = s.adv(f.clear_flux_up, per_col); f.clear_flux_dn = s.adv(f.clear_flux_dn, per_col);
    f.clear_flux_net = s.adv(f.clear_flux_net, per_col); f.clear_flux_dn_dir = s.adv(f.clear_flux_dn_dir, per_col);
}
inline void slice_flux(rrtmgp_flux_out &f, rrtmgp_solve_opts &o, const ColumnSlice &s, size_t nlev, size_t ncol_total) {
    slice_flux_arrays(f, s, nlev, ncol_total);
    slice_opts(o, s, nlev);
}

// `ncol` = columns of the array being sliced (the leading dimension of its inc_flux unless the caller gave one)
inline void slice_lw_bcs(rrtmgp_lw_bcs &b, const ColumnSlice &s, size_t nbnd, size_t ncol) {
    b.sfc_emis = s.adv(b.sfc_emis, nbnd);
    if (b.inc_flux) {
        if (b.inc_flux_ld <= 0) b.inc_flux_ld = (int32_t)ncol;
        b.inc_flux = s.adv(b.inc_flux, 1);
    }
}
inline void slice_sw_bcs(rrtmgp_sw_bcs &b, const ColumnSlice &s, size_t nbnd) {
    b.cos_zenith = s.adv(b.cos_zenith, 1); b.toa_flux = s.adv(b.toa_flux, 1);
    b.sfc_alb_direct = s.adv(b.sfc_alb_direct, nbnd); b.sfc_alb_diffuse = s.adv(b.sfc_alb_diffuse, nbnd);
}
inline void slice_gray(rrtmgp_gray_state &g, const ColumnSlice &s, size_t nc) {
    const size_t nlay = g.nlay, nlev = nlay + 1;
    g.ncol = (int64_t)nc;
    g.lat = s.adv(g.lat, 1); g.t_sfc = s.adv(g.t_sfc, 1);
    g.p_lay = s.adv(g.p_lay, nlay); g.t_lay = s.adv(g.t_lay, nlay);
    g.p_lev = s.adv(g.p_lev, nlev); g.t_lev = s.adv(g.t_lev, nlev);
}

// `chunk(slice, n_columns, stager, vmr_is_gm)` stages and launches the columns [slice.c0, slice.c0 + n_columns)
template <typename F>
int run_column_pipeline(rrtmgp_workspace *ws, size_t ncol, size_t E, bool vmr_gm, F &&chunk) {
    TRY(pipeline_resources(ws));
    // Chunk sizes.  Two costs pull in opposite directions: nothing overlaps the first chunk's upload and the last chunk's
    // download (small chunks at both ends), and every launch ends with the tail of its persistent grid — the last column of
    // each workgroup finishes alone — which a step cut into 16 equal chunks pays 16 times per kernel (4.4 ms of a 38 ms
    // Layer-2 step, tools/experiments/README.md).  So the chunks ramp: 4 096 columns first, doubling up to `ramp_max`, the
    // same downwards at the end, the middle in equal pieces no larger than `ramp_max`.  RRTMGP_HIP_HOST_CHUNK_COLUMNS = n:
    // equal chunks of about n columns instead (the pipeline of rounds 2-3 with n = 8192).
    static const size_t fixed_chunk = getenv("RRTMGP_HIP_HOST_CHUNK_COLUMNS") ? (size_t)atol(getenv("RRTMGP_HIP_HOST_CHUNK_COLUMNS")) : 0;
    static const size_t ramp_max = getenv("RRTMGP_HIP_HOST_RAMP_MAX") ? (size_t)atol(getenv("RRTMGP_HIP_HOST_RAMP_MAX")) : 32768;
    std::vector<size_t> edge{0};
    {
        std::vector<size_t> up;
        size_t ramp = 0;
        for (size_t n = 4096; n < ramp_max; n *= 2) { up.push_back(n); ramp += n; }
        if (fixed_chunk || ncol < 2 * ramp + ramp_max / 2 || up.empty()) {
            const size_t per_chunk = fixed_chunk ? fixed_chunk : 8192;
            const size_t nc = std::min<size_t>(32, std::max<size_t>(2, ncol / std::max<size_t>(per_chunk, 1024)));
            const size_t per = (ncol + nc - 1) / nc;
            for (size_t c = 1; c <= nc && edge.back() < ncol; c++) edge.push_back(std::min(ncol, per * c));
        } else {
            for (size_t n : up) edge.push_back(edge.back() + n);
            const size_t mid = ncol - 2 * ramp, nmid = (mid + ramp_max - 1) / ramp_max;
            for (size_t c = 1; c <= nmid; c++) edge.push_back(ramp + mid * c / nmid);
            for (size_t i = up.size(); i-- > 0;) edge.push_back(edge.back() + up[i]);
        }
    }
    const int nchunk = (int)edge.size() - 1;
    RR_HIP(hipStreamSynchronize(ws->stream));  // earlier work of the caller on this workspace
    // One copy stream, two staging sets: chunk c + 1 is uploaded while chunk c is solved, and chunk c - 1 comes home behind
    // that upload.  (Downloads on a second stream over three staging sets — PCIe is full duplex — measured slower: 38.3 vs
    // 37.4-37.8 ms for the Layer-2 step, 39.7 vs 38.0 ms for two separate solves; every cross-stream event costs the copy
    // queue more than the overlap returns, and an event recorded between the copies of one stream makes the runtime fall
    // back to shader copies that queue behind the persistent solve grid.  tools/experiments/README.md, round 4.)
    Stager prev{ws, {}};
    prev.cs = ws->copy_stream;
    int rc = RRTMGP_OK;
    // (the staging buffers grow to the largest chunk's size during the first call only — hipFree waits for the device, so a
    // buffer is never released under a copy in flight — and stay there: warm calls allocate nothing)
    for (int c = 0; c < nchunk && rc == RRTMGP_OK; c++) {
        const size_t c0 = edge[c], c1 = edge[c + 1];
        ColumnSlice sl{E, c0};
        std::swap(ws->stage, ws->stage_alt);  // the set chunk c - 2 used; its downloads are ahead of these uploads in the copy stream
        Stager st{ws, {}};
        st.cs = ws->copy_stream;
        // the well-mixed vmr vector (VmrGM) does not depend on the column range: chunks 0 and 1 put it into the two
        // staging sets, later chunks leave it there.  A copy that small is done by a blit KERNEL, which has to wait
        // for a workgroup slot of the persistent solve grid (0.3 ms per chunk in the copy queue: rocprofv3 timeline,
        // tools/experiments/host_timeline.sh)
        static const bool restage_all = getenv("RRTMGP_HIP_HOST_RESTAGE_ALL") != nullptr;
        if (!restage_all && c >= 2 && vmr_gm) st.keep |= 1ull << S_VMR;
        rc = chunk(sl, c1 - c0, st);
        if (rc == RRTMGP_OK && hipEventRecord(ws->ev_k[c & 1], ws->stream) != hipSuccess) rc = set_error(RRTMGP_EHIP, "hipEventRecord");
        // chunk c - 1: its kernel is older than chunk c's, wait for it on the copy stream and bring the fluxes home
        if (rc == RRTMGP_OK && c > 0) {
            if (hipStreamWaitEvent(ws->copy_stream, ws->ev_k[(c - 1) & 1], 0) != hipSuccess) rc = set_error(RRTMGP_EHIP, "hipStreamWaitEvent");
            else rc = prev.copy_back();
        }
        prev.backs = std::move(st.backs);
    }
    if (rc == RRTMGP_OK) {
        // the last chunk's kernel is the newest work on the compute stream
        if (hipStreamSynchronize(ws->stream) != hipSuccess) rc = set_error(RRTMGP_EHIP, "hipStreamSynchronize");
        else rc = prev.copy_back();
    }
    (void)hipStreamSynchronize(ws->copy_stream);
    (void)hipStreamSynchronize(ws->stream);
    return rc;
}
// one solver: `solve_chunk(as_c, flux_c, opts_c, slice, stager)` stages and launches one chunk
template <typename F>
int run_host_pipeline(rrtmgp_workspace *ws, const rrtmgp_atmos_state *as, const rrtmgp_flux_out *flux,
                             const rrtmgp_solve_opts *opts, size_t E, F &&solve_chunk) {
    const size_t nlev = as->nlay + 1, ncol = as->ncol;
    return run_column_pipeline(ws, ncol, E, as->vmr_kind == RRTMGP_VMR_GM, [&](const ColumnSlice &sl, size_t nc, Stager &st) {
        rrtmgp_atmos_state a = *as;
        rrtmgp_flux_out f = *flux;
        rrtmgp_solve_opts o{};
        if (opts) o = *opts; else o.n_gauss_angles = 1;
        slice_state(a, sl, nc);
        slice_flux(f, o, sl, nlev, ncol);
        return solve_chunk(a, f, o, sl, st);
    });
}

// The registration pass over the caller's WHOLE host arrays: which of them are page-locked (and, opt-in, locking the large
// ones on first sight).  What it verifies stays marked as used by `ws` until the PinScope of the solve ends.
struct PinScope {
    rrtmgp_workspace *ws;
    ~PinScope() { host_pin_end(ws); }
};

}  // namespace rrtmgp
