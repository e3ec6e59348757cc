// common.h — device-side table/state views and host-side handles of libhip_rrtmgp.so.
//
// gfx950 (MI355X) only.  Tables are re-laid-out g-point-innermost at lookup
// creation so that the 64 lanes of a wavefront (= 64 consecutive g-points of one
// column) read each interpolation corner as one coalesced 256-byte segment.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <functional>
#include <string>
#include <map>
#include <utility>
#include <vector>

#include "../../include/rrtmgp_hip.h"

namespace rrtmgp {

// ---- device views (trivially copyable; passed to kernels by value) -------------

// LookUpLW / LookUpSW (src/optics/LookUpTables.jl:130-201), device layout.
template <typename FT>
struct DevGas {
    int is_sw, n_gpt, n_bnd, n_eta, n_pp /* n_p_ref + 1 */, n_t_ref, n_gases, n_t_plnk, idx_h2o;
    // per-band fluxes: the lanes of a workgroup are laid out band by band, every band starting on a 16-lane DPP row
    // (padding lanes idle); band_rows = rows of that layout, 0 when it needs more than the 16 rows of a workgroup.
    // With whole 16-g-point bands (rrtmgp-data g256 / g224) the layout is the identity.
    int band_rows;
    const int *band_row_lo;    // [n_bnd + 1] first row of each band
    const int *band_lane_gpt;  // [256] g-point (0-based) of each lane of that layout, -1 for padding
    const int *lane_gpt;       // [256] the same for the broadband instances: whole bands dealt to the wavefronts by minor-gas slot count (build_gas)
    FT p_ref_tropo;
    // the tables the g-point lanes gather from live in ONE allocation (one scalar base address for
    // every global_load of the hot loop); offsets in bytes:
    const char *arena;
    // [t][p][eta][gpt] entries of KMAJOR_ENTRY_BYTES = 16: everything ONE gather per (T plane[, p plane]) corner needs.
    //   LW Float32 {k(e), pf(e), k(e+1), pf(e+1)}   LW Float64 {k(e), pf(e)}          (pf = planck_fraction)
    //   SW Float32 {k(e,p), k(e+1,p), k(e,p+1), k(e+1,p+1)}   SW Float64 {k(e), k(e+1)}
    // (the neighbour of the last eta / p row repeats that row: the interpolation never starts there)
    unsigned off_kmajor;
    // [t][eta][row] of 16-byte (Float32) entries {c_2p(e), c_2p(e+1), c_2p+1(e), c_2p+1(e+1)} at koff[b] + (p*ng_b + (g - lo_b))*4:
    // slot pair p of the g-point at eta and eta + 1; region 0 lower, 1 upper.  Slot j = contributor i for the LW lookup;
    // SW: j = i + 1, and slot 0 holds the Rayleigh coefficient of the region (krayl) — RAYLEIGH_SLOT
    unsigned off_kminor[2];
    const FT *t_planck;    // [n_t_plnk]                  (LW)
    const FT *tot_planck;  // [bnd][n_t_plnk]             (LW; = reference (n_t_plnk, n_bnd))
    const FT *ln_p_ref;    // [n_p_ref]
    const FT *t_ref;       // [n_t_ref]
    const FT *vmr_ref;     // (2, n_gases, n_t_ref) as the reference stores it
    const FT *eta_half;    // [tropo][bnd][n_t_ref]: vmr_ref ratio of the band's two key species (gas_optics.jl:140-143)
    const int *key_species;  // (2, 2, n_bnd), gas indices as in the reference
    const int *gpt2bnd;      // [n_gpt] 0-based band
    const int *bnd_lo;       // [n_bnd] first g-point (0-based) of the band
    const int *bnd_ng;       // [n_bnd] g-points in the band
    // minor gases, region 0 = lower, 1 = upper atmosphere
    const int *m_bnd_st[2];   // [n_bnd+1] 0-based start into gasdata columns
    const int *m_gasdata[2];  // (4, n_min_absrb)
    const int *m_koff[2];     // [n_bnd] offset of the band's block along the (padded) contributor axis
    const int *m_st2[2];      // [n_bnd] first scaling slot of the band (a multiple of MINOR_PAIR)
    const int *m_slot_int[2]; // [m_nslot] gasdata column of each scaling slot, -1 for padding, RAYLEIGH_SLOT for the Rayleigh slot (SW)
    int m_ncontrib[2];        // row length of kminor in elements: 4 per (band pair, g-point)
    int m_nint[2];            // minor intervals (gasdata columns) per region
    int m_nslot[2];           // scaling slots per region (every band padded to whole pairs)
    const FT *solar_src_scaled;  // [n_gpt]         (SW)
};

// LookUpCld (LookUpTables.jl:239-284); data kept in the reference layout.
template <typename FT>
struct DevCld {
    int nband, nrghice, nsize_liq, nsize_ice;
    FT radliq_lwr, radliq_upr, radice_lwr, radice_upr;
    const FT *liqdata;  // (3*nsize_liq, nband)
    const FT *icedata;  // (3*nsize_ice, nband, nrghice)
};

// LookUpAerosolMerra (LookUpTables.jl:312-325); reference layouts.
template <typename FT>
struct DevAero {
    int nband, nbin, nrh, iband_550nm;
    const FT *size_bin_limits, *rh_levels, *dust, *sea_salt, *sulfate, *black_carbon_rh, *black_carbon,
        *organic_carbon_rh, *organic_carbon;
};

// AtmosphericState (+ cloud / aerosol state) device view, reference layouts.
template <typename FT>
struct DevState {
    int ncol, nlay, ngas, vmr_kind;
    const FT *layerdata, *t_lev, *t_sfc;
    const FT *vmr_h2o, *vmr_o3, *vmr;
    const FT *cld_r_eff_liq, *cld_r_eff_ice, *cld_path_liq, *cld_path_ice, *cld_frac;
    FT *cld_cover;  // LW or SW cover for this solve, or nullptr
    int ice_rgh;
    const FT *aero_size, *aero_mass;
    FT *aod_sw_ext, *aod_sw_sca;
};

template <typename FT>
struct DevFlux {
    FT *up, *dn, *net, *dir;
    int layout;
    int ld;  // RRTMGP_LAYOUT_NCOL_NLEV: elements between consecutive levels of the flux arrays as the kernel writes them (>= ncol)
    const FT *metric;  // (nlev, ncol) or nullptr
    FT *band_up, *band_dn, *band_net;  // optional FluxBand (nlev, band_ncol, nbnd); band_net may be null on its own
    int band_ncol;                     // second dimension of the band arrays as the kernel writes them (>= ncol)
    FT *clear_up, *clear_dn, *clear_net, *clear_dir;  // optional clear-sky diagnostic (same layout as up/dn/net/dir)
};

// ---- host-side handles ------------------------------------------------------------

struct DeviceBuffer {
    void *ptr = nullptr;
    size_t bytes = 0;
};

enum LookupKind { LK_GAS = 1, LK_CLOUD = 2, LK_AEROSOL = 3 };

constexpr unsigned KMAJOR_ENTRY_BYTES = 16;
constexpr int MINOR_PAIR = 2;    // minor-gas slots served by one gather per T plane
constexpr int MINOR_ENTRY = 4;   // elements of a kminor entry: MINOR_PAIR slots x (eta, eta + 1)
constexpr int RAYLEIGH_SLOT = -2;  // m_slot_int value of the slot that carries krayl; its scaling is (h2o + 1) col_dry

}  // namespace rrtmgp

namespace rrtmgp { struct ShardWorkers; void shard_workers_destroy(ShardWorkers *); }

struct rrtmgp_lookup {
    int kind;
    int ftype;
    int device;
    std::vector<void *> allocs;  // every device allocation owned by this lookup
    // one of the following is valid, by (kind, ftype)
    rrtmgp::DevGas<float> gas32;
    rrtmgp::DevGas<double> gas64;
    rrtmgp::DevCld<float> cld32;
    rrtmgp::DevCld<double> cld64;
    rrtmgp::DevAero<float> aero32;
    rrtmgp::DevAero<double> aero64;
    int max_minor;  // largest per-band minor count (either region)
    int max_int;    // scaling slots of the larger region (m_nslot)
    // *_lookup_create_multi: replicas on the other devices (owned by this head; each has no replicas of its own)
    std::vector<rrtmgp_lookup *> replicas;
};

struct rrtmgp_workspace {
    int device;
    int ftype;
    int64_t ncol, nlay;
    hipStream_t own_stream = nullptr;
    hipStream_t stream = nullptr;
    hipEvent_t ev_start = nullptr, ev_stop = nullptr;
    bool timed = false;
    int n_cu = 0;
    rrtmgp_workspace *head = nullptr;  // shard of a multi-device workspace: its head
    // staging mirrors for host-memory callers, keyed by slot; the second set and the copy stream serve the
    // pipelined host path (column chunks: chunk c+1 is uploaded while chunk c is being solved)
    std::vector<rrtmgp::DeviceBuffer> stage, stage_alt;
    hipStream_t copy_stream = nullptr;
    // second compute lane of a SHORT Layer-2 step (step.hip step_launch): the SW kernels run on this stream with this sweep
    // scratch while the LW kernels run on `stream`, so that one solver's workgroups fill the slots the other's tail frees
    hipStream_t alt_stream = nullptr;
    rrtmgp::DeviceBuffer alt_scratch;
    int *col_queue[2] = {nullptr, nullptr};   // per lane: {next column, workgroups done} (queue_ensure)
    hipEvent_t ev_in[2] = {nullptr, nullptr}, ev_k[3] = {nullptr, nullptr, nullptr};
    // small host-array solves: every array travels through ONE page-locked bounce buffer (a host memcpy per array, one
    // DMA each way) instead of one DMA per array (~15 us each, 17 arrays per solve)
    char *bounce_h = nullptr, *bounce_d = nullptr;
    size_t bounce_bytes = 0;
    // bytes staged host -> device / device -> host since creation (rrtmgp_hip_workspace_transfer_bytes)
    uint64_t h2d_bytes = 0, d2h_bytes = 0;
    // per-(block, level, lane) scratch of the vertical sweeps
    rrtmgp::DeviceBuffer scratch;
    // resident workgroups per CU of each (kernel, dynamic LDS size) launched so far
    std::map<std::pair<const void *, size_t>, int> occupancy;
    // workspace_create_multi: one single-device workspace per shard (owned by this head, which holds no device
    // resources of its own); shard s covers the global columns [shard_c0[s], shard_c0[s + 1])
    std::vector<rrtmgp_workspace *> shards;
    std::vector<int64_t> shard_c0;
    rrtmgp::ShardWorkers *workers = nullptr;  // persistent host threads of the shards (multi.hip), created by the first call
};

namespace rrtmgp {

int set_error(int code, const std::string &msg);
int hip_fail(hipError_t e, const char *what, const char *file, int line);

#define RR_HIP(call)                                                        \
    do {                                                                    \
        hipError_t _e = (call);                                             \
        if (_e != hipSuccess) return rrtmgp::hip_fail(_e, #call, __FILE__, __LINE__); \
    } while (0)

#define RR_CHECK(cond, msg)                                         \
    do {                                                            \
        if (!(cond)) return rrtmgp::set_error(RRTMGP_EINVAL, msg);  \
    } while (0)

// ensure ws->stage[slot] holds at least `bytes`
int stage_ensure(rrtmgp_workspace *ws, int slot, size_t bytes);
// A lane = the stream a launch goes to and the sweep scratch it uses.  Every launch helper takes one explicitly where the
// workspace has two (the fused step runs short SW solves on the second lane): nothing swaps fields of the workspace in and
// out around a call any more.  nullptr = the workspace's main lane.
struct Lane {
    hipStream_t stream;
    DeviceBuffer *scratch;
    int idx;   // 0 = main, 1 = second: which column-queue counters (rrtmgp_workspace::col_queue) its launches use
};
int scratch_ensure(rrtmgp_workspace *ws, size_t bytes, const Lane *lane = nullptr);
// The column queue of a lane's launches: {next column, workgroups done}, 256 bytes of device memory allocated and zeroed ONCE.
// The kernels leave it zeroed themselves (queue_release, device.h: the last workgroup out resets both counters), so a launch
// needs no memset in front of it (round 5: two fill kernels + their dependencies were ~15 us of a 250 us step of 512 columns).
int queue_ensure(rrtmgp_workspace *ws, int lane_idx, int **out);
// reads the counters of both lanes back (the lanes must be idle) and fails loudly if a kernel left them dirty; `reset_only`:
// just re-zeroes them (workspace_set_stream).  Every kernel that takes columns from the queue must end with queue_release.
int queue_check(rrtmgp_workspace *ws, bool reset_only);

// every device allocation of the library goes through these two (rrtmgp_hip_allocation_counts)
hipError_t rr_malloc(void **p, size_t bytes);
hipError_t rr_free(void *p);
// page-locked host arrays (process-wide registry, runtime.hip): is this whole caller array registered (explicitly, or —
// opt-in — by the library on first sight)?  `ws` uses it until host_pin_end(ws)
bool host_pin(rrtmgp_workspace *ws, const void *p, size_t bytes);
void host_pin_begin(rrtmgp_workspace *ws);
void host_pin_end(rrtmgp_workspace *ws);
void host_range_check(const rrtmgp_workspace *ws, const void *p, size_t bytes);  // releases stale registrations under [p, p + bytes)

// the replica of `lk` that lives on `device` (lk itself when it does), or nullptr
const rrtmgp_lookup *lookup_on(const rrtmgp_lookup *lk, int device);
const std::string &last_error_string();

// Runs `shard_call(shard_workspace, first_column, n_columns)` for every shard of a multi-device workspace,
// concurrently (one host thread per shard), and returns the first failure (message preserved).
// `device_arrays`: the caller's arrays live in device memory (all shards on that device): work queued on the device is
// waited for first, and the call returns when every shard's stream has drained.
int multi_run(rrtmgp_workspace *ws, const std::function<int(rrtmgp_workspace *, size_t, size_t)> &shard_call,
              bool device_arrays = false);

// Launchers implemented in the .hip translation units.  `which`: 1 = two-stream, 0 = no-scattering.
template <typename FT>
int launch_lw(rrtmgp_workspace *ws, int twostream, const DevGas<FT> &lk, const DevCld<FT> *cld, const DevAero<FT> *aero,
              const DevState<FT> &as, const FT *sfc_emis, const FT *inc_flux, int inc_ld, const DevFlux<FT> &fl, int n_angles,
              uint64_t seed, int64_t col_offset, int max_minor);
template <typename FT>
int launch_sw(rrtmgp_workspace *ws, int twostream, const DevGas<FT> &lk, const DevCld<FT> *cld, const DevAero<FT> *aero,
              const DevState<FT> &as, const FT *cos_zenith, const FT *toa_flux, const FT *alb_dir, const FT *alb_dif,
              const DevFlux<FT> &fl, uint64_t seed, int64_t col_offset, int max_minor, const Lane *lane = nullptr);

struct GrayArgs {
    int otp_kind;
    double otp[5];
    double stefan;
};
template <typename FT>
int launch_gray_lw(rrtmgp_workspace *ws, int twostream, int ncol, int nlay, const GrayArgs &ga, const FT *lat,
                   const FT *p_lay, const FT *p_lev, const FT *t_lay, const FT *t_lev, const FT *t_sfc,
                   const FT *sfc_emis, const FT *inc_flux, const DevFlux<FT> &fl);
template <typename FT>
int launch_gray_sw(rrtmgp_workspace *ws, int twostream, int ncol, int nlay, const GrayArgs &ga, const FT *p_lay,
                   const FT *p_lev, const FT *cos_zenith, const FT *toa_flux, const FT *alb_dir, const FT *alb_dif,
                   const DevFlux<FT> &fl);
// a 2-D array argument as the reference passes it (rrtmgp_view2d): element (i, j) at p[i * s0 + j * s1]
template <typename T>
struct View2 {
    T *p;
    int64_t s0, s1;
    __host__ __device__ __forceinline__ T &operator()(int i, int j) const { return p[(int64_t)i * s0 + (int64_t)j * s1]; }
};
template <typename FT>
int launch_col_gas(rrtmgp_workspace *ws, int ncol, int nlay, View2<const FT> p_lev, View2<FT> col_dry, const rrtmgp_params &ps,
                   View2<const FT> vmr_h2o, const FT *lat);
template <typename FT>
int launch_rel_hum(rrtmgp_workspace *ws, int ncol, int nlay, View2<FT> rh, View2<const FT> p_lay, View2<const FT> t_lay,
                   const rrtmgp_params &ps, View2<const FT> vmr_h2o);

template <typename FT>
int launch_heating_rate(rrtmgp_workspace *ws, int ncol, int nlay, View2<FT> hr_lay, View2<const FT> flux_net,
                        View2<const FT> p_lev, double grav, double cp_d);

// prepare_atmosphere!: the state through layer / level accessors, so that AtmosphericState
// (layerdata rows, element stride 4) and GrayAtmosphericState (separate arrays) share one kernel.
template <typename FT>
struct PrepView {
    int ncol, nlay, ls /* element stride of the four layer arrays */;
    FT *p_lay, *t_lay, *rel_hum, *col_dry;  // element (k, col) at [ls * (k + nlay*col)]
    FT *p_lev, *t_lev;
    const FT *t_sfc, *lat;
    FT *vmr_h2o;  // (k, col) at [hs * (k + nlay*col)]
    int hs;
    FT *vmr_o3, *vmr_full;
    int ngas;
    FT *cld[5];
    FT *aero[2];
    const FT *center_z, *face_z;
};
template <typename FT>
struct PrepArgs {
    int steps, interpolation, bottom_mode, iso, clamp_t;
    FT p_min, t_min, t_max, R, cp, g, mol_m_dry, mol_m_h2o, avogadro;
};
template <typename FT>
int launch_prepare(rrtmgp_workspace *ws, const PrepView<FT> &v, const rrtmgp_params &ps, const rrtmgp_prepare_opts &o,
                   bool gray);

}  // namespace rrtmgp
