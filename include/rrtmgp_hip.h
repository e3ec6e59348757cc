/*
 * rrtmgp_hip.h — C ABI of libhip_rrtmgp.so, the MI355X (gfx950) device back end
 * for RRTMGP.jl's per-column radiative-transfer hot path.
 *
 * The reference has no FFI: its device back end is selected by Julia multiple
 * dispatch on `context.device`.  Each entry point below
 * replaces one device method of ext/RRTMGPCUDAExt.jl and the files under ext/cuda; the
 * reference method it stands in for is cited next to it as file:line relative
 * to the reference tree.  The Julia-side binding (`ccall`) a maintainer would
 * add is shown in INTEGRATION.md and ext/RRTMGPHIPExt.jl.
 *
 * Conventions
 *  - Plain pointers and sizes only; no C++ / torch types cross this boundary.
 *  - All arrays use the reference's own in-memory layouts (Julia column-major;
 *    the FIRST listed dimension is fastest).  Index tables are Julia `Int`
 *    (int64_t) and 1-based, exactly as the reference structs hold them.
 *  - `ftype` is sizeof(FT): 4 = Float32, 8 = Float64.  `const void*` arrays
 *    hold FT elements.
 *  - Vertical index 1 (C offset 0) is the surface, nlev = nlay+1 the top.
 *  - Lookup tables are always HOST pointers; they are re-laid-out
 *    (g-point-innermost) and uploaded once by *_create, mirroring the one-off
 *    `DA(...)` uploads in ext/lookup_constructors.jl:83,407,727,18.
 *  - State / boundary-condition / flux arrays are host OR device pointers as
 *    declared by `mem` in each struct (device pointers are used in place; host
 *    pointers are staged through the workspace's own HBM mirrors).
 *  - Every function returns 0 on success or a negative RRTMGP_E* code;
 *    rrtmgp_hip_last_error() gives the message.  Like the reference kernels,
 *    out-of-range physical inputs are clamped, never rejected.
 *  - A workspace may be used by one host thread at a time; different
 *    workspaces are independent and may be driven from different host threads
 *    concurrently (tests/test_abi_contracts.py).  Calls are stream-ordered on the workspace
 *    stream and block until results are in the caller's arrays only when the
 *    arrays are host memory.
 */
#ifndef RRTMGP_HIP_H
#define RRTMGP_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RRTMGP_F32 4
#define RRTMGP_F64 8

#define RRTMGP_MEM_HOST 0
#define RRTMGP_MEM_DEVICE 1

/* flux array layouts */
#define RRTMGP_LAYOUT_NCOL_NLEV 0 /* (ncol, nlev): FluxLW/FluxSW on a GPU device, Fluxes.jl:45-49 */
#define RRTMGP_LAYOUT_NLEV_NCOL 1 /* (nlev, ncol): FluxPresentation, Fluxes.jl:384 */

#define RRTMGP_VMR_GM 0   /* VmrGM, VolumeMixingRatios.jl:34-46 */
#define RRTMGP_VMR_FULL 1 /* Vmr,   VolumeMixingRatios.jl:75-80 */

#define RRTMGP_OK 0
#define RRTMGP_EINVAL -1   /* bad argument / inconsistent dimensions */
#define RRTMGP_ENODEV -2   /* no usable HIP device */
#define RRTMGP_EHIP -3     /* HIP runtime error */
#define RRTMGP_ENOMEM -4
#define RRTMGP_EUNSUPPORTED -5

#define RRTMGP_N_AEROSOLS 15 /* src/api/aerosols.jl:18-34 */

/* ---- lookup tables (reference in-memory form) -------------------------- */

/* LookUpMinor, src/optics/LookUpTables.jl:36-56 */
typedef struct rrtmgp_minor_desc {
    int64_t n_min_absrb;    /* size(gasdata, 2) */
    int64_t n_contrib;      /* size(kminor, 3) */
    const int64_t *bnd_st;  /* (n_bnd+1)  1-based start into gasdata columns */
    const int64_t *gpt_st;  /* (n_gpt+1)  1-based start into kminor axis 3 */
    const int64_t *gasdata; /* (4, n_min_absrb): gas idx, scaling-gas idx, scales_with_density, scale_by_complement */
    const void *kminor;     /* FT (n_eta, n_t_ref, n_contrib) */
} rrtmgp_minor_desc;

/* LookUpLW (LookUpTables.jl:130-143) and LookUpSW (:185-201) */
typedef struct rrtmgp_gas_lookup_desc {
    int32_t ftype;
    int32_t is_sw;
    int64_t n_gpt;    /* 256 LW / 224 SW in rrtmgp-data v1.9 */
    int64_t n_bnd;
    int64_t n_eta;    /* size(kmajor, 1) = 9 */
    int64_t n_p_ref;  /* length(ln_p_ref) = 59; size(kmajor, 2) = n_p_ref + 1 */
    int64_t n_t_ref;  /* 14 */
    int64_t n_gases;  /* size(vmr_ref, 2) = ngas + 1 (slot 1 = dry air) */
    int64_t n_t_plnk; /* LW: length(t_planck) = 196 */
    int64_t idx_h2o;
    double p_ref_tropo;
    double p_ref_min;
    double t_ref_min;
    double t_ref_max;
    double solar_src_tot;        /* SW */
    const int64_t *key_species;  /* (2, 2, n_bnd) */
    const int64_t *major_gpt2bnd;/* (n_gpt) */
    const void *kmajor;          /* FT (n_eta, n_p_ref+1, n_t_ref, n_gpt) */
    const void *planck_fraction; /* LW: FT, same shape as kmajor */
    const void *t_planck;        /* LW: FT (n_t_plnk) */
    const void *tot_planck;      /* LW: FT (n_t_plnk, n_bnd) */
    const void *ln_p_ref;        /* FT (n_p_ref) */
    const void *t_ref;           /* FT (n_t_ref) */
    const void *vmr_ref;         /* FT (2, n_gases, n_t_ref) */
    rrtmgp_minor_desc minor_lower;
    rrtmgp_minor_desc minor_upper;
    const void *rayl_lower;       /* SW: FT (n_eta, n_t_ref, n_gpt) */
    const void *rayl_upper;       /* SW */
    const void *solar_src_scaled; /* SW: FT (n_gpt) */
} rrtmgp_gas_lookup_desc;

/* LookUpCld, LookUpTables.jl:239-284 */
typedef struct rrtmgp_cloud_lookup_desc {
    int32_t ftype;
    int32_t _pad;
    int64_t nband, nrghice, nsize_liq, nsize_ice; /* dims[1:4] */
    const void *bounds;  /* FT (4): radliq_lwr, radliq_upr, radice_lwr, radice_upr */
    const void *liqdata; /* FT (3*nsize_liq, nband): ext | ssa | asy stacked */
    const void *icedata; /* FT (3*nsize_ice, nband, nrghice) */
} rrtmgp_cloud_lookup_desc;

/* LookUpAerosolMerra, LookUpTables.jl:312-325 */
typedef struct rrtmgp_aerosol_lookup_desc {
    int32_t ftype;
    int32_t _pad;
    int64_t nband, nbin, nrh;
    int64_t iband_550nm;          /* 0 if none */
    const void *size_bin_limits;  /* FT (2, nbin) */
    const void *rh_levels;        /* FT (nrh) */
    const void *dust;             /* FT (3, nbin, nband) */
    const void *sea_salt;         /* FT (3, nrh, nbin, nband) */
    const void *sulfate;          /* FT (3, nrh, nband) */
    const void *black_carbon_rh;  /* FT (3, nrh, nband) */
    const void *black_carbon;     /* FT (3, nband) */
    const void *organic_carbon_rh;/* FT (3, nrh, nband) */
    const void *organic_carbon;   /* FT (3, nband) */
} rrtmgp_aerosol_lookup_desc;

/* ---- caller-owned state, boundary conditions, outputs ------------------- */

/* AtmosphericState (+ CloudState, AerosolState, Vmr/VmrGM),
 * src/optics/AtmosphericStates.jl:70-82,236-248,292-298 */
typedef struct rrtmgp_atmos_state {
    int32_t mem;      /* RRTMGP_MEM_HOST / RRTMGP_MEM_DEVICE for every pointer below */
    int32_t vmr_kind; /* RRTMGP_VMR_GM / RRTMGP_VMR_FULL */
    int64_t ncol;
    int64_t nlay;
    int64_t ngas;          /* GM: length(vmr); FULL: size(vmr, 1) */
    const void *layerdata; /* FT (4, nlay, ncol): col_dry, p_lay, t_lay, rel_hum */
    const void *p_lev;     /* FT (nlev, ncol); read only by compute_col_gas */
    const void *t_lev;     /* FT (nlev, ncol) */
    const void *t_sfc;     /* FT (ncol) */
    const void *lat;       /* FT (ncol) or NULL */
    const void *vmr_h2o;   /* GM: FT (nlay, ncol) */
    const void *vmr_o3;    /* GM: FT (nlay, ncol) */
    const void *vmr;       /* GM: FT (ngas); FULL: FT (ngas, nlay, ncol) */
    /* CloudState; all NULL when the solve has no clouds */
    const void *cld_r_eff_liq; /* FT (nlay, ncol) [um] */
    const void *cld_r_eff_ice;
    const void *cld_path_liq;  /* [g/m2] */
    const void *cld_path_ice;
    const void *cld_frac;
    void *cld_cover_lw; /* out FT (ncol) or NULL */
    void *cld_cover_sw; /* out FT (ncol) or NULL */
    int64_t ice_rgh;    /* 1..3 */
    /* AerosolState; NULL when the solve has no aerosols */
    const void *aero_size; /* FT (15, nlay, ncol) [um] */
    const void *aero_mass; /* FT (15, nlay, ncol) [kg/m2] */
    void *aod_sw_ext;      /* out FT (ncol) or NULL */
    void *aod_sw_sca;      /* out FT (ncol) or NULL */
} rrtmgp_atmos_state;

/* LwBCs, src/optics/BCs.jl:17-26 */
typedef struct rrtmgp_lw_bcs {
    int32_t mem;
    int32_t inc_flux_ld;  /* leading dimension of inc_flux: elements between consecutive g-points; 0 = ncol.  Larger
                           * values describe a block of columns inside a wider (ncol_total, ngpt) array: that is how the
                           * library itself hands column ranges of inc_flux (the one array whose FASTEST dimension is
                           * ncol) to shards and pipeline chunks. */
    const void *sfc_emis; /* FT (nbnd_lw, ncol) */
    const void *inc_flux; /* FT (ncol, ngpt) or NULL */
} rrtmgp_lw_bcs;

/* SwBCs, src/optics/BCs.jl:40-59 (inc_flux_diffuse is stored but never read
 * by the reference solver, shortwave_2stream.jl:331, so it is not passed) */
typedef struct rrtmgp_sw_bcs {
    int32_t mem;
    int32_t _pad;
    const void *cos_zenith;      /* FT (ncol) */
    const void *toa_flux;        /* FT (ncol) */
    const void *sfc_alb_direct;  /* FT (nbnd_sw, ncol) */
    const void *sfc_alb_diffuse; /* FT (nbnd_sw, ncol) */
} rrtmgp_sw_bcs;

/* FluxLW / FluxSW broadband accumulators, src/optics/Fluxes.jl:93-149 */
typedef struct rrtmgp_flux_out {
    int32_t mem;
    int32_t layout;    /* RRTMGP_LAYOUT_* */
    void *flux_up;     /* FT, ncol*nlev */
    void *flux_dn;
    void *flux_net;
    void *flux_dn_dir; /* SW only; NULL for LW */
    /* FluxBand, src/optics/Fluxes.jl:170-215: optional per-band fluxes, FT (nlev, ncol, nbnd)
     * whatever `layout` says (the reference keeps them vertical-first), all three NULL = off.
     * Two-stream non-gray solvers only (src/api/getters.jl:404).  Zeroed, accumulated per
     * g-point into the g-point's band and metric-scaled as RTESolver.jl:141,246 does;
     * `band_flux_net` (may be NULL on its own) = scaled up - scaled dn, which the reference
     * fills one step later in update_net_fluxes! (src/api/update_fluxes.jl:198-201).
     * The per-band kernels lay the g-points out band by band on 16-lane rows: any band structure
     * whose bands, each rounded up to 16 g-points, fit 256 lanes (rrtmgp-data v1.9 g256 / g224 and
     * the reduced g128 / g112 sets do); RRTMGP_EUNSUPPORTED otherwise. */
    void *band_flux_up;
    void *band_flux_dn;
    void *band_flux_net;
    int64_t band_flux_ncol; /* columns of the band arrays' second dimension; 0 = ncol.  Larger: the arrays are a block of
                             * columns inside wider (nlev, band_flux_ncol, nbnd) arrays — how the library hands
                             * column ranges of them to shards and pipeline chunks */
    /* AllSkyRadiationWithClearSkyDiagnostics, src/api/update_fluxes.jl:39-65,101-128: the reference
     * solves twice (clear, then all-sky) and snapshots the first result.  When clear_flux_up is
     * non-NULL (clear_flux_dn / _net required with it; clear_flux_dn_dir for SW) and a cloud lookup
     * is passed to a TWO-STREAM solve, the same launch also carries the recurrences without the
     * cloud increment — gas optics, sources and aerosols are shared — and writes the clear-sky
     * fluxes here, same layout and metric scaling as flux_up/dn/net(/dn_dir).  Cannot be combined
     * with band_flux_* in one call. */
    void *clear_flux_up;
    void *clear_flux_dn;
    void *clear_flux_net;
    void *clear_flux_dn_dir;
    int64_t flux_ncol; /* RRTMGP_LAYOUT_NCOL_NLEV only: columns of the first dimension of the (clear_)flux_* arrays; 0 = ncol.
                        * Larger: the arrays are a block of columns inside wider (flux_ncol, nlev) arrays (shards, chunks) */
} rrtmgp_flux_out;

/* Per-call options. */
typedef struct rrtmgp_solve_opts {
    int32_t n_gauss_angles;      /* LW no-scattering: 1..4 (AngularDiscretizations.jl:34-63) */
    int32_t metric_mem;          /* mem kind of metric_scaling */
    const void *metric_scaling;  /* FT (nlev, ncol) or NULL; apply_metric_scaling!, Fluxes.jl:295-304 */
    uint64_t seed;               /* McICA stream key (see rrtmgp_hip_mcica_uniform) */
    int64_t col_offset;          /* global index of column 0 of this shard: keys the
                                    McICA stream so results do not depend on how
                                    columns are sharded across GPUs */
} rrtmgp_solve_opts;

/* Gray-atmosphere state, src/optics/gray_atmospheric_states.jl:100-128 */
typedef struct rrtmgp_gray_state {
    int32_t mem;
    int32_t otp_kind; /* 0 = GrayOpticalThicknessSchneider2004, 1 = ...OGorman2008 */
    int64_t ncol;
    int64_t nlay;
    const void *lat;   /* FT (ncol) degrees */
    const void *p_lay; /* FT (nlay, ncol) */
    const void *p_lev; /* FT (nlev, ncol) */
    const void *t_lay; /* FT (nlay, ncol) */
    const void *t_lev; /* FT (nlev, ncol) */
    const void *t_sfc; /* FT (ncol) */
    double otp[5];     /* Schneider: alpha, te, tt, dt ; OGorman: alpha, fl, tau_e, tau_p, tau_0 */
    double stefan;     /* RP.Stefan(param_set) */
} rrtmgp_gray_state;

/* RRTMGPParameters subset used on the device path, src/Parameters.jl:6-14 */
typedef struct rrtmgp_params {
    double grav, molmass_dryair, molmass_water, gas_constant, kappa_d, stefan, avogad;
} rrtmgp_params;

typedef struct rrtmgp_lookup rrtmgp_lookup;       /* opaque device-resident lookup */
typedef struct rrtmgp_workspace rrtmgp_workspace; /* opaque per-(ncol,nlay,FT) scratch */

/* ---- lifecycle ---------------------------------------------------------- */

/* Number of visible HIP devices, or RRTMGP_ENODEV. */
int rrtmgp_hip_device_count(void);

/* One-off table uploads; replace DA(...) in ext/lookup_constructors.jl:83 (LookUpLW),
 * :407 (LookUpSW), :727 (LookUpCld), :18 (LookUpAerosolMerra).
 *
 * LIMITS of this back end (each is checked and refused with RRTMGP_EUNSUPPORTED / RRTMGP_EINVAL and a message, never
 * silently truncated; rrtmgp-data v1.9 — 256 + 224 g-points, 16 + 14 bands, 9 x 60 x 14 table axes — is inside all of them):
 *   - n_gpt <= 256 per lookup (one lane per g-point, one workgroup of at most 4 wavefronts per column);
 *   - n_bnd <= 16 per lookup (the per-(layer, band) LDS records are laid out for 16 bands);
 *   - table axes n_eta, n_p_ref + 1, n_t_ref <= 255 each (indices travel packed in bytes);
 *   - a column's LDS records must fit the CU's 160 KB: about 590 layers in Float32, 250 in Float64 (fewer with per-band
 *     fluxes or the one-pass clear-sky diagnostic), reported by the solve that meets it;
 *   - ncol < 2^31 and nlay < 4096 per workspace; per-band fluxes need every band, padded to 16 g-points, to fit 256 lanes;
 *   - every re-laid-out table must be addressable with 32-bit byte offsets (4 GB per lookup). */
int rrtmgp_hip_gas_lookup_create(const rrtmgp_gas_lookup_desc *desc, int device, rrtmgp_lookup **out);
int rrtmgp_hip_cloud_lookup_create(const rrtmgp_cloud_lookup_desc *desc, int device, rrtmgp_lookup **out);
int rrtmgp_hip_aerosol_lookup_create(const rrtmgp_aerosol_lookup_desc *desc, int device, rrtmgp_lookup **out);
int rrtmgp_hip_lookup_destroy(rrtmgp_lookup *lk);

/* Scratch that the reference keeps in op/src/fluxb/state_cache/masks
 * (src/rte/RTE.jl:53,111,177,229); allocated once, zero allocation per solve
 * (update_fluxes.jl:215-218). */
int rrtmgp_hip_workspace_create(int device, int64_t ncol, int64_t nlay, int32_t ftype, rrtmgp_workspace **out);
int rrtmgp_hip_workspace_destroy(rrtmgp_workspace *ws);
/* Run this workspace's launches on an existing hipStream_t (e.g. torch's current
 * stream); NULL selects the HIP null (legacy default) stream.  Until this is called a
 * workspace launches on a private non-blocking stream it creates for itself. */
int rrtmgp_hip_workspace_set_stream(rrtmgp_workspace *ws, void *hip_stream);
/* Block until everything queued on the workspace stream has finished. */
int rrtmgp_hip_workspace_synchronize(rrtmgp_workspace *ws);
/* Milliseconds the solver kernel of the most recent LAUNCH of this workspace took on the device (HIP events around the
 * launch, on the stream it ran on); synchronizes.  After rrtmgp_hip_update_fluxes that is the step's LAST solver launch:
 * the shortwave kernel (on the workspace's second lane when a short step runs the two solvers side by side) — a per-kernel
 * breakdown of a fused step comes from two solve calls or from rocprofv3, not from this counter.  After a pipelined host
 * solve: the last chunk's kernel. */
int rrtmgp_hip_workspace_last_kernel_ms(rrtmgp_workspace *ws, double *ms);

/* ---- spectral solvers (K1-K4 of SURVEY.md §2.2) -------------------------- */

/* rte_lw_2stream_solve!(device::CUDADevice, flux, flux_lw, band_flux, src_lw, bcs_lw, op, as,
 *   state_cache, lookup_lw, lookup_lw_cld, lookup_lw_aero)  ext/cuda/rte_longwave_2stream.jl:48
 * (+ apply_metric_scaling!, src/rte/RTESolver.jl:140). cld / aero may be NULL. */
int rrtmgp_hip_rte_lw_2stream_solve(rrtmgp_workspace *ws, const rrtmgp_lookup *lookup_lw,
                                    const rrtmgp_lookup *lookup_lw_cld, const rrtmgp_lookup *lookup_lw_aero,
                                    const rrtmgp_atmos_state *as, const rrtmgp_lw_bcs *bcs,
                                    const rrtmgp_flux_out *flux, const rrtmgp_solve_opts *opts);

/* rte_lw_noscat_solve!(device::CUDADevice, ..., angle_disc, ...)  ext/cuda/rte_longwave_noscat.jl:54 */
int rrtmgp_hip_rte_lw_noscat_solve(rrtmgp_workspace *ws, const rrtmgp_lookup *lookup_lw,
                                   const rrtmgp_lookup *lookup_lw_cld, const rrtmgp_lookup *lookup_lw_aero,
                                   const rrtmgp_atmos_state *as, const rrtmgp_lw_bcs *bcs,
                                   const rrtmgp_flux_out *flux, const rrtmgp_solve_opts *opts);

/* rte_sw_2stream_solve!(device::CUDADevice, ...)  ext/cuda/rte_shortwave_2stream.jl:58 */
int rrtmgp_hip_rte_sw_2stream_solve(rrtmgp_workspace *ws, const rrtmgp_lookup *lookup_sw,
                                    const rrtmgp_lookup *lookup_sw_cld, const rrtmgp_lookup *lookup_sw_aero,
                                    const rrtmgp_atmos_state *as, const rrtmgp_sw_bcs *bcs,
                                    const rrtmgp_flux_out *flux, const rrtmgp_solve_opts *opts);

/* rte_sw_noscat_solve!(device::CUDADevice, ...)  ext/cuda/rte_shortwave_noscat.jl:54 */
int rrtmgp_hip_rte_sw_noscat_solve(rrtmgp_workspace *ws, const rrtmgp_lookup *lookup_sw,
                                   const rrtmgp_atmos_state *as, const rrtmgp_sw_bcs *bcs,
                                   const rrtmgp_flux_out *flux, const rrtmgp_solve_opts *opts);

/* ---- gray solvers (K5-K8) ------------------------------------------------ */

/* ext/cuda/rte_longwave_2stream.jl:1, rte_longwave_noscat.jl:1 */
int rrtmgp_hip_rte_lw_2stream_solve_gray(rrtmgp_workspace *ws, const rrtmgp_gray_state *as,
                                         const rrtmgp_lw_bcs *bcs, const rrtmgp_flux_out *flux,
                                         const rrtmgp_solve_opts *opts);
int rrtmgp_hip_rte_lw_noscat_solve_gray(rrtmgp_workspace *ws, const rrtmgp_gray_state *as,
                                        const rrtmgp_lw_bcs *bcs, const rrtmgp_flux_out *flux,
                                        const rrtmgp_solve_opts *opts);
/* ext/cuda/rte_shortwave_2stream.jl:1, rte_shortwave_noscat.jl:1 */
int rrtmgp_hip_rte_sw_2stream_solve_gray(rrtmgp_workspace *ws, const rrtmgp_gray_state *as,
                                         const rrtmgp_sw_bcs *bcs, const rrtmgp_flux_out *flux,
                                         const rrtmgp_solve_opts *opts);
int rrtmgp_hip_rte_sw_noscat_solve_gray(rrtmgp_workspace *ws, const rrtmgp_gray_state *as,
                                        const rrtmgp_sw_bcs *bcs, const rrtmgp_flux_out *flux,
                                        const rrtmgp_solve_opts *opts);

/* ---- state preparation (K9, K10) ----------------------------------------- */

/* A 2-D array argument exactly as the reference hands it to these device methods: a dense `Array`, or one of the
 * strided views the host code builds — `view(as.layerdata, i, :, :)` (src/optics/AtmosphericStates.jl:96-106: element
 * stride 4), `view(vmr.vmr, idx_h2o, :, :)` (src/api/grid_adaptation.jl:204: element stride ngas), the getters' domain
 * views `view(x, 1:n, :)` (src/api/getters.jl:42-43: column stride = rows of the parent).  Element (i, j) lives at
 * ((FT *)ptr)[i * stride0 + j * stride1]; strides are in ELEMENTS and are what Julia's `strides(a)` returns.  A dense
 * (n0, n1) array has stride0 = 1, stride1 = n0.  Both strides must be >= 1.  Nothing is copied on the caller's side:
 * host views are staged as the memory span they cover (views of one parent array share one upload; a written view is
 * staged in and out, so the parent's other elements come back unchanged), device views are used in place. */
typedef struct rrtmgp_view2d {
    void *ptr;
    int64_t stride0; /* elements between (i, j) and (i + 1, j) */
    int64_t stride1; /* elements between (i, j) and (i, j + 1) */
} rrtmgp_view2d;

/* The three calls below take their extents explicitly and check them against the workspace: ncol <= the workspace's
 * ncol (== for a multi-device workspace, whose shard ranges are fixed) and nlay <= its nlay — `heating_rate` passes the
 * DOMAIN layer count, one less than the solver's when there is an isothermal boundary layer (src/api/standalone.jl:106-122). */

/* compute_col_gas!(device::CUDADevice, p_lev, col_dry, param_set, vmr_h2o, lat)  ext/cuda/optics.jl:2
 * (caller: update_concentrations!, src/api/grid_adaptation.jl:278-292, which passes getview_col_dry(as) and
 * _vmr_h2o(as.vmr, idx_h2o)).  p_lev (nlay+1, ncol) -> col_dry (nlay, ncol); vmr_h2o (nlay, ncol) or NULL (a view whose
 * `ptr` is NULL also means "absent": a binding can always pass a struct); lat FT (ncol), dense, or NULL. */
int rrtmgp_hip_compute_col_gas(rrtmgp_workspace *ws, int32_t mem, int64_t ncol, int64_t nlay, const rrtmgp_view2d *p_lev,
                               const rrtmgp_view2d *col_dry, const rrtmgp_params *params, const rrtmgp_view2d *vmr_h2o,
                               const void *lat);

/* compute_relative_humidity!(device::CUDADevice, rh, p_lay, t_lay, param_set, vmr_h2o)  ext/cuda/optics.jl:35
 * (the reference's drivers pass rows 4, 2, 3 of layerdata: test/read_clear_sky.jl:162-169); all (nlay, ncol). */
int rrtmgp_hip_compute_relative_humidity(rrtmgp_workspace *ws, int32_t mem, int64_t ncol, int64_t nlay,
                                         const rrtmgp_view2d *rh, const rrtmgp_view2d *p_lay, const rrtmgp_view2d *t_lay,
                                         const rrtmgp_params *params, const rrtmgp_view2d *vmr_h2o);

/* prepare_atmosphere!(s::RRTMGPSolver)  src/api/update_fluxes.jl:252-281 — the whole
 * preparation cascade of src/api/grid_adaptation.jl, in place, one kernel, columns independent:
 *   RRTMGP_PREP_INTERPOLATE  interpolate_levels! (:73-113): interior faces and the top face by
 *                            `interpolation`, the bottom face by `bottom_extrapolation`
 *                            (formulas: src/api/interpolation.jl:176-252);
 *   RRTMGP_PREP_ISOTHERMAL   add_isothermal_boundary_layer! (:137-173): the state's LAST layer
 *                            is the extra layer (interpolation then covers nlay - 1 layers);
 *   RRTMGP_PREP_CLIP         clip! (:232-258): vmr_h2o >= 0, p >= p_min, T in [t_min, t_max]
 *                            (t_min > t_max skips the temperature clamp);
 *   RRTMGP_PREP_COL_DRY      update_concentrations! (:278-292) = compute_col_gas!.
 * Every array of `as` named below is read AND written (the const in rrtmgp_atmos_state is
 * cast away): layerdata, p_lev, t_lev, vmr_h2o / vmr_o3 / vmr, cloud and aerosol inputs. */
#define RRTMGP_PREP_INTERPOLATE 1
#define RRTMGP_PREP_ISOTHERMAL 2
#define RRTMGP_PREP_CLIP 4
#define RRTMGP_PREP_COL_DRY 8
#define RRTMGP_PREP_ALL 15
/* optional, NOT part of prepare_atmosphere! (the reference's tests call compute_relative_humidity! themselves,
 * test/read_clear_sky.jl:162-169): also refresh layerdata row 4 (rel_hum) from the clipped p_lay, t_lay, vmr_h2o in the
 * same launch, compute_relative_humidity_kernel! src/optics/gas_optics.jl:58-80.  AtmosphericState only. */
#define RRTMGP_PREP_REL_HUM 16

#define RRTMGP_INTERP_NONE 0            /* NoInterpolation, interpolation.jl:47 */
#define RRTMGP_INTERP_ARITHMETIC_MEAN 1 /* :55 */
#define RRTMGP_INTERP_GEOMETRIC_MEAN 2  /* :64 */
#define RRTMGP_INTERP_UNIFORM_Z 3       /* :73 */
#define RRTMGP_INTERP_UNIFORM_P 4       /* :82 */
#define RRTMGP_INTERP_BEST_FIT 5        /* :90, needs center_z / face_z */

#define RRTMGP_BOTTOM_SAME_AS_INTERPOLATION 0 /* :119 */
#define RRTMGP_BOTTOM_USE_SURFACE_TEMP 1      /* :128 */
#define RRTMGP_BOTTOM_HYDROSTATIC 2           /* :136, needs center_z / face_z */

typedef struct rrtmgp_prepare_opts {
    int32_t steps;                     /* OR of RRTMGP_PREP_* */
    int32_t interpolation;             /* RRTMGP_INTERP_* */
    int32_t bottom_extrapolation;      /* RRTMGP_BOTTOM_* */
    int32_t isothermal_boundary_layer; /* grid has the extra top layer (also shortens the interpolation) */
    int32_t z_mem;                     /* mem kind of center_z / face_z */
    int32_t idx_h2o;                   /* 1-based row of h2o in a FULL vmr */
    const void *center_z;              /* FT (nlay, ncol) or NULL */
    const void *face_z;                /* FT (nlev, ncol) or NULL */
    double p_min, t_min, t_max;
} rrtmgp_prepare_opts;

int rrtmgp_hip_prepare_atmosphere(rrtmgp_workspace *ws, const rrtmgp_atmos_state *as, const rrtmgp_params *params,
                                  const rrtmgp_prepare_opts *opts);
/* Gray state: interpolation, isothermal layer (p, T only) and the pressure clip
 * (grid_adaptation.jl:147-156, 215-227); col_dry does not exist. */
int rrtmgp_hip_prepare_atmosphere_gray(rrtmgp_workspace *ws, const rrtmgp_gray_state *as,
                                       const rrtmgp_params *params, const rrtmgp_prepare_opts *opts);

/* compute_gray_heating_rate!(device::CUDADevice, hr_lay, p_lev, ncol, nlay, flux_net, cp_d_, grav_)
 * ext/cuda/gray_atmosphere.jl:42-61 (body src/optics/GrayAtmosphere.jl:152-167; caller `heating_rate`,
 * src/api/standalone.jl:106-122, which passes the domain views level_pressure(s) / net_flux(s)):
 * hr_lay(nlay, ncol) = grav (F_net[k+1] - F_net[k]) / (p_lev[k+1] - p_lev[k]) / cp_d with flux_net, p_lev (nlay+1, ncol). */
int rrtmgp_hip_compute_gray_heating_rate(rrtmgp_workspace *ws, int32_t mem, int64_t ncol, int64_t nlay,
                                         const rrtmgp_view2d *hr_lay, const rrtmgp_view2d *p_lev,
                                         const rrtmgp_view2d *flux_net, double cp_d, double grav);

/* ---- the whole radiation step (Layer 2): update_fluxes!(s::RRTMGPSolver) ---------------------------------------
 *
 * update_fluxes!(s, seedval)  src/api/update_fluxes.jl:223-233 =
 *     prepare_atmosphere!(s) (:252-281) -> update_lw_fluxes!(s) (:12-65) -> update_sw_fluxes!(s) (:74-128)
 *     -> update_net_fluxes!(s) (:165-194),
 * for the spectral radiation methods (ClearSkyRadiation, AllSkyRadiation, AllSkyRadiationWithClearSkyDiagnostics), in ONE
 * call: the caller's state crosses to the device once (the reference's extension path stages it for every solve: LW, SW,
 * twice each with diagnostics, plus in and out for compute_col_gas!), then on the workspace stream
 *     [prepare kernel]  ->  LW solve  ->  SW solve  ->  net = lw_net + sw_net,
 * and what comes back is the fluxes, the diagnostics, and — only when `prepare` is given — the prepared state arrays
 * (layerdata, p_lev, t_lev, vmr_h2o or the full vmr; the cloud / aerosol inputs and vmr_o3 as well when the isothermal
 * boundary layer is filled).  Large host-array steps run as a pipeline of column chunks (uploads of chunk c + 1 and
 * downloads of chunk c - 1 overlap the kernels of chunk c); multi-device workspaces shard the columns; device arrays are
 * used in place.  The bits are those of the separate calls.
 *
 *  - `flux_lw` / `flux_sw` are what the step leaves for the getters: a Julia host passes the arrays of
 *    `s.presented_flux_lw` / `_sw` (FluxPresentation, (nlev, ncol): RRTMGP_LAYOUT_NLEV_NCOL), which makes the
 *    update_presentation! copies (update_fluxes.jl:15,77) unnecessary.  Their `clear_flux_*` slots take the arrays of
 *    `s.clear_flux_lw` / `_sw` (AllSkyRadiationWithClearSkyDiagnostics): two-stream solvers fill them in the same launch
 *    as the all-sky fluxes (the reference solves twice, :39-65, :101-128); a no-scattering LW solver, or a step that also
 *    keeps per-band fluxes, runs the cloudless solve first on the staged state, as the reference does.
 *  - `net_flux` = flux_lw.flux_net + flux_sw.flux_net and `clear_net_flux` = the clear-sky pair, both FT (nlev, ncol)
 *    whatever the layout of the flux arrays (transpose_sum_into!, src/optics/Fluxes.jl:407-424); either may be NULL.
 *    `mem` of these two arrays is `flux_lw->mem`.
 *  - `lookup_*_cld` NULL = no clouds (ClearSkyRadiation); `lookup_*_aero` NULL = `aerosol_radiation = false`.
 *  - `prepare` NULL = the state is already prepared (no kernel, nothing of the state is copied back).
 *  - `opts->seed` keys the McICA streams of both solves (LW and SW draw from separate streams, `is_sw`);
 *    `opts->n_gauss_angles` applies to a no-scattering LW solver; `opts->metric_scaling` = deep_atmosphere_inverse_scaling. */
#define RRTMGP_LW_TWOSTREAM 1 /* TwoStreamLWRTE */
#define RRTMGP_LW_NOSCAT 0    /* NoScatLWRTE */
typedef struct rrtmgp_update_fluxes_args {
    const rrtmgp_lookup *lookup_lw, *lookup_sw;         /* s.lookups.lookup_lw / lookup_sw */
    const rrtmgp_lookup *lookup_lw_cld, *lookup_sw_cld; /* or NULL */
    const rrtmgp_lookup *lookup_lw_aero, *lookup_sw_aero; /* or NULL */
    const rrtmgp_atmos_state *as;   /* s.as */
    const rrtmgp_lw_bcs *bcs_lw;    /* s.lws.bcs */
    const rrtmgp_sw_bcs *bcs_sw;    /* s.sws.bcs */
    const rrtmgp_flux_out *flux_lw; /* flux_dn_dir NULL */
    const rrtmgp_flux_out *flux_sw;
    void *net_flux;                 /* FT (nlev, ncol) or NULL: s.net_flux_buffer */
    void *clear_net_flux;           /* FT (nlev, ncol) or NULL: s.clear_net_flux_buffer */
    const rrtmgp_params *params;    /* needed with `prepare` */
    const rrtmgp_prepare_opts *prepare; /* or NULL */
    const rrtmgp_solve_opts *opts;  /* or NULL */
    int32_t lw_solver;              /* RRTMGP_LW_TWOSTREAM / RRTMGP_LW_NOSCAT; the SW solver is two-stream (solver.jl:176-182) */
    int32_t _pad;
} rrtmgp_update_fluxes_args;

int rrtmgp_hip_update_fluxes(rrtmgp_workspace *ws, const rrtmgp_update_fluxes_args *args);

/* update_fluxes!(s, seedval) for GrayRadiation (src/api/update_fluxes.jl:223-233 with :19-23 / :81-85, the gray methods of
 * update_lw_fluxes! / update_sw_fluxes!), in ONE call:
 *     prepare_atmosphere!(s) (gray form: interpolation, isothermal layer, pressure clip; src/api/grid_adaptation.jl:147-156,
 *     215-227)  ->  gray LW solve (K5 / K6)  ->  gray SW solve (K7 / K8)  ->  net_flux = lw_net + sw_net,
 * with the gray state staged once.  The reference runs this through its generic Layer-2 path, whose presentation copies and
 * net sum are array broadcasts (Fluxes.jl:408,424): on device-resident arrays of a host language without a GPU array package
 * (ext/RRTMGPHIPExt.jl `HIPArray`) there is nothing to run those, so the step is offered whole — `flux_lw` / `flux_sw` take
 * the presentation arrays ((nlev, ncol), RRTMGP_LAYOUT_NLEV_NCOL) exactly as in rrtmgp_hip_update_fluxes.
 *  - `lw_solver`: RRTMGP_LW_TWOSTREAM / RRTMGP_LW_NOSCAT; `sw_twostream`: 1 = TwoStreamSWRTE, 0 = NoScatSWRTE (gray radiation
 *    may pair a non-scattering shortwave solver, src/api/solver.jl:110-112).
 *  - `net_flux` FT (nlev, ncol) or NULL, memory kind of `flux_lw`; `prepare` NULL = the state is already prepared;
 *    `opts->metric_scaling` = deep_atmosphere_inverse_scaling (n_gauss_angles must be <= 1: gray radiation uses one angle).
 * The bits are those of the separate calls (rrtmgp_hip_prepare_atmosphere_gray, rrtmgp_hip_rte_*_solve_gray + a host sum).
 * Multi-device workspaces shard the columns; host arrays are staged, device arrays used in place. */
typedef struct rrtmgp_update_fluxes_gray_args {
    const rrtmgp_gray_state *as;    /* s.as */
    const rrtmgp_lw_bcs *bcs_lw;    /* s.lws.bcs: sfc_emis (1, ncol), inc_flux (ncol) or NULL */
    const rrtmgp_sw_bcs *bcs_sw;    /* s.sws.bcs: albedos (1, ncol) */
    const rrtmgp_flux_out *flux_lw; /* flux_dn_dir NULL */
    const rrtmgp_flux_out *flux_sw;
    void *net_flux;                 /* FT (nlev, ncol) or NULL */
    const rrtmgp_params *params;    /* needed with `prepare` */
    const rrtmgp_prepare_opts *prepare; /* or NULL */
    const rrtmgp_solve_opts *opts;  /* or NULL */
    int32_t lw_solver;
    int32_t sw_twostream;
} rrtmgp_update_fluxes_gray_args;

int rrtmgp_hip_update_fluxes_gray(rrtmgp_workspace *ws, const rrtmgp_update_fluxes_gray_args *args);

/* Bytes this workspace (all shards) has moved host -> device and device -> host since it was created: what a host-array
 * call costs on PCIe (bench.py reports bytes per column of the Layer-2 step; tests count the staging of strided views). */
int rrtmgp_hip_workspace_transfer_bytes(const rrtmgp_workspace *ws, uint64_t *h2d, uint64_t *d2h);

/* ---- several GPUs from ONE host process (SURVEY.md §8(b) "Threading", §8(e)) ---------------
 *
 * The reference shards columns over devices above its API (one ClimaComms context per rank);
 * a Julia host that owns all columns in one process gets the same contiguous-range sharding
 * behind the C ABI instead:
 *
 *  - `*_lookup_create_multi` uploads one replica of the lookup per DISTINCT device of
 *    `device_ids`; the returned handle stands for all of them and is destroyed by
 *    rrtmgp_hip_lookup_destroy.
 *  - `rrtmgp_hip_workspace_create_multi` creates one shard per entry of `device_ids` (an id
 *    may repeat: several shards, each with its own stream, on one GPU).  Shard s owns the
 *    contiguous global columns [s*ncol/ndev, (s+1)*ncol/ndev); since ncol is the slowest
 *    dimension of every state / boundary / flux array, each shard's slab is contiguous.
 *  - Every solver / preparation entry point accepts such a workspace: the call runs the
 *    shards concurrently (one host thread + one stream per shard), sets `col_offset` per
 *    shard so that the McICA stream stays keyed by the global column, and returns when every
 *    shard's results are in the caller's arrays.  The bits are those of a single launch.
 *  - Arrays must be host memory (RRTMGP_MEM_HOST) unless every shard is on the same device
 *    as the pointers.  The arrays whose column ranges are not contiguous slabs — LwBCs.inc_flux
 *    (ncol fastest) and the per-band fluxes (ncol in the middle) — are handed to the shards as
 *    strided blocks (inc_flux_ld, band_flux_ncol), and so are fluxes in the RRTMGP_LAYOUT_NCOL_NLEV
 *    layout (flux_ncol).
 *  - Host arrays cross PCIe as true asynchronous DMA when they are page-locked: rrtmgp_hip_host_register below.
 *  - Device arrays (all shards on the device that holds them): the call waits for work queued on that device before
 *    the shards start (they run on private streams that nothing else orders behind the caller's producer stream) and
 *    returns when every shard's kernels have finished.
 */
int rrtmgp_hip_gas_lookup_create_multi(const rrtmgp_gas_lookup_desc *desc, const int32_t *device_ids, int ndev,
                                       rrtmgp_lookup **out);
int rrtmgp_hip_cloud_lookup_create_multi(const rrtmgp_cloud_lookup_desc *desc, const int32_t *device_ids, int ndev,
                                         rrtmgp_lookup **out);
int rrtmgp_hip_aerosol_lookup_create_multi(const rrtmgp_aerosol_lookup_desc *desc, const int32_t *device_ids, int ndev,
                                           rrtmgp_lookup **out);
int rrtmgp_hip_workspace_create_multi(const int32_t *device_ids, int ndev, int64_t ncol, int64_t nlay, int32_t ftype,
                                      rrtmgp_workspace **out);
/* The CPUs next to a GPU's PCIe root: the shard worker of that GPU is bound to them (staging copies of one process at 8 GPUs
 * only reach their rate from the memory controllers next to each GPU; RRTMGP_HIP_NO_NUMA_BIND=1 leaves the threads unbound).
 * Reads <RRTMGP_HIP_SYSFS_ROOT or /sys>/bus/pci/devices/<pci_bus_id, lower-cased>/local_cpulist ("0-31,64-95"); writes up to
 * `cap` CPU numbers and returns how many the list names, or a negative error when there is no such file.  Host logic only:
 * callable without a GPU (tests/test_sharding.py drives it against a fake sysfs tree). */
int rrtmgp_hip_local_cpus(const char *pci_bus_id, int32_t *cpus, int cap);

/* Number of shards of a workspace (1 for a single-device workspace). */
int rrtmgp_hip_workspace_shards(const rrtmgp_workspace *ws);

/* ---- page-locked host arrays: explicit lifetime ---------------------------------------------------
 *
 * With RRTMGP_MEM_HOST every solve copies state in and fluxes out.  Those copies are asynchronous DMA (and the pipelined
 * host path overlaps them with the kernels) only from / to page-locked memory.  The OWNER of a host array registers it
 * once, keeps it registered while it hands it to solves, and unregisters it before the memory is freed; a binding does
 * that from the array's owner and finalizer (ext/RRTMGPHIPExt.jl `pin!`, rrtmgp.jl_amd/states.py `pin_host_array`).
 *   - register is reference counted per exact (ptr, bytes) range; a range that overlaps a live registration of other
 *     extents is refused (RRTMGP_EINVAL);
 *   - any workspace, shard or pipeline chunk uses a registered range (or a sub-range of it) as it is; the library never
 *     releases an explicit registration on its own, and unregister fails while a running solve is using the range;
 *   - hipHostRegister locks whole pages: the array should own its pages.  glibc mmaps every allocation of at least 32 MB
 *     (and most above 128 KB); smaller heap arrays share their first / last page with other objects, and a page lock the
 *     HIP runtime takes and drops on such a neighbour for a pageable copy unmaps the shared page under the registration
 *     (GPU memory access fault) — the bindings therefore register arrays of at least 32 MB only.
 * Unregistered arrays work too (pageable copies).  RRTMGP_HIP_AUTO_HOST_REGISTER=1 makes the library itself register
 * every unregistered host array of at least 32 MB (RRTMGP_HIP_HOST_REGISTER_MIN_BYTES) the first time a solve sees it and
 * release it when other memory shows up on its pages or 64 registration passes did not touch it: a heuristic for callers
 * that cannot manage lifetimes (an array freed and re-allocated at the same address with the same size cannot be told
 * from the old one), hence opt-in. */
int rrtmgp_hip_host_register(void *ptr, size_t bytes);
int rrtmgp_hip_host_unregister(void *ptr);
/* Live registrations (explicit and automatic), for tests. */
int rrtmgp_hip_host_registered_count(void);

/* ---- caller-owned device arrays ------------------------------------------------------------- */

/* What a host language needs to keep its arrays in HBM without a GPU array package of its own (the Julia glue's
 * `HIPArray`, ext/RRTMGPHIPExt.jl; the reference gets this from CUDA.jl's CuArray, ext/RRTMGPCUDAExt.jl:1-66): allocation,
 * release, blocking copies (ordered behind everything queued on the device) and byte fills on device `device`.  A pointer
 * from rrtmgp_hip_device_malloc is what the descriptors take with `mem = RRTMGP_MEM_DEVICE`.  Not part of the library's own
 * allocation accounting (rrtmgp_hip_allocation_counts).  The copies and fills synchronise the WHOLE device (that is what orders
 * them behind solves queued on any workspace's private stream): meant for setup, checkpoints and getters between radiation
 * steps, not for a per-step path - a host that drives several workspaces concurrently serialises them with every such call. */
enum { RRTMGP_COPY_H2D = 1, RRTMGP_COPY_D2H = 2, RRTMGP_COPY_D2D = 3 };
int rrtmgp_hip_device_malloc(int device, size_t bytes, void **out);
int rrtmgp_hip_device_free(int device, void *ptr);
int rrtmgp_hip_memcpy(int device, void *dst, const void *src, size_t bytes, int32_t kind);
int rrtmgp_hip_memset(int device, void *dst, int32_t byte_value, size_t bytes);

/* ---- allocation accounting (zero-allocation contract, update_fluxes.jl:215-218) ------------ */

/* Device allocations (hipMalloc) and page-locked host memory events (hipHostRegister of caller arrays,
 * hipHostMalloc of the small-solve bounce buffer) the library has made since it was loaded, all workspaces and lookups together.  A warm solve must not change them:
 * tests/test_abi_contracts.py. */
int rrtmgp_hip_allocation_counts(int64_t *device_allocs, int64_t *device_frees, int64_t *host_registrations);

/* ---- McICA stream -------------------------------------------------------- */

/* The reference draws Random.rand() (Float64) per (g-point, column) inside
 * build_cloud_mask! (src/optics/cloud_optics.jl:279,291) and documents the
 * stream as device-dependent and not reproducible.  This back end defines a
 * counter-based stream instead: draw number `draw` of g-point `igpt` (1-based)
 * of global column `gcol` (1-based) in band set `is_sw` under `seed` is
 * rrtmgp_hip_mcica_uniform(...) in [0, 1).  Host-callable so tests can pin it. */
double rrtmgp_hip_mcica_uniform(uint64_t seed, int64_t gcol, int64_t igpt, int32_t is_sw, int32_t draw);

/* ---- device math forms ---------------------------------------------------- */

/* The per-g-point loops do not call libm or the compiler's `/`: they use the forms of rrtmgp.jl_amd/csrc/device.h (v_exp_f32 /
 * v_rcp_f32 / v_rsq_f32 based).  This entry evaluates one of them element-wise on HOST arrays of `n` values of `ftype` (`y`
 * only for the quotients, else NULL) so that a test can state their accuracy against a wider type: the reference computes
 * with Julia's exp (< 1 ulp) and IEEE `/` and sqrt (docs/src/precision.md).  Measured maxima: tests/test_primitives.py. */
enum {
    RRTMGP_PRIM_EXP_NEG = 0,      /* e^-x, x >= 0: transmissivities, the direct beam */
    RRTMGP_PRIM_EXP_PAIR_E1 = 1,  /* e^-x of the two-stream pair (longwave_2stream.jl:167, shortwave_2stream.jl:204) */
    RRTMGP_PRIM_EXP_PAIR_OM1 = 2, /* 1 - e^-x = -expm1(-x) of the same pair */
    RRTMGP_PRIM_RCP = 3,          /* 1 / x */
    RRTMGP_PRIM_DIV = 4,          /* x / y */
    RRTMGP_PRIM_SQRT_POS = 5,     /* sqrt(x), x positive and normal */
    RRTMGP_PRIM_IEEE_DIV = 6      /* x / y of increment_2stream (optics_utils.jl:189-223): correctly rounded in every build */
};
int rrtmgp_hip_eval_primitive(int device, int32_t op, int32_t ftype, const void *x, const void *y, void *out, int64_t n);

/* ---- diagnostics ---------------------------------------------------------- */

/* Copies the calling thread's last error message (NUL-terminated) into buf. */
int rrtmgp_hip_last_error(char *buf, size_t n);
/* "major.minor.patch", followed by " [flags]" when the library was not built as shipped */
const char *rrtmgp_hip_version(void);
/* The compile-time switches this library was built with, space separated: "" for the shipped build (IEEE-accurate
 * Float32 forms), "RR_FAST_F32" for the raw-instruction Float32 build (`make fast`); anything else (RR_EXP_*, tuning values) marks an experimental build whose results may be
 * wrong by construction (rrtmgp.jl_amd/csrc/variants.h). */
const char *rrtmgp_hip_build_flags(void);
/* sizeof() of ABI struct number `which` as compiled into the library (0 minor_desc,
 * 1 gas_lookup_desc, 2 cloud_lookup_desc, 3 aerosol_lookup_desc, 4 atmos_state, 5 lw_bcs,
 * 6 sw_bcs, 7 flux_out, 8 solve_opts, 9 gray_state, 10 params, 11 prepare_opts, 12 view2d, 13 update_fluxes_args,
 * 14 update_fluxes_gray_args); -1 otherwise.  Lets a
 * foreign-language binding verify its struct mirror at load time. */
int rrtmgp_hip_abi_sizeof(int which);

#ifdef __cplusplus
}
#endif
#endif /* RRTMGP_HIP_H */
