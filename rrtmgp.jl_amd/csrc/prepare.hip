// prepare_atmosphere! on the device: level interpolation / extrapolation, the isothermal
// boundary layer, clipping and col_dry in ONE launch (reference: four to nine host/broadcast
// passes over (nlay, ncol) arrays, src/api/grid_adaptation.jl:73-292, interpolation.jl:148-252,
// update_fluxes.jl:252-281).  One 64-lane workgroup per column, lanes over levels; the steps of
// a column are separated by workgroup barriers because each reads what the previous one wrote.
#include "common.h"
#include "device.h"

namespace rrtmgp {

template <typename FT>
__device__ __forceinline__ FT uniform_z_p(FT T, FT p1, FT T1, FT p2, FT T2) {  // interpolation.jl:155-156
    return T1 == T2 ? m_sqrt(p1 * p2) : p1 * m_pow(p2 / p1, m_log(T / T1) / m_log(T2 / T1));
}
template <typename FT>
__device__ __forceinline__ FT best_fit_p(FT T, FT z, FT p1, FT T1, FT z1, FT p2, FT T2, FT z2) {  // :164-166
    return T1 == T2 ? p1 * m_pow(p2 / p1, (z - z1) / (z2 - z1)) : p1 * m_pow(p2 / p1, m_log(T / T1) / m_log(T2 / T1));
}

// interp! (interpolation.jl:176-196)
template <typename FT>
__device__ void interp_face(int mode, FT &p, FT &T, FT z, FT pd, FT Td, FT zd, FT pu, FT Tu, FT zu) {
    switch (mode) {
        case RRTMGP_INTERP_ARITHMETIC_MEAN: T = (Td + Tu) / FT(2); p = (pd + pu) / FT(2); break;
        case RRTMGP_INTERP_GEOMETRIC_MEAN: T = m_sqrt(Td * Tu); p = m_sqrt(pd * pu); break;
        case RRTMGP_INTERP_UNIFORM_Z: T = (Td + Tu) / FT(2); p = uniform_z_p(T, pd, Td, pu, Tu); break;
        case RRTMGP_INTERP_UNIFORM_P:
            p = (pd + pu) / FT(2);
            T = Td * m_pow(Tu / Td, m_log(p / pd) / m_log(pu / pd));
            break;
        default:
            T = Td + (Tu - Td) * (z - zd) / (zu - zd);
            p = best_fit_p(T, z, pd, Td, zd, pu, Tu, zu);
    }
}

// extrap! (interpolation.jl:206-252); mode >= 100 are the bottom-only schemes
template <typename FT>
__device__ void extrap_face(int mode, FT &p, FT &T, FT z, FT p1, FT T1, FT z1, FT p2, FT T2, FT z2, FT Ts, FT R,
                            FT cp, FT g) {
    switch (mode) {
        case RRTMGP_INTERP_ARITHMETIC_MEAN: T = (FT(3) * T1 - T2) / FT(2); p = (FT(3) * p1 - p2) / FT(2); break;
        case RRTMGP_INTERP_GEOMETRIC_MEAN: T = m_sqrt(T1 * T1 * T1 / T2); p = m_sqrt(p1 * p1 * p1 / p2); break;
        case RRTMGP_INTERP_UNIFORM_Z: T = (FT(3) * T1 - T2) / FT(2); p = uniform_z_p(T, p1, T1, p2, T2); break;
        case RRTMGP_INTERP_UNIFORM_P:
            p = (FT(3) * p1 - p2) / FT(2);
            T = T1 * m_pow(T2 / T1, m_log(p / p1) / m_log(p2 / p1));
            break;
        case RRTMGP_INTERP_BEST_FIT:
            T = T1 + (T2 - T1) * (z - z1) / (z2 - z1);
            p = best_fit_p(T, z, p1, T1, z1, p2, T2, z2);
            break;
        case 100 + RRTMGP_BOTTOM_USE_SURFACE_TEMP: T = Ts; p = p1 * m_pow(T / T1, cp / R); break;
        default:
            T = T1 + g / cp * (z1 - z);
            p = p1 * m_pow(T / T1, cp / R);
    }
}

template <typename FT>
__global__ void __launch_bounds__(64) prepare_kernel(const PrepView<FT> v, const PrepArgs<FT> a) {
    const int col = blockIdx.x, lane = threadIdx.x;
    const int nlay_all = v.nlay, nlev_all = nlay_all + 1, ls = v.ls;
    const size_t lay0 = (size_t)nlay_all * col, lev0 = (size_t)nlev_all * col;
    FT *p_lay = v.p_lay + ls * lay0, *t_lay = v.t_lay + ls * lay0;
    FT *p_lev = v.p_lev + lev0, *t_lev = v.t_lev + lev0;
    const FT *zc = v.center_z ? v.center_z + lay0 : nullptr, *zf = v.face_z ? v.face_z + lev0 : nullptr;
    auto ZC = [&](int k) { return zc ? zc[k] : FT(0); };
    auto ZF = [&](int k) { return zf ? zf[k] : FT(0); };

    if ((a.steps & RRTMGP_PREP_INTERPOLATE) && a.interpolation != RRTMGP_INTERP_NONE) {
        const int nlay = nlay_all - (a.iso ? 1 : 0);   // interpolate_levels!, grid_adaptation.jl:94
        const FT ts = v.t_sfc[col];
        for (int k = lane; k <= nlay; k += 64) {       // level k (0-based) sits between layers k-1 and k
            FT p, T;
            if (k == 0)
                extrap_face(a.bottom_mode, p, T, ZF(0), p_lay[0], t_lay[0], ZC(0), p_lay[ls], t_lay[ls], ZC(1), ts, a.R,
                            a.cp, a.g);
            else if (k == nlay)
                extrap_face(a.interpolation, p, T, ZF(nlay), p_lay[ls * (nlay - 1)], t_lay[ls * (nlay - 1)], ZC(nlay - 1),
                            p_lay[ls * (nlay - 2)], t_lay[ls * (nlay - 2)], ZC(nlay - 2), ts, a.R, a.cp, a.g);
            else
                interp_face(a.interpolation, p, T, ZF(k), p_lay[ls * (k - 1)], t_lay[ls * (k - 1)], ZC(k - 1),
                            p_lay[ls * k], t_lay[ls * k], ZC(k));
            p_lev[k] = p;
            t_lev[k] = T;
        }
        __syncthreads();
    }
    if ((a.steps & RRTMGP_PREP_ISOTHERMAL) && a.iso) {  // add_isothermal_boundary_layer!, :137-173
        const int e = nlay_all - 1;                     // the extra layer; its upper level is e + 1
        if (lane == 0) {
            p_lay[ls * e] = (p_lev[e] + a.p_min) / FT(2);
            p_lev[e + 1] = a.p_min;
            t_lay[ls * e] = t_lev[e];
            t_lev[e + 1] = t_lev[e];
            if (v.rel_hum) v.rel_hum[ls * (lay0 + e)] = v.rel_hum[ls * (lay0 + e - 1)];
            if (!v.vmr_full) {
                if (v.vmr_h2o) v.vmr_h2o[lay0 + e] = v.vmr_h2o[lay0 + e - 1];
                if (v.vmr_o3) v.vmr_o3[lay0 + e] = v.vmr_o3[lay0 + e - 1];
            }
            for (int i = 0; i < 5; i++)
                if (v.cld[i]) v.cld[i][lay0 + e] = v.cld[i][lay0 + e - 1];
        }
        if (v.vmr_full)
            for (int ig = lane; ig < v.ngas; ig += 64)
                v.vmr_full[ig + (size_t)v.ngas * (lay0 + e)] = v.vmr_full[ig + (size_t)v.ngas * (lay0 + e - 1)];
        for (int i = 0; i < 2; i++)
            if (v.aero[i] && lane < RRTMGP_N_AEROSOLS)
                v.aero[i][lane + (size_t)RRTMGP_N_AEROSOLS * (lay0 + e)] =
                    v.aero[i][lane + (size_t)RRTMGP_N_AEROSOLS * (lay0 + e - 1)];
        __syncthreads();
    }
    if (a.steps & RRTMGP_PREP_CLIP) {  // clip!, :215-258 (gray: pressures only)
        // Julia's max(x, lo) and clamp(x, lo, hi) keep a NaN x (a face extrapolated to a negative pressure gives NaN through
        // log, interpolation.jl:228): a bad input must surface as NaN fluxes, not as a value clamped to a table bound
        auto jl_max = [](FT x, FT lo) { return x != x ? x : (x > lo ? x : lo); };
        auto jl_clamp = [](FT x, FT lo, FT hi) { return x > hi ? hi : (x < lo ? lo : x); };
        for (int k = lane; k < nlev_all; k += 64) {
            if (k < nlay_all) {
                if (v.vmr_h2o) {
                    FT &h = v.vmr_h2o[(size_t)v.hs * (lay0 + k)];
                    h = jl_max(h, FT(0));
                }
                p_lay[ls * k] = jl_max(p_lay[ls * k], a.p_min);
                if (a.clamp_t) t_lay[ls * k] = jl_clamp(t_lay[ls * k], a.t_min, a.t_max);
            }
            p_lev[k] = jl_max(p_lev[k], a.p_min);
            if (a.clamp_t) t_lev[k] = jl_clamp(t_lev[k], a.t_min, a.t_max);
        }
        __syncthreads();
    }
    if ((a.steps & RRTMGP_PREP_COL_DRY) && v.col_dry) {  // compute_col_gas!, src/optics/gas_optics.jl:16-47
        const FT g0 = v.lat ? a.g - FT(0.02586) * m_cos(FT(2) * Num<FT>::pi() * v.lat[col] / FT(180)) : a.g;
        for (int k = lane; k < nlay_all; k += 64) {
            const FT dp = p_lev[k] - p_lev[k + 1];
            const FT h2o = v.vmr_h2o ? v.vmr_h2o[(size_t)v.hs * (lay0 + k)] : FT(0);
            const FT m_air = (a.mol_m_dry + a.mol_m_h2o * h2o);
            v.col_dry[ls * (lay0 + k)] = (dp * a.avogadro / (FT(100 * 100) * m_air * g0));
        }
    }
    if ((a.steps & RRTMGP_PREP_REL_HUM) && v.rel_hum && v.vmr_h2o) {  // compute_relative_humidity_kernel!, gas_optics.jl:58-80
        const FT mwd = a.mol_m_h2o / a.mol_m_dry;
        for (int k = lane; k < nlay_all; k += 64) {
            const FT mmr = v.vmr_h2o[(size_t)v.hs * (lay0 + k)] * mwd;
            const FT q_tmp = m_max(FT(1e-7), mmr / (FT(1) + mmr));
            const FT t = t_lay[ls * k];
            const FT es = m_exp((FT(17.67) * (t - FT(273.16))) / (t - FT(29.65)));
            v.rel_hum[ls * (lay0 + k)] = m_max(FT(0.01) * (FT(0.263) * p_lay[ls * k] * q_tmp) / es, FT(0));
        }
    }
}

template <typename FT>
int launch_prepare(rrtmgp_workspace *ws, const PrepView<FT> &v, const rrtmgp_params &ps, const rrtmgp_prepare_opts &o,
                   bool gray) {
    PrepArgs<FT> a{};
    a.steps = o.steps;
    a.interpolation = o.interpolation;
    a.iso = o.isothermal_boundary_layer != 0;
    if ((a.steps & RRTMGP_PREP_INTERPOLATE) && a.interpolation != RRTMGP_INTERP_NONE) {
        RR_CHECK(a.interpolation >= RRTMGP_INTERP_ARITHMETIC_MEAN && a.interpolation <= RRTMGP_INTERP_BEST_FIT,
                 "unknown interpolation scheme");
        RR_CHECK(o.bottom_extrapolation >= 0 && o.bottom_extrapolation <= RRTMGP_BOTTOM_HYDROSTATIC,
                 "unknown bottom extrapolation scheme");
        a.bottom_mode = o.bottom_extrapolation == RRTMGP_BOTTOM_SAME_AS_INTERPOLATION ? a.interpolation
                                                                                        : 100 + o.bottom_extrapolation;
        const bool needs_z = a.interpolation == RRTMGP_INTERP_BEST_FIT || a.bottom_mode == 100 + RRTMGP_BOTTOM_HYDROSTATIC;
        RR_CHECK(!needs_z || (v.center_z && v.face_z), "BestFit / HydrostaticBottom need center_z and face_z");
        RR_CHECK(v.nlay - (a.iso ? 1 : 0) >= 2, "level interpolation needs at least two layers");
    }
    RR_CHECK(!a.iso || v.nlay >= 2, "the isothermal boundary layer needs at least two layers");
    a.p_min = (FT)o.p_min; a.t_min = (FT)o.t_min; a.t_max = (FT)o.t_max;
    a.clamp_t = !gray && o.t_min <= o.t_max;
    a.R = (FT)ps.gas_constant / (FT)ps.molmass_dryair;
    a.cp = a.R / (FT)ps.kappa_d;
    a.g = (FT)ps.grav;
    a.mol_m_dry = (FT)ps.molmass_dryair; a.mol_m_h2o = (FT)ps.molmass_water; a.avogadro = (FT)ps.avogad;
    if (v.ncol == 0) return RRTMGP_OK;
    hipLaunchKernelGGL((prepare_kernel<FT>), dim3((unsigned)v.ncol), dim3(64), 0, ws->stream, v, a);
    RR_HIP(hipGetLastError());
    return RRTMGP_OK;
}

template int launch_prepare<float>(rrtmgp_workspace *, const PrepView<float> &, const rrtmgp_params &,
                                   const rrtmgp_prepare_opts &, bool);
template int launch_prepare<double>(rrtmgp_workspace *, const PrepView<double> &, const rrtmgp_params &,
                                    const rrtmgp_prepare_opts &, bool);

}  // namespace rrtmgp
