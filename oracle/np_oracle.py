"""TEST INFRASTRUCTURE: an independent numpy restatement of the hot path, used only to
cross-check the C oracle (oracle/rrtmgp_oracle.c) — two transcriptions of the reference
that share no code.  Vectorised over g-points, explicit loops over columns and layers;
Float64; clouds with a deterministic mask (cld_frac in {0, 1}), no aerosols.

Follows (reference file:line): gas optics src/optics/gas_optics.jl:87-444 and
optics_utils.jl:34-223; LW sources compute_optical_props.jl:129-198; cloud optics
cloud_optics.jl:70-244; two-stream solvers src/rte/longwave_2stream.jl:149-334 and
shortwave_2stream.jl:189-392; no-scattering LW longwave_noscat.jl:171-301.
"""
import numpy as np

EPS = np.finfo(np.float64).eps


def _interp1d_equi(xi, x, y):
    if xi < x[0]:
        return y[0]
    if xi > x[-1]:
        return y[-1]
    dx = x[1] - x[0]
    n = len(x)
    loc = 0 if xi <= x[0] else (n - 2 if xi >= x[-1] else min(int((xi - x[0]) / dx), n - 2))
    f = (xi - x[loc]) / dx
    return y[loc] * (1 - f) + y[loc + 1] * f


def _vmr(as_, ig, k, c):
    if ig == 0:
        return 1.0
    v = as_.vmr
    if hasattr(v, "vmr_h2o"):
        return v.vmr_h2o[k, c] if ig == 1 else v.vmr_o3[k, c] if ig == 3 else v.vmr[ig - 1]
    return v.vmr[ig - 1, k, c]


def gas_optics_column(lk, as_, c):
    """tau, ssa, pfrac for every (layer, g-point) of column c -> arrays (nlay, ngpt)."""
    nlay = as_.layerdata.shape[1]
    ngpt = lk.kmajor.shape[3]
    n_eta = lk.kmajor.shape[0]
    bnd = lk.major_gpt2bnd - 1
    tau = np.zeros((nlay, ngpt)); ssa = np.zeros((nlay, ngpt)); pfrac = np.zeros((nlay, ngpt))
    for k in range(nlay):
        col_dry, p, t = as_.layerdata[0, k, c], as_.layerdata[1, k, c], as_.layerdata[2, k, c]
        tropo = 0 if p > lk.p_ref_tropo else 1
        h2o = _vmr(as_, lk.idx_h2o, k, c)
        dT = lk.t_ref[1] - lk.t_ref[0]
        nt = len(lk.t_ref)
        jT = 0 if t <= lk.t_ref[0] else (nt - 2 if t >= lk.t_ref[-1] else min(int((t - lk.t_ref[0]) / dT), nt - 2))
        fT = (t - lk.t_ref[jT]) / dT
        dlp = lk.ln_p_ref[0] - lk.ln_p_ref[1]
        lp = np.log(p)
        jp = min(max(int((lk.ln_p_ref[0] - lp) / dlp) + 1, 1), len(lk.ln_p_ref) - 1) + 1   # 1-based jpress
        fP = (lk.ln_p_ref[jp - 2] - lp) / dlp
        jpt = jp + tropo            # 1-based jpresst = jpress + tropo1 - 1, tropo1 = tropo + 1
        pA, pB = jpt - 2, jpt - 1   # 0-based planes
        for b in range(lk.key_species.shape[2]):
            gs = np.nonzero(bnd == b)[0]
            ig = lk.key_species[:, tropo, b]
            v1, v2 = _vmr(as_, ig[0], k, c), _vmr(as_, ig[1], k, c)
            je, fe, cm = [], [], []
            for it in range(2):
                eh = lk.vmr_ref[tropo, ig[0], jT + it] / lk.vmr_ref[tropo, ig[1], jT + it]
                mix = v1 + eh * v2
                eta = v1 * (1.0 / mix) if mix > 0 else 0.5
                loc = eta * (n_eta - 1)
                j = min(int(loc), n_eta - 2)
                je.append(j); fe.append(loc - j); cm.append(mix)

            def tri(tab, s1, s2):
                a = tab[:, :, :, gs]
                return (s1 * ((1 - fP) * ((1 - fT) * ((1 - fe[0]) * a[je[0], pA, jT] + fe[0] * a[je[0] + 1, pA, jT])) +
                              fP * ((1 - fT) * ((1 - fe[0]) * a[je[0], pB, jT] + fe[0] * a[je[0] + 1, pB, jT]))) +
                        s2 * ((1 - fP) * (fT * ((1 - fe[1]) * a[je[1], pA, jT + 1] + fe[1] * a[je[1] + 1, pA, jT + 1])) +
                              fP * (fT * ((1 - fe[1]) * a[je[1], pB, jT + 1] + fe[1] * a[je[1] + 1, pB, jT + 1]))))

            def bil(tab_g):   # tab_g: (n_eta, n_t, len(gs))
                return ((1 - fe[0]) * (1 - fT) * tab_g[je[0], jT] + fe[0] * (1 - fT) * tab_g[je[0] + 1, jT] +
                        (1 - fe[1]) * fT * tab_g[je[1], jT + 1] + fe[1] * fT * tab_g[je[1] + 1, jT + 1])
            tmaj = tri(lk.kmajor, cm[0], cm[1]) * col_dry
            mn = lk.minor_lower if tropo == 0 else lk.minor_upper
            st, en = mn.bnd_st[b] - 1, mn.bnd_st[b + 1] - 1
            tmin = np.zeros(len(gs))
            for i in range(en - st):
                gas, sgas, dens, comp = mn.gasdata[:, st + i]
                vm = _vmr(as_, gas, k, c)
                if vm > 0:
                    sc = vm * col_dry
                    if dens == 1:
                        sc *= 0.01 * p / t
                        if sgas > 0:
                            vs = _vmr(as_, sgas, k, c) * (1.0 / (1.0 + h2o))
                            sc *= (1 - vs) if comp == 1 else vs
                    idx = mn.gpt_st[gs] - 1 + i
                    tmin += bil(mn.kminor[:, :, idx]) * sc
            if not lk.is_sw:
                tau[k, gs] = np.maximum(tmaj + tmin, 0)
                pfrac[k, gs] = tri(lk.planck_fraction, 1.0, 1.0)
            else:
                ray = lk.rayl_lower if tropo == 0 else lk.rayl_upper
                tray = bil(ray[:, :, gs]) * (h2o + 1) * col_dry
                tt = np.maximum(tmaj + tmin + tray, 0)
                tau[k, gs] = tt
                ssa[k, gs] = np.where(tt > 0, tray / np.where(tt > 0, tt, 1), 0)
    return tau, ssa, pfrac


def cloud_increment(tau, ssa, g, lkc, as_, c, bnd, delta):
    cs = as_.cloud_state
    nlay = tau.shape[0]
    nl, ni = int(lkc.dims[2]), int(lkc.dims[3])
    lo_l, up_l, lo_i, up_i = lkc.bounds

    def props(tab, n, lo, up, re, path):
        if not path > EPS:
            return 0.0, 0.0, 0.0
        dr = (up - lo) / (n - 1)
        re = max(min(re, up), lo)
        loc = max(min(int((re - lo) / dr) + 1, n - 1), 1)
        fac = (re - lo - (loc - 1) * dr) / dr
        ext = (1 - fac) * tab[loc - 1] + fac * tab[loc]
        s = (1 - fac) * tab[n + loc - 1] + fac * tab[n + loc]
        a = (1 - fac) * tab[2 * n + loc - 1] + fac * tab[2 * n + loc]
        t = max(ext * path, 0.0)
        return t, s * t, a * s * t
    for k in range(nlay):
        if not cs.cld_frac[k, c] > 0:
            continue
        for b in range(lkc.liqdata.shape[1]):
            gs = np.nonzero(bnd == b)[0]
            tl, tls, tlsg = props(lkc.liqdata[:, b], nl, lo_l, up_l, cs.cld_r_eff_liq[k, c], cs.cld_path_liq[k, c])
            ti, tis, tisg = props(lkc.icedata[:, b, cs.ice_rgh - 1], ni, lo_i, up_i, cs.cld_r_eff_ice[k, c],
                                  cs.cld_path_ice[k, c])
            t2, s2 = tl + ti, tls + tis
            g2 = (tlsg + tisg) / max(EPS, s2)
            s2 = s2 / max(EPS, t2)
            if delta:
                w = s2 * (1 - g2) * (1 + g2)
                om = (1 - s2) + w
                t2, s2, g2 = om * t2, w / max(EPS, om), g2 / max(EPS, 1 + g2)
            t1, s1, g1 = tau[k, gs], ssa[k, gs], g[k, gs]
            tt = t1 + t2
            ss = t1 * s1 + t2 * s2
            g[k, gs] = (t1 * s1 * g1 + t2 * s2 * g2) / np.maximum(EPS, ss)
            ssa[k, gs] = ss / np.maximum(EPS, tt)
            tau[k, gs] = tt


def lw_sources(lk, as_, c, pfrac):
    nlay, ngpt = pfrac.shape
    bnd = lk.major_gpt2bnd - 1

    def B(T):
        return np.array([_interp1d_equi(T, lk.t_planck, lk.tot_planck[:, b]) for b in range(lk.tot_planck.shape[1])])[bnd]
    lev = np.zeros((nlay + 1, ngpt))
    lay = np.zeros((nlay, ngpt))
    inc_prev = None
    for k in range(nlay):
        dec = B(as_.t_lev[k, c]) * pfrac[k]
        inc = B(as_.t_lev[k + 1, c]) * pfrac[k]
        lay[k] = B(as_.layerdata[2, k, c]) * pfrac[k]
        lev[k] = dec if k == 0 else np.sqrt(inc_prev * dec)
        inc_prev = inc
    lev[nlay] = inc_prev
    sfc = B(as_.t_sfc[c]) * pfrac[0]
    return lay, lev, sfc


def lw_coeffs(tau, ssa, g, bot, top):
    D = 1.66
    g1 = D * (1 - 0.5 * ssa * (1 + g))
    g2 = D * 0.5 * ssa * (1 - g)
    k = np.sqrt(np.maximum(D * (1 - ssa) * (g1 + g2), np.sqrt(EPS)))
    e1 = np.exp(-tau * k)
    om1 = -np.expm1(-tau * k)
    om2 = om1 * (1 + e1)
    RT = 1 / (k * (1 + e1 * e1) + g1 * om2)
    R, T = RT * g2 * om2, RT * 2 * k * e1
    dB = bot - top
    gs = g1 + g2
    tsafe = np.where(tau > 0, tau, 1)
    emis = om1 * (k * om1 + D * (1 - ssa) * (1 + e1)) * RT
    dBz = dB * (om1 / tsafe) * (k * om1 + gs * (1 + e1)) * RT / np.maximum(gs, EPS)
    su = np.where(tau > 0, np.pi * (top * emis - T * dB + dBz), 0)
    sd = np.where(tau > 0, np.pi * (bot * emis + T * dB - dBz), 0)
    return R, T, su, sd


def solve_lw_2stream(lk, as_, bcs, lkc=None, per_gpoint=False):
    """`per_gpoint=True` returns (nlev, ncol, n_gpt) arrays instead of the g-point sums."""
    nlay, ncol = as_.layerdata.shape[1:]
    bnd = lk.major_gpt2bnd - 1
    shape = (nlay + 1, ncol, lk.n_gpt) if per_gpoint else (nlay + 1, ncol)
    red = (lambda x: x) if per_gpoint else (lambda x: x.sum())
    up = np.zeros(shape); dn = np.zeros(shape)
    for c in range(ncol):
        tau, ssa, pf = gas_optics_column(lk, as_, c)
        g = np.zeros_like(tau)
        if lkc is not None:
            cloud_increment(tau, ssa, g, lkc, as_, c, bnd, False)
        _, lev, sfc = lw_sources(lk, as_, c, pf)
        emis = bcs.sfc_emis[bnd, c]
        alb = [1 - emis]
        src = [np.pi * emis * sfc]
        for k in range(nlay):
            R, T, su, sd = lw_coeffs(tau[k], ssa[k], g[k], lev[k], lev[k + 1])
            den = 1 / (1 - R * alb[k])
            alb.append(R + T * T * alb[k] * den)
            src.append(su + T * den * (src[k] + alb[k] * sd))
        F = bcs.inc_flux[c] if bcs.inc_flux is not None else np.zeros(tau.shape[1])
        dn[nlay, c] = red(F); up[nlay, c] = red(F * alb[nlay] + src[nlay])
        for k in range(nlay - 1, -1, -1):
            R, T, su, sd = lw_coeffs(tau[k], ssa[k], g[k], lev[k], lev[k + 1])
            den = 1 / (1 - R * alb[k])
            F = (T * F + R * src[k] + sd) * den
            dn[k, c] = red(F); up[k, c] = red(F * alb[k] + src[k])
    return up, dn


def solve_lw_noscat(lk, as_, bcs, lkc=None, Ds=1.0 / 0.6096748751, w=1.0):
    nlay, ncol = as_.layerdata.shape[1:]
    bnd = lk.major_gpt2bnd - 1
    up = np.zeros((nlay + 1, ncol)); dn = np.zeros((nlay + 1, ncol))
    thr = np.sqrt(np.sqrt(EPS))
    for c in range(ncol):
        tau, ssa, pf = gas_optics_column(lk, as_, c)
        if lkc is not None:
            g = np.zeros_like(tau); s2 = np.zeros_like(tau); t2 = np.zeros_like(tau)
            cloud_increment(t2, s2, g, lkc, as_, c, bnd, False)   # combined cloud tau, ssa on a zero background
            tau = tau + (t2 - t2 * s2)                              # absorption only, cloud_optics.jl:45
        lay, lev, sfc = lw_sources(lk, as_, c, pf)
        emis = bcs.sfc_emis[bnd, c]

        def source(levs, lays, tl, tr):
            ts = np.where(tl > thr, tl, 1)
            fact = np.where(tl > thr, (1 - tr) / ts - tr, tl * (0.5 + tl * (-1 / 3 + tl / 8)))
            return (1 - tr) * levs + 2 * fact * (lays - levs)
        I = (bcs.inc_flux[c] / np.pi) if bcs.inc_flux is not None else np.zeros(tau.shape[1])
        dn[nlay, c] = (I * np.pi * w).sum()
        for k in range(nlay - 1, -1, -1):
            tl = tau[k] * Ds; tr = np.exp(-tl)
            I = tr * I + source(lev[k], lay[k], tl, tr)
            dn[k, c] = (I * np.pi * w).sum()
        I = I * (1 - emis) + emis * sfc
        up[0, c] = (I * np.pi * w).sum()
        for k in range(1, nlay + 1):
            tl = tau[k - 1] * Ds; tr = np.exp(-tl)
            I = tr * I + source(lev[k], lay[k - 1], tl, tr)
            up[k, c] = (I * np.pi * w).sum()
    return up, dn


def sw_coeffs(tau, ssa, g, mu0):
    g1 = (8 - ssa * (5 + 3 * g)) * 0.25
    g2 = 3 * (ssa * (1 - g)) * 0.25
    g3 = (2 - (3 * mu0) * g) * 0.25
    g4 = 1 - g3
    a1, a2 = g1 * g4 + g2 * g3, g1 * g3 + g2 * g4
    k = np.sqrt(np.maximum(2 * (1 - ssa) * (g1 + g2), np.sqrt(EPS)))
    e1 = np.exp(-tau * k); e2 = e1 * e1
    om1 = -np.expm1(-tau * k); om2 = om1 * (1 + e1)
    RT = 1 / (k * (1 + e2) + g1 * om2)
    Rdif, Tdif = RT * g2 * om2, RT * 2 * k * e1
    T0 = np.exp(-tau / max(mu0, EPS))
    kmu = k * mu0
    kmu2 = kmu * kmu
    d = 1 - kmu2
    win = np.sqrt(EPS)
    near = np.abs(d) < win
    kmu2 = np.where(near, np.where(d >= 0, 1 - win, 1 + win), kmu2)
    kmu = np.where(near, np.sqrt(kmu2), kmu)
    kg3, kg4 = k * g3, k * g4
    RT2 = ssa * RT / (1 - kmu2)
    Rdir = RT2 * ((1 - kmu) * (a2 + kg3) - (1 + kmu) * (a2 - kg3) * e2 - 2 * (kg3 - a2 * kmu) * e1 * T0)
    Tdir = -RT2 * ((1 + kmu) * (a1 + kg4) * T0 - (1 - kmu) * (a1 - kg4) * e2 * T0 - 2 * (kg4 + a1 * kmu) * e1)
    Rdir, Tdir = np.maximum(0, Rdir), np.maximum(0, Tdir)
    av = np.maximum(0, 1 - T0)
    tot = Rdir + Tdir
    sc = np.where(tot > av, av / np.maximum(EPS, tot), 1.0)
    return Rdir * sc, Tdir * sc, Rdif, Tdif


def solve_sw_2stream(lk, as_, bcs, lkc=None):
    nlay, ncol = as_.layerdata.shape[1:]
    bnd = lk.major_gpt2bnd - 1
    up = np.zeros((nlay + 1, ncol)); dn = np.zeros((nlay + 1, ncol)); dr = np.zeros((nlay + 1, ncol))
    for c in range(ncol):
        mu0 = bcs.cos_zenith[c]
        if not mu0 > 0:
            continue
        tau, ssa, _ = gas_optics_column(lk, as_, c)
        g = np.zeros_like(tau)
        if lkc is not None:
            cloud_increment(tau, ssa, g, lkc, as_, c, bnd, True)
        top = bcs.toa_flux[c] * lk.solar_src_scaled * mu0
        inv = 1 / max(mu0, EPS)
        dirs = np.zeros((nlay + 1, tau.shape[1]))
        dirs[nlay] = top
        cum = np.zeros(tau.shape[1])
        for k in range(nlay - 1, -1, -1):
            cum = cum + tau[k]
            dirs[k] = top * np.exp(-cum * inv)
        alb = [bcs.sfc_alb_diffuse[bnd, c]]
        src = [dirs[0] * bcs.sfc_alb_direct[bnd, c]]
        co = []
        for k in range(nlay):
            Rdir, Tdir, R, T = sw_coeffs(tau[k], ssa[k], g[k], mu0)
            co.append((Tdir, R, T))
            den = 1 / (1 - R * alb[k])
            alb.append(R + T * T * alb[k] * den)
            src.append(Rdir * dirs[k + 1] + T * den * (src[k] + alb[k] * (Tdir * dirs[k + 1])))
        F = np.zeros(tau.shape[1])
        up[nlay, c] = (F * alb[nlay] + src[nlay]).sum(); dn[nlay, c] = (F + top).sum(); dr[nlay, c] = top.sum()
        for k in range(nlay - 1, -1, -1):
            Tdir, R, T = co[k]
            den = 1 / (1 - R * alb[k])
            F = (T * F + R * src[k] + Tdir * dirs[k + 1]) * den
            up[k, c] = (F * alb[k] + src[k]).sum(); dn[k, c] = (F + dirs[k]).sum(); dr[k, c] = dirs[k].sum()
    return up, dn, dr
