// lookups.hip — lookup tables: re-layout (g-point innermost, 16-byte gather entries, slot pairs, bands dealt to the
// wavefronts) and one-off upload; replaces the `DA(...)` uploads of ext/lookup_constructors.jl:83,407,727,18.
#include "host.h"
#include "device.h"

namespace rrtmgp {

// ---- upload helpers -----------------------------------------------------------------
template <typename T>
static int upload(rrtmgp_lookup *lk, const std::vector<T> &h, const T **out) {
    void *d = nullptr;
    const size_t bytes = std::max<size_t>(h.size(), 1) * sizeof(T);
    RR_HIP(rr_malloc(&d, bytes));
    lk->allocs.push_back(d);
    if (!h.empty()) {
        host_range_check(nullptr, h.data(), h.size() * sizeof(T));  // no stale registration under this buffer (host_pin)
        RR_HIP(hipMemcpy(d, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice));
    }
    *out = (const T *)d;
    return RRTMGP_OK;
}

template <typename FT>
static int upload_raw(rrtmgp_lookup *lk, const void *src, size_t n, const FT **out) {
    std::vector<FT> h((const FT *)src, (const FT *)src + n);
    return upload(lk, h, out);
}

template <typename FT>
static int build_gas(rrtmgp_lookup *lk, const rrtmgp_gas_lookup_desc *d, DevGas<FT> &g) {
    const int64_t NE = d->n_eta, NP = d->n_p_ref + 1, NT = d->n_t_ref, NG = d->n_gpt, NB = d->n_bnd;
    RR_CHECK(NE >= 2 && NP >= 3 && NT >= 2 && NG >= 1 && NB >= 1, "bad gas lookup dimensions");
    RR_CHECK((double)NE * NP * NT * NG * 16.0 < 4.0e9, "gas lookup too large for 32-bit table offsets");
    RR_CHECK(d->kmajor && d->ln_p_ref && d->t_ref && d->vmr_ref && d->key_species && d->major_gpt2bnd,
             "gas lookup: missing table");
    g.is_sw = d->is_sw; g.n_gpt = (int)NG; g.n_bnd = (int)NB; g.n_eta = (int)NE; g.n_pp = (int)NP; g.n_t_ref = (int)NT;
    g.n_gases = (int)d->n_gases; g.n_t_plnk = (int)d->n_t_plnk; g.idx_h2o = (int)d->idx_h2o;
    g.p_ref_tropo = (FT)d->p_ref_tropo;
    // host image of the gather arena; pieces start on 256-byte boundaries
    std::vector<FT> arena;
    auto arena_piece = [&](size_t n) -> size_t {
        const size_t at = (arena.size() + 63) & ~size_t(63);
        arena.resize(at + n, FT(0));
        return at;
    };
    g.t_planck = nullptr; g.tot_planck = nullptr;
    {   // kmajor (and planck_fraction): one 16-byte entry per (t, p, eta, g) that carries the neighbours ONE gather should
        // bring (common.h DevGas::off_kmajor; device.h gas_issue reads them in this order)
        RR_CHECK(d->is_sw || (d->planck_fraction && d->t_planck && d->tot_planck && d->n_t_plnk >= 2), "LW lookup: missing Planck tables");
        const FT *sk = (const FT *)d->kmajor, *sp = (const FT *)d->planck_fraction;
        constexpr size_t NV = KMAJOR_ENTRY_BYTES / sizeof(FT);   // values per entry: 4 (Float32) or 2 (Float64)
        const size_t at = arena_piece(NV * NE * NP * NT * NG);
        for (int64_t gq = 0; gq < NG; gq++)
            for (int64_t t = 0; t < NT; t++)
                for (int64_t p = 0; p < NP; p++)
                    for (int64_t e = 0; e < NE; e++) {
                        const int64_t e1 = std::min(e + 1, NE - 1), p1 = std::min(p + 1, NP - 1);
                        auto src = [&](int64_t ee, int64_t pp) { return (size_t)(ee + NE * (pp + NP * (t + NT * gq))); };
                        FT *o = &arena[at + NV * (((t * NP + p) * NE + e) * NG + gq)];
                        if (d->is_sw && NV == 4) { o[0] = sk[src(e, p)]; o[1] = sk[src(e1, p)]; o[2] = sk[src(e, p1)]; o[3] = sk[src(e1, p1)]; }
                        else if (d->is_sw) { o[0] = sk[src(e, p)]; o[1] = sk[src(e1, p)]; }
                        else if (NV == 4) { o[0] = sk[src(e, p)]; o[1] = sp[src(e, p)]; o[2] = sk[src(e1, p)]; o[3] = sp[src(e1, p)]; }
                        else { o[0] = sk[src(e, p)]; o[1] = sp[src(e, p)]; }
                    }
        g.off_kmajor = (unsigned)(at * sizeof(FT));
    }
    if (!d->is_sw) {
        TRY(upload_raw<FT>(lk, d->t_planck, d->n_t_plnk, &g.t_planck));
        TRY(upload_raw<FT>(lk, d->tot_planck, d->n_t_plnk * NB, &g.tot_planck));
    }
    TRY(upload_raw<FT>(lk, d->ln_p_ref, d->n_p_ref, &g.ln_p_ref));
    TRY(upload_raw<FT>(lk, d->t_ref, NT, &g.t_ref));
    TRY(upload_raw<FT>(lk, d->vmr_ref, 2 * d->n_gases * NT, &g.vmr_ref));
    std::vector<int> ks(4 * NB), g2b(NG), lo(NB, -1), ng(NB, 0);
    for (int64_t i = 0; i < 4 * NB; i++) {
        RR_CHECK(d->key_species[i] >= 0 && d->key_species[i] < d->n_gases, "key_species out of range");
        ks[i] = (int)d->key_species[i];
    }
    for (int64_t i = 0; i < NG; i++) {
        const int64_t b = d->major_gpt2bnd[i] - 1;
        RR_CHECK(b >= 0 && b < NB, "major_gpt2bnd out of range");
        RR_CHECK(i == 0 || b >= d->major_gpt2bnd[i - 1] - 1, "g-points of a band must be contiguous");
        g2b[i] = (int)b;
        if (lo[b] < 0) lo[b] = (int)i;
        ng[b]++;
    }
    // lane layout of the per-band flux variants (common.h): band by band on 16-lane rows
    std::vector<int> row_lo(NB + 1, 0), lane_gpt(256, -1);
    for (int64_t b = 0; b < NB; b++) row_lo[b + 1] = row_lo[b] + (ng[b] + 15) / 16;
    g.band_rows = row_lo[NB] <= 16 ? row_lo[NB] : 0;
    if (g.band_rows)
        for (int64_t b = 0; b < NB; b++)
            for (int i = 0; i < ng[b]; i++) lane_gpt[row_lo[b] * 16 + i] = lo[b] + i;
    TRY(upload(lk, row_lo, &g.band_row_lo));
    TRY(upload(lk, lane_gpt, &g.band_lane_gpt));
    TRY(upload(lk, ks, &g.key_species));
    {   // the reference-ratio of the two key species of a band, vmr_ref[tropo, ig0 + 1, jT] / vmr_ref[tropo, ig1 + 1, jT]
        // (compute_interp_frac_eta, gas_optics.jl:140-143): formed once here, in FT with the IEEE division of the reference's
        // CPU path, instead of per (layer, band, T plane) on the device
        const FT *vr = (const FT *)d->vmr_ref;
        std::vector<FT> eh((size_t)2 * NB * NT);
        for (int tropo = 0; tropo < 2; tropo++)
            for (int64_t b = 0; b < NB; b++)
                for (int64_t t = 0; t < NT; t++) {
                    const int ig0 = ks[0 + 2 * (tropo + 2 * b)], ig1 = ks[1 + 2 * (tropo + 2 * b)];
                    eh[((size_t)tropo * NB + b) * NT + t] = vr[tropo + 2 * (ig0 + d->n_gases * t)] / vr[tropo + 2 * (ig1 + d->n_gases * t)];
                }
        TRY(upload(lk, eh, &g.eta_half));
    }
    TRY(upload(lk, g2b, &g.gpt2bnd));
    TRY(upload(lk, lo, &g.bnd_lo));
    TRY(upload(lk, ng, &g.bnd_ng));
    const rrtmgp_minor_desc *md[2] = {&d->minor_lower, &d->minor_upper};
    std::vector<int> slots[2] = {std::vector<int>(NB, 0), std::vector<int>(NB, 0)};  // slots per band and region (Rayleigh included)
    for (int r = 0; r < 2; r++) {
        const rrtmgp_minor_desc *m = md[r];
        RR_CHECK(m->bnd_st && m->gpt_st && (m->n_min_absrb == 0 || m->gasdata), "minor lookup: missing table");
        std::vector<int> bst(NB + 1), gd(4 * std::max<int64_t>(m->n_min_absrb, 1), 0), koff(NB, 0);
        for (int64_t b = 0; b <= NB; b++) bst[b] = (int)(m->bnd_st[b] - 1);
        for (int64_t i = 0; i < 4 * m->n_min_absrb; i++) gd[i] = (int)m->gasdata[i];
        for (int64_t i = 0; i < m->n_min_absrb; i++)
            RR_CHECK(gd[4 * i] >= 0 && gd[4 * i] < d->n_gases && gd[4 * i + 1] >= 0 && gd[4 * i + 1] < d->n_gases,
                     "minor gas index out of range");
        // reference order: contributor (gpt_st[g] - 1) + i.  Device order: the slots of a g-point (SW: slot 0 = the Rayleigh
        // coefficient, then the contributors) come in PAIRS, and the 16-byte (Float32) entry of a pair at (t, eta) holds both
        // slots at eta AND at eta + 1:   {c_2p(e), c_2p(e+1), c_2p+1(e), c_2p+1(e+1)}   at
        //     koff[b] + (p * ng_b + (g - lo_b)) * 4        along the row of (t, eta)
        // so that ONE gather per T plane serves two contributors (interp2d, optics_utils.jl:85-98, reads eta and eta + 1 of
        // jT at jeta[1] and of jT + 1 at jeta[2]: two gathers per pair).  The vector memory pipeline prices a gather by the
        // instruction (gas_issue, device.h); with 4 contributors x 1 corner per gather (rounds 2-4) a band with 1-2 slots paid
        // 4 gathers and one with 5-6 paid 8, now 2 and 6.  A band with n_b slots owns max(1, ceil(n_b / 2)) pairs; padding
        // entries are 0 and carry a zero scaling.  The scalings of a layer are laid out the same way (slot = 2 * pair + j % 2).
        // SW: compute_tau_rayleigh rides in slot 0 (krayl[:, :, g] has the same (t, eta) rows and interp2d weights; its
        // "scaling" is the layer's (h2o + 1) col_dry) instead of costing gathers of its own (RAYLEIGH_SLOT).
        const int64_t lead = d->is_sw ? 1 : 0;
        std::vector<int64_t> dst(std::max<int64_t>(m->n_contrib, 1), 0), rayl_dst(d->is_sw ? NG : 0, 0);
        std::vector<int> st2(NB, 0), slot_int;
        int64_t off = 0;
        for (int64_t b = 0; b < NB; b++) {
            const int64_t nb = bst[b + 1] - bst[b];
            RR_CHECK(nb >= 0, "minor bnd_st must be non-decreasing");
            // at least one pair per band: a band without contributors reads its own all-zero pair with zero scalings
            const int64_t npair = std::max<int64_t>(1, (nb + lead + MINOR_PAIR - 1) / MINOR_PAIR);
            koff[b] = (int)off;
            st2[b] = (int)slot_int.size();
            for (int64_t i = 0; i < npair * MINOR_PAIR; i++)
                slot_int.push_back(i < lead ? RAYLEIGH_SLOT : i < nb + lead ? (int)(bst[b] + i - lead) : -1);
            lk->max_minor = std::max<int>(lk->max_minor, (int)nb);
            slots[r][b] = (int)(nb + lead);
            for (int64_t gi = 0; gi < ng[b]; gi++) {
                const int64_t gq = lo[b] + gi;
                RR_CHECK(m->gpt_st[gq + 1] - m->gpt_st[gq] == nb, "minor gpt_st inconsistent with bnd_st");
                // position of the slot's value AT ITS OWN eta inside the entry; the eta + 1 copy sits one element further
                if (lead) rayl_dst[gq] = off + gi * MINOR_ENTRY;
                for (int64_t i = 0; i < nb; i++) {
                    const int64_t src = m->gpt_st[gq] - 1 + i, j = i + lead;
                    RR_CHECK(src >= 0 && src < m->n_contrib, "minor contributor index out of range");
                    dst[src] = off + ((j / MINOR_PAIR) * ng[b] + gi) * MINOR_ENTRY + (j % MINOR_PAIR) * 2;
                }
            }
            off += npair * ng[b] * MINOR_ENTRY;
        }
        const int64_t row = off;
        g.m_ncontrib[r] = (int)row;
        RR_CHECK(m->n_min_absrb <= 255 && slot_int.size() <= 510, "more than 255 minor-gas intervals per region are not supported");
        g.m_nint[r] = (int)m->n_min_absrb;
        g.m_nslot[r] = (int)slot_int.size();
        lk->max_int = std::max<int>(lk->max_int, (int)slot_int.size());
        TRY(upload(lk, bst, &g.m_bnd_st[r]));
        TRY(upload(lk, gd, &g.m_gasdata[r]));
        TRY(upload(lk, koff, &g.m_koff[r]));
        TRY(upload(lk, st2, &g.m_st2[r]));
        TRY(upload(lk, slot_int, &g.m_slot_int[r]));
        RR_CHECK(m->n_contrib == 0 || m->kminor, "minor lookup: missing kminor");
        {   // (n_eta, n_t, n) -> [t][eta][row]: source element c of (e, t) goes to its slot of row (t, e) and, as the eta + 1
            // neighbour, to the element behind it in row (t, e - 1); the last eta row repeats itself (never a base row)
            const size_t at = arena_piece((size_t)NE * NT * row);
            auto put = [&](const FT *s, int64_t nsrc, const std::vector<int64_t> &where) {
                for (int64_t c = 0; c < nsrc; c++)
                    for (int64_t t = 0; t < NT; t++)
                        for (int64_t e = 0; e < NE; e++) {
                            const FT v = s[e + NE * (t + NT * c)];
                            arena[at + (t * NE + e) * row + where[c]] = v;
                            if (e > 0) arena[at + (t * NE + e - 1) * row + where[c] + 1] = v;
                            if (e == NE - 1) arena[at + (t * NE + e) * row + where[c] + 1] = v;
                        }
            };
            put((const FT *)m->kminor, m->n_contrib, dst);
            if (d->is_sw) {   // krayl (n_eta, n_t, n_gpt) of this region into the leading slots of the same rows
                const FT *ry = (const FT *)(r == 0 ? d->rayl_lower : d->rayl_upper);
                RR_CHECK(ry, "SW lookup: missing Rayleigh tables");
                put(ry, NG, rayl_dst);
            }
            g.off_kminor[r] = (unsigned)(at * sizeof(FT));
        }
    }
    {   // lane -> g-point of the broadband (not per-band) instances.  A wavefront issues the minor-gas gathers of its
        // LARGEST band (gas_issue): with whole 16-g-point bands the bands are dealt to the wavefronts so that bands with
        // many slots share wavefronts, minimising  sum over wavefronts of (max pairs, lower) + (max pairs, upper).  Which
        // lane solves a g-point enters nothing but the (fixed) order of the g-point sums.  RRTMGP_HIP_BAND_ORDER=identity
        // keeps the bands where the lookup has them.
        std::vector<int> lane_g(256, -1);
        for (int64_t i = 0; i < std::min<int64_t>(NG, 256); i++) lane_g[i] = (int)i;
        bool whole = NG <= 256 && NB * 16 == NG;
        for (int64_t b = 0; b < NB && whole; b++) whole = ng[b] == 16;
        const char *ord = getenv("RRTMGP_HIP_BAND_ORDER");
        if (whole && NB > 4 && !(ord && !strcmp(ord, "identity"))) {
            auto pairs = [&](int r, int b) { return std::max(1, (slots[r][b] + MINOR_PAIR - 1) / MINOR_PAIR); };
            auto cost = [&](const std::vector<int> &perm) {
                int c = 0;
                for (int64_t w = 0; w * 4 < NB; w++) {
                    int m0 = 0, m1 = 0;
                    for (int64_t i = w * 4; i < std::min<int64_t>(NB, w * 4 + 4); i++) { m0 = std::max(m0, pairs(0, perm[i])); m1 = std::max(m1, pairs(1, perm[i])); }
                    c += m0 + m1;
                }
                return c;
            };
            std::vector<int> best(NB);
            for (int64_t b = 0; b < NB; b++) best[b] = (int)b;
            int cbest = cost(best);
            // starts: the bands sorted by (lower, upper), by (upper, lower) and by their sum; then pairwise exchanges
            for (int key = 0; key < 3; key++) {
                std::vector<int> p(NB);
                for (int64_t b = 0; b < NB; b++) p[b] = (int)b;
                std::stable_sort(p.begin(), p.end(), [&](int x, int y) {
                    const int x0 = pairs(0, x), x1 = pairs(1, x), y0 = pairs(0, y), y1 = pairs(1, y);
                    if (key == 0) return x0 != y0 ? x0 > y0 : x1 > y1;
                    if (key == 1) return x1 != y1 ? x1 > y1 : x0 > y0;
                    return x0 + x1 > y0 + y1;
                });
                int c = cost(p);
                for (bool moved = true; moved;) {
                    moved = false;
                    for (int64_t i = 0; i < NB; i++)
                        for (int64_t j = i + 1; j < NB; j++) {
                            if (i / 4 == j / 4) continue;
                            std::swap(p[i], p[j]);
                            const int c2 = cost(p);
                            if (c2 < c) { c = c2; moved = true; } else std::swap(p[i], p[j]);
                        }
                }
                if (c < cbest) { cbest = c; best = p; }
            }
            for (int64_t i = 0; i < NB; i++)
                for (int q = 0; q < 16; q++) lane_g[i * 16 + q] = lo[best[i]] + q;
        }
        TRY(upload(lk, lane_g, &g.lane_gpt));
    }
    g.solar_src_scaled = nullptr;
    if (d->is_sw) {
        RR_CHECK(d->solar_src_scaled, "SW lookup: missing solar source table");
        TRY(upload_raw<FT>(lk, d->solar_src_scaled, NG, &g.solar_src_scaled));
    }
    arena_piece(64);  // keeps the last table off the end of the allocation
    RR_CHECK((double)arena.size() * sizeof(FT) < 4.0e9, "gas lookup too large for 32-bit table offsets");
    {
        const FT *dev = nullptr;
        TRY(upload(lk, arena, &dev));
        g.arena = (const char *)dev;
    }
    return RRTMGP_OK;
}

template <typename FT>
static int build_cld(rrtmgp_lookup *lk, const rrtmgp_cloud_lookup_desc *d, DevCld<FT> &c) {
    RR_CHECK(d->bounds && d->liqdata && d->icedata, "cloud lookup: missing table");
    RR_CHECK(d->nsize_liq >= 2 && d->nsize_ice >= 2 && d->nband >= 1 && d->nrghice >= 1, "bad cloud lookup dimensions");
    c.nband = (int)d->nband; c.nrghice = (int)d->nrghice; c.nsize_liq = (int)d->nsize_liq; c.nsize_ice = (int)d->nsize_ice;
    const FT *b = (const FT *)d->bounds;
    c.radliq_lwr = b[0]; c.radliq_upr = b[1]; c.radice_lwr = b[2]; c.radice_upr = b[3];
    TRY(upload_raw<FT>(lk, d->liqdata, 3 * d->nsize_liq * d->nband, &c.liqdata));
    TRY(upload_raw<FT>(lk, d->icedata, 3 * d->nsize_ice * d->nband * d->nrghice, &c.icedata));
    return RRTMGP_OK;
}

template <typename FT>
static int build_aero(rrtmgp_lookup *lk, const rrtmgp_aerosol_lookup_desc *d, DevAero<FT> &a) {
    RR_CHECK(d->size_bin_limits && d->rh_levels && d->dust && d->sea_salt && d->sulfate && d->black_carbon_rh &&
                 d->black_carbon && d->organic_carbon_rh && d->organic_carbon,
             "aerosol lookup: missing table");
    RR_CHECK(d->nbin >= 1 && d->nbin <= 255 && d->nrh >= 2 && d->nband >= 1, "bad aerosol lookup dimensions");
    RR_CHECK(d->iband_550nm >= 0 && d->iband_550nm <= d->nband, "iband_550nm must be 0 (none) or a band index");
    a.nband = (int)d->nband; a.nbin = (int)d->nbin; a.nrh = (int)d->nrh; a.iband_550nm = (int)d->iband_550nm;
    TRY(upload_raw<FT>(lk, d->size_bin_limits, 2 * d->nbin, &a.size_bin_limits));
    TRY(upload_raw<FT>(lk, d->rh_levels, d->nrh, &a.rh_levels));
    TRY(upload_raw<FT>(lk, d->dust, 3 * d->nbin * d->nband, &a.dust));
    TRY(upload_raw<FT>(lk, d->sea_salt, 3 * d->nrh * d->nbin * d->nband, &a.sea_salt));
    TRY(upload_raw<FT>(lk, d->sulfate, 3 * d->nrh * d->nband, &a.sulfate));
    TRY(upload_raw<FT>(lk, d->black_carbon_rh, 3 * d->nrh * d->nband, &a.black_carbon_rh));
    TRY(upload_raw<FT>(lk, d->black_carbon, 3 * d->nband, &a.black_carbon));
    TRY(upload_raw<FT>(lk, d->organic_carbon_rh, 3 * d->nrh * d->nband, &a.organic_carbon_rh));
    TRY(upload_raw<FT>(lk, d->organic_carbon, 3 * d->nband, &a.organic_carbon));
    return RRTMGP_OK;
}

}  // namespace rrtmgp

using namespace rrtmgp;

extern "C" {

int rrtmgp_hip_gas_lookup_create(const rrtmgp_gas_lookup_desc *desc, int device, rrtmgp_lookup **out) {
    RR_CHECK(desc && out, "null argument");
    RR_CHECK(desc->ftype == RRTMGP_F32 || desc->ftype == RRTMGP_F64, "ftype must be 4 or 8");
    TRY(select_device(device));
    auto *lk = new rrtmgp_lookup();
    lk->kind = LK_GAS; lk->ftype = desc->ftype; lk->device = device; lk->max_minor = 0; lk->max_int = 0;
    int rc = desc->ftype == RRTMGP_F32 ? build_gas<float>(lk, desc, lk->gas32) : build_gas<double>(lk, desc, lk->gas64);
    if (rc) { rrtmgp_hip_lookup_destroy(lk); return rc; }
    *out = lk;
    return RRTMGP_OK;
}

int rrtmgp_hip_cloud_lookup_create(const rrtmgp_cloud_lookup_desc *desc, int device, rrtmgp_lookup **out) {
    RR_CHECK(desc && out, "null argument");
    RR_CHECK(desc->ftype == RRTMGP_F32 || desc->ftype == RRTMGP_F64, "ftype must be 4 or 8");
    TRY(select_device(device));
    auto *lk = new rrtmgp_lookup();
    lk->kind = LK_CLOUD; lk->ftype = desc->ftype; lk->device = device; lk->max_minor = 0; lk->max_int = 0;
    int rc = desc->ftype == RRTMGP_F32 ? build_cld<float>(lk, desc, lk->cld32) : build_cld<double>(lk, desc, lk->cld64);
    if (rc) { rrtmgp_hip_lookup_destroy(lk); return rc; }
    *out = lk;
    return RRTMGP_OK;
}

int rrtmgp_hip_aerosol_lookup_create(const rrtmgp_aerosol_lookup_desc *desc, int device, rrtmgp_lookup **out) {
    RR_CHECK(desc && out, "null argument");
    RR_CHECK(desc->ftype == RRTMGP_F32 || desc->ftype == RRTMGP_F64, "ftype must be 4 or 8");
    TRY(select_device(device));
    auto *lk = new rrtmgp_lookup();
    lk->kind = LK_AEROSOL; lk->ftype = desc->ftype; lk->device = device; lk->max_minor = 0; lk->max_int = 0;
    int rc = desc->ftype == RRTMGP_F32 ? build_aero<float>(lk, desc, lk->aero32) : build_aero<double>(lk, desc, lk->aero64);
    if (rc) { rrtmgp_hip_lookup_destroy(lk); return rc; }
    *out = lk;
    return RRTMGP_OK;
}

int rrtmgp_hip_lookup_destroy(rrtmgp_lookup *lk) {
    if (!lk) return RRTMGP_OK;
    (void)hipSetDevice(lk->device);
    for (void *p : lk->allocs) (void)rr_free(p);
    for (rrtmgp_lookup *r : lk->replicas) rrtmgp_hip_lookup_destroy(r);
    delete lk;
    return RRTMGP_OK;
}

}  // extern "C"
