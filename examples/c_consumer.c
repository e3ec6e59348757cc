/* A plain-C consumer of include/rrtmgp_hip.h: what a `ccall` / cgo / FFI binding sees.  C99, the header and libc only -
 * no Python mirror of the structs, no C++.
 *
 *   gcc -std=c99 -Iinclude examples/c_consumer.c -Lrrtmgp.jl_amd -lhip_rrtmgp -lm -Wl,-rpath,$PWD/rrtmgp.jl_amd -o c_consumer
 *   ./c_consumer [case.bin]
 *
 * Without a GPU (tests/test_abi.py): the struct sizes a binding would mirror, the host-callable McICA stream, and that a
 * solve without a device fails with a status and a message instead of crashing.
 * With a GPU (tests/test_c_consumer.py, `-m gpu`):
 *   1. gray longwave, no scattering, on an ISOTHERMAL column over a black surface: the one-angle transport is exact there,
 *      flux_dn(surface) = sigma T^4 (1 - exp(-D tau)), flux_up = sigma T^4 at every level - the property
 *      test/angular_discretization.jl:102-153 of the reference pins, with tau from the published optical-thickness profile
 *      (Schneider 2004; src/optics/gray_optics_kernels.jl:171-190);
 *   2. with a case file (examples/make_c_consumer_case.py): a spectral all-sky solve - lookups created from the raw tables,
 *      two-stream LW + SW with McICA clouds - compared with the fluxes the CPU oracle wrote next to the inputs. */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "rrtmgp_hip.h"

static int fail(const char *what) {
    char msg[512];
    rrtmgp_hip_last_error(msg, sizeof msg);
    printf("FAILED: %s: %s\n", what, msg);
    return 1;
}
#define TRY(call) do { if ((call) != RRTMGP_OK) return fail(#call); } while (0)

/* ---- 1. gray longwave on an isothermal column ------------------------------------------------------------------- */
static int gray_isothermal(void) {
    enum { NLAY = 30, NLEV = NLAY + 1, NCOL = 3 };
    const double T = 255.0, p0 = 100000.0, pe = 900.0, sigma = 5.670374419e-8, D = 1.0 / 0.6096748751;
    const double alpha = 3.5, te = 300.0, tt = 200.0, dt = 60.0; /* GrayOpticalThicknessSchneider2004 defaults */
    const double lat[NCOL] = {0.0, 35.0, -70.0};
    static double p_lay[NLAY * NCOL], p_lev[NLEV * NCOL], t_lay[NLAY * NCOL], t_lev[NLEV * NCOL], t_sfc[NCOL], emis[NCOL];
    static double up[NLEV * NCOL], dn[NLEV * NCOL], net[NLEV * NCOL];
    double tau[NCOL];
    int bad = 0;
    for (int c = 0; c < NCOL; c++) {
        const double s = sin(lat[c] / 180.0 * 3.14159265358979323846);
        const double r = (te + dt * (1.0 / 3.0 - s * s)) / tt, d0 = r * r * r * r - 1.0;
        tau[c] = 0.0;
        for (int k = 0; k < NLEV; k++) { p_lev[k + NLEV * c] = p0 - (p0 - pe) * k / NLAY; t_lev[k + NLEV * c] = T; }
        for (int k = 0; k < NLAY; k++) {
            const double p = 0.5 * (p_lev[k + NLEV * c] + p_lev[k + 1 + NLEV * c]);
            p_lay[k + NLAY * c] = p; t_lay[k + NLAY * c] = T;
            tau[c] += fabs(alpha * d0 * pow(p / p0, alpha) / p * (p_lev[k + 1 + NLEV * c] - p_lev[k + NLEV * c]));
        }
        t_sfc[c] = T; emis[c] = 1.0;
    }
    rrtmgp_workspace *ws = NULL;
    TRY(rrtmgp_hip_workspace_create(0, NCOL, NLAY, RRTMGP_F64, &ws));
    rrtmgp_gray_state gs;
    memset(&gs, 0, sizeof gs);
    gs.mem = RRTMGP_MEM_HOST; gs.otp_kind = 0; gs.ncol = NCOL; gs.nlay = NLAY;
    gs.lat = lat; gs.p_lay = p_lay; gs.p_lev = p_lev; gs.t_lay = t_lay; gs.t_lev = t_lev; gs.t_sfc = t_sfc;
    gs.otp[0] = alpha; gs.otp[1] = te; gs.otp[2] = tt; gs.otp[3] = dt; gs.stefan = sigma;
    rrtmgp_lw_bcs bcs;
    memset(&bcs, 0, sizeof bcs);
    bcs.mem = RRTMGP_MEM_HOST; bcs.sfc_emis = emis; /* (nbnd = 1, ncol) */
    rrtmgp_flux_out fl;
    memset(&fl, 0, sizeof fl);
    fl.mem = RRTMGP_MEM_HOST; fl.layout = RRTMGP_LAYOUT_NLEV_NCOL; fl.flux_up = up; fl.flux_dn = dn; fl.flux_net = net;
    rrtmgp_solve_opts o;
    memset(&o, 0, sizeof o);
    o.n_gauss_angles = 1;
    TRY(rrtmgp_hip_rte_lw_noscat_solve_gray(ws, &gs, &bcs, &fl, &o));
    for (int c = 0; c < NCOL; c++) {
        const double B = sigma * T * T * T * T, want_dn = B * (1.0 - exp(-D * tau[c]));
        const double e_dn = fabs(dn[NLEV * c] - want_dn) / want_dn;
        double e_up = 0.0;
        for (int k = 0; k < NLEV; k++) e_up = fmax(e_up, fabs(up[k + NLEV * c] - B) / B);
        printf("gray isothermal, lat %6.1f: tau %.4f  flux_dn(sfc) %.9f (exact %.9f, rel %.1e)  max rel |flux_up - sigma T^4| %.1e\n",
               lat[c], tau[c], dn[NLEV * c], want_dn, e_dn, e_up);
        if (!(e_dn < 1e-12) || !(e_up < 1e-12) || dn[NLAY + NLEV * c] != 0.0) bad++;
    }
    TRY(rrtmgp_hip_workspace_destroy(ws));
    return bad;
}

/* ---- 2. a spectral all-sky solve from a raw case file --------------------------------------------------------------- */
typedef struct { char name[32]; int32_t dtype, ndim; int64_t dims[4], nbytes; } rec_head;
static char *g_file;
static size_t g_size;

static const void *find(const char *name, int64_t *dims /* [4] or NULL */) {
    size_t off = 8;
    while (off + sizeof(rec_head) <= g_size) {
        const rec_head *h = (const rec_head *)(g_file + off);
        off += sizeof(rec_head);
        if (strncmp(h->name, name, 32) == 0) {
            if (dims) memcpy(dims, h->dims, sizeof h->dims);
            return g_file + off;
        }
        off += (size_t)((h->nbytes + 7) / 8 * 8);
    }
    printf("case file: no array named %s\n", name);
    exit(2);
}
static const void *findf(const char *prefix, const char *name, int64_t *dims) {
    char full[64];
    snprintf(full, sizeof full, "%s%s", prefix, name);
    return find(full, dims);
}

static rrtmgp_minor_desc minor_desc(const char *prefix) {
    rrtmgp_minor_desc m;
    int64_t d[4];
    memset(&m, 0, sizeof m);
    m.bnd_st = (const int64_t *)findf(prefix, ".bnd_st", NULL);
    m.gpt_st = (const int64_t *)findf(prefix, ".gpt_st", NULL);
    m.gasdata = (const int64_t *)findf(prefix, ".gasdata", d);
    m.n_min_absrb = d[1];
    m.kminor = findf(prefix, ".kminor", d);
    m.n_contrib = d[2];
    return m;
}

static int gas_lookup(const char *tag, int is_sw, rrtmgp_lookup **out) {
    rrtmgp_gas_lookup_desc g;
    int64_t d[4];
    char pre[16], sub[40];
    memset(&g, 0, sizeof g);
    snprintf(pre, sizeof pre, "%s", tag);
    const double *sc = (const double *)findf(pre, ".scalars", NULL);
    g.ftype = RRTMGP_F64; g.is_sw = is_sw;
    g.idx_h2o = (int64_t)sc[0]; g.p_ref_tropo = sc[1]; g.p_ref_min = sc[2]; g.t_ref_min = sc[3]; g.t_ref_max = sc[4]; g.solar_src_tot = sc[5];
    g.kmajor = findf(pre, ".kmajor", d);
    g.n_eta = d[0]; g.n_p_ref = d[1] - 1; g.n_t_ref = d[2]; g.n_gpt = d[3];
    g.key_species = (const int64_t *)findf(pre, ".key_species", d);
    g.n_bnd = d[2];
    g.major_gpt2bnd = (const int64_t *)findf(pre, ".major_gpt2bnd", NULL);
    g.ln_p_ref = findf(pre, ".ln_p_ref", NULL);
    g.t_ref = findf(pre, ".t_ref", NULL);
    g.vmr_ref = findf(pre, ".vmr_ref", d);
    g.n_gases = d[1];
    snprintf(sub, sizeof sub, "%s.minor_lower", tag); g.minor_lower = minor_desc(sub);
    snprintf(sub, sizeof sub, "%s.minor_upper", tag); g.minor_upper = minor_desc(sub);
    if (is_sw) {
        g.rayl_lower = findf(pre, ".rayl_lower", NULL); g.rayl_upper = findf(pre, ".rayl_upper", NULL);
        g.solar_src_scaled = findf(pre, ".solar_src_scaled", NULL);
    } else {
        g.planck_fraction = findf(pre, ".planck_fraction", NULL);
        g.t_planck = findf(pre, ".t_planck", d); g.n_t_plnk = d[0];
        g.tot_planck = findf(pre, ".tot_planck", NULL);
    }
    return rrtmgp_hip_gas_lookup_create(&g, 0, out);
}

static int cloud_lookup(const char *tag, rrtmgp_lookup **out) {
    rrtmgp_cloud_lookup_desc c;
    memset(&c, 0, sizeof c);
    const int64_t *dims = (const int64_t *)findf(tag, ".dims", NULL);
    c.ftype = RRTMGP_F64; c.nband = dims[0]; c.nrghice = dims[1]; c.nsize_liq = dims[2]; c.nsize_ice = dims[3];
    c.bounds = findf(tag, ".bounds", NULL); c.liqdata = findf(tag, ".liqdata", NULL); c.icedata = findf(tag, ".icedata", NULL);
    return rrtmgp_hip_cloud_lookup_create(&c, 0, out);
}

static double maxdiff(const double *a, const double *b, size_t n) {
    double m = 0.0;
    for (size_t i = 0; i < n; i++) {
        const double d = fabs(a[i] - b[i]);
        if (!(d <= m)) m = d; /* a NaN sticks */
    }
    return m;
}

static int spectral_case(const char *path) {
    FILE *fh = fopen(path, "rb");
    if (!fh) { printf("cannot open %s\n", path); return 1; }
    fseek(fh, 0, SEEK_END);
    g_size = (size_t)ftell(fh);
    fseek(fh, 0, SEEK_SET);
    g_file = (char *)malloc(g_size);
    if (!g_file || fread(g_file, 1, g_size, fh) != g_size || memcmp(g_file, "RRCASE1", 8) != 0) { printf("bad case file\n"); return 1; }
    fclose(fh);

    rrtmgp_lookup *lw = NULL, *sw = NULL, *cld_lw = NULL, *cld_sw = NULL;
    TRY(gas_lookup("lw", 0, &lw));
    TRY(gas_lookup("sw", 1, &sw));
    TRY(cloud_lookup("cld_lw", &cld_lw));
    TRY(cloud_lookup("cld_sw", &cld_sw));

    int64_t d[4];
    rrtmgp_atmos_state as;
    memset(&as, 0, sizeof as);
    as.mem = RRTMGP_MEM_HOST; as.vmr_kind = RRTMGP_VMR_GM;
    as.layerdata = find("as.layerdata", d);
    const int64_t nlay = d[1], ncol = d[2], nlev = nlay + 1;
    as.ncol = ncol; as.nlay = nlay;
    as.p_lev = find("as.p_lev", NULL); as.t_lev = find("as.t_lev", NULL); as.t_sfc = find("as.t_sfc", NULL);
    as.vmr_h2o = find("as.vmr_h2o", NULL); as.vmr_o3 = find("as.vmr_o3", NULL); as.vmr = find("as.vmr", d);
    as.ngas = d[0];
    as.cld_r_eff_liq = find("as.cld_r_eff_liq", NULL); as.cld_r_eff_ice = find("as.cld_r_eff_ice", NULL);
    as.cld_path_liq = find("as.cld_path_liq", NULL); as.cld_path_ice = find("as.cld_path_ice", NULL);
    as.cld_frac = find("as.cld_frac", NULL);
    as.ice_rgh = *(const int64_t *)find("as.ice_rgh", NULL);
    double *cover_lw = (double *)malloc(sizeof(double) * (size_t)ncol), *cover_sw = (double *)malloc(sizeof(double) * (size_t)ncol);
    as.cld_cover_lw = cover_lw; as.cld_cover_sw = cover_sw;

    rrtmgp_lw_bcs lb;
    memset(&lb, 0, sizeof lb);
    lb.mem = RRTMGP_MEM_HOST; lb.sfc_emis = find("lw_bcs.sfc_emis", NULL);
    rrtmgp_sw_bcs sb;
    memset(&sb, 0, sizeof sb);
    sb.mem = RRTMGP_MEM_HOST; sb.cos_zenith = find("sw_bcs.cos_zenith", NULL); sb.toa_flux = find("sw_bcs.toa_flux", NULL);
    sb.sfc_alb_direct = find("sw_bcs.sfc_alb_direct", NULL); sb.sfc_alb_diffuse = find("sw_bcs.sfc_alb_diffuse", NULL);

    const size_t n = (size_t)(nlev * ncol);
    double *buf = (double *)calloc(7 * n, sizeof(double));
    rrtmgp_flux_out fl, fs;
    memset(&fl, 0, sizeof fl); memset(&fs, 0, sizeof fs);
    fl.mem = fs.mem = RRTMGP_MEM_HOST; fl.layout = fs.layout = RRTMGP_LAYOUT_NLEV_NCOL;
    fl.flux_up = buf; fl.flux_dn = buf + n; fl.flux_net = buf + 2 * n;
    fs.flux_up = buf + 3 * n; fs.flux_dn = buf + 4 * n; fs.flux_net = buf + 5 * n; fs.flux_dn_dir = buf + 6 * n;
    rrtmgp_solve_opts o;
    memset(&o, 0, sizeof o);
    o.n_gauss_angles = 1; o.seed = (uint64_t) * (const int64_t *)find("seed", NULL);

    rrtmgp_workspace *ws = NULL;
    TRY(rrtmgp_hip_workspace_create(0, ncol, nlay, RRTMGP_F64, &ws));
    TRY(rrtmgp_hip_rte_lw_2stream_solve(ws, lw, cld_lw, NULL, &as, &lb, &fl, &o));
    TRY(rrtmgp_hip_rte_sw_2stream_solve(ws, sw, cld_sw, NULL, &as, &sb, &fs, &o));

    const char *lw_names[3] = {"expect.lw.flux_up", "expect.lw.flux_dn", "expect.lw.flux_net"};
    const char *sw_names[4] = {"expect.sw.flux_up", "expect.sw.flux_dn", "expect.sw.flux_net", "expect.sw.flux_dn_dir"};
    double e_lw = 0.0, e_sw = 0.0;
    for (int i = 0; i < 3; i++) e_lw = fmax(e_lw, maxdiff(buf + (size_t)i * n, (const double *)find(lw_names[i], NULL), n));
    for (int i = 0; i < 4; i++) e_sw = fmax(e_sw, maxdiff(buf + (size_t)(3 + i) * n, (const double *)find(sw_names[i], NULL), n));
    const double e_cov = fmax(maxdiff(cover_lw, (const double *)find("expect.cld_cover_lw", NULL), (size_t)ncol),
                              maxdiff(cover_sw, (const double *)find("expect.cld_cover_sw", NULL), (size_t)ncol));
    printf("spectral all-sky, %lld columns x %lld layers, Float64: max |HIP - oracle| LW %.3e  SW %.3e W/m2, cloud cover %.1e "
           "(surface LW up %.3f, TOA SW dn %.3f)\n", (long long)ncol, (long long)nlay, e_lw, e_sw, e_cov, buf[0], buf[4 * n + (size_t)nlay]);
    const int bad = !(e_lw < 1e-8) + !(e_sw < 1e-8) + !(e_cov == 0.0) + !(buf[0] > 100.0);
    TRY(rrtmgp_hip_workspace_destroy(ws));
    TRY(rrtmgp_hip_lookup_destroy(lw)); TRY(rrtmgp_hip_lookup_destroy(sw));
    TRY(rrtmgp_hip_lookup_destroy(cld_lw)); TRY(rrtmgp_hip_lookup_destroy(cld_sw));
    free(buf); free(cover_lw); free(cover_sw); free(g_file);
    return bad;
}

int main(int argc, char **argv) {
    char msg[256];
    int bad = 0;
    /* struct mirrors, in the library's numbering (rrtmgp_hip_abi_sizeof) */
    const size_t sizes[] = {sizeof(rrtmgp_minor_desc),        sizeof(rrtmgp_gas_lookup_desc), sizeof(rrtmgp_cloud_lookup_desc),
                            sizeof(rrtmgp_aerosol_lookup_desc), sizeof(rrtmgp_atmos_state),    sizeof(rrtmgp_lw_bcs),
                            sizeof(rrtmgp_sw_bcs),            sizeof(rrtmgp_flux_out),        sizeof(rrtmgp_solve_opts),
                            sizeof(rrtmgp_gray_state),        sizeof(rrtmgp_params),          sizeof(rrtmgp_prepare_opts)};
    for (int i = 0; i < (int)(sizeof sizes / sizeof sizes[0]); i++)
        if (rrtmgp_hip_abi_sizeof(i) != (int)sizes[i]) {
            printf("struct %d: header says %zu bytes, library %d\n", i, sizes[i], rrtmgp_hip_abi_sizeof(i));
            bad++;
        }
    /* the McICA stream is a pure function of (seed, column, g-point, band set, draw) */
    const double u = rrtmgp_hip_mcica_uniform(42u, 7, 3, 0, 1);
    if (!(u >= 0.0 && u < 1.0) || u != rrtmgp_hip_mcica_uniform(42u, 7, 3, 0, 1) || u == rrtmgp_hip_mcica_uniform(42u, 7, 3, 1, 1)) bad++;
    if (rrtmgp_hip_device_count() <= 0) {
        /* no device: a negative status and a message */
        rrtmgp_workspace *ws = NULL;
        const int rc = rrtmgp_hip_workspace_create(0, 16, 8, RRTMGP_F32, &ws);
        if (rc >= 0 || ws != NULL) bad++;
        if (rrtmgp_hip_last_error(msg, sizeof msg) != 0 || strlen(msg) == 0) bad++;
        if (argc > 1) { printf("a case file was given but there is no GPU\n"); bad++; }
    } else {
        bad += gray_isothermal();
        if (argc > 1) bad += spectral_case(argv[1]);
    }
    printf("%s %s: %d problem(s)\n", "libhip_rrtmgp", rrtmgp_hip_version(), bad);
    return bad;
}
