// views.hip — the three calls that take 2-D views as the reference passes them (compute_col_gas!, compute_relative_humidity!,
// compute_gray_heating_rate!): uncopied device views, host views gathered / scattered element-exactly.
#include "host.h"
#include "device.h"

namespace rrtmgp {

// ---- 2-D view arguments (rrtmgp_view2d) of compute_col_gas! / compute_relative_humidity! / compute_gray_heating_rate! ----
// The reference passes strided views (rows of layerdata: element stride 4; a row of Vmr.vmr: element stride ngas; domain
// views of level arrays: column stride = rows of the parent).  Device views are used in place.  A host view moves exactly
// its own elements, once, and reaches the kernel as a dense (n0, n1) array:
//   * dense                                  -> one copy;
//   * unit stride0, padded columns           -> one 2-D DMA (rows of n0 elements, stride1 apart), when a row is >= 64 bytes;
//   * anything else (stride0 > 1, C order)   -> gathered by the CPU into the workspace's page-locked bounce buffer (a few
//     threads for large arrays), one DMA; written views come back the same way and are scattered into place.
// Round 3 staged the memory SPAN of a strided view in and out instead (4x the bytes for a row of layerdata, and — with a
// sharded workspace and column-fastest views — overlapping spans that the shards' write-backs raced on).
struct ViewArg {
    const rrtmgp_view2d *v;
    size_t n0, n1;
    bool out;
    char *dev = nullptr;          // device address of element (0, 0)
    int64_t ds0 = 0, ds1 = 0;     // element strides of the array the kernel sees
    size_t pack_off = ~size_t(0); // offset in the bounce buffer when the view is gathered / scattered by the CPU
    bool dense() const { return v->stride0 == 1 && (size_t)v->stride1 == n0; }
};
template <typename T>
static void strided_copy_t(T *dense, T *strided, size_t n0, size_t n1, size_t s0, size_t s1, bool to_dense, size_t j0, size_t j1) {
    for (size_t j = j0; j < j1; j++) {
        T *d = dense + j * n0, *q = strided + j * s1;
        if (to_dense) for (size_t i = 0; i < n0; i++) d[i] = q[i * s0];
        else for (size_t i = 0; i < n0; i++) q[i * s0] = d[i];
    }
}
// Three helper threads that live as long as the library (the gather used to spawn and join up to three
// std::threads on EVERY call of a large strided view).  One gather at a time uses them (try_lock): a second caller — the
// shard workers of a multi-device workspace run concurrently, each bound to its GPU's CPUs — simply copies on its own
// thread, which is the parallelism a sharded call already has.
class CopyHelpers {
    static constexpr int N = 3;
    std::thread th[N];
    std::mutex mu, busy;
    std::condition_variable cv_go, cv_done;
    const std::function<void(size_t, size_t)> *job = nullptr;
    size_t lo[N] = {0}, hi[N] = {0};
    unsigned pending = 0, epoch = 0;
    bool quit = false, started = false;
    void run(int t) {
        unsigned seen = 0;
        std::unique_lock<std::mutex> lk(mu);
        for (;;) {
            cv_go.wait(lk, [&] { return quit || epoch != seen; });
            if (quit) return;
            seen = epoch;
            const auto *j = job;
            const size_t a = lo[t], b = hi[t];
            lk.unlock();
            if (b > a) (*j)(a, b);
            lk.lock();
            if (--pending == 0) cv_done.notify_all();
        }
    }
public:
    ~CopyHelpers() {
        { std::lock_guard<std::mutex> lk(mu); quit = true; }
        cv_go.notify_all();
        if (started) for (auto &t : th) t.join();
    }
    // part(j0, j1) over [0, n): the caller takes the first quarter, the helpers the rest; false = helpers are taken
    bool parallel(size_t n, const std::function<void(size_t, size_t)> &part) {
        std::unique_lock<std::mutex> own(busy, std::try_to_lock);
        if (!own.owns_lock()) return false;
        {
            std::lock_guard<std::mutex> lk(mu);
            if (!started) { for (int t = 0; t < N; t++) th[t] = std::thread([this, t] { run(t); }); started = true; }
            for (int t = 0; t < N; t++) { lo[t] = n * (t + 1) / (N + 1); hi[t] = n * (t + 2) / (N + 1); }
            job = &part; pending = N; epoch++;
        }
        cv_go.notify_all();
        part(0, n / (N + 1));
        std::unique_lock<std::mutex> lk(mu);
        cv_done.wait(lk, [&] { return pending == 0; });
        return true;
    }
};
static CopyHelpers g_copy_helpers;

static void strided_copy(void *dense, void *strided, size_t n0, size_t n1, size_t s0, size_t s1, size_t E, bool to_dense) {
    const std::function<void(size_t, size_t)> part = [&](size_t j0, size_t j1) {
        if (E == 4) strided_copy_t<uint32_t>((uint32_t *)dense, (uint32_t *)strided, n0, n1, s0, s1, to_dense, j0, j1);
        else strided_copy_t<uint64_t>((uint64_t *)dense, (uint64_t *)strided, n0, n1, s0, s1, to_dense, j0, j1);
    };
    // columns are disjoint in both layouts: stage_views refuses written views whose elements overlap
    if (n0 * n1 >= (size_t(1) << 20) && n1 >= 4 && g_copy_helpers.parallel(n1, part)) return;
    part(0, n1);
}
// distinct (i, j) -> distinct elements?  Sufficient for every view Julia's `view` / numpy basic slicing can make of a dense
// parent: one stride spans the other dimension entirely.
// (The last element of a line along the faster dimension sits at (n - 1) * stride: the next line may start right behind it.
// Round 5 asked for n * stride, which refused `view(A, 1:2:5, :)` of a 5-row parent: n0 = 3, s0 = 2, s1 = 5.  ADVICE r5.)
static bool view_is_injective(size_t n0, size_t n1, int64_t s0, int64_t s1) {
    if (n0 <= 1 || n1 <= 1) return (n0 <= 1 || s0 >= 1) && (n1 <= 1 || s1 >= 1);
    return (s0 >= 1 && (size_t)s1 > (n0 - 1) * (size_t)s0) || (s1 >= 1 && (size_t)s0 > (n1 - 1) * (size_t)s1);
}
struct ViewPack {   // the CPU-gathered views of one call
    rrtmgp_workspace *ws;
    ViewArg *a = nullptr;
    int n = 0;
    size_t E = 0, in_hi = 0, out_lo = ~size_t(0), out_hi = 0;
    // after the kernel: the written views' dense images come home in one DMA and are scattered into the caller's arrays
    int finish() {
        if (out_hi <= out_lo) return RRTMGP_OK;
        RR_HIP(hipMemcpyAsync(ws->bounce_h + out_lo, ws->bounce_d + out_lo, out_hi - out_lo, hipMemcpyDeviceToHost, ws->stream));
        ws->d2h_bytes += out_hi - out_lo;
        RR_HIP(hipStreamSynchronize(ws->stream));
        for (int i = 0; i < n; i++)
            if (a[i].v && a[i].out && a[i].pack_off != ~size_t(0))
                strided_copy(ws->bounce_h + a[i].pack_off, a[i].v->ptr, a[i].n0, a[i].n1, (size_t)a[i].v->stride0, (size_t)a[i].v->stride1, E, false);
        return RRTMGP_OK;
    }
};
static int stage_views(Stager &st, ViewPack &vp, int mem, ViewArg *a, int n, size_t E) {
    static const int slots[4] = {S_PLEV, S_PLAY, S_TLAY, S_AUX0};
    RR_CHECK(n <= 4, "internal: too many view arguments");
    vp.a = a; vp.n = n; vp.E = E;
    size_t pack = 0;
    for (int pass = 0; pass < 2; pass++)   // gathered inputs first, then the written views: the one upload covers inputs only
        for (int i = 0; i < n; i++) {
            if (!a[i].v || (pass == 1) != a[i].out) continue;
            RR_CHECK(a[i].v->ptr && a[i].v->stride0 >= 1 && a[i].v->stride1 >= 1, "view2d: null pointer or non-positive stride");
            RR_CHECK(!a[i].out || view_is_injective(a[i].n0, a[i].n1, a[i].v->stride0, a[i].v->stride1),
                     "view2d: a written view whose elements overlap (stride1 < n0 * stride0 and stride0 < n1 * stride1) is not supported");
            a[i].ds0 = a[i].v->stride0; a[i].ds1 = a[i].v->stride1;
            if (mem == RRTMGP_MEM_DEVICE) { a[i].dev = (char *)a[i].v->ptr; continue; }
            a[i].ds0 = 1; a[i].ds1 = (int64_t)a[i].n0;   // every host view reaches the kernel dense
            const bool rows2d = a[i].v->stride0 == 1 && (size_t)a[i].v->stride1 > a[i].n0 && a[i].n0 * E >= 64;
            if (!a[i].dense() && !rows2d) { a[i].pack_off = pack; pack += Stager::al(a[i].n0 * a[i].n1 * E); }
        }
    if (mem == RRTMGP_MEM_DEVICE) return RRTMGP_OK;
    if (pack) TRY(bounce_ensure(st.ws, pack));
    for (int i = 0; i < n; i++) {
        if (!a[i].v) continue;
        const size_t bytes = a[i].n0 * a[i].n1 * E, width = a[i].n0 * E, pitch = (size_t)a[i].v->stride1 * E;
        void *dev = nullptr;
        if (a[i].pack_off != ~size_t(0)) {
            dev = st.ws->bounce_d + a[i].pack_off;
            if (!a[i].out) {
                strided_copy(st.ws->bounce_h + a[i].pack_off, a[i].v->ptr, a[i].n0, a[i].n1, (size_t)a[i].v->stride0, (size_t)a[i].v->stride1, E, true);
                vp.in_hi = std::max(vp.in_hi, a[i].pack_off + bytes);
            } else {
                vp.out_lo = std::min(vp.out_lo, a[i].pack_off);
                vp.out_hi = std::max(vp.out_hi, a[i].pack_off + bytes);
            }
        } else if (a[i].dense()) {
            if (a[i].out) TRY(st.out(mem, slots[i], a[i].v->ptr, bytes, &dev));
            else TRY(st.in(mem, slots[i], a[i].v->ptr, bytes, (const void **)&dev));
        } else {   // unit stride0, padded columns: n1 rows of n0 elements
            if (a[i].out) TRY(st.out2d(slots[i], a[i].v->ptr, width, a[i].n1, pitch, &dev));
            else TRY(st.in2d(slots[i], a[i].v->ptr, width, a[i].n1, pitch, (const void **)&dev));
        }
        a[i].dev = (char *)dev;
    }
    if (vp.in_hi) {   // the gathered inputs: one DMA
        RR_HIP(hipMemcpyAsync(st.ws->bounce_d, st.ws->bounce_h, vp.in_hi, hipMemcpyHostToDevice, st.ws->stream));
        st.ws->h2d_bytes += vp.in_hi;
    }
    return RRTMGP_OK;
}
template <typename T>
static View2<T> dev_view(const ViewArg &a) {
    return a.v ? View2<T>{(T *)a.dev, a.ds0, a.ds1} : View2<T>{nullptr, 0, 0};
}

template <typename FT>
static int col_gas_t(rrtmgp_workspace *ws, int32_t mem, size_t ncol, size_t nlay, const rrtmgp_view2d *p_lev,
                     const rrtmgp_view2d *col_dry, const rrtmgp_params *ps, const rrtmgp_view2d *vmr_h2o, const void *lat) {
    Stager st{ws, {}};
    ViewPack vp{ws};
    ViewArg a[3] = {{p_lev, nlay + 1, ncol, false}, {col_dry, nlay, ncol, true}, {vmr_h2o, nlay, ncol, false}};
    TRY(stage_views(st, vp, mem, a, 3, sizeof(FT)));
    const FT *la;
    TRY(st.in(mem, S_LAT, lat, ncol * sizeof(FT), (const void **)&la));
    TRY(launch_col_gas<FT>(ws, (int)ncol, (int)nlay, dev_view<const FT>(a[0]), dev_view<FT>(a[1]), *ps, dev_view<const FT>(a[2]), la));
    TRY(st.finish());
    return vp.finish();
}

template <typename FT>
static int rel_hum_t(rrtmgp_workspace *ws, int32_t mem, size_t ncol, size_t nlay, const rrtmgp_view2d *rh,
                     const rrtmgp_view2d *p_lay, const rrtmgp_view2d *t_lay, const rrtmgp_params *ps, const rrtmgp_view2d *vmr_h2o) {
    Stager st{ws, {}};
    ViewPack vp{ws};
    ViewArg a[4] = {{rh, nlay, ncol, true}, {p_lay, nlay, ncol, false}, {t_lay, nlay, ncol, false}, {vmr_h2o, nlay, ncol, false}};
    TRY(stage_views(st, vp, mem, a, 4, sizeof(FT)));
    TRY(launch_rel_hum<FT>(ws, (int)ncol, (int)nlay, dev_view<FT>(a[0]), dev_view<const FT>(a[1]), dev_view<const FT>(a[2]), *ps,
                           dev_view<const FT>(a[3])));
    TRY(st.finish());
    return vp.finish();
}

template <typename FT>
static int heating_rate_t(rrtmgp_workspace *ws, int32_t mem, size_t ncol, size_t nlay, const rrtmgp_view2d *hr_lay,
                          const rrtmgp_view2d *p_lev, const rrtmgp_view2d *flux_net, double cp_d, double grav) {
    Stager st{ws, {}};
    ViewPack vp{ws};
    ViewArg a[3] = {{hr_lay, nlay, ncol, true}, {p_lev, nlay + 1, ncol, false}, {flux_net, nlay + 1, ncol, false}};
    TRY(stage_views(st, vp, mem, a, 3, sizeof(FT)));
    TRY(launch_heating_rate<FT>(ws, (int)ncol, (int)nlay, dev_view<FT>(a[0]), dev_view<const FT>(a[2]), dev_view<const FT>(a[1]),
                                grav, cp_d));
    TRY(st.finish());
    return vp.finish();
}


}  // namespace rrtmgp

using namespace rrtmgp;

extern "C" {

// extents of the three view-based calls against the workspace; a view of a multi-device workspace advanced to a shard's
// first column keeps its strides
static int check_view_extents(const rrtmgp_workspace *ws, int64_t ncol, int64_t nlay) {
    RR_CHECK(ncol >= 1 && nlay >= 1, "ncol and nlay must be positive");
    RR_CHECK(nlay <= ws->nlay, "nlay exceeds the workspace's");
    if (!ws->shards.empty() || ws->head) RR_CHECK(ncol == ws->ncol, "ncol differs from the (multi-device) workspace's");
    else RR_CHECK(ncol <= ws->ncol, "ncol exceeds the workspace's");
    return RRTMGP_OK;
}
static rrtmgp_view2d col_adv(const rrtmgp_view2d *v, size_t E, size_t c0) {
    rrtmgp_view2d r = *v;
    r.ptr = (char *)v->ptr + (size_t)v->stride1 * c0 * E;
    return r;
}

int rrtmgp_hip_compute_col_gas(rrtmgp_workspace *ws, int32_t mem, int64_t ncol, int64_t nlay, const rrtmgp_view2d *p_lev,
                               const rrtmgp_view2d *col_dry, const rrtmgp_params *params, const rrtmgp_view2d *vmr_h2o,
                               const void *lat) {
    RR_CHECK(ws && p_lev && col_dry && params, "null argument");
    if (vmr_h2o && !vmr_h2o->ptr) vmr_h2o = nullptr;  // an absent optional array may also be a view with a null pointer
    TRY(check_view_extents(ws, ncol, nlay));
    if (!ws->shards.empty()) {
        RR_CHECK(p_lev->ptr && col_dry->ptr, "view2d: null pointer");
        const size_t E = (size_t)ws->ftype;
        return multi_run(ws, [&](rrtmgp_workspace *sw, size_t c0, size_t nc) -> int {
            const ColumnSlice sl{E, c0};
            const rrtmgp_view2d pl = col_adv(p_lev, E, c0), cd = col_adv(col_dry, E, c0);
            rrtmgp_view2d h{};
            if (vmr_h2o) h = col_adv(vmr_h2o, E, c0);
            return rrtmgp_hip_compute_col_gas(sw, mem, (int64_t)nc, nlay, &pl, &cd, params, vmr_h2o ? &h : nullptr, sl.adv(lat, 1));
        }, mem == RRTMGP_MEM_DEVICE);
    }
    RR_HIP(hipSetDevice(ws->device));
    return ws->ftype == RRTMGP_F32 ? col_gas_t<float>(ws, mem, (size_t)ncol, (size_t)nlay, p_lev, col_dry, params, vmr_h2o, lat)
                                   : col_gas_t<double>(ws, mem, (size_t)ncol, (size_t)nlay, p_lev, col_dry, params, vmr_h2o, lat);
}

int rrtmgp_hip_compute_relative_humidity(rrtmgp_workspace *ws, int32_t mem, int64_t ncol, int64_t nlay,
                                         const rrtmgp_view2d *rh, const rrtmgp_view2d *p_lay, const rrtmgp_view2d *t_lay,
                                         const rrtmgp_params *params, const rrtmgp_view2d *vmr_h2o) {
    RR_CHECK(ws && rh && p_lay && t_lay && params && vmr_h2o, "null argument");
    TRY(check_view_extents(ws, ncol, nlay));
    if (!ws->shards.empty()) {
        RR_CHECK(rh->ptr && p_lay->ptr && t_lay->ptr && vmr_h2o->ptr, "view2d: null pointer");
        const size_t E = (size_t)ws->ftype;
        return multi_run(ws, [&](rrtmgp_workspace *sw, size_t c0, size_t nc) -> int {
            const rrtmgp_view2d r = col_adv(rh, E, c0), p = col_adv(p_lay, E, c0), t = col_adv(t_lay, E, c0), h = col_adv(vmr_h2o, E, c0);
            return rrtmgp_hip_compute_relative_humidity(sw, mem, (int64_t)nc, nlay, &r, &p, &t, params, &h);
        }, mem == RRTMGP_MEM_DEVICE);
    }
    RR_HIP(hipSetDevice(ws->device));
    return ws->ftype == RRTMGP_F32 ? rel_hum_t<float>(ws, mem, (size_t)ncol, (size_t)nlay, rh, p_lay, t_lay, params, vmr_h2o)
                                   : rel_hum_t<double>(ws, mem, (size_t)ncol, (size_t)nlay, rh, p_lay, t_lay, params, vmr_h2o);
}

int rrtmgp_hip_compute_gray_heating_rate(rrtmgp_workspace *ws, int32_t mem, int64_t ncol, int64_t nlay,
                                         const rrtmgp_view2d *hr_lay, const rrtmgp_view2d *p_lev,
                                         const rrtmgp_view2d *flux_net, double cp_d, double grav) {
    RR_CHECK(ws && hr_lay && p_lev && flux_net, "null argument");
    RR_CHECK(cp_d != 0.0, "cp_d must not be zero");
    TRY(check_view_extents(ws, ncol, nlay));
    if (!ws->shards.empty()) {
        RR_CHECK(hr_lay->ptr && p_lev->ptr && flux_net->ptr, "view2d: null pointer");
        const size_t E = (size_t)ws->ftype;
        return multi_run(ws, [&](rrtmgp_workspace *sw, size_t c0, size_t nc) -> int {
            const rrtmgp_view2d h = col_adv(hr_lay, E, c0), p = col_adv(p_lev, E, c0), f = col_adv(flux_net, E, c0);
            return rrtmgp_hip_compute_gray_heating_rate(sw, mem, (int64_t)nc, nlay, &h, &p, &f, cp_d, grav);
        }, mem == RRTMGP_MEM_DEVICE);
    }
    RR_HIP(hipSetDevice(ws->device));
    return ws->ftype == RRTMGP_F32 ? heating_rate_t<float>(ws, mem, (size_t)ncol, (size_t)nlay, hr_lay, p_lev, flux_net, cp_d, grav)
                                   : heating_rate_t<double>(ws, mem, (size_t)ncol, (size_t)nlay, hr_lay, p_lev, flux_net, cp_d, grav);
}

}  // extern "C"
