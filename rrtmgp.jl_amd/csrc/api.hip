// api.hip — host side of the C ABI declared in include/rrtmgp_hip.h:
// lookup re-layout + upload, workspaces, host<->HBM staging, solver dispatch.
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
#include "device.h"

namespace rrtmgp {

static thread_local std::string g_last_error;

int set_error(int code, const std::string &msg) {
    g_last_error = msg;
    return code;
}

const std::string &last_error_string() { return g_last_error; }

// Debug aid (RRTMGP_HIP_BACKTRACE_ON_ABORT=1): the C call stack of an abort() — e.g. one raised inside the HIP runtime —
// on stderr before the process dies.
static void abort_backtrace(int sig) {
    void *frames[64];
    const int n = backtrace(frames, 64);
    static const char msg[] = "rrtmgp_hip: SIGABRT, C call stack:\n";
    (void)!write(2, msg, sizeof msg - 1);
    backtrace_symbols_fd(frames, n, 2);
    signal(sig, SIG_DFL);
    raise(sig);
}
static const bool g_abort_hook = [] {
    if (getenv("RRTMGP_HIP_BACKTRACE_ON_ABORT")) signal(SIGABRT, abort_backtrace);
    return true;
}();

static std::atomic<int64_t> g_dev_allocs{0}, g_dev_frees{0}, g_host_regs{0};
hipError_t rr_malloc(void **p, size_t bytes) {
    const hipError_t e = hipMalloc(p, bytes);
    if (e == hipSuccess) g_dev_allocs++;
    return e;
}
hipError_t rr_free(void *p) {
    g_dev_frees++;
    return hipFree(p);
}

// ---- page-locked host arrays ---------------------------------------------------------------------------------------
// hipMemcpyAsync is real asynchronous DMA only from / to page-locked memory (from pageable memory the runtime stages
// through its own bounce buffers and the "asynchronous" copy blocks the host thread).  Page-locking has an EXPLICIT
// lifetime: the owner of a host array registers it (rrtmgp_hip_host_register, include/rrtmgp_hip.h) and unregisters it
// before the memory is freed; the bindings do that from the array's owner + finalizer (ext/RRTMGPHIPExt.jl `pin!`,
// rrtmgp.jl_amd/states.py).  The registry is process-wide, keyed by address range, reference counted per exact range.
//
// RRTMGP_HIP_AUTO_HOST_REGISTER=1 additionally registers, the first time a solve sees it, every host array of at least
// 32 MB (RRTMGP_HIP_HOST_REGISTER_MIN_BYTES) that nobody registered.  That is a heuristic and therefore opt-in: a
// registration that outlives its array keeps the OLD physical pages mapped for the GPU, and a new array that the allocator
// places at the same address with the same size cannot be told from the old one (the copies would silently use the
// old pages).  The floor is there because hipHostRegister locks whole PAGES: a smaller array comes from the allocator's
// heap and shares its first and last page with unrelated objects; when the runtime later locks and unlocks one of those
// for a pageable copy of its own, it unmaps the shared page under the registration and the next DMA through it dies with
// "Memory access fault by GPU" (seen with 100 KB numpy arrays: 4 of 24 runs of the test suite).  From 32 MB on glibc always
// mmaps, so the pages belong to the array alone.  Auto entries are released when a different array shows up on their
// pages, or after 64 registration passes that did not touch them.
//
// Concurrency: a solve marks the entries it verified as in use (`users`) until its results are home; an entry with users
// is never released by anybody (two workspaces on two threads may share the caller's state arrays), and a sub-range
// request (the slab of a shard or of a pipeline chunk) counts as registered only inside an explicit entry or one this
// workspace (or its multi-device head) is using.
struct PinEntry {
    size_t bytes;
    int refs;                 // explicit registrations of exactly this range (0: an auto entry)
    uint64_t last_used;       // g_pin_pass at the last touch
    std::vector<const rrtmgp_workspace *> users;  // solves in flight that verified it
    bool used_by(const rrtmgp_workspace *ws) const {
        return ws && std::find(users.begin(), users.end(), ws) != users.end();
    }
};
static std::mutex g_pin_mu;
static std::map<const char *, PinEntry> g_pins;
static uint64_t g_pin_pass = 0;

static void drop(std::map<const char *, PinEntry>::iterator &it) {
    (void)hipHostUnregister(const_cast<char *>(it->first));
    (void)hipGetLastError();
    it = g_pins.erase(it);
}
static bool droppable(const PinEntry &e) { return e.refs == 0 && e.users.empty(); }
// Looks at every registration that overlaps [p, p + bytes).  Returns true when a valid one covers the whole buffer;
// auto entries that overlap it otherwise belong to memory that has been re-allocated since and are released (unless a
// running solve is using them).  `use`: mark the covering entry as used by `ws`.  (Caller holds g_pin_mu.)
static bool settle(const rrtmgp_workspace *ws, const char *p, size_t bytes, bool use) {
    bool covered = false;
    auto it = g_pins.upper_bound(p);
    if (it != g_pins.begin()) --it;
    while (it != g_pins.end() && it->first < p + bytes) {
        const char *a = it->first, *b = a + it->second.bytes;
        if (b <= p) { ++it; continue; }
        PinEntry &e = it->second;
        const bool exact = a == p && e.bytes == bytes;
        const bool inside = a <= p && p + bytes <= b;
        if (exact || (inside && (e.refs > 0 || e.used_by(ws) || (ws && e.used_by(ws->head))))) {
            covered = true;
            e.last_used = g_pin_pass;
            if (use && ws && !e.used_by(ws)) e.users.push_back(ws);
            ++it;
        } else if (droppable(e)) {
            drop(it);
        } else {
            ++it;  // somebody else's live registration: left alone (the copy through it is then an ordinary pageable one)
        }
    }
    return covered;
}
void host_range_check(const rrtmgp_workspace *ws, const void *ptr, size_t bytes) {
    if (!ptr || !bytes) return;
    std::lock_guard<std::mutex> lock(g_pin_mu);
    if (!g_pins.empty()) (void)settle(ws, (const char *)ptr, bytes, false);
}
static size_t auto_register_min() {  // ~0: never (the default)
    static const size_t v = !getenv("RRTMGP_HIP_AUTO_HOST_REGISTER") || getenv("RRTMGP_HIP_NO_HOST_REGISTER") ? ~size_t(0)
                            : getenv("RRTMGP_HIP_HOST_REGISTER_MIN_BYTES") ? (size_t)atoll(getenv("RRTMGP_HIP_HOST_REGISTER_MIN_BYTES"))
                                                                            : (size_t)32 << 20;
    return v;
}
// One of the caller's WHOLE host arrays, seen by the registration pass of a solve: true when it is page-locked.
bool host_pin(rrtmgp_workspace *ws, const void *ptr, size_t bytes) {
    if (!ptr || !bytes) return false;
    const char *p = (const char *)ptr;
    std::lock_guard<std::mutex> lock(g_pin_mu);
    if (settle(ws, p, bytes, true)) return true;
    if (bytes < auto_register_min() || g_pins.size() >= 1024) return false;
    // nothing may be left under the new range (a live registration of other extents: stay pageable)
    auto it = g_pins.upper_bound(p);
    if (it != g_pins.begin()) --it;
    for (; it != g_pins.end() && it->first < p + bytes; ++it)
        if (it->first + it->second.bytes > p) return false;
    if (hipHostRegister(const_cast<char *>(p), bytes, hipHostRegisterDefault) != hipSuccess) {
        (void)hipGetLastError();
        return false;
    }
    g_host_regs++;
    g_pins.emplace(p, PinEntry{bytes, 0, g_pin_pass, {ws}});
    return true;
}
// A registration pass of `ws` (one per host-array solve) begins ...
void host_pin_begin(rrtmgp_workspace *ws) {
    std::lock_guard<std::mutex> lock(g_pin_mu);
    g_pin_pass++;
    (void)ws;
}
// ... and the solve it belongs to has its results home: `ws` stops using what it verified, and auto entries that no
// pass has touched for a while belong to arrays the caller no longer hands over — probably freed — and are released.
void host_pin_end(rrtmgp_workspace *ws) {
    std::lock_guard<std::mutex> lock(g_pin_mu);
    for (auto it = g_pins.begin(); it != g_pins.end();) {
        auto &u = it->second.users;
        u.erase(std::remove(u.begin(), u.end(), (const rrtmgp_workspace *)ws), u.end());
        if (droppable(it->second) && it->second.last_used + 64 < g_pin_pass) drop(it);
        else ++it;
    }
}
int host_register_explicit(void *ptr, size_t bytes) {
    if (!ptr || !bytes) return set_error(RRTMGP_EINVAL, "host_register: null pointer or zero size");
    const char *p = (const char *)ptr;
    std::lock_guard<std::mutex> lock(g_pin_mu);
    auto it = g_pins.upper_bound(p);
    if (it != g_pins.begin()) --it;
    while (it != g_pins.end() && it->first < p + bytes) {
        if (it->first + it->second.bytes <= p) { ++it; continue; }
        if (it->first == p && it->second.bytes == bytes) { it->second.refs++; return RRTMGP_OK; }  // (an auto entry becomes explicit)
        if (!droppable(it->second)) return set_error(RRTMGP_EINVAL, "host_register: the range overlaps a live registration of other extents");
        drop(it);
    }
    const hipError_t e = hipHostRegister(ptr, bytes, hipHostRegisterDefault);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        return set_error(RRTMGP_EHIP, std::string("hipHostRegister failed: ") + hipGetErrorString(e));
    }
    g_host_regs++;
    g_pins.emplace(p, PinEntry{bytes, 1, g_pin_pass, {}});
    return RRTMGP_OK;
}
int host_unregister_explicit(void *ptr) {
    std::lock_guard<std::mutex> lock(g_pin_mu);
    auto it = g_pins.find((const char *)ptr);
    if (it == g_pins.end() || it->second.refs == 0) return set_error(RRTMGP_EINVAL, "host_unregister: not a registered range");
    if (it->second.refs == 1 && !it->second.users.empty())
        return set_error(RRTMGP_EINVAL, "host_unregister: a running solve is using the range");
    if (--it->second.refs == 0) drop(it);
    return RRTMGP_OK;
}
int host_registered_count() {
    std::lock_guard<std::mutex> lock(g_pin_mu);
    return (int)g_pins.size();
}

int hip_fail(hipError_t e, const char *what, const char *file, int line) {
    char buf[512];
    snprintf(buf, sizeof buf, "HIP error %d (%s) in %s at %s:%d", (int)e, hipGetErrorString(e), what, file, line);
    return set_error(RRTMGP_EHIP, buf);
}

int stage_ensure(rrtmgp_workspace *ws, int slot, size_t bytes) {
    if ((int)ws->stage.size() <= slot) ws->stage.resize(slot + 1);
    DeviceBuffer &b = ws->stage[slot];
    if (b.bytes >= bytes && b.ptr) return RRTMGP_OK;
    if (b.ptr) RR_HIP(rr_free(b.ptr));
    b.ptr = nullptr;
    b.bytes = 0;
    RR_HIP(rr_malloc(&b.ptr, bytes ? bytes : 16));
    b.bytes = bytes;
    return RRTMGP_OK;
}

static int bounce_ensure(rrtmgp_workspace *ws, size_t bytes) {
    if (ws->bounce_bytes >= bytes) return RRTMGP_OK;
    RR_HIP(hipStreamSynchronize(ws->stream));
    if (ws->bounce_h) RR_HIP(hipHostFree(ws->bounce_h));
    if (ws->bounce_d) RR_HIP(rr_free(ws->bounce_d));
    ws->bounce_h = ws->bounce_d = nullptr;
    ws->bounce_bytes = 0;
    const size_t cap = std::max<size_t>(bytes + bytes / 4, 1 << 16);
    RR_HIP(hipHostMalloc((void **)&ws->bounce_h, cap, hipHostMallocDefault));
    g_host_regs++;  // page-locked host memory: counted with the registrations (rrtmgp_hip_allocation_counts)
    RR_HIP(rr_malloc((void **)&ws->bounce_d, cap));
    ws->bounce_bytes = cap;
    return RRTMGP_OK;
}

int scratch_ensure(rrtmgp_workspace *ws, size_t bytes, const Lane *lane) {
    DeviceBuffer &b = lane ? *lane->scratch : ws->scratch;
    if (b.bytes >= bytes && b.ptr) return RRTMGP_OK;
    if (b.ptr) {
        RR_HIP(hipStreamSynchronize(lane ? lane->stream : ws->stream));   // its last user
        RR_HIP(rr_free(b.ptr));
    }
    b.ptr = nullptr;
    b.bytes = 0;
    RR_HIP(rr_malloc(&b.ptr, bytes));
    b.bytes = bytes;
    return RRTMGP_OK;
}

int queue_ensure(rrtmgp_workspace *ws, int lane_idx, int **out) {
    int *&q = ws->col_queue[lane_idx & 1];
    if (!q) {
        void *p = nullptr;
        RR_HIP(rr_malloc(&p, 256));
        // once: the zeroing must be COMPLETE before any stream of this workspace launches a kernel that reads the counters
        // (the workspace streams are non-blocking: nothing orders them behind the null stream a plain hipMemset runs on;
        // an unordered first launch read uninitialised counters — a negative "next column" — and faulted)
        RR_HIP(hipMemset(p, 0, 256));
        RR_HIP(hipDeviceSynchronize());
        q = (int *)p;
    }
    *out = q;
    return RRTMGP_OK;
}

// The column kernels leave their queue counters at {0, 0} (queue_release, device.h) and rely on finding them so: a kernel that
// returned early, a new kernel that forgot queue_release, or a launch still running when the workspace moved to another
// stream would leave them dirty and every later launch on the lane would silently skip or repeat columns.
// Called where the lanes are known to be idle: reads both counters back and fails loudly instead.
int queue_check(rrtmgp_workspace *ws, bool reset_only) {
    for (int *q : ws->col_queue) {
        if (!q) continue;
        int v[2] = {0, 0};
        if (!reset_only) RR_HIP(hipMemcpy(v, q, sizeof v, hipMemcpyDeviceToHost));
        if (reset_only || v[0] || v[1]) {
            RR_HIP(hipMemset(q, 0, 256));
            RR_HIP(hipDeviceSynchronize());
        }
        if (v[0] || v[1])
            return set_error(RRTMGP_EHIP, "internal: a column kernel left its queue counters at {" + std::to_string(v[0]) + ", " +
                             std::to_string(v[1]) + "} (every kernel on the queue must end with queue_release); the results of the "
                             "last launch on this workspace are not to be trusted.  The counters have been reset");
    }
    return RRTMGP_OK;
}

// Number of workgroups for a one-workgroup-per-column kernel: every column gets its
// own group up to a few resident generations per CU, then groups stride over columns.
// Persistent grid of the column kernels: exactly the workgroups that are resident at once
// (registers AND LDS decide, so ask the runtime), each striding over columns.  A larger grid
// would run in waves of workgroups and leave the chip half empty during the last one.
int column_grid(rrtmgp_workspace *ws, int ncol, int threads, size_t lds_bytes, const void *kernel) {
    if (lds_bytes > 160 * 1024) return set_error(RRTMGP_EUNSUPPORTED, "column does not fit the 160 KB LDS");
    const auto key = std::make_pair(kernel, lds_bytes);
    auto it = ws->occupancy.find(key);
    if (it == ws->occupancy.end()) {  // once per kernel variant: the two runtime calls cost more than a small solve
        // the permission is per kernel, not per workspace: always ask for the whole LDS so that workspaces of
        // different sizes cannot lower each other's limit
        RR_HIP(hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        int n = 0;
        RR_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, kernel, threads, lds_bytes));
        it = ws->occupancy.emplace(key, std::max(n, 1)).first;
        if (getenv("RRTMGP_HIP_TRACE_LAUNCH"))  // tuning aid: what decides the resident workgroups of this kernel variant
            fprintf(stderr, "rrtmgp_hip: kernel %p: %d threads, %zu B LDS -> %d workgroups per CU\n", kernel, threads, lds_bytes, n);
    }
    static const int cap_per_cu = getenv("RRTMGP_HIP_MAX_WG_PER_CU") ? atoi(getenv("RRTMGP_HIP_MAX_WG_PER_CU")) : 0;  // tuning aid
    const int per_cu = cap_per_cu > 0 ? std::min(it->second, cap_per_cu) : it->second;
    const int cap = ws->n_cu * per_cu;
    return std::max(1, std::min(ncol, cap));
}

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

#define TRY(x)            \
    do {                  \
        int _rc = (x);    \
        if (_rc) return _rc; \
    } while (0)

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

static int select_device(int device) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return set_error(RRTMGP_ENODEV, "no HIP device visible");
    if (device < 0 || device >= n) return set_error(RRTMGP_EINVAL, "device index out of range");
    RR_HIP(hipSetDevice(device));
    return RRTMGP_OK;
}

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
// `nrghice`: roughness classes of the cloud lookup the solve uses; 0 = NO cloud lookup reads the cloud arrays (they are staged
// for the isothermal-layer preparation only: whichever of them the state carries, and `ice_rgh` is nobody's business).
// `aero_lookup` likewise for the aerosol arrays.
template <typename FT>
static int stage_state(Stager &st, const rrtmgp_atmos_state *as, bool use_cld, bool use_aero, bool lw, DevState<FT> &d,
                       int64_t nrghice = 1, StateRW rw = StateRW(), bool aero_lookup = true) {
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
static int stage_flux(Stager &st, const rrtmgp_flux_out *f, const rrtmgp_solve_opts *opts, size_t ncol, size_t nlev,
                      bool sw, DevFlux<FT> &d, size_t nbnd = 0, int so = 0 /* S_X_FLUX_UP - S_FLUX_UP: the second block of slots */,
                      const FT *staged_metric = nullptr /* the metric factors are on the device already */) {
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

static int check_common(rrtmgp_workspace *ws, const rrtmgp_lookup *gas, int want_sw, const rrtmgp_lookup *cld,
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

template <typename FT>
static int solve_lw_t(rrtmgp_workspace *ws, int twostream, const DevGas<FT> &lk, const DevCld<FT> *cld,
                      const DevAero<FT> *aero, int max_minor, const rrtmgp_atmos_state *as, const rrtmgp_lw_bcs *bcs,
                      const rrtmgp_flux_out *flux, const rrtmgp_solve_opts *opts, Stager *chunk = nullptr) {
    RR_CHECK(bcs && bcs->sfc_emis, "LwBCs: sfc_emis is required");
    RR_CHECK(!cld || cld->nband == lk.n_bnd, "cloud lookup band count differs from the gas lookup");
    RR_CHECK(!aero || aero->nband == lk.n_bnd, "aerosol lookup band count differs from the gas lookup");
    const int n_angles = opts ? opts->n_gauss_angles : 1;
    RR_CHECK(twostream || (n_angles >= 1 && n_angles <= 4), "n_gauss_angles must be 1..4");
    Stager own{ws, {}};
    Stager &st = chunk ? *chunk : own;
    DevState<FT> ds;
    TRY(stage_state(st, as, cld != nullptr, aero != nullptr, true, ds, cld ? cld->nrghice : 1));
    const FT *emis, *inc;
    TRY(st.in(bcs->mem, S_BC0, bcs->sfc_emis, (size_t)lk.n_bnd * as->ncol * sizeof(FT), (const void **)&emis));
    // inc_flux is the one array whose fastest dimension is ncol: a column range of it is a 2-D block
    const size_t inc_ld_in = bcs->inc_flux_ld > 0 ? (size_t)bcs->inc_flux_ld : (size_t)as->ncol;
    RR_CHECK(!bcs->inc_flux || inc_ld_in >= (size_t)as->ncol, "LwBCs.inc_flux_ld is smaller than ncol");
    int inc_ld = (int)as->ncol;  // what the kernel sees: host blocks are compacted while they are staged
    if (bcs->inc_flux && inc_ld_in != (size_t)as->ncol && bcs->mem == RRTMGP_MEM_HOST) {
        TRY(st.in2d(S_BC1, bcs->inc_flux, (size_t)as->ncol * sizeof(FT), (size_t)lk.n_gpt, inc_ld_in * sizeof(FT), (const void **)&inc));
    } else {
        if (bcs->inc_flux && bcs->mem == RRTMGP_MEM_DEVICE) inc_ld = (int)inc_ld_in;
        TRY(st.in(bcs->mem, S_BC1, bcs->inc_flux, (size_t)lk.n_gpt * as->ncol * sizeof(FT), (const void **)&inc));
    }
    DevFlux<FT> fl;
    TRY(stage_flux(st, flux, opts, as->ncol, as->nlay + 1, false, fl, twostream ? (size_t)lk.n_bnd : 0));
    if (st.pin_only) return RRTMGP_OK;
    if (chunk && !st.packed) {  // pipelined host path: the uploads ran on the copy stream
        RR_HIP(hipEventRecord(ws->ev_in[0], st.copy_stream()));
        RR_HIP(hipStreamWaitEvent(ws->stream, ws->ev_in[0], 0));
    }
    TRY(st.flush());  // packed small solve: the one upload
    TRY(launch_lw<FT>(ws, twostream, lk, cld, aero, ds, emis, inc, inc_ld, fl, n_angles, opts ? opts->seed : 0,
                      opts ? opts->col_offset : 0, max_minor));
    return chunk && !st.packed ? RRTMGP_OK : st.finish();
}

template <typename FT>
static int solve_sw_t(rrtmgp_workspace *ws, int twostream, const DevGas<FT> &lk, const DevCld<FT> *cld,
                      const DevAero<FT> *aero, int max_minor, const rrtmgp_atmos_state *as, const rrtmgp_sw_bcs *bcs,
                      const rrtmgp_flux_out *flux, const rrtmgp_solve_opts *opts, Stager *chunk = nullptr) {
    RR_CHECK(bcs && bcs->cos_zenith && bcs->toa_flux, "SwBCs: cos_zenith and toa_flux are required");
    RR_CHECK(!twostream || (bcs->sfc_alb_direct && bcs->sfc_alb_diffuse), "SwBCs: surface albedos are required");
    RR_CHECK(!cld || cld->nband == lk.n_bnd, "cloud lookup band count differs from the gas lookup");
    RR_CHECK(!aero || aero->nband == lk.n_bnd, "aerosol lookup band count differs from the gas lookup");
    RR_CHECK(flux && flux->flux_dn_dir, "FluxSW: flux_dn_dir is required");
    Stager own{ws, {}};
    Stager &st = chunk ? *chunk : own;
    DevState<FT> ds;
    TRY(stage_state(st, as, cld != nullptr, aero != nullptr, false, ds, cld ? cld->nrghice : 1));
    const FT *mu0, *toa, *adir, *adif;
    const size_t E = sizeof(FT), ncol = as->ncol;
    TRY(st.in(bcs->mem, S_BC0, bcs->cos_zenith, ncol * E, (const void **)&mu0));
    TRY(st.in(bcs->mem, S_BC1, bcs->toa_flux, ncol * E, (const void **)&toa));
    TRY(st.in(bcs->mem, S_BC2, bcs->sfc_alb_direct, (size_t)lk.n_bnd * ncol * E, (const void **)&adir));
    TRY(st.in(bcs->mem, S_BC3, bcs->sfc_alb_diffuse, (size_t)lk.n_bnd * ncol * E, (const void **)&adif));
    DevFlux<FT> fl;
    TRY(stage_flux(st, flux, opts, ncol, as->nlay + 1, true, fl, twostream ? (size_t)lk.n_bnd : 0));
    if (st.pin_only) return RRTMGP_OK;
    if (chunk && !st.packed) {  // pipelined host path: the uploads ran on the copy stream
        RR_HIP(hipEventRecord(ws->ev_in[0], st.copy_stream()));
        RR_HIP(hipStreamWaitEvent(ws->stream, ws->ev_in[0], 0));
    }
    TRY(st.flush());  // packed small solve: the one upload
    TRY(launch_sw<FT>(ws, twostream, lk, cld, aero, ds, mu0, toa, adir, adif, fl, opts ? opts->seed : 0,
                      opts ? opts->col_offset : 0, max_minor));
    return chunk && !st.packed ? RRTMGP_OK : st.finish();
}

// ---- pipelined host path ------------------------------------------------------------------
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

static void slice_state(rrtmgp_atmos_state &a, const ColumnSlice &s, size_t nc) {
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
static void slice_opts(rrtmgp_solve_opts &o, const ColumnSlice &s, size_t nlev) {
    o.metric_scaling = s.adv(o.metric_scaling, nlev);
    o.col_offset += (int64_t)s.c0;
}
static void slice_flux_arrays(rrtmgp_flux_out &f, const ColumnSlice &s, size_t nlev, size_t ncol_total) {
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
static void slice_flux(rrtmgp_flux_out &f, rrtmgp_solve_opts &o, const ColumnSlice &s, size_t nlev, size_t ncol_total) {
    slice_flux_arrays(f, s, nlev, ncol_total);
    slice_opts(o, s, nlev);
}

// `ncol` = columns of the array being sliced (the leading dimension of its inc_flux unless the caller gave one)
static void slice_lw_bcs(rrtmgp_lw_bcs &b, const ColumnSlice &s, size_t nbnd, size_t ncol) {
    b.sfc_emis = s.adv(b.sfc_emis, nbnd);
    if (b.inc_flux) {
        if (b.inc_flux_ld <= 0) b.inc_flux_ld = (int32_t)ncol;
        b.inc_flux = s.adv(b.inc_flux, 1);
    }
}
static void slice_sw_bcs(rrtmgp_sw_bcs &b, const ColumnSlice &s, size_t nbnd) {
    b.cos_zenith = s.adv(b.cos_zenith, 1); b.toa_flux = s.adv(b.toa_flux, 1);
    b.sfc_alb_direct = s.adv(b.sfc_alb_direct, nbnd); b.sfc_alb_diffuse = s.adv(b.sfc_alb_diffuse, nbnd);
}
static void slice_gray(rrtmgp_gray_state &g, const ColumnSlice &s, size_t nc) {
    const size_t nlay = g.nlay, nlev = nlay + 1;
    g.ncol = (int64_t)nc;
    g.lat = s.adv(g.lat, 1); g.t_sfc = s.adv(g.t_sfc, 1);
    g.p_lay = s.adv(g.p_lay, nlay); g.t_lay = s.adv(g.t_lay, nlay);
    g.p_lev = s.adv(g.p_lev, nlev); g.t_lev = s.adv(g.t_lev, nlev);
}

// What a multi-device workspace can shard in one call (include/rrtmgp_hip.h): everything whose slowest dimension is ncol.
static int check_multi(const rrtmgp_workspace *ws, int state_mem, int bcs_mem, const rrtmgp_flux_out *flux,
                       const rrtmgp_solve_opts *opts, const void *inc_flux) {
    if (ws->shards.size() <= 1) return RRTMGP_OK;
    bool one_device = true;
    for (auto *s : ws->shards) one_device = one_device && s->device == ws->shards[0]->device;
    const bool any_dev = state_mem == RRTMGP_MEM_DEVICE || bcs_mem == RRTMGP_MEM_DEVICE || (flux && flux->mem == RRTMGP_MEM_DEVICE) ||
                         (opts && opts->metric_scaling && opts->metric_mem == RRTMGP_MEM_DEVICE);
    if (any_dev && !one_device) return set_error(RRTMGP_EINVAL, "a workspace spanning several devices needs host arrays");
    return RRTMGP_OK;
}

static bool host_pipeline_applies(const rrtmgp_atmos_state *as, int bcs_mem, const rrtmgp_flux_out *flux,
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
static int fork_join_events(rrtmgp_workspace *ws) {
    for (int i = 0; i < 3; i++)
        if (!ws->ev_k[i]) RR_HIP(hipEventCreateWithFlags(&ws->ev_k[i], hipEventDisableTiming));
    return RRTMGP_OK;
}
static int pipeline_resources(rrtmgp_workspace *ws) {
    if (ws->copy_stream) return RRTMGP_OK;
    RR_HIP(hipStreamCreateWithFlags(&ws->copy_stream, hipStreamNonBlocking));
    for (int i = 0; i < 2; i++) RR_HIP(hipEventCreateWithFlags(&ws->ev_in[i], hipEventDisableTiming));
    TRY(fork_join_events(ws));
    ws->stage_alt.resize(ws->stage.size());
    return RRTMGP_OK;
}
// second compute lane of a short Layer-2 step (step_t)
static int lane_resources(rrtmgp_workspace *ws) {
    if (ws->alt_stream) return RRTMGP_OK;
    RR_HIP(hipStreamCreateWithFlags(&ws->alt_stream, hipStreamNonBlocking));
    return fork_join_events(ws);
}

// `chunk(slice, n_columns, stager, vmr_is_gm)` stages and launches the columns [slice.c0, slice.c0 + n_columns)
template <typename F>
static int run_column_pipeline(rrtmgp_workspace *ws, size_t ncol, size_t E, bool vmr_gm, F &&chunk) {
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
static int run_host_pipeline(rrtmgp_workspace *ws, const rrtmgp_atmos_state *as, const rrtmgp_flux_out *flux,
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

// Below this many staged host bytes a solve goes through the bounce buffer (Stager::packed): per-array DMA commands
// cost ~15 us each whatever their size, a host memcpy ~0.1 us per KB (RRTMGP_HIP_HOST_PACK_BYTES overrides, 0 = never).
static size_t host_pack_max() {
    static const size_t v = getenv("RRTMGP_HIP_HOST_PACK_BYTES") ? (size_t)atoll(getenv("RRTMGP_HIP_HOST_PACK_BYTES")) : (size_t)2 << 20;
    return v;
}

// The registration pass over the caller's WHOLE host arrays: which of them are page-locked (and, opt-in, locking the large
// ones on first sight).  What it verifies stays marked as used by `ws` until the PinScope of the solve ends.
struct PinScope {
    rrtmgp_workspace *ws;
    ~PinScope() { host_pin_end(ws); }
};
template <typename FT>
static int pin_lw(rrtmgp_workspace *ws, int twostream, const DevGas<FT> &lk, const DevCld<FT> *cld, const DevAero<FT> *aero,
                  int max_minor, const rrtmgp_atmos_state *as, const rrtmgp_lw_bcs *bcs, const rrtmgp_flux_out *flux,
                  const rrtmgp_solve_opts *opts, size_t *need = nullptr) {
    host_pin_begin(ws);
    Stager pin{ws, {}};
    pin.pin_only = true;
    const int rc = solve_lw_t<FT>(ws, twostream, lk, cld, aero, max_minor, as, bcs, flux, opts, &pin);
    if (need) *need = pin.need;  // host bytes the solve stages (what the packed small-solve path sizes its bounce buffer from)
    return rc;
}
template <typename FT>
static int pin_sw(rrtmgp_workspace *ws, int twostream, const DevGas<FT> &lk, const DevCld<FT> *cld, const DevAero<FT> *aero,
                  int max_minor, const rrtmgp_atmos_state *as, const rrtmgp_sw_bcs *bcs, const rrtmgp_flux_out *flux,
                  const rrtmgp_solve_opts *opts, size_t *need = nullptr) {
    host_pin_begin(ws);
    Stager pin{ws, {}};
    pin.pin_only = true;
    const int rc = solve_sw_t<FT>(ws, twostream, lk, cld, aero, max_minor, as, bcs, flux, opts, &pin);
    if (need) *need = pin.need;  // host bytes the solve stages (what the packed small-solve path sizes its bounce buffer from)
    return rc;
}

template <typename FT>
static int solve_lw_host(rrtmgp_workspace *ws, int twostream, const DevGas<FT> &lk, const DevCld<FT> *cld,
                         const DevAero<FT> *aero, int max_minor, const rrtmgp_atmos_state *as, const rrtmgp_lw_bcs *bcs,
                         const rrtmgp_flux_out *flux, const rrtmgp_solve_opts *opts) {
    size_t need = 0;
    PinScope scope{ws};
    TRY(pin_lw<FT>(ws, twostream, lk, cld, aero, max_minor, as, bcs, flux, opts, &need));
    if (!bcs || !host_pipeline_applies(as, bcs->mem, flux, opts)) {
        if (need && need <= host_pack_max()) {   // small solve: one bounce buffer, one DMA each way
            TRY(bounce_ensure(ws, need));
            Stager st{ws, {}};
            st.packed = true;
            return solve_lw_t<FT>(ws, twostream, lk, cld, aero, max_minor, as, bcs, flux, opts, &st);
        }
        return solve_lw_t<FT>(ws, twostream, lk, cld, aero, max_minor, as, bcs, flux, opts);
    }
    return run_host_pipeline(ws, as, flux, opts, sizeof(FT),
                             [&](rrtmgp_atmos_state &a, rrtmgp_flux_out &f, rrtmgp_solve_opts &o, const ColumnSlice &sl, Stager &st) {
                                 rrtmgp_lw_bcs b = *bcs;
                                 slice_lw_bcs(b, sl, (size_t)lk.n_bnd, (size_t)as->ncol);
                                 return solve_lw_t<FT>(ws, twostream, lk, cld, aero, max_minor, &a, &b, &f, &o, &st);
                             });
}

template <typename FT>
static int solve_sw_host(rrtmgp_workspace *ws, int twostream, const DevGas<FT> &lk, const DevCld<FT> *cld,
                         const DevAero<FT> *aero, int max_minor, const rrtmgp_atmos_state *as, const rrtmgp_sw_bcs *bcs,
                         const rrtmgp_flux_out *flux, const rrtmgp_solve_opts *opts) {
    size_t need = 0;
    PinScope scope{ws};
    TRY(pin_sw<FT>(ws, twostream, lk, cld, aero, max_minor, as, bcs, flux, opts, &need));
    if (!bcs || !host_pipeline_applies(as, bcs->mem, flux, opts)) {
        if (need && need <= host_pack_max()) {
            TRY(bounce_ensure(ws, need));
            Stager st{ws, {}};
            st.packed = true;
            return solve_sw_t<FT>(ws, twostream, lk, cld, aero, max_minor, as, bcs, flux, opts, &st);
        }
        return solve_sw_t<FT>(ws, twostream, lk, cld, aero, max_minor, as, bcs, flux, opts);
    }
    return run_host_pipeline(ws, as, flux, opts, sizeof(FT),
                             [&](rrtmgp_atmos_state &a, rrtmgp_flux_out &f, rrtmgp_solve_opts &o, const ColumnSlice &sl, Stager &st) {
                                 rrtmgp_sw_bcs b = *bcs;
                                 b.cos_zenith = sl.adv(b.cos_zenith, 1); b.toa_flux = sl.adv(b.toa_flux, 1);
                                 b.sfc_alb_direct = sl.adv(b.sfc_alb_direct, (size_t)lk.n_bnd);
                                 b.sfc_alb_diffuse = sl.adv(b.sfc_alb_diffuse, (size_t)lk.n_bnd);
                                 return solve_sw_t<FT>(ws, twostream, lk, cld, aero, max_minor, &a, &b, &f, &o, &st);
                             });
}

// prepare_atmosphere! (update_fluxes.jl:252-281): stage every array the cascade touches as in/out
template <typename FT>
static int prepare_t(rrtmgp_workspace *ws, const rrtmgp_atmos_state *as, const rrtmgp_params *ps,
                     const rrtmgp_prepare_opts *o) {
    const size_t E = sizeof(FT), ncol = as->ncol, nlay = as->nlay, nlev = nlay + 1, n2 = nlay * ncol * E;
    RR_CHECK(as->layerdata && as->p_lev && as->t_lev && as->t_sfc, "prepare_atmosphere: missing array");
    Stager st{ws, {}};
    PrepView<FT> v{};
    v.ncol = (int)ncol; v.nlay = (int)nlay; v.ls = 4;
    const int mem = as->mem;
    FT *ld;
    TRY(st.inout(mem, S_LAYERDATA, as->layerdata, 4 * n2, (void **)&ld));
    v.col_dry = ld; v.p_lay = ld + 1; v.t_lay = ld + 2; v.rel_hum = ld + 3;
    TRY(st.inout(mem, S_PLEV, as->p_lev, nlev * ncol * E, (void **)&v.p_lev));
    TRY(st.inout(mem, S_TLEV, as->t_lev, nlev * ncol * E, (void **)&v.t_lev));
    TRY(st.in(mem, S_TSFC, as->t_sfc, ncol * E, (const void **)&v.t_sfc));
    TRY(st.in(mem, S_LAT, as->lat, ncol * E, (const void **)&v.lat));
    if (as->vmr_kind == RRTMGP_VMR_GM) {
        RR_CHECK(as->vmr_h2o, "VmrGM: vmr_h2o is required");
        TRY(st.inout(mem, S_VMR_H2O, as->vmr_h2o, n2, (void **)&v.vmr_h2o));
        TRY(st.inout(mem, S_VMR_O3, as->vmr_o3, n2, (void **)&v.vmr_o3));
        v.hs = 1;
    } else {
        RR_CHECK(as->vmr && o->idx_h2o >= 1 && o->idx_h2o <= as->ngas, "Vmr: idx_h2o out of range");
        TRY(st.inout(mem, S_VMR, as->vmr, (size_t)as->ngas * n2, (void **)&v.vmr_full));
        v.ngas = (int)as->ngas; v.hs = (int)as->ngas; v.vmr_h2o = v.vmr_full + (o->idx_h2o - 1);
    }
    if ((o->steps & RRTMGP_PREP_ISOTHERMAL) && o->isothermal_boundary_layer) {
        const void *cl[5] = {as->cld_r_eff_liq, as->cld_r_eff_ice, as->cld_path_liq, as->cld_path_ice, as->cld_frac};
        const int slot[5] = {S_CLD_RL, S_CLD_RI, S_CLD_PL, S_CLD_PI, S_CLD_F};
        for (int i = 0; i < 5; i++) TRY(st.inout(mem, slot[i], cl[i], n2, (void **)&v.cld[i]));
        TRY(st.inout(mem, S_AERO_SIZE, as->aero_size, RRTMGP_N_AEROSOLS * n2, (void **)&v.aero[0]));
        TRY(st.inout(mem, S_AERO_MASS, as->aero_mass, RRTMGP_N_AEROSOLS * n2, (void **)&v.aero[1]));
    }
    TRY(st.in(o->z_mem, S_ZC, o->center_z, n2, (const void **)&v.center_z));
    TRY(st.in(o->z_mem, S_ZF, o->face_z, nlev * ncol * E, (const void **)&v.face_z));
    TRY(launch_prepare<FT>(ws, v, *ps, *o, false));
    return st.finish();
}

template <typename FT>
static int prepare_gray_t(rrtmgp_workspace *ws, const rrtmgp_gray_state *gs, const rrtmgp_params *ps,
                          const rrtmgp_prepare_opts *o) {
    const size_t E = sizeof(FT), ncol = gs->ncol, nlay = gs->nlay, nlev = nlay + 1;
    RR_CHECK(gs->p_lay && gs->p_lev && gs->t_lay && gs->t_lev && gs->t_sfc, "prepare_atmosphere (gray): missing array");
    Stager st{ws, {}};
    PrepView<FT> v{};
    v.ncol = (int)ncol; v.nlay = (int)nlay; v.ls = 1;
    TRY(st.inout(gs->mem, S_PLAY, gs->p_lay, nlay * ncol * E, (void **)&v.p_lay));
    TRY(st.inout(gs->mem, S_TLAY, gs->t_lay, nlay * ncol * E, (void **)&v.t_lay));
    TRY(st.inout(gs->mem, S_PLEV, gs->p_lev, nlev * ncol * E, (void **)&v.p_lev));
    TRY(st.inout(gs->mem, S_TLEV, gs->t_lev, nlev * ncol * E, (void **)&v.t_lev));
    TRY(st.in(gs->mem, S_TSFC, gs->t_sfc, ncol * E, (const void **)&v.t_sfc));
    TRY(st.in(o->z_mem, S_ZC, o->center_z, nlay * ncol * E, (const void **)&v.center_z));
    TRY(st.in(o->z_mem, S_ZF, o->face_z, nlev * ncol * E, (const void **)&v.face_z));
    rrtmgp_prepare_opts og = *o;
    og.steps &= ~RRTMGP_PREP_COL_DRY;
    TRY(launch_prepare<FT>(ws, v, *ps, og, true));
    return st.finish();
}

template <typename FT>
static int solve_gray_lw_t(rrtmgp_workspace *ws, int twostream, const rrtmgp_gray_state *gs, const rrtmgp_lw_bcs *bcs,
                           const rrtmgp_flux_out *flux, const rrtmgp_solve_opts *opts) {
    const size_t E = sizeof(FT), ncol = gs->ncol, nlay = gs->nlay, nlev = nlay + 1;
    Stager st{ws, {}};
    const FT *lat, *p_lay, *p_lev, *t_lay, *t_lev, *t_sfc, *emis, *inc;
    TRY(st.in(gs->mem, S_LAT, gs->lat, ncol * E, (const void **)&lat));
    TRY(st.in(gs->mem, S_PLAY, gs->p_lay, nlay * ncol * E, (const void **)&p_lay));
    TRY(st.in(gs->mem, S_PLEV, gs->p_lev, nlev * ncol * E, (const void **)&p_lev));
    TRY(st.in(gs->mem, S_TLAY, gs->t_lay, nlay * ncol * E, (const void **)&t_lay));
    TRY(st.in(gs->mem, S_TLEV, gs->t_lev, nlev * ncol * E, (const void **)&t_lev));
    TRY(st.in(gs->mem, S_TSFC, gs->t_sfc, ncol * E, (const void **)&t_sfc));
    TRY(st.in(bcs->mem, S_BC0, bcs->sfc_emis, ncol * E, (const void **)&emis));
    TRY(st.in(bcs->mem, S_BC1, bcs->inc_flux, ncol * E, (const void **)&inc));
    DevFlux<FT> fl;
    TRY(stage_flux(st, flux, opts, ncol, nlev, false, fl));
    GrayArgs ga;
    ga.otp_kind = gs->otp_kind;
    for (int i = 0; i < 5; i++) ga.otp[i] = gs->otp[i];
    ga.stefan = gs->stefan;
    TRY(launch_gray_lw<FT>(ws, twostream, (int)ncol, (int)nlay, ga, lat, p_lay, p_lev, t_lay, t_lev, t_sfc, emis, inc, fl));
    return st.finish();
}

template <typename FT>
static int solve_gray_sw_t(rrtmgp_workspace *ws, int twostream, const rrtmgp_gray_state *gs, const rrtmgp_sw_bcs *bcs,
                           const rrtmgp_flux_out *flux, const rrtmgp_solve_opts *opts) {
    const size_t E = sizeof(FT), ncol = gs->ncol, nlay = gs->nlay, nlev = nlay + 1;
    Stager st{ws, {}};
    const FT *p_lay, *p_lev, *mu0, *toa, *adir, *adif;
    TRY(st.in(gs->mem, S_PLAY, gs->p_lay, nlay * ncol * E, (const void **)&p_lay));
    TRY(st.in(gs->mem, S_PLEV, gs->p_lev, nlev * ncol * E, (const void **)&p_lev));
    TRY(st.in(bcs->mem, S_BC0, bcs->cos_zenith, ncol * E, (const void **)&mu0));
    TRY(st.in(bcs->mem, S_BC1, bcs->toa_flux, ncol * E, (const void **)&toa));
    TRY(st.in(bcs->mem, S_BC2, bcs->sfc_alb_direct, ncol * E, (const void **)&adir));
    TRY(st.in(bcs->mem, S_BC3, bcs->sfc_alb_diffuse, ncol * E, (const void **)&adif));
    DevFlux<FT> fl;
    TRY(stage_flux(st, flux, opts, ncol, nlev, true, fl));
    GrayArgs ga;
    ga.otp_kind = gs->otp_kind;
    for (int i = 0; i < 5; i++) ga.otp[i] = gs->otp[i];
    ga.stefan = gs->stefan;
    TRY(launch_gray_sw<FT>(ws, twostream, (int)ncol, (int)nlay, ga, p_lay, p_lev, mu0, toa, adir, adif, fl));
    return st.finish();
}

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

// One chunk (or the whole batch) of the step: stage everything once, [prepare] -> LW -> SW -> net sums, copy back.
template <typename FT>
static int step_t(rrtmgp_workspace *ws, const StepLookups<FT> &L, const rrtmgp_update_fluxes_args *a, Stager *chunk = nullptr) {
    const rrtmgp_atmos_state *as = a->as;
    const rrtmgp_solve_opts *opts = a->opts;
    const rrtmgp_prepare_opts *po = a->prepare;
    const size_t E = sizeof(FT), ncol = as->ncol, nlay = as->nlay, nlev = nlay + 1;
    const int twostream_lw = a->lw_solver == RRTMGP_LW_TWOSTREAM;
    const int n_angles = opts ? opts->n_gauss_angles : 1;
    RR_CHECK(twostream_lw || (n_angles >= 1 && n_angles <= 4), "n_gauss_angles must be 1..4");
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
    const bool use_cld = L.lw_cld || L.sw_cld, use_aero = L.lw_aero || L.sw_aero;
    const bool prep = po != nullptr;
    const bool iso = prep && (po->steps & RRTMGP_PREP_ISOTHERMAL) && po->isothermal_boundary_layer;
    const int mem = as->mem;

    // ---- the state, once.  stage_state as the LW solve stages it (t_lev included; cld_cover = the LW cover) ...
    DevState<FT> ds;
    const int64_t nrgh = std::min<int64_t>(L.lw_cld ? L.lw_cld->nrghice : INT32_MAX, L.sw_cld ? L.sw_cld->nrghice : INT32_MAX);
    rrtmgp_atmos_state as_lw = *as;
    if (!L.lw_cld) as_lw.cld_cover_lw = nullptr;   // a LW solve without clouds writes no cover: nothing to bring back
    // The isothermal boundary layer fills the extra layer of EVERY cloud / aerosol array the state carries, whatever the
    // radiation method reads (prepare_t does; AllSkyRadiation with aerosol_radiation = false on a state with an
    // AerosolState): stage them for the preparation even when no lookup asks for them.
    const bool st_cld = use_cld || (iso && as->cld_frac), st_aero = use_aero || (iso && as->aero_mass);
    TRY(stage_state(st, &as_lw, st_cld, st_aero, true, ds, use_cld ? nrgh : 0, StateRW{prep, iso}, use_aero));
    // ... and what only the SW solve writes
    DevState<FT> ds_sw = ds;
    ds_sw.cld_cover = nullptr;
    if (L.sw_cld) TRY(st.out(mem, S_CLD_COVER2, as->cld_cover_sw, ncol * E, (void **)&ds_sw.cld_cover));
    if (L.sw_aero) {
        RR_CHECK((as->aod_sw_ext == nullptr) == (as->aod_sw_sca == nullptr), "aod_sw_ext and aod_sw_sca go together");
        TRY(st.out(mem, S_AOD_EXT, as->aod_sw_ext, ncol * E, (void **)&ds_sw.aod_sw_ext));
        TRY(st.out(mem, S_AOD_SCA, as->aod_sw_sca, ncol * E, (void **)&ds_sw.aod_sw_sca));
    }
    // ... and what only the preparation touches
    PrepView<FT> pv{};
    if (prep) {
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
    }

    // ---- boundary conditions
    const FT *emis, *inc, *mu0, *toa, *adir, *adif;
    const rrtmgp_lw_bcs *bl = a->bcs_lw;
    const rrtmgp_sw_bcs *bs = a->bcs_sw;
    TRY(st.in(bl->mem, S_LW_BC0, bl->sfc_emis, (size_t)L.lw->n_bnd * ncol * E, (const void **)&emis));
    const size_t inc_ld_in = bl->inc_flux_ld > 0 ? (size_t)bl->inc_flux_ld : ncol;
    RR_CHECK(!bl->inc_flux || inc_ld_in >= ncol, "LwBCs.inc_flux_ld is smaller than ncol");
    int inc_ld = (int)ncol;
    if (bl->inc_flux && inc_ld_in != ncol && bl->mem == RRTMGP_MEM_HOST) {
        TRY(st.in2d(S_LW_BC1, bl->inc_flux, ncol * E, (size_t)L.lw->n_gpt, inc_ld_in * E, (const void **)&inc));
    } else {
        if (bl->inc_flux && bl->mem == RRTMGP_MEM_DEVICE) inc_ld = (int)inc_ld_in;
        TRY(st.in(bl->mem, S_LW_BC1, bl->inc_flux, (size_t)L.lw->n_gpt * ncol * E, (const void **)&inc));
    }
    TRY(st.in(bs->mem, S_BC0, bs->cos_zenith, ncol * E, (const void **)&mu0));
    TRY(st.in(bs->mem, S_BC1, bs->toa_flux, ncol * E, (const void **)&toa));
    TRY(st.in(bs->mem, S_BC2, bs->sfc_alb_direct, (size_t)L.sw->n_bnd * ncol * E, (const void **)&adir));
    TRY(st.in(bs->mem, S_BC3, bs->sfc_alb_diffuse, (size_t)L.sw->n_bnd * ncol * E, (const void **)&adif));

    // ---- outputs
    DevFlux<FT> fl_lw, fl_sw;
    TRY(stage_flux(st, a->flux_lw, opts, ncol, nlev, false, fl_lw, twostream_lw ? (size_t)L.lw->n_bnd : (a->flux_lw->clear_flux_up ? 1 : 0)));
    TRY(stage_flux(st, a->flux_sw, opts, ncol, nlev, true, fl_sw, (size_t)L.sw->n_bnd, S_X_FLUX_UP - S_FLUX_UP, fl_lw.metric));
    if (!fl_lw.metric) fl_sw.metric = nullptr;
    // (decided from the caller's descriptors: in the registration pass the staged pointers are all null)
    const bool band_lw = a->flux_lw->band_flux_up != nullptr, band_sw = a->flux_sw->band_flux_up != nullptr;
    RR_CHECK(twostream_lw || !band_lw, "per-band fluxes are only available from the two-stream solvers");
    const bool diag_lw = a->flux_lw->clear_flux_up != nullptr, diag_sw = a->flux_sw->clear_flux_up != nullptr;
    RR_CHECK(!(diag_lw || diag_sw) || (L.lw_cld && L.sw_cld), "the clear-sky diagnostic needs the cloud lookups (AllSkyRadiationWithClearSkyDiagnostics)");
    RR_CHECK(!a->clear_net_flux || (diag_lw && diag_sw), "clear_net_flux needs the clear-sky fluxes of both solvers");
    FT *net = nullptr, *clear_net = nullptr;
    TRY(st.out(a->flux_lw->mem, S_NET, a->net_flux, ncol * nlev * E, (void **)&net));
    TRY(st.out(a->flux_lw->mem, S_CLR_NETSUM, a->clear_net_flux, ncol * nlev * E, (void **)&clear_net));
    if (st.pin_only) return RRTMGP_OK;

    if (chunk && !st.packed) {  // pipelined host path: the uploads ran on the copy stream
        RR_HIP(hipEventRecord(ws->ev_in[0], st.copy_stream()));
        RR_HIP(hipStreamWaitEvent(ws->stream, ws->ev_in[0], 0));
    }
    TRY(st.flush());  // packed small step: the one upload

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
    static const int force_overlap = getenv("RRTMGP_HIP_STEP_OVERLAP") ? atoi(getenv("RRTMGP_HIP_STEP_OVERLAP")) : -1;
    const double f_slots = (double)ncol / (4.0 * (double)ws->n_cu);
    const bool short_step = f_slots <= 0.65 || (f_slots > 1.05 && f_slots < 1.5) || (use_aero && f_slots > 2.1 && f_slots <= 6.0);
    const bool overlap = !(chunk && !st.packed) && (force_overlap >= 0 ? force_overlap != 0 : short_step);
    if (overlap) TRY(lane_resources(ws));
    if (prep) TRY(launch_prepare<FT>(ws, pv, *a->params, *po, false));
    const uint64_t seed = opts ? opts->seed : 0;
    const int64_t coff = opts ? opts->col_offset : 0;
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
    FT *lw_clear_net = fl_lw.clear_net, *sw_clear_net = fl_sw.clear_net;
    auto lw_lane = [&]() -> int {
        if (diag_lw && (!twostream_lw || band_lw)) {
            const DevFlux<FT> c = clear_first(fl_lw);
            TRY(launch_lw<FT>(ws, twostream_lw, *L.lw, nullptr, L.lw_aero, ds, emis, inc, inc_ld, c, n_angles, seed, coff, L.lw_max_int));
        }
        return launch_lw<FT>(ws, twostream_lw, *L.lw, L.lw_cld, L.lw_aero, ds, emis, inc, inc_ld, fl_lw, n_angles, seed, coff, L.lw_max_int);
    };
    auto sw_lane = [&](const Lane *lane) -> int {
        if (diag_sw && band_sw) {
            const DevFlux<FT> c = clear_first(fl_sw);
            DevState<FT> dc = ds_sw;
            dc.aod_sw_ext = dc.aod_sw_sca = nullptr;   // the all-sky solve writes the same values
            TRY(launch_sw<FT>(ws, 1, *L.sw, nullptr, L.sw_aero, dc, mu0, toa, adir, adif, c, seed, coff, L.sw_max_int, lane));
        }
        return launch_sw<FT>(ws, 1, *L.sw, L.sw_cld, L.sw_aero, ds_sw, mu0, toa, adir, adif, fl_sw, seed, coff, L.sw_max_int, lane);
    };
    auto net_sums = [&](hipStream_t stream) -> int {
        if (net) TRY(launch_net_sum<FT>(ws, fl_lw.net, fl_sw.net, net, ncol, nlev, fl_lw.layout, fl_lw.ld, fl_sw.ld, stream));
        if (clear_net) TRY(launch_net_sum<FT>(ws, lw_clear_net, sw_clear_net, clear_net, ncol, nlev, fl_lw.layout, fl_lw.ld, fl_sw.ld, stream));
        return RRTMGP_OK;
    };
    if (overlap) {
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
        if (order == 0) {          // (shipped) LW queued first, net sums on the main lane behind the join
            TRY(lw_lane());
            TRY(sw_lane(&second));
            RR_HIP(hipEventRecord(ws->ev_k[1], second.stream));
            RR_HIP(hipStreamWaitEvent(ws->stream, ws->ev_k[1], 0));
            TRY(net_sums(nullptr));
        } else {
            if (order == 1) { TRY(lw_lane()); TRY(sw_lane(&second)); }   // LW first, net sums on the second lane
            else { TRY(sw_lane(&second)); TRY(lw_lane()); }
            RR_HIP(hipEventRecord(ws->ev_k[2], ws->stream));   // LW done
            RR_HIP(hipStreamWaitEvent(second.stream, ws->ev_k[2], 0));
            TRY(net_sums(second.stream));
            RR_HIP(hipEventRecord(ws->ev_k[1], second.stream));   // join
            RR_HIP(hipStreamWaitEvent(ws->stream, ws->ev_k[1], 0));
        }
    } else {
        TRY(lw_lane());
        TRY(sw_lane(nullptr));
        TRY(net_sums(nullptr));
    }
    return chunk && !st.packed ? RRTMGP_OK : st.finish();
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

}  // namespace rrtmgp

using namespace rrtmgp;

// ======================================================================================
extern "C" {

int rrtmgp_hip_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return set_error(RRTMGP_ENODEV, "no HIP device visible");
    return n;
}

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

int rrtmgp_hip_workspace_create(int device, int64_t ncol, int64_t nlay, int32_t ftype, rrtmgp_workspace **out) {
    RR_CHECK(out, "null argument");
    RR_CHECK(ftype == RRTMGP_F32 || ftype == RRTMGP_F64, "ftype must be 4 or 8");
    RR_CHECK(ncol >= 1 && nlay >= 2 && ncol < (1LL << 31) && nlay < 4096, "bad workspace dimensions");
    TRY(select_device(device));
    auto *ws = new rrtmgp_workspace();
    ws->device = device; ws->ftype = ftype; ws->ncol = ncol; ws->nlay = nlay;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) != hipSuccess) { delete ws; return set_error(RRTMGP_EHIP, "hipGetDeviceProperties failed"); }
    ws->n_cu = prop.multiProcessorCount;
    if (hipStreamCreateWithFlags(&ws->own_stream, hipStreamNonBlocking) != hipSuccess ||
        hipEventCreate(&ws->ev_start) != hipSuccess || hipEventCreate(&ws->ev_stop) != hipSuccess) {
        delete ws;
        return set_error(RRTMGP_EHIP, "stream / event creation failed");
    }
    ws->stream = ws->own_stream;
    ws->timed = true;
    ws->stage.resize(S_NSLOTS);
    *out = ws;
    return RRTMGP_OK;
}

int rrtmgp_hip_workspace_destroy(rrtmgp_workspace *ws) {
    if (!ws) return RRTMGP_OK;
    (void)hipSetDevice(ws->device);
    (void)hipStreamSynchronize(ws->stream);
    host_pin_end(ws);
    if (!ws->shards.empty()) {  // a multi-device head owns its shards (and their worker threads) and nothing else
        shard_workers_destroy(ws->workers);
        for (rrtmgp_workspace *s : ws->shards) rrtmgp_hip_workspace_destroy(s);
        delete ws;
        return RRTMGP_OK;
    }
    for (auto &b : ws->stage) if (b.ptr) (void)rr_free(b.ptr);
    for (auto &b : ws->stage_alt) if (b.ptr) (void)rr_free(b.ptr);

    for (int i = 0; i < 2; i++) {
        if (ws->ev_in[i]) (void)hipEventDestroy(ws->ev_in[i]);
        if (ws->ev_k[i]) (void)hipEventDestroy(ws->ev_k[i]);
    }
    if (ws->ev_k[2]) (void)hipEventDestroy(ws->ev_k[2]);
    if (ws->copy_stream) (void)hipStreamDestroy(ws->copy_stream);
    if (ws->alt_stream) (void)hipStreamDestroy(ws->alt_stream);
    if (ws->scratch.ptr) (void)rr_free(ws->scratch.ptr);
    if (ws->alt_scratch.ptr) (void)rr_free(ws->alt_scratch.ptr);
    for (int *q : ws->col_queue) if (q) (void)rr_free(q);
    if (ws->bounce_h) (void)hipHostFree(ws->bounce_h);
    if (ws->bounce_d) (void)rr_free(ws->bounce_d);
    if (ws->ev_start) (void)hipEventDestroy(ws->ev_start);
    if (ws->ev_stop) (void)hipEventDestroy(ws->ev_stop);
    if (ws->own_stream) (void)hipStreamDestroy(ws->own_stream);
    delete ws;
    return RRTMGP_OK;
}

int rrtmgp_hip_workspace_set_stream(rrtmgp_workspace *ws, void *hip_stream) {
    RR_CHECK(ws, "null workspace");
    RR_CHECK(ws->shards.empty(), "a multi-device workspace runs each shard on its own stream");
    if (ws->stream != (hipStream_t)hip_stream) {
        // nothing of this workspace may still be running on the stream it leaves: its kernels share the queue counters and
        // the sweep scratch with whatever the new stream launches next
        RR_HIP(hipSetDevice(ws->device));
        RR_HIP(hipStreamSynchronize(ws->stream));
        if (ws->alt_stream) RR_HIP(hipStreamSynchronize(ws->alt_stream));
        TRY(queue_check(ws, true));
    }
    ws->stream = (hipStream_t)hip_stream;  // NULL is the HIP null (legacy default) stream
    return RRTMGP_OK;
}

int rrtmgp_hip_workspace_synchronize(rrtmgp_workspace *ws) {
    RR_CHECK(ws, "null workspace");
    for (rrtmgp_workspace *s : ws->shards) TRY(rrtmgp_hip_workspace_synchronize(s));
    if (!ws->shards.empty()) return RRTMGP_OK;
    RR_HIP(hipSetDevice(ws->device));
    RR_HIP(hipStreamSynchronize(ws->stream));
    return RRTMGP_OK;
}

int rrtmgp_hip_workspace_last_kernel_ms(rrtmgp_workspace *ws, double *ms) {
    RR_CHECK(ws && ms, "null argument");
    if (!ws->shards.empty()) {  // the slowest shard
        *ms = 0;
        for (rrtmgp_workspace *s : ws->shards) {
            double m = 0;
            TRY(rrtmgp_hip_workspace_last_kernel_ms(s, &m));
            *ms = std::max(*ms, m);
        }
        return RRTMGP_OK;
    }
    RR_HIP(hipSetDevice(ws->device));
    RR_HIP(hipEventSynchronize(ws->ev_stop));
    float f = 0;
    RR_HIP(hipEventElapsedTime(&f, ws->ev_start, ws->ev_stop));
    *ms = f;
    // a timed workspace is a measured / tested one: the cheap place to verify the queue invariant (the timed launch is over;
    // another lane may still run, so only when the whole workspace is idle)
    static const bool check = getenv("RRTMGP_HIP_NO_QUEUE_CHECK") == nullptr;
    if (check && hipStreamQuery(ws->stream) == hipSuccess && (!ws->alt_stream || hipStreamQuery(ws->alt_stream) == hipSuccess))
        TRY(queue_check(ws, false));
    return RRTMGP_OK;
}

#define GAS_DISPATCH(ws, fn, lk, cld, aero, ...)                                                                      \
    ((ws)->ftype == RRTMGP_F32                                                                                        \
         ? fn<float>(ws, twostream, (lk)->gas32, (cld) ? &(cld)->cld32 : nullptr, (aero) ? &(aero)->aero32 : nullptr, \
                     (lk)->max_int, __VA_ARGS__)                                                                    \
         : fn<double>(ws, twostream, (lk)->gas64, (cld) ? &(cld)->cld64 : nullptr, (aero) ? &(aero)->aero64 : nullptr, \
                      (lk)->max_int, __VA_ARGS__))

}  // extern "C" (templates need C++ linkage)

// ---- multi-device dispatch of the spectral solvers: slice, pick the replicas, re-enter the single-device entry ----
static int n_bnd_of(const rrtmgp_workspace *ws, const rrtmgp_lookup *gas) {
    return ws->ftype == RRTMGP_F32 ? gas->gas32.n_bnd : gas->gas64.n_bnd;
}
template <typename BCS, typename SliceBcs, typename Call>
static int multi_spectral(rrtmgp_workspace *ws, const rrtmgp_lookup *gas, const rrtmgp_lookup *cld, const rrtmgp_lookup *aero,
                          const rrtmgp_atmos_state *as, const BCS *bcs, const rrtmgp_flux_out *flux,
                          const rrtmgp_solve_opts *opts, SliceBcs slice_bcs, Call call) {
    const size_t E = (size_t)ws->ftype, nlev = (size_t)as->nlay + 1;
    return multi_run(ws, [&](rrtmgp_workspace *sw, size_t c0, size_t nc) -> int {
        const ColumnSlice sl{E, c0};
        rrtmgp_atmos_state a = *as;
        BCS b = *bcs;
        rrtmgp_flux_out f = *flux;
        rrtmgp_solve_opts o{};
        if (opts) o = *opts; else o.n_gauss_angles = 1;
        slice_state(a, sl, nc);
        slice_flux(f, o, sl, nlev, (size_t)as->ncol);
        slice_bcs(b, sl);
        const rrtmgp_lookup *g = lookup_on(gas, sw->device), *c = lookup_on(cld, sw->device), *ae = lookup_on(aero, sw->device);
        if (!g || (cld && !c) || (aero && !ae))
            return set_error(RRTMGP_EINVAL, "a lookup has no replica on one of the workspace's devices (use *_lookup_create_multi)");
        return call(sw, g, c, ae, &a, &b, &f, &o);
    }, as->mem == RRTMGP_MEM_DEVICE || bcs->mem == RRTMGP_MEM_DEVICE || flux->mem == RRTMGP_MEM_DEVICE ||
       (opts && opts->metric_scaling && opts->metric_mem == RRTMGP_MEM_DEVICE));
}
static int check_multi_spectral(rrtmgp_workspace *ws, const rrtmgp_lookup *gas, const rrtmgp_atmos_state *as, const void *bcs,
                                int bcs_mem, const rrtmgp_flux_out *flux, const rrtmgp_solve_opts *opts, const void *inc_flux) {
    RR_CHECK(ws && gas && as && bcs && flux, "null argument");
    RR_CHECK(gas->kind == LK_GAS && gas->ftype == ws->ftype, "expected a gas lookup of the workspace's precision");
    RR_CHECK(as->ncol == ws->ncol && as->nlay == ws->nlay, "state dimensions differ from the workspace");
    return check_multi(ws, as->mem, bcs_mem, flux, opts, inc_flux);
}
static const rrtmgp_lookup *head_replica(const rrtmgp_workspace *ws, const rrtmgp_lookup *lk) {
    return lk ? lookup_on(lk, ws->shards[0]->device) : nullptr;
}

extern "C" {

int rrtmgp_hip_rte_lw_2stream_solve(rrtmgp_workspace *ws, const rrtmgp_lookup *lookup_lw, const rrtmgp_lookup *cld,
                                    const rrtmgp_lookup *aero, const rrtmgp_atmos_state *as, const rrtmgp_lw_bcs *bcs,
                                    const rrtmgp_flux_out *flux, const rrtmgp_solve_opts *opts) {
    const int twostream = 1;
    if (ws && !ws->shards.empty()) {
        TRY(check_multi_spectral(ws, lookup_lw, as, bcs, bcs ? bcs->mem : 0, flux, opts, bcs ? bcs->inc_flux : nullptr));
        const rrtmgp_lookup *g = head_replica(ws, lookup_lw), *c = head_replica(ws, cld), *ae = head_replica(ws, aero);
        RR_CHECK(g && (!cld || c) && (!aero || ae), "a lookup has no replica on the workspace's first device");
        PinScope scope{ws};
        TRY(GAS_DISPATCH(ws, pin_lw, g, c, ae, as, bcs, flux, opts));
        const size_t nb = (size_t)n_bnd_of(ws, lookup_lw);
        return multi_spectral(ws, lookup_lw, cld, aero, as, bcs, flux, opts,
                              [nb, as](rrtmgp_lw_bcs &b, const ColumnSlice &sl) { slice_lw_bcs(b, sl, nb, (size_t)as->ncol); },
                              rrtmgp_hip_rte_lw_2stream_solve);
    }
    TRY(check_common(ws, lookup_lw, 0, cld, aero, as));
    return GAS_DISPATCH(ws, solve_lw_host, lookup_lw, cld, aero, as, bcs, flux, opts);
}

int rrtmgp_hip_rte_lw_noscat_solve(rrtmgp_workspace *ws, const rrtmgp_lookup *lookup_lw, const rrtmgp_lookup *cld,
                                   const rrtmgp_lookup *aero, const rrtmgp_atmos_state *as, const rrtmgp_lw_bcs *bcs,
                                   const rrtmgp_flux_out *flux, const rrtmgp_solve_opts *opts) {
    const int twostream = 0;
    if (ws && !ws->shards.empty()) {
        TRY(check_multi_spectral(ws, lookup_lw, as, bcs, bcs ? bcs->mem : 0, flux, opts, bcs ? bcs->inc_flux : nullptr));
        const rrtmgp_lookup *g = head_replica(ws, lookup_lw), *c = head_replica(ws, cld), *ae = head_replica(ws, aero);
        RR_CHECK(g && (!cld || c) && (!aero || ae), "a lookup has no replica on the workspace's first device");
        PinScope scope{ws};
        TRY(GAS_DISPATCH(ws, pin_lw, g, c, ae, as, bcs, flux, opts));
        const size_t nb = (size_t)n_bnd_of(ws, lookup_lw);
        return multi_spectral(ws, lookup_lw, cld, aero, as, bcs, flux, opts,
                              [nb, as](rrtmgp_lw_bcs &b, const ColumnSlice &sl) { slice_lw_bcs(b, sl, nb, (size_t)as->ncol); },
                              rrtmgp_hip_rte_lw_noscat_solve);
    }
    TRY(check_common(ws, lookup_lw, 0, cld, aero, as));
    return GAS_DISPATCH(ws, solve_lw_host, lookup_lw, cld, aero, as, bcs, flux, opts);
}

int rrtmgp_hip_rte_sw_2stream_solve(rrtmgp_workspace *ws, const rrtmgp_lookup *lookup_sw, const rrtmgp_lookup *cld,
                                    const rrtmgp_lookup *aero, const rrtmgp_atmos_state *as, const rrtmgp_sw_bcs *bcs,
                                    const rrtmgp_flux_out *flux, const rrtmgp_solve_opts *opts) {
    const int twostream = 1;
    if (ws && !ws->shards.empty()) {
        TRY(check_multi_spectral(ws, lookup_sw, as, bcs, bcs ? bcs->mem : 0, flux, opts, nullptr));
        const rrtmgp_lookup *g = head_replica(ws, lookup_sw), *c = head_replica(ws, cld), *ae = head_replica(ws, aero);
        RR_CHECK(g && (!cld || c) && (!aero || ae), "a lookup has no replica on the workspace's first device");
        PinScope scope{ws};
        TRY(GAS_DISPATCH(ws, pin_sw, g, c, ae, as, bcs, flux, opts));
        const size_t nb = (size_t)n_bnd_of(ws, lookup_sw);
        return multi_spectral(ws, lookup_sw, cld, aero, as, bcs, flux, opts,
                              [nb](rrtmgp_sw_bcs &b, const ColumnSlice &sl) { slice_sw_bcs(b, sl, nb); },
                              rrtmgp_hip_rte_sw_2stream_solve);
    }
    TRY(check_common(ws, lookup_sw, 1, cld, aero, as));
    return GAS_DISPATCH(ws, solve_sw_host, lookup_sw, cld, aero, as, bcs, flux, opts);
}

int rrtmgp_hip_rte_sw_noscat_solve(rrtmgp_workspace *ws, const rrtmgp_lookup *lookup_sw, const rrtmgp_atmos_state *as,
                                   const rrtmgp_sw_bcs *bcs, const rrtmgp_flux_out *flux, const rrtmgp_solve_opts *opts) {
    const int twostream = 0;
    const rrtmgp_lookup *cld = nullptr, *aero = nullptr;
    if (ws && !ws->shards.empty()) {
        TRY(check_multi_spectral(ws, lookup_sw, as, bcs, bcs ? bcs->mem : 0, flux, opts, nullptr));
        const rrtmgp_lookup *g = head_replica(ws, lookup_sw);
        RR_CHECK(g, "the lookup has no replica on the workspace's first device");
        PinScope scope{ws};
        TRY(GAS_DISPATCH(ws, pin_sw, g, cld, aero, as, bcs, flux, opts));
        const size_t nb = (size_t)n_bnd_of(ws, lookup_sw);
        return multi_spectral(ws, lookup_sw, cld, aero, as, bcs, flux, opts,
                              [nb](rrtmgp_sw_bcs &b, const ColumnSlice &sl) { slice_sw_bcs(b, sl, nb); },
                              [](rrtmgp_workspace *sw, const rrtmgp_lookup *g2, const rrtmgp_lookup *, const rrtmgp_lookup *,
                                 const rrtmgp_atmos_state *a, const rrtmgp_sw_bcs *b, const rrtmgp_flux_out *f,
                                 const rrtmgp_solve_opts *o) { return rrtmgp_hip_rte_sw_noscat_solve(sw, g2, a, b, f, o); });
    }
    TRY(check_common(ws, lookup_sw, 1, nullptr, nullptr, as));
    return GAS_DISPATCH(ws, solve_sw_host, lookup_sw, cld, aero, as, bcs, flux, opts);
}

}  // extern "C"

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

int rrtmgp_hip_workspace_transfer_bytes(const rrtmgp_workspace *ws, uint64_t *h2d, uint64_t *d2h) {
    RR_CHECK(ws, "null workspace");
    uint64_t a = ws->h2d_bytes, b = ws->d2h_bytes;
    for (const rrtmgp_workspace *s : ws->shards) { a += s->h2d_bytes; b += s->d2h_bytes; }
    if (h2d) *h2d = a;
    if (d2h) *d2h = b;
    return RRTMGP_OK;
}

}  // extern "C"

// gray solves / preparation steps on a multi-device workspace: same slicing, no lookups
template <typename BCS, typename SliceBcs, typename Call>
static int multi_gray(rrtmgp_workspace *ws, const rrtmgp_gray_state *gs, const BCS *bcs, const rrtmgp_flux_out *flux,
                      const rrtmgp_solve_opts *opts, SliceBcs slice_bcs, Call call) {
    RR_CHECK(gs && bcs && flux, "null argument");
    RR_CHECK(gs->ncol == ws->ncol && gs->nlay == ws->nlay, "state dimensions differ from the workspace");
    TRY(check_multi(ws, gs->mem, bcs->mem, flux, opts, nullptr));
    const size_t E = (size_t)ws->ftype, nlev = (size_t)gs->nlay + 1;
    return multi_run(ws, [&](rrtmgp_workspace *sw, size_t c0, size_t nc) -> int {
        const ColumnSlice sl{E, c0};
        rrtmgp_gray_state g = *gs;
        BCS b = *bcs;
        rrtmgp_flux_out f = *flux;
        rrtmgp_solve_opts o{};
        if (opts) o = *opts; else o.n_gauss_angles = 1;
        slice_gray(g, sl, nc);
        slice_flux(f, o, sl, nlev, (size_t)gs->ncol);
        slice_bcs(b, sl);
        return call(sw, &g, &b, &f, &o);
    }, gs->mem == RRTMGP_MEM_DEVICE || bcs->mem == RRTMGP_MEM_DEVICE || flux->mem == RRTMGP_MEM_DEVICE ||
       (opts && opts->metric_scaling && opts->metric_mem == RRTMGP_MEM_DEVICE));
}
static void slice_gray_lw_bcs(rrtmgp_lw_bcs &b, const ColumnSlice &sl) { b.sfc_emis = sl.adv(b.sfc_emis, 1); b.inc_flux = sl.adv(b.inc_flux, 1); }
static void slice_gray_sw_bcs(rrtmgp_sw_bcs &b, const ColumnSlice &sl) { slice_sw_bcs(b, sl, 1); }

extern "C" {

static int check_gray(rrtmgp_workspace *ws, const rrtmgp_gray_state *gs, const void *bcs, const rrtmgp_flux_out *flux) {
    RR_CHECK(ws && gs && bcs && flux, "null argument");
    RR_CHECK(gs->ncol == ws->ncol && gs->nlay == ws->nlay, "state dimensions differ from the workspace");
    RR_CHECK(gs->otp_kind == 0 || gs->otp_kind == 1, "unknown gray optical-thickness kind");
    RR_CHECK(gs->p_lay && gs->p_lev, "gray state: missing pressure arrays");
    RR_HIP(hipSetDevice(ws->device));
    return RRTMGP_OK;
}

int rrtmgp_hip_rte_lw_2stream_solve_gray(rrtmgp_workspace *ws, const rrtmgp_gray_state *as, const rrtmgp_lw_bcs *bcs,
                                         const rrtmgp_flux_out *flux, const rrtmgp_solve_opts *opts) {
    if (ws && !ws->shards.empty()) return multi_gray(ws, as, bcs, flux, opts, slice_gray_lw_bcs, rrtmgp_hip_rte_lw_2stream_solve_gray);
    TRY(check_gray(ws, as, bcs, flux));
    RR_CHECK(as->lat && as->t_lay && as->t_lev && as->t_sfc && bcs->sfc_emis, "gray LW: missing array");
    return ws->ftype == RRTMGP_F32 ? solve_gray_lw_t<float>(ws, 1, as, bcs, flux, opts)
                                   : solve_gray_lw_t<double>(ws, 1, as, bcs, flux, opts);
}
int rrtmgp_hip_rte_lw_noscat_solve_gray(rrtmgp_workspace *ws, const rrtmgp_gray_state *as, const rrtmgp_lw_bcs *bcs,
                                        const rrtmgp_flux_out *flux, const rrtmgp_solve_opts *opts) {
    if (ws && !ws->shards.empty()) return multi_gray(ws, as, bcs, flux, opts, slice_gray_lw_bcs, rrtmgp_hip_rte_lw_noscat_solve_gray);
    TRY(check_gray(ws, as, bcs, flux));
    RR_CHECK(as->lat && as->t_lay && as->t_lev && as->t_sfc && bcs->sfc_emis, "gray LW: missing array");
    RR_CHECK(!opts || opts->n_gauss_angles <= 1, "gray radiation is solved with a single quadrature angle");
    return ws->ftype == RRTMGP_F32 ? solve_gray_lw_t<float>(ws, 0, as, bcs, flux, opts)
                                   : solve_gray_lw_t<double>(ws, 0, as, bcs, flux, opts);
}
int rrtmgp_hip_rte_sw_2stream_solve_gray(rrtmgp_workspace *ws, const rrtmgp_gray_state *as, const rrtmgp_sw_bcs *bcs,
                                         const rrtmgp_flux_out *flux, const rrtmgp_solve_opts *opts) {
    if (ws && !ws->shards.empty()) return multi_gray(ws, as, bcs, flux, opts, slice_gray_sw_bcs, rrtmgp_hip_rte_sw_2stream_solve_gray);
    TRY(check_gray(ws, as, bcs, flux));
    RR_CHECK(bcs->cos_zenith && bcs->toa_flux && bcs->sfc_alb_direct && bcs->sfc_alb_diffuse && flux->flux_dn_dir,
             "gray SW: missing array");
    return ws->ftype == RRTMGP_F32 ? solve_gray_sw_t<float>(ws, 1, as, bcs, flux, opts)
                                   : solve_gray_sw_t<double>(ws, 1, as, bcs, flux, opts);
}
int rrtmgp_hip_rte_sw_noscat_solve_gray(rrtmgp_workspace *ws, const rrtmgp_gray_state *as, const rrtmgp_sw_bcs *bcs,
                                        const rrtmgp_flux_out *flux, const rrtmgp_solve_opts *opts) {
    if (ws && !ws->shards.empty()) return multi_gray(ws, as, bcs, flux, opts, slice_gray_sw_bcs, rrtmgp_hip_rte_sw_noscat_solve_gray);
    TRY(check_gray(ws, as, bcs, flux));
    RR_CHECK(bcs->cos_zenith && bcs->toa_flux && flux->flux_dn_dir, "gray SW: missing array");
    return ws->ftype == RRTMGP_F32 ? solve_gray_sw_t<float>(ws, 0, as, bcs, flux, opts)
                                   : solve_gray_sw_t<double>(ws, 0, as, bcs, flux, opts);
}

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

int rrtmgp_hip_prepare_atmosphere(rrtmgp_workspace *ws, const rrtmgp_atmos_state *as, const rrtmgp_params *params,
                                  const rrtmgp_prepare_opts *opts) {
    RR_CHECK(ws && as && params && opts, "null argument");
    RR_CHECK(as->ncol >= 0 && as->ncol <= ws->ncol && as->nlay == ws->nlay, "state does not fit the workspace");
    if (!ws->shards.empty()) {
        RR_CHECK(as->ncol == ws->ncol, "state dimensions differ from the workspace");
        const size_t nlay = (size_t)as->nlay;
        return multi_run(ws, [&](rrtmgp_workspace *sw, size_t c0, size_t nc) -> int {
            const ColumnSlice sl{(size_t)ws->ftype, c0};
            rrtmgp_atmos_state a = *as;
            rrtmgp_prepare_opts o = *opts;
            slice_state(a, sl, nc);
            o.center_z = sl.adv(o.center_z, nlay); o.face_z = sl.adv(o.face_z, nlay + 1);
            return rrtmgp_hip_prepare_atmosphere(sw, &a, params, &o);
        }, as->mem == RRTMGP_MEM_DEVICE);
    }
    RR_HIP(hipSetDevice(ws->device));
    return ws->ftype == RRTMGP_F32 ? prepare_t<float>(ws, as, params, opts) : prepare_t<double>(ws, as, params, opts);
}

int rrtmgp_hip_prepare_atmosphere_gray(rrtmgp_workspace *ws, const rrtmgp_gray_state *as, const rrtmgp_params *params,
                                       const rrtmgp_prepare_opts *opts) {
    RR_CHECK(ws && as && params && opts, "null argument");
    RR_CHECK(as->ncol >= 0 && as->ncol <= ws->ncol && as->nlay == ws->nlay, "state does not fit the workspace");
    if (!ws->shards.empty()) {
        RR_CHECK(as->ncol == ws->ncol, "state dimensions differ from the workspace");
        const size_t nlay = (size_t)as->nlay;
        return multi_run(ws, [&](rrtmgp_workspace *sw, size_t c0, size_t nc) -> int {
            const ColumnSlice sl{(size_t)ws->ftype, c0};
            rrtmgp_gray_state g = *as;
            rrtmgp_prepare_opts o = *opts;
            slice_gray(g, sl, nc);
            o.center_z = sl.adv(o.center_z, nlay); o.face_z = sl.adv(o.face_z, nlay + 1);
            return rrtmgp_hip_prepare_atmosphere_gray(sw, &g, params, &o);
        }, as->mem == RRTMGP_MEM_DEVICE);
    }
    RR_HIP(hipSetDevice(ws->device));
    return ws->ftype == RRTMGP_F32 ? prepare_gray_t<float>(ws, as, params, opts)
                                   : prepare_gray_t<double>(ws, as, params, opts);
}

int rrtmgp_hip_host_register(void *ptr, size_t bytes) { return host_register_explicit(ptr, bytes); }
int rrtmgp_hip_host_unregister(void *ptr) { return host_unregister_explicit(ptr); }
int rrtmgp_hip_host_registered_count(void) { return host_registered_count(); }

// ---- caller-owned device arrays (a host language without its own GPU array package: ext/RRTMGPHIPExt.jl HIPArray) --------
// Plain hipMalloc / hipFree / hipMemcpy / hipMemset on the named device: these are the CALLER's arrays, so they do not enter
// the library's allocation accounting (rrtmgp_hip_allocation_counts is about what a solve allocates).
int rrtmgp_hip_device_malloc(int device, size_t bytes, void **out) {
    RR_CHECK(out, "null output pointer");
    RR_HIP(hipSetDevice(device));
    RR_HIP(hipMalloc(out, bytes ? bytes : 16));
    return RRTMGP_OK;
}
int rrtmgp_hip_device_free(int device, void *ptr) {
    if (!ptr) return RRTMGP_OK;
    RR_HIP(hipSetDevice(device));
    RR_HIP(hipFree(ptr));
    return RRTMGP_OK;
}
int rrtmgp_hip_memcpy(int device, void *dst, const void *src, size_t bytes, int32_t kind) {
    RR_CHECK(kind >= RRTMGP_COPY_H2D && kind <= RRTMGP_COPY_D2D, "kind must be 1 (host to device), 2 (device to host) or 3 (device to device)");
    if (bytes == 0) return RRTMGP_OK;
    RR_CHECK(dst && src, "null pointer");
    RR_HIP(hipSetDevice(device));
    // blocking, and ordered behind everything queued on the device (the solves run on the workspaces' own streams)
    RR_HIP(hipDeviceSynchronize());
    RR_HIP(hipMemcpy(dst, src, bytes, kind == RRTMGP_COPY_H2D ? hipMemcpyHostToDevice
                                      : kind == RRTMGP_COPY_D2H ? hipMemcpyDeviceToHost : hipMemcpyDeviceToDevice));
    return RRTMGP_OK;
}
int rrtmgp_hip_memset(int device, void *dst, int32_t byte_value, size_t bytes) {
    if (bytes == 0) return RRTMGP_OK;
    RR_CHECK(dst, "null pointer");
    RR_HIP(hipSetDevice(device));
    RR_HIP(hipDeviceSynchronize());
    RR_HIP(hipMemset(dst, byte_value, bytes));
    RR_HIP(hipDeviceSynchronize());
    return RRTMGP_OK;
}

int rrtmgp_hip_allocation_counts(int64_t *device_allocs, int64_t *device_frees, int64_t *host_registrations) {
    if (device_allocs) *device_allocs = g_dev_allocs.load();
    if (device_frees) *device_frees = g_dev_frees.load();
    if (host_registrations) *host_registrations = g_host_regs.load();
    return RRTMGP_OK;
}

double rrtmgp_hip_mcica_uniform(uint64_t seed, int64_t gcol, int64_t igpt, int32_t is_sw, int32_t draw) {
    return mcica_draw(mcica_key(seed, gcol, igpt, is_sw), draw);
}

int rrtmgp_hip_last_error(char *buf, size_t n) {
    if (!buf || n == 0) return RRTMGP_EINVAL;
    snprintf(buf, n, "%s", g_last_error.c_str());
    return RRTMGP_OK;
}

const char *rrtmgp_hip_build_flags(void) {
    static const std::string s = [] { std::string f = RR_BUILD_FLAGS; return f.empty() ? f : f.substr(1); }();
    return s.c_str();
}
const char *rrtmgp_hip_version(void) {
    static const std::string s = std::string("0.5.0") + (*rrtmgp_hip_build_flags() ? std::string(" [") + rrtmgp_hip_build_flags() + "]" : std::string());
    return s.c_str();
}

/* sizes of the ABI structs as compiled, for binding self-checks (tests/test_abi.py) */
int rrtmgp_hip_abi_sizeof(int which) {
    switch (which) {
        case 0: return (int)sizeof(rrtmgp_minor_desc);
        case 1: return (int)sizeof(rrtmgp_gas_lookup_desc);
        case 2: return (int)sizeof(rrtmgp_cloud_lookup_desc);
        case 3: return (int)sizeof(rrtmgp_aerosol_lookup_desc);
        case 4: return (int)sizeof(rrtmgp_atmos_state);
        case 5: return (int)sizeof(rrtmgp_lw_bcs);
        case 6: return (int)sizeof(rrtmgp_sw_bcs);
        case 7: return (int)sizeof(rrtmgp_flux_out);
        case 8: return (int)sizeof(rrtmgp_solve_opts);
        case 9: return (int)sizeof(rrtmgp_gray_state);
        case 10: return (int)sizeof(rrtmgp_params);
        case 11: return (int)sizeof(rrtmgp_prepare_opts);
        case 12: return (int)sizeof(rrtmgp_view2d);
        case 13: return (int)sizeof(rrtmgp_update_fluxes_args);
        default: return -1;
    }
}

}  // extern "C"
