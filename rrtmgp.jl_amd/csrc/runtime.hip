// runtime.hip — errors, allocation accounting, page-locked host arrays, workspace buffers and streams, and the entry points
// that only touch those: device count, workspaces, caller-owned device arrays, diagnostics (include/rrtmgp_hip.h).
#include "host.h"
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

int bounce_ensure(rrtmgp_workspace *ws, size_t bytes) {
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

int select_device(int device) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return set_error(RRTMGP_ENODEV, "no HIP device visible");
    if (device < 0 || device >= n) return set_error(RRTMGP_EINVAL, "device index out of range");
    RR_HIP(hipSetDevice(device));
    return RRTMGP_OK;
}

}  // namespace rrtmgp

using namespace rrtmgp;

extern "C" {

int rrtmgp_hip_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return set_error(RRTMGP_ENODEV, "no HIP device visible");
    return n;
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

int rrtmgp_hip_workspace_transfer_bytes(const rrtmgp_workspace *ws, uint64_t *h2d, uint64_t *d2h) {
    RR_CHECK(ws, "null workspace");
    uint64_t a = ws->h2d_bytes, b = ws->d2h_bytes;
    for (const rrtmgp_workspace *s : ws->shards) { a += s->h2d_bytes; b += s->d2h_bytes; }
    if (h2d) *h2d = a;
    if (d2h) *d2h = b;
    return RRTMGP_OK;
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
        case 14: return (int)sizeof(rrtmgp_update_fluxes_gray_args);
        default: return -1;
    }
}

}  // extern "C"
