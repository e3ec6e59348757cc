// multi.hip — several GPUs from one host process: replicated lookups, sharded workspaces,
// one host thread + one stream per shard (include/rrtmgp_hip.h, "several GPUs from ONE host process").
//
// Columns shard embarrassingly: shard s owns the contiguous global columns
// [s*ncol/n, (s+1)*ncol/n), which is one contiguous slab of every state / boundary / flux array because ncol is
// their slowest dimension.  There is no collective: every shard stages its own slab in, solves it with
// `col_offset` advanced to its first global column (the McICA stream is keyed by the global column), and
// stages its fluxes back into the caller's arrays.  The public entry points (solve.hip, step.hip, views.hip) do the slicing and
// re-enter themselves with a shard workspace; this file owns the objects and the fan-out.
#include <pthread.h>
#include <sched.h>

#include <cctype>
#include <condition_variable>
#include <cstring>
#include <memory>
#include <mutex>
#include <thread>

#include "common.h"

namespace rrtmgp {

const rrtmgp_lookup *lookup_on(const rrtmgp_lookup *lk, int device) {
    if (!lk) return nullptr;
    if (lk->device == device) return lk;
    for (const rrtmgp_lookup *r : lk->replicas)
        if (r->device == device) return r;
    return nullptr;
}

// ---- which CPUs sit next to a GPU -------------------------------------------------------------------------------------
// "<sysfs>/bus/pci/devices/<bus id>/local_cpulist" holds ranges like "0-31,64-95".  Host logic only (no HIP call): exported as
// rrtmgp_hip_local_cpus so that it can be tested against a fake sysfs tree (RRTMGP_HIP_SYSFS_ROOT, default /sys) on a box
// whose single GPU never exercises the multi-socket case.
int parse_cpulist(const char *text, std::vector<int> &cpus) {
    cpus.clear();
    std::string line(text ? text : "");
    char *save = nullptr;
    for (char *tok = strtok_r(&line[0], ",\n \t", &save); tok; tok = strtok_r(nullptr, ",\n \t", &save)) {
        int lo = 0, hi = 0;
        const int k = sscanf(tok, "%d-%d", &lo, &hi);
        if (k < 1) continue;
        if (k == 1) hi = lo;
        if (lo < 0 || hi < lo) continue;
        for (int c = lo; c <= hi && c < CPU_SETSIZE; c++) cpus.push_back(c);
    }
    return (int)cpus.size();
}
int local_cpus_of(const char *pci_bus_id, std::vector<int> &cpus) {
    cpus.clear();
    if (!pci_bus_id || !*pci_bus_id) return -1;
    std::string id(pci_bus_id);
    for (char &c : id) c = (char)tolower(c);   // hipDeviceGetPCIBusId prints upper-case hex, sysfs names are lower-case
    const char *root = getenv("RRTMGP_HIP_SYSFS_ROOT");
    const std::string path = std::string(root && *root ? root : "/sys") + "/bus/pci/devices/" + id + "/local_cpulist";
    FILE *f = fopen(path.c_str(), "r");
    if (!f) return -1;
    char line[4096] = {0};
    const bool ok = fgets(line, sizeof line, f) != nullptr;
    fclose(f);
    return ok ? parse_cpulist(line, cpus) : 0;
}

// ---- shard workers ---------------------------------------------------------------------------------------------------
// One persistent host thread per shard (round 4: shard 0 too — on the calling thread it ran unbound, wherever the host
// model's thread happened to be), parked on a condition variable between calls: a call costs two notifications per shard instead of a thread spawn + join (multi_run used to
// create and join std::threads on every entry).  A worker is bound to the CPUs that are local to its GPU's PCIe root
// (/sys/bus/pci/devices/<bus id>/local_cpulist): at 8 GPUs the staging copies of one process move > 100 GB/s, which only
// works from the memory controllers next to each GPU.  RRTMGP_HIP_NO_NUMA_BIND=1 leaves the threads unbound.
struct ShardWorkers {
    struct Worker {
        std::thread th;
        std::mutex mu;
        std::condition_variable cv;
        const std::function<void()> *job = nullptr;  // set by the caller, cleared by the worker when done
        bool quit = false;
    };
    std::vector<std::unique_ptr<Worker>> w;

    static void bind_near(int device) {
        if (getenv("RRTMGP_HIP_NO_NUMA_BIND")) return;
        char bus[64] = {0};
        if (hipDeviceGetPCIBusId(bus, sizeof bus, device) != hipSuccess) { (void)hipGetLastError(); return; }
        std::vector<int> cpus;
        if (local_cpus_of(bus, cpus) <= 0) return;
        cpu_set_t set;
        CPU_ZERO(&set);
        for (int c : cpus) CPU_SET(c, &set);
        (void)pthread_setaffinity_np(pthread_self(), sizeof set, &set);  // failure (cgroup limits) is harmless
    }
    explicit ShardWorkers(const std::vector<rrtmgp_workspace *> &shards) {
        for (size_t s = 0; s < shards.size(); s++) {
            w.emplace_back(new Worker());
            Worker *me = w.back().get();
            const int device = shards[s]->device;
            me->th = std::thread([me, device] {
                bind_near(device);
                std::unique_lock<std::mutex> lk(me->mu);
                for (;;) {
                    me->cv.wait(lk, [me] { return me->job || me->quit; });
                    if (me->quit) return;
                    const std::function<void()> *j = me->job;
                    lk.unlock();
                    (*j)();
                    lk.lock();
                    me->job = nullptr;
                    me->cv.notify_all();
                }
            });
        }
    }
    void start(size_t i, const std::function<void()> *job) {
        std::lock_guard<std::mutex> lk(w[i]->mu);
        w[i]->job = job;
        w[i]->cv.notify_all();
    }
    void wait(size_t i) {
        std::unique_lock<std::mutex> lk(w[i]->mu);
        w[i]->cv.wait(lk, [&] { return w[i]->job == nullptr; });
    }
    ~ShardWorkers() {
        for (auto &x : w) {
            { std::lock_guard<std::mutex> lk(x->mu); x->quit = true; }
            x->cv.notify_all();
            x->th.join();
        }
    }
};
void shard_workers_destroy(ShardWorkers *p) { delete p; }

int multi_run(rrtmgp_workspace *ws, const std::function<int(rrtmgp_workspace *, size_t, size_t)> &shard_call, bool device_arrays) {
    const size_t n = ws->shards.size();
    if (device_arrays) {
        // the shards run on private streams: whatever the caller queued on this device (the producer of the arrays) first
        RR_HIP(hipSetDevice(ws->shards[0]->device));
        RR_HIP(hipDeviceSynchronize());
    }
    std::vector<int> rc(n, RRTMGP_OK);
    std::vector<std::string> msg(n);
    auto run = [&](size_t s) {
        const size_t c0 = (size_t)ws->shard_c0[s], c1 = (size_t)ws->shard_c0[s + 1];
        if (c1 == c0) return;
        rc[s] = shard_call(ws->shards[s], c0, c1 - c0);
        if (rc[s] != RRTMGP_OK) { msg[s] = last_error_string(); return; }  // the message lives in the worker's thread-local slot
        // host arrays are home already (the shard's entry point blocks); device arrays: the kernels have finished
        if (device_arrays && hipStreamSynchronize(ws->shards[s]->stream) != hipSuccess) { rc[s] = RRTMGP_EHIP; msg[s] = "hipStreamSynchronize"; }
    };
    static const bool spawn = getenv("RRTMGP_HIP_SPAWN_SHARD_THREADS") != nullptr;  // A/B: the old thread-per-call fan-out
    if (spawn) {
        std::vector<std::thread> workers;
        for (size_t s = 1; s < n; s++) workers.emplace_back(run, s);
        run(0);
        for (auto &t : workers) t.join();
    } else if (n == 1) {
        run(0);  // HIPDevice(0) / Workspace(device=[0]): one shard, nothing to fan out — no pool, the caller's thread
    } else {
        if (!ws->workers) ws->workers = new ShardWorkers(ws->shards);
        std::vector<std::function<void()>> jobs(n);
        for (size_t s = 0; s < n; s++) {
            jobs[s] = [&run, s] { run(s); };
            ws->workers->start(s, &jobs[s]);
        }
        for (size_t s = 0; s < n; s++) ws->workers->wait(s);
    }
    for (size_t s = 0; s < n; s++)
        if (rc[s] != RRTMGP_OK) return set_error(rc[s], "shard " + std::to_string(s) + ": " + msg[s]);
    return RRTMGP_OK;
}

// one lookup per distinct device: the first is the head, the others hang off it
template <typename Desc, typename Create>
static int create_lookup_multi(const Desc *desc, const int32_t *device_ids, int ndev, rrtmgp_lookup **out, Create create) {
    RR_CHECK(desc && device_ids && out && ndev >= 1 && ndev <= 64, "bad argument");
    rrtmgp_lookup *head = nullptr;
    for (int i = 0; i < ndev; i++) {
        if (lookup_on(head, device_ids[i])) continue;
        rrtmgp_lookup *lk = nullptr;
        const int rc = create(desc, device_ids[i], &lk);
        if (rc != RRTMGP_OK) {
            const std::string m = last_error_string();
            rrtmgp_hip_lookup_destroy(head);
            return set_error(rc, m);
        }
        if (!head) head = lk; else head->replicas.push_back(lk);
    }
    *out = head;
    return RRTMGP_OK;
}

}  // namespace rrtmgp

using namespace rrtmgp;

extern "C" {

int rrtmgp_hip_gas_lookup_create_multi(const rrtmgp_gas_lookup_desc *desc, const int32_t *device_ids, int ndev,
                                       rrtmgp_lookup **out) {
    return create_lookup_multi(desc, device_ids, ndev, out, rrtmgp_hip_gas_lookup_create);
}
int rrtmgp_hip_cloud_lookup_create_multi(const rrtmgp_cloud_lookup_desc *desc, const int32_t *device_ids, int ndev,
                                         rrtmgp_lookup **out) {
    return create_lookup_multi(desc, device_ids, ndev, out, rrtmgp_hip_cloud_lookup_create);
}
int rrtmgp_hip_aerosol_lookup_create_multi(const rrtmgp_aerosol_lookup_desc *desc, const int32_t *device_ids, int ndev,
                                           rrtmgp_lookup **out) {
    return create_lookup_multi(desc, device_ids, ndev, out, rrtmgp_hip_aerosol_lookup_create);
}

int rrtmgp_hip_workspace_create_multi(const int32_t *device_ids, int ndev, int64_t ncol, int64_t nlay, int32_t ftype,
                                      rrtmgp_workspace **out) {
    RR_CHECK(device_ids && out && ndev >= 1 && ndev <= 64, "bad argument");
    RR_CHECK(ftype == RRTMGP_F32 || ftype == RRTMGP_F64, "ftype must be 4 or 8");
    RR_CHECK(ncol >= ndev && nlay >= 2 && ncol < (1LL << 31) && nlay < 4096, "bad workspace dimensions (need ncol >= shards)");
    auto *head = new rrtmgp_workspace();
    head->device = device_ids[0]; head->ftype = ftype; head->ncol = ncol; head->nlay = nlay;
    head->shard_c0.resize(ndev + 1);
    for (int s = 0; s <= ndev; s++) head->shard_c0[s] = ncol * s / ndev;   // the ranges of rrtmgp.jl_amd/sharding.py
    for (int s = 0; s < ndev; s++) {
        rrtmgp_workspace *w = nullptr;
        const int rc = rrtmgp_hip_workspace_create(device_ids[s], head->shard_c0[s + 1] - head->shard_c0[s], nlay, ftype, &w);
        if (rc != RRTMGP_OK) {
            const std::string m = last_error_string();
            rrtmgp_hip_workspace_destroy(head);
            return set_error(rc, m);
        }
        w->head = head;
        head->shards.push_back(w);
    }
    *out = head;
    return RRTMGP_OK;
}

int rrtmgp_hip_local_cpus(const char *pci_bus_id, int32_t *cpus, int cap) {
    std::vector<int> v;
    const int n = local_cpus_of(pci_bus_id, v);
    if (n < 0) return set_error(RRTMGP_EINVAL, "no local_cpulist for this PCI bus id");
    for (int i = 0; i < n && i < cap && cpus; i++) cpus[i] = v[i];
    return n;
}

int rrtmgp_hip_workspace_shards(const rrtmgp_workspace *ws) {
    if (!ws) return set_error(RRTMGP_EINVAL, "null workspace");
    return ws->shards.empty() ? 1 : (int)ws->shards.size();
}

}  // extern "C"
