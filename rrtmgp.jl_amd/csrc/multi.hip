// multi.hip — several GPUs from one host process: replicated lookups, sharded workspaces,
// one host thread + one stream per shard (include/rrtmgp_hip.h, "several GPUs from ONE host process").
//
// Columns shard embarrassingly: shard s owns the contiguous global columns
// [s*ncol/n, (s+1)*ncol/n), which is one contiguous slab of every state / boundary / flux array because ncol is
// their slowest dimension.  There is no collective: every shard stages its own slab in, solves it with
// `col_offset` advanced to its first global column (the McICA stream is keyed by the global column), and
// stages its fluxes back into the caller's arrays.  The public entry points (api.hip) do the slicing and
// re-enter themselves with a shard workspace; this file owns the objects and the fan-out.
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

int multi_run(rrtmgp_workspace *ws, const std::function<int(rrtmgp_workspace *, size_t, size_t)> &shard_call) {
    const size_t n = ws->shards.size();
    std::vector<int> rc(n, RRTMGP_OK);
    std::vector<std::string> msg(n);
    auto run = [&](size_t s) {
        const size_t c0 = (size_t)ws->shard_c0[s], c1 = (size_t)ws->shard_c0[s + 1];
        if (c1 == c0) return;
        rc[s] = shard_call(ws->shards[s], c0, c1 - c0);
        if (rc[s] != RRTMGP_OK) msg[s] = last_error_string();  // the message lives in the worker's thread-local slot
    };
    std::vector<std::thread> workers;
    workers.reserve(n);
    for (size_t s = 1; s < n; s++) workers.emplace_back(run, s);
    run(0);
    for (auto &t : workers) t.join();
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

int rrtmgp_hip_workspace_shards(const rrtmgp_workspace *ws) {
    if (!ws) return set_error(RRTMGP_EINVAL, "null workspace");
    return ws->shards.empty() ? 1 : (int)ws->shards.size();
}

}  // extern "C"
