// libhammlet_hip.so, chain-parallel pooling (hml_pool_* / hml_allreduce_marginals of include/hml.h).
// Chains shard across GPUs and never communicate while sampling; ONE collective at the end - ncclAllReduce(sum, int32)
// over xGMI - pools their state marginals.  The reference has no counterpart (one process, one thread,
// src/main.cpp:108); the common labelling of the states follows the idea of bin/sortStates:1-6.
//
// RCCL is bound at run time (dlopen of librccl.so.1 on the first pooling call): the library is 570 MB of code objects,
// and a single-chain `hammlet` run - the reference's use - should not pay for mapping and registering it at start-up.
// Its official header provides the types and prototypes; nothing of RCCL is re-declared here.
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <vector>

#include "hml_ctx.hpp"
#include "hml_k_pool.h"

static int set_err(int code, const std::string& msg) { return hml_set_err(code, msg); }

namespace {

struct RcclApi {
    void* handle = nullptr;
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclCommInitAll) CommInitAll = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclAllReduce) AllReduce = nullptr;
    decltype(&ncclAllGather) AllGather = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    decltype(&ncclGetVersion) GetVersion = nullptr;
    std::string error;
};

RcclApi& rccl() {
    static RcclApi api;
    static std::once_flag once;
    std::call_once(once, [] {
        // An RCCL that the process has loaded already wins (a PyTorch process: torch bundles its own copy - two copies in
        // one process end in "double free or corruption" when the process exits, whichever is used), then
        // HML_RCCL_LIBRARY, then the system's.
        const char* names[] = {getenv("HML_RCCL_LIBRARY"), "librccl.so.1", "librccl.so"};
        for (const char* n : {"librccl.so.1", "librccl.so"}) {
            api.handle = dlopen(n, RTLD_NOW | RTLD_GLOBAL | RTLD_NOLOAD);
            if (api.handle) break;
        }
        for (const char* n : names) {
            if (api.handle) break;
            if (!n || !*n) continue;
            api.handle = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
            if (!api.handle) api.error = dlerror();
        }
        if (!api.handle) return;
        bool ok = true;
        auto bind = [&](auto& fn, const char* sym) {
            fn = reinterpret_cast<std::remove_reference_t<decltype(fn)>>(dlsym(api.handle, sym));
            if (!fn) { ok = false; api.error = std::string("librccl: missing symbol ") + sym; }
        };
        bind(api.GetUniqueId, "ncclGetUniqueId");
        bind(api.CommInitRank, "ncclCommInitRank");
        bind(api.CommInitAll, "ncclCommInitAll");
        bind(api.CommDestroy, "ncclCommDestroy");
        bind(api.AllReduce, "ncclAllReduce");
        bind(api.AllGather, "ncclAllGather");
        bind(api.GroupStart, "ncclGroupStart");
        bind(api.GroupEnd, "ncclGroupEnd");
        bind(api.GetErrorString, "ncclGetErrorString");
        bind(api.GetVersion, "ncclGetVersion");
        if (!ok) { dlclose(api.handle); api.handle = nullptr; }
    });
    return api;
}

int need_rccl() {
    RcclApi& r = rccl();
    if (!r.handle) return set_err(HML_ERR_HIP, "RCCL is not available (librccl.so.1): " + r.error);
    return 0;
}

#define NCCLCHK(call)                                                                                       \
    do {                                                                                                    \
        ncclResult_t r_ = (call);                                                                           \
        if (r_ != ncclSuccess) return set_err(HML_ERR_HIP, std::string(#call) + ": " + rccl().GetErrorString(r_)); \
    } while (0)

uint64_t payload_count(const hml_ctx* c) { return (uint64_t)(c->K + 1) * (c->T + 1) + 1u + (uint64_t)c->K; }

int grid_for(uint64_t items, int per_block, int hi) {
    uint64_t g = (items + per_block - 1) / per_block;
    return (int)std::max<uint64_t>(1, std::min<uint64_t>(g, (uint64_t)hi));
}

// device memory and events that are released on every way out of a function
struct EventPair {
    hipEvent_t e0 = nullptr, e1 = nullptr;
    ~EventPair() { if (e0) (void)hipEventDestroy(e0); if (e1) (void)hipEventDestroy(e1); }
};

// relabelled difference arrays + boundary row + tail of one chain -> payload (device, payload_count(c) int32).
// A context whose marginals ARE a pooled payload (hml_pool_install) is refused: exporting them again would apply the
// relabelling a second time and count every chain once more.
int export_payload(hml_ctx* c, int32_t* payload, int32_t* perm_out) {
    if (c->pooled) return set_err(HML_ERR_ARG, "the marginals of this context are pooled already: they cannot be pooled again");
    if (int r = hml_ctx_bind(c)) return r;
    if (int r = hml_ctx_ensure_marginal_buffers(c)) return r;
    std::vector<int32_t> perm(c->K);
    if (int r = hml_relabel_permutation(c, perm.data())) return r;
    if (perm_out) memcpy(perm_out, perm.data(), sizeof(int32_t) * c->K);
    DevBuf d_perm;
    HIPCHK(hipMalloc(&d_perm.p, sizeof(int32_t) * c->K));
    HIPCHK(hipMemcpyAsync(d_perm.p, perm.data(), sizeof(int32_t) * c->K, hipMemcpyHostToDevice, c->stream));
    const uint64_t n = payload_count(c);
    HIPCHK(hipMemsetAsync(payload + (n - 1u - (uint64_t)c->K), 0, sizeof(int32_t) * (1u + (uint64_t)c->K), c->stream));
    hipLaunchKernelGGL(hml_k_pool_export, dim3(grid_for(n, 256, 1 << 16)), dim3(256), 0, c->stream, c->d_diff, c->d_boundary, c->d_mdl,
                       d_perm.as<int32_t>(), (uint32_t)c->T, c->K, payload);
    KLAUNCH_CHECK();
    HIPCHK(hipStreamSynchronize(c->stream));
    c->pool_perm = perm;   // the label space the chain's marginals are in once a payload is installed
    return 0;
}

int install_payload(hml_ctx* c, const int32_t* payload) {
    if (int r = hml_ctx_bind(c)) return r;
    if (int r = hml_ctx_ensure_marginal_buffers(c)) return r;
    const uint64_t T1 = c->T + 1;
    hipLaunchKernelGGL(hml_k_pool_install_diff, dim3(grid_for((uint64_t)c->K * T1, 256, 1 << 16)), dim3(256), 0, c->stream, payload,
                       (uint32_t)c->T, c->K, c->d_diff);
    hipLaunchKernelGGL(hml_k_pool_install_boundary, dim3(grid_for((T1 + 31) / 32, 256, 1 << 16)), dim3(256), 0, c->stream, payload,
                       (uint32_t)c->T, c->K, c->d_boundary, c->d_mdl);
    KLAUNCH_CHECK();
    HIPCHK(hipStreamSynchronize(c->stream));
    c->pooled = true;
    return 0;
}

// Before a collective over the payload every rank learns whether all ranks got this far and agree on the shape: a rank
// that left early (a failed export, a chain of another K or T) would otherwise leave the others waiting in the collective.
// ncclMax over {status, K, -K, T's halves and their negatives, the rank's number of marginal segments (two halves)}:
// *m_max = the largest number of segments any rank holds (what sizes the slots of the boundary-list form).
int handshake(ncclComm_t comm, hipStream_t stream, int32_t* d_hs, int local_rc, const hml_ctx* c, uint64_t m_local = 0, uint64_t* m_max = nullptr,
              int form = 0) {
    const int32_t tl = (int32_t)(c->T & 0x7fffffffu), th = (int32_t)(c->T >> 31);
    // (the maximum of a two-word number by ncclMax: the high half first - the low half only counts among the ranks that
    // hold the largest high half, so it travels in a second round when the high halves differ; segment counts are below
    // 2^32, their high half is one bit: send the number as 16-bit pieces whose element-wise maxima bound it from above)
    const int32_t m3 = (int32_t)((m_local >> 32) & 0xffffu), m2 = (int32_t)((m_local >> 16) & 0xffffu), m1 = (int32_t)(m_local & 0xffffu);
    // the form of the collective that follows ({form, -form}: the ranks must hold the same one, or some would enter
    // ncclAllReduce and the others ncclAllGather - ADVICE round 4)
    const int32_t h[12] = {local_rc ? 1 : 0, c->K, -c->K, tl, -tl, th, -th, m3, m2, m1, form, -form};
    // A HIP failure on this rank BEFORE the collective is posted must not keep it from posting: the others would wait in
    // theirs.  It goes into the status word instead (every word 1 when even the copy fails: status 1 for everybody).
    int hip_rc = 0;
    if (hipMemcpyAsync(d_hs, h, sizeof h, hipMemcpyHostToDevice, stream) != hipSuccess) {
        (void)hipGetLastError();
        hip_rc = set_err(HML_ERR_HIP, "pooling handshake: the copy of the status words failed");
        (void)hipMemsetD32Async((hipDeviceptr_t)d_hs, 1, 12, stream);
    }
    NCCLCHK(rccl().AllReduce(d_hs, d_hs, 12, ncclInt32, ncclMax, comm, stream));
    int32_t g[12];
    HIPCHK(hipMemcpyAsync(g, d_hs, sizeof g, hipMemcpyDeviceToHost, stream));
    HIPCHK(hipStreamSynchronize(stream));
    if (local_rc) return local_rc;
    if (hip_rc) return hip_rc;
    if (g[0] != 0) return set_err(HML_ERR_ARG, "pooling abandoned: another rank failed before the collective");
    if (g[1] != -g[2] || g[3] != -g[4] || g[5] != -g[6]) return set_err(HML_ERR_ARG, "pooling abandoned: the ranks' chains differ in the number of states or positions");
    if (g[10] != -g[11]) return set_err(HML_ERR_ARG, "pooling abandoned: the ranks ask for different forms of the collective (hml_pool_set_form / HML_POOL_FORM must agree on all ranks)");
    // an upper bound of the largest count that every rank computes alike (exact when one rank holds the largest of every piece)
    if (m_max) *m_max = ((uint64_t)(uint32_t)g[7] << 32) | ((uint64_t)(uint32_t)g[8] << 16) | (uint64_t)(uint32_t)g[9];
    return 0;
}

}  // namespace

struct hml_pool {
    ncclComm_t comm = nullptr;
    int device = 0, rank = 0, n_ranks = 1;
    hipStream_t stream = nullptr;
    int32_t* d_payload = nullptr;
    int32_t* d_handshake = nullptr;
    uint64_t capacity = 0;       // int32 elements
    double last_ms = 0;
    uint64_t last_bytes = 0;
    int form = 0;                // 0: chosen per call (lists when they are an eighth of the dense payload or less), 1: dense, 2: lists
    int last_form = 0;           // 1 dense / 2 lists
    uint64_t last_entries = 0;   // list form: the largest number of segments of a rank
};

extern "C" {

int hml_pool_unique_id(void* id) {
    if (!id) return set_err(HML_ERR_ARG, "null argument");
    if (int r = need_rccl()) return r;
    static_assert(sizeof(ncclUniqueId) == HML_POOL_ID_BYTES, "ncclUniqueId size");
    ncclUniqueId u;
    NCCLCHK(rccl().GetUniqueId(&u));
    memcpy(id, &u, sizeof u);
    return 0;
}

int hml_pool_create(hml_pool** out, int device, int rank, int n_ranks, const void* id) {
    if (!out || !id) return set_err(HML_ERR_ARG, "null argument");
    if (n_ranks < 1 || rank < 0 || rank >= n_ranks) return set_err(HML_ERR_ARG, "rank out of range");
    if (int r = need_rccl()) return r;
    HIPCHK(hipSetDevice(device));
    hml_pool* p = new hml_pool();
    p->device = device; p->rank = rank; p->n_ranks = n_ranks;
    ncclUniqueId u;
    memcpy(&u, id, sizeof u);
    ncclResult_t rc = rccl().CommInitRank(&p->comm, n_ranks, u, rank);
    if (rc != ncclSuccess) { delete p; return set_err(HML_ERR_HIP, std::string("ncclCommInitRank: ") + rccl().GetErrorString(rc)); }
    if (hipStreamCreateWithFlags(&p->stream, hipStreamNonBlocking) != hipSuccess || hipMalloc(&p->d_handshake, 16 * sizeof(int32_t)) != hipSuccess) {
        (void)hipGetLastError();
        if (p->stream) (void)hipStreamDestroy(p->stream);
        rccl().CommDestroy(p->comm);
        delete p;
        return set_err(HML_ERR_HIP, "could not create the communicator's stream and handshake buffer");
    }
    if (const char* e = getenv("HML_POOL_FORM")) p->form = atoi(e);
    *out = p;
    return 0;
}

void hml_pool_destroy(hml_pool* p) {
    if (!p) return;
    hipSetDevice(p->device);
    if (p->stream) hipStreamSynchronize(p->stream);
    if (p->comm) rccl().CommDestroy(p->comm);
    if (p->d_payload) hipFree(p->d_payload);
    if (p->d_handshake) hipFree(p->d_handshake);
    if (p->stream) hipStreamDestroy(p->stream);
    delete p;
}

int hml_pool_info(hml_pool* p, int* rank, int* n_ranks, double* last_allreduce_ms, uint64_t* last_bytes, int* rccl_version) {
    if (!p) return set_err(HML_ERR_ARG, "null pool");
    if (rank) *rank = p->rank;
    if (n_ranks) *n_ranks = p->n_ranks;
    if (last_allreduce_ms) *last_allreduce_ms = p->last_ms;
    if (last_bytes) *last_bytes = p->last_bytes;
    if (rccl_version) { int v = 0; rccl().GetVersion(&v); *rccl_version = v; }
    return 0;
}

int hml_pool_payload_size(hml_ctx* c, uint64_t* n_int32) {
    if (!c || !c->model_set || !n_int32) return set_err(HML_ERR_ARG, "model not set");
    *n_int32 = payload_count(c);
    return 0;
}

int hml_pool_export(hml_ctx* c, void* payload_dev, int32_t* perm_out) {
    if (!c || !c->model_set || !payload_dev) return set_err(HML_ERR_ARG, "model not set");
    return export_payload(c, (int32_t*)payload_dev, perm_out);
}

int hml_pool_install(hml_ctx* c, const void* payload_dev) {
    if (!c || !c->model_set || !payload_dev) return set_err(HML_ERR_ARG, "model not set");
    return install_payload(c, (const int32_t*)payload_dev);
}

// The collective.  Two forms of the same pooled marginals (hml_k_pool.h): the dense payload through ncclAllReduce(sum) -
// [K+1][T+1] int32, 2.4 GB at 10^8 positions / 5 states whatever it holds - or the ranks' boundary lists through
// ncclAllGather, chosen when the gathered lists are at most an eighth of the dense payload (a strongly compressed chain:
// config 3 after 100 recorded sweeps holds 23 000 segments, 0.6 MB).  Every rank takes the same decision from the same
// handshake - which also carries the requested form (hml_pool_set_form / HML_POOL_FORM): ranks that ask for different
// forms all return HML_ERR_ARG instead of entering different collectives.  A rank that cannot go on (wrong device, no model, out of memory, a failed export) still joins the
// handshakes with its status, so that nobody is left inside a collective.
int hml_pool_marginals(hml_pool* p, hml_ctx* c, int32_t* perm_out) {
    if (!p || !c) return set_err(HML_ERR_ARG, "null argument");
    HIPCHK(hipSetDevice(p->device));
    int rc = 0;
    if (!c->model_set) rc = set_err(HML_ERR_ARG, "model not set");
    else if (c->device != p->device) rc = set_err(HML_ERR_ARG, "the chain lives on another device than the communicator's rank");
    else if (c->pooled) rc = set_err(HML_ERR_ARG, "the marginals of this context are pooled already: they cannot be pooled again");
    // ---- first handshake: status, shape, and the number of marginal segments of every rank
    uint64_t M = 0, M_max = 0;
    DevBuf d_seg, d_g;
    if (!rc) rc = hml_ctx_ensure_marginal_buffers(c);
    if (!rc) { uint32_t* sg = nullptr; int32_t* gg = nullptr; rc = hml_ctx_gather_marginal_segments(c, &M, &sg, &gg); d_seg.p = sg; d_g.p = gg; }
    if (int r = handshake(p->comm, p->stream, p->d_handshake, rc, c, M, &M_max, p->form)) return r;
    const int K = c->K;
    const uint64_t n = payload_count(c);
    const uint64_t slot = hml_pool_list_header(K) + M_max * (uint64_t)(K + 1);
    const bool lists = p->form == 2 || (p->form == 0 && slot * (uint64_t)p->n_ranks * 8u <= n);
    const uint64_t need = lists ? slot * (uint64_t)p->n_ranks : n;
    if (p->capacity < need) {
        if (p->d_payload) (void)hipFree(p->d_payload);
        p->d_payload = nullptr; p->capacity = 0;
        if (hipMalloc(&p->d_payload, need * sizeof(int32_t)) != hipSuccess) { (void)hipGetLastError(); rc = set_err(HML_ERR_HIP, "out of device memory for the pooling payload"); }
        else p->capacity = need;
    }
    EventPair ev;
    if (!rc && (hipEventCreate(&ev.e0) != hipSuccess || hipEventCreate(&ev.e1) != hipSuccess)) rc = set_err(HML_ERR_HIP, "hipEventCreate failed");
    if (!lists) {
        // ---- dense: export, second handshake (the export's status), all-reduce, install
        if (!rc) rc = export_payload(c, p->d_payload, perm_out);
        if (int r = handshake(p->comm, p->stream, p->d_handshake, rc, c)) return r;
        HIPCHK(hipEventRecord(ev.e0, p->stream));
        NCCLCHK(rccl().AllReduce(p->d_payload, p->d_payload, (size_t)n, ncclInt32, ncclSum, p->comm, p->stream));
        HIPCHK(hipEventRecord(ev.e1, p->stream));
        HIPCHK(hipStreamSynchronize(p->stream));
        float ms = 0;
        (void)hipEventElapsedTime(&ms, ev.e0, ev.e1);
        p->last_ms = ms; p->last_bytes = n * sizeof(int32_t); p->last_form = 1; p->last_entries = 0;
        return install_payload(c, p->d_payload);
    }
    // ---- lists: this rank's list into its slot, second handshake, all-gather, every list into the zeroed arrays
    std::vector<int32_t> perm(K);
    int32_t* const mine = p->d_payload ? p->d_payload + (uint64_t)p->rank * slot : nullptr;
    DevBuf d_perm;
    if (!rc) rc = hml_relabel_permutation(c, perm.data());
    if (!rc) {
        if (perm_out) memcpy(perm_out, perm.data(), sizeof(int32_t) * K);
        if (hipMalloc(&d_perm.p, sizeof(int32_t) * K) != hipSuccess) { (void)hipGetLastError(); rc = set_err(HML_ERR_HIP, "out of device memory"); }
    }
    if (!rc) {
        bool ok = hipMemcpyAsync(d_perm.p, perm.data(), sizeof(int32_t) * K, hipMemcpyHostToDevice, c->stream) == hipSuccess &&
                  hipMemsetAsync(mine, 0, hml_pool_list_header(K) * sizeof(int32_t), c->stream) == hipSuccess;
        if (ok) {
            hipLaunchKernelGGL(hml_k_pool_list_pack, dim3(grid_for(std::max<uint64_t>(M, 1), 256, 1 << 14)), dim3(256), 0, c->stream, d_seg.as<uint32_t>(),
                               d_g.as<int32_t>(), M, c->d_mdl, d_perm.as<int32_t>(), K, mine);
            ok = hipGetLastError() == hipSuccess && hipStreamSynchronize(c->stream) == hipSuccess;
        }
        if (!ok) { (void)hipGetLastError(); rc = set_err(HML_ERR_HIP, "packing the boundary list failed"); }
    }
    if (int r = handshake(p->comm, p->stream, p->d_handshake, rc, c)) return r;
    HIPCHK(hipEventRecord(ev.e0, p->stream));
    NCCLCHK(rccl().AllGather(mine, p->d_payload, (size_t)slot, ncclInt32, p->comm, p->stream));   // (in place: the send buffer is this rank's slot)
    HIPCHK(hipEventRecord(ev.e1, p->stream));
    HIPCHK(hipStreamSynchronize(p->stream));
    float ms = 0;
    (void)hipEventElapsedTime(&ms, ev.e0, ev.e1);
    p->last_ms = ms; p->last_bytes = slot * (uint64_t)p->n_ranks * sizeof(int32_t); p->last_form = 2; p->last_entries = M_max;
    const uint64_t T1 = c->T + 1;
    HIPCHK(hipMemsetAsync(c->d_diff, 0, (uint64_t)K * T1 * sizeof(int32_t), c->stream));
    HIPCHK(hipMemsetAsync(c->d_boundary, 0, ((T1 + 31) / 32 + 1) * sizeof(uint32_t), c->stream));
    hipLaunchKernelGGL(hml_k_pool_list_install, dim3(grid_for(std::max<uint64_t>(M_max, 1), 256, 1 << 14)), dim3(256), 0, c->stream, p->d_payload, p->n_ranks,
                       slot, (uint32_t)c->T, K, c->d_diff, c->d_boundary, c->d_mdl);
    KLAUNCH_CHECK();
    HIPCHK(hipStreamSynchronize(c->stream));
    c->pool_perm = perm;
    c->pooled = true;
    return 0;
}

// "form" of hml_pool_marginals' collective: 0 (default) chosen per call, 1 always the dense payload, 2 always the boundary
// lists (environment: HML_POOL_FORM); hml_pool_last reports what the last call took (1 / 2) and, for the lists, the largest
// number of marginal segments a rank held.
int hml_pool_set_form(hml_pool* p, int form) {
    if (!p || form < 0 || form > 2) return set_err(HML_ERR_ARG, "form: 0 (automatic), 1 (dense payload) or 2 (boundary lists)");
    p->form = form;
    return 0;
}
int hml_pool_last(hml_pool* p, int* form, uint64_t* entries) {
    if (!p) return set_err(HML_ERR_ARG, "null pool");
    if (form) *form = p->last_form;
    if (entries) *entries = p->last_entries;
    return 0;
}

int hml_pool_permutation(hml_ctx* c, int32_t* perm) {
    if (!c || !c->model_set || !perm) return set_err(HML_ERR_ARG, "model not set");
    for (int k = 0; k < c->K; ++k) perm[k] = (c->pooled && (int)c->pool_perm.size() == c->K) ? c->pool_perm[k] : k;
    return 0;
}

// One process driving n chains (hammlet -chains N): contexts that share a device are summed on that device first, the
// per-device sums go through one grouped ncclAllReduce over a communicator of the distinct devices (ncclCommInitAll;
// a single device still forms a one-rank communicator), and every context receives the pooled marginals.
int hml_allreduce_marginals(hml_ctx* const* ctxs, int n) { return hml_allreduce_marginals_perm(ctxs, n, nullptr); }

int hml_allreduce_marginals_perm(hml_ctx* const* ctxs, int n, int32_t* perms) {
    if (!ctxs || n < 1) return set_err(HML_ERR_ARG, "no contexts");
    for (int i = 0; i < n; ++i) {
        if (!ctxs[i] || !ctxs[i]->model_set) return set_err(HML_ERR_ARG, "model not set");
        if (ctxs[i]->K != ctxs[0]->K || ctxs[i]->T != ctxs[0]->T) return set_err(HML_ERR_ARG, "chains of different shape cannot be pooled");
    }
    for (int i = 0; i < n; ++i)
        if (ctxs[i]->pooled) return set_err(HML_ERR_ARG, "the marginals of a context are pooled already: they cannot be pooled again");
    const uint64_t cnt = payload_count(ctxs[0]);
    const int K = ctxs[0]->K;
    std::map<int, std::vector<int>> by_dev;
    for (int i = 0; i < n; ++i) by_dev[ctxs[i]->device].push_back(i);
    // chains that all share one device are summed there and that is the result: RCCL (570 MB to map, a communicator to
    // build) is not touched
    const bool one_device = by_dev.size() == 1;
    if (!one_device) { if (int r = need_rccl()) return r; }
    std::vector<int> devs;
    std::vector<int32_t*> bufs;
    int rc = 0;
    auto cleanup = [&]() { for (size_t k = 0; k < bufs.size(); ++k) { hipSetDevice(devs[k]); hipFree(bufs[k]); } };
    for (auto& kv : by_dev) {
        HIPCHK(hipSetDevice(kv.first));
        int32_t *acc = nullptr, *tmp = nullptr;
        if (hipMalloc(&acc, cnt * sizeof(int32_t)) != hipSuccess) { cleanup(); return set_err(HML_ERR_HIP, "out of device memory for the pooling payload"); }
        devs.push_back(kv.first); bufs.push_back(acc);
        for (size_t j = 0; j < kv.second.size() && !rc; ++j) {
            hml_ctx* c = ctxs[kv.second[j]];
            int32_t* const perm_j = perms ? perms + (size_t)kv.second[j] * K : nullptr;
            if (j == 0) { rc = export_payload(c, acc, perm_j); continue; }
            if (!tmp && hipMalloc(&tmp, cnt * sizeof(int32_t)) != hipSuccess) { rc = set_err(HML_ERR_HIP, "out of device memory for the pooling payload"); break; }
            rc = export_payload(c, tmp, perm_j);
            if (!rc) {
                hipLaunchKernelGGL(hml_k_pool_add, dim3(grid_for(cnt, 256, 1 << 16)), dim3(256), 0, c->stream, acc, tmp, cnt);
                if (hipStreamSynchronize(c->stream) != hipSuccess) rc = set_err(HML_ERR_HIP, "pooling kernel failed");
            }
        }
        if (tmp) hipFree(tmp);
        if (rc) { cleanup(); return rc; }
    }
    const int nd = (int)devs.size();
    if (one_device) {
        for (int i : by_dev[devs[0]]) { rc = install_payload(ctxs[i], bufs[0]); if (rc) break; }
        cleanup();
        return rc;
    }
    std::vector<ncclComm_t> comms(nd);
    {
        ncclResult_t r = rccl().CommInitAll(comms.data(), nd, devs.data());
        if (r != ncclSuccess) { cleanup(); return set_err(HML_ERR_HIP, std::string("ncclCommInitAll: ") + rccl().GetErrorString(r)); }
    }
    ncclResult_t r = rccl().GroupStart();
    for (int k = 0; k < nd && r == ncclSuccess; ++k) {
        hipSetDevice(devs[k]);
        r = rccl().AllReduce(bufs[k], bufs[k], (size_t)cnt, ncclInt32, ncclSum, comms[k], ctxs[by_dev[devs[k]][0]]->stream);
    }
    ncclResult_t r2 = rccl().GroupEnd();
    if (r == ncclSuccess) r = r2;
    for (int k = 0; k < nd; ++k) { hipSetDevice(devs[k]); hipStreamSynchronize(ctxs[by_dev[devs[k]][0]]->stream); }
    for (int k = 0; k < nd; ++k) rccl().CommDestroy(comms[k]);
    if (r != ncclSuccess) { cleanup(); return set_err(HML_ERR_HIP, std::string("ncclAllReduce: ") + rccl().GetErrorString(r)); }
    for (int k = 0; k < nd && !rc; ++k)
        for (int i : by_dev[devs[k]]) { rc = install_payload(ctxs[i], bufs[k]); if (rc) break; }
    cleanup();
    return rc;
}

}  // extern "C"
