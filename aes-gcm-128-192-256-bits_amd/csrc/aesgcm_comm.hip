// aesgcm_comm.hip -- the inter-GPU step of the sharded path, in the library (no PyTorch anywhere on it).
//
// One GCM message shards over G GPUs with no data-path exchange except 16 bytes per rank per message: rank g
// en/decrypts its block range with aesgcm_shard_crypt_dev, which leaves the WEIGHTED partial
//     W_g = (sum_{i in shard} X_i * H^(end_g-1-i)) * H^(n_blocks - end_g)
// in device memory; the partials are all-gathered (RCCL has no XOR reduction, rccl.h ncclRedOp_t, hence gather +
// fold) and aesgcm_shard_finalize_dev XOR-folds them on the device into the tag.  The reference's nearest
// counterpart is the two-way split of one multiplication, src/gcm_ghash.vhd:317-333: X*H = (Xhi||0)*H ^ (0||Xlo)*H.
//
// Two shapes of the same thing:
//   aesgcm_comm_*   one PROCESS per GPU (how bench.py --gpus N runs under torch.distributed.run): the caller
//                   distributes a 128-byte RCCL unique id (rank 0 makes it), every rank calls ncclCommInitRank.
//   aesgcm_mgpu_*   one process driving ndev GPUs: ncclCommInitAll + grouped all-gather.
// RCCL is reached through dlopen("librccl.so.1") at first use, so the library loads (and every single-GPU entry
// point works) on a box without RCCL; without it these entry points return AESGCM_ERCCL.
#include "../../include/aesgcm.h"

#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>      // types and prototypes only: nothing here links against librccl

#include <algorithm>
#include <mutex>
#include <new>
#include <stdio.h>
#include <string.h>
#include <vector>

namespace {

thread_local char g_cerr[256] = "";

struct Rccl {
    void *h = nullptr;
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclCommInitAll) CommInitAll = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclCommCount) CommCount = nullptr;
    decltype(&ncclCommUserRank) CommUserRank = nullptr;
    decltype(&ncclAllGather) AllGather = nullptr;
    decltype(&ncclAllReduce) AllReduce = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    bool ok = false;
};
Rccl g_rccl;
std::once_flag g_rccl_once;

void rccl_load() {
#ifdef AESGCM_TEST_RCCL_LIB
    const char *names[] = {AESGCM_TEST_RCCL_LIB};             // tests/fake_hip: the host side linked against a fake runtime (no product build defines this)
#else
    const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
#endif
    for (const char *n : names) { g_rccl.h = dlopen(n, RTLD_NOW | RTLD_GLOBAL); if (g_rccl.h) break; }
    if (!g_rccl.h) return;
#define SYM(f) g_rccl.f = reinterpret_cast<decltype(g_rccl.f)>(dlsym(g_rccl.h, "nccl" #f)); if (!g_rccl.f) return
    SYM(GetUniqueId); SYM(CommInitRank); SYM(CommInitAll); SYM(CommDestroy); SYM(CommCount); SYM(CommUserRank);
    SYM(AllGather); SYM(AllReduce); SYM(GroupStart); SYM(GroupEnd); SYM(GetErrorString);
#undef SYM
    g_rccl.ok = true;
}
int rccl_get(Rccl **r) {
    std::call_once(g_rccl_once, rccl_load);
    if (!g_rccl.ok) { snprintf(g_cerr, sizeof g_cerr, "librccl.so.1 not loadable: %s", dlerror() ? dlerror() : "symbol missing"); return AESGCM_ERCCL; }
    *r = &g_rccl;
    return AESGCM_OK;
}
int nccl_fail(ncclResult_t e, const char *what) {
    snprintf(g_cerr, sizeof g_cerr, "%s: %s", what, g_rccl.GetErrorString ? g_rccl.GetErrorString(e) : "rccl error");
    return AESGCM_ERCCL;
}
int hip_fail(hipError_t e, const char *what) {
    snprintf(g_cerr, sizeof g_cerr, "%s: %s", what, hipGetErrorString(e));
    return AESGCM_EHIP;
}
#define NCHK(call) do { ncclResult_t _e = (call); if (_e != ncclSuccess) return nccl_fail(_e, #call); } while (0)
#define HCHK(call) do { hipError_t _e = (call); if (_e != hipSuccess) return hip_fail(_e, #call); } while (0)

}  // namespace

struct aesgcm_comm {
    int device = 0, n_ranks = 0, rank = 0;
    ncclComm_t comm = nullptr;
    hipStream_t stream = nullptr;      // for the host-value collectives (barrier, max)
    double *d_scalar = nullptr;        // 2 doubles of device scratch
};

#define MGPU_RING 8                     /* messages that may be queued (tag = NULL) before their tags are collected */
struct aesgcm_mgpu {
    int ndev = 0;
    std::vector<int> dev;
    std::vector<aesgcm_ctx *> ctx;
    std::vector<ncclComm_t> comm;
    std::vector<hipStream_t> st;
    std::vector<unsigned char *> part, all;    // per device: own 16-byte partial, ndev gathered partials.  all[0] is a ring of MGPU_RING such rows: a queued message's row waits there for its finalize
    // the queued messages, oldest first: what their finalize needs
    unsigned long long seq = 0;                 // messages queued so far
    int pending = 0;
    uint8_t iv[MGPU_RING][12];
    size_t aad_len[MGPU_RING];
    uint64_t total[MGPU_RING];
};

extern "C" {

const char *aesgcm_comm_last_error(void) { return g_cerr; }

int aesgcm_comm_unique_id(uint8_t id[AESGCM_COMM_ID_BYTES]) {
    if (!id) return AESGCM_EARG;
    Rccl *r; int rc = rccl_get(&r); if (rc) return rc;
    static_assert(sizeof(ncclUniqueId) == AESGCM_COMM_ID_BYTES, "ncclUniqueId size");
    ncclUniqueId u;
    NCHK(r->GetUniqueId(&u));
    memcpy(id, &u, sizeof u);
    return AESGCM_OK;
}

int aesgcm_comm_create(aesgcm_comm **out, int device, const uint8_t id[AESGCM_COMM_ID_BYTES], int n_ranks, int rank) {
    if (!out || !id || n_ranks < 1 || rank < 0 || rank >= n_ranks) return AESGCM_EARG;
    *out = nullptr;
    Rccl *r; int rc = rccl_get(&r); if (rc) return rc;
    HCHK(hipSetDevice(device));
    aesgcm_comm *c = new (std::nothrow) aesgcm_comm();
    if (!c) return AESGCM_ENOMEM;
    c->device = device;
    ncclUniqueId u;
    memcpy(&u, id, sizeof u);
    ncclResult_t e = r->CommInitRank(&c->comm, n_ranks, u, rank);
    if (e != ncclSuccess) { delete c; return nccl_fail(e, "ncclCommInitRank"); }
    // what RCCL itself reports, not what the caller asked for
    if (r->CommCount(c->comm, &c->n_ranks) != ncclSuccess || r->CommUserRank(c->comm, &c->rank) != ncclSuccess) {
        r->CommDestroy(c->comm); delete c; snprintf(g_cerr, sizeof g_cerr, "ncclCommCount/UserRank failed"); return AESGCM_ERCCL;
    }
    hipError_t he = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
    if (he == hipSuccess) he = hipMalloc(&c->d_scalar, 2 * sizeof(double));
    if (he != hipSuccess) { aesgcm_comm_destroy(c); return hip_fail(he, "comm scratch"); }
    *out = c;
    return AESGCM_OK;
}

int aesgcm_comm_ranks(const aesgcm_comm *c, int *n_ranks, int *rank) {
    if (!c) return AESGCM_EARG;
    if (n_ranks) *n_ranks = c->n_ranks;
    if (rank) *rank = c->rank;
    return AESGCM_OK;
}

int aesgcm_comm_allgather_dev(aesgcm_comm *c, const void *d_send, void *d_recv, size_t bytes_per_rank, void *stream) {
    if (!c || !d_send || !d_recv || !bytes_per_rank) return AESGCM_EARG;
    HCHK(hipSetDevice(c->device));
    NCHK(g_rccl.AllGather(d_send, d_recv, bytes_per_rank, ncclUint8, c->comm, stream ? (hipStream_t)stream : c->stream));
    return AESGCM_OK;
}

// all-reduce of ONE host double over the ranks (op 0 = max, 1 = min, 2 = sum); synchronous.  bench.py's barrier and
// max-over-ranks timing.
int aesgcm_comm_allreduce_f64(aesgcm_comm *c, double *value, int op) {
    if (!c || !value || op < 0 || op > 2) return AESGCM_EARG;
    HCHK(hipSetDevice(c->device));
    HCHK(hipMemcpyAsync(c->d_scalar, value, sizeof(double), hipMemcpyHostToDevice, c->stream));
    NCHK(g_rccl.AllReduce(c->d_scalar, c->d_scalar + 1, 1, ncclDouble, op == 0 ? ncclMax : op == 1 ? ncclMin : ncclSum, c->comm, c->stream));
    HCHK(hipMemcpyAsync(value, c->d_scalar + 1, sizeof(double), hipMemcpyDeviceToHost, c->stream));
    HCHK(hipStreamSynchronize(c->stream));
    return AESGCM_OK;
}

int aesgcm_comm_barrier(aesgcm_comm *c) {
    double one = 1.0;
    return aesgcm_comm_allreduce_f64(c, &one, 2);
}

int aesgcm_comm_destroy(aesgcm_comm *c) {
    if (!c) return AESGCM_OK;
    hipSetDevice(c->device);
    if (c->stream) hipStreamSynchronize(c->stream);
    if (c->comm && g_rccl.ok) g_rccl.CommDestroy(c->comm);
    if (c->d_scalar) hipFree(c->d_scalar);
    if (c->stream) hipStreamDestroy(c->stream);
    delete c;
    return AESGCM_OK;
}

// ---------------------------------------------------------------- one process, ndev GPUs
int aesgcm_mgpu_create(aesgcm_mgpu **out, int ndev, const int *devices, const uint8_t *key, size_t key_len) {
    if (!out || ndev < 1 || !devices || !key) return AESGCM_EARG;
    *out = nullptr;
    if (key_len != 16 && key_len != 24 && key_len != 32) return AESGCM_EKEYLEN;
    Rccl *r; int rc = rccl_get(&r); if (rc) return rc;
    aesgcm_mgpu *m = new (std::nothrow) aesgcm_mgpu();
    if (!m) return AESGCM_ENOMEM;
    m->ndev = ndev;
    m->dev.assign(devices, devices + ndev);
    m->ctx.assign(ndev, nullptr); m->comm.assign(ndev, nullptr); m->st.assign(ndev, nullptr);
    m->part.assign(ndev, nullptr); m->all.assign(ndev, nullptr);
    for (int g = 0; g < ndev; g++) {
        // every device derives round keys, H and its tables locally from the key: nothing is broadcast
        if ((rc = aesgcm_ctx_create(&m->ctx[g], devices[g], key, key_len))) { aesgcm_mgpu_destroy(m); return rc; }
        hipError_t e = hipSetDevice(devices[g]);
        if (e == hipSuccess) e = hipStreamCreate(&m->st[g]);     // a blocking stream, as a context's own: ordered behind work the caller left on the NULL stream
        if (e == hipSuccess) e = hipMalloc(&m->part[g], 16);
        if (e == hipSuccess) e = hipMalloc(&m->all[g], (size_t)16 * ndev * (g == 0 ? MGPU_RING : 1));
        if (e != hipSuccess) { aesgcm_mgpu_destroy(m); return hip_fail(e, "mgpu buffers"); }
    }
    ncclResult_t e = r->CommInitAll(m->comm.data(), ndev, devices);
    if (e != ncclSuccess) { for (auto &c : m->comm) c = nullptr; aesgcm_mgpu_destroy(m); return nccl_fail(e, "ncclCommInitAll"); }
    *out = m;
    return AESGCM_OK;
}

int aesgcm_mgpu_ranks(const aesgcm_mgpu *m, int *n_ranks) {
    if (!m || !n_ranks) return AESGCM_EARG;
    int n = 0;
    NCHK(g_rccl.CommCount(m->comm[0], &n));      // the size RCCL reports for the communicator
    *n_ranks = n;
    return AESGCM_OK;
}

// device g's context, borrowed (the bench's kernel timing and launch geometry; it stays the mgpu object's)
int aesgcm_mgpu_ctx(aesgcm_mgpu *m, int g, aesgcm_ctx **out) {
    if (!m || !out || g < 0 || g >= m->ndev) return AESGCM_EARG;
    *out = m->ctx[g];
    return AESGCM_OK;
}

// the tags of the OLDEST n queued messages, in the order they were queued, finalized in one launch per contiguous run of ring rows (aesgcm_shard_finalize_batch_dev) on
// device 0.  The queue is a FIFO: the window starts at the oldest message still waiting (seq - pending) and what is collected leaves from that end, so a partial
// collect -- pending 5, last_tags(2), then last_tags(3) -- returns messages {0, 1} and then {2, 3, 4}, and the `pending >= MGPU_RING` gate of aesgcm_mgpu_crypt_dev
// protects exactly the ring rows that have not been read.  (Round 5 took the NEWEST n and still counted the oldest as collected: the advisor's finding.)
int aesgcm_mgpu_last_tags(aesgcm_mgpu *m, size_t n, uint8_t *tags) {
    if (!m || !tags || n > (size_t)m->pending) return AESGCM_EARG;
    if (!n) return AESGCM_OK;
    HCHK(hipSetDevice(m->dev[0]));
    const unsigned long long first = m->seq - (unsigned long long)m->pending;
    size_t done = 0;
    while (done < n) {
        const size_t slot = (size_t)((first + done) % MGPU_RING);
        const size_t run = std::min(n - done, (size_t)MGPU_RING - slot);
        uint8_t ivs[MGPU_RING * 12];
        size_t al[MGPU_RING];
        uint64_t tl[MGPU_RING];
        for (size_t k = 0; k < run; k++) { memcpy(ivs + 12 * k, m->iv[slot + k], 12); al[k] = m->aad_len[slot + k]; tl[k] = m->total[slot + k]; }
        const int rc = aesgcm_shard_finalize_batch_dev(m->ctx[0], run, ivs, m->all[0] + slot * 16 * (size_t)m->ndev, (size_t)m->ndev, 16, (size_t)16 * m->ndev, al, tl, tags + 16 * done, m->st[0]);
        if (rc) return rc;
        done += run;
    }
    m->pending -= (int)n;
    return AESGCM_OK;
}

// tag != NULL: the message's tag, all streams synchronised on return.  tag == NULL: the call only enqueues -- every device's shard, the one grouped all-gather -- and
// the message waits in a ring of MGPU_RING rows on device 0 for aesgcm_mgpu_last_tags; no host synchronisation anywhere (what a caller that keeps several messages
// in flight wants: bench.py --single-process queues its four)
int aesgcm_mgpu_crypt_dev(aesgcm_mgpu *m, int decrypt, const uint8_t iv[12], const void *d_aad_on_dev0, size_t aad_len,
                          const void *const *d_in, const size_t *shard_len, void *const *d_out, uint8_t tag[16]) {
    if (!m || !iv || !d_in || !shard_len || !d_out || (aad_len && !d_aad_on_dev0)) return AESGCM_EARG;
    // a call that wants its tag back while older messages still wait for aesgcm_mgpu_last_tags would have to jump the queue: refused before anything is enqueued
    if (tag && m->pending > 0) { snprintf(g_cerr, sizeof g_cerr, "%d messages queued with tag = NULL: collect their tags (aesgcm_mgpu_last_tags) before a call that returns its own", m->pending); return AESGCM_ESTATE; }
    if (m->pending >= MGPU_RING) { snprintf(g_cerr, sizeof g_cerr, "%d messages queued: collect their tags first (aesgcm_mgpu_last_tags)", m->pending); return AESGCM_ESTATE; }
    uint64_t total = 0;
    for (int g = 0; g < m->ndev; g++) {
        if (g + 1 < m->ndev && (shard_len[g] & 15)) return AESGCM_EARG;      // shards are cut at 16-byte block boundaries; only the last may be ragged
        total += shard_len[g];
    }
    int rc;
    uint64_t first = 0;
    for (int g = 0; g < m->ndev; g++) {
        rc = aesgcm_shard_crypt_dev(m->ctx[g], decrypt, iv, g == 0 ? d_aad_on_dev0 : nullptr, g == 0 ? aad_len : 0,
                                    d_in[g], shard_len[g], d_out[g], first, total, m->part[g], m->st[g]);
        if (rc) return rc;
        first += shard_len[g] / 16;
    }
    const size_t slot = (size_t)(m->seq % MGPU_RING);
    // the ONE exchange of the path: 16 bytes per device
    NCHK(g_rccl.GroupStart());
    for (int g = 0; g < m->ndev; g++) {
        HCHK(hipSetDevice(m->dev[g]));
        ncclResult_t e = g_rccl.AllGather(m->part[g], g == 0 ? m->all[0] + slot * 16 * (size_t)m->ndev : m->all[g], 16, ncclUint8, m->comm[g], m->st[g]);
        if (e != ncclSuccess) { g_rccl.GroupEnd(); return nccl_fail(e, "ncclAllGather"); }
    }
    NCHK(g_rccl.GroupEnd());
    memcpy(m->iv[slot], iv, 12); m->aad_len[slot] = aad_len; m->total[slot] = total;
    ++m->seq; ++m->pending;
    if (!tag) return AESGCM_OK;
    if ((rc = aesgcm_mgpu_last_tags(m, 1, tag))) return rc;                   // waits for stream 0
    for (int g = 1; g < m->ndev; g++) { HCHK(hipSetDevice(m->dev[g])); HCHK(hipStreamSynchronize(m->st[g])); }
    HCHK(hipSetDevice(m->dev[0]));
    return AESGCM_OK;
}
// every stream of the object drained: what a caller of the queued form does before it reads the outputs on the host side
int aesgcm_mgpu_sync(aesgcm_mgpu *m) {
    if (!m) return AESGCM_EARG;
    for (int g = 0; g < m->ndev; g++) { HCHK(hipSetDevice(m->dev[g])); HCHK(hipStreamSynchronize(m->st[g])); }
    return AESGCM_OK;
}

int aesgcm_mgpu_destroy(aesgcm_mgpu *m) {
    if (!m) return AESGCM_OK;
    for (int g = 0; g < m->ndev; g++) {
        hipSetDevice(m->dev[g]);
        if (m->st[g]) hipStreamSynchronize(m->st[g]);
        if (m->comm[g] && g_rccl.ok) g_rccl.CommDestroy(m->comm[g]);
        if (m->part[g]) hipFree(m->part[g]);
        if (m->all[g]) hipFree(m->all[g]);
        if (m->st[g]) hipStreamDestroy(m->st[g]);
        if (m->ctx[g]) aesgcm_ctx_destroy(m->ctx[g]);
    }
    delete m;
    return AESGCM_OK;
}

}  // extern "C"
