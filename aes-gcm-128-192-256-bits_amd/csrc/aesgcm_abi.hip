// aesgcm_abi.hip -- the C ABI of include/aesgcm.h: every exported entry point of libaesgcm_hip.so except the inter-GPU exchange (aesgcm_comm.hip).  Argument checks,
// the order of calls into the host runtime (aesgcm_host.hip) and the launch of the packet kernels' shapes; no device code (aesgcm_internal.h).
#include "aesgcm_internal.h"

#include <algorithm>
#include <new>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

extern "C" {

int aesgcm_abi_version(void) { return AESGCM_ABI_VERSION; }


const char *aesgcm_strerror(int code) {
    switch (code) {
    case AESGCM_OK: return "ok";
    case AESGCM_EARG: return "invalid argument";
    case AESGCM_EKEYLEN: return "key length must be 16, 24 or 32 bytes";
    case AESGCM_EIVLEN: return "IV must be 12 bytes";
    case AESGCM_ETOOLONG: return "message exceeds the GCM counter space (2^36 - 32 bytes)";
    case AESGCM_EAUTH: return "authentication tag mismatch";
    case AESGCM_EHIP: return "HIP runtime error (see aesgcm_last_error)";
    case AESGCM_ENOMEM: return "out of device memory";
    case AESGCM_ESTATE: return "streaming call out of order";
    case AESGCM_EALIGN: return "device data pointer must be 16-byte aligned";
    case AESGCM_ERCCL: return "RCCL unavailable or a collective failed (see aesgcm_comm_last_error)";
    default: return "unknown error";
    }
}

const char *aesgcm_last_error(void) { return g_err; }


int aesgcm_device_count(int *n) {
    if (!n) return AESGCM_EARG;
    HIPCHK(hipGetDeviceCount(n));
    return AESGCM_OK;
}

int aesgcm_device_name(int device, char *buf, size_t buflen) {
    if (!buf || !buflen) return AESGCM_EARG;
    hipDeviceProp_t prop;
    HIPCHK(hipGetDeviceProperties(&prop, device));
    snprintf(buf, buflen, "%s %s (%d CUs)", prop.name, prop.gcnArchName, prop.multiProcessorCount);
    return AESGCM_OK;
}


int aesgcm_ctx_create(aesgcm_ctx **out, int device, const uint8_t *key, size_t key_len) {
    if (key_len != 16 && key_len != 24 && key_len != 32) return AESGCM_EKEYLEN;
    return ctx_create_common(out, device, key, key_len, 0);
}

int aesgcm_ctx_create_preexpanded(aesgcm_ctx **out, int device, const uint8_t *rk, int nr) {
    if (nr != 10 && nr != 12 && nr != 14) return AESGCM_EKEYLEN;
    return ctx_create_common(out, device, rk, (size_t)(4 * (nr - 6)), nr);
}

// A new key for an existing context (the reference core's "load key" between frames, tb/gcm_gctr.py:144-175; H is recomputed only then, src/gcm_gctr.vhd:142-144):
// everything the context owns stays -- stream, scratch, host slot, options -- only the key material is rebuilt.  Waits for the context's queued work first.
int aesgcm_ctx_rekey(aesgcm_ctx *c, const uint8_t *key, size_t key_len) {
    if (!c || !key) return AESGCM_EARG;
    if (key_len != 16 && key_len != 24 && key_len != 32) return AESGCM_EKEYLEN;
    if (c->s.active) return AESGCM_ESTATE;
    HIPCHK(hipSetDevice(c->device));
    // every *_dev entry point takes a caller's stream, so work that reads this context's key material may be queued on any stream of the device: wait for them all
    // (round 4 waited for the context's own stream only -- a message in flight on another stream would have read half-rebuilt tables)
    HIPCHK(hipDeviceSynchronize());
    return ctx_load_key(c, key, key_len, 0);
}

int aesgcm_ctx_destroy(aesgcm_ctx *c) {
    if (!c) return AESGCM_OK;
    { std::lock_guard<std::mutex> lk(g_mu); g_ctxs.erase(std::remove(g_ctxs.begin(), g_ctxs.end(), c), g_ctxs.end()); }
    hipSetDevice(c->device);
    if (c->stream) hipStreamSynchronize(c->stream);
    for (auto &e : c->ev) { hipEventDestroy(e.first); hipEventDestroy(e.second); }
    for (auto &e : c->ev_pool) { hipEventDestroy(e.first); hipEventDestroy(e.second); }
    if (c->km) { hipMemset(c->km, 0, sizeof(KeyMaterial)); hipFree(c->km); }
    if (c->d_keystage) hipFree(c->d_keystage);
    if (c->parts) hipFree(c->parts);
    if (c->fold_a) hipFree(c->fold_a);
    if (c->fold_b) hipFree(c->fold_b);
    if (c->d_counter) hipFree(c->d_counter);
    if (c->d_cyc) hipFree(c->d_cyc);
    if (c->d_tag) hipFree(c->d_tag);
    if (c->h_tag) hipHostFree(c->h_tag);
    if (c->h_mtag) hipHostFree(c->h_mtag);
    if (c->d_mtag) hipFree(c->d_mtag);
    if (c->d_trace) hipFree(c->d_trace);
    if (c->rows_buf) hipFree(c->rows_buf);
    if (c->side) { hipStreamSynchronize(c->side); hipStreamDestroy(c->side); }
    if (c->ev_fork) hipEventDestroy(c->ev_fork);
    if (c->ev_join) hipEventDestroy(c->ev_join);
    pipeline_release(c);
    if (c->st_in) hipFree(c->st_in);
    if (c->st_out) hipFree(c->st_out);
    if (c->st_aad) hipFree(c->st_aad);
    if (c->ev_sync) hipEventDestroy(c->ev_sync);
    if (c->ev_fused) hipEventDestroy(c->ev_fused);
    if (c->stream) {                                            // idle by now (synchronised above): kept for the device's next context, up to 64 of them
        std::lock_guard<std::mutex> lk(g_mu);
        if (c->device >= 0 && c->device < (int)g_dev.size() && g_dev[c->device].streams.size() < 64) { g_dev[c->device].streams.push_back(c->stream); c->stream = nullptr; }
    }
    if (c->stream) hipStreamDestroy(c->stream);
    delete c;
    return AESGCM_OK;
}

int aesgcm_ctx_device(const aesgcm_ctx *c) { return c ? c->device : AESGCM_EARG; }

// which launch structure the context's last whole-message call took (AESGCM_LAUNCH_*): the choice between the full and the half shape of the cyclic rows depends on
// what other contexts had under way at the moment of the call, so benches and profiles ask instead of assuming
int aesgcm_ctx_last_launch(const aesgcm_ctx *c, int *shape) {
    if (!c || !shape) return AESGCM_EARG;
    *shape = c->last_shape;
    return AESGCM_OK;
}

// Tunables of ONE context, for tests and profiling scripts (the defaults are the measured best; nothing in the library reads the environment).
// Every value selects between paths that produce the same bytes; the parity tests use them to reach each path at sizes a CPU check finishes in seconds.
int aesgcm_ctx_set_option(aesgcm_ctx *c, const char *key, int64_t value) {
    if (!c || !key || value < 0) return AESGCM_EARG;
    const u64 v = (u64)value;
    if (!strcmp(key, "tw")) c->tw_override = (u32)v;                                   // rows per chunk of the dealt kernels (0 = the library's rule)
    else if (!strcmp(key, "body_min")) {                                              // bytes from which a range's aligned middle goes through k_body
        c->body_min = v;
        if (v >= (1ull << 60)) c->cyc_max = c->cyc_max_pieces = c->cyc_max_fused = 0;  // "never k_body" means the cyclic rows too
    }
    else if (!strcmp(key, "cyc_min")) c->cyc_min = c->cyc_min_fused = v;               // bytes: ranges in [cyc_min, cyc_max) take k_body's cyclic rows; both 0 = never
    else if (!strcmp(key, "cyc_max")) c->cyc_max = c->cyc_max_pieces = c->cyc_max_fused = v;
    else if (!strcmp(key, "cyc_half")) { if (v > 2) return AESGCM_EARG; c->cyc_half = (int)v; }   // whole messages below 80 MiB as k_bodyh (two workgroups per CU): 0 never, 1 always, 2 when another context has a message under way
    else if (!strcmp(key, "cyc_close")) c->cyc_fuse = v != 0;                          // 1: a whole message's cyclic launch closes the tag itself; 0: k_fold + k_combine behind it
    else if (!strcmp(key, "fold_close")) c->fold_close = v != 0;                       // 1: behind the dealt k_body the first (or second) k_fold level closes the tag
    else if (!strcmp(key, "cyc_prio")) c->cyc_prio = (u32)v;                           // rows between rotations of the waves' issue priorities in a cyclic launch (0 = off)
    else if (!strcmp(key, "pkt_order")) { }                                           // accepted and ignored since round 6: with offset arrays the packet kernels always take their messages by falling size class (the routing sort makes the order anyway)
    else if (!strcmp(key, "wipe_on_auth_fail")) c->wipe_on_auth_fail = v != 0;         // decrypt with verification: zero the output of a message / packet whose tag does not match
    else if (!strcmp(key, "rows_min")) c->rows_min = v;                                // bytes per packet from which aesgcm_packets_crypt_dev goes by rows (k_rows); 0 = never
    else if (!strcmp(key, "route_mid_min")) c->route_mid_min = v > 0xFFFFFFFFull ? 0xFFFFFFFFu : (u32)v;   // routed calls: messages between the marks from which the high mark applies
    else if (!strcmp(key, "route_blocks_min")) c->route_blocks_min = v;                  // routed calls: blocks of short messages below which nothing goes to the packet kernels
    else if (!strcmp(key, "route_top_min")) c->route_top_min = v > 0xFFFFFFFFull ? 0xFFFFFFFFu : (u32)v;   // routed calls: messages between rows_min and 16 320 bytes from which the packet kernels take them too
    else if (!strcmp(key, "rows_block")) c->rows_block = (u32)v;                       // units (rows, tails) per dealt block of k_rows; 0 = the library's cut
    else if (!strcmp(key, "poll_us")) c->poll_ns = 1000L * (long)v;                    // how long a tag is polled for in the host slot before the call blocks in the runtime
    else return AESGCM_EARG;
    return AESGCM_OK;
}

int aesgcm_ctx_stream(const aesgcm_ctx *c, void **stream) {
    if (!c || !stream) return AESGCM_EARG;
    *stream = (void *)c->stream;
    return AESGCM_OK;
}

// Everything enqueued from now on on `c`'s own stream starts only after everything enqueued so far on `other`'s own stream
// has completed (one event record + one stream wait; no host synchronisation).  Two contexts of one key on one device
// have separate scratch sets and streams, so consecutive messages can alternate between them and message m+1's fused
// kernel starts while message m's k_fold / k_combine drain; this call orders the step that needs both (the all-gather).
// the context's status word (pinned host memory, h_tag[2]: {code, 0, detail lo, detail hi}; written by k_rows_plan* when it refuses a call): as an error code, without clearing it
static int status_pending(const aesgcm_ctx *c) {
    const u32 code = __atomic_load_n(reinterpret_cast<const u32 *>(c->h_tag + 2), __ATOMIC_ACQUIRE);
    if (code == AESGCM_STATUS_OK) return AESGCM_OK;
    snprintf(g_err, sizeof g_err, "an asynchronous call on this context was refused on the device (status %u; aesgcm_ctx_status has the detail)", code);
    return code == AESGCM_STATUS_PLAN ? AESGCM_ESTATE : AESGCM_ETOOLONG;
}
int aesgcm_ctx_status(aesgcm_ctx *c, int *code, uint64_t *detail) {
    if (!c || !code) return AESGCM_EARG;
    volatile u32 *w = reinterpret_cast<volatile u32 *>(c->h_tag + 2);
    const u32 st = __atomic_load_n(reinterpret_cast<const u32 *>(c->h_tag + 2), __ATOMIC_ACQUIRE);
    *code = (int)st;
    if (detail) *detail = st ? ((u64)w[3] << 32) | w[2] : 0;
    if (st) { w[2] = 0; w[3] = 0; __atomic_store_n(reinterpret_cast<u32 *>(c->h_tag + 2), 0u, __ATOMIC_RELEASE); }
    return AESGCM_OK;
}

int aesgcm_ctx_wait(aesgcm_ctx *c, aesgcm_ctx *other) {
    if (!c || !other) return AESGCM_EARG;
    { const int rc = status_pending(other); if (rc) return rc; }            // what `other` was asked to do and refused (aesgcm_ctx_status)
    { const int rc = status_pending(c); if (rc) return rc; }
    if (c == other) return AESGCM_OK;
    if (c->device != other->device) return AESGCM_EARG;
    HIPCHK(hipSetDevice(c->device));
    if (!other->ev_sync) HIPCHK(hipEventCreateWithFlags(&other->ev_sync, hipEventDisableTiming));
    HIPCHK(hipEventRecord(other->ev_sync, other->stream));
    HIPCHK(hipStreamWaitEvent(c->stream, other->ev_sync, 0));
    return AESGCM_OK;
}

// As aesgcm_ctx_wait, but only up to `other`'s most recently enqueued FUSED kernel (k_body / k_main), not its fold / combine
// tail: message m+1's fused kernel (on `c`) then follows message m's (on `other`) back to back, and m's k_fold / k_combine
// launches run beside it.  (Two contexts that simply start together share the CUs -- k_body is one 141 KiB workgroup per CU --
// and finish together: that hides one tail in two; chained, all tails but the last hide.)  The event is recorded from the
// first call on; a wait issued before `other` has launched anything is a no-op.
int aesgcm_ctx_wait_fused(aesgcm_ctx *c, aesgcm_ctx *other) {
    if (!c || !other) return AESGCM_EARG;
    if (c == other) return AESGCM_OK;
    if (c->device != other->device) return AESGCM_EARG;
    HIPCHK(hipSetDevice(c->device));
    if (!other->ev_fused) { HIPCHK(hipEventCreateWithFlags(&other->ev_fused, hipEventDisableTiming)); return AESGCM_OK; }
    HIPCHK(hipStreamWaitEvent(c->stream, other->ev_fused, 0));
    return AESGCM_OK;
}

int aesgcm_ctx_geometry(const aesgcm_ctx *c, int *n_wg, int *wg_lanes, int *lds_bytes) {
    if (!c) return AESGCM_EARG;
    if (n_wg) *n_wg = c->G;
    if (wg_lanes) *wg_lanes = AESGCM_MAIN_WG;
    if (lds_bytes) *lds_bytes = AESGCM_LDS_BYTES;
    return AESGCM_OK;
}


int aesgcm_ctx_body_geometry(const aesgcm_ctx *c, int *n_wg, int *wg_lanes, int *lds_bytes) {
    if (!c) return AESGCM_EARG;
#if AESGCM_T4
    if (n_wg) *n_wg = c->G / 2;
#else
    if (n_wg) *n_wg = c->G;
#endif
    if (wg_lanes) *wg_lanes = AESGCM_BODY_WG;
    if (lds_bytes) *lds_bytes = AESGCM_BODY_LDS;
    return AESGCM_OK;
}


int aesgcm_ctx_split(const aesgcm_ctx *c, size_t len, uint64_t first_block, uint64_t *head_blocks, uint64_t *body_blocks) {
    if (!c) return AESGCM_EARG;
    if (cyc_capable(c)) {                                                  // cyclic rows: the body is every whole row behind the head
        const u64 nfull = len / 16, head = (256 - (first_block & 255)) & 255;
        const u64 R = nfull > head ? (nfull - head) / 64 : 0;
        if (R && R * 1024 >= c->cyc_min && R * 1024 < (((first_block & 255) || (len & 1023)) ? c->cyc_max_pieces : c->cyc_max)) {
            if (head_blocks) *head_blocks = head;
            if (body_blocks) *body_blocks = 64 * R;
            return AESGCM_OK;
        }
    }
    BodySplit b;
    const bool split = ctx_body_split(c, len, first_block, &b);
    if (head_blocks) *head_blocks = split ? b.head_blocks : 0;
    if (body_blocks) *body_blocks = split ? b.body_blocks : 0;
    return AESGCM_OK;
}


// ---------------------------------------------------------------- unit-level
int aesgcm_key_expand(int device, const uint8_t *key, size_t key_len, uint8_t rk[240], int *nr) {
    if (!key || !rk) return AESGCM_EARG;
    if (key_len != 16 && key_len != 24 && key_len != 32) return AESGCM_EKEYLEN;
    aesgcm_ctx *c = nullptr;
    int rc = aesgcm_ctx_create(&c, device, key, key_len);
    if (rc) return rc;
    hipError_t e = hipMemcpy(rk, c->km->rk_bytes, (size_t)16 * (c->nr + 1), hipMemcpyDeviceToHost);
    if (nr) *nr = c->nr;
    aesgcm_ctx_destroy(c);
    if (e != hipSuccess) return hip_fail(e, "hipMemcpy");
    return AESGCM_OK;
}


int aesgcm_get_h(aesgcm_ctx *c, uint8_t h[16]) {
    if (!c || !h) return AESGCM_EARG;
    HIPCHK(hipSetDevice(c->device));
    HIPCHK(hipMemcpy(h, &c->km->h, 16, hipMemcpyDeviceToHost));
    return AESGCM_OK;
}


int aesgcm_gfmul(int device, const uint8_t *h, const uint8_t *x, uint8_t *z, size_t n) {
    if (!h || !x || !z) return AESGCM_EARG;
    if (!n) return AESGCM_OK;
    DeviceState *ds;
    int rc = device_state(device, &ds);
    if (rc) return rc;
    HIPCHK(hipSetDevice(device));
    uint4 *d = nullptr;
    HIPCHK(hipMalloc(&d, 48 * n));
    hipError_t e = hipMemcpy(d, h, 16 * n, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(d + n, x, 16 * n, hipMemcpyHostToDevice);
    if (e == hipSuccess) {
        e = klaunch_gfmul(d, d + n, d + 2 * n, n);
    }
    if (e == hipSuccess) e = hipMemcpy(z, d + 2 * n, 16 * n, hipMemcpyDeviceToHost);
    hipFree(d);
    if (e != hipSuccess) return hip_fail(e, "aesgcm_gfmul");
    return AESGCM_OK;
}


int aesgcm_encrypt_dev(aesgcm_ctx *c, const uint8_t iv[12], const void *d_aad, size_t aad_len,
                       const void *d_pt, size_t len, void *d_ct, uint8_t tag[16], void *stream) {
    if (!c || !iv) return AESGCM_EARG;
    hipStream_t st = pick_stream(c, stream);
    int rc = crypt_dev(c, 0, iv, d_aad, aad_len, d_pt, len, d_ct, st);
    if (rc) return rc;
    if (tag) return fetch_tag(c, st, tag);
    return AESGCM_OK;
}

int aesgcm_decrypt_dev(aesgcm_ctx *c, const uint8_t iv[12], const void *d_aad, size_t aad_len,
                       const void *d_ct, size_t len, void *d_pt, const uint8_t *expect_tag, uint8_t tag_out[16], void *stream) {
    if (!c || !iv) return AESGCM_EARG;
    hipStream_t st = pick_stream(c, stream);
    int rc = crypt_dev(c, 1, iv, d_aad, aad_len, d_ct, len, d_pt, st);
    if (rc) return rc;
    if (tag_out || expect_tag) {
        uint8_t t[16];
        if ((rc = fetch_tag(c, st, t))) return rc;
        if (tag_out) memcpy(tag_out, t, 16);
        if (expect_tag && !ct_compare16(t, expect_tag)) {
            if (c->wipe_on_auth_fail && len) { HIPCHK(hipMemsetAsync(d_pt, 0, len, st)); HIPCHK(hipStreamSynchronize(st)); }   // nothing unauthenticated is left in the caller's buffer
            return AESGCM_EAUTH;
        }
    }
    return AESGCM_OK;
}

// the tag of the message most recently enqueued with tag = NULL (aesgcm_encrypt_dev / aesgcm_decrypt_dev): through the host slot, as if the call had asked for it
int aesgcm_last_tag(aesgcm_ctx *c, uint8_t tag[16], void *stream) {
    if (!c || !tag) return AESGCM_EARG;
    hipStream_t st = pick_stream(c, stream);
    HIPCHK(hipSetDevice(c->device));
    { const int rc = status_pending(c); if (rc) return rc; }                // an asynchronous call on this context was refused on the device: say so rather than hand out a tag from before it
    return fetch_tag(c, st, tag);
}


int aesgcm_encrypt(aesgcm_ctx *c, const uint8_t iv[12], const uint8_t *aad, size_t aad_len,
                   const uint8_t *pt, size_t len, uint8_t *ct, uint8_t tag[16]) {
    if (!c || !iv || !tag || (aad_len && !aad) || (len && (!pt || !ct))) return AESGCM_EARG;
    int rc = check_lengths(aad_len, len);
    if (rc) return rc;
    HIPCHK(hipSetDevice(c->device));
    if ((rc = stage_in(c, aad, aad_len, pt, len))) return rc;
    if ((rc = crypt_dev(c, 0, iv, c->st_aad, aad_len, c->st_in, len, c->st_out, c->stream))) return rc;
    // the call returns data synchronously (tb/gcm_model.py:26): the tag's generation number is published by k_combine BEFORE
    // this copy starts, and with a page-locked `ct` (aesgcm_host_alloc) the copy is truly asynchronous -- wait for it.  With
    // len == 0 nothing is copied and the tag alone is polled for.
    if (len) { HIPCHK(hipMemcpyAsync(ct, c->st_out, len, hipMemcpyDeviceToHost, c->stream)); HIPCHK(hipStreamSynchronize(c->stream)); }
    return fetch_tag(c, c->stream, tag);
}

int aesgcm_decrypt(aesgcm_ctx *c, const uint8_t iv[12], const uint8_t *aad, size_t aad_len,
                   const uint8_t *ct, size_t len, uint8_t *pt, const uint8_t *expect_tag, uint8_t tag_out[16]) {
    if (!c || !iv || (aad_len && !aad) || (len && (!ct || !pt))) return AESGCM_EARG;
    int rc = check_lengths(aad_len, len);
    if (rc) return rc;
    HIPCHK(hipSetDevice(c->device));
    if ((rc = stage_in(c, aad, aad_len, ct, len))) return rc;
    if ((rc = crypt_dev(c, 1, iv, c->st_aad, aad_len, c->st_in, len, c->st_out, c->stream))) return rc;
    uint8_t t[16];
    const bool hold = expect_tag && c->wipe_on_auth_fail;        // the plaintext leaves the device only once its tag has been checked
    if (len && !hold) { HIPCHK(hipMemcpyAsync(pt, c->st_out, len, hipMemcpyDeviceToHost, c->stream)); HIPCHK(hipStreamSynchronize(c->stream)); }   // as aesgcm_encrypt: never return while `pt` is still landing
    if ((rc = fetch_tag(c, c->stream, t))) return rc;
    if (tag_out) memcpy(tag_out, t, 16);
    if (expect_tag && !ct_compare16(t, expect_tag)) {
        if (hold && len) { memset(pt, 0, len); HIPCHK(hipMemsetAsync(c->st_out, 0, len, c->stream)); HIPCHK(hipStreamSynchronize(c->stream)); }
        return AESGCM_EAUTH;
    }
    if (len && hold) { HIPCHK(hipMemcpyAsync(pt, c->st_out, len, hipMemcpyDeviceToHost, c->stream)); HIPCHK(hipStreamSynchronize(c->stream)); }
    return AESGCM_OK;
}


int aesgcm_ecb_encrypt(aesgcm_ctx *c, const uint8_t *in, size_t nblocks, uint8_t *out) {
    if (!c || (nblocks && (!in || !out))) return AESGCM_EARG;
    if (!nblocks) return AESGCM_OK;
    HIPCHK(hipSetDevice(c->device));
    int rc;
    if ((rc = stage_in(c, nullptr, 0, in, 16 * nblocks))) return rc;
    if ((rc = enqueue_main(c, MODE_ECB, nullptr, nullptr, 0, c->st_in, 16 * (u64)nblocks, c->st_out, 0, c->stream, nullptr))) return rc;
    HIPCHK(hipMemcpyAsync(out, c->st_out, 16 * nblocks, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    return AESGCM_OK;
}


int aesgcm_keystream_dev(aesgcm_ctx *c, const uint8_t iv[12], uint64_t first_block, uint64_t nblocks, void *d_out, void *stream) {
    if (!c || !iv || (nblocks && !d_out)) return AESGCM_EARG;
    if (first_block + nblocks > (((u64)1) << 32) - 2) return AESGCM_ETOOLONG;
    if (!nblocks) return AESGCM_OK;
    HIPCHK(hipSetDevice(c->device));
    return enqueue_main(c, MODE_KS, iv, nullptr, 0, d_out /*unused in*/, 16 * nblocks, d_out, first_block, pick_stream(c, stream), nullptr);
}

int aesgcm_keystream(aesgcm_ctx *c, const uint8_t iv[12], uint64_t first_block, uint64_t nblocks, uint8_t *out) {
    if (!c || !iv || (nblocks && !out)) return AESGCM_EARG;
    if (!nblocks) return AESGCM_OK;
    HIPCHK(hipSetDevice(c->device));
    int rc;
    if ((rc = grow(&c->st_out, &c->st_out_cap, 16 * nblocks))) return rc;
    if ((rc = aesgcm_keystream_dev(c, iv, first_block, nblocks, c->st_out, nullptr))) return rc;
    HIPCHK(hipMemcpyAsync(out, c->st_out, 16 * nblocks, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    return AESGCM_OK;
}


int aesgcm_ghash(aesgcm_ctx *c, const uint8_t *data, size_t len, uint8_t y[16]) {
    if (!c || !y || (len && !data)) return AESGCM_EARG;
    if ((len + 15) / 16 >= MAX_SEQ_BLOCKS) return AESGCM_ETOOLONG;
    HIPCHK(hipSetDevice(c->device));
    int rc;
    if ((rc = stage_in(c, data, len, nullptr, 0))) return rc;
    Partials pp;
    uint8_t iv0[12] = {0};
    // the data rides in the AAD slot of the GHASH sequence (GHASH only, no AES)
    if ((rc = enqueue_main(c, MODE_ENC, iv0, c->st_aad, len, c->st_in, 0, c->st_out, 0, c->stream, &pp))) return rc;
    if ((rc = enqueue_combine(c, combine_with_items(plan_combine_poly(pp.ptr, pp.np, pp.kind, 1, c->d_tag), pp.eA), c->stream))) return rc;   // Y = P * H
    HIPCHK(hipMemcpyAsync(y, c->d_tag, 16, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    return AESGCM_OK;
}


// ---------------------------------------------------------------- shards
int aesgcm_shard_crypt_dev(aesgcm_ctx *c, int decrypt, const uint8_t iv[12], const void *d_aad, size_t aad_len,
                           const void *d_in, size_t len, void *d_out, uint64_t first_block, uint64_t total_len,
                           void *d_partial, void *stream) {
    if (!c || !iv || !d_partial || (len && (!d_in || !d_out))) return AESGCM_EARG;
    int rc = check_lengths(first_block == 0 ? aad_len : 0, total_len);
    if (rc) return rc;
    const u64 total_blocks = (total_len + 15) / 16;
    const u64 my_blocks = ((u64)len + 15) / 16;
    if (first_block + my_blocks > total_blocks) return AESGCM_EARG;
    if ((len & 15) && first_block + my_blocks != total_blocks) return AESGCM_EARG;   // only the last shard may be ragged
    if (first_block != 0 && aad_len) return AESGCM_EARG;
    HIPCHK(hipSetDevice(c->device));
    hipStream_t st = pick_stream(c, stream);
    const u64 after = total_blocks - (first_block + my_blocks);            // blocks of the message behind this shard
    {   // mid-size shards: one k_body launch of cyclic rows, its items straight to the weighted partial W = P H^after
        Partials pc;
        bool took;
        if ((rc = enqueue_cyc(c, decrypt ? MODE_DEC : MODE_ENC, iv, d_aad, aad_len, d_in, len, d_out, first_block, st, &pc, &took))) return rc;
        if (took) return enqueue_combine(c, combine_with_items(plan_combine_poly(pc.ptr, pc.np, pc.kind, after, (uint4 *)d_partial), pc.eA, pc.tail_item, pc.tail_blocks), st);
    }
    BodySplit b;
    if (ctx_body_split(c, len, first_block, &b)) {
        if (!aad_len && !b.head_blocks && len == 16 * b.body_blocks) {
            // the shard is one aligned body (the 8-GPU job's shape: 4 GiB at a multiple of 256 blocks): its items go straight to the
            // weighted partial W = P H^after -- no chaining value, one k_combine instead of memset + carry combine + weighting combine
            Partials pb;
            if ((rc = enqueue_body(c, decrypt ? MODE_DEC : MODE_ENC, iv, b, d_in, d_out, first_block, st, &pb))) return rc;
            return enqueue_combine(c, combine_with_items(plan_combine_poly(pb.ptr, pb.np, pb.kind, after, (uint4 *)d_partial), pb.eA), st);
        }
        uint4 *state = c->d_tag + 2;
        HIPCHK(hipMemsetAsync(state, 0, 16, st));
        if ((rc = absorb_range(c, decrypt ? MODE_DEC : MODE_ENC, iv, d_aad, aad_len, d_in, len, d_out, first_block, st, state))) return rc;
        CombineParams q = plan_combine_poly(nullptr, 0, PARTS_NONE, 0, (uint4 *)d_partial);       // W = Y * H^after
        q.carry = state; q.has_carry = 1; q.e_carry = after;
        return enqueue_combine(c, q, st);
    }
    Partials pp;
    if ((rc = enqueue_main(c, decrypt ? MODE_DEC : MODE_ENC, iv, d_aad, aad_len, d_in, len, d_out, first_block, st, &pp))) return rc;
    return enqueue_combine(c, combine_with_items(plan_combine_poly(pp.ptr, pp.np, pp.kind, after, (uint4 *)d_partial), pp.eA), st);
}

int aesgcm_shard_finalize_strided_dev(aesgcm_ctx *c, const uint8_t iv[12], const void *d_partials, size_t n_partials, size_t stride_bytes,
                                      size_t aad_len, uint64_t total_len, uint8_t tag[16], void *stream) {
    if (!c || !iv || (n_partials && !d_partials) || n_partials > AESGCM_GMAX) return AESGCM_EARG;
    if (stride_bytes < 16 || (stride_bytes & 15) || stride_bytes / 16 > 0xFFFFFFFFull) return AESGCM_EARG;
    HIPCHK(hipSetDevice(c->device));
    hipStream_t st = pick_stream(c, stream);
    CombineParams q = plan_combine_tag((const uint4 *)d_partials, (u32)n_partials, PARTS_GATHERED, iv, aad_len, total_len, c->d_tag);
    q.stride = (u32)(stride_bytes / 16);
    int rc = enqueue_combine(c, q, st);
    if (rc) return rc;
    if (tag) return fetch_tag(c, st, tag);
    return AESGCM_OK;
}

// The tags of n_msgs messages in ONE launch (one workgroup per message) and one wait: what a multi-GPU step does after its single
// all-gather.  Per message a k_combine launch costs ~15 us (E_K(IV || 1) bytewise on one lane) plus a host round trip for its tag;
// four of them were ~140 us of a 17 ms rank step (profiles/archive/r03/rank_step_trace.txt).
int aesgcm_shard_finalize_batch_dev(aesgcm_ctx *c, size_t n_msgs, const uint8_t *ivs, const void *d_partials, size_t n_partials,
                                    size_t stride_bytes, size_t msg_stride_bytes, const size_t *aad_lens, const uint64_t *total_lens,
                                    uint8_t *tags, void *stream) {
    if (!c || !ivs || !total_lens || !tags || (n_partials && !d_partials) || n_partials > AESGCM_GMAX) return AESGCM_EARG;
    if (!n_msgs) return AESGCM_OK;
    if (n_msgs > COMBINE_BATCH_MAX) return AESGCM_EARG;
    if (stride_bytes < 16 || (stride_bytes & 15) || stride_bytes / 16 > 0xFFFFFFFFull || (msg_stride_bytes & 15)) return AESGCM_EARG;
    HIPCHK(hipSetDevice(c->device));
    if (!c->h_mtag) {
        HIPCHK(hipHostMalloc((void **)&c->h_mtag, 32 * COMBINE_BATCH_MAX, hipHostMallocMapped | hipHostMallocCoherent));
        memset(c->h_mtag, 0, 32 * COMBINE_BATCH_MAX);
        HIPCHK(hipHostGetDevicePointer((void **)&c->h_mtag_dev, c->h_mtag, 0));
        HIPCHK(hipMalloc(&c->d_mtag, 16 * COMBINE_BATCH_MAX));
    }
    hipStream_t st = pick_stream(c, stream);
    CombineBatch b;
    memset(&b, 0, sizeof b);
    const u64 gen = gen_take(c);
    for (size_t m = 0; m < n_msgs; m++) {
        CombineParams q = plan_combine_tag((const uint4 *)((const unsigned char *)d_partials + m * msg_stride_bytes), (u32)n_partials, PARTS_GATHERED,
                                           ivs + 12 * m, aad_lens ? aad_lens[m] : 0, total_lens[m], c->d_mtag + m);
        q.stride = (u32)(stride_bytes / 16);
        q.out_host = c->h_mtag_dev + 2 * m; q.gen = gen;
        b.p[m] = q;
    }
    { const hipError_t le = klaunch_combine_batch((unsigned)n_msgs, st, c->km, c->tables, b); if (le != hipSuccess) { gen_give_back(c); return hip_fail(le, "k_combine_batch launch"); } }
    // every workgroup publishes its own generation word behind its tag: poll them all (short), then fall back to the stream
    struct timespec t0, t1;
    clock_gettime(CLOCK_MONOTONIC, &t0);
    bool seen = false;
    for (u32 spin = 0; !seen; ++spin) {
        seen = true;
        for (size_t m = 0; m < n_msgs; m++)
            if (__atomic_load_n(reinterpret_cast<volatile u64 *>(c->h_mtag + 2 * m + 1), __ATOMIC_ACQUIRE) != gen) { seen = false; break; }
        if (seen) break;
        if ((spin & 63u) == 63u) {
            clock_gettime(CLOCK_MONOTONIC, &t1);
            if ((t1.tv_sec - t0.tv_sec) * 1000000000L + (t1.tv_nsec - t0.tv_nsec) > c->poll_ns) break;
        }
#if defined(__x86_64__)
        __builtin_ia32_pause();
#endif
    }
    if (!seen) HIPCHK(hipStreamSynchronize(st));
    for (size_t m = 0; m < n_msgs; m++) memcpy(tags + 16 * m, c->h_mtag + 2 * m, 16);
    return AESGCM_OK;
}

int aesgcm_shard_finalize_dev(aesgcm_ctx *c, const uint8_t iv[12], const void *d_partials, size_t n_partials,
                              size_t aad_len, uint64_t total_len, uint8_t tag[16], void *stream) {
    return aesgcm_shard_finalize_strided_dev(c, iv, d_partials, n_partials, 16, aad_len, total_len, tag, stream);
}


// ---------------------------------------------------------------- streaming
// state Y (c->d_tag[1]) = polynomial of everything absorbed so far: sum X_i H^(n-1-i); the bookkeeping is aesgcm_ctx::StreamState (aesgcm_internal.h)
int aesgcm_stream_begin(aesgcm_ctx *c, const uint8_t iv[12], int decrypt) {
    if (!c || !iv) return AESGCM_EARG;
    HIPCHK(hipSetDevice(c->device));
    return stream_open(c, iv, decrypt, c->stream);
}

int aesgcm_stream_aad(aesgcm_ctx *c, const uint8_t *aad, size_t len) {
    if (!c || (len && !aad)) return AESGCM_EARG;
    if (!c->s.active || c->s.data || c->s.ragged) return AESGCM_ESTATE;
    if (!len) return AESGCM_OK;
    if (check_lengths(c->s.aad_len + len, 0)) return AESGCM_ETOOLONG;
    HIPCHK(hipSetDevice(c->device));
    int rc;
    if ((rc = stage_in(c, aad, len, nullptr, 0))) return rc;
    if ((rc = stream_absorb(c, c->st_aad, len, c->st_in, 0, c->st_out, 0))) return rc;
    c->s.aad_len += len;
    if (len & 15) c->s.ragged = true;
    HIPCHK(hipStreamSynchronize(c->stream));
    return AESGCM_OK;
}

int aesgcm_stream_update(aesgcm_ctx *c, const uint8_t *in, size_t len, uint8_t *out) {
    if (!c || (len && (!in || !out))) return AESGCM_EARG;
    if (!c->s.active) return AESGCM_ESTATE;
    if (c->s.data && c->s.ragged) return AESGCM_ESTATE;      // a ragged data chunk must be the last one
    if (!len) return AESGCM_OK;
    if (check_lengths(c->s.aad_len, c->s.len + len)) return AESGCM_ETOOLONG;
    HIPCHK(hipSetDevice(c->device));
    int rc;
    if ((rc = stage_in(c, nullptr, 0, in, len))) return rc;
    c->s.ragged = false;
    if ((rc = stream_absorb(c, nullptr, 0, c->st_in, len, c->st_out, c->s.len / 16))) return rc;
    c->s.data = true;
    c->s.len += len;
    if (len & 15) c->s.ragged = true;
    HIPCHK(hipMemcpyAsync(out, c->st_out, len, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    return AESGCM_OK;
}

// The same step on DEVICE pointers, asynchronous on `stream` (round 6): a chunk of any size -- it takes the launch structure a shard of that size takes -- of a message
// whose total length nobody knows yet.  The chunks of one session must be stream-ordered (one stream, or the caller orders them), as every call on a context.
int aesgcm_stream_update_dev(aesgcm_ctx *c, const void *d_in, size_t len, void *d_out, void *stream) {
    if (!c || (len && (!d_in || !d_out))) return AESGCM_EARG;
    if (!c->s.active) return AESGCM_ESTATE;
    if (c->s.data && c->s.ragged) return AESGCM_ESTATE;
    if (!len) return AESGCM_OK;
    if (check_lengths(c->s.aad_len, c->s.len + len)) return AESGCM_ETOOLONG;
    if (((uintptr_t)d_in | (uintptr_t)d_out) & 15) return AESGCM_EALIGN;
    HIPCHK(hipSetDevice(c->device));
    const bool was_ragged = c->s.ragged;
    c->s.ragged = false;
    const int rc = stream_absorb(c, nullptr, 0, d_in, len, d_out, c->s.len / 16, pick_stream(c, stream), true);
    if (rc) { c->s.ragged = was_ragged; return rc; }
    c->s.data = true;
    c->s.len += len;
    if (len & 15) c->s.ragged = true;
    return AESGCM_OK;
}

int aesgcm_stream_final(aesgcm_ctx *c, uint8_t tag[16]) {
    if (!c || !tag) return AESGCM_EARG;
    if (!c->s.active) return AESGCM_ESTATE;
    HIPCHK(hipSetDevice(c->device));
    int rc = enqueue_combine(c, plan_combine_final(c->d_tag + 1, c->s.iv, c->s.aad_len, c->s.len, c->d_tag), c->stream);
    if (rc) return rc;
    if ((rc = fetch_tag(c, c->stream, tag))) return rc;
    c->s.active = false;
    return AESGCM_OK;
}

// ---- the state of a message under way as 64 bytes a caller can keep, move to another context, device or process, and pick up again (SURVEY.md 5 "checkpoint /
// resume", 8(f2): the state the RTL and the pycryptodome model cannot export -- the Y register, src/gcm_ghash.vhd:174-186, and the counter, src/aes_icb.vhd:97-100).
//   [0] version 1   [1] direction (1 = decrypt)   [2] bit 0: data has begun, bit 1: the last chunk was ragged   [3] 0
//   [4, 16) IV      [16, 24) AAD bytes so far     [24, 32) data bytes so far (the next counter is 2 + this / 16)      [32, 48) Y, in the library's form (the RTL's Y / H)
//   [48, 56) GHASH blocks so far    [56, 60) key check: the first four bytes of E_K(A5 .. A5) -- NOT key material (H = E_K(0) is, and stays out)   [60, 64) sum check over [0, 60)
// No key, no H, no table.  Y itself is a secret-dependent value of the same kind as a tag before its final XOR: treat the blob like the message's tag-in-progress.
#define STREAM_BLOB_VERSION 1
static u32 blob_sum(const uint8_t *b) { u32 s = 0x5EC0DE5u; for (int i = 0; i < 60; i++) s = (s << 5 | s >> 27) ^ b[i]; return s; }
static int key_check(aesgcm_ctx *c, uint8_t out[4]) {
    uint8_t in[16], ct[16];
    memset(in, 0xA5, 16);
    const int rc = aesgcm_ecb_encrypt(c, in, 1, ct);
    if (rc) return rc;
    memcpy(out, ct, 4);
    return AESGCM_OK;
}
int aesgcm_stream_export(aesgcm_ctx *c, uint8_t blob[64]) {
    if (!c || !blob) return AESGCM_EARG;
    if (!c->s.active) return AESGCM_ESTATE;
    HIPCHK(hipSetDevice(c->device));
    memset(blob, 0, 64);
    blob[0] = STREAM_BLOB_VERSION; blob[1] = c->s.dec ? 1 : 0; blob[2] = (uint8_t)((c->s.data ? 1 : 0) | (c->s.ragged ? 2 : 0));
    memcpy(blob + 4, c->s.iv, 12);
    memcpy(blob + 16, &c->s.aad_len, 8); memcpy(blob + 24, &c->s.len, 8); memcpy(blob + 48, &c->s.blocks, 8);
    // everything this session enqueued -- on the context's stream or, for aesgcm_stream_update_dev, on the caller's -- before Y is read
    HIPCHK(hipDeviceSynchronize());
    HIPCHK(hipMemcpy(blob + 32, c->d_tag + 1, 16, hipMemcpyDeviceToHost));
    int rc = key_check(c, blob + 56);
    if (rc) return rc;
    const u32 sum = blob_sum(blob);
    memcpy(blob + 60, &sum, 4);
    return AESGCM_OK;
}
int aesgcm_stream_import(aesgcm_ctx *c, const uint8_t blob[64]) {
    if (!c || !blob) return AESGCM_EARG;
    if (c->s.active) return AESGCM_ESTATE;                      // a session of this context's own is open
    u32 sum;
    memcpy(&sum, blob + 60, 4);
    if (blob[0] != STREAM_BLOB_VERSION || blob[1] > 1 || (blob[2] & ~3u) || blob[3] || sum != blob_sum(blob)) return AESGCM_EARG;
    HIPCHK(hipSetDevice(c->device));
    uint8_t kc[4];
    int rc = key_check(c, kc);
    if (rc) return rc;
    if (memcmp(kc, blob + 56, 4)) { snprintf(g_err, sizeof g_err, "aesgcm_stream_import: the state was exported under another key"); return AESGCM_EARG; }
    aesgcm_ctx::StreamState st;
    st.active = true; st.dec = blob[1]; st.data = (blob[2] & 1) != 0; st.ragged = (blob[2] & 2) != 0;
    memcpy(st.iv, blob + 4, 12);
    memcpy(&st.aad_len, blob + 16, 8); memcpy(&st.len, blob + 24, 8); memcpy(&st.blocks, blob + 48, 8);
    if (check_lengths(st.aad_len, st.len) || st.blocks != (st.aad_len + 15) / 16 + (st.len + 15) / 16 || (!st.ragged && (st.len & 15)) || (st.data ? false : st.len != 0)) return AESGCM_EARG;
    HIPCHK(hipMemcpyAsync(c->d_tag + 1, blob + 32, 16, hipMemcpyHostToDevice, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    c->s = st;
    return AESGCM_OK;
}

int aesgcm_wipe_failed_dev(int device, size_t n_pkts, void *d_out, size_t pkt_len, const uint64_t *d_data_off, const int *d_auth, void *stream) {
    if (n_pkts >= (((size_t)1) << 31) || pkt_len >= (((size_t)1) << 32)) return AESGCM_ETOOLONG;
    if (n_pkts && (!d_out || !d_auth)) return AESGCM_EARG;
    return wipe_failed(device, n_pkts, d_out, pkt_len, (const u64 *)d_data_off, d_auth, (hipStream_t)stream);
}


// ---------------------------------------------------------------- messages wherever they live: by rows, always
int aesgcm_messages_crypt_dev(aesgcm_ctx *c, int decrypt, size_t n_msgs, const void *d_ivs,
                              const uint64_t *d_aad_ptr, const uint32_t *d_aad_len,
                              const uint64_t *d_in_ptr, const uint32_t *d_len, const uint64_t *d_out_ptr,
                              void *d_tags, const void *d_expect_tags, int *d_auth, void *stream) {
    if (!c) return AESGCM_EARG;
    if (!n_msgs) return AESGCM_OK;
    if (!d_ivs || !d_tags || !d_in_ptr || !d_len || !d_out_ptr || ((d_aad_ptr != nullptr) != (d_aad_len != nullptr))) return AESGCM_EARG;
    if (decrypt && c->wipe_on_auth_fail && d_expect_tags && !d_auth) return AESGCM_EARG;      // "no unauthenticated plaintext" needs the per-message verdicts: without d_auth nothing would be compared, nothing wiped
    if (n_msgs >= (((size_t)1) << 31)) return AESGCM_ETOOLONG;
    HIPCHK(hipSetDevice(c->device));
    RowsParams r;
    memset(&r, 0, sizeof r);
    r.ivs = (const unsigned char *)d_ivs; r.tags = (unsigned char *)d_tags; r.expect = (const unsigned char *)d_expect_tags; r.auth = d_auth;
    r.in_ptr = (const u64 *)d_in_ptr; r.out_ptr = (const u64 *)d_out_ptr; r.aad_ptr = (const u64 *)d_aad_ptr; r.len_arr = d_len; r.alen_arr = d_aad_len;
    r.n_pkts = (u32)n_msgs;
    // routed per message (round 6): the short ones are the packet kernels', which read the same arrays
    PktParams k;
    memset(&k, 0, sizeof k);
    k.ivs = r.ivs; k.tags = r.tags; k.expect = r.expect; k.auth = d_auth;
    k.aligned = 1;                                                                      // per message: its two addresses decide (pkt_info)
    const int rc = packets_rows(c, decrypt, r, pick_stream(c, stream), &k);
    if (!rc && decrypt && c->wipe_on_auth_fail && d_expect_tags) return wipe_failed(c->device, n_msgs, nullptr, 0, nullptr, d_auth, pick_stream(c, stream), (const u64 *)d_out_ptr, d_len);
    return rc;
}

// ---------------------------------------------------------------- packets under the context's key
int aesgcm_packets_crypt_dev(aesgcm_ctx *c, int decrypt, size_t n_pkts, const void *d_ivs,
                             const void *d_aad, size_t aad_len, const uint64_t *d_aad_off,
                             const void *d_in, size_t pkt_len, const uint64_t *d_data_off, void *d_out,
                             void *d_tags, const void *d_expect_tags, int *d_auth, void *stream) {
    if (!c) return AESGCM_EARG;
    if (!n_pkts) return AESGCM_OK;
    if (!d_ivs || !d_tags || ((aad_len || d_aad_off) && !d_aad) || ((pkt_len || d_data_off) && (!d_in || !d_out))) return AESGCM_EARG;
    if (decrypt && c->wipe_on_auth_fail && d_expect_tags && !d_auth) return AESGCM_EARG;      // "no unauthenticated plaintext" needs the per-packet verdicts: without d_auth nothing would be compared, nothing wiped
    if (n_pkts >= (((size_t)1) << 31) || pkt_len >= (((size_t)1) << 28) || aad_len >= (((size_t)1) << 28)) return AESGCM_ETOOLONG;
    HIPCHK(hipSetDevice(c->device));
    PktParams p;
    memset(&p, 0, sizeof p);
    p.ivs = (const unsigned char *)d_ivs; p.aad = (const unsigned char *)d_aad; p.in = (const unsigned char *)d_in;
    p.out = (unsigned char *)d_out; p.tags = (unsigned char *)d_tags; p.expect = (const unsigned char *)d_expect_tags; p.auth = d_auth;
    p.data_off = (const u64 *)d_data_off; p.aad_off = (const u64 *)d_aad_off;
    p.n_pkts = (u32)n_pkts; p.pkt_len = (u32)pkt_len; p.aad_len = (u32)aad_len;
    p.aligned = (((uintptr_t)d_in | (uintptr_t)d_out) & 15) == 0 && (d_data_off || pkt_len % 16 == 0);
    // Offset arrays: the lengths are on the device, and so is the choice (round 6) -- every message is ROUTED by its own size, the long ones by rows, the short ones
    // through the packet kernels, in one call (the reference's own traffic is both at once: tb/gcm_gctr.py:279-281 draws lengths from a U-shaped distribution).
    // pkt_len is ignored in this form.  Fixed-size records: the host knows the one size and routes the whole call (packets_by_rows).
    const bool routed = d_data_off != nullptr;
    if (routed || packets_by_rows(c, n_pkts, pkt_len)) {                          // message-sized packets: the rows of all of them through k_body's row loop
        RowsParams r;
        memset(&r, 0, sizeof r);
        r.ivs = (const unsigned char *)d_ivs; r.aad = (const unsigned char *)d_aad; r.in = (const unsigned char *)d_in;
        r.out = (unsigned char *)d_out; r.tags = (unsigned char *)d_tags; r.expect = (const unsigned char *)d_expect_tags; r.auth = d_auth;
        r.data_off = (const u64 *)d_data_off; r.aad_off = (const u64 *)d_aad_off;
        r.n_pkts = (u32)n_pkts; r.pkt_len = routed ? 0u : (u32)pkt_len; r.aad_len = (u32)aad_len;
        const int rc = packets_rows(c, decrypt, r, pick_stream(c, stream), routed ? &p : nullptr);
        if (!rc && decrypt && c->wipe_on_auth_fail && d_expect_tags) return wipe_failed(c->device, n_pkts, d_out, pkt_len, (const u64 *)d_data_off, d_auth, pick_stream(c, stream));
        return rc;
    }
    const u32 n_cu = (u32)c->G / 2;                                                 // c->G = two workgroups per CU
    int lg = packets_pick_lg(n_cu, n_pkts, pkt_len);                       // (fixed-size records from here on: offset arrays took the routed path above)
#ifdef AESGCM_DEBUG_KNOBS
    if (g_force.pkt_lanes) lg = g_force.pkt_lanes == 1 ? 0 : g_force.pkt_lanes == 64 ? 6 : g_force.pkt_lanes == 16 ? 4 : g_force.pkt_lanes == 8 ? 3 : 2;
#endif
    const int shape = lg == 0 ? 'l' : lg == 6 ? 'w' : 'g';
    hipError_t launch_err = hipSuccess;
    hipStream_t st = pick_stream(c, stream);
    p.counter = c->d_counter; p.counter_base = c->counter_base;
    if (shape == 'l') {
        const u32 nb = (u32)((n_pkts + 63) / 64);
        // the ILP form (512-lane workgroups, eight independent keystream chains per line) while the packets fit one round of it; its workgroups are spread over
        // all CUs, a wave of 64 packets each first
        // Measured, AES-256, GiB/s 768-lane form / ILP form (profiles/r04/packets_sweep_ilp_aes256.txt): 1 KiB packets 16384 66 / 78, 65536 255 / 306, 131072 481 / 592;
        // 256 B 32768 99 / 95, 98304 245 / 266, 131072 295 / 330; 64 B (no whole line to work on) 16384 27 / 19.
        // Packets shorter than two lines gain from it only once they fill the chip (fewer, fatter waves): 196608 x 256 B 380 / 414, 262144 442 / 460 (2^20: 682 / 642);
        // 64 B 196608 127 / 146, 393216 183 / 201, 2^20 254 / 266.
        bool ilp = n_pkts <= (size_t)n_cu * AESGCM_PKTL_WG_ILP ? (pkt_len >= 512 || (pkt_len >= 256 && n_pkts >= 49152))
                                                                : (n_pkts >= (size_t)n_cu * AESGCM_PKTL_WG && (pkt_len <= 64 || (pkt_len <= 256 && n_pkts <= 300000)));
#ifdef AESGCM_DEBUG_KNOBS
        if (g_force.pkt_ilp) ilp = g_force.pkt_ilp == 1;
#endif
        const u32 waves_per_wg = (ilp ? AESGCM_PKTL_WG_ILP : AESGCM_PKTL_WG) / 64;
        u32 wgs = ilp ? nb : (nb + waves_per_wg - 1) / waves_per_wg;
        if (wgs > n_cu) wgs = n_cu;                                                  // one workgroup per CU (registers, and with four T-tables the LDS)
        c->counter_base += nb + wgs * waves_per_wg;                                 // every wave ends on one failing fetch
        launch_err = klaunch_pktl(c->nr, decrypt, ilp, wgs, st, c->km, c->tables, p);
    } else {
        const u32 P = 64u >> lg;                                                    // packets per wave-iteration
        p.plain = (lg == 6 || lg == 2) && !d_aad_off && !aad_len && p.aligned && pkt_len && pkt_len % ((size_t)16 << lg) == 0;
        const u32 waves_per_wg = (u32)PKTG_WG(lg) / 64;
        // deal: about 4 dispenser fetches per resident wave, a multiple of P, at most 64 packets (one E_K(J0) pass per fetch)
        u32 deal = (u32)(n_pkts / ((size_t)n_cu * waves_per_wg * 4));
        deal = deal / P * P;
        deal = deal < P ? P : deal > PKTG_MAX_DEAL ? PKTG_MAX_DEAL : deal;
#ifdef AESGCM_DEBUG_KNOBS
        if (g_force.pkt_deal >= 1 && g_force.pkt_deal <= (int)PKTG_MAX_DEAL) deal = ((u32)g_force.pkt_deal + P - 1) / P * P;
#endif
        p.deal = deal;
        const u32 nb = (u32)((n_pkts + deal - 1) / deal);
        u32 wgs = (nb + waves_per_wg - 1) / waves_per_wg;
        if (wgs > n_cu) wgs = n_cu;                                                  // one workgroup per CU (LDS)
        c->counter_base += nb + wgs * waves_per_wg;                                 // every wave ends on one failing fetch
        launch_err = klaunch_pktg(c->nr, decrypt, lg, wgs, st, c->km, c->tables, p);
    }
    const hipError_t le = launch_err;
    if (le != hipSuccess) { c->counter_base = p.counter_base; return hip_fail(le, "k_pkt launch"); }
    if (decrypt && c->wipe_on_auth_fail && d_expect_tags) return wipe_failed(c->device, n_pkts, d_out, pkt_len, nullptr, d_auth, st);
    return AESGCM_OK;
}


// What the device decided about the context's most recent ROUTED call (offset arrays, messages wherever they live): waits for the device, then reads the header
// k_len_scan and the plan left in the context's scratch.  For benches, profiles and tests -- the host never needs it.
int aesgcm_ctx_last_route(aesgcm_ctx *c, uint64_t out[4]) {
    if (!c || !out) return AESGCM_EARG;
    if (!c->rows_buf) return AESGCM_ESTATE;
    HIPCHK(hipSetDevice(c->device));
    HIPCHK(hipDeviceSynchronize());
    RowsHdr h;
    HIPCHK(hipMemcpy(&h, c->rows_buf, sizeof h, hipMemcpyDeviceToHost));      // (the header is the first thing rows_carve hands out)
    out[0] = h.route_min; out[1] = h.n_small; out[2] = h.n_small ? 1ull << h.pkt_lg : 0ull; out[3] = h.G;
    return AESGCM_OK;
}

// The instruction stream of the packet kernels WITHOUT the data's HBM traffic (k_pktl<NR, 2, 0> / k_pktg<NR, 2, LG>: IVs, offsets, AAD and tags still move): what the
// formulation of the frame path costs by itself -- one lane (or lane group) per frame, T-table AES, the H-table GHASH, per-frame E_K(J0) and length block -- on this chip
// at this moment's clocks.  The yardstick bench.py --config frames prints as roofline.formulation_ceiling, as aesgcm_ctx_ceiling_probe is for the stream kernel and
// aesgcm_batch_ceiling_probe_dev for cfg5.  The call takes the routed path of aesgcm_packets_crypt_dev with offset arrays (the sort, the shape chosen on the device for the
// count), every frame to the packet kernels whatever its size; no data buffer is touched (none is passed).  Asynchronous on `stream`; the caller times it (aesgcm_timer_*).
int aesgcm_frames_ceiling_probe_dev(aesgcm_ctx *c, size_t n_pkts, const void *d_ivs, const void *d_aad, const uint64_t *d_aad_off, const uint64_t *d_data_off, void *d_tags, void *stream) {
    if (!c || !n_pkts || !d_ivs || !d_tags || !d_data_off || (d_aad_off && !d_aad)) return AESGCM_EARG;
    if (n_pkts >= (((size_t)1) << 31)) return AESGCM_ETOOLONG;
    HIPCHK(hipSetDevice(c->device));
    PktParams p;
    memset(&p, 0, sizeof p);
    p.ivs = (const unsigned char *)d_ivs; p.aad = (const unsigned char *)d_aad; p.tags = (unsigned char *)d_tags;
    p.data_off = (const u64 *)d_data_off; p.aad_off = (const u64 *)d_aad_off;
    p.n_pkts = (u32)n_pkts; p.aligned = 1;
    RowsParams r;
    memset(&r, 0, sizeof r);
    r.ivs = p.ivs; r.aad = p.aad; r.tags = p.tags; r.data_off = p.data_off; r.aad_off = p.aad_off; r.n_pkts = (u32)n_pkts;
    return packets_rows(c, 2, r, pick_stream(c, stream), &p);
}

int aesgcm_batch_crypt_dev(int device, int decrypt, size_t n_pkts, size_t key_len, const void *d_keys, const void *d_ivs,
                           const void *d_aad, size_t aad_len, const void *d_in, size_t pkt_len, void *d_out,
                           void *d_tags, const void *d_expect_tags, int *d_auth, void *stream) {
    if (!n_pkts) return AESGCM_OK;
    if (!d_keys || !d_ivs || !d_tags || (aad_len && !d_aad) || (pkt_len && (!d_in || !d_out))) return AESGCM_EARG;
    if (pkt_len >= (((size_t)1) << 28) || aad_len >= (((size_t)1) << 28)) return AESGCM_ETOOLONG;
    BatchParams p;
    memset(&p, 0, sizeof p);
    p.keys = (const unsigned char *)d_keys; p.ivs = (const unsigned char *)d_ivs; p.aad = (const unsigned char *)d_aad;
    p.in = (const unsigned char *)d_in; p.out = (unsigned char *)d_out; p.tags = (unsigned char *)d_tags;
    p.expect = (const unsigned char *)d_expect_tags; p.auth = d_auth;
    p.pkt_len = (u32)pkt_len; p.aad_len = (u32)aad_len;
    p.aligned = (pkt_len % 16 == 0) && (((uintptr_t)d_in | (uintptr_t)d_out) & 15) == 0;
    return batch_launch(device, decrypt, n_pkts, key_len, p, stream);
}


// The instruction stream of the batch kernel WITHOUT the data's HBM traffic (k_batch3<NR, 2, 3>: keys, IVs and tags still move, 44 - 60 bytes per packet): what
// the formulation costs by itself on this chip at this moment's clocks -- the yardstick bench.py --config cfg5 prints as roofline.formulation_ceiling, as
// aesgcm_ctx_ceiling_probe is for the stream kernel.  Exists for calls that take the 8-lanes-per-packet shape (BASELINE config 5 does); AESGCM_EARG otherwise.
int aesgcm_batch_ceiling_probe_dev(int device, size_t n_pkts, size_t key_len, const void *d_keys, const void *d_ivs, size_t pkt_len, void *d_tags, void *stream) {
    if (!n_pkts || !d_keys || !d_ivs || !d_tags || !pkt_len) return AESGCM_EARG;
    if (pkt_len >= (((size_t)1) << 28)) return AESGCM_ETOOLONG;
    BatchParams p;
    memset(&p, 0, sizeof p);
    p.keys = (const unsigned char *)d_keys; p.ivs = (const unsigned char *)d_ivs; p.tags = (unsigned char *)d_tags;
    p.pkt_len = (u32)pkt_len;
    p.aligned = pkt_len % 16 == 0;
    return batch_launch(device, 2, n_pkts, key_len, p, stream);
}

int aesgcm_batch_crypt_var_dev(int device, int decrypt, size_t n_pkts, size_t key_len, const void *d_keys, const void *d_ivs,
                               const void *d_aad, const uint64_t *d_aad_off, const void *d_in, const uint64_t *d_data_off,
                               void *d_out, void *d_tags, const void *d_expect_tags, int *d_auth, void *stream) {
    if (!n_pkts) return AESGCM_OK;
    if (!d_keys || !d_ivs || !d_tags || !d_data_off || !d_in || !d_out || (d_aad_off && !d_aad)) return AESGCM_EARG;
    BatchParams p;
    memset(&p, 0, sizeof p);
    p.keys = (const unsigned char *)d_keys; p.ivs = (const unsigned char *)d_ivs; p.aad = d_aad_off ? (const unsigned char *)d_aad : nullptr;
    p.in = (const unsigned char *)d_in; p.out = (unsigned char *)d_out; p.tags = (unsigned char *)d_tags;
    p.expect = (const unsigned char *)d_expect_tags; p.auth = d_auth;
    p.data_off = (const u64 *)d_data_off; p.aad_off = (const u64 *)d_aad_off;
    p.aligned = (((uintptr_t)d_in | (uintptr_t)d_out) & 15) == 0;      // per packet: and its offset is a multiple of 16
    return batch_launch(device, decrypt, n_pkts, key_len, p, stream);
}


// Which kernel shape a call with these arguments takes (lanes per packet: 1 = one lane per packet, 4 / 8 / 16 = a lane group, 64 = a whole wave); pkt_len = 0
// with var_len != 0 describes the offset-array forms.  What bench.py and the profiling scripts print beside their numbers.
int aesgcm_batch_shape(int device, size_t n_pkts, size_t pkt_len, int var_len, int *lanes_per_packet) {
    if (!lanes_per_packet || !n_pkts) return AESGCM_EARG;
    DeviceState *ds;
    int rc = device_state(device, &ds);
    if (rc) return rc;
    int lg = batch_pick_lg(ds->n_cu, n_pkts, pkt_len, var_len != 0);
#ifdef AESGCM_DEBUG_KNOBS
    if (g_force.batch_lanes) lg = g_force.batch_lanes == 8 ? 3 : g_force.batch_lanes == 16 ? 4 : 6;
#endif
    *lanes_per_packet = 1 << lg;
    return AESGCM_OK;
}

int aesgcm_packets_shape(const aesgcm_ctx *c, size_t n_pkts, size_t pkt_len, int var_len, int *lanes_per_packet) {
    if (!c || !lanes_per_packet || !n_pkts) return AESGCM_EARG;
    if (var_len) { *lanes_per_packet = AESGCM_SHAPE_MIXED; return AESGCM_OK; }         // the lengths are on the device: every message is routed there, by rows or to the packet kernels
    if (packets_by_rows(c, n_pkts, pkt_len)) { *lanes_per_packet = AESGCM_SHAPE_ROWS; return AESGCM_OK; }
    int lg = packets_pick_lg((u32)c->G / 2, n_pkts, pkt_len);
#ifdef AESGCM_DEBUG_KNOBS
    if (g_force.pkt_lanes) lg = g_force.pkt_lanes == 1 ? 0 : g_force.pkt_lanes == 64 ? 6 : g_force.pkt_lanes == 16 ? 4 : g_force.pkt_lanes == 8 ? 3 : 2;
#endif
    *lanes_per_packet = 1 << lg;
    return AESGCM_OK;
}


int aesgcm_encrypt_pipelined(aesgcm_ctx *c, const uint8_t iv[12], const uint8_t *aad, size_t aad_len,
                             const uint8_t *pt, size_t len, uint8_t *ct, uint8_t tag[16], size_t chunk_bytes) {
    if (!c || !iv || !tag || (aad_len && !aad) || (len && (!pt || !ct))) return AESGCM_EARG;
    return crypt_pipelined(c, 0, iv, aad, aad_len, pt, len, ct, tag, chunk_bytes);
}

int aesgcm_decrypt_pipelined(aesgcm_ctx *c, const uint8_t iv[12], const uint8_t *aad, size_t aad_len,
                             const uint8_t *ct, size_t len, uint8_t *pt, const uint8_t *expect_tag, uint8_t tag_out[16],
                             size_t chunk_bytes) {
    if (!c || !iv || (aad_len && !aad) || (len && (!ct || !pt))) return AESGCM_EARG;
    uint8_t t[16];
    int rc = crypt_pipelined(c, 1, iv, aad, aad_len, ct, len, pt, t, chunk_bytes);
    if (rc) return rc;
    if (tag_out) memcpy(tag_out, t, 16);
    if (expect_tag && !ct_compare16(t, expect_tag)) {
        if (c->wipe_on_auth_fail && len) {                       // the chunks have landed in `pt` already (that is the pipeline): wipe them, and the device's two chunk slots
            memset(pt, 0, len);
            for (int i = 0; i < 2; i++) if (c->pl_buf[i]) HIPCHK(hipMemsetAsync(c->pl_buf[i], 0, c->pl_cap, c->stream));
            HIPCHK(hipStreamSynchronize(c->stream));
        }
        return AESGCM_EAUTH;
    }
    return AESGCM_OK;
}

// page-locked host memory, so that the pipelined path's copies are true DMA (pageable buffers work, slower)
int aesgcm_host_alloc(void **p, size_t bytes) {
    if (!p) return AESGCM_EARG;
    hipError_t e = hipHostMalloc(p, bytes ? bytes : 16, hipHostMallocDefault);
    if (e == hipErrorOutOfMemory) return AESGCM_ENOMEM;
    if (e != hipSuccess) return hip_fail(e, "hipHostMalloc");
    return AESGCM_OK;
}

int aesgcm_host_free(void *p) {
    HIPCHK(hipHostFree(p));
    return AESGCM_OK;
}


// ---------------------------------------------------------------- memory helpers
int aesgcm_dev_alloc(int device, void **p, size_t bytes) {
    if (!p) return AESGCM_EARG;
    HIPCHK(hipSetDevice(device));
    hipError_t e = hipMalloc(p, bytes ? bytes : 16);
    if (e == hipErrorOutOfMemory) return AESGCM_ENOMEM;
    if (e != hipSuccess) return hip_fail(e, "hipMalloc");
    return AESGCM_OK;
}

int aesgcm_dev_free(int device, void *p) {
    HIPCHK(hipSetDevice(device));
    HIPCHK(hipFree(p));
    return AESGCM_OK;
}

int aesgcm_dev_upload(int device, void *d, const void *h, size_t n) {
    HIPCHK(hipSetDevice(device));
    if (n) HIPCHK(hipMemcpy(d, h, n, hipMemcpyHostToDevice));
    return AESGCM_OK;
}

int aesgcm_dev_download(int device, void *h, const void *d, size_t n) {
    HIPCHK(hipSetDevice(device));
    if (n) HIPCHK(hipMemcpy(h, d, n, hipMemcpyDeviceToHost));
    return AESGCM_OK;
}

int aesgcm_dev_sync(int device) {
    HIPCHK(hipSetDevice(device));
    HIPCHK(hipDeviceSynchronize());
    return AESGCM_OK;
}

int aesgcm_dev_copy(int device, void *d_dst, const void *d_src, size_t bytes, void *stream) {
    if (bytes && (!d_dst || !d_src)) return AESGCM_EARG;
    if (((uintptr_t)d_dst | (uintptr_t)d_src | bytes) & 15) return AESGCM_EALIGN;
    if (!bytes) return AESGCM_OK;
    HIPCHK(hipSetDevice(device));
    const u64 n16 = bytes / 16;
    if ((n16 + 255) / 256 > 0x7FFFFFFFull) return AESGCM_ETOOLONG;
    HIPCHK(klaunch_copy16((hipStream_t)stream, (uint4 *)d_dst, (const uint4 *)d_src, n16));
    return AESGCM_OK;
}

int aesgcm_fill_splitmix64_dev(int device, void *d_buf, size_t len, uint64_t seed, uint64_t first_word, void *stream) {
    if (len && !d_buf) return AESGCM_EARG;
    if ((uintptr_t)d_buf & 7) return AESGCM_EALIGN;
    if (!len) return AESGCM_OK;
    HIPCHK(hipSetDevice(device));
    size_t nw = len / 8;
    size_t blocks = (nw + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    if (blocks < 1) blocks = 1;
    HIPCHK(klaunch_fill_splitmix64((hipStream_t)stream, (unsigned)blocks, (u64 *)d_buf, nw, len & 7, seed, first_word));
    return AESGCM_OK;
}


// ---------------------------------------------------------------- timing
// a pair of HIP events for callers that time launches of the context-free entry points (batch) on the stream they use
struct aesgcm_timer { int device; hipEvent_t a, b; };

int aesgcm_timer_create(aesgcm_timer **out, int device) {
    if (!out) return AESGCM_EARG;
    *out = nullptr;
    HIPCHK(hipSetDevice(device));
    aesgcm_timer *t = new (std::nothrow) aesgcm_timer();
    if (!t) return AESGCM_ENOMEM;
    t->device = device; t->a = t->b = nullptr;
    hipError_t e = hipEventCreate(&t->a);
    if (e == hipSuccess) e = hipEventCreate(&t->b);
    if (e != hipSuccess) { if (t->a) hipEventDestroy(t->a); delete t; return hip_fail(e, "hipEventCreate"); }
    *out = t;
    return AESGCM_OK;
}

int aesgcm_timer_start(aesgcm_timer *t, void *stream) { if (!t) return AESGCM_EARG; HIPCHK(hipSetDevice(t->device)); HIPCHK(hipEventRecord(t->a, (hipStream_t)stream)); return AESGCM_OK; }

int aesgcm_timer_stop(aesgcm_timer *t, void *stream) { if (!t) return AESGCM_EARG; HIPCHK(hipSetDevice(t->device)); HIPCHK(hipEventRecord(t->b, (hipStream_t)stream)); return AESGCM_OK; }

int aesgcm_timer_ms(aesgcm_timer *t, double *ms) {
    if (!t || !ms) return AESGCM_EARG;
    HIPCHK(hipSetDevice(t->device));
    HIPCHK(hipEventSynchronize(t->b));
    float f = 0;
    HIPCHK(hipEventElapsedTime(&f, t->a, t->b));
    *ms = f;
    return AESGCM_OK;
}

int aesgcm_timer_destroy(aesgcm_timer *t) {
    if (!t) return AESGCM_OK;
    hipSetDevice(t->device);
    hipEventDestroy(t->a); hipEventDestroy(t->b);
    delete t;
    return AESGCM_OK;
}


int aesgcm_ctx_timing_enable(aesgcm_ctx *c, int on) {
    if (!c) return AESGCM_EARG;
    c->timing = on != 0;
    return AESGCM_OK;
}

int aesgcm_ctx_timing_read(aesgcm_ctx *c, uint64_t *n, double *total_ms, int reset) {
    if (!c) return AESGCM_EARG;
    HIPCHK(hipSetDevice(c->device));
    double tot = 0;
    for (auto &e : c->ev) {
        HIPCHK(hipEventSynchronize(e.second));
        float ms = 0;
        HIPCHK(hipEventElapsedTime(&ms, e.first, e.second));
        tot += ms;
    }
    if (n) *n = c->ev.size();
    if (total_ms) *total_ms = tot;
    if (reset) { for (auto &e : c->ev) c->ev_pool.push_back(e); c->ev.clear(); }
    return AESGCM_OK;
}


// The fused kernel's instruction stream WITHOUT its HBM traffic: k_body<NR, MODE_PROBE> over a virtual range of `nbytes`
// (same chunking, same dispensers, same LDS tables, same scalar loads, same GHASH; no global load, no global store
// except the chunk items).  Its time is the ceiling of the T-table formulation on this chip at this moment's clocks.
int aesgcm_ctx_ceiling_probe(aesgcm_ctx *c, size_t nbytes, double *ms, uint64_t *blocks) {
    if (!c || !ms) return AESGCM_EARG;
    HIPCHK(hipSetDevice(c->device));
    BodySplit b;
    if (!plan_body_split(nbytes, 0, c->tw_override, 0, &b)) return AESGCM_EARG;
    const uint8_t iv[12] = {0};
    Partials pp;
    const bool was = c->timing;
    c->timing = true;
    HIPCHK(hipStreamSynchronize(c->stream));
    const size_t mark = c->ev.size();
    int rc = enqueue_body(c, MODE_PROBE, iv, b, nullptr, nullptr, 0, c->stream, &pp);
    c->timing = was;
    if (rc) return rc;
    HIPCHK(hipStreamSynchronize(c->stream));
    float t = 0;
    HIPCHK(hipEventElapsedTime(&t, c->ev[mark].first, c->ev[mark].second));
    c->ev_pool.push_back(c->ev[mark]);
    c->ev.erase(c->ev.begin() + (long)mark);
    *ms = t;
    if (blocks) *blocks = b.body_blocks;
    return AESGCM_OK;
}


int aesgcm_ctx_wg_trace(aesgcm_ctx *c, uint64_t *out, size_t max_wgs, size_t *n_wgs) {
    if (!c || !out || !n_wgs) return AESGCM_EARG;
    HIPCHK(hipSetDevice(c->device));
    size_t n = c->last_np < max_wgs ? c->last_np : max_wgs;
    HIPCHK(hipDeviceSynchronize());
    if (n) HIPCHK(hipMemcpy(out, c->d_trace, n * 4 * sizeof(u64), hipMemcpyDeviceToHost));
    *n_wgs = n;
    return AESGCM_OK;
}

}  // extern "C"
