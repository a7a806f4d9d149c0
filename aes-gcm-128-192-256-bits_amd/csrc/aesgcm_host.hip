// aesgcm_host.hip -- the host runtime of libaesgcm_hip.so: contexts and per-device state, the launch planners (which launches a message takes: DESIGN.md section 6),
// the shape rules of the packet paths (section 8), the scratch and the cut of the row path (aesgcm_rows.h), the pipelined host-buffer path.  No device code and no
// kernel name in this file: launches go through the klaunch_* functions of aesgcm_kernels.hip (aesgcm_internal.h).
#include "aesgcm_internal.h"

#include <algorithm>
#include <new>
#include <stddef.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

thread_local char g_err[256] = "";
int hip_fail(hipError_t e, const char *what) {
    snprintf(g_err, sizeof g_err, "%s: %s", what, hipGetErrorString(e));
    return AESGCM_EHIP;
}

std::mutex &g_mu = *new std::mutex();
std::vector<DeviceState> &g_dev = *new std::vector<DeviceState>();     // never destroyed (as g_ctxs): contexts may outlive this library's static destructors

int device_state(int device, DeviceState **out) {
    std::lock_guard<std::mutex> lk(g_mu);
    int n = 0;
    HIPCHK(hipGetDeviceCount(&n));
    if (device < 0 || device >= n) { snprintf(g_err, sizeof g_err, "device %d out of range (%d visible)", device, n); return AESGCM_EHIP; }
    if ((int)g_dev.size() < n) g_dev.resize(n);
    DeviceState &d = g_dev[device];
    if (!d.tables) {
        HIPCHK(hipSetDevice(device));
        hipDeviceProp_t prop;
        HIPCHK(hipGetDeviceProperties(&prop, device));
        d.n_cu = prop.multiProcessorCount;
        DevTables *t = nullptr;
        HIPCHK(hipMalloc(&t, sizeof(DevTables)));
        HIPCHK(klaunch_init_tables(t));
        HIPCHK(hipDeviceSynchronize());
        d.tables = t;
        HIPCHK(hipMalloc(&d.batch_counter, 4 * BATCH_DISPENSERS));
        HIPCHK(hipMemset(d.batch_counter, 0, 4 * BATCH_DISPENSERS));
    }
    *out = &d;
    return AESGCM_OK;
}




int set_lds_attrs(int device, DeviceState *ds) {
    // 72 KiB and more of dynamic LDS per workgroup exceed the 64 KiB default cap: opt in once per kernel instance and device (klaunch_set_attributes, in the kernels' translation unit)
    std::lock_guard<std::mutex> lk(g_mu);
    if (ds->attrs) return AESGCM_OK;
    HIPCHK(hipSetDevice(device));
    HIPCHK(klaunch_set_attributes());
    ds->attrs = true;
    return AESGCM_OK;
}


int grow_parts(aesgcm_ctx *c, size_t need) {
    if (need <= c->parts_cap) return AESGCM_OK;
    if (c->parts) { HIPCHK(hipDeviceSynchronize()); HIPCHK(hipFree(c->parts)); c->parts = nullptr; c->parts_cap = 0; }   // rare: first big message
    size_t n = need < 4096 ? 4096 : need;
    hipError_t e = hipMalloc(&c->parts, n * 64 * sizeof(uint4));
    if (e == hipErrorOutOfMemory) return AESGCM_ENOMEM;
    if (e != hipSuccess) return hip_fail(e, "hipMalloc");
    c->parts_cap = n;
    return AESGCM_OK;
}

// device address of the key's precomputed table of H^e, or NULL
static const uint4 *ptab_ptr(const aesgcm_ctx *c, u64 e) {
    const int k = ptab_index(e);
    return k < 0 ? nullptr : reinterpret_cast<const uint4 *>(reinterpret_cast<const char *>(c->km) + offsetof(KeyMaterial, ptab)) + (size_t)k * 512;
}
// k_fold levels: n items (period, eA, eB as in FoldParams) -> one item (left in parts, fold_a or fold_b)
int enqueue_fold(aesgcm_ctx *c, const uint4 *items, u32 n, u32 period, u64 eA, u64 eB, hipStream_t st, Partials *po, const FoldClose *close) {
    const uint4 *cur = items;
    int which = 0;
    while (n > 1) {
        // the last level(s) can be k_combine's own: up to 64 items whose spacing has precomputed tables
        if (period <= 1 && n <= COMBINE_MAX_ITEMS && ptab_ptr(c, eA) && (n <= 4 || ptab_ptr(c, 4 * eA)) && (n <= 16 || ptab_ptr(c, 16 * eA))) {
            po->ptr = cur; po->np = n; po->kind = PARTS_ITEM; po->eA = eA;
            return AESGCM_OK;
        }
        FoldParams f;
        plan_fold(f, cur, which ? c->fold_b : c->fold_a, n, period, eA, eB);
        f.tabA = ptab_ptr(c, f.eA); f.tabB = ptab_ptr(c, f.eB); f.tabC = ptab_ptr(c, f.eC);
        const u32 G = fold_wgs(n, f.group);
        if (close && G <= FOLD_CLOSE_MAX_WGS) {
            // a whole message: this level closes the tag itself (FoldClose) -- no further level, no k_combine.  Every closing workgroup stages the lanes' tables (33 KB)
            // and spends ~4 us: the 256 workgroups of a 1 GiB message's first level are one round on the chip and the step gains 11 us (cfg2: 976 -> 963 us); the
            // 2048 of 16 GiB would be eight rounds and cost what they save (profiles/archive/r03c/fold_close_ab.txt), so there the first level stays plain and the second
            // (64 workgroups) closes
            f.close = *close;
            f.close.on = 1; f.close.step = fold_out_step(f);
            HIPCHK(klaunch_fold(G, true, st, c->km, f));
            po->done = true;
            return AESGCM_OK;
        }
        if (G > (which ? FOLD_B_ITEMS : FOLD_A_ITEMS)) { snprintf(g_err, sizeof g_err, "k_fold: %u output items do not fit the level's buffer", G); return AESGCM_EHIP; }
        HIPCHK(klaunch_fold(G, false, st, c->km, f));
        eA = fold_out_step(f); eB = 0; period = 1;
        cur = f.out; n = G; which ^= 1;
    }
    po->ptr = cur; po->np = 1; po->kind = PARTS_ITEM;
    return AESGCM_OK;
}

// the context's cut of a range into head / k_body (dealt chunks) / tail
bool ctx_body_split(const aesgcm_ctx *c, u64 len, u64 first_block, BodySplit *b) {
    return plan_body_split(len, first_block, c->tw_override, c->body_min, b);
}
// Enqueue the fused kernel over (aad, data) and the k_fold levels over its chunk items; describe the result for k_combine.  mode ENC/DEC: GHASH partials.  mode KS/ECB: no GHASH.
int enqueue_main(aesgcm_ctx *c, int mode, const uint8_t iv[12], const void *d_aad, u64 aad_len,
                        const void *d_in, u64 len, void *d_out, u64 first_block, hipStream_t st, Partials *po, bool want_tail) {
    const bool gh = (mode == MODE_ENC || mode == MODE_DEC);
    if (po) *po = Partials();
    if (((uintptr_t)d_in & 15) || ((uintptr_t)d_out & 15)) return AESGCM_EALIGN;
    MainParams p;
    memset(&p, 0, sizeof p);
    const u32 C = plan_main(p, mode, c->tw_override, iv, d_aad, aad_len, d_in, len, d_out, first_block, nullptr);
    if (!C) return AESGCM_OK;
    int rc;
    if (gh && (rc = grow_parts(c, C))) return rc;
    p.parts = c->parts;
    u32 wgs = (C + 1 + AESGCM_MAIN_WG / 64 - 1) / (AESGCM_MAIN_WG / 64);      // one wave per chunk is enough for small inputs (+ one spare for E_K(J0))
    if (wgs > (u32)c->G) wgs = (u32)c->G;
    plan_queues(C, &p.nq, &p.seg);
    if ((u64)wgs * (AESGCM_MAIN_WG / 64) >= (u64)C + 1) p.nq = 0;   // a wave per chunk and a spare: static assignment, the dispensers are not touched
    p.counter = c->d_counter + 16 * (1 + AESGCM_NQ * c->qset);
    p.counter_zero = c->d_counter + 16 * (1 + AESGCM_NQ * (c->qset ^ 1u));
    if (gh && po) { p.ej0 = c->d_tag + 3; po->ej0 = p.ej0; }
    if (gh && po && want_tail && C == 1) { p.tail = 1; p.tag_out = c->d_tag; p.tag_host = c->h_tag_dev; p.gen = gen_take(c); po->done = true; }
    p.trace = nullptr;
    const bool timed = c->timing && !c->timing_mute;
    if (timed) {
        p.trace = c->d_trace;
        HIPCHK(hipMemsetAsync(c->d_trace, 0, sizeof(u64) * 4 * AESGCM_GMAX, st));
    }
    if (!c->timing_mute) c->last_np = wgs;
    std::pair<hipEvent_t, hipEvent_t> evp;
    if (timed) {
        if (!c->ev_pool.empty()) { evp = c->ev_pool.back(); c->ev_pool.pop_back(); }
        else { HIPCHK(hipEventCreate(&evp.first)); HIPCHK(hipEventCreate(&evp.second)); }
        HIPCHK(hipEventRecord(evp.first, st));
    }
    {
        const hipError_t le = klaunch_main(mode, c->nr, wgs, st, c->km, c->tables, p);
        if (le != hipSuccess) {                      // nothing ran: the queues were not touched on the device
            if (timed) c->ev_pool.push_back(evp);
            if (p.tail) gen_give_back(c);
            return hip_fail(le, "k_main launch");
        }
        if (p.nq) c->qset ^= 1u;                      // the launch leaves the other set zeroed for the next dynamic one
        if (c->ev_fused) HIPCHK(hipEventRecord(c->ev_fused, st));
    }
    if (timed) { HIPCHK(hipEventRecord(evp.second, st)); c->ev.push_back(evp); }
    if (gh && po && po->done) return AESGCM_OK;                   // the launch finished the tag itself
    if (gh && po) {
        const u64 eA = (u64)64 * p.Tw;
        if (C <= COMBINE_MAX_ITEMS && (C == 1 || (ptab_ptr(c, eA) && (C <= 4 || ptab_ptr(c, 4 * eA)) && (C <= 16 || ptab_ptr(c, 16 * eA))))) {
            po->ptr = c->parts; po->np = C; po->kind = PARTS_ITEM; po->eA = eA;   // few chunks: k_combine folds them, no k_fold launch
            return AESGCM_OK;
        }
        return enqueue_fold(c, c->parts, C, 1, eA, 0, st, po);
    }
    return AESGCM_OK;
}


// one k_body launch (dealt chunks or cyclic rows) with the context's timing and event bookkeeping
int launch_body(aesgcm_ctx *c, int mode, BodyParams &p, u32 wgs, hipStream_t st) {
    const bool cyc = p.cyc != 0, half = cyc && p.cw == BODY_CYC_WAVES_HALF;
    if (cyc && mode == MODE_PROBE) return AESGCM_EARG;
    if (half && !p.fuse) return AESGCM_EARG;                    // the half shape exists with the in-launch closing only
    if (c->timing) { p.trace = c->d_trace; HIPCHK(hipMemsetAsync(c->d_trace, 0, sizeof(u64) * 4 * AESGCM_GMAX, st)); }
    c->last_np = wgs;
    std::pair<hipEvent_t, hipEvent_t> evp;
    if (c->timing) {
        if (!c->ev_pool.empty()) { evp = c->ev_pool.back(); c->ev_pool.pop_back(); }
        else { HIPCHK(hipEventCreate(&evp.first)); HIPCHK(hipEventCreate(&evp.second)); }
        HIPCHK(hipEventRecord(evp.first, st));
    }
    const hipError_t le = klaunch_body(mode, c->nr, cyc, half, wgs, st, c->km, c->tables, p);
    if (le != hipSuccess) {
        if (c->timing) c->ev_pool.push_back(evp);
        return hip_fail(le, "k_body launch");
    }
    if (!cyc) c->qset ^= 1u;
    if (c->timing) { HIPCHK(hipEventRecord(evp.second, st)); c->ev.push_back(evp); }
    if (c->ev_fused) HIPCHK(hipEventRecord(c->ev_fused, st));
    return AESGCM_OK;
}
// k_body over the planned split + the k_fold levels over its interleaved chunk items
int enqueue_body(aesgcm_ctx *c, int mode, const uint8_t iv[12], const BodySplit &b, const void *d_in, void *d_out,
                        u64 first_block, hipStream_t st, Partials *po, const FoldClose *close) {
    BodyParams p;
    memset(&p, 0, sizeof p);
    int rc = grow_parts(c, (size_t)4 * b.S);
    if (rc) return rc;
    plan_body(p, b, iv, d_in, d_out, first_block, c->parts);
    p.ej0 = c->d_tag + 3; po->ej0 = p.ej0;
    const u32 waves_per_wg = AESGCM_BODY_WG / 64;
    u32 wgs = (p.C + waves_per_wg - 1) / waves_per_wg;
#if AESGCM_T4
    if (wgs > (u32)c->G / 2) wgs = (u32)c->G / 2;                 // one 136 KiB workgroup per CU
#else
    if (wgs > (u32)c->G) wgs = (u32)c->G;
#endif
    plan_queues(p.C, &p.nq, &p.seg);
    p.counter = c->d_counter + 16 * (1 + AESGCM_NQ * c->qset);
    p.counter_zero = c->d_counter + 16 * (1 + AESGCM_NQ * (c->qset ^ 1u));
    if ((rc = launch_body(c, mode, p, wgs, st))) return rc;
    // items 4s + v: phases 64 blocks apart inside a super-chunk, super-chunks 256 T blocks apart
    return enqueue_fold(c, c->parts, p.C, 4, 64, (u64)256 * b.T, st, po, close);
}
// A whole range -- AAD, data from any first block, ragged end -- as ONE k_body launch of cyclic rows (plan_body_cyc) and the k_fold level over its
// 4096 items, when the range is of that size (*took says whether it was).  po describes the items and the partial last row for k_combine.
bool cyc_capable(const aesgcm_ctx *c) {
#if AESGCM_T4
    return (u32)c->G / 2 * (AESGCM_BODY_WG / 64) == BODY_CYC_WAVES && c->cyc_max_pieces > c->cyc_min_fused;
#else
    return false;
#endif
}
// Is a message of ANOTHER context of this device under way right now?  Every result goes to its context's pinned host slot with the generation number of its
// launch behind it, so "under way" is: the slot does not show the generation last launched.  What the half shape of the cyclic rows is for (two messages
// share every CU); asked once per whole-message launch, a mutex and a few loads.  Contexts register in ctx_create_common and leave in aesgcm_ctx_destroy.
std::vector<aesgcm_ctx *> &g_ctxs = *new std::vector<aesgcm_ctx *>();          // never destroyed: contexts may be destroyed after this library's static destructors have run
bool others_in_flight(const aesgcm_ctx *c) {
    std::lock_guard<std::mutex> lk(g_mu);
    for (const aesgcm_ctx *o : g_ctxs) {
        if (o == c || o->device != c->device || !o->h_tag) continue;
        const u64 launched = __atomic_load_n(&o->tag_gen, __ATOMIC_RELAXED);
        if (__atomic_load_n(reinterpret_cast<const u64 *>(o->h_tag + 1), __ATOMIC_RELAXED) != launched) return true;
    }
    return false;
}
int enqueue_cyc(aesgcm_ctx *c, int mode, const uint8_t iv[12], const void *d_aad, u64 aad_len, const void *d_in, u64 len, void *d_out,
                       u64 first_block, hipStream_t st, Partials *po, bool *took, bool whole_message_tag) {
    *took = false;
    const bool fused = whole_message_tag && c->cyc_fuse;
    if (!cyc_capable(c) || len < (fused ? c->cyc_min_fused : c->cyc_min)) return AESGCM_OK;
    if (((uintptr_t)d_in & 15) || ((uintptr_t)d_out & 15)) return AESGCM_OK;     // the caller's other path reports the alignment
    int rc = grow_parts(c, (size_t)BODY_CYC_WAVES + 1);
    if (rc) return rc;
    BodyParams p;
    // a range with pieces around its body (AAD, an odd first block, a ragged end) costs the other paths a launch pair per piece (+45 .. 80 us,
    // profiles/archive/r03c/general_shape.txt): for those the cyclic launch stays ahead for longer
    const bool pieces = aad_len || (first_block & 255) || (len & 1023);
    const u64 lo = fused ? c->cyc_min_fused : c->cyc_min, hi = pieces ? c->cyc_max_pieces : fused ? c->cyc_max_fused : c->cyc_max;
    const bool half = fused && len < c->cyc_half_max && (c->cyc_half == 1 || (c->cyc_half == 2 && others_in_flight(c)));   // two workgroups per CU: for messages in flight beside each other
    if (!plan_body_cyc(p, mode, iv, d_aad, aad_len, d_in, len, d_out, first_block, c->parts, lo, hi, half ? BODY_CYC_WAVES_HALF : BODY_CYC_WAVES)) return AESGCM_OK;
    *took = true;
    *po = Partials();
    p.prio_rows = c->cyc_prio;
    if (fused) {                                                                // the launch closes the tag itself (cyc_close): nothing behind it
        p.fuse = 1; p.aad_len = aad_len; p.ct_len = len; p.acc = c->d_cyc;
        p.tag_out = c->d_tag; p.tag_host = c->h_tag_dev; p.gen = gen_take(c);
        po->done = true;
        c->last_shape = half ? AESGCM_LAUNCH_CYCLIC_HALF : AESGCM_LAUNCH_CYCLIC;
        rc = launch_body(c, mode, p, half ? BODY_CYC_WAVES_HALF / (AESGCM_BODYH_WG / 64) : BODY_CYC_WAVES / (AESGCM_BODY_WG / 64), st);
        if (rc) gen_give_back(c);
        return rc;
    }
    p.ej0 = c->d_tag + 3; po->ej0 = p.ej0;
    if ((rc = launch_body(c, mode, p, BODY_CYC_WAVES / (AESGCM_BODY_WG / 64), st))) return rc;
    if ((rc = enqueue_fold(c, c->parts, BODY_CYC_WAVES, 1, 64, 0, st, po))) return rc;       // always BODY_CYC_WAVES items, 64 blocks apart
    if (p.tb) { po->tail_item = c->parts + (size_t)BODY_CYC_WAVES * 64; po->tail_blocks = p.tb; }
    return AESGCM_OK;
}

// Y' = Y * H^nb ^ P(aad, data) for a whole range, Y in *state (device).  Large ranges go head / k_body / tail,
// each piece folded into the state in order; small ones are a single k_main launch.
int absorb_range(aesgcm_ctx *c, int mode, const uint8_t iv[12], const void *d_aad, u64 aad_len, const void *d_in, u64 len,
                        void *d_out, u64 first_block, hipStream_t st, uint4 *state, const uint4 **ej0) {
    BodySplit b;
    Partials pp;
    int rc;
    {   // mid-size ranges: the whole range in one k_body launch of cyclic rows
        bool took;
        if ((rc = enqueue_cyc(c, mode, iv, d_aad, aad_len, d_in, len, d_out, first_block, st, &pp, &took))) return rc;
        if (took) {
            if (ej0) *ej0 = pp.ej0;
            const u64 nb = (aad_len + 15) / 16 + (len + 15) / 16;
            return enqueue_combine(c, combine_with_items(plan_combine_carry(pp.ptr, pp.np, pp.kind, state, nb), pp.eA, pp.tail_item, pp.tail_blocks), st);
        }
    }
    if (!ctx_body_split(c, len, first_block, &b)) {
        if ((rc = enqueue_main(c, mode, iv, d_aad, aad_len, d_in, len, d_out, first_block, st, &pp))) return rc;
        const u64 nb = (aad_len + 15) / 16 + (len + 15) / 16;
        return nb ? enqueue_combine(c, combine_with_items(plan_combine_carry(pp.ptr, pp.np, pp.kind, state, nb), pp.eA), st) : AESGCM_OK;
    }
    if (((uintptr_t)d_in & 15) || ((uintptr_t)d_out & 15)) return AESGCM_EALIGN;
    const u64 n_aad = (aad_len + 15) / 16;
    if (n_aad + b.head_blocks) {
        c->timing_mute = true;
        rc = enqueue_main(c, mode, iv, d_aad, aad_len, d_in, 16 * b.head_blocks, d_out, first_block, st, &pp);
        c->timing_mute = false;
        if (rc) return rc;
        if ((rc = enqueue_combine(c, combine_with_items(plan_combine_carry(pp.ptr, pp.np, pp.kind, state, n_aad + b.head_blocks), pp.eA), st))) return rc;
    }
    if ((rc = enqueue_body(c, mode, iv, b, d_in, d_out, first_block, st, &pp))) return rc;
    if (ej0) *ej0 = pp.ej0;                                      // valid until the next launch on this context overwrites the slot: consumed by the caller's final combine
    if ((rc = enqueue_combine(c, combine_with_items(plan_combine_carry(pp.ptr, pp.np, pp.kind, state, b.body_blocks), pp.eA), st))) return rc;
    const u64 done = b.head_blocks + b.body_blocks, tail = len - 16 * done;
    if (tail) {
        c->timing_mute = true;
        rc = enqueue_main(c, mode, iv, nullptr, 0, (const unsigned char *)d_in + 16 * done, tail, (unsigned char *)d_out + 16 * done,
                          first_block + done, st, &pp);
        c->timing_mute = false;
        if (rc) return rc;
        if ((rc = enqueue_combine(c, combine_with_items(plan_combine_carry(pp.ptr, pp.np, pp.kind, state, (tail + 15) / 16), pp.eA), st))) return rc;
    }
    return AESGCM_OK;
}

int enqueue_combine(aesgcm_ctx *c, const CombineParams &p0, hipStream_t st) {
    CombineParams p = p0;
    const bool to_slot = p.out == c->d_tag;
    if (to_slot) { p.out_host = c->h_tag_dev; p.gen = gen_take(c); }          // results that go to the tag slot are mirrored to the pinned host slot
    if (p.kind == PARTS_ITEM && p.np > 1) {                       // the launch folds the items itself: tables of H^eA, H^(8 eA)
        p.tabA = ptab_ptr(c, p.eA);
        p.tabB = p.np > 4 ? ptab_ptr(c, 4 * p.eA) : nullptr;
        p.tabC = p.np > 16 ? ptab_ptr(c, 16 * p.eA) : nullptr;
        if (p.np > COMBINE_MAX_ITEMS || !p.tabA || (p.np > 4 && !p.tabB) || (p.np > 16 && !p.tabC)) { if (to_slot) gen_give_back(c); snprintf(g_err, sizeof g_err, "k_combine: %u items, spacing %llu not foldable in the launch", p.np, (unsigned long long)p.eA); return AESGCM_EHIP; }
    }
    const hipError_t le = klaunch_combine(st, c->km, c->tables, p);
    if (le != hipSuccess) { if (to_slot) gen_give_back(c); return hip_fail(le, "k_combine launch"); }
    return AESGCM_OK;
}

int check_lengths(u64 aad_len, u64 len) {
    if (len > MAX_DATA) return AESGCM_ETOOLONG;
    if ((aad_len + 15) / 16 + (len + 15) / 16 >= MAX_SEQ_BLOCKS) return AESGCM_ETOOLONG;
    return AESGCM_OK;
}

// whole message on device pointers; leaves the tag in c->d_tag[0]
int crypt_dev(aesgcm_ctx *c, int dec, const uint8_t iv[12], const void *d_aad, u64 aad_len,
                     const void *d_in, u64 len, void *d_out, hipStream_t st) {
    int rc = check_lengths(aad_len, len);
    if (rc) return rc;
    if (aad_len && !d_aad) return AESGCM_EARG;
    if (len && (!d_in || !d_out)) return AESGCM_EARG;
    HIPCHK(hipSetDevice(c->device));
    {   // mid-size messages: k_body as cyclic rows takes AAD, data and the ragged end in one launch; its items go straight to the tag
        Partials pc;
        bool took;
        if ((rc = enqueue_cyc(c, dec ? MODE_DEC : MODE_ENC, iv, d_aad, aad_len, d_in, len, d_out, 0, st, &pc, &took, true))) return rc;
        if (took && pc.done) return AESGCM_OK;                   // the launch left the tag in d_tag and in the host slot
        if (took) {
            c->last_shape = AESGCM_LAUNCH_CYCLIC;
            CombineParams q = combine_with_items(plan_combine_tag(pc.ptr, pc.np, pc.kind, iv, aad_len, len, c->d_tag), pc.eA, pc.tail_item, pc.tail_blocks);
            q.ej0 = pc.ej0;
            return enqueue_combine(c, q, st);
        }
    }
    BodySplit b;
    if (ctx_body_split(c, len, 0, &b)) {
        c->last_shape = AESGCM_LAUNCH_DEALT;
        if (!aad_len && !b.head_blocks && len == 16 * b.body_blocks) {
            // the whole message is one aligned body (the benchmark's shape): no chaining value to carry, k_body's items go
            // straight to the tag
            Partials pb;
            FoldClose fc = {};
            if (c->fold_close) {                                 // k_fold's first level closes the tag (when there is a k_fold launch at all)
                fc.aad_len = 0; fc.ct_len = len; fc.ej0 = c->d_tag + 3; fc.acc = c->d_cyc;
                fc.tag_out = c->d_tag; fc.tag_host = c->h_tag_dev; fc.gen = gen_now(c) + 1;
            }
            if ((rc = enqueue_body(c, dec ? MODE_DEC : MODE_ENC, iv, b, d_in, d_out, 0, st, &pb, c->fold_close ? &fc : nullptr))) return rc;
            if (pb.done) { gen_take(c); return AESGCM_OK; }
            CombineParams q = combine_with_items(plan_combine_tag(pb.ptr, pb.np, pb.kind, iv, 0, len, c->d_tag), pb.eA);
            q.ej0 = pb.ej0;
            return enqueue_combine(c, q, st);
        }
        // large message: head / k_body / tail folded into a device-side chaining value, then the tag from it
        uint4 *state = c->d_tag + 2;
        HIPCHK(hipMemsetAsync(state, 0, 16, st));
        const uint4 *ej0 = nullptr;
        if ((rc = absorb_range(c, dec ? MODE_DEC : MODE_ENC, iv, d_aad, aad_len, d_in, len, d_out, 0, st, state, &ej0))) return rc;
        CombineParams q = plan_combine_final(state, iv, aad_len, len, c->d_tag);
        q.ej0 = ej0;                                             // left by k_body (every piece of this message writes the same value)
        return enqueue_combine(c, q, st);
    }
    Partials pp;
    c->last_shape = AESGCM_LAUNCH_MAIN;
    rc = enqueue_main(c, dec ? MODE_DEC : MODE_ENC, iv, d_aad, aad_len, d_in, len, d_out, 0, st, &pp, true);
    if (rc) return rc;
    if (pp.done) return AESGCM_OK;                               // single chunk: k_main's tail left the tag in d_tag and in the host slot
    CombineParams q = combine_with_items(plan_combine_tag(pp.ptr, pp.np, pp.kind, iv, aad_len, len, c->d_tag), pp.eA);
    q.ej0 = pp.ej0;                                              // same IV, same stream: k_main left E_K(IV || 1) behind
    return enqueue_combine(c, q, st);
}

// The tag of the last result enqueued for the host slot: the kernel stores it in pinned host memory and then publishes
// the generation number; the host polls that number for a short while (a kernel-completion interrupt costs ~10 us on
// this platform, a poll of coherent host memory well under one) and falls back to a stream synchronisation for long-
// running work or if anything went wrong.
// What has happened when this returns: a tag published from INSIDE a launch (k_body's cyclic rows, cyc_close; k_fold's closing, acc_arrive) is seen while that
// launch is still running, and this function does NOT wait for its end -- the stream is not synchronised.  Every byte of the result is in device memory all the
// same: the rows store through the L2 (global_store ... sc0 sc1, gstore16_wt / gstore*_wt_at, AESGCM_BODY_WT), each workgroup waits for the acknowledgement of
// its own stores (s_waitcnt vmcnt(0)) before it counts itself arrived, and the tag is published by the workgroup that counts the last arrival; the launch
// retires a few microseconds later.  examples/early_read.cpp (tests/test_gpu_cyclic.py) is the standing check: a copy ordered behind nothing reads the whole
// result the moment the tag is there.  Tags that come from k_combine or k_main's tail are published by the last kernel of the call.
int fetch_tag(aesgcm_ctx *c, hipStream_t st, uint8_t tag[16]) {
    const u64 want = gen_now(c);
    volatile u64 *gen = reinterpret_cast<volatile u64 *>(c->h_tag + 1);
    bool seen = false;
    struct timespec t0, t1;
    clock_gettime(CLOCK_MONOTONIC, &t0);
    for (u32 spin = 0;; ++spin) {                                 // poll for at most ~200 us, then block in the runtime
        if (__atomic_load_n(gen, __ATOMIC_ACQUIRE) == want) { seen = true; break; }
        if ((spin & 63u) == 63u) {
            clock_gettime(CLOCK_MONOTONIC, &t1);
            if ((t1.tv_sec - t0.tv_sec) * 1000000000L + (t1.tv_nsec - t0.tv_nsec) > c->poll_ns) break;
        }
#if defined(__x86_64__)
        __builtin_ia32_pause();
#endif
    }
    if (!seen) {
        HIPCHK(hipStreamSynchronize(st));
        // the stream the message was enqueued on has drained: its tag is there -- unless `st` is not that stream, or the launch failed after the number was taken
        if (__atomic_load_n(gen, __ATOMIC_ACQUIRE) != want) {
            snprintf(g_err, sizeof g_err, "the host slot shows generation %llu, not %llu: the stream passed is not the one the message was enqueued on", (unsigned long long)__atomic_load_n(gen, __ATOMIC_ACQUIRE), (unsigned long long)want);
            return AESGCM_ESTATE;
        }
    }
    memcpy(tag, c->h_tag, 16);
    return AESGCM_OK;
}

int ct_compare16(const uint8_t *a, const uint8_t *b) {
    unsigned d = 0;
    for (int i = 0; i < 16; i++) d |= (unsigned)(a[i] ^ b[i]);
    return d == 0;
}

int grow(unsigned char **p, size_t *cap, size_t need) {
    if (need <= *cap) return AESGCM_OK;
    if (*p) { hipError_t e = hipFree(*p); *p = nullptr; *cap = 0; if (e != hipSuccess) return hip_fail(e, "hipFree"); }
    size_t n = need < 4096 ? 4096 : need;
    hipError_t e = hipMalloc((void **)p, n);
    if (e == hipErrorOutOfMemory) return AESGCM_ENOMEM;
    if (e != hipSuccess) return hip_fail(e, "hipMalloc");
    *cap = n;
    return AESGCM_OK;
}


// Key material of a context from a key (or a pre-expanded schedule): aes_kexp, H, the H-power tables -- k_setup and k_setup_ptab on the context's stream, waited
// for.  The staging buffer for the key bytes belongs to the context (aesgcm_ctx_rekey comes through here without an allocation) and is wiped behind the kernels.
int ctx_load_key(aesgcm_ctx *c, const uint8_t *key, size_t key_len, int pre_nr) {
    hipError_t e;
    if (!c->d_keystage && (e = hipMalloc((void **)&c->d_keystage, 256)) != hipSuccess) return hip_fail(e, "hipMalloc");
    const size_t kb = pre_nr ? (size_t)16 * (pre_nr + 1) : key_len;
    e = hipMemcpyAsync(c->d_keystage, key, kb, hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess) {
        e = klaunch_setup(c->stream, c->km, c->tables, c->d_keystage, (int)key_len, pre_nr, (u32)c->G);
    }
    if (e == hipSuccess) e = hipMemsetAsync(c->d_keystage, 0, 256, c->stream);    // do not leave key bytes behind
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    if (e != hipSuccess) return hip_fail(e, "k_setup");
    c->nr = pre_nr ? pre_nr : (int)(key_len / 4 + 6);                               // only now: a load that failed leaves the context's round count with its old key material
    return AESGCM_OK;
}

int ctx_create_common(aesgcm_ctx **out, int device, const uint8_t *key, size_t key_len, int pre_nr) {
    if (!out || !key) return AESGCM_EARG;
    *out = nullptr;
    DeviceState *ds;
    int rc = device_state(device, &ds);
    if (rc) return rc;
    rc = set_lds_attrs(device, ds);
    if (rc) return rc;
    aesgcm_ctx *c = new (std::nothrow) aesgcm_ctx();
    if (!c) return AESGCM_ENOMEM;
    c->device = device;
    c->tables = ds->tables;
    c->nr = pre_nr ? pre_nr : (int)(key_len / 4 + 6);
    const int per_cu = 2;
    int G = per_cu * ds->n_cu;
    if (G > AESGCM_GMAX) G = AESGCM_GMAX;
    if (G < 1) G = 1;
    c->G = G;
    hipError_t e;
    if ((e = hipSetDevice(device)) != hipSuccess) { delete c; return hip_fail(e, "hipSetDevice"); }
    {   // a stream a destroyed context left behind, or a new one
        std::lock_guard<std::mutex> lk(g_mu);
        if (!ds->streams.empty()) { c->stream = ds->streams.back(); ds->streams.pop_back(); }
    }
    if (!c->stream && (e = hipStreamCreate(&c->stream)) != hipSuccess) { delete c; return hip_fail(e, "hipStreamCreate"); }
    if ((e = hipMalloc(&c->km, sizeof(KeyMaterial))) != hipSuccess ||
        (e = hipMalloc(&c->fold_a, sizeof(uint4) * 64 * FOLD_A_ITEMS)) != hipSuccess ||
        (e = hipMalloc(&c->fold_b, sizeof(uint4) * 64 * FOLD_B_ITEMS)) != hipSuccess ||
        (e = hipMalloc(&c->d_counter, 64 * (1 + 2 * AESGCM_NQ))) != hipSuccess ||
        (e = hipMemset(c->d_counter, 0, 64 * (1 + 2 * AESGCM_NQ))) != hipSuccess ||
        (e = hipMalloc(&c->d_cyc, 8 * (2 * CYC_ACC_SLOTS + 1))) != hipSuccess ||
        (e = hipMemset(c->d_cyc, 0, 8 * (2 * CYC_ACC_SLOTS + 1))) != hipSuccess ||
        (e = hipMalloc(&c->d_tag, sizeof(uint4) * 4)) != hipSuccess ||
        (e = hipHostMalloc((void **)&c->h_tag, 64, hipHostMallocMapped | hipHostMallocCoherent)) != hipSuccess ||
        (e = hipHostGetDevicePointer((void **)&c->h_tag_dev, c->h_tag, 0)) != hipSuccess ||
        (memset(c->h_tag, 0, 64), false) ||
        (e = hipMalloc(&c->d_trace, sizeof(u64) * 4 * AESGCM_GMAX)) != hipSuccess) { aesgcm_ctx_destroy(c); return hip_fail(e, "hipMalloc"); }
    if ((rc = ctx_load_key(c, key, key_len, pre_nr))) { aesgcm_ctx_destroy(c); return rc; }
    { std::lock_guard<std::mutex> lk(g_mu); g_ctxs.push_back(c); }
    *out = c;
    return AESGCM_OK;
}


// ---------------------------------------------------------------- host-pointer wrappers
int stage_in(aesgcm_ctx *c, const uint8_t *aad, size_t aad_len, const uint8_t *in, size_t len) {
    int rc;
    if ((rc = grow(&c->st_aad, &c->st_aad_cap, aad_len))) return rc;
    if ((rc = grow(&c->st_in, &c->st_in_cap, len))) return rc;
    if ((rc = grow(&c->st_out, &c->st_out_cap, len))) return rc;
    if (aad_len) HIPCHK(hipMemcpyAsync(c->st_aad, aad, aad_len, hipMemcpyHostToDevice, c->stream));
    if (len) HIPCHK(hipMemcpyAsync(c->st_in, in, len, hipMemcpyHostToDevice, c->stream));
    return AESGCM_OK;
}

// A message under way: Y <- 0 and the bookkeeping of aesgcm_ctx::StreamState (aesgcm_stream_begin, the pipelined path, aesgcm_stream_import build on this)
int stream_open(aesgcm_ctx *c, const uint8_t iv[12], int decrypt, hipStream_t st) {
    c->s = aesgcm_ctx::StreamState();
    memcpy(c->s.iv, iv, 12);
    c->s.active = true; c->s.dec = decrypt ? 1 : 0;
    HIPCHK(hipMemsetAsync(c->d_tag + 1, 0, 16, st));
    return AESGCM_OK;
}
// Y' = Y * H^nb ^ P(aad, data) on `st` (NULL: the context's stream).  large: the range may be of any size on device pointers (aesgcm_stream_update_dev) -- it takes the
// launch structure a shard of that size takes (cyclic rows, dealt chunks; absorb_range); else one k_main launch (chunks that came through the staging buffers).
int stream_absorb(aesgcm_ctx *c, const void *d_aad, u64 aad_len, const void *d_in, u64 len, void *d_out, u64 first_block, hipStream_t st, bool large) {
    if (!st) st = c->stream;
    const u64 nb = (aad_len + 15) / 16 + (len + 15) / 16;
    if (large) {
        const int rc = absorb_range(c, c->s.dec ? MODE_DEC : MODE_ENC, c->s.iv, d_aad, aad_len, d_in, len, d_out, first_block, st, c->d_tag + 1);
        if (rc) return rc;
        c->s.blocks += nb;
        return AESGCM_OK;
    }
    Partials pp;
    int rc = enqueue_main(c, c->s.dec ? MODE_DEC : MODE_ENC, c->s.iv, d_aad, aad_len, d_in, len, d_out, first_block, st, &pp);
    if (rc) return rc;
    c->s.blocks += nb;
    return enqueue_combine(c, combine_with_items(plan_combine_carry(pp.ptr, pp.np, pp.kind, c->d_tag + 1, nb), pp.eA), st);       // Y' = Y * H^nb ^ P
}


// ---------------------------------------------------------------- shapes of the packet kernels
#ifdef AESGCM_DEBUG_KNOBS
// Test / profiling builds only (libaesgcm_hip_dbg.so, -DAESGCM_DEBUG_KNOBS; include/aesgcm_debug.h): force the kernel shape the next launches take, so that every
// shape can be checked on inputs the host's own rule would give to another.  The product library has no such switch and reads no environment.
ForceShape g_force = {0, 0, 0, 0, 0, 0, 0};
extern "C" __attribute__((visibility("default"))) int aesgcm_debug_force_shape(const char *what, int value) {
    if (!what) return AESGCM_EARG;
    if (!strcmp(what, "pkt_lanes")) { if (value != 0 && value != 1 && value != 4 && value != 8 && value != 16 && value != 64) return AESGCM_EARG; g_force.pkt_lanes = value; }
    else if (!strcmp(what, "pkt_deal")) g_force.pkt_deal = value;
    else if (!strcmp(what, "batch_lanes")) { if (value != 0 && value != 8 && value != 16 && value != 64) return AESGCM_EARG; g_force.batch_lanes = value; }
    else if (!strcmp(what, "batch_deal")) g_force.batch_deal = value;
    else if (!strcmp(what, "pkt_ilp")) { if (value < 0 || value > 2) return AESGCM_EARG; g_force.pkt_ilp = value; }              // k_pktl's ILP form: 0 = the library's rule, 1 = always, 2 = never
    else if (!strcmp(what, "pkt_rows")) { if (value < 0 || value > 2) return AESGCM_EARG; g_force.pkt_rows = value; }            // aesgcm_packets_crypt_dev by rows (k_rows): 0 = the library's rule, 1 = always, 2 = never
    else if (!strcmp(what, "batch_order")) { if (value < 0 || value > 2) return AESGCM_EARG; g_force.batch_order = value; }      // variable-length batches by length class: 0 = the library's rule, 1 = always, 2 = never
    else return AESGCM_EARG;
    return AESGCM_OK;
}
#endif

// Packets under ONE key: how many lanes work on one packet, as log2 (0 = one LANE per packet, k_pktl; 2, 3, 4 = a lane GROUP of 4, 8, 16, k_pktg; 6 = a whole
// wave, k_pktg<.., 6>).  Measured (profiles/archive/r03/packets_sweep_aes256.txt, GiB/s wave / g16 / g8 / g4 / lane): the best shape is the one that just fills the
// resident lanes (256 CUs x 16 waves x 64) -- 65536 x 1 KiB 203 / 232 / 340 / 384 / 194, 16384 x 4 KiB 235 / 367 / 290 / 177 / 53, 4096 x 16 KiB
// 362 / 172 / 95 / 49 / 13 (the one regime where a whole wave per packet is right: at most 4096 packets of at least 4 KiB) -- but never more lanes than an
// eighth of the packet's blocks once the machine is full (closing cost per byte: 16384 x 1 KiB 62 / 128 / 176 / 138 / 50, 16384 x 256 B 16 / 35 / 58 / 72 / 41),
// a quarter when it is not (4096 x 1 KiB 34 / 69 / 57 / 38 / 13).  Lanes win from 131072 packets (2^20 x 1 KiB 303 / 592 / 657 / 742 / 767; 262144 x 4 KiB
// 496 / 656 / 704 / 722 / 724), short packets from 32768 (65536 x 256 B 51 / 61 / 95 / 129 / 148).
int packets_pick_lg(u32 n_cu, size_t n_pkts, size_t pkt_len) {
    const size_t lanes_total = (size_t)n_cu * (AESGCM_PKT_WG / 64) * 64, lanes_l = (size_t)n_cu * AESGCM_PKTL_WG;
    const size_t blocks = (pkt_len + 15) / 16;
    // One lane per packet once the packets fill k_pktl's resident lanes (256 x 768); frames of up to 1 KiB from three quarters of that, short ones much earlier.
    // Round 4 (profiles/r04/packets_sweep_aes256.txt, after k_pktl's rebuild): 131072 x 4 KiB 553 by lanes against 722 by groups of 4 (196608: 795 / 713),
    // 131072 x 16 KiB 573 / 789, 131072 x 1 KiB 488 / 509 (196608: 677 / 577), 49152 x 256 B 142 / 124, 16384 x 64 B 28 / 23.
    // (Offset arrays -- the host does not know the lengths -- are routed on the device since round 6: route_pick_lg, aesgcm_pkt.h, with the measurements behind it.)
    // k_pktl's ILP form (512-lane workgroups) moves the 1 KiB mark down: 131072 x 1 KiB 592 by lanes against 500 by groups of 4, 98304: 454 / 456.
    const size_t lanes_ilp = (size_t)n_cu * AESGCM_PKTL_WG_ILP;
    if (n_pkts >= lanes_l || (pkt_len <= 1024 && 8 * n_pkts >= 7 * lanes_ilp) || (pkt_len <= 256 && n_pkts >= 32768) || (pkt_len <= 64 && n_pkts >= 16384)) return 0;
    // Lane groups: the group that just fills the resident lanes.  Packets of 4 KiB and more round the fill UP to a power of two (half again as many lanes as
    // are resident is cheaper than rows twice as long: 49152 x 4 KiB 474 with 4 lanes, 576 with 8; x 16 KiB 542 / 722), shorter ones down (49152 x 1 KiB 325 / 291).
    size_t fill = lanes_total / n_pkts;
    if (pkt_len >= 4096 && (fill & (fill - 1))) { size_t f = 1; while (f < fill) f <<= 1; fill = f; }
    const size_t cap = n_pkts >= 16384 ? blocks / 8 : blocks / 4;
    const size_t g = fill < cap ? fill : cap;
    return g >= 64 ? 6 : g >= 16 ? 4 : g >= 8 ? 3 : 2;
}

// Packets with their OWN key (k_batch3): lanes per packet as log2 (3, 4, 6 = 8 / 16 lanes, a whole wave; the two-pass kernel k_batch of rounds 2 - 3 that
// the numbers below call by name is gone since round 4: k_batch3<.., 6> took its place, 4096 x 1 MiB 443 -> 637 GiB/s).  16 lanes once
// there are packets enough to fill the machine that way (one 1024-lane workgroup per CU = 64 packets per CU) or the packets are short, else one wave per packet.
// Measured, AES-128, GiB/s k_batch / k_batch3 (profiles/archive/r03/batch_sweep_aes128.txt): 4096 x 1 KiB 30 / 56, 4096 x 256 B 7.5 / 17, 1024 x 1 KiB 14 / 16.5; 1024 x 4 KiB
// 45 / 33, 4096 x 4 KiB 108 / 120, 4096 x 16 KiB 286 / 168; from 16384 packets k_batch3 wins at every size (4 KiB 179 / 350).  8 lanes (eight packets per wave
// share what a wave-iteration pays once) when there are packets enough to fill the chip that way and they are not long: 2^20 packets of 64 B 42 -> 74 GiB/s,
// 256 B 163 -> 265, 1 KiB 424 -> 560, 1500 B 484 -> 598, 4 KiB 658 -> 706, 16 KiB 770 -> 736; 16384 packets: 1 KiB 125 -> 155, 4 KiB 352 -> 273
// (profiles/archive/r03c/batch_sweep_lanes8_aes128.txt).  Batches with per-packet lengths (offset arrays on the device: the host does not know the lengths) go by count
// alone and assume frames of MACsec size, where 8 lanes gain most; a batch of frames beyond 8 KiB loses ~5 % by it.
int batch_pick_lg(int n_cu, size_t n_pkts, size_t pkt_len, bool var) {
    int lg = (n_pkts >= (size_t)64 * n_cu || (!var && pkt_len <= 2048)) ? 4 : 6;
    if (lg == 4 && (var ? n_pkts >= (size_t)64 * n_cu
                        : ((n_pkts >= (size_t)256 * n_cu && pkt_len <= 8192) || (n_pkts >= (size_t)64 * n_cu && pkt_len <= 2048)))) lg = 3;
    return lg;
}


// The order in which a launch takes packets of mixed length: counting sort by falling length class on the launch's stream (k_len_hist, k_len_scan,
// k_len_scatter).  *perm = NULL when it does not pay or is switched off.  Three launches of about 10 us in front of the packet kernel: mixed frames of
// 64 .. 1514 bytes, AES-256, best shape each (profiles/r04/packets_sweep_mixed_*.txt): 16384 frames 101 GiB/s in array order, 79 by class; 65536 223 / 199; 98304
// 256 / 271; 131072 284 / 320; 262144 382 / 429; 2^20 426 / 717 -- the order pays once the machine is full, and the default threshold is there.
int order_launch(OrderSlot &o, const u64 *d_off, size_t n_pkts, hipStream_t st, const u32 **perm) {
    if (!o.done) HIPCHK(hipEventCreateWithFlags(&o.done, hipEventDisableTiming));
    else HIPCHK(hipStreamWaitEvent(st, o.done, 0));                                // the slot's previous reader, on whatever stream it ran
    // no memory for the scratch (4 bytes per packet): the launch takes the packets as they come -- slower, never wrong
    if (!o.bins && hipMalloc((void **)&o.bins, LEN_SORT_ENTRIES * sizeof(u32)) != hipSuccess) { o.bins = nullptr; (void)hipGetLastError(); *perm = nullptr; return AESGCM_OK; }
    if (o.cap < n_pkts) {
        if (o.perm) { HIPCHK(hipFree(o.perm)); o.perm = nullptr; o.cap = 0; }     // hipFree waits for the launches that may still read it
        if (hipMalloc((void **)&o.perm, n_pkts * sizeof(u32)) != hipSuccess) { o.perm = nullptr; (void)hipGetLastError(); *perm = nullptr; return AESGCM_OK; }
        o.cap = n_pkts;
    }
    LenSrc src = {d_off, nullptr, nullptr, nullptr, 0u};                            // by data length (a batch packet's AAD is short)
    RouteCfg none = {nullptr, (u32)n_pkts, 0u, 0u, 0u, 0u, 0xFFu, 0u, 0, 0, 0, 0, 0, 0};    // a plain order: nothing is routed
    HIPCHK(klaunch_len_sort(st, src, (u32)n_pkts, o.bins, o.perm, none));
    *perm = o.perm;
    return AESGCM_OK;
}


// ---------------------------------------------------------------- many messages under the context's key: by rows (aesgcm_rows.h)
// the scratch of the path, carved out of one allocation: per message 16 + 4 bytes and (offset arrays) the three prefix sums, 32 bytes per record slot.  Zero at rest.
struct RowsScratch { RowsHdr *hdr; u32 *queues; u64 *prefix, *sprefix, *plan_part; u32 *slot_base; RowsRec *rec; unsigned long long *acc; u32 *cnt; u32 *perm, *bins; u64 *bad_part; PktDesc *desc; };

#define ROWS_DESC_MAX_N ((size_t)1 << 24)
size_t rows_carve(unsigned char *base, size_t slots, size_t n, RowsScratch *r) {
    size_t o = 0;
    auto take = [&](size_t bytes) { unsigned char *q = base ? base + o : nullptr; o += (bytes + 255) & ~(size_t)255; return q; };
    RowsScratch t;
    t.hdr = (RowsHdr *)take(sizeof(RowsHdr));
    t.queues = (u32 *)take(64 * ROWS_NQ);
    t.prefix = (u64 *)take(8 * (n + 1));
    t.sprefix = (u64 *)take(8 * (n + 1));
    t.plan_part = (u64 *)take(8 * 4 * (n / 1024 + 2));                   // the planner's sums per workgroup of 1024 messages (units, smalls, slots, first refused length)
    t.slot_base = (u32 *)take(4 * (n + 1));
    t.rec = (RowsRec *)take(sizeof(RowsRec) * slots);
    t.acc = (unsigned long long *)take(16 * n);
    t.cnt = (u32 *)take(4 * n);
    t.perm = (u32 *)take(4 * n);                                         // a routed call: the launch order of the messages that take the packet kernels (k_len_*)
    t.bins = (u32 *)take(4 * (size_t)LEN_SORT_ENTRIES);
    t.bad_part = (u64 *)take(8 * (size_t)LEN_SORT_WGS);                  // ... and the first length each slice of the sort found it cannot take
    t.desc = n <= ROWS_DESC_MAX_N ? (PktDesc *)take(sizeof(PktDesc) * n) : nullptr;      // ... and, for a lane per packet, the launch's packet records (48 bytes each: not for calls of more than 2^24 messages)
    if (r) *r = t;
    return o;
}

int rows_scratch(aesgcm_ctx *c, size_t slots, size_t n, hipStream_t st, RowsScratch *r) {
    if (slots > c->rows_cap_slots || n > c->rows_cap_n) {
        if (c->rows_buf) { HIPCHK(hipFree(c->rows_buf)); c->rows_buf = nullptr; c->rows_cap_slots = c->rows_cap_n = 0; }    // hipFree waits for the launches that may still use it
        const size_t cs = slots < 65536 ? 65536 : slots, cn = n < 4096 ? 4096 : n;
        const hipError_t e = hipMalloc((void **)&c->rows_buf, rows_carve(nullptr, cs, cn, nullptr));
        if (e == hipErrorOutOfMemory) return AESGCM_ENOMEM;
        if (e != hipSuccess) return hip_fail(e, "hipMalloc");
        c->rows_cap_slots = cs; c->rows_cap_n = cn; c->rows_dirty = true;
    }
    if (c->rows_dirty) HIPCHK(hipMemsetAsync(c->rows_buf, 0, rows_carve(nullptr, c->rows_cap_slots, c->rows_cap_n, nullptr), st));   // fresh scratch, or a launch failed half way through a call
    rows_carve(c->rows_buf, c->rows_cap_slots, c->rows_cap_n, r);
    return AESGCM_OK;
}

// The marks of a ROUTED call as length classes (64 bytes each) for k_len_scan: the context's "rows_min" (8 KiB) and a quarter of it -- the two marks packets_by_rows applies
// to fixed-size records, chosen between per call on the device by what lies between them (k_len_scan).  The classes resolve up to 16320 bytes.
static void route_marks(const aesgcm_ctx *c, u32 *c_hi, u32 *c_lo) {
    const u64 hi = c->rows_min / 64, lo = c->rows_min / 4 / 64;
    *c_hi = c->rows_min ? (u32)(hi < PKT_LEN_CLASSES ? hi : PKT_LEN_CLASSES - 1u) : PKT_LEN_CLASSES;      // rows_min = 0: never by rows
    *c_lo = c->rows_min ? (u32)(lo < PKT_LEN_CLASSES ? lo : PKT_LEN_CLASSES - 1u) : PKT_LEN_CLASSES;
#ifdef AESGCM_DEBUG_KNOBS
    if (g_force.pkt_rows == 1) *c_hi = *c_lo = 0;                                          // everything by rows
    else if (g_force.pkt_rows == 2 || g_force.pkt_lanes) *c_hi = *c_lo = PKT_LEN_CLASSES;  // everything through the packet kernels (a forced shape of theirs means them)
#endif
}

// p: the caller's pointers, counts and lengths; the cut and the scratch are filled in here.  k != NULL: the call is ROUTED per message (round 6; the lengths are on the
// device): *k describes the same call for the packet kernels, which take the messages below the mark k_len_scan chooses; the row launches see those as nothing.
int packets_rows(aesgcm_ctx *c, int decrypt, RowsParams &p, hipStream_t st, PktParams *k) {
    const size_t n = p.n_pkts;
    RowsScratch r;
    int rc;
    const bool var = p.data_off != nullptr || p.aad_off != nullptr || p.len_arr != nullptr;          // a length of any kind on the device: the plan is made there
    if (k && !var) return AESGCM_EARG;
    u32 wgs = (u32)c->G / 2;                                                 // one 141 KiB workgroup per CU
    const u32 n_cu = wgs;
    hipStream_t rows_st = st;                                                // where the row launches go: the caller's stream, or (a routed call) the side stream
    size_t slots;
    if (!var) {
        const RowsGeom g = rows_geom(p.pkt_len);
        const u32 na = rows_na(p.aad_len);
        p.U = rows_units(g, na); p.S = rows_smalls(g, na);
        p.G = (u64)n * p.U;
        const u64 need = (p.G + AESGCM_BODY_WG / 64 - 1) / (AESGCM_BODY_WG / 64);     // at least a unit per wave
        if (need < wgs) wgs = (u32)need;
        p.waves = wgs * (AESGCM_BODY_WG / 64);
        rows_cut(p.G, p.waves, c->rows_block, (u64)1 << 30, &p.D, &p.NB, &p.dyn);
        p.SM = p.U ? rows_nat_count(g, na) + (p.U - 1u) / p.D + 1u : 0u;
        if ((u64)n * p.SM >= (1ull << 31)) return AESGCM_ETOOLONG;
        slots = n * p.SM;
    } else {
        p.waves = wgs * (AESGCM_BODY_WG / 64);
        slots = ROWS_SLOTS_PER_MSG * n + ROWS_NB_CAP;                        // a run and a long AAD per message, and one more slot per block boundary inside its rows
        if (slots >= (1ull << 31)) return AESGCM_ETOOLONG;
    }
    if ((rc = rows_scratch(c, slots, n, st, &r))) return rc;
    p.slot_cap = (u32)slots;
    p.rec = r.rec; p.acc = r.acc; p.cnt = r.cnt; p.queues = r.queues;
    c->rows_dirty = true;                                                    // until both launches are enqueued
    if (var) {
        p.hdr = r.hdr; p.prefix = r.prefix; p.sprefix = r.sprefix; p.slot_base = r.slot_base;
        if (k) {
            // the route: a counting sort of the messages by falling size class (data + AAD) whose scan also decides -- which messages go by rows, how many are the packet
            // kernels', and in which shape (aesgcm_rows.h RowsHdr)
            LenSrc src = {p.data_off, p.aad_off, p.len_arr, p.alen_arr, p.aad_len};
            RouteCfg cfg = {r.hdr, (u32)n, n_cu, 0u, 0u, c->route_mid_min, 0xFFu, 0u, c->route_blocks_min, (u64)(uintptr_t)p.in_ptr, (u64)(uintptr_t)p.out_ptr, (u64)(uintptr_t)p.aad_ptr, (u64)(uintptr_t)p.len_arr, (u64)(uintptr_t)p.alen_arr};
            k->scattered = p.len_arr ? 1u : 0u;
            cfg.top_min = c->route_top_min;
            route_marks(c, &cfg.c_hi, &cfg.c_lo);
            const bool probe = decrypt == 2;                                 // aesgcm_frames_ceiling_probe_dev: the packet kernels' instruction stream without the data's traffic -- every message theirs, no row launch
            if (probe) { if (p.len_arr) return AESGCM_EARG; cfg.c_hi = cfg.c_lo = PKT_LEN_CLASSES; }
#ifdef AESGCM_DEBUG_KNOBS
            if (g_force.pkt_lanes) cfg.force_lg = g_force.pkt_lanes == 1 ? 0u : g_force.pkt_lanes == 64 ? (p.len_arr ? 4u : 6u) : g_force.pkt_lanes == 16 ? 4u : g_force.pkt_lanes == 8 ? 3u : 2u;      // (messages wherever they live have no wave-per-packet instance: 16 lanes)
            if (g_force.pkt_deal >= 1 && g_force.pkt_deal <= (int)PKTG_MAX_DEAL) cfg.force_deal = (u32)g_force.pkt_deal;
#endif
            // (the sort also checks every length: a call with one of 2^28 bytes or more is refused by its scan -- hdr->bad -- and every launch behind returns at once)
            const DescSrc ds = {r.desc, p.ivs, p.in_ptr, p.out_ptr, p.aad_ptr};                 // (r.desc is NULL beyond 2^24 messages: rows_carve)
            HIPCHK(klaunch_len_sort(st, src, (u32)n, r.bins, r.perm, cfg, r.bad_part, reinterpret_cast<u32 *>(c->h_tag_dev + 2), ds));
            k->perm = r.perm; k->route = r.hdr; k->desc = ds.desc; k->counter = &r.hdr->pkt_counter; k->counter_base = 0; k->plain = 0;
            k->n_pkts = (u32)n;
            // Behind the sort the call FORKS: the row launches (plan, k_rows, k_rows_close) go to the context's side stream, the packet kernels stay on the caller's, and
            // the caller's stream waits for the side stream at the end.  The two halves share nothing but the header the scan left (read-only from here on, except the
            // plan's own fields); a call of frames alone no longer waits for seven launches that find nothing to do.  (The two big launches do not share the chip: k_pktl and
            // k_rows each hold a CU with one 141 KiB workgroup, so k_rows runs when the packet launch ends -- profiles/r06/route_band.txt.)
            if (!probe && !c->side) {
                HIPCHK(hipStreamCreateWithFlags(&c->side, hipStreamNonBlocking));
                HIPCHK(hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming));
                HIPCHK(hipEventCreateWithFlags(&c->ev_join, hipEventDisableTiming));
            }
            if (!probe) {
                HIPCHK(hipEventRecord(c->ev_fork, st));
                HIPCHK(hipStreamWaitEvent(c->side, c->ev_fork, 0));
                rows_st = c->side;
                HIPCHK(klaunch_rows_plan(rows_st, p, true, c->rows_block, (u32)ROWS_NB_CAP, r.plan_part, reinterpret_cast<u32 *>(c->h_tag_dev + 2)));
            }
            // every shape the count of small messages -- anything up to n -- could ask for; all but the one k_len_scan named return before they stage a table
            const u32 lg_min = cfg.force_lg != 0xFFu ? cfg.force_lg : route_pick_lg(n_cu, n), lg_max = cfg.force_lg != 0xFFu ? cfg.force_lg : 4u;
            static const u32 shapes[] = {0u, 2u, 3u, 4u, 6u};
            for (u32 lg : shapes) {
                if (lg < lg_min || lg > lg_max || cfg.c_hi == 0u) continue;                  // (c_hi = 0: everything by rows, forced)
                if (lg == 0u) {
                    const u32 waves_per_wg = AESGCM_PKTL_WG / 64, nb = (u32)((n + 63) / 64);
                    u32 w = (nb + waves_per_wg - 1) / waves_per_wg;
                    if (w > n_cu) w = n_cu;
                    HIPCHK(klaunch_pktl(c->nr, decrypt, false, w, st, c->km, c->tables, *k));
                } else {
                    const u32 P = 64u >> lg, waves_per_wg = (u32)PKTG_WG(lg) / 64, nb = (u32)((n + P - 1) / P);
                    u32 w = (nb + waves_per_wg - 1) / waves_per_wg;
                    if (w > n_cu) w = n_cu;
                    HIPCHK(klaunch_pktg(c->nr, decrypt, (int)lg, w, st, c->km, c->tables, *k));
                }
            }
            if (probe) { c->rows_dirty = false; return AESGCM_OK; }           // (nothing of the row path ran: its scratch is at rest)
        } else {
            HIPCHK(klaunch_rows_plan(st, p, false, c->rows_block, (u32)ROWS_NB_CAP, r.plan_part, reinterpret_cast<u32 *>(c->h_tag_dev + 2)));
        }
    }
    p.prio_rows = c->cyc_prio;
    if (wgs) HIPCHK(klaunch_rows(c->nr, decrypt, wgs, rows_st, c->km, c->tables, p));                          // (fixed-size records of no bytes and no AAD have no units: their tags are the closing's alone)
    size_t close_lanes = p.slot_cap > n ? p.slot_cap : n;                                                    // a lane per record slot and per message; the lanes stride, so the grid is capped (and with offset arrays most slots of the worst case are never given out)
    if (close_lanes > (size_t)4096 * ROWS_CLOSE_WG) close_lanes = (size_t)4096 * ROWS_CLOSE_WG;
    HIPCHK(klaunch_rows_close(decrypt, (unsigned)((close_lanes + ROWS_CLOSE_WG - 1) / ROWS_CLOSE_WG), rows_st, c->km, c->tables, p));
    if (rows_st != st) { HIPCHK(hipEventRecord(c->ev_join, rows_st)); HIPCHK(hipStreamWaitEvent(st, c->ev_join, 0)); }      // the join
    c->rows_dirty = false;
    return AESGCM_OK;
}

// Does a call of FIXED-SIZE records go by rows?  (With offset arrays every message is routed by its own size on the device: k_len_scan applies the same two marks per
// message, aesgcm_rows.h.)  From rows_min bytes per packet (8 KiB), and from a quarter of that while the packets are few: the packet kernels need a packet per lane
// (or per lane group) to fill the chip, the rows of a call fill it whatever the count.  Measured on one box with the smalls in the closing launch, AES-256,
// GiB/s by rows / by the packet kernels (profiles/r05/rows_min_sweep2.txt, rows_min_sweep3.txt): 32 KiB x 131072 885 / 721, + 16 bytes 870 / 666; 16 KiB x 262144
// 845 / 792, + 16 827 / 749, x 16384 634 / 563, x 4096 571 / 400; 8 KiB x 524288 770 / 727, + 16 746 / 695, x 32768 596 / 545, x 4096 464 / 254, x 1024 189 / 129;
// 6 KiB x 699050 709 / 802, x 65536 613 / 613, x 8192 501 / 282; 4 KiB x 2^20 620 / 810, x 131072 557 / 667, x 32768 495 / 485, x 16384 502 / 384, x 4096 339 / 151;
// 2 KiB x 2^20 429 / 804, x 16384 352 / 242, x 4096 206 / 115; 1 KiB x 262144 264 / 616, x 4096 116 / 83.
bool packets_by_rows(const aesgcm_ctx *c, size_t n_pkts, size_t pkt_len) {
#ifdef AESGCM_DEBUG_KNOBS
    if (g_force.pkt_rows) return g_force.pkt_rows == 1;
    if (g_force.pkt_lanes) return false;                                     // a forced shape of the packet kernels means the packet kernels
#endif
    if (!c->rows_min) return false;
    if (n_pkts <= 16384) return 4 * pkt_len >= c->rows_min;
    if (pkt_len < c->rows_min) return false;
    // many packets of 8 .. 16 KiB whose last, partial row has more than a few blocks: what is not a whole row costs a message about a row and a half either way (a
    // pass of a wave in the row launch, or a lane per block of the closing launch), and the lane-per-packet kernel keeps them -- 262 144 x 9000 bytes (8 rows + 51
    // blocks) 613 by rows, 768 there; x 8448 (8 rows + 16 blocks) 696 / 821; from 16 KiB rows win again: 131 072 x 16 656 744 / 682
    // (profiles/r05/rows_ragged_many.txt)
    if (pkt_len < 2 * c->rows_min && rows_geom(pkt_len).tb > ROWS_FEW_TAIL) return false;
    return true;
}

// zero the output of every packet whose d_auth[] entry is 0 (behind the launch that wrote it, on the same stream)
int wipe_failed(int device, size_t n_pkts, void *d_out, size_t pkt_len, const u64 *d_data_off, const int *d_auth, hipStream_t st, const u64 *d_out_ptr, const u32 *d_len) {
    if (!n_pkts || !d_auth || (!d_out && !d_len)) return AESGCM_OK;
    HIPCHK(hipSetDevice(device));
    HIPCHK(klaunch_wipe_failed(st, (unsigned char *)d_out, d_auth, d_data_off, (u32)n_pkts, (u32)pkt_len, d_out_ptr, d_len));
    return AESGCM_OK;
}


// ---------------------------------------------------------------- batch (per-packet key and IV)
// Variable-length batches by length class: mixed frames of 64 .. 1514 bytes, GiB/s in array order / by class (profiles/r04/batch_mixed_*.txt): AES-128 65536 packets
// 202 / 178, 262144 313 / 307, 393216 332 / 342, 2^20 362 / 493; AES-256 65536 177 / 167, 98304 205 / 213, 262144 266 / 290, 2^20 301 / 437.
#define BATCH_ORDER_MIN(nr) ((nr) == 10 ? 262144u : 98304u)
int batch_launch(int device, int decrypt, size_t n_pkts, size_t key_len, BatchParams &p, void *stream) {
    if (key_len != 16 && key_len != 24 && key_len != 32) return AESGCM_EKEYLEN;
    if (n_pkts >= (((size_t)1) << 31)) return AESGCM_ETOOLONG;
    DeviceState *ds;
    int rc = device_state(device, &ds);
    if (rc) return rc;
    if ((rc = set_lds_attrs(device, ds))) return rc;
    HIPCHK(hipSetDevice(device));
    p.n_pkts = (u32)n_pkts;
    u32 wgs = 0;
    {   // a fresh dispenser per launch (zeroed on the launch stream), so launches on different streams may overlap
        std::lock_guard<std::mutex> lk(g_mu);
        p.counter = ds->batch_counter + (ds->batch_slot++ % BATCH_DISPENSERS);
        p.counter_base = 0;
    }
    const int nr = (int)(key_len / 4 + 6);
    hipStream_t st = (hipStream_t)stream;
    HIPCHK(hipMemsetAsync(p.counter, 0, 4, st));
    int lg = batch_pick_lg(ds->n_cu, n_pkts, p.pkt_len, p.data_off != nullptr);          // k_batch3 with 8 / 16 / 64 lanes per packet
#ifdef AESGCM_DEBUG_KNOBS
    if (g_force.batch_lanes) lg = g_force.batch_lanes == 8 ? 3 : g_force.batch_lanes == 16 ? 4 : 6;
#endif
    if (decrypt == 2 && lg != 3) { snprintf(g_err, sizeof g_err, "the probe of the batch kernel exists in the 8-lanes-per-packet shape; this call takes %d", 1 << lg); return AESGCM_EARG; }
    if (lg <= 6) {
        // packets of mixed length: by falling length class once the batch fills the machine several times over (BATCH_ORDER_MIN; as aesgcm_packets_crypt_dev)
        OrderSlot *oslot = nullptr;
        bool ordered = lg < 6 && p.data_off && n_pkts >= BATCH_ORDER_MIN(nr);
#ifdef AESGCM_DEBUG_KNOBS
        if (g_force.batch_order) ordered = lg < 6 && p.data_off && g_force.batch_order == 1;
#endif
        std::unique_lock<std::mutex> order_lock(g_mu, std::defer_lock);             // held from the choice of the slot to the event behind its reader: callers on other threads queue up here
        if (ordered) {
            order_lock.lock();
            oslot = &ds->order[ds->order_next++ & 3u];
            if ((rc = order_launch(*oslot, p.data_off, n_pkts, st, &p.perm))) return rc;
        }
        p.plain = !p.data_off && !p.aad_off && !p.aad_len && p.aligned && p.pkt_len && p.pkt_len % (16u << lg) == 0;
        const u32 waves_per_wg = (u32)BATCH3_LANES(nr) / 64;
        const u32 P = 64u >> lg, per_wg = waves_per_wg * P;
        wgs = (u32)((n_pkts + per_wg - 1) / per_wg);
        if (wgs > (u32)ds->n_cu) wgs = (u32)ds->n_cu;
        u32 deal = (u32)(n_pkts / ((size_t)wgs * waves_per_wg * 16));
        deal = deal < P ? P : deal > 8 * P ? 8 * P : (deal + P - 1) / P * P;
#ifdef AESGCM_DEBUG_KNOBS
        if (g_force.batch_deal >= 1 && g_force.batch_deal <= 4096) deal = ((u32)g_force.batch_deal + P - 1) / P * P;
#endif
        p.deal = deal;
        HIPCHK(klaunch_batch3(nr, decrypt, lg, wgs, st, ds->tables, p));
        if (oslot && p.perm) HIPCHK(hipEventRecord(oslot->done, st));
        return AESGCM_OK;
    }
    return AESGCM_EARG;                                         // batch_pick_lg gives 3, 4 or 6
}


// ---------------------------------------------------------------- pipelined host-buffer path
// H2D of chunk k+1, the fused kernel on chunk k and D2H of chunk k-1 overlap on three streams; the GHASH
// value is carried from chunk to chunk on the device (Y' = Y*H^blocks ^ P, the same combine the beat-by-beat
// interface uses), so the result is bit-identical to one launch over the whole message.
void pipeline_release(aesgcm_ctx *c) {
    for (int i = 0; i < 2; i++) {
        if (c->pl_buf[i]) { hipFree(c->pl_buf[i]); c->pl_buf[i] = nullptr; }
        if (c->pl_ev_h2d[i]) { hipEventDestroy(c->pl_ev_h2d[i]); c->pl_ev_h2d[i] = nullptr; }
        if (c->pl_ev_k[i]) { hipEventDestroy(c->pl_ev_k[i]); c->pl_ev_k[i] = nullptr; }
        if (c->pl_ev_d2h[i]) { hipEventDestroy(c->pl_ev_d2h[i]); c->pl_ev_d2h[i] = nullptr; }
    }
    if (c->pl_in) { hipStreamDestroy(c->pl_in); c->pl_in = nullptr; }
    if (c->pl_out) { hipStreamDestroy(c->pl_out); c->pl_out = nullptr; }
    c->pl_cap = 0;
}

// all or nothing: either both streams, all six events and both chunk slots of `chunk` bytes exist afterwards, or none
// of them does (pl_in == NULL, pl_cap == 0) and the next call starts from scratch
int pipeline_prepare(aesgcm_ctx *c, size_t chunk) {
    hipError_t e = hipSuccess;
    if (!c->pl_in) {
        if (e == hipSuccess) e = hipStreamCreateWithFlags(&c->pl_in, hipStreamNonBlocking);
        if (e == hipSuccess) e = hipStreamCreateWithFlags(&c->pl_out, hipStreamNonBlocking);
        for (int i = 0; i < 2 && e == hipSuccess; i++) {
            e = hipEventCreateWithFlags(&c->pl_ev_h2d[i], hipEventDisableTiming);
            if (e == hipSuccess) e = hipEventCreateWithFlags(&c->pl_ev_k[i], hipEventDisableTiming);
            if (e == hipSuccess) e = hipEventCreateWithFlags(&c->pl_ev_d2h[i], hipEventDisableTiming);
        }
        if (e != hipSuccess) { pipeline_release(c); return hip_fail(e, "pipeline streams/events"); }
    }
    if (chunk > c->pl_cap) {
        c->pl_cap = 0;
        for (int i = 0; i < 2 && e == hipSuccess; i++) {
            if (c->pl_buf[i]) { e = hipFree(c->pl_buf[i]); c->pl_buf[i] = nullptr; }
            if (e == hipSuccess) e = hipMalloc((void **)&c->pl_buf[i], chunk);
        }
        if (e != hipSuccess) {
            pipeline_release(c);
            return e == hipErrorOutOfMemory ? AESGCM_ENOMEM : hip_fail(e, "pipeline chunk slots");
        }
        c->pl_cap = chunk;
    }
    return AESGCM_OK;
}


int crypt_pipelined(aesgcm_ctx *c, int dec, const uint8_t iv[12], const uint8_t *aad, size_t aad_len,
                           const uint8_t *in, size_t len, uint8_t *out, uint8_t tag[16], size_t chunk) {
    int rc = check_lengths(aad_len, len);
    if (rc) return rc;
    // the chunk-to-chunk GHASH value lives in the streaming slot (d_tag[1], s_iv, s_dec): refuse to run inside an open
    // stream_begin .. stream_final session instead of silently corrupting its running GHASH
    if (c->s.active) return AESGCM_ESTATE;
    if (!chunk) chunk = (size_t)64 << 20;
    chunk = (chunk + 1023) / 1024 * 1024;                    // whole rows, 16-byte aligned chunk starts
    if (chunk > len) chunk = (len + 1023) / 1024 * 1024;
    if (!chunk) chunk = 1024;
    HIPCHK(hipSetDevice(c->device));
    if ((rc = pipeline_prepare(c, chunk))) return rc;
    // state Y <- 0, then the AAD (small; through the staging buffer on the compute stream): the same StreamState the beat-by-beat interface keeps, open for the
    // length of this call
    if ((rc = stream_open(c, iv, dec, c->stream))) return rc;
    struct Close { aesgcm_ctx *c; ~Close() { c->s.active = false; } } close_on_return{c};
    if (aad_len) {
        if ((rc = stage_in(c, aad, aad_len, nullptr, 0))) return rc;
        if ((rc = stream_absorb(c, c->st_aad, aad_len, c->st_in, 0, c->st_out, 0))) return rc;
    }
    const size_t n_chunks = (len + chunk - 1) / chunk;
    for (size_t k = 0; k < n_chunks; k++) {
        const int s = (int)(k & 1);
        const size_t off = k * chunk, m = (len - off < chunk) ? len - off : chunk;
        if (k >= 2) HIPCHK(hipStreamWaitEvent(c->pl_in, c->pl_ev_d2h[s], 0));     // slot free again
        HIPCHK(hipMemcpyAsync(c->pl_buf[s], in + off, m, hipMemcpyHostToDevice, c->pl_in));
        HIPCHK(hipEventRecord(c->pl_ev_h2d[s], c->pl_in));
        HIPCHK(hipStreamWaitEvent(c->stream, c->pl_ev_h2d[s], 0));
        if ((rc = stream_absorb(c, nullptr, 0, c->pl_buf[s], m, c->pl_buf[s], off / 16))) return rc;   // in place
        HIPCHK(hipEventRecord(c->pl_ev_k[s], c->stream));
        HIPCHK(hipStreamWaitEvent(c->pl_out, c->pl_ev_k[s], 0));
        HIPCHK(hipMemcpyAsync(out + off, c->pl_buf[s], m, hipMemcpyDeviceToHost, c->pl_out));
        HIPCHK(hipEventRecord(c->pl_ev_d2h[s], c->pl_out));
    }
    if ((rc = enqueue_combine(c, plan_combine_final(c->d_tag + 1, iv, aad_len, len, c->d_tag), c->stream))) return rc;
    if ((rc = fetch_tag(c, c->stream, tag))) return rc;
    HIPCHK(hipStreamSynchronize(c->pl_out));
    return AESGCM_OK;
}
