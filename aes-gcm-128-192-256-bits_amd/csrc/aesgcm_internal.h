// aesgcm_internal.h -- what the three translation units of libaesgcm_hip.so share (round 5: the library was one 2 900-line file until then).
//
//   aesgcm_kernels.hip   the kernels and, at its end, the LAUNCHERS: one plain function per kernel family (klaunch_*) that picks the template instance and launches
//                        it.  Nothing outside that file names a kernel, so the other two units hold no device code at all -- the host side builds and runs against
//                        a fake HIP runtime on a machine without a GPU (tests/fake_hip: which device is current at every allocation, stream, event and launch).
//   aesgcm_host.hip      the host runtime: contexts and per-device state, the launch planners (which launches a message takes), the shape rules of the packet
//                        paths, the scratch of the row path, the pipelined host-buffer path.
//   aesgcm_abi.hip       the C ABI of include/aesgcm.h: argument checks and the calls into the runtime.
//   aesgcm_comm.hip      the inter-GPU exchange (RCCL through dlopen), as before.
#pragma once
#include "aesgcm_dev.h"
#include "aesgcm_rows.h"
#include "../../include/aesgcm.h"

#include <mutex>
#include <vector>

// ---------------------------------------------------------------- launch geometry of the kernels (workgroup sizes, LDS bytes): the planners size their grids by these
#ifndef AESGCM_WAVES_PER_SIMD
#define AESGCM_WAVES_PER_SIMD (2 * AESGCM_MAIN_WG / 256)   /* two workgroups per CU */
#endif
#if AESGCM_T4
#ifndef AESGCM_BODY_WG_T4
#define AESGCM_BODY_WG_T4 1024               /* lanes of k_body's workgroup (the cyclic rows and their closing need 1024; 768 = 3 waves per SIMD was the round-4 energy A/B, profiles/r04/energy_ab.txt) */
#endif
#define AESGCM_BODY_WG AESGCM_BODY_WG_T4     /* one workgroup per CU (136 KiB of LDS), 4 waves per SIMD, 128 registers */
#define AESGCM_BODY_WPS ((AESGCM_BODY_WG + 255) / 256)
#define AESGCM_BODY_LDS AESGCM_LDS_BYTES_T4
#else
#define AESGCM_BODY_WG AESGCM_MAIN_WG
#define AESGCM_BODY_WPS AESGCM_WAVES_PER_SIMD
#define AESGCM_BODY_LDS AESGCM_LDS_BYTES
#endif
#define AESGCM_BODYH_WG 512
#ifdef BATCH3_WG
#define BATCH3_LANES(NR) BATCH3_WG
#else
#define BATCH3_LANES(NR) AESGCM_WG
#endif
#ifdef AESGCM_PKTG_WG
#define PKTG_WG(LG) AESGCM_PKTG_WG
#else
#define PKTG_WG(LG) ((LG) == 6 ? 768 : AESGCM_PKT_WG)
#endif
#define PKTG_WAVE_SLOT 1280u                                                                                /* per wave: 64 E_K(J0) values and the 64 packet numbers of its dispenser block */
#define PKTG_LDS_TOTAL(LG) (PKTG_LDS_BYTES(LG) + ((LG) <= 4 ? (u32)(PKTG_WG(LG) / 64) * PKTG_WAVE_SLOT : 0u))
#define AESGCM_PKTL_LDS (AESGCM_PKTL_T4 ? AESGCM_LDS_BYTES_T4 : AESGCM_LDS_BYTES)
#ifndef AESGCM_PKTL_WAVES
#define AESGCM_PKTL_WAVES ((AESGCM_PKTL_WG + 255) / 256)          // waves per SIMD the register budget is sized for (one workgroup per CU)
#endif
#define AESGCM_PKTL_WG_ILP 512
#define LEN_SORT_WGS 256u
#define LEN_SORT_ENTRIES (PKT_LEN_CLASSES * LEN_SORT_WGS)
#define COMBINE_BATCH_MAX 8
struct CombineBatch { CombineParams p[COMBINE_BATCH_MAX]; };

// ---------------------------------------------------------------- launchers (aesgcm_kernels.hip); every one returns hipGetLastError() of its launch(es)
hipError_t klaunch_set_attributes();                                          // hipFuncSetAttribute(MaxDynamicSharedMemorySize) of every instance, on the current device
hipError_t klaunch_init_tables(DevTables *t);
hipError_t klaunch_setup(hipStream_t st, KeyMaterial *km, const DevTables *tb, const uint8_t *d_key, int key_len, int pre_nr, u32 G);      // k_setup + k_setup_ptab
hipError_t klaunch_gfmul(const uint4 *h, const uint4 *x, uint4 *z, size_t n);
hipError_t klaunch_copy16(hipStream_t st, uint4 *dst, const uint4 *src, u64 n16);
hipError_t klaunch_fill_splitmix64(hipStream_t st, unsigned blocks, u64 *buf, size_t n_words, size_t tail_bytes, u64 seed, u64 first_word);
hipError_t klaunch_main(int mode, int nr, unsigned wgs, hipStream_t st, const KeyMaterial *km, const DevTables *tb, const MainParams &p);
hipError_t klaunch_body(int mode, int nr, bool cyc, bool half, unsigned wgs, hipStream_t st, const KeyMaterial *km, const DevTables *tb, const BodyParams &p);
hipError_t klaunch_fold(unsigned wgs, bool closing, hipStream_t st, const KeyMaterial *km, const FoldParams &p);
hipError_t klaunch_combine(hipStream_t st, const KeyMaterial *km, const DevTables *tb, const CombineParams &p);
hipError_t klaunch_combine_batch(unsigned n, hipStream_t st, const KeyMaterial *km, const DevTables *tb, const CombineBatch &b);
hipError_t klaunch_pktl(int nr, int dec, bool ilp, unsigned wgs, hipStream_t st, const KeyMaterial *km, const DevTables *tb, const PktParams &p);
hipError_t klaunch_pktg(int nr, int dec, int lg, unsigned wgs, hipStream_t st, const KeyMaterial *km, const DevTables *tb, const PktParams &p);
hipError_t klaunch_batch3(int nr, int dec, int lg, unsigned wgs, hipStream_t st, const DevTables *tb, const BatchParams &p);
hipError_t klaunch_len_sort(hipStream_t st, const LenSrc &src, u32 n, u32 *bins, u32 *perm, const RouteCfg &rc, u64 *bad_part = nullptr, u32 *host_status = nullptr, const DescSrc &ds = DescSrc{});                           // k_len_hist, k_len_scan (+ the route of the call), k_len_scatter
hipError_t klaunch_rows_plan(hipStream_t st, const RowsParams &p, bool routed, u32 force_d, u32 nb_cap, u64 *part, u32 *host_status);      // p: the lengths' arrays, n_pkts, waves, slot_cap, and the scratch arrays the plan fills (hdr, prefix, sprefix, slot_base)
hipError_t klaunch_rows(int nr, int dec, unsigned wgs, hipStream_t st, const KeyMaterial *km, const DevTables *tb, const RowsParams &p);
hipError_t klaunch_rows_close(int dec, unsigned wgs, hipStream_t st, const KeyMaterial *km, const DevTables *tb, const RowsParams &p);
hipError_t klaunch_wipe_failed(hipStream_t st, unsigned char *out, const int *auth, const u64 *data_off, u32 n_pkts, u32 pkt_len, const u64 *out_ptr = nullptr, const u32 *len_arr = nullptr);

// ---------------------------------------------------------------- host runtime (aesgcm_host.hip)
extern thread_local char g_err[256];
int hip_fail(hipError_t e, const char *what);
#define HIPCHK(call) do { hipError_t _e = (call); if (_e != hipSuccess) return hip_fail(_e, #call); } while (0)
#define FOLD_CLOSE_MAX_WGS 512u

#define BATCH_DISPENSERS 256
// The launch order of packets of mixed length (k_len_*): scratch of one launch.  `done` is recorded behind the packet kernel that reads the order, and the next
// user of the slot makes its stream wait for it: slots may be reused by launches on other streams at any rate.
struct OrderSlot { u32 *perm = nullptr; size_t cap = 0; u32 *bins = nullptr; hipEvent_t done = nullptr; };
// `streams`: the streams of destroyed contexts, for the next context of the device -- hipStreamCreate takes 2 ms and hipStreamDestroy half a millisecond on this
// runtime (profiles/microbench/runtime_costs.cpp), more than everything else a context costs together (k_setup: 0.4 ms).
struct DeviceState { DevTables *tables = nullptr; int n_cu = 0; bool attrs = false; u32 *batch_counter = nullptr; u32 batch_slot = 0; OrderSlot order[4]; unsigned order_next = 0;
                     std::vector<hipStream_t> streams; };   // ring of dispensers: concurrent batch launches never share one
extern std::mutex &g_mu;
extern std::vector<DeviceState> &g_dev;

struct aesgcm_ctx {
    int device = 0;
    int nr = 0;
    int G = 0;                         // workgroups per full launch
    DevTables *tables = nullptr;
    KeyMaterial *km = nullptr;
    uint4 *parts = nullptr;            // one item (64 lane accumulators) per chunk, grown on demand
    size_t parts_cap = 0;              // items
    uint4 *fold_a = nullptr, *fold_b = nullptr;   // k_fold ping-pong: MAX_CHUNKS/128 items; the second level leaves at most max(MAX_CHUNKS/65536, COMBINE_MAX_ITEMS) (fold_group)
    u32 *d_counter = nullptr;          // chunk dispenser
    u32 counter_base = 0;              // value the packet dispenser (d_counter[0]) holds before the next launch
    u32 qset = 0;                      // which of the two sets of chunk queues (d_counter[16 (1 + 16 set + q)]) the next dynamic launch of k_main / k_body uses; that launch zeroes the other set
    u32 tw_override = 0;               // option "tw": rows per chunk of the dealt kernels, 0 = the library's rule (main_geometry)
    u64 body_min = (u64)256 << 20;     // ranges with an aligned middle of at least this many bytes go through k_body (option "body_min").  Since k_main
                                       // got cheaper below 256 MiB (dispensers, k_fold: profiles/archive/r02f/split_threshold.txt) the cut pays from 256 MiB:
                                       // 128 MiB -16 %, 256 MiB +0.8 %, 512 MiB +4 %, 1 GiB +11 %, 2 GiB +7 %; it was 128 MiB before, 0.7 GiB in round 1
    long poll_ns = 200000L;            // how long fetch_tag polls the host slot before it blocks in the runtime (option "poll_us")
    unsigned long long *d_cyc = nullptr;   // the accumulators and the arrival counter of the fused closing of a cyclic launch (zero between launches)
    // The tag of a fused cyclic launch appears while the launch is still running, and the call's contract is that the ciphertext is in memory by then.  Three ways were
    // built and measured in round 3 (profiles/archive/r03c/cyc_end.txt, us per message at 64 KiB / 16 MiB): the rows store THROUGH the L2 (sc0 sc1), so no line is left dirty --
    // 24 / 40, what ships (AESGCM_BODY_WT); every workgroup writes its XCD's L2 back before it counts itself arrived -- 29 / 46 (what a -DAESGCM_BODY_WT=0 build does);
    // the host waits for the end of the launch behind the tag -- 38 / 54 (deleted in round 4 with the run-time switch between the three).
    bool fold_close = true;            // whole messages through the dealt k_body: k_fold's first level closes the tag (FoldClose; option "fold_close" 0: further levels and k_combine)
    u32 cyc_prio = 2;                  // rows between rotations of the waves' issue priorities in a cyclic launch (body_prio; option "cyc_prio", 0 = off).  Without it the oldest wave of
                                       // every SIMD runs ahead and the youngest finishes alone: 256 MiB 321 -> 291 us, 1 GiB 1238 -> 1105 (dealt chunks: 1090), profiles/archive/r03c/cyc_prio_*.txt
    int cyc_half = 2;                  // option "cyc_half": whole messages below cyc_half_max bytes take the HALF shape of the cyclic rows (k_bodyh: 256 workgroups of 512 lanes, two per
                                       // CU) -- for callers that keep two or more messages in flight on contexts of their own, where one message's staging and closing then run
                                       // beside another's rows; alone on the chip the half shape is slower than the full one.  0 = never, 1 = always, 2 (default) = when another
                                       // context of the device has a message under way at the moment of the call (others_in_flight)
    u64 cyc_half_max = (u64)80 << 20;  // sustained GiB/s, AES-256, full shape with 2 in flight / half shape with 3 (profiles/r04/inflight_threshold.txt): 8 MiB 409 / 562, 24 MiB 677 / 797,
                                       // 32 MiB 730 / 816, 48 MiB 810 / 836, 64 MiB 828 / 846, 96 MiB 870 / 862, 128 MiB 877 / 867 -- the two-table round costs what the overlap buys from there
    bool cyc_fuse = true;              // whole messages: the cyclic launch closes the tag itself (option "cyc_close" 0: k_fold + k_combine behind it, as for shards and streaming chunks)
    // Which ranges go through k_body as cyclic rows (body_cyc_lane: one launch for AAD, data and ragged end, no dispenser, 4096 items whatever the size).  options "cyc_min" / "cyc_max"
    // (bytes; both 0 = never); needs one k_body workgroup per CU on 256 CUs.  Whole messages close their tag inside the launch (cyc_close): 24 us from 16 KiB to 2 MiB where
    // k_main + k_fold + k_combine take 27 (64 KiB) .. 39 (256 KiB) .. 34 (1 MiB), profiles/archive/r03c/cyc_small.txt -- from 64 KiB.  Shards and streaming chunks keep k_fold + k_combine
    // behind the launch and start at 4 MiB (2 MiB: 35 -> 37 us, 4 MiB: 39 -> 38).  The upper end: with the waves' priorities rotating (cyc_prio) equal shares hold up to about
    // 1 GiB -- AES-256, us per message, dealt chunks / cyclic rows: 512 MiB 577 / 555, 768 MiB 824 / 818, 896 MiB 970 / 932, 1 GiB 1069 / 1099, 1.25 GiB 1370 / 1381
    // (profiles/archive/r03c/cyc_prio_fine_*.txt); a range with pieces around its body costs the dealt form a launch pair per piece (+45 .. 80 us), so those stay cyclic a little longer.
    u64 cyc_min_fused = (u64)64 << 10, cyc_min = (u64)4 << 20;
    u64 cyc_max = (u64)1 << 30, cyc_max_fused = (u64)1 << 30, cyc_max_pieces = (u64)1280 << 20;
    uint4 *h_tag = nullptr;            // 64 bytes of pinned, device-mapped host memory: k_combine leaves the tag here too, so fetching it
    uint4 *h_tag_dev = nullptr;        //   is a host read -- no copy kernel, no interrupt-driven stream wait (its device address)
    u64 tag_gen = 0;                   // generation number of the last result sent to the host slot (the kernel publishes it behind the tag)
    uint4 *h_mtag = nullptr, *h_mtag_dev = nullptr;   // COMBINE_BATCH_MAX slots of {tag, generation} in pinned host memory (batched finalize); created on first use
    uint4 *d_mtag = nullptr;
    uint4 *d_tag = nullptr;            // [0] tag / poly result, [1] streaming state Y
    int last_shape = AESGCM_LAUNCH_NONE;   // which launch structure the context's last whole-message call took (aesgcm_ctx_last_launch)
    bool wipe_on_auth_fail = false;    // option "wipe_on_auth_fail": decrypt calls that verify a tag zero the output of what fails (the reference's model returns the plaintext and raises: default off)
    u64 *d_trace = nullptr;            // per-workgroup trace of the last k_main launch (timing mode only)
    u32 last_np = 0;
    uint8_t *d_keystage = nullptr;     // 256 bytes: where a key (or schedule) waits for k_setup; zeroed behind it
    hipStream_t stream = nullptr;
    hipEvent_t ev_sync = nullptr;      // aesgcm_ctx_wait: marks "everything enqueued so far on this context's stream"
    hipEvent_t ev_fused = nullptr;     // aesgcm_ctx_wait_fused: recorded behind every fused-kernel launch once somebody has asked for it
    // host-API staging
    unsigned char *st_in = nullptr, *st_out = nullptr, *st_aad = nullptr;
    size_t st_in_cap = 0, st_out_cap = 0, st_aad_cap = 0;
    // pipelined host path: two device chunk slots, copy streams and events
    unsigned char *pl_buf[2] = {nullptr, nullptr};
    size_t pl_cap = 0;
    hipStream_t pl_in = nullptr, pl_out = nullptr;
    hipEvent_t pl_ev_h2d[2] = {nullptr, nullptr}, pl_ev_k[2] = {nullptr, nullptr}, pl_ev_d2h[2] = {nullptr, nullptr};
    // many messages through the row kernel (k_rows, aesgcm_rows.h): one block of device scratch, grown on demand
    hipStream_t side = nullptr;        // a routed call's row launches (plan, k_rows, k_rows_close) run here, beside the packet kernels on the caller's stream: forked and joined with the two events
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    unsigned char *rows_buf = nullptr;
    size_t rows_cap_slots = 0, rows_cap_n = 0;
    bool rows_dirty = true;            // the scratch is not known to be zero (fresh, or a launch failed between k_rows and k_rows_close)
    u64 rows_min = (u64)8 << 10;       // packets of at least this many bytes go by rows, and from a quarter of it while they are at most 16384 (option "rows_min"; 0 = never).  With offset arrays the device applies the same marks per message (k_len_scan)
    u32 route_mid_min = 65536;         // option "route_mid_min": a routed call takes the high mark (rows_min) when at least this many messages lie between a quarter of it and it, else the low one (k_len_scan)
    u64 route_blocks_min = 1u << 17;   // option "route_blocks_min": ... and sends nothing to the packet kernels when the short messages hold fewer 16-byte blocks than this + 3.5 per message in all (0: always split)
    u32 route_top_min = 458752;        // option "route_top_min": ... and the mark rises to the last size class (16 320 bytes) when at least this many messages lie between "rows_min" and it (0: never)
    u32 rows_block = 0;                // option "rows_block": units per dealt block of k_rows (0 = the library's cut: one block per wave, or blocks of ROWS_DYN_BLOCK for large calls)
    // The state of a message under way (round 6: ONE struct for the beat-by-beat interface, the pipelined host-buffer path and what aesgcm_stream_export carries between
    // contexts, devices and processes): the RTL's Y register (src/gcm_ghash.vhd:174-186) and its counter (src/aes_icb.vhd:97-100) are d_tag[1] on the device -- in the
    // library's form, the polynomial sum X_i H^(n-1-i) of the blocks absorbed so far (the RTL's Y is this value times H) -- and `len` here (the next block's counter is
    // 2 + len / 16).
    struct StreamState {
        bool active = false, data = false, ragged = false;     // a session is open; data has begun (no more AAD); the last chunk was ragged (nothing but the tag may follow)
        int dec = 0;
        uint8_t iv[12] = {0};
        u64 aad_len = 0, len = 0, blocks = 0;                  // bytes of AAD and of data absorbed so far; GHASH blocks absorbed so far
    } s;
    // timing
    bool timing = false;
    bool timing_mute = false;          // head / tail launches beside k_body are not the measured kernel
    std::vector<std::pair<hipEvent_t, hipEvent_t>> ev;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> ev_pool;
};

extern std::vector<aesgcm_ctx *> &g_ctxs;

// the generation number of the context's host slot: written by the thread that launches, read by other contexts' launches (others_in_flight) -- atomics both ways;
// a launch that fails after taking a number gives it back, so that no context is ever taken for "under way" on account of a launch that never ran
static inline u64 gen_take(aesgcm_ctx *c) { return __atomic_add_fetch(&c->tag_gen, 1, __ATOMIC_RELAXED); }
static inline void gen_give_back(aesgcm_ctx *c) { __atomic_sub_fetch(&c->tag_gen, 1, __ATOMIC_RELAXED); }
static inline u64 gen_now(const aesgcm_ctx *c) { return __atomic_load_n(&c->tag_gen, __ATOMIC_RELAXED); }
static inline hipStream_t pick_stream(aesgcm_ctx *c, void *s) { return s ? (hipStream_t)s : c->stream; }

static const u64 MAX_DATA = (((u64)1) << 36) - 32;      // aes_icb.vhd:114
static const u64 MAX_SEQ_BLOCKS = ((u64)1) << 36;

// What the fold stage needs to know about the partials a launch produced.
struct Partials { const uint4 *ptr = nullptr; u32 np = 0; u32 kind = PARTS_NONE; const uint4 *ej0 = nullptr; u64 eA = 0; bool done = false; const uint4 *tail_item = nullptr; u32 tail_blocks = 0; };   // eA: blocks between chunk items when k_combine folds them itself (np > 1)
struct RowsScratch;
int set_lds_attrs(int device, DeviceState *ds);
int device_state(int device, DeviceState **out);
int grow_parts(aesgcm_ctx *c, size_t need);
int enqueue_fold(aesgcm_ctx *c, const uint4 *items, u32 n, u32 period, u64 eA, u64 eB, hipStream_t st, Partials *po, const FoldClose *close = nullptr);
bool ctx_body_split(const aesgcm_ctx *c, u64 len, u64 first_block, BodySplit *b);
int enqueue_main(aesgcm_ctx *c, int mode, const uint8_t iv[12], const void *d_aad, u64 aad_len, const void *d_in, u64 len, void *d_out, u64 first_block, hipStream_t st, Partials *po, bool want_tail = false);
int launch_body(aesgcm_ctx *c, int mode, BodyParams &p, u32 wgs, hipStream_t st);
int enqueue_body(aesgcm_ctx *c, int mode, const uint8_t iv[12], const BodySplit &b, const void *d_in, void *d_out, u64 first_block, hipStream_t st, Partials *po, const FoldClose *close = nullptr);
bool cyc_capable(const aesgcm_ctx *c);
bool others_in_flight(const aesgcm_ctx *c);
int enqueue_cyc(aesgcm_ctx *c, int mode, const uint8_t iv[12], const void *d_aad, u64 aad_len, const void *d_in, u64 len, void *d_out, u64 first_block, hipStream_t st, Partials *po, bool *took, bool whole_message_tag = false);
int absorb_range(aesgcm_ctx *c, int mode, const uint8_t iv[12], const void *d_aad, u64 aad_len, const void *d_in, u64 len, void *d_out, u64 first_block, hipStream_t st, uint4 *state, const uint4 **ej0 = nullptr);
int enqueue_combine(aesgcm_ctx *c, const CombineParams &p0, hipStream_t st);
int check_lengths(u64 aad_len, u64 len);
int crypt_dev(aesgcm_ctx *c, int dec, const uint8_t iv[12], const void *d_aad, u64 aad_len, const void *d_in, u64 len, void *d_out, hipStream_t st);
int fetch_tag(aesgcm_ctx *c, hipStream_t st, uint8_t tag[16]);
int ct_compare16(const uint8_t *a, const uint8_t *b);
int grow(unsigned char **p, size_t *cap, size_t need);
int ctx_load_key(aesgcm_ctx *c, const uint8_t *key, size_t key_len, int pre_nr);
int ctx_create_common(aesgcm_ctx **out, int device, const uint8_t *key, size_t key_len, int pre_nr);
int stage_in(aesgcm_ctx *c, const uint8_t *aad, size_t aad_len, const uint8_t *in, size_t len);
int stream_absorb(aesgcm_ctx *c, const void *d_aad, u64 aad_len, const void *d_in, u64 len, void *d_out, u64 first_block, hipStream_t st = nullptr, bool large = false);
int stream_open(aesgcm_ctx *c, const uint8_t iv[12], int decrypt, hipStream_t st);
int packets_pick_lg(u32 n_cu, size_t n_pkts, size_t pkt_len);
int batch_pick_lg(int n_cu, size_t n_pkts, size_t pkt_len, bool var);
int order_launch(OrderSlot &o, const u64 *d_off, size_t n_pkts, hipStream_t st, const u32 **perm);
size_t rows_carve(unsigned char *base, size_t slots, size_t n, RowsScratch *r);
int rows_scratch(aesgcm_ctx *c, size_t slots, size_t n, hipStream_t st, RowsScratch *r);
int packets_rows(aesgcm_ctx *c, int decrypt, RowsParams &p, hipStream_t st, PktParams *k = nullptr);
bool packets_by_rows(const aesgcm_ctx *c, size_t n_pkts, size_t pkt_len);
int wipe_failed(int device, size_t n_pkts, void *d_out, size_t pkt_len, const u64 *d_data_off, const int *d_auth, hipStream_t st, const u64 *d_out_ptr = nullptr, const u32 *d_len = nullptr);
int batch_launch(int device, int decrypt, size_t n_pkts, size_t key_len, BatchParams &p, void *stream);
void pipeline_release(aesgcm_ctx *c);
int pipeline_prepare(aesgcm_ctx *c, size_t chunk);
int crypt_pipelined(aesgcm_ctx *c, int dec, const uint8_t iv[12], const uint8_t *aad, size_t aad_len, const uint8_t *in, size_t len, uint8_t *out, uint8_t tag[16], size_t chunk);

#ifdef AESGCM_DEBUG_KNOBS
struct ForceShape { int pkt_lanes, pkt_deal, batch_lanes, batch_deal, batch_order, pkt_ilp, pkt_rows; };
extern ForceShape g_force;
#endif
