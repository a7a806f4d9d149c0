// aesgcm_rows.h -- MANY messages under one key through k_body's row code (round 5): the lane pieces and the plan.
//
// The reference's deployment is frame after frame under one key (tb/gcm_test.py:76-85; H is kept while no new key is loaded,
// src/gcm_gctr.vhd:142-144).  Until round 4 a message-sized packet of aesgcm_packets_crypt_dev went through the wave-per-packet group
// kernel (two-table round, LDS full of tree tables: 4096 x 1 MiB 731 GiB/s against 980 for one message of the same bytes).  Here the rows
// of ALL messages of a call are one pool of work for k_body's row loop (four T-tables, rounds 1 - 2 from per-strand lane constants and the
// scalar cache, a five-bit GHASH table in LDS):
//
//   * a message of `len` bytes is R = len / 1024 whole rows of 64 blocks, aligned to its first block (every packet starts at counter 2,
//     src/aes_icb.vhd:97-118, so every message IS an aligned body), plus a TAIL of tb <= 64 blocks (the last one ragged), plus its AAD;
//   * the ROWS are laid on an axis of UNITS: a unit per row of every message, in their order (and one for an AAD of more than 64 blocks, as rows of its
//     own).  A RUN of consecutive rows is body_rows_lane's: k_body's row code with the lane constants of all four row phases in registers and one Horner
//     accumulator of stride H^64, so a run of any length leaves ONE value behind.  A tail of more than 16 blocks is one more unit behind the message's rows: a
//     right-aligned row through the general cipher code, one pass of a wave;
//   * the axis is cut into BLOCKS of D units.  A small or mid-size call is cut into exactly one block per wave of the launch (equal shares, no dispenser:
//     the launch is as long as its rows and nothing waits for a last chunk); a large one into blocks of 64 units dealt from dispensers, as k_body deals
//     its chunks.  Where a block's range meets the boundaries of a message it falls into PIECES: a run of rows, a long AAD;
//   * at the end of a run the wave does not leave its 64 lane accumulators: lane L's value times H^(63 - L) through the key's per-lane Shoup tables
//     (KeyMaterial::ltab, read from DEVICE memory: the vector-memory path is idle in this kernel and its LDS is not -- the same tables staged in the 19 KiB of
//     LDS behind the T-tables made every size slower on one box, tail-only calls 0.95 -> 1.25 ms, 64 KiB messages 4.55 -> 4.61 ms,
//     profiles/r05/rows_ab2/lds_lane_tables_ab.txt), XORed over the wave, is the polynomial of the run up to its last block -- a 32-byte RECORD per piece:
//     those 16 bytes, the message, and the exponent still due: H^(blocks behind the piece + 2), e = 64 (R - 1 - last row) + tb + 2 (the long AAD:
//     64 R + tb + 2).  HBM traffic is 1.00 x algorithmic whatever the cut;
//   * everything else -- the AAD blocks (up to 64) and the blocks of SHORT tails (up to 16) of every message, and what a message owes once: the length block and
//     E_K(J0) -- is the SMALLS: message after message, block after block, one more axis, walked by the LANES of the closing launch (k_rows_close), a lane per
//     block, whatever message it belongs to.  The lane finds its message (arithmetic, or a search in the plan's prefix sums), runs the cipher under ITS
//     message's IV through a 4 KiB copy of the four T-tables in the workgroup's LDS, and weighs its block with the power of H its place in its message asks
//     for (rows_small_block).  (Until the middle of round 5 every message had a unit of the row launch for its tail -- which also held the length block and
//     encrypted counter 1 -- and one for its AAD: a wave's pass each, for a block or two.  64 KiB messages with a 13-byte header and a ragged end ran 7 %
//     below whole ones, 16 KiB ones 16 %, profiles/r05/rows_aad_cost.txt.  Packing the smalls into units of the row launch itself was built too: its code
//     around the row loop cost AES-256 nine scratch accesses per row.)
//   * the same small launch closes every tag: a LANE per record slot multiplies by H^e bit-serially (any exponent, all lanes in parallel); records, the
//     segments of the smalls axis (one message's AAD, or its tail: the lanes of a segment XOR their terms together first -- one address serves 87 M atomics a
//     second) and the message's own lane each XOR their product into the message's accumulator with memory-side atomics and count themselves arrived.  The
//     lane that counts a message's last arrival holds its tag  P H^2 ^ L H ^ E_K(J0)  (gcm_ghash.vhd:257,293 re-associated), stores it and, for decrypt,
//     compares.
//
// Fixed-size records need no plan: message m owns the units [m U, (m + 1) U) and the smalls blocks [m S, (m + 1) S).  With offset arrays the lengths are
// on the device, and one small launch in front (k_rows_plan) makes the three prefix sums there: units, smalls blocks and record slots in front of every
// message.  Everything the launches share between calls -- accumulators, arrival counts, record flags, dispensers -- is zero at rest: whoever consumes a
// thing puts the zero back.
#pragma once
#include "aesgcm_dev.h"

#define ROWS_NQ AESGCM_NQ                     /* dispenser queues of k_rows (one cache line each) */
#define ROWS_DYN_BLOCK 64u                    /* units per dealt block (measured, 4096 x 1 MiB: 16 units 885 GiB/s, 32: 925, 64: 932, 128: 922, 256: 876; profiles/r05/rows_v1_chunk_sweep.txt) */
#define ROWS_STATIC_MAX 192u                  /* units per wave up to which a call is cut into one block per wave */
#define ROWS_CLOSE_WG 256u                    /* lanes per k_rows_close workgroup */
#define ROWS_NB_CAP 65536u                    /* offset-array form: blocks the scratch is sized for (k_rows_plan falls back to one block per wave beyond) */
#define ROWS_SMALL_AAD 64u                    /* an AAD of up to this many blocks lies on the smalls axis; a longer one is a unit of the row launch (rows of its own, Horner with H^64) */
#define ROWS_SMALL_TAIL 16u                   /* a tail of up to this many blocks lies on the smalls axis; a longer one is a unit of the row launch: one pass of a wave (a block of the smalls costs the
                                                 closing launch about a sixteenth of such a pass; 262 144 x 9000 bytes -- tails of 51 blocks -- 325 GiB/s with every tail in the closing launch,
                                                 profiles/r05/rows_ragged_few.txt) */
#define ROWS_FEW_TAIL 4u                      /* the routing rule's word for "ends (almost) on a row" (packets_by_rows) */
#define ROWS_SLOTS_PER_MSG 3u                 /* offset-array form: record slots per message besides one per block boundary -- a run, a long tail, a long AAD */
#define ROWS_PLAN_ONE_WG 4096u                /* offset-array form: up to this many messages one workgroup makes the plan (16 us); beyond, five small launches with a thread per message */
#define ROWS_REC_VALID 1u
#define ROWS_REC_WEIGH 2u

struct RowsRec { G128 w; u64 e; u32 msg; u32 flags; };     // one piece: 32 bytes
// What the device decides about a call whose lengths are on the device (64 bytes of the context's scratch, rewritten by every such call):
//   * by k_len_scan (round 6, the ROUTE of the call): route_min -- messages of at least this many bytes (data + AAD) go by rows, the others to the packet kernels --,
//     n_small (how many take the packet kernels: the first n_small entries of the launch order `perm`), pkt_lg / pkt_deal (the packet kernel shape for that count and
//     its packets per dispenser fetch), pkt_counter (that launch's dispenser, zero again); and the VERDICT on the call's lengths (k_len_hist checks every one it reads):
//     bad != 0 -- NOTHING of the call runs (every kernel behind returns at once, outputs and tags are untouched) -- with the reason in status / detail, also stored in the
//     context's pinned host slot (aesgcm_ctx_status);
//   * by k_rows_plan*: the cut (G, D, NB, dyn) -- and the verdict itself for a call that is not routed (fixed-size data with an AAD offset array), or when the cut does not
//     fit (PLAN, UNITS: not reachable with the scratch the host sizes; the row launches do not run then);
//   * for the packet kernels of a call whose messages live WHEREVER (aesgcm_messages_crypt_dev): the five arrays of addresses and lengths (sc_*; zero = offsets from the
//     call's buffers).  They ride here, behind the one pointer the packet kernels get, rather than in the kernels' own arguments: ten more scalar registers held across
//     the packet loops cost k_pktg its last free vector register (16 - 24 bytes of scratch in the 4-lane shape).
struct RowsHdr { u64 G; u32 D, NB, dyn, bad; u32 route_min, n_small, pkt_lg, pkt_deal, pkt_counter, status; u64 detail; u32 pad[2];
                 u64 sc_in, sc_out, sc_aad, sc_len, sc_alen; u64 pad2[3]; };
static_assert(sizeof(RowsHdr) == 128, "RowsHdr: two 64-byte lines");
HD void pkt_place_scattered(const RowsHdr *h, u32 pkt, u64 *doff, u64 *ooff, u64 *aoff, u32 *pkt_len, u32 *aad_len) {
    const u64 *in_ptr = reinterpret_cast<const u64 *>((uintptr_t)h->sc_in), *out_ptr = reinterpret_cast<const u64 *>((uintptr_t)h->sc_out), *aad_ptr = reinterpret_cast<const u64 *>((uintptr_t)h->sc_aad);
    const u32 *len_arr = reinterpret_cast<const u32 *>((uintptr_t)h->sc_len), *alen_arr = reinterpret_cast<const u32 *>((uintptr_t)h->sc_alen);
    *doff = in_ptr[pkt]; *ooff = out_ptr[pkt]; *pkt_len = len_arr[pkt];
    *aoff = aad_ptr ? aad_ptr[pkt] : 0; *aad_len = aad_ptr ? alen_arr[pkt] : 0u;
}
#define ROWS_LEN_LIMIT (1ull << 28)           /* a message's data and its AAD: each below this (include/aesgcm.h) */
#define ROWS_ST_OK 0u
#define ROWS_ST_PLAN_FIT 1u                   /* AESGCM_STATUS_PLAN: the plan does not fit the scratch the host sized for it */
#define ROWS_ST_LENGTH 2u                     /* AESGCM_STATUS_LENGTH: a length of 2^28 bytes or more, or offsets that do not rise (detail: the first such message) */
#define ROWS_ST_UNITS 3u                      /* AESGCM_STATUS_UNITS: the cut does not fit 32-bit block numbers */
#define ROWS_ROUTE_NEVER 0xFFFFFFFFu          /* route_min: nothing goes by rows */
// where the sizes of a launch order (k_len_*) come from: offset arrays (aoff NULL: fixed aad_len) or per-message length arrays
struct LenSrc { const u64 *off, *aoff; const u32 *len_arr, *alen_arr; u32 aad_len; };
// how k_len_scan routes a call: hdr NULL = no route (a plain launch order); marks as length classes (64 bytes each; >= PKT_LEN_CLASSES = never by rows): c_hi when at
// least mid_min messages lie between the marks, else c_lo; below blocks_min blocks of short messages in all, everything goes by rows (the rule and its measurements:
// k_len_scan); force_lg != 0xFF / force_deal != 0: the debug library's forced packet kernel shape
struct RouteCfg { RowsHdr *hdr; u32 n, n_cu, c_hi, c_lo, mid_min, force_lg, force_deal; u64 blocks_min; u64 sc_in, sc_out, sc_aad, sc_len, sc_alen; u32 top_min, pad; };

struct RowsParams {
    const unsigned char *ivs;                 // n_pkts * 12 bytes
    const unsigned char *aad;
    const unsigned char *in;
    unsigned char *out;
    unsigned char *tags;                      // n_pkts * 16
    const unsigned char *expect;              // dec: expected tags or NULL
    int *auth;                                // dec: per message 1 / 0, or NULL
    const u64 *data_off, *aad_off;            // n_pkts + 1 byte offsets, or NULL = fixed pkt_len / aad_len records
    // ... or messages WHEREVER THEY LIVE (aesgcm_messages_crypt_dev): device addresses and lengths per message; in / out / aad above are NULL then
    const u64 *in_ptr, *out_ptr, *aad_ptr;    // n_pkts device addresses each (aad_ptr: or NULL = no AAD)
    const u32 *len_arr, *alen_arr;            // n_pkts lengths each (alen_arr: NULL with aad_ptr)
    u32 n_pkts, pkt_len, aad_len;
    u32 waves;                                // waves of the k_rows launch
    // the cut: fixed-size form (the host knows it) ...
    u64 G;                                    // units of the call
    u32 D, NB, dyn;                           // units per block, blocks, 1 = blocks come from the dispensers / 0 = wave w takes block w
    u32 U, S, SM;                             // per message: units, blocks of the smalls axis, record slots
    // ... or with offset arrays (k_rows_plan made it)
    const RowsHdr *hdr;
    const u64 *prefix;                        // units in front of message m (n_pkts + 1 entries)
    const u64 *sprefix;                       // smalls blocks in front of message m (n_pkts + 1 entries)
    const u32 *slot_base;                     // record slots in front of message m (n_pkts + 1 entries)
    u32 slot_cap;                             // record slots the scratch holds (k_rows_close has a lane for each)
    RowsRec *rec;
    unsigned long long *acc;                  // per message {hi, lo}: the XOR of everything that makes its tag
    u32 *cnt;                                 // per message: pieces arrived (k_rows_close; the number due is rows_pieces)
    u32 *queues;                              // ROWS_NQ dispensers, 16 u32 apart
    u32 prio_rows;                            // rotate the waves' issue priorities every so many rows (one block per wave: equal shares must also run at equal speed)
};

// ---- geometry (host, planner and kernels agree through these) -------------------------------------
struct RowsGeom { u32 R, Q, rho, tb; };
HD RowsGeom rows_geom(u64 len) {
    RowsGeom g;
    g.R = (u32)(len >> 10); g.Q = g.R >> 2; g.rho = g.R & 3u;
    g.tb = ((u32)(len & 1023u) + 15u) >> 4;
    return g;
}
// the size a routed call goes by: data + AAD (a lane of the packet kernels walks both, block by block); saturating, so that lengths beyond the limit -- which the plan refuses -- route somewhere defined
HD u32 rows_route_size(u64 len, u64 alen) { const u64 t = len + alen; return t < len || t > 0xFFFFFFFEull ? 0xFFFFFFFEu : (u32)t; }
HD u64 len_src_data(const LenSrc &s, u32 i) { return s.len_arr ? (u64)s.len_arr[i] : s.off[i + 1] - s.off[i]; }
HD u64 len_src_aad(const LenSrc &s, u32 i) { return s.len_arr ? (s.alen_arr ? (u64)s.alen_arr[i] : 0ull) : s.aoff ? s.aoff[i + 1] - s.aoff[i] : (u64)s.aad_len; }
HD u32 len_src_size(const LenSrc &s, u32 i) { return rows_route_size(len_src_data(s, i), len_src_aad(s, i)); }
// what the sort needs beside the lengths to write the launch's packet records (aesgcm_pkt.h PktDesc; desc NULL: none wanted): the IVs, and for messages wherever they live
// the arrays of addresses
struct DescSrc { PktDesc *desc; const unsigned char *ivs; const u64 *in_ptr, *out_ptr, *aad_ptr; };
HD PktDesc len_src_desc(const LenSrc &s, const DescSrc &ds, u32 i, u64 dl, u64 al) {
    PktDesc d;
    if (s.len_arr) { d.a = ds.in_ptr[i]; d.b = ds.aad_ptr ? ds.aad_ptr[i] : 0ull; d.c = ds.out_ptr[i]; }
    else { d.a = s.off[i]; d.b = s.aoff ? s.aoff[i] : (u64)i * s.aad_len; d.c = 0; }
    d.len = (u32)dl; d.alen = (u32)al; d.pkt = i;
    const unsigned char *iv = ds.ivs + (size_t)i * 12;
    d.iv0 = gload4_any(iv); d.iv1 = gload4_any(iv + 4); d.iv2 = gload4_any(iv + 8);
    return d;
}
// A message of a ROUTED call below the mark is the packet kernels': to the row launches it is a message of NO bytes and NO AAD that also owes nothing -- no unit, no
// smalls block, no record slot, no arrival (its tag comes from the packet kernel).  Everything that counts a message's share goes through these two (the geometry of
// an empty message, an AAD of no blocks), so the counts below need no word about routing
HD bool rows_is_small(u64 len, u64 alen, u32 route_min) { return rows_route_size(len, alen) < route_min; }
HD RowsGeom rows_geom_routed(u64 len, u64 alen, u32 route_min) { return rows_geom(rows_is_small(len, alen, route_min) ? 0ull : len); }
HD u32 rows_na_routed(u64 len, u64 alen, u32 route_min) { return rows_is_small(len, alen, route_min) ? 0u : (u32)((alen + 15u) >> 4); }
HD u32 rows_na(u32 alen) { return (alen + 15u) >> 4; }                                   // AAD blocks
HD u32 rows_long_aad(u32 na) { return na > ROWS_SMALL_AAD ? 1u : 0u; }
HD u32 rows_small_aad(u32 na) { return na > ROWS_SMALL_AAD ? 0u : na; }
HD u32 rows_long_tail(const RowsGeom &g) { return g.tb > ROWS_SMALL_TAIL ? 1u : 0u; }
HD u32 rows_small_tail(const RowsGeom &g) { return g.tb > ROWS_SMALL_TAIL ? 0u : g.tb; }
HD u32 rows_units(const RowsGeom &g, u32 na) { return g.R + rows_long_tail(g) + rows_long_aad(na); }      // units of the row launch: the rows, the long tail, the long AAD (0 for a short message: only the closing sees it)
HD u32 rows_smalls(const RowsGeom &g, u32 na) { return rows_small_aad(na) + rows_small_tail(g); }         // blocks on the smalls axis: the (short) AAD, then the (short) tail
// the natural segment of unit u of a message: its rows (when it has any), then the long tail, then the long AAD
HD u32 rows_nat(const RowsGeom &g, u32 u) { return u < g.R ? 0u : (g.R ? 1u : 0u) + (u - g.R); }
HD u32 rows_nat_count(const RowsGeom &g, u32 na) { return (g.R ? 1u : 0u) + rows_long_tail(g) + rows_long_aad(na); }
// record slots of a message whose units are [g0, g0 + U): a slot per (natural segment, block) pair it can have -- slot = base + nat + (block - first block)
HD u32 rows_slots(const RowsGeom &g, u32 na, u64 g0, u32 D) {
    const u32 U = rows_units(g, na);
    return U ? rows_nat_count(g, na) + (u32)((g0 + U - 1u) / D - g0 / D) : 0u;
}
// arrivals k_rows_close counts for a message: the pieces it falls into under the cut -- its run of rows one per block it touches, the long tail, the long AAD --,
// its blocks of the smalls axis, and the message's own lane
HD u32 rows_pieces(const RowsGeom &g, u32 na, u64 g0, u32 D) {
    return (g.R ? (u32)((g0 + g.R - 1u) / D - g0 / D) + 1u : 0u) + rows_long_tail(g) + rows_long_aad(na) + rows_smalls(g, na) + 1u;      // (never asked for a message of the packet kernels: nothing of it arrives here)
}
// the cut of a call of G units for `waves` waves: one block per wave while that is at most ROWS_STATIC_MAX units (or when the dealt cut would not fit the
// scratch: nb_cap blocks), else blocks of ROWS_DYN_BLOCK units from the dispensers.  force_d > 0: dealt blocks of that many units (tests)
HD bool rows_cut(u64 G, u32 waves, u32 force_d, u64 nb_cap, u32 *D, u32 *NB, u32 *dyn) {
    if (waves < 1) waves = 1;
    u64 d = (G + waves - 1u) / waves;
    u32 dy = 0;
    if (d < 1) d = 1;
    const u64 dd = force_d ? force_d : ROWS_DYN_BLOCK;
    if ((force_d || d > ROWS_STATIC_MAX) && (G + dd - 1u) / dd <= nb_cap) { d = dd; dy = 1; }
    *D = (u32)d; *NB = (u32)((G + d - 1u) / d); *dyn = dy;
    return d <= 0xFFFFFFFFull && (G + d - 1u) / d <= 0xFFFFFFFFull;                    // false: the cut does not fit its 32-bit numbers (ROWS_ST_UNITS; more rows than any memory holds)
}
// the message's lengths and where it lies: offsets from the call's in / out / aad (the scattered form: from 0, i.e. addresses)
struct RowsMsg { u64 doff, ooff, aoff; u32 len, alen; };
HD RowsMsg rows_msg(const RowsParams &p, u32 m) {
    RowsMsg q;
    if (p.len_arr) {
        q.doff = p.in_ptr[m]; q.ooff = p.out_ptr[m]; q.len = p.len_arr[m];
        q.aoff = p.aad_ptr ? p.aad_ptr[m] : 0; q.alen = p.aad_ptr ? p.alen_arr[m] : 0u;
        return q;
    }
    q.len = p.pkt_len; q.alen = p.aad_len;
    q.doff = (u64)m * p.pkt_len; q.aoff = (u64)m * p.aad_len;
    if (p.data_off) { q.doff = p.data_off[m]; q.len = (u32)(p.data_off[m + 1] - q.doff); }
    if (p.aad_off) { q.aoff = p.aad_off[m]; q.alen = (u32)(p.aad_off[m + 1] - q.aoff); }
    q.ooff = q.doff;
    return q;
}
// (the (u32) casts above are safe: k_rows_plan* refuses a call -- hdr->bad, nothing runs -- in which any length or offset difference is 2^28 or more)
// a ROUTED call (hdr->route_min, made by k_len_scan; the plan of a call that is not routed writes 0 there): messages below the mark are the packet kernels', the row launches see them as nothing
HD u32 rows_route_min(const RowsParams &p) { return p.hdr ? p.hdr->route_min : 0u; }
HD RowsGeom rows_geom_of(const RowsMsg &q, u32 route_min) { return rows_geom_routed(q.len, q.alen, route_min); }
HD u32 rows_na_of(const RowsMsg &q, u32 route_min) { return rows_na_routed(q.len, q.alen, route_min); }
HD const unsigned char *rows_src(const RowsParams &p, const RowsMsg &q) { return reinterpret_cast<const unsigned char *>((uintptr_t)p.in + q.doff); }
HD unsigned char *rows_dst(const RowsParams &p, const RowsMsg &q) { return reinterpret_cast<unsigned char *>((uintptr_t)p.out + q.ooff); }
HD const unsigned char *rows_aadp(const RowsParams &p, const RowsMsg &q) { return reinterpret_cast<const unsigned char *>((uintptr_t)p.aad + q.aoff); }
// the LAST message of [lo, hi) whose entry of a prefix array is <= x: the one that owns position x (messages without a share have their successor's start)
HD u32 rows_search(const u64 *pre, u64 x, u32 lo, u32 hi) {
    while (hi - lo > 1u) { const u32 mid = lo + ((hi - lo) >> 1); if (pre[mid] <= x) lo = mid; else hi = mid; }
    return lo;
}
// the message that owns unit g, and the units / smalls blocks / record slots in front of a message
HD u32 rows_find_msg(const RowsParams &p, u64 g) { return p.prefix ? rows_search(p.prefix, g, 0u, p.n_pkts) : (u32)(g / p.U); }
HD u64 rows_unit_base(const RowsParams &p, u32 m) { return p.prefix ? p.prefix[m] : (u64)m * p.U; }
HD u64 rows_small_base(const RowsParams &p, u32 m) { return p.sprefix ? p.sprefix[m] : (u64)m * p.S; }
HD u64 rows_small_total(const RowsParams &p) { return p.sprefix ? p.sprefix[p.n_pkts] : (u64)p.n_pkts * p.S; }
HD u32 rows_slot_base(const RowsParams &p, u32 m) { return p.slot_base ? p.slot_base[m] : m * p.SM; }

// ---- k_rows: one piece ---------------------------------------------------------------------------
enum { ROWS_RUN = 0, ROWS_TAIL = 1, ROWS_AAD = 2 };
struct RowsPiece { u32 kind, r0, len, slot; u64 e; };                   // run: rows [r0, r0 + len); len = units taken
// the piece that starts at unit u of the message (geometry g, first unit g0) and may take up to `room` units; D = units per block
HD RowsPiece rows_piece(const RowsGeom &g, u32 slot_base, u64 g0, u32 u, u64 room, u32 D) {
    RowsPiece pc;
    pc.slot = slot_base + rows_nat(g, u) + (u32)((g0 + u) / D - g0 / D);
    pc.len = 1; pc.r0 = 0;
    if (u < g.R) {
        pc.kind = ROWS_RUN;
        pc.r0 = u;
        const u32 left = g.R - u;
        pc.len = room < left ? (u32)room : left;
        pc.e = 64ull * (g.R - (pc.r0 + pc.len)) + g.tb + 2u;              // blocks behind the run's last row, and H^2
    } else if (rows_long_tail(g) && u == g.R) {
        pc.kind = ROWS_TAIL;                                              // the long tail, whole: its lanes weigh their blocks in full
        pc.e = 0;
    } else {
        pc.kind = ROWS_AAD;                                               // the long AAD, whole
        pc.e = 64ull * g.R + g.tb + 2u;
    }
    return pc;
}
// lane `lane` of the wave that runs a piece of rows: CTR over the rows r0 .. r0 + len - 1 and the lane's Horner accumulator (stride H^64)
template <int NR, int MODE>
HD uint4 rows_run_lane(const KeyMaterial *__restrict__ km, const DevTables *__restrict__ tb, const RowsParams &p, const RowsMsg &q, const RowsPiece &pc,
                       const unsigned char *smem, const CtrConsts &cc, u32 lane, u32 prio_rows, u32 prio_slot) {
    return body_rows_lane<NR, MODE>(km, tb, rows_src(p, q), rows_dst(p, q), 0u, smem, cc, pc.r0, pc.len, lane, prio_rows, prio_slot);
}
// the lane's term of the run's polynomial: B_L H^(63 - L) (XOR over the wave = the polynomial of the run up to its last block)
HD G128 rows_run_term(const KeyMaterial *__restrict__ km, uint4 acc, u32 lane) { return shoup2_gmul_lds(mo_to_be(acc), km->ltab[63u - lane]); }
// A tail of more than ROWS_SMALL_TAIL blocks (data blocks 64 R ..., tb of them, tb <= 64, the last one ragged) as ONE right-aligned row: lane L >= 64 - tb runs the cipher on counter
// 2 + block index (aes_icb.vhd:97-118).  Returns the lane's term  X_L H^(65 - L)  of  (tail polynomial) H^2.
template <int NR, int DEC>
HD G128 rows_tail_lane(const KeyMaterial *__restrict__ km, const RowsParams &p, const RowsMsg &q, const unsigned char *smem, const CtrConsts &cc, u32 lane) {
    const u32 *__restrict__ rk = km->rk;
    const u32 lb = (lane & 31u) << 2;
    const RowsGeom g = rows_geom(q.len);
    const u32 pad = 64u - g.tb;
    const unsigned char *src = rows_src(p, q);
    unsigned char *dst = rows_dst(p, q);
    G128 z = {{0, 0, 0, 0}};
    if (lane >= pad) {                                                                  // (the table multiply too: its 32 reads per lane go to 64 different tables, and the memory path takes them a lane at a time)
        uint4 gin;
        const u32 i = 64u * g.R + (lane - pad);                                         // block index in the message
        u32 s0, s1, s2, s3;
        ctr_rounds_lds<NR>(bswap32(2u + i), cc, s0, s1, s2, s3, rk, smem, lb);
        const u32 off = 16u * i, rem = q.len - off;
        const bool full = rem >= 16u;
        const uint4 x = full ? gload16_any(src + off) : load_block_bytes(src + off, rem);
        uint4 y = make_uint4(x.x ^ s0, x.y ^ s1, x.z ^ s2, x.w ^ s3);                    // gcm_gctr.vhd:150
        if (!full) y = mask_block(y, rem);
        if (full) gstore16_any(dst + off, y); else store_block_bytes(dst + off, y, rem);
        gin = DEC ? x : y;                                                              // aes_gcm.vhd:207-211
        z = shoup2_gmul_lds(mo_to_be(gin), km->ltab[65u - lane]);
    }
    return z;
}
// An AAD of more than ROWS_SMALL_AAD blocks as rows of its own (right-aligned, Horner with H^64): the lane's term  A_L H^(63 - L)  of the AAD's polynomial
HD G128 rows_aad_lane(const KeyMaterial *__restrict__ km, const RowsParams &p, const RowsMsg &q, const unsigned char *smem, u32 lane) {
    const u32 n_aad = rows_na(q.alen), rows = (n_aad + 63u) >> 6, pad = 64u * rows - n_aad;
    const unsigned char *a = rows_aadp(p, q);
    uint4 acc = make_uint4(0, 0, 0, 0);
    for (u32 k = 0; k < rows; k++) {
        if (k) acc = ghash_mul_const_lds(acc, smem);
        const u32 slot = k * 64u + lane;
        if (slot < pad) continue;
        const u32 off = 16u * (slot - pad), rem = q.alen - off;
        acc = xor4(acc, rem >= 16u ? gload16_any(a + off) : load_block_bytes(a + off, rem));
    }
    return shoup2_gmul_lds(mo_to_be(acc), km->ltab[63u - lane]);
}

// ---- k_rows_close: what a message owes once: (length block) H ^ E_K(J0) -------------------------------
#define ROWS_CLOSE_LDS_TE 0u                  /* te0 | te1 | te2 | te3 of DevTables, 4 KiB, one copy: the lanes of a wave read where their bytes say (a few-way bank conflict, in a launch that is short) */
HD void rows_close_fill_te(unsigned char *smem, const DevTables *__restrict__ tb, u32 tid) {          // thread tid of ROWS_CLOSE_WG = 256: one uint4 each
    *reinterpret_cast<uint4 *>(smem + ROWS_CLOSE_LDS_TE + 16u * tid) = reinterpret_cast<const uint4 *>(tb->te0)[tid];     // te0 .. te3 are consecutive
}
// one block through the cipher with the four T-tables at `te` (aes_round.vhd's SB -> SR -> MC as table lookups, config/config_aes_round.py:120-124; the last round,
// aes_last_round.vhd:76, takes S[x] from byte 1 of te0[x]); memory-order words in and out
HD uint4 aes_encrypt_te(const u32 *te, const u32 *__restrict__ rk, u32 nr, uint4 in) {
    const u32 *te0 = te, *te1 = te + 256, *te2 = te + 512, *te3 = te + 768;
    u32 s0 = in.x ^ rk[0], s1 = in.y ^ rk[1], s2 = in.z ^ rk[2], s3 = in.w ^ rk[3];
    for (u32 r = 1; r < nr; r++) {
        const u32 *k = rk + 4u * r;
        const u32 t0 = te0[s0 & 255u] ^ te1[(s1 >> 8) & 255u] ^ te2[(s2 >> 16) & 255u] ^ te3[s3 >> 24] ^ k[0];
        const u32 t1 = te0[s1 & 255u] ^ te1[(s2 >> 8) & 255u] ^ te2[(s3 >> 16) & 255u] ^ te3[s0 >> 24] ^ k[1];
        const u32 t2 = te0[s2 & 255u] ^ te1[(s3 >> 8) & 255u] ^ te2[(s0 >> 16) & 255u] ^ te3[s1 >> 24] ^ k[2];
        const u32 t3 = te0[s3 & 255u] ^ te1[(s0 >> 8) & 255u] ^ te2[(s1 >> 16) & 255u] ^ te3[s2 >> 24] ^ k[3];
        s0 = t0; s1 = t1; s2 = t2; s3 = t3;
    }
    const u32 *k = rk + 4u * nr;
#define ROWS_SB(x) ((te0[(x) & 255u] >> 8) & 255u)
    const u32 o0 = (ROWS_SB(s0) | (ROWS_SB(s1 >> 8) << 8) | (ROWS_SB(s2 >> 16) << 16) | (ROWS_SB(s3 >> 24) << 24)) ^ k[0];
    const u32 o1 = (ROWS_SB(s1) | (ROWS_SB(s2 >> 8) << 8) | (ROWS_SB(s3 >> 16) << 16) | (ROWS_SB(s0 >> 24) << 24)) ^ k[1];
    const u32 o2 = (ROWS_SB(s2) | (ROWS_SB(s3 >> 8) << 8) | (ROWS_SB(s0 >> 16) << 16) | (ROWS_SB(s1 >> 24) << 24)) ^ k[2];
    const u32 o3 = (ROWS_SB(s3) | (ROWS_SB(s0 >> 8) << 8) | (ROWS_SB(s1 >> 16) << 16) | (ROWS_SB(s2 >> 24) << 24)) ^ k[3];
#undef ROWS_SB
    return make_uint4(o0, o1, o2, o3);
}
// the lane of message m: [8 len(A)]_64 || [8 len(C)]_64 (gcm_ghash.vhd:257) times H, and E_K(IV || 0^31 1), the J0 block the RTL latches first (gcm_ghash.vhd:158-169)
HD G128 rows_msg_term(const KeyMaterial *__restrict__ km, const u32 *te, const RowsParams &p, u32 m) {
    const RowsMsg q = rows_msg(p, m);
    G128 z = tag_len_term(km, q.alen, q.len);
    const unsigned char *ivp = p.ivs + (size_t)m * 12;
    const G128 e = mo_to_be(aes_encrypt_te(te, km->rk, km->nr, make_uint4(load_le32(ivp), load_le32(ivp + 4), load_le32(ivp + 8), 0x01000000u)));
    z.w[0] ^= e.w[0]; z.w[1] ^= e.w[1]; z.w[2] ^= e.w[2]; z.w[3] ^= e.w[3];
    return z;
}

// ---- k_rows_close: block t of the smalls axis ------------------------------------------------------------
// Block i of the short AAD, or of the short tail, of the message that owns t.  Returns the message; *z is the block times the part of its power of H that the
// key's per-exponent Shoup tables hold (KeyMaterial::ltab, e < 130), *e_run the exponent that is still due -- the same for every block of the segment, so the
// lanes of a segment are XORed together first and ONE of them pays the bit-serial power (k_rows_close):
//   AAD block i of na:   A H^(na - 1 - i),  still due H^(64 R + tb + 2);
//   tail block i of tb:  the cipher on counter 2 + 64 R + i under the message's IV (aes_icb.vhd:97-118), the data block XORed with it (gcm_gctr.vhd:150), the
//                        ciphertext X -- the input, for decrypt (aes_gcm.vhd:207-211) -- times H^(tb - i + 1): nothing due.
template <int DEC>
HD u32 rows_small_block(const KeyMaterial *__restrict__ km, const u32 *te, const RowsParams &p, u64 t, G128 *z, u64 *e_run) {
    const u32 m = p.sprefix ? rows_search(p.sprefix, t, 0u, p.n_pkts) : ((t >> 32) ? (u32)(t / p.S) : (u32)t / p.S);
    const RowsMsg q = rows_msg(p, m);
    const RowsGeom g = rows_geom(q.len);
    const u32 nas = rows_small_aad(rows_na(q.alen)), r = (u32)(t - rows_small_base(p, m));      // (a block behind the short AAD is a block of a SHORT tail: a long one is not on this axis)
    uint4 x;
    u32 e;
    if (r < nas) {
        const unsigned char *a = rows_aadp(p, q);
        const u32 off = 16u * r, rem = q.alen - off;
        x = rem >= 16u ? gload16_any(a + off) : load_block_bytes(a + off, rem);
        e = nas - 1u - r;
        *e_run = 64ull * g.R + g.tb + 2u;
    } else {
        const u32 i = r - nas, bi = 64u * g.R + i;                                       // block of the tail, block of the message
        const unsigned char *ivp = p.ivs + (size_t)m * 12;
        const uint4 ks = aes_encrypt_te(te, km->rk, km->nr, make_uint4(load_le32(ivp), load_le32(ivp + 4), load_le32(ivp + 8), bswap32(2u + bi)));
        const unsigned char *src = rows_src(p, q);
        unsigned char *dst = rows_dst(p, q);
        const u32 off = 16u * bi, rem = q.len - off;
        const bool full = rem >= 16u;
        const uint4 in = full ? gload16_any(src + off) : load_block_bytes(src + off, rem);
        uint4 y = make_uint4(in.x ^ ks.x, in.y ^ ks.y, in.z ^ ks.z, in.w ^ ks.w);
        if (!full) y = mask_block(y, rem);
        if (full) gstore16_any(dst + off, y); else store_block_bytes(dst + off, y, rem);
        x = DEC ? in : y;
        e = g.tb - i + 1u;
        *e_run = 0;
    }
    *z = shoup2_gmul_lds(mo_to_be(x), km->ltab[e]);                                     // e <= 64 < AESGCM_NLTAB
    return m;
}
// the rest of a segment's power, paid once for the XOR of its blocks
HD G128 rows_small_due(const KeyMaterial *__restrict__ km, const G128 &z, u64 e_run) { return e_run ? gf_mul(z, gf_pow_h_serial(km, e_run)) : z; }

// ---- k_rows_close: a record's contribution ----------------------------------------------------------
HD G128 rows_weigh(const KeyMaterial *__restrict__ km, const RowsRec &r) { return (r.flags & ROWS_REC_WEIGH) ? gf_mul(r.w, gf_pow_h_serial(km, r.e)) : r.w; }
