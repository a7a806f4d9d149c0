/*
 * aesgcm.h -- C ABI of libaesgcm_hip.so: MI355X (gfx950) AES-GCM bulk path.
 *
 * This is the drop-in boundary for the hot path of BLu85/AES-GCM-128-192-256-bits.  The reference
 * has no FFI of its own: its software surface is the Python class tb/gcm_model.py:5-51, which
 * forwards every call to pycryptodome (tb/gcm_model.py:18 AES.new(..MODE_GCM..), :22 update,
 * :26 encrypt, :30 decrypt, :35 digest, :44 verify).  Each entry point below names the reference
 * item whose arithmetic it replaces (paths relative to the reference checkout); the Python binding
 * a maintainer would add is shown in INTEGRATION.md.
 *
 * Conventions
 *   - extern "C", plain pointers and sizes; no C++/torch/HIP types in any signature
 *     (a HIP stream is passed as void*; NULL = the context's own stream).
 *   - every function returns AESGCM_OK (0) or a negative AESGCM_E* code; nothing throws or aborts.
 *   - ALL cryptographic arithmetic (key expansion, H, GHASH tables, CTR, GHASH, tag) runs in HIP
 *     kernels on the selected device.  There is no CPU fallback: without a usable HIP device every
 *     compute entry point fails with AESGCM_EHIP.
 *   - the caller owns every buffer it passes; the library owns only opaque contexts.
 *   - a context is not thread-safe; distinct contexts may be used from distinct threads.  Calls on ONE
 *     context must be stream-ordered (same stream, or the caller synchronises between streams): a context
 *     owns one set of scratch buffers (two streams driving aesgcm_shard_crypt_dev on ONE context would corrupt
 *     each other's GHASH scratch).  The batch entry points have no context; launches on different streams may
 *     overlap, up to 256 of them in flight per device (each takes the next slot of a 256-entry ring of packet
 *     dispensers, zeroed on its own stream).
 *   - IV is always 96 bits (src/gcm_pkg.vhd:15-17, tb/gcm_gctr.py:251); tag is the full 128 bits
 *     (src/gcm_ghash.vhd:293).
 *   - length rule: data <= 2^36 - 32 bytes (the 32-bit block counter stops at all-ones,
 *     src/aes_icb.vhd:114) and AAD blocks + data blocks < 2^36 -> AESGCM_ETOOLONG otherwise.
 */
#ifndef AESGCM_H
#define AESGCM_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define AESGCM_ABI_VERSION 5   /* 5 (round 6): calls with offset arrays and aesgcm_messages_crypt_dev are ROUTED per message on the device (AESGCM_SHAPE_MIXED; pkt_len is no longer a hint),
                                  aesgcm_ctx_status (what an asynchronous call could not say when it returned), aesgcm_stream_export / _import / _update_dev, aesgcm_frames_ceiling_probe_dev,
                                  aesgcm_mgpu_last_tags collects the OLDEST queued messages;
                                  4 (round 5): packets of message size by rows (AESGCM_SHAPE_ROWS), aesgcm_messages_crypt_dev, aesgcm_ctx_last_launch, aesgcm_wipe_failed_dev and the option "wipe_on_auth_fail",
                                  aesgcm_mgpu_crypt_dev with tag = NULL + aesgcm_mgpu_last_tags / aesgcm_mgpu_sync;
                                  2: aesgcm_ctx_wait, aesgcm_comm_* / aesgcm_mgpu_*, packet and batch entry points; 3: aesgcm_ctx_set_option (the library no longer reads
                                  any environment variable), aesgcm_batch_shape / aesgcm_packets_shape, aesgcm_mgpu_ctx, aesgcm_last_tag through the host slot */

#if defined(__GNUC__)
#define AESGCM_API __attribute__((visibility("default")))
#else
#define AESGCM_API
#endif

#define AESGCM_OK        0
#define AESGCM_EARG     (-1)   /* NULL/invalid argument                                        */
#define AESGCM_EKEYLEN  (-2)   /* key length not 16/24/32 (aes_pkg.vhd:60-63 modes 128/192/256) */
#define AESGCM_EIVLEN   (-3)   /* reserved: IV is fixed at 12 bytes by the signatures            */
#define AESGCM_ETOOLONG (-4)   /* message exceeds the counter space (aes_icb.vhd:114)            */
#define AESGCM_EAUTH    (-5)   /* tag mismatch on decrypt (tb/gcm_model.py:47 ValueError branch) */
#define AESGCM_EHIP     (-6)   /* HIP runtime error / no device; see aesgcm_last_error()         */
#define AESGCM_ENOMEM   (-7)
#define AESGCM_ESTATE   (-8)   /* streaming call out of order (AAD after data, ragged chunk)     */
#define AESGCM_EALIGN   (-9)   /* device data pointer not 16-byte aligned                        */
#define AESGCM_ERCCL    (-10)  /* RCCL missing or a collective failed; see aesgcm_comm_last_error() */

typedef struct aesgcm_ctx aesgcm_ctx;

/* ---------------------------------------------------------------- library / device */
AESGCM_API int         aesgcm_abi_version(void);
AESGCM_API const char *aesgcm_strerror(int code);
AESGCM_API const char *aesgcm_last_error(void);              /* thread-local detail of the last AESGCM_EHIP */
AESGCM_API int         aesgcm_device_count(int *n);
AESGCM_API int         aesgcm_device_name(int device, char *buf, size_t buflen);

/* ---------------------------------------------------------------- unit-level entry points
 * One per arithmetic block of the RTL so that each can be parity-tested in isolation.  All run
 * on the GPU (small kernels), results are copied back to the host buffers given. */

/* FIPS-197 KeyExpansion.  Replaces aes_kexp (config/config_aes_kexp.py:113-159, window update
 * :189-219) and its software twin tb/key_exp.py:79-121 aes_expand_key.  rk receives
 * 16*(nr+1) bytes, stage i = bytes 16i..16i+15 (the layout load_pre_exp_key streams,
 * tb/gcm_gctr.py:199-207); *nr = 10/12/14 (aes_pkg.vhd:31-33). */
AESGCM_API int aesgcm_key_expand(int device, const uint8_t *key, size_t key_len, uint8_t rk[240], int *nr);

/* nblocks independent ECB encryptions under the context key through the same LDS T-table round
 * code the CTR kernel uses.  Replaces aes_round x Nr + aes_last_round
 * (config/config_aes_round.py:120-126, src/aes_last_round.vhd:76) as instantiated by aes_ecb
 * (config/config_aes_ecb.py:250-327). */
AESGCM_API int aesgcm_ecb_encrypt(aesgcm_ctx *ctx, const uint8_t *in, size_t nblocks, uint8_t *out);

/* n independent GF(2^128) products z[i] = x[i] * h[i] (16-byte big-endian blocks, GCM bit order).
 * Replaces ghash_gfmul (src/ghash_gfmul.vhd:37-64). */
AESGCM_API int aesgcm_gfmul(int device, const uint8_t *h, const uint8_t *x, uint8_t *z, size_t n);

/* GHASH chaining value after absorbing `len` bytes (last block zero-padded) from Y = 0 under the
 * context's H = E_K(0^128):  Y_i = (Y_{i-1} xor X_i) * H  (src/gcm_ghash.vhd:174-186, :259-272),
 * WITHOUT the length block.  Computed by the parallel H-power path, not by serial Horner. */
AESGCM_API int aesgcm_ghash(aesgcm_ctx *ctx, const uint8_t *data, size_t len, uint8_t y[16]);

/* H = E_K(0^128) as latched by gcm_ghash (src/gcm_gctr.vhd:141-144, src/gcm_ghash.vhd:128-139). */
AESGCM_API int aesgcm_get_h(aesgcm_ctx *ctx, uint8_t h[16]);

/* ---------------------------------------------------------------- context
 * A context = (device, expanded key, H, H-power tables, scratch).  Creating it runs the on-GPU key
 * expansion and table build once per key (the RTL's "load key" phase, tb/gcm_gctr.py:144-175). */
AESGCM_API int aesgcm_ctx_create(aesgcm_ctx **out, int device, const uint8_t *key, size_t key_len);
/* Pre-expanded key load path (config/config_aes_kprexp.py:66-106, tb/gcm_gctr.py:180-214):
 * rk = 16*(nr+1) bytes exactly as aesgcm_key_expand / tb/key_exp.py produce them. */
AESGCM_API int aesgcm_ctx_create_preexpanded(aesgcm_ctx **out, int device, const uint8_t *rk, int nr);
/* A new key for an existing context: the reference core's "load key" between frames (tb/gcm_gctr.py:144-175; H is recomputed only then,
 * src/gcm_gctr.vhd:142-144).  The context keeps its stream, scratch, host slot and options; only the key schedule, H and the H-power tables
 * are rebuilt (0.4 ms, half of what destroying the context and creating another costs).  Waits for the context's queued
 * work first; AESGCM_ESTATE inside an open aesgcm_stream_* session. */
AESGCM_API int aesgcm_ctx_rekey(aesgcm_ctx *ctx, const uint8_t *key, size_t key_len);
AESGCM_API int aesgcm_ctx_destroy(aesgcm_ctx *ctx);
AESGCM_API int aesgcm_ctx_device(const aesgcm_ctx *ctx);
/* Which launch structure the context's last whole-message call (aesgcm_encrypt[_dev] / aesgcm_decrypt[_dev]) took.  The library chooses by size and -- for the
 * half shape of the cyclic rows -- by whether another context of the device had a message under way at that moment; benches and profiles report it. */
#define AESGCM_LAUNCH_NONE 0
#define AESGCM_LAUNCH_MAIN 1            /* k_main (+ k_combine): below 64 KiB */
#define AESGCM_LAUNCH_CYCLIC 2          /* one k_body launch of cyclic rows */
#define AESGCM_LAUNCH_CYCLIC_HALF 3     /* ... in its half shape (k_bodyh) */
#define AESGCM_LAUNCH_DEALT 4           /* k_body's dealt chunks (+ k_fold): from 1 GiB */
AESGCM_API int aesgcm_ctx_last_launch(const aesgcm_ctx *ctx, int *shape);
/* Tunables of one context, for tests and profiling scripts; the library reads no environment variable and the defaults are the
 * measured best.  Every value selects between paths that produce the same bytes.  Keys (value >= 0):
 *   "tw"          rows of 64 blocks per chunk of the dealt kernels, 0 = the library's rule
 *   "body_min"    bytes from which a range's aligned middle goes through k_body's dealt chunks (>= 2^60: never, nor cyclic rows)
 *   "cyc_min", "cyc_max"   bytes: ranges in [cyc_min, cyc_max) take k_body's cyclic rows (one launch per message); both 0 = never
 *   "cyc_close"   1: that launch closes the tag itself; 0: k_fold + k_combine behind it
 *   "cyc_half"    whole messages below 80 MiB take the launch in its half shape (256 workgroups of 512 lanes, two per CU), in which one message's
 *                 table staging and closing run beside another's rows: 0 never, 1 always, 2 (default) when another context of the device has a
 *                 message under way at the moment of the call (its host slot does not yet show its last launch) -- i.e. for callers that keep
 *                 messages in flight on contexts of their own; a single message alone on the chip is slower in that shape
 *   "fold_close"  1: behind the dealt k_body a k_fold level closes the tag; 0: further levels and k_combine
 *   "cyc_prio"    rows between rotations of the waves' issue priorities in a cyclic launch, 0 = off
 *   "pkt_order"   accepted and ignored since round 6 (it was the packet count from which a call with offset arrays took its packets by falling length class: the
 *                 routing sort of such a call makes that order anyway)
 *   "wipe_on_auth_fail"  1: a decrypt call that verifies a tag (expect_tag / d_expect_tags) leaves ZEROS, not unauthenticated plaintext, where verification fails:
 *                 aesgcm_decrypt and aesgcm_decrypt_pipelined wipe the caller's buffer (aesgcm_decrypt does not even copy the plaintext out before the tag is
 *                 checked), aesgcm_decrypt_dev the device buffer, aesgcm_packets_crypt_dev / aesgcm_messages_crypt_dev every packet whose d_auth entry is 0 (d_auth must be given:
 *                 AESGCM_EARG for a decrypt call with d_expect_tags and without d_auth while the option is on).  Default 0:
 *                 the reference model returns the plaintext and raises (tb/gcm_model.py:29-30,47-51), and so does the class that mirrors it.
 *   "rows_min"    bytes per packet from which aesgcm_packets_crypt_dev goes by rows (default 8192; fixed-size records: from a quarter of it while the packets are at most 16384; 0 = never).
 *                 With offset arrays the mark is applied per message on the device, to data + AAD, in steps of 64 bytes and up to 16320 (see "route_mid_min")
 *   "route_mid_min", "route_blocks_min"   how a call with offset arrays is routed on the device: the mark is "rows_min" when at least route_mid_min (65536) of its messages lie
 *                 between a quarter of rows_min and rows_min, else that quarter; and nothing goes to the packet kernels at all while the messages below the mark hold
 *                 fewer than route_blocks_min (2^17) + 3.5 per message 16-byte blocks between them -- unless the call has at most 4096 messages and none above the mark: that is one packet
 *                 launch against three row launches (csrc/aesgcm_kernels.hip route_decide has the measurements).  0 / 0: always the high mark, always split
 *   "route_top_min"   ... and the mark rises to 16320 bytes, the last size the sort resolves, when at least this many messages (458752) lie between "rows_min" and it; 0 = never
 *   "rows_block"  units (rows of 64 blocks) per dealt block of the row kernel, 0 = the library's cut (one block per wave; blocks of 64 for large calls)
 *   "poll_us"     how long a tag is polled for in the pinned host slot before the call blocks in the runtime
 * AESGCM_EARG for an unknown key. */
AESGCM_API int aesgcm_ctx_set_option(aesgcm_ctx *ctx, const char *key, int64_t value);
/* the context's own HIP stream (what `stream = NULL` means everywhere): lets a caller order other work -- the
 * aesgcm_comm_allgather_dev of the partials -- behind the context's kernels without a host synchronisation */
AESGCM_API int aesgcm_ctx_stream(const aesgcm_ctx *ctx, void **stream);
/* Stream-order two contexts of one device without a host synchronisation: what is enqueued on ctx's own stream AFTER this call
 * starts only when everything enqueued on other's own stream BEFORE it has completed.  A context owns one scratch set, so
 * back-to-back messages (the shards of bench.py's N > 1 step) alternate between two contexts of the same key: message m+1's
 * fused kernel then runs while message m's fold / combine kernels drain; the step that consumes both (the all-gather of the
 * partials) is ordered with this call.  No reference counterpart (the RTL has one pipeline, src/aes_gcm.vhd). */
AESGCM_API int aesgcm_ctx_wait(aesgcm_ctx *ctx, aesgcm_ctx *other);
/* The same, but only up to other's most recently enqueued FUSED kernel (the AES-CTR + GHASH launch), not the fold / combine
 * launches behind it: chaining message m+1 (on ctx) to message m (on other) this way runs the fused kernels back to back and
 * lets every fold / combine tail but the last hide behind the next message.  Tracking starts with the first call (which,
 * like any call made before other has launched a fused kernel, waits for nothing). */
AESGCM_API int aesgcm_ctx_wait_fused(aesgcm_ctx *ctx, aesgcm_ctx *other);
/* What an ASYNCHRONOUS call could not say when it returned (round 6).  aesgcm_packets_crypt_dev with offset arrays and aesgcm_messages_crypt_dev read their lengths on the
 * device, behind the call: when the plan kernel finds one it cannot take -- the RTL raises a flag when its counter cannot go on (src/aes_icb.vhd:65,98,114,119) -- NOTHING
 * of that call runs (outputs, tags and d_auth are left as they were) and the reason goes to a status word in the context's pinned host memory, next to its tag slot:
 *   AESGCM_STATUS_LENGTH   a message's data or AAD is 2^28 bytes or more, or an offset array does not rise; *detail = the first such message
 *   AESGCM_STATUS_PLAN     the call's plan does not fit the scratch the library sized for it; *detail = the record slots it asked for
 *   AESGCM_STATUS_UNITS    more rows than 32-bit block numbers hold; *detail = the rows
 * aesgcm_ctx_status reads the word and clears it (no wait: synchronise the stream the call ran on first -- a status is there when that stream has drained); 0 = nothing
 * to report.  While a status is unread, aesgcm_last_tag and aesgcm_ctx_wait on the context return AESGCM_ETOOLONG (LENGTH, UNITS) or AESGCM_ESTATE (PLAN). */
#define AESGCM_STATUS_OK     0
#define AESGCM_STATUS_PLAN   1
#define AESGCM_STATUS_LENGTH 2
#define AESGCM_STATUS_UNITS  3
AESGCM_API int aesgcm_ctx_status(aesgcm_ctx *ctx, int *code, uint64_t *detail);

/* ---------------------------------------------------------------- whole messages, host pointers
 * Replaces the model's update/encrypt/digest sequence (tb/gcm_model.py:21-35) i.e. the aes_gcm
 * top level in encrypt mode (src/aes_gcm.vhd:207-211: GHASH consumes the GCTR output).
 * Copies H2D/D2H around the device path below; meant for parity tests and small messages. */
AESGCM_API int aesgcm_encrypt(aesgcm_ctx *ctx, const uint8_t iv[12], const uint8_t *aad, size_t aad_len,
                   const uint8_t *pt, size_t len, uint8_t *ct, uint8_t tag[16]);
/* Decrypt mode (tb/gcm_model.py:29-30,43-51; src/aes_gcm.vhd:207-211: GHASH consumes the input).
 * Plaintext is always written (the model emits data before the tag is checked).  tag_out (may be
 * NULL) receives the computed tag.  If expect_tag != NULL it is compared in constant time and
 * AESGCM_EAUTH is returned on mismatch. */
AESGCM_API int aesgcm_decrypt(aesgcm_ctx *ctx, const uint8_t iv[12], const uint8_t *aad, size_t aad_len,
                   const uint8_t *ct, size_t len, uint8_t *pt, const uint8_t *expect_tag, uint8_t tag_out[16]);

/* ---------------------------------------------------------------- whole messages, device pointers
 * The benchmarked path: d_in/d_out are device pointers (16-byte aligned, may alias for in-place),
 * d_aad a device pointer (any alignment) or NULL.  Work is enqueued on `stream` (a hipStream_t
 * passed as void*, NULL = context stream); the only host traffic is the 16-byte tag.  When the call
 * returns with a tag, the whole result is in device memory: the kernel that finishes the tag stores it
 * in a pinned host slot behind the data (messages of 64 KiB .. 1 GiB: the one launch that encrypts
 * the message, whose ciphertext stores go through the L2 for that), and the call returns when the
 * slot shows it -- long work falls back to a stream synchronisation.  The launch itself may retire a
 * few microseconds later; anything ordered behind it on `stream`, hipStreamSynchronize and blocking
 * copies see it complete as always.  Pass tag = NULL to skip the wait: the call only enqueues, and
 * aesgcm_last_tag() later collects the tag of the context's most recent message the same way (a poll of
 * the host slot) -- several contexts of one key, each with its own stream, keep several messages in
 * flight that way (bench.py --inflight). */
AESGCM_API int aesgcm_encrypt_dev(aesgcm_ctx *ctx, const uint8_t iv[12], const void *d_aad, size_t aad_len,
                       const void *d_pt, size_t len, void *d_ct, uint8_t tag[16], void *stream);
AESGCM_API int aesgcm_decrypt_dev(aesgcm_ctx *ctx, const uint8_t iv[12], const void *d_aad, size_t aad_len,
                       const void *d_ct, size_t len, void *d_pt, const uint8_t *expect_tag,
                       uint8_t tag_out[16], void *stream);
/* (`stream` must be the stream the message was enqueued on: AESGCM_ESTATE if that stream has drained and the host slot still does not show the message's tag) */
AESGCM_API int aesgcm_last_tag(aesgcm_ctx *ctx, uint8_t tag[16], void *stream);

/* CTR keystream blocks [first_block, first_block+nblocks): E_K(IV || (2+i) mod 2^32)
 * (src/aes_icb.vhd:97-100,118; src/gcm_gctr.vhd:150 before the xor). */
AESGCM_API int aesgcm_keystream(aesgcm_ctx *ctx, const uint8_t iv[12], uint64_t first_block, uint64_t nblocks, uint8_t *out);
AESGCM_API int aesgcm_keystream_dev(aesgcm_ctx *ctx, const uint8_t iv[12], uint64_t first_block, uint64_t nblocks,
                         void *d_out, void *stream);

/* ---------------------------------------------------------------- one message sharded over ranks
 * (no reference counterpart: the RTL is one pipeline; the algebra is gcm_ghash.vhd:317-333's
 * linear split generalised).  Rank g owns data blocks [first_block, first_block + ceil(len/16)) of a
 * message with total_len bytes of data and aad_len bytes of AAD; only the LAST shard may have a
 * ragged length.  The rank that owns first_block == 0 also absorbs the AAD (pass d_aad there, NULL
 * elsewhere).  The call en/decrypts the shard and writes its 16-byte WEIGHTED GHASH partial
 *    W_g = (sum_{i in shard} X_i * H^(end_g-1-i)) * H^(n_total_blocks - end_g)
 * to d_partial (device memory).  Partials of all ranks are exchanged by the caller (one 16-byte
 * all-gather, e.g. RCCL) and handed to aesgcm_shard_finalize, which XOR-folds them on the device and
 * produces tag = GHASH xor E_K(IV||1). */
AESGCM_API int aesgcm_shard_crypt_dev(aesgcm_ctx *ctx, int decrypt, const uint8_t iv[12],
                           const void *d_aad, size_t aad_len,
                           const void *d_in, size_t len, void *d_out,
                           uint64_t first_block, uint64_t total_len,
                           void *d_partial, void *stream);
AESGCM_API int aesgcm_shard_finalize_dev(aesgcm_ctx *ctx, const uint8_t iv[12], const void *d_partials, size_t n_partials,
                              size_t aad_len, uint64_t total_len, uint8_t tag[16], void *stream);
/* The same with the n_partials partials stride_bytes apart (a multiple of 16): ONE all-gather of M messages' partials
 * leaves them as [rank][message][16]; message m is finalised from d_partials + 16 m with stride 16 M. */
AESGCM_API int aesgcm_shard_finalize_strided_dev(aesgcm_ctx *ctx, const uint8_t iv[12], const void *d_partials, size_t n_partials,
                              size_t stride_bytes, size_t aad_len, uint64_t total_len, uint8_t tag[16], void *stream);

/* The tags of n_msgs (<= 8) messages after ONE all-gather, in one launch and one wait: message m has IV ivs + 12 m, lengths
 * aad_lens[m] (NULL = all zero) / total_lens[m], its n_partials partials start at d_partials + m * msg_stride_bytes and lie
 * stride_bytes apart; tags receives n_msgs * 16 bytes.  With the [rank][message][16] layout: msg_stride_bytes = 16,
 * stride_bytes = 16 * n_msgs.  Same arithmetic as aesgcm_shard_finalize_strided_dev called n_msgs times. */
AESGCM_API int aesgcm_shard_finalize_batch_dev(aesgcm_ctx *ctx, size_t n_msgs, const uint8_t *ivs, const void *d_partials, size_t n_partials,
                              size_t stride_bytes, size_t msg_stride_bytes, const size_t *aad_lens, const uint64_t *total_lens,
                              uint8_t *tags, void *stream);

/* ---------------------------------------------------------------- the exchange step, in the library
 * (BASELINE north_star "a single RCCL reduce of per-shard partial tags over xGMI".)  RCCL has
 * no XOR reduction (rccl.h ncclRedOp_t), so the 16-byte partials are all-gathered and folded on the device by
 * aesgcm_shard_finalize_dev.  RCCL is loaded with dlopen("librccl.so.1") on first use; without it these return
 * AESGCM_ERCCL and everything else in the library still works.
 *
 * One PROCESS per GPU: rank 0 calls aesgcm_comm_unique_id and hands the 128 bytes to the other ranks by any means
 * (bench.py: a file under /tmp keyed by the launcher's pid); every rank then calls aesgcm_comm_create, which is
 * ncclCommInitRank on `device`.  aesgcm_comm_ranks returns what RCCL reports (ncclCommCount / ncclCommUserRank).
 * aesgcm_comm_allgather_dev: d_recv[r * bytes_per_rank ..] = rank r's d_send, asynchronous on `stream`.
 * aesgcm_comm_allreduce_f64: one host double, op 0 = max, 1 = min, 2 = sum, synchronous (bench timing);
 * aesgcm_comm_barrier is the sum of ones. */
#define AESGCM_COMM_ID_BYTES 128
typedef struct aesgcm_comm aesgcm_comm;
AESGCM_API const char *aesgcm_comm_last_error(void);
AESGCM_API int aesgcm_comm_unique_id(uint8_t id[AESGCM_COMM_ID_BYTES]);
AESGCM_API int aesgcm_comm_create(aesgcm_comm **out, int device, const uint8_t id[AESGCM_COMM_ID_BYTES], int n_ranks, int rank);
AESGCM_API int aesgcm_comm_ranks(const aesgcm_comm *comm, int *n_ranks, int *rank);
AESGCM_API int aesgcm_comm_allgather_dev(aesgcm_comm *comm, const void *d_send, void *d_recv, size_t bytes_per_rank, void *stream);
AESGCM_API int aesgcm_comm_allreduce_f64(aesgcm_comm *comm, double *value, int op);
AESGCM_API int aesgcm_comm_barrier(aesgcm_comm *comm);
AESGCM_API int aesgcm_comm_destroy(aesgcm_comm *comm);

/* One process driving ndev GPUs (ncclCommInitAll).  Shard g = d_in[g] / d_out[g] (device memory of devices[g]),
 * shard_len[g] bytes, owning the message's blocks right after shard g-1's; every length but the last must be a
 * multiple of 16.  Each device expands the key and builds H and its tables itself (nothing is broadcast); the AAD
 * (device memory of devices[0]) is absorbed by shard 0.  The call en/decrypts all shards concurrently, performs ONE
 * grouped ncclAllGather of 16 bytes per device, folds on devices[0] and returns the tag; all streams are
 * synchronised on return.  aesgcm_mgpu_ranks returns the communicator size RCCL reports. */
typedef struct aesgcm_mgpu aesgcm_mgpu;
AESGCM_API int aesgcm_mgpu_create(aesgcm_mgpu **out, int ndev, const int *devices, const uint8_t *key, size_t key_len);
AESGCM_API int aesgcm_mgpu_ranks(const aesgcm_mgpu *m, int *n_ranks);
AESGCM_API int aesgcm_mgpu_ctx(aesgcm_mgpu *m, int g, aesgcm_ctx **out);     /* device g's context, borrowed: never destroy it */
AESGCM_API int aesgcm_mgpu_crypt_dev(aesgcm_mgpu *m, int decrypt, const uint8_t iv[12], const void *d_aad_on_dev0, size_t aad_len,
                          const void *const *d_in, const size_t *shard_len, void *const *d_out, uint8_t tag[16]);
/* tag = NULL in aesgcm_mgpu_crypt_dev only ENQUEUES the message -- shards, the all-gather -- without a host synchronisation on any device; up to 8 such messages may
 * wait.  The queue is a FIFO: aesgcm_mgpu_last_tags finalizes the OLDEST n of them in one launch on devices[0] and returns their tags in the order they were queued (16 n bytes;
 * n may be less than what waits: the rest stays queued); aesgcm_mgpu_sync drains every device's stream (before the outputs are read by anything not ordered behind those
 * streams).  AESGCM_ESTATE when a ninth message is queued, and when a call with tag != NULL is made while messages wait (it would have to jump the queue). */
AESGCM_API int aesgcm_mgpu_last_tags(aesgcm_mgpu *m, size_t n, uint8_t *tags);
AESGCM_API int aesgcm_mgpu_sync(aesgcm_mgpu *m);
AESGCM_API int aesgcm_mgpu_destroy(aesgcm_mgpu *m);

/* ---------------------------------------------------------------- many packets under the context's key
 * The RTL keeps H across packets while no new key is loaded (src/gcm_gctr.vhd:142-144) and takes a new IV per
 * packet (src/aes_icb.vhd:60-70 "load IV"): this is that mode.  Per packet: ivs[p] (12 bytes), optional AAD
 * and data either as fixed-size records (aad_len / pkt_len, offset arrays NULL) or delimited by uint64 offset
 * arrays with n_pkts + 1 entries (then aad_len / pkt_len are ignored); tags[p] receives the computed tag; for
 * decrypt d_auth[p] (optional) = 1 if it equals d_expect_tags[p].  Asynchronous on `stream`.
 * Two families of kernels do the work.  Packets of message size -- from 8 KiB each (context option "rows_min"; from 2 KiB unless there are many between the two), up to
 * 2^28 - 1 bytes -- go BY ROWS (round 5): the 64-block rows of all such messages are one pool of work for the row loop a single large message runs through
 * (csrc/aesgcm_rows.h), and one small launch behind it takes what is not a whole row -- headers, ragged ends -- block by block and closes every tag; 4096 x 1 MiB then runs
 * at the rate of one 4 GiB message.  Shorter packets take the packet kernels: a lane, or a group of 4 .. 16 lanes, per packet.
 * FIXED-SIZE records: the host knows the one size and the whole call goes one way (many records of 8 .. 16 KiB whose last partial row is longer than 4 blocks stay with the
 * packet kernels).  OFFSET ARRAYS: the lengths are on the device, and so is the choice -- every message is ROUTED BY ITS OWN SIZE (data + AAD) inside the one call (round 6):
 * a counting sort by size class on the device (three small launches on `stream`) splits the call at the mark, hands the short messages to the packet kernels longest first
 * (the lanes of a wave run to the longest packet among them) and the others to the rows; which path a message takes never shows in its bytes or its tag.  The reference's
 * own traffic is of both kinds at once (tb/gcm_gctr.py:279-281: lengths from a U-shaped distribution).  pkt_len is IGNORED with offset arrays (until round 5 it was a hint
 * that sent the whole call one way).  Lengths of 2^28 bytes or more, or offsets that do not rise, are found on the device: nothing of the call runs then, see aesgcm_ctx_status.
 * CAPTURE: for given pointers and count the host's side of a routed call is a fixed sequence of launches on `stream` and the context's side stream (forked and joined by
 * events).  After one ordinary call of the same context with sizes at least as large, this call and aesgcm_messages_crypt_dev allocate nothing, wait for nothing and read
 * nothing back, so they may run under hipStreamBeginCapture on `stream`; the graph may be replayed over other lengths, offsets and bytes (examples/graph_replay.cpp,
 * tests/test_gpu_mixed.py; on ROCm 7.2 a replay costs what the direct call costs). */
AESGCM_API int aesgcm_packets_crypt_dev(aesgcm_ctx *ctx, int decrypt, size_t n_pkts, const void *d_ivs,
                             const void *d_aad, size_t aad_len, const uint64_t *d_aad_off,
                             const void *d_in, size_t pkt_len, const uint64_t *d_data_off, void *d_out,
                             void *d_tags, const void *d_expect_tags, int *d_auth, void *stream);

/* Messages WHEREVER THEY LIVE under the context's key (round 5): the same work as aesgcm_packets_crypt_dev with offset arrays, but every message has its own buffers --
 * the reference's harness hands the core one frame after the other, each its own object (tb/gcm_test.py:76-85, tb/gcm_gctr.py:233-276); a caller with a queue
 * of messages in separate allocations has exactly that, and copying them into one buffer to batch them would cost what the batch saves.  All arrays are in
 * device memory, n_msgs entries each: d_in_ptr / d_out_ptr device addresses of the messages' input and output (in == out allowed), d_len their lengths
 * (each < 2^28 bytes), d_aad_ptr / d_aad_len the same for AAD (both NULL: none); d_ivs n_msgs x 12 bytes, d_tags n_msgs x 16; decrypt: d_expect_tags / d_auth
 * as aesgcm_packets_crypt_dev.  Routed per message like the offset-array form (round 6: until then always by rows, and a call of tiny messages paid for it); the
 * context option "wipe_on_auth_fail" applies; a length of 2^28 or more is reported through aesgcm_ctx_status.  Asynchronous on `stream`. */
AESGCM_API int aesgcm_messages_crypt_dev(aesgcm_ctx *ctx, int decrypt, size_t n_msgs, const void *d_ivs,
                              const uint64_t *d_aad_ptr, const uint32_t *d_aad_len,
                              const uint64_t *d_in_ptr, const uint32_t *d_len, const uint64_t *d_out_ptr,
                              void *d_tags, const void *d_expect_tags, int *d_auth, void *stream);

/* ---------------------------------------------------------------- batch: independent packets, per-packet key + IV
 * (BASELINE config 5; the RTL equivalent is reloading key and IV between packets, tb/gcm_gctr.py:144-175,
 * with aes_kexp run per packet, config/config_aes_kexp.py:113-159.)  All arrays are contiguous device memory:
 * keys[n][key_len], ivs[n][12], aad[n][aad_len] (or NULL), in[n][pkt_len], out[n][pkt_len], tags[n][16].
 * One wave per packet; key expansion, H, E_K(J0) and the GHASH tables are rebuilt per packet on the GPU.
 * decrypt != 0: GHASH runs over the input, tags[] receives the COMPUTED tags; if d_auth != NULL it receives
 * one int per packet (1 = equals d_expect_tags[p], 0 = mismatch; all 1 when d_expect_tags is NULL).
 * Asynchronous on `stream`; in == out is allowed. */
AESGCM_API int aesgcm_batch_crypt_dev(int device, int decrypt, size_t n_pkts, size_t key_len, const void *d_keys, const void *d_ivs,
                           const void *d_aad, size_t aad_len, const void *d_in, size_t pkt_len, void *d_out,
                           void *d_tags, const void *d_expect_tags, int *d_auth, void *stream);

/* Zero the output of every packet whose d_auth entry is 0: what the context option "wipe_on_auth_fail" does behind aesgcm_packets_crypt_dev, for callers of the
 * context-free batch entry points (fixed-size records: d_data_off = NULL).  Asynchronous on `stream`, which must be the stream the decrypt call ran on. */
AESGCM_API int aesgcm_wipe_failed_dev(int device, size_t n_pkts, void *d_out, size_t pkt_len, const uint64_t *d_data_off, const int *d_auth, void *stream);

/* Measurement support: the batch kernel's instruction stream without the data's loads and stores (keys, IVs and tags still move) over n_pkts virtual packets of
 * pkt_len bytes -- the ceiling of its formulation (per-packet aes_kexp, T-table AES, Shoup GHASH) on this chip at this moment's clocks; the caller times it
 * (aesgcm_timer_*).  Only for calls that take the 8-lanes-per-packet shape, as BASELINE config 5 does (AESGCM_EARG otherwise). */
AESGCM_API int aesgcm_batch_ceiling_probe_dev(int device, size_t n_pkts, size_t key_len, const void *d_keys, const void *d_ivs, size_t pkt_len, void *d_tags, void *stream);

/* The same for the frame path (round 6): the packet kernels' instruction stream over n_pkts frames delimited by d_data_off (and d_aad_off: or NULL) as aesgcm_packets_crypt_dev
 * takes them, WITHOUT the data's loads and stores (IVs, offsets, AAD and tags still move; no data buffer is passed) -- every frame to the packet kernels, in the shape the
 * device chooses for the count.  bench.py --config frames prints it as roofline.formulation_ceiling.  Asynchronous on `stream`; the caller times it. */
AESGCM_API int aesgcm_frames_ceiling_probe_dev(aesgcm_ctx *ctx, size_t n_pkts, const void *d_ivs, const void *d_aad, const uint64_t *d_aad_off, const uint64_t *d_data_off, void *d_tags, void *stream);

/* Which kernel shape a call with these arguments takes: lanes per packet (1 = one lane per packet, 4 / 8 / 16 = a lane group, 64 = a
 * whole wave; aesgcm_packets_shape: AESGCM_SHAPE_ROWS = by rows, every message over the whole chip).  var_len != 0 describes the offset-array forms:
 * aesgcm_batch_shape goes by count (the host does not know the lengths); aesgcm_packets_shape answers AESGCM_SHAPE_MIXED -- every message is routed by
 * its own size on the device, by rows or to the packet kernel shape chosen there for the count of short ones (pkt_len is ignored). */
#define AESGCM_SHAPE_ROWS (1 << 20)
#define AESGCM_SHAPE_MIXED (1 << 21)
/* ... and what the device DID decide for the context's most recent call of that kind (waits for the device): out[0] = the mark in bytes -- messages of at least that size
 * (data + AAD) went by rows; 0 = all of them, 0xFFFFFFFF = none --, out[1] = messages that took the packet kernels, out[2] = lanes per packet of the packet kernel shape
 * chosen for that count (0: no packet launch did anything), out[3] = units (rows, long tails, long AADs) of the row launch.  AESGCM_ESTATE before the first such call. */
AESGCM_API int aesgcm_ctx_last_route(aesgcm_ctx *ctx, uint64_t out[4]);
AESGCM_API int aesgcm_batch_shape(int device, size_t n_pkts, size_t pkt_len, int var_len, int *lanes_per_packet);
AESGCM_API int aesgcm_packets_shape(const aesgcm_ctx *ctx, size_t n_pkts, size_t pkt_len, int var_len, int *lanes_per_packet);

/* Variable-length form (MACsec-shaped traffic like the reference's README vectors: short frames with a
 * per-frame header as AAD, README.md:251-257): packet p occupies bytes [d_data_off[p], d_data_off[p+1]) of
 * in/out and, when d_aad_off != NULL, bytes [d_aad_off[p], d_aad_off[p+1]) of aad.  Offset arrays have
 * n_pkts + 1 uint64 entries in device memory.  Each packet < 2^28 bytes.  Packets whose data offset is a
 * multiple of 16 take the aligned fast path.  From 262144 packets (AES-128; 98304 for the longer keys) the launch takes them by falling
 * length class, as aesgcm_packets_crypt_dev does (scratch: 4 bytes per packet, kept per device). */
AESGCM_API int aesgcm_batch_crypt_var_dev(int device, int decrypt, size_t n_pkts, size_t key_len, const void *d_keys, const void *d_ivs,
                               const void *d_aad, const uint64_t *d_aad_off, const void *d_in, const uint64_t *d_data_off,
                               void *d_out, void *d_tags, const void *d_expect_tags, int *d_auth, void *stream);

/* ---------------------------------------------------------------- streaming (beat-by-beat) interface
 * Mirrors the call order the reference harness drives its model with (tb/gcm_test.py:76-85 ->
 * tb/gcm_model.py:21-35): all AAD first, then data; every chunk except the last of its kind must be
 * a multiple of 16 bytes (the harness sends 16-byte beats, tb/gcm_sequencer.py:129-140).  Output for a
 * chunk is complete when the call returns.  State (running GHASH value, block counter) lives on the
 * device between calls, and can be exported and imported (below). */
AESGCM_API int aesgcm_stream_begin(aesgcm_ctx *ctx, const uint8_t iv[12], int decrypt);
AESGCM_API int aesgcm_stream_aad(aesgcm_ctx *ctx, const uint8_t *aad, size_t len);
AESGCM_API int aesgcm_stream_update(aesgcm_ctx *ctx, const uint8_t *in, size_t len, uint8_t *out);
AESGCM_API int aesgcm_stream_final(aesgcm_ctx *ctx, uint8_t tag[16]);
/* The same step on DEVICE pointers (round 6): d_in / d_out 16-byte aligned, may alias; asynchronous on `stream` (NULL = the context's own; the chunks of one session must be
 * stream-ordered, as every call on a context).  A chunk of any size takes the launch structure a shard of that size takes, so a message of unknown total length that is
 * already on the GPU runs at the rate of aesgcm_shard_crypt_dev.  Every chunk but the last a multiple of 16 bytes. */
AESGCM_API int aesgcm_stream_update_dev(aesgcm_ctx *ctx, const void *d_in, size_t len, void *d_out, void *stream);
/* The state of the open session as 64 bytes the caller can keep, move and pick up again -- in another context of the same key, on another device, in another process
 * (SURVEY.md 5 "checkpoint / resume", 8(f2)): what the RTL holds in its Y register (src/gcm_ghash.vhd:174-186) and its counter (src/aes_icb.vhd:97-100) and cannot hand out.
 * blob: version, direction, IV, AAD and data bytes so far, GHASH blocks so far, the running GHASH value (in the library's form: the RTL's Y divided by H), a four-byte key
 * check (E_K of a constant block; NOT key material -- no key, no H and no table is in the blob) and a sum check.  The running GHASH value depends on the key and the data
 * like a tag before its final XOR: handle the blob as you would the tag-in-progress.  aesgcm_stream_export waits for everything the session has enqueued (a device
 * synchronisation) and leaves the session open; aesgcm_stream_import opens a session in `ctx` at exactly that point (AESGCM_ESTATE if one is open already; AESGCM_EARG for
 * a blob that is damaged, of another version, or exported under another key), after which aesgcm_stream_aad / _update / _update_dev / _final go on as if nothing had happened. */
#define AESGCM_STREAM_STATE_BYTES 64
AESGCM_API int aesgcm_stream_export(aesgcm_ctx *ctx, uint8_t blob[AESGCM_STREAM_STATE_BYTES]);
AESGCM_API int aesgcm_stream_import(aesgcm_ctx *ctx, const uint8_t blob[AESGCM_STREAM_STATE_BYTES]);

/* ---------------------------------------------------------------- whole messages, host pointers, pipelined
 * For data that does not start on the GPU: the message is cut into chunk_bytes
 * pieces (0 = 64 MiB); H2D of chunk k+1, the fused kernel on chunk k and D2H of chunk k-1 overlap on three
 * HIP streams, and the running GHASH value is carried between chunks on the device (state the RTL and the
 * pycryptodome model cannot export).  Results are bit-identical to aesgcm_encrypt/aesgcm_decrypt.  Buffers from
 * aesgcm_host_alloc (page-locked) make the copies true DMA; pageable buffers work but copy slower.  The chunk-to-
 * chunk GHASH value uses the context's streaming slot: inside an open aesgcm_stream_begin .. aesgcm_stream_final
 * session these calls return AESGCM_ESTATE. */
AESGCM_API int aesgcm_encrypt_pipelined(aesgcm_ctx *ctx, const uint8_t iv[12], const uint8_t *aad, size_t aad_len,
                             const uint8_t *pt, size_t len, uint8_t *ct, uint8_t tag[16], size_t chunk_bytes);
AESGCM_API int aesgcm_decrypt_pipelined(aesgcm_ctx *ctx, const uint8_t iv[12], const uint8_t *aad, size_t aad_len,
                             const uint8_t *ct, size_t len, uint8_t *pt, const uint8_t *expect_tag, uint8_t tag_out[16],
                             size_t chunk_bytes);
AESGCM_API int aesgcm_host_alloc(void **h_ptr, size_t bytes);
AESGCM_API int aesgcm_host_free(void *h_ptr);

/* ---------------------------------------------------------------- device memory helpers
 * (so that a Python/ctypes host needs no other GPU runtime binding) */
AESGCM_API int aesgcm_dev_alloc(int device, void **d_ptr, size_t bytes);
AESGCM_API int aesgcm_dev_free(int device, void *d_ptr);
AESGCM_API int aesgcm_dev_upload(int device, void *d_dst, const void *h_src, size_t bytes);
AESGCM_API int aesgcm_dev_download(int device, void *h_dst, const void *d_src, size_t bytes);
AESGCM_API int aesgcm_dev_sync(int device);
/* device-to-device copy by a plain 16-bytes-per-lane kernel (asynchronous on `stream`): the measured HBM
 * read+write figure bench.py reports beside the datasheet peak (SURVEY.md 8(d) "measured copy-kernel figure"). */
AESGCM_API int aesgcm_dev_copy(int device, void *d_dst, const void *d_src, size_t bytes, void *stream);
/* SplitMix64 counter-based synthetic stream: little-endian 64-bit word w of stream
 * `seed` for w = first_word ..; bytes [0, len) of the buffer. */
AESGCM_API int aesgcm_fill_splitmix64_dev(int device, void *d_buf, size_t len, uint64_t seed, uint64_t first_word, void *stream);

/* A pair of HIP events: aesgcm_timer_start / _stop record them on `stream` (NULL = the default stream) around whatever the
 * caller enqueues there; aesgcm_timer_ms waits for the second and returns the elapsed device time.  For timing launches
 * of the context-free entry points (bench.py --config cfg5: aesgcm_batch_crypt_dev) on the stream they run on. */
typedef struct aesgcm_timer aesgcm_timer;
AESGCM_API int aesgcm_timer_create(aesgcm_timer **out, int device);
AESGCM_API int aesgcm_timer_start(aesgcm_timer *t, void *stream);
AESGCM_API int aesgcm_timer_stop(aesgcm_timer *t, void *stream);
AESGCM_API int aesgcm_timer_ms(aesgcm_timer *t, double *ms);
AESGCM_API int aesgcm_timer_destroy(aesgcm_timer *t);

/* ---------------------------------------------------------------- measurement support
 * When enabled, every launch of the fused CTR+GHASH kernel on this context is bracketed with HIP
 * events on the stream it is launched on (see aesgcm_ctx_split for split ranges).  aesgcm_ctx_timing_read synchronises those events and
 * returns the number of launches and their summed duration since the last reset. */
AESGCM_API int aesgcm_ctx_timing_enable(aesgcm_ctx *ctx, int on);
AESGCM_API int aesgcm_ctx_timing_read(aesgcm_ctx *ctx, uint64_t *n_launches, double *total_ms, int reset);
/* Per-workgroup trace of the most recent fused-kernel launch made while timing was enabled: for each of
 * the *n_wgs workgroups four uint64 {start, end of its last wave (100 MHz wall clock), HW_ID | XCC_ID << 32,
 * (chunks its waves processed) | (sum over its waves of shader-clock kilocycles resident) << 32}. */
AESGCM_API int aesgcm_ctx_wg_trace(aesgcm_ctx *ctx, uint64_t *out, size_t max_wgs, size_t *n_wgs);
/* Ceiling of the formulation (SURVEY.md 8(d) "measured LDS/VALU ceilings next to the result"): runs the fused kernel's
 * instruction stream over a virtual range of `nbytes` with its global loads and stores removed (no buffer is touched)
 * and returns its duration and the number of 16-byte blocks it covered.  bench.py prints 32 B x blocks / time beside
 * the achieved figure, measured in the same process. */
AESGCM_API int aesgcm_ctx_ceiling_probe(aesgcm_ctx *ctx, size_t nbytes, double *ms, uint64_t *blocks);
/* geometry the context chose (workgroups, lanes per workgroup, LDS bytes per workgroup) */
AESGCM_API int aesgcm_ctx_geometry(const aesgcm_ctx *ctx, int *n_workgroups, int *wg_lanes, int *lds_bytes);
/* the same for k_body, the kernel that runs the aligned middle of ranges >= 256 MiB (four T-tables: one 1024-lane workgroup per CU) */
AESGCM_API int aesgcm_ctx_body_geometry(const aesgcm_ctx *ctx, int *n_workgroups, int *wg_lanes, int *lds_bytes);
/* How a data range of `len` bytes starting at block `first_block` of its message is launched: large ranges are cut
 * into head (k_main), an aligned middle whose counters start at a multiple of 256 (k_body: rounds 1-2 without LDS
 * lookups) and tail (k_main).  *body_blocks = 0 means one k_main launch.  With timing enabled, only the k_body
 * launch of a split range is timed and traced (it is the measured kernel). */
AESGCM_API int aesgcm_ctx_split(const aesgcm_ctx *ctx, size_t len, uint64_t first_block, uint64_t *head_blocks, uint64_t *body_blocks);

#ifdef __cplusplus
}
#endif
#endif /* AESGCM_H */
