/* aesgcm_debug.h -- exported ONLY by libaesgcm_hip_dbg.so, the -DAESGCM_DEBUG_KNOBS build of the same sources (csrc/Makefile, target `debug`).
 *
 * The product library (libaesgcm_hip.so) picks its kernel shapes itself and has no switch to override them: it reads no environment
 * variable and exports nothing below.  The test-suite and the profiling scripts need every shape on inputs the host's own rule would
 * give to another shape -- to check each against the CPU restatement, and to measure the rule -- and get it from this one process-wide call.
 */
#ifndef AESGCM_DEBUG_H
#define AESGCM_DEBUG_H

#include "aesgcm.h"

#ifdef __cplusplus
extern "C" {
#endif

/* what = "pkt_lanes"   : aesgcm_packets_crypt_dev takes 1 (k_pktl), 4 / 8 / 16 (k_pktg lane groups) or 64 (k_pktg, a wave per packet) lanes per packet
 *        "pkt_deal"    : packets per dispenser fetch of k_pktg (rounded up to a multiple of the packets per wave, at most 64)
 *        "batch_lanes" : aesgcm_batch_crypt[_var]_dev takes 8 / 16 / 64 lanes per packet (k_batch3)
 *        "batch_deal"  : packets per dispenser fetch of k_batch3 (rounded up to a multiple of the packets per wave)
 *        "pkt_ilp"     : k_pktl in its form for batches that do not fill the chip (512-lane workgroups, eight keystream chains per line) always (1) or never (2)
 *        "pkt_rows"    : aesgcm_packets_crypt_dev / aesgcm_messages_crypt_dev by rows (k_rows / k_rows_close, csrc/aesgcm_rows.h) always (1) or never (2) instead of the library's
 *                        rule (fixed-size records: from 8 KiB per packet, from 2 KiB while the packets are at most 16384; offset arrays: per message, on the device)
 *        (a forced "pkt_lanes" also means never by rows; with offset arrays it names the shape the routed call's packet launch takes)
 *        "batch_order" : aesgcm_batch_crypt_var_dev takes its packets by falling length class always (1) or never (2) instead of from 262144 (AES-128) / 98304 packets
 * value 0 = the library's own choice again.  Not thread-safe; set it between launches. */
AESGCM_API int aesgcm_debug_force_shape(const char *what, int value);

#ifdef __cplusplus
}
#endif
#endif
