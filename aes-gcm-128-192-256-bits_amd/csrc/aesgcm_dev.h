// aesgcm_dev.h -- per-lane arithmetic of the MI355X AES-GCM path (gfx950).
//
// Everything here is __host__ __device__ so that the exact lane code the kernels run can also be
// driven by a CPU harness (tests/host_emul) in the GPU-less build container.  The kernels that
// compose these pieces are in aesgcm_kernels.hip.
//
// Data conventions (DESIGN.md "Layout"):
//   * a 16-byte block lives in registers as 4 dwords in MEMORY order ("mo"): d0 = bytes 0..3 loaded
//     little-endian, exactly what global_load_dwordx4 returns.  AES state columns are therefore
//     little-endian words (row 0 in the low byte); the T-table is built for that convention, so no
//     byte swap is ever needed on the data path.
//   * GF(2^128) arithmetic that needs shifts (the bit-serial multiply) works on big-endian words
//     ("be"): w[0] holds GCM bits 0..31 with bit 0 in the MSB (src/ghash_gfmul.vhd:44-57: VHDL bit
//     127 = leftmost).  mo <-> be is one byte swap per word.
#pragma once
// Since round 5 this header is an umbrella; the code lives in
//   aesgcm_base.h     types, global-memory accessors, the literal AES / GF(2^128) arithmetic, LDS access, SplitMix64
//   aesgcm_aes.h      AES rounds through T-tables in LDS
//   aesgcm_ghash.h    GHASH multiplies by a launch constant through tables in LDS
//   aesgcm_stream.h   structures, lane code and host-side plans of the stream kernels (k_setup, k_main, k_fold, k_body, k_combine)
//   aesgcm_batch.h    packets with a key each (k_batch3)
//   aesgcm_pkt.h      packets under one key (k_pktg, k_pktl)
// and, on top of them, aesgcm_rows.h (many messages under one key through k_body's row code: k_rows).
#include "aesgcm_base.h"
#include "aesgcm_aes.h"
#include "aesgcm_ghash.h"
#include "aesgcm_stream.h"
#include "aesgcm_batch.h"
#include "aesgcm_pkt.h"
