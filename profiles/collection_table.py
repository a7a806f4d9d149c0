#!/usr/bin/env python3
"""The markdown table of a final collection (profiles/README.md): python3 profiles/collection_table.py r04"""
import json, os, sys
R = sys.argv[1] if len(sys.argv) > 1 else "r04"
D = os.path.join(os.path.dirname(os.path.abspath(__file__)), R)
ALG = {"cfg3_n1": 2 * 2**34 + 16, "cfg3_dec": 2 * 2**34 + 16, "cfg2_n1": 2 * 2**30 + 16, "cfg5_n1": 2**20 * (8192 + 44), "cfg5_aes256": 2**20 * (8192 + 60), "cfg5_dec": 2**20 * (8192 + 44),
       "cyc_64m": 2 * 2**26 + 16, "cyc_1m": 2 * 2**20 + 16, "half_16m": 2 * 2**24 + 16, "pktl_1k": 2**20 * (2048 + 28), "pktl_1k_dec": 2**20 * (2048 + 28), "pktg_1k": 2**20 * (2048 + 28),
       "pktg8_1k": 2**20 * (2048 + 28), "pktg4_1k": 2**20 * (2048 + 28), "pktw_16k": 4096 * (32768 + 28),
       "pktw_1m": 4096 * (2 * 2**20 + 28), "pktl_ilp_1k": 131072 * (2048 + 28), "batchw_1m": 4096 * (2 * 2**20 + 60)}
NOTE = {"cyc_1m": "; the key's tables and the closing's, once per XCD: 1.1 MB beside 2.1 MB of data", "pktg4_1k": "; a 4-lane group touches its 128-byte line in two 64-byte steps",
        "half_16m": "; three launches overlap", "pktw_1m": "; 4096 x 1 MiB packets under one key (call 69)",
        "pktl_ilp_1k": "; 131072 x 1 KiB, the ILP form (call 69)", "batchw_1m": "; 4096 x 1 MiB packets with an AES-256 key each (call 69)"}
print("| tag | kernel | avg ms (rocprof) | HBM bytes vs alg | LDS array busy | bank conflicts / array cycles | frac (under the profiler) |")
print("|---|---|---|---|---|---|---|")
for t, alg in ALG.items():
    p = os.path.join(D, t, "pmc_%s.json" % t)
    if not os.path.exists(p):
        continue
    j = json.load(open(p)); lds = j.get("lds", {})
    ms = j["kernel_avg_ns_under_rocprof"] / 1e6
    conf, act = lds.get("SQ_LDS_BANK_CONFLICT_per_launch"), lds.get("SQ_LDS_IDX_ACTIVE_per_launch")
    print("| %s | `%s` | %.4g | %.4g (%.3f x%s) | %s | %s | %.3f |" % (t, j["kernel"], ms, j["hbm_bytes_per_launch"], j["hbm_bytes_per_launch"] / alg, NOTE.get(t, ""), lds.get("lds_busy_frac"),
          "%.1f %%" % (100 * conf / act) if conf is not None and act else "--", alg / (ms * 1e-3) / 8e12))
for f in ("default", "dec", "cfg2", "cfg5", "cfg5_dec", "cfg5_aes256", "emu_r0", "emu_r7"):
    j = json.load(open(os.path.join(D, "bench_%s.json" % f))); r = j["roofline"]
    print("bench_%s: %.1f GiB/s, step %.3f ms, frac %.4f, sclk %s" % (f, j["value"], j["ms_per_step"], r["frac"], r.get("sclk_mhz")))
