#!/usr/bin/env python3
"""Condense a rocprofv3 output tree (profiles/collect.sh) into a small text/JSON summary that is committed under
profiles/: per-kernel stats and per-launch PMC sums for the hot kernel, plus pmc_<tag>.json -- the file bench.py reads
for `roofline.traffic`, stamped with the git hash and the SHA-256 of libaesgcm_hip.so it was measured on (bench.py
reports the traffic only when the library that is running is that very build).
    python3 profiles/summarize.py <out dir> <tag> <hot kernel regex>"""
import csv
import glob
import hashlib
import json
import os
import re
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def rows(path):
    with open(path, newline="") as f:
        return list(csv.DictReader(f))


def sha256_file(path):
    h = hashlib.sha256()
    with open(path, "rb") as f:
        for b in iter(lambda: f.read(1 << 20), b""):
            h.update(b)
    return h.hexdigest()


def main(out, tag, hot_re):
    summ = {"tag": tag}
    hot_re = re.compile(hot_re)
    for p in glob.glob(os.path.join(out, "stats", "**", "*kernel_stats.csv"), recursive=True):
        print("== kernel stats", os.path.relpath(p, out))
        for r in rows(p):
            print("  %-70s calls %6s  avg %12s ns  total %14s ns  %6s%%" % (r.get("Name", "")[:70], r.get("Calls"), r.get("AverageNs"), r.get("TotalDurationNs"), r.get("Percentage")))
            if hot_re.search(r.get("Name", "")) and float(r["TotalDurationNs"]) > summ.get("_hot_total", 0.0):
                summ["_hot_total"] = float(r["TotalDurationNs"])
                summ["hot_kernel"] = r["Name"].split("(")[0].replace("void ", "")
                summ["hot_avg_ns"] = float(r["AverageNs"])
                summ["hot_calls"] = int(r["Calls"])
    for d in sorted(glob.glob(os.path.join(out, "pmc_*"))):
        if not os.path.isdir(d):
            continue
        for p in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            acc = defaultdict(lambda: defaultdict(float))
            info, disp = {}, defaultdict(set)
            for r in rows(p):
                k = r.get("Kernel_Name", "")
                acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
                info[k] = (r.get("VGPR_Count"), r.get("SGPR_Count"), r.get("LDS_Block_Size"), r.get("Workgroup_Size"), r.get("Grid_Size"))
                disp[k].add(r.get("Dispatch_Id"))
            print("== pmc", os.path.relpath(p, out))
            hot = summ.get("hot_kernel", "k_main")
            for k in acc:
                if hot not in k:
                    continue
                n = len(disp[k])
                print("  %s  dispatches %d  vgpr/sgpr/lds/wg/grid %s" % (k[:60], n, info[k]))
                summ["launch"] = dict(zip(("vgpr", "sgpr", "lds_block", "workgroup", "grid"), info[k]))
                for c, v in sorted(acc[k].items()):
                    print("    %-28s per-launch %.6g" % (c, v / n))
                    summ.setdefault("pmc", {})[c] = v / n
    pm = summ.get("pmc", {})
    pj = {"source": "profiles/%s (rocprofv3 --pmc, one counter group per pass, never with tracing)" % os.path.basename(out),
          "tag": tag, "kernel": summ.get("hot_kernel"), "kernel_avg_ns_under_rocprof": summ.get("hot_avg_ns")}
    so = os.environ.get("AESGCM_LIB") or os.path.join(ROOT, "aes-gcm-128-192-256-bits_amd", "libaesgcm_hip.so")
    try:
        pj["so_sha256"] = sha256_file(so)
    except OSError:
        pj["so_sha256"] = None
    pj["git"] = (os.environ.get("GIT_HEAD") or "")[:7] or None        # the GPU box has no .git: the caller passes the hash
    if "FETCH_SIZE" in pm or "WRITE_SIZE" in pm:
        # FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE reads exactly 1/2 of a wide coalesced streaming read
        # (MI355X_MICROARCH.md "HBM"; calibrated here too, microbench/fetch_calib.hip) -> doubled.  WRITE_SIZE calibrated exact.
        f = pm.get("FETCH_SIZE", 0.0) * 1024 * 2
        w = pm.get("WRITE_SIZE", 0.0) * 1024
        pj.update({"FETCH_SIZE_KiB": pm.get("FETCH_SIZE"), "WRITE_SIZE_KiB": pm.get("WRITE_SIZE"),
                   "correction": "FETCH_SIZE x2 (gfx950 tallies 128-byte read requests at 64 B); WRITE_SIZE raw",
                   "hbm_read_bytes_per_launch": f, "hbm_write_bytes_per_launch": w, "hbm_bytes_per_launch": f + w})
    if "TCC_EA0_RDREQ_sum" in pm:
        # read requests at the L2's memory side by size; what is not 32 / 64 / 128 B is taken as 64 B
        n, n32, n64, n128 = pm["TCC_EA0_RDREQ_sum"], pm.get("TCC_EA0_RDREQ_32B_sum", 0.0), pm.get("TCC_EA0_RDREQ_64B_sum", 0.0), pm.get("TCC_EA0_RDREQ_128B_sum", 0.0)
        pj["tcc_read"] = {"requests": n, "32B": n32, "64B": n64, "128B": n128, "bytes_by_request_size": 32 * n32 + 64 * n64 + 128 * n128 + 64 * max(0.0, n - n32 - n64 - n128)}
    if "TCC_EA0_WRREQ_sum" in pm:
        n, n64 = pm["TCC_EA0_WRREQ_sum"], pm.get("TCC_EA0_WRREQ_64B_sum", 0.0)
        pj["tcc_write"] = {"requests": n, "64B": n64, "bytes_64B_full_else_32B": 64 * n64 + 32 * max(0.0, n - n64),
                           "l2_hit_frac": round(pm["TCC_HIT_sum"] / (pm["TCC_HIT_sum"] + pm["TCC_MISS_sum"]), 4) if pm.get("TCC_HIT_sum") is not None and (pm.get("TCC_HIT_sum", 0) + pm.get("TCC_MISS_sum", 0)) else None}
    if "SQ_LDS_IDX_ACTIVE" in pm and "GRBM_GUI_ACTIVE" in pm:
        cu = 256
        pj["lds"] = {"SQ_LDS_IDX_ACTIVE_per_launch": pm["SQ_LDS_IDX_ACTIVE"], "SQ_INSTS_LDS_per_launch": pm.get("SQ_INSTS_LDS"),
                     "SQ_LDS_BANK_CONFLICT_per_launch": pm.get("SQ_LDS_BANK_CONFLICT"),
                     "GRBM_GUI_ACTIVE_per_launch_sum_of_8_XCDs": pm["GRBM_GUI_ACTIVE"], "cu_count": cu,
                     "lds_busy_frac": round(pm["SQ_LDS_IDX_ACTIVE"] / (cu * pm["GRBM_GUI_ACTIVE"] / 8.0), 4),
                     "lds_array_cycles_per_lds_instruction": round(pm["SQ_LDS_IDX_ACTIVE"] / pm["SQ_INSTS_LDS"], 3) if pm.get("SQ_INSTS_LDS") else None}
    summ["pmc_json"] = pj
    print(json.dumps(summ, indent=1))
    with open(os.path.join(out, "summary.json"), "w") as fo:
        json.dump(summ, fo, indent=1)
    with open(os.path.join(out, "pmc_%s.json" % tag), "w") as fo:
        json.dump(pj, fo, indent=1)


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else "run", sys.argv[3] if len(sys.argv) > 3 else "k_main|k_body")
