#!/usr/bin/env python3
"""Condense a rocprofv3 output tree (profiles/collect.sh) into a small text/JSON summary that is committed
under profiles/: per-kernel stats and per-launch PMC sums for the fused kernel."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict


def rows(path):
    with open(path, newline="") as f:
        return list(csv.DictReader(f))


def main(out):
    summ = {}
    for p in glob.glob(os.path.join(out, "stats", "**", "*kernel_stats.csv"), recursive=True):
        print("== kernel stats", os.path.relpath(p, out))
        for r in rows(p):
            print("  %-70s calls %6s  avg %12s ns  total %14s ns  %6s%%" % (r.get("Name", "")[:70], r.get("Calls"), r.get("AverageNs"), r.get("TotalDurationNs"), r.get("Percentage")))
            # the fused kernel of the run: k_body for split ranges, else k_main -- whichever took most of the time
            if ("k_main" in r.get("Name", "") or "k_body" in r.get("Name", "")) and float(r["TotalDurationNs"]) > summ.get("_hot_total", 0.0):
                summ["_hot_total"] = float(r["TotalDurationNs"])
                summ["hot_kernel"] = r["Name"].split("(")[0].replace("void ", "")
                summ["hot_avg_ns"] = float(r["AverageNs"])
                summ["hot_calls"] = int(r["Calls"])
    for d in sorted(glob.glob(os.path.join(out, "pmc_*"))):
        if not os.path.isdir(d):
            continue
        for p in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            acc = defaultdict(lambda: defaultdict(float))
            cnt = defaultdict(int)
            info = {}
            for r in rows(p):
                k = r.get("Kernel_Name", "")
                acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
                info[k] = (r.get("VGPR_Count"), r.get("SGPR_Count"), r.get("LDS_Block_Size"), r.get("Workgroup_Size"), r.get("Grid_Size"))
            # dispatches per kernel = rows / counters
            disp = defaultdict(set)
            for r in rows(p):
                disp[r.get("Kernel_Name", "")].add(r.get("Dispatch_Id"))
            print("== pmc", os.path.relpath(p, out))
            hot = summ.get("hot_kernel", "k_main")
            for k in acc:
                if hot not in k:
                    continue
                n = len(disp[k])
                print("  %s  dispatches %d  vgpr/sgpr/lds/wg/grid %s" % (k[:60], n, info[k]))
                for c, v in sorted(acc[k].items()):
                    print("    %-28s per-launch %.6g" % (c, v / n))
                    summ.setdefault("pmc", {})[c] = v / n
    pm = summ.get("pmc", {})
    if "FETCH_SIZE" in pm or "WRITE_SIZE" in pm:
        # FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE reads exactly 1/2 of a wide coalesced
        # streaming read (MI355X_MICROARCH.md "HBM") -> doubled.  WRITE_SIZE is uncalibrated there: reported raw.
        f = pm.get("FETCH_SIZE", 0.0) * 1024 * 2
        w = pm.get("WRITE_SIZE", 0.0) * 1024
        summ["hbm_read_bytes_per_launch_corrected"] = f
        summ["hbm_write_bytes_per_launch_raw"] = w
        summ["hbm_bytes_per_launch"] = f + w
    print(json.dumps(summ, indent=1))
    with open(os.path.join(out, "summary.json"), "w") as fo:
        json.dump(summ, fo, indent=1)


if __name__ == "__main__":
    main(sys.argv[1])
