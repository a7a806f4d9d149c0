#!/usr/bin/env python3
"""Every number the round-6 documents quote, read from profiles/r06/ (after profiles/runs/r06_adopt.sh): python3 profiles/r06_numbers.py"""
import glob, json, os
R = os.path.join(os.path.dirname(os.path.abspath(__file__)), "r06")
print("library", open(os.path.join(R, "so_sha256.txt")).read().split()[0][:12], "|", open(os.path.join(R, "pytest_tail.txt")).read().strip().splitlines()[-7:][0] if os.path.exists(os.path.join(R, "pytest_tail.txt")) else "")
for l in open(os.path.join(R, "pytest_tail.txt")):
    if "passed" in l: print(l.strip())
for f in sorted(glob.glob(os.path.join(R, "bench_*.json")) + glob.glob(os.path.join(R, "*_box", "bench_*.json"))):
    d = json.loads(open(f).read().strip().splitlines()[-1]); r = d["roofline"]; c = d.get("cpu_baseline") or {}; fc = r.get("formulation_ceiling") or {}
    print("%-44s %7.1f GiB/s step %.3f ms frac %.4f traffic %s ceiling %s (probe %s ms %s GiB/s) cpu %s mframes %s" % (os.path.relpath(f, R), d["value"], d["ms_per_step"], r["frac"],
          ("%.3f x" % (r["traffic"] / r["alg_bytes_per_launch"])) if r.get("traffic") else None, r.get("achieved_over_ceiling"), fc.get("avg_launch_ms"), fc.get("gib_per_s"), c.get("value"), d["config"].get("mframes_per_s")))
for l in open(os.path.join(R, "mixed_u.jsonl")):
    d = json.loads(l)
    print("mixed n %7d max %8d aad %2d AES-%d%s%s: mixed %6.1f (%.3f ms) rows %6.1f pkt %6.1f short %.3f + long %.3f ms vs_comb %.3f" % (d["n"], d["max_len"], d["aad"], d["key_bits"], " scattered" if d["scattered"] else "", " dec" if d["decrypt"] else "",
          d["mixed"]["gib_per_s"], d["mixed"]["ms_median"], d["all_rows"]["gib_per_s"], d["all_pkt"]["gib_per_s"], d["short"]["ms_median"], d["long"]["ms_median"], d["vs_combination"]))
for name in ("route_sweep.jsonl", "route_band.jsonl"):
    p = os.path.join(R, name)
    if not os.path.exists(p): continue
    w = 9
    for l in open(p):
        d = json.loads(l); w = min(w, d["lib_vs_best"])
        if name == "route_band.jsonl" or d["lib_vs_best"] < 0.95 or (d["kind"] == "frames" and d["n"] == 1 << 20):
            print(name[:-6], d["kind"], d["n"], "rows %.1f pkt %.1f lib %.1f GiB/s (%.4f / %.4f / %.4f ms) lib/best %.3f" % (d["gib_s_rows"], d["gib_s_pkt"], d["gib_s_lib"], d["ms_rows"], d["ms_pkt"], d["ms_lib"], d["lib_vs_best"]))
    print(name, "worst", w)
print(open(os.path.join(R, "size_sweep.txt")).read().strip().split("\n", 1)[1])
for t in ("cfg3_n1", "cfg2_n1", "cfg5_n1", "rows_1m", "frames", "frames_probe", "pktl_1k", "pktg_1k", "mixed_u", "mixed_u_1m"):
    j = json.load(open(os.path.join(R, t, "pmc_%s.json" % t)))
    print("%-13s %-22s %.4g ms  hbm %.4g  lds %s  so %s" % (t, j["kernel"], j["kernel_avg_ns_under_rocprof"] / 1e6, j["hbm_bytes_per_launch"], j.get("lds", {}).get("lds_busy_frac"), j["so_sha256"][:8]))
if os.path.exists(os.path.join(R, "graph_replay.jsonl")):
    for l in open(os.path.join(R, "graph_replay.jsonl")):
        if l.startswith("{"):
            d = json.loads(l); print("graph n %8d direct %.1f us (host %.1f) replay %.1f us (host %.1f)" % (d["n_frames"], d["direct_us_per_call"], d["direct_host_us_per_call"], d["graph_us_per_call"], d["graph_host_us_per_call"]))
