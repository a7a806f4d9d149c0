#!/usr/bin/env python3
"""ISA census of every kernel instance in csrc/aesgcm_kernels.gfx950.s (`make -C csrc asm`): VGPRs, SGPRs, scratch bytes,
static LDS, and -- per loop depth, from the assembler's own loop comments -- the instruction mix with the scratch_* and
v_readlane/v_writelane counts, so that a spill that lands inside a hot loop shows up as a number and not as a surprise in a
counter run.  `hot` = basic blocks that hold at least HOT_LDS ds_read ops (an AES row or a GHASH table multiply).

    python tools/isa_census.py [path.s]            -> table on stdout (profiles/archive/r03/isa_census.txt is this output)
    census(path) -> {kernel: {...}} for tests/test_isa_cpu.py
"""
import os
import re
import subprocess
import sys
from collections import Counter, defaultdict

HOT_LDS = 40
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DEFAULT_S = os.path.join(ROOT, "aes-gcm-128-192-256-bits_amd", "csrc", "aesgcm_kernels.gfx950.s")


def demangle(names):
    try:
        out = subprocess.run(["c++filt"] + names, stdout=subprocess.PIPE, text=True, check=True).stdout.split("\n")
        return {n: re.sub(r"\(.*", "", re.sub(r"^void ", "", d)) for n, d in zip(names, out)}
    except Exception:
        return {n: n for n in names}


def wt_store_hazards(path=None, need=5):
    """The through-the-L2 stores (gstore16_wt / gstore*_wt_at in aesgcm_dev.h) are inline asm, so the compiler's hazard recognizer does not see that they read
    an SGPR pair as their base: `VALU writes SGPR -> VMEM reads that SGPR` needs 5 wait states (gfx940 and later), and only instruction distance provides
    them.  -> [(line number, store, writer, wait states)] for every `global_store ... sc0 sc1` whose base SGPR is written by a VALU instruction
    (v_readfirstlane / v_readlane / v_cmp into an SGPR) fewer than `need` wait states earlier in the same basic block.  tests/test_isa_cpu.py wants []."""
    lines = open(path or DEFAULT_S).read().split("\n")
    bad = []
    for i, ln in enumerate(lines):
        m = re.match(r"\s*global_store_\w+\s+v\d+, v(?:\[[\d:]+\]|\d+), s\[(\d+):(\d+)\].*sc0 sc1", ln)
        if not m:
            continue
        regs = {"s%s" % m.group(1), "s%s" % m.group(2)}
        waits, j = 0, i - 1
        while j >= 0 and waits < need:
            t = lines[j].strip()
            j -= 1
            if not t or t.startswith((";", "//")) or t.startswith(".") and not t.endswith(":"):
                continue
            if t.endswith(":") or re.match(r"^\.?\w+:", t):          # a label: the block starts here, whatever came before is a branch away
                break
            op = t.split()[0]
            dst = re.match(r"\S+\s+(s\d+|s\[(\d+):(\d+)\])", t)
            if op.startswith("v_") and dst:
                d = {dst.group(1)} if dst.group(2) is None else {"s%d" % k for k in range(int(dst.group(2)), int(dst.group(3)) + 1)}
                if d & regs:
                    bad.append((i + 1, ln.strip(), t, waits))
                    break
            nop = re.match(r"s_nop\s+(\d+)", t)
            waits += int(nop.group(1)) + 1 if nop else 1
    return bad


def classify(op):
    if op.startswith("scratch_"):
        return "scratch"
    if op.startswith(("v_readlane", "v_writelane")):
        return "lane"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith(("global_", "flat_", "buffer_")):
        return "vmem"
    if op.startswith("v_"):
        return "valu"
    if op.startswith(("s_load", "s_buffer_load")):
        return "smem"
    if op.startswith("s_waitcnt"):
        return "wait"
    if op.startswith("s_"):
        return "salu"
    return "other"


def census(path=DEFAULT_S):
    kernels = {}
    meta = {}
    cur = None
    lines = open(path).read().split("\n")
    # ---- code: functions start at "<name>:" after a .type <name>,@function; blocks at .LBBn_m:
    funcs = set(re.findall(r"^\s*\.type\s+(\S+),@function", "\n".join(lines), flags=re.M))
    i = 0
    while i < len(lines):
        line = lines[i]
        m = re.match(r"^(\S+):\s*(;.*)?$", line)
        if m and m.group(1) in funcs:
            cur = kernels.setdefault(m.group(1), {"blocks": []})
            blk = {"label": "entry", "depth": 0, "ops": Counter()}
            cur["blocks"].append(blk)
            i += 1
            continue
        if cur is not None:
            s = line.strip()
            if s.startswith(".Lfunc_end"):
                cur = None
                i += 1
                continue
            m = re.match(r"^(\.LBB\d+_\d+):(.*)$", line)
            if m:
                # loop comments ride on the label line and on the comment-only lines right after it
                text = m.group(2)
                j = i + 1
                while j < len(lines) and lines[j].strip().startswith(";"):
                    text += " " + lines[j]
                    j += 1
                d = 0
                mm = re.search(r"This (?:Inner )?Loop Header: Depth=(\d+)", text) or re.search(r"in Loop: Header=\S+ Depth=(\d+)", text)
                if mm:
                    d = int(mm.group(1))
                blk = {"label": m.group(1), "depth": d, "ops": Counter()}
                cur["blocks"].append(blk)
                i += 1
                continue
            if s and not s.startswith((";", ".")):
                blk["ops"][classify(s.split()[0])] += 1
                if s.split()[0].startswith("ds_read"):
                    blk["ops"]["ds_read"] += 1
        i += 1
    # ---- metadata (amdhsa.kernels YAML at the end of the file)
    text = "\n".join(lines)
    for m in re.finditer(r"- \.agpr_count:.*?(?=\n  - \.agpr_count:|\namdhsa\.target|\Z)", text, flags=re.S):
        blob = m.group(0)
        name = re.search(r"\.name:\s+(\S+)", blob)
        if not name:
            continue
        def num(key):
            mm = re.search(r"\.%s:\s+(\d+)" % key, blob)
            return int(mm.group(1)) if mm else None
        meta[name.group(1)] = dict(vgpr=num("vgpr_count"), sgpr=num("sgpr_count"), scratch=num("private_segment_fixed_size"),
                                   lds_static=num("group_segment_fixed_size"), spill_v=num("vgpr_spill_count"), spill_s=num("sgpr_spill_count"))
    names = demangle(sorted(kernels))
    out = {}
    for k, v in kernels.items():
        if k not in meta:
            continue
        by_depth = defaultdict(Counter)
        hot = Counter()
        for b in v["blocks"]:
            by_depth[b["depth"]].update(b["ops"])
            if b["ops"]["ds_read"] >= HOT_LDS:
                hot.update(b["ops"])
                hot["blocks"] += 1
        out[names[k]] = dict(meta[k], depth={d: dict(c) for d, c in sorted(by_depth.items())}, hot=dict(hot), mangled=k)
    return out


def main():
    path = sys.argv[1] if len(sys.argv) > 1 else DEFAULT_S
    c = census(path)
    print("# ISA census of %s  (tools/isa_census.py; hot = blocks with >= %d ds_read ops)" % (os.path.basename(path), HOT_LDS))
    print("# %-34s %5s %5s %8s | scratch_* ops by loop depth | v_read/writelane by depth | hot blocks: n, valu, lds, vmem, scratch, lane" % ("kernel", "vgpr", "sgpr", "scratchB"))
    for name in sorted(c):
        k = c[name]
        sc = " ".join("d%d=%d" % (d, v.get("scratch", 0)) for d, v in k["depth"].items())
        ln = " ".join("d%d=%d" % (d, v.get("lane", 0)) for d, v in k["depth"].items())
        h = k["hot"]
        print("%-36s %5s %5s %8s | %-28s | %-28s | n=%d valu=%d lds=%d vmem=%d scratch=%d lane=%d" % (
            name, k["vgpr"], k["sgpr"], k["scratch"], sc, ln, h.get("blocks", 0), h.get("valu", 0), h.get("lds", 0), h.get("vmem", 0), h.get("scratch", 0), h.get("lane", 0)))


if __name__ == "__main__":
    main()
