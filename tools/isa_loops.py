#!/usr/bin/env python3
"""Instruction census of one kernel in a gfx950 .s file: per basic block (label to label) counts of VALU / SALU /
LDS / VMEM / scratch / readlane-writelane instructions, so that the hot loop's mix can be read off.
    python tools/isa_loops.py aesgcm_kernels.gfx950.s _Z6k_bodyILi14ELi0EE"""
import re
import sys
from collections import Counter


def census(path, prefix):
    blocks, cur, name, on = [], Counter(), None, False
    for line in open(path):
        if not on:
            if line.startswith(prefix) and ":" in line:
                on, name = True, "entry"
                cur = Counter()
            continue
        s = line.strip()
        if s.startswith(".Lfunc_end") or s.startswith(".section") or s.startswith(".rodata"):
            break
        m = re.match(r"^(\.LBB\d+_\d+):", s)
        if m:
            blocks.append((name, cur))
            name, cur = m.group(1), Counter()
            continue
        if not s or s.startswith(";") or s.startswith("."):
            continue
        op = s.split()[0]
        if op.startswith("scratch_"):
            cur["scratch"] += 1
        elif op.startswith(("v_readlane", "v_writelane", "v_readfirstlane")):
            cur["lane"] += 1
        elif op.startswith("ds_"):
            cur["lds:" + op] += 1
            cur["lds"] += 1
        elif op.startswith(("global_", "flat_", "buffer_")):
            cur["vmem"] += 1
        elif op.startswith("v_"):
            cur["valu"] += 1
            if op.startswith(("v_accvgpr",)):
                cur["accvgpr"] += 1
        elif op.startswith("s_load") or op.startswith("s_buffer_load"):
            cur["smem"] += 1
        elif op.startswith("s_waitcnt"):
            cur["wait"] += 1
        elif op.startswith("s_"):
            cur["salu"] += 1
        cur["total"] += 1
    blocks.append((name, cur))
    return blocks


if __name__ == "__main__":
    for name, c in census(sys.argv[1], sys.argv[2]):
        if c["total"] >= int(sys.argv[3]) if len(sys.argv) > 3 else 1:
            keys = ["total", "valu", "salu", "lds", "vmem", "smem", "scratch", "lane", "wait", "accvgpr"]
            print("%-12s" % name, " ".join("%s=%d" % (k, c[k]) for k in keys if c[k]))
