"""CPU: ISA regression guard.  Builds the gfx950 assembly of the shipped kernels (`make -C csrc asm`, hipcc cross-compiles without
a GPU) and reads it with tools/isa_census.py: a register spill that lands inside a hot loop costs HBM traffic and issue slots
(round 1: three scratch_load per row in k_body = 13 % extra traffic; round 2: nine scratch ops in k_pkt's row loop), and it
arrives silently with any change of a launch bound or of the lane code.  Fails if

  * k_body (every key size, ENC / DEC / PROBE, dealt chunks and cyclic rows with their fused closing), k_pktl, k_pktg (every shape), their twins for messages wherever
    they live (k_pktls, k_pktgs: round 6), k_batch3 (every shape) or the KS / ECB instances of k_main use scratch at all;
  * any scratch_* op sits at the innermost loop depth of k_main ENC / DEC (the row loop), of k_pktg (the iteration loop of a
    packet's lane group), of k_batch3 (the block loops);
  * a kernel needs more registers than its launch geometry allows;
  * a retired kernel is back in the library (k_batch2 and k_batch, deleted in round 4), or the kernel set is not the one DESIGN.md lists;
  * a through-the-L2 store (inline asm: the compiler's hazard recognizer does not see it) reads a base SGPR that a VALU instruction wrote fewer than 5 wait
    states earlier (tools/isa_census.py wt_store_hazards).

The full table is committed as profiles/r04/isa_census.txt."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
CSRC = os.path.join(ROOT, "aes-gcm-128-192-256-bits_amd", "csrc")


@pytest.fixture(scope="module")
def census():
    if not os.path.exists("/opt/rocm/bin/hipcc"):
        pytest.skip("no hipcc")
    subprocess.run(["make", "-C", CSRC, "-s", "asm"], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    import isa_census
    c = isa_census.census(os.path.join(CSRC, "aesgcm_kernels.gfx950.s"))
    assert len(c) > 60
    return c


def _inner_scratch(k):
    """scratch ops at the deepest loop depth that holds LDS reads"""
    depths = [d for d, ops in k["depth"].items() if ops.get("ds_read", 0) >= 16]
    assert depths, k
    return k["depth"][max(depths)].get("scratch", 0), max(depths)


def test_scratch_free_kernels(census):
    for name, k in census.items():
        if name.startswith(("k_body<", "k_pktl<", "k_pktg<", "k_pktls<", "k_pktgs<", "k_batch3<", "k_batch<")) or name.startswith("k_main<") and name.endswith((", 2>", ", 3>")):
            assert k["scratch"] == 0, (name, k["scratch"])


def test_no_scratch_in_the_hot_loops(census):
    seen = 0
    for name, k in census.items():
        if name.startswith(("k_main<", "k_body<", "k_pktg<", "k_pktl<", "k_pktgs<", "k_pktls<", "k_batch<", "k_batch3<")):
            n, depth = _inner_scratch(k)
            assert n == 0, "%s: %d scratch ops at loop depth %d" % (name, n, depth)
            seen += 1
    assert seen == 12 + 15 + 33 + 15 + 18 + 6 + 21      # k_main (3 key sizes x 4 modes), k_body (x ENC, DEC dealt and cyclic + the dealt probe), k_pktg (x 2 x 4 shapes + the probes of 3 shapes), k_pktl (x 2 forms + the probe), k_pktgs (x 2 x 3 shapes), k_pktls, k_batch3 (x 3 shapes)


def test_register_budgets(census):
    for name, k in census.items():
        if name.startswith("k_main<"):
            assert k["vgpr"] <= 80, (name, k["vgpr"])          # 768 lanes x 2 workgroups per CU = 6 waves per SIMD
        wide = name.startswith("k_pktg<") and name.endswith(", 6>") or name.startswith("k_pktl<") and name.endswith(", 0>") or name.startswith("k_pktls<")      # 768-lane workgroups: 3 waves per SIMD, 168 registers
        if wide:
            assert k["vgpr"] <= 168 and k["scratch"] == 0, (name, k["vgpr"], k["scratch"])
        elif name.startswith("k_pktl<"):                       # the ILP form: 512-lane workgroups, 2 waves per SIMD, 256 registers
            assert k["vgpr"] <= 256 and k["scratch"] == 0, (name, k["vgpr"], k["scratch"])
        elif name.startswith(("k_body<", "k_bodyh<", "k_pktg<", "k_pktgs<", "k_batch<", "k_batch3<")):
            assert k["vgpr"] <= 128, (name, k["vgpr"])         # one 1024-lane workgroup per CU = 4 waves per SIMD
            if name.startswith(("k_pktg<", "k_pktgs<", "k_batch3<")):
                assert k["scratch"] == 0, (name, k["scratch"])  # nothing spilled (ds_swizzle exchanges, per-packet values parked in LDS, fresh lane id)


def test_kernel_set(census):
    """the shipped kernels, by family; nothing retired, nothing unlisted"""
    fam = {}
    for name in census:
        fam.setdefault(name.split("<")[0], []).append(name)
    assert sorted(fam) == ["k_batch3", "k_body", "k_bodyh", "k_combine", "k_combine_batch", "k_copy16", "k_fill_splitmix64", "k_fold", "k_gfmul", "k_init_tables",
                           "k_len_hist", "k_len_scan", "k_len_scatter", "k_len_sort1", "k_main", "k_pktg", "k_pktgs", "k_pktl", "k_pktls", "k_rows", "k_rows_close", "k_rows_plan", "k_rows_plan_base", "k_rows_plan_cut", "k_rows_plan_place", "k_rows_plan_slots", "k_rows_plan_sums", "k_setup", "k_setup_ptab", "k_wipe_failed"], sorted(fam)
    assert (len(fam["k_main"]), len(fam["k_body"]), len(fam["k_bodyh"]), len(fam["k_pktg"]), len(fam["k_pktl"]), len(fam["k_batch3"])) == (12, 15, 6, 33, 15, 21)
    assert (len(fam["k_rows"]), len(fam["k_rows_close"]), len(fam["k_pktgs"]), len(fam["k_pktls"])) == (6, 2, 18, 6)


def test_no_sgpr_hazard_in_front_of_the_write_through_stores(census):
    import isa_census
    bad = isa_census.wt_store_hazards(os.path.join(CSRC, "aesgcm_kernels.gfx950.s"))
    assert not bad, bad[:5]


def test_half_shape_keeps_its_spills_out_of_the_row_loop(census):
    """k_bodyh (128 registers, the two-table round) spills three to five dwords around the GENERAL row code -- the front row a strand runs at most once -- and
    nothing in the body row loop: the basic blocks with the (NR - 2) x 16 T-table lookups of rounds 3 .. NR, or with the 52 reads of the GHASH table multiply,
    hold no scratch op."""
    import isa_loops
    path = os.path.join(CSRC, "aesgcm_kernels.gfx950.s")
    for nr in (10, 12, 14):
        for dec in (0, 1):
            k = census["k_bodyh<%d, %d>" % (nr, dec)]
            assert k["scratch"] <= 32 and sum(ops.get("scratch", 0) for ops in k["depth"].values()) <= 6, k
            blocks = isa_loops.census(path, "_Z7k_bodyhILi%dELi%dEE" % (nr, dec))
            rows = [c for _, c in blocks if c["lds"] == 16 * (nr - 2) or c["lds"] == 52]
            assert rows and all(c["scratch"] == 0 for c in rows), (nr, dec, [(c["lds"], c["scratch"]) for c in rows])


def test_row_kernel_keeps_its_spills_out_of_the_row_loop(census):
    """k_rows (round 5: k_body's row loop over runs of rows of many messages, with the tail / AAD code and the piece walk of a whole call around it) may spill a
    few dwords around the general code -- a tail or an AAD is one row per message -- but nothing in the row loop: the basic blocks with the (NR - 2) x 16 T-table
    lookups of rounds 3 .. NR, or with the 52 reads of the GHASH table multiply, hold no scratch access and at most two lane moves (AES-256 with sixteen more
    registers for the row phases spilled five round keys per row; its build computes the phase constants per row and keeps 24 key words in vector registers)."""
    import isa_loops
    path = os.path.join(CSRC, "aesgcm_kernels.gfx950.s")
    for nr in (10, 12, 14):
        for dec in (0, 1):
            k = census["k_rows<%d, %d>" % (nr, dec)]
            assert k["vgpr"] <= 128 and k["scratch"] <= 64, k
            blocks = isa_loops.census(path, "_Z6k_rowsILi%dELi%dEE" % (nr, dec))
            rows = [c for _, c in blocks if c["lds"] == 16 * (nr - 2) or c["lds"] == 52]
            assert len(rows) >= 2 and all(c["scratch"] == 0 and c["lane"] <= 2 for c in rows), (nr, dec, [(c["lds"], c["scratch"], c["lane"]) for c in rows])
