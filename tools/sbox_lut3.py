#!/usr/bin/env python3
"""Bitsliced AES S-box for gfx950's v_bitop3_b32 (any 3-input boolean function in one VALU instruction).

Starting point: the Boyar-Peralta depth-16 circuit for the AES S-box (128 XOR/XNOR/AND gates, "A depth-16 circuit for
the AES S-box", 2011) -- a published circuit, restated here as data and VERIFIED below against the S-box computed from
its FIPS-197 definition (GF(2^8) inverse + affine map; the reference's table is src/aes_func.vhd:228-301).  The circuit
is then technology-mapped into 3-input LUTs by cut enumeration and area-oriented covering (area flow, then exact local
area with reference counting), which fuses XOR chains into XOR3 and AND-XOR pairs into single instructions.

    python tools/sbox_lut3.py            # map, verify, print statistics
    python tools/sbox_lut3.py --emit tests/host_emul/aesgcm_bs_sbox.inc   # write the mapped S-box (bs_sbox(u32 &x0..&x7))

Bit convention: U0 = most significant input bit (x7), S0 = most significant output bit.
"""
import itertools
import sys

# ---------------------------------------------------------------- the circuit (data)
BP = """
T1 = U0 + U3
T2 = U0 + U5
T3 = U0 + U6
T4 = U3 + U5
T5 = U4 + U6
T6 = T1 + T5
T7 = U1 + U2
T8 = U7 + T6
T9 = U7 + T7
T10 = T6 + T7
T11 = U1 + U5
T12 = U2 + U5
T13 = T3 + T4
T14 = T6 + T11
T15 = T5 + T11
T16 = T5 + T12
T17 = T9 + T16
T18 = U3 + U7
T19 = T7 + T18
T20 = T1 + T19
T21 = U6 + U7
T22 = T7 + T21
T23 = T2 + T22
T24 = T2 + T10
T25 = T20 + T17
T26 = T3 + T16
T27 = T1 + T12
M1 = T13 x T6
M2 = T23 x T8
M3 = T14 + M1
M4 = T19 x U7
M5 = M4 + M1
M6 = T3 x T16
M7 = T22 x T9
M8 = T26 + M6
M9 = T20 x T17
M10 = M9 + M6
M11 = T1 x T15
M12 = T4 x T27
M13 = M12 + M11
M14 = T2 x T10
M15 = M14 + M11
M16 = M3 + M2
M17 = M5 + T24
M18 = M8 + M7
M19 = M10 + M15
M20 = M16 + M13
M21 = M17 + M15
M22 = M18 + M13
M23 = M19 + T25
M24 = M22 + M23
M25 = M22 x M20
M26 = M21 + M25
M27 = M20 + M21
M28 = M23 + M25
M29 = M28 x M27
M30 = M26 x M24
M31 = M20 x M23
M32 = M27 x M31
M33 = M27 + M25
M34 = M21 x M22
M35 = M24 x M34
M36 = M24 + M25
M37 = M21 + M29
M38 = M32 + M33
M39 = M23 + M30
M40 = M35 + M36
M41 = M38 + M40
M42 = M37 + M39
M43 = M37 + M38
M44 = M39 + M40
M45 = M42 + M41
M46 = M44 x T6
M47 = M40 x T8
M48 = M39 x U7
M49 = M43 x T16
M50 = M38 x T9
M51 = M37 x T17
M52 = M42 x T15
M53 = M45 x T27
M54 = M41 x T10
M55 = M44 x T13
M56 = M40 x T23
M57 = M39 x T19
M58 = M43 x T3
M59 = M38 x T22
M60 = M37 x T20
M61 = M42 x T1
M62 = M45 x T4
M63 = M41 x T2
L0 = M61 + M62
L1 = M50 + M56
L2 = M46 + M48
L3 = M47 + M55
L4 = M54 + M58
L5 = M49 + M61
L6 = M62 + L5
L7 = M46 + L3
L8 = M51 + M59
L9 = M52 + M53
L10 = M53 + L4
L11 = M60 + L2
L12 = M48 + M51
L13 = M50 + L0
L14 = M52 + M61
L15 = M55 + L1
L16 = M56 + L0
L17 = M57 + L1
L18 = M58 + L8
L19 = M63 + L4
L20 = L0 + L1
L21 = L1 + L7
L22 = L3 + L12
L23 = L18 + L2
L24 = L15 + L9
L25 = L6 + L10
L26 = L7 + L9
L27 = L8 + L10
L28 = L11 + L14
L29 = L11 + L17
S0 = L6 + L24
S1 = L16 # L26
S2 = L19 # L28
S3 = L6 + L21
S4 = L20 + L22
S5 = L25 + L29
S6 = L13 # L27
S7 = L6 # L23
"""


def sbox_table():
    def xt(a):
        return ((a << 1) ^ (0x1B if a & 0x80 else 0)) & 0xFF

    def mul(a, b):
        r = 0
        while b:
            if b & 1:
                r ^= a
            a = xt(a)
            b >>= 1
        return r
    t = []
    for x in range(256):
        inv = 0
        if x:
            p, b, e = 1, x, 254
            while e:
                if e & 1:
                    p = mul(p, b)
                b = mul(b, b)
                e >>= 1
            inv = p
        s = inv
        r = inv
        for _ in range(4):
            r = ((r << 1) | (r >> 7)) & 0xFF
            s ^= r
        t.append(s ^ 0x63)
    return t


# ---------------------------------------------------------------- netlist
class Net:
    """nodes: name -> (op, a, b) with op in {'in', 'xor', 'and', 'xnor'}; order = topological list of names"""

    def __init__(self):
        self.nodes, self.order, self.outputs = {}, [], []

    def add_in(self, n):
        self.nodes[n] = ("in", None, None)
        self.order.append(n)

    def add(self, n, op, a, b):
        assert n not in self.nodes and a in self.nodes and b in self.nodes, (n, a, b)
        self.nodes[n] = (op, a, b)
        self.order.append(n)

    def simulate(self, invals):
        """invals: dict input name -> int bitmask (parallel evaluation over many patterns); returns dict of all nodes"""
        v = dict(invals)
        full = invals["__mask__"]
        for n in self.order:
            op, a, b = self.nodes[n]
            if op == "in":
                continue
            if op == "xor":
                v[n] = v[a] ^ v[b]
            elif op == "and":
                v[n] = v[a] & v[b]
            elif op == "xnor":
                v[n] = (v[a] ^ v[b]) ^ full
        return v


def parse_bp():
    net = Net()
    for i in range(8):
        net.add_in("U%d" % i)
    for line in BP.strip().splitlines():
        dst, rhs = [s.strip() for s in line.split("=")]
        a, op, b = rhs.split()
        net.add(dst, {"+": "xor", "x": "and", "#": "xnor"}[op], a, b)
    net.outputs = ["S%d" % i for i in range(8)]
    return net


def verify_sbox(net, outputs=None):
    full = (1 << 256) - 1
    vals = {"__mask__": full}
    for i in range(8):          # U0 = MSB
        m = 0
        for x in range(256):
            if (x >> (7 - i)) & 1:
                m |= 1 << x
        vals["U%d" % i] = m
    v = net.simulate(vals)
    sb = sbox_table()
    outs = outputs or net.outputs
    for x in range(256):
        y = 0
        for i in range(8):
            y |= ((v[outs[i]] >> x) & 1) << (7 - i)
        if y != sb[x]:
            return False
    return True


# ---------------------------------------------------------------- LUT3 mapping
A8, B8, C8 = 0xF0, 0xCC, 0xAA       # truth-table columns of the three LUT inputs (v_bitop3_b32 / LOP3 convention)


def enumerate_cuts(net, K=3, max_cuts=60, leaf_ok=None):
    """cuts[n] = list of (leaves tuple sorted, truthtable over those leaves as a function: dict)"""
    cuts = {}
    for n in net.order:
        op, a, b = net.nodes[n]
        triv = (n,)
        if op == "in":
            cuts[n] = [triv]
            continue
        cs = {triv}
        for ca in cuts[a]:
            for cb in cuts[b]:
                u = tuple(sorted(set(ca) | set(cb)))
                if len(u) <= K and (leaf_ok is None or leaf_ok(u)):
                    cs.add(u)
        # drop dominated cuts (a cut that is a superset of another non-trivial cut)
        lst = sorted(cs, key=lambda c: (len(c), c))
        keep = []
        for c in lst:
            if c == triv or not any(set(k) < set(c) for k in keep if k != triv):
                keep.append(c)
        cuts[n] = keep[:max_cuts]
    return cuts


def cut_function(net, n, leaves):
    """truth table (int, 2^len(leaves) bits) of node n over the leaves"""
    k = len(leaves)
    full = (1 << (1 << k)) - 1
    vals = {}
    for i, l in enumerate(leaves):
        m = 0
        for x in range(1 << k):
            if (x >> (k - 1 - i)) & 1:          # leaf 0 = most significant index bit (the 0xF0 column)
                m |= 1 << x
        vals[l] = m

    def ev(x):
        if x in vals:
            return vals[x]
        op, a, b = net.nodes[x]
        assert op != "in", (n, leaves, x)
        va, vb = ev(a), ev(b)
        r = va ^ vb if op == "xor" else (va & vb if op == "and" else (va ^ vb) ^ full)
        vals[x] = r
        return r
    return ev(n)


def map_lut3(net, K=3, rounds=6, leaf_ok=None, verbose=False):
    cuts = enumerate_cuts(net, K, leaf_ok=leaf_ok)
    fanout = {n: 0 for n in net.order}
    for n in net.order:
        op, a, b = net.nodes[n]
        if op != "in":
            fanout[a] += 1
            fanout[b] += 1
    for o in net.outputs:
        fanout[o] += 1
    is_in = {n: net.nodes[n][0] == "in" for n in net.order}
    # pass 1: area flow
    af, best = {}, {}
    for n in net.order:
        if is_in[n]:
            af[n] = 0.0
            continue
        bc, bv = None, None
        for c in cuts[n]:
            if c == (n,):
                continue
            v = (1.0 + sum(af[l] for l in c)) / max(1, fanout[n])
            if bv is None or v < bv - 1e-12:
                bc, bv = c, v
        best[n], af[n] = bc, bv

    def cover(best):
        used, stack = set(), list(net.outputs)
        while stack:
            n = stack.pop()
            if n in used or is_in[n]:
                continue
            used.add(n)
            stack.extend(best[n])
        return used

    # exact-area refinement
    def refs_of(best):
        refs = {n: 0 for n in net.order}
        for o in net.outputs:
            refs[o] += 1
        for n in cover(best):
            for l in best[n]:
                refs[l] += 1
        return refs

    for it in range(rounds):
        refs = refs_of(best)

        def deref(n):
            a = 1
            for l in best[n]:
                if is_in[l]:
                    continue
                refs[l] -= 1
                if refs[l] == 0:
                    a += deref(l)
            return a

        def ref(n, cut=None):
            c = cut if cut is not None else best[n]
            a = 1
            for l in c:
                if is_in[l]:
                    continue
                if refs[l] == 0:
                    a += ref(l)
                refs[l] += 1
            return a

        def deref_cut(c):
            a = 1
            for l in c:
                if is_in[l]:
                    continue
                refs[l] -= 1
                if refs[l] == 0:
                    a += deref(l)
            return a
        changed = 0
        for n in net.order:
            if is_in[n] or refs[n] == 0:
                continue
            deref(n)                                     # free the current implementation of n
            bc, ba = None, None
            for c in cuts[n]:
                if c == (n,):
                    continue
                a = ref(n, c)
                deref_cut(c)
                if ba is None or a < ba or (a == ba and c == best[n]):
                    bc, ba = c, a
            if bc != best[n]:
                changed += 1
            best[n] = bc
            ref(n)
        if verbose:
            print("  refinement %d: %d LUTs, %d changes" % (it, len(cover(best)), changed))
        if not changed:
            break
    used = cover(best)
    luts = []
    for n in net.order:
        if n in used:
            leaves = best[n]
            tt = cut_function(net, n, leaves)
            luts.append((n, leaves, tt))
    return luts


def verify_luts(luts, net):
    """simulate the LUT network on all 256 inputs"""
    full = (1 << 256) - 1
    v = {}
    for i in range(8):
        m = 0
        for x in range(256):
            if (x >> (7 - i)) & 1:
                m |= 1 << x
        v["U%d" % i] = m
    for n, leaves, tt in luts:
        k = len(leaves)
        r = 0
        for idx in range(1 << k):
            if not (tt >> idx) & 1:
                continue
            term = full
            for i, l in enumerate(leaves):
                bit = (idx >> (k - 1 - i)) & 1
                term &= v[l] if bit else (v[l] ^ full)
            r |= term
        v[n] = r
    sb = sbox_table()
    for x in range(256):
        y = 0
        for i in range(8):
            y |= ((v["S%d" % i] >> x) & 1) << (7 - i)
        if y != sb[x]:
            return False
    return True


def tt8(tt, k):
    """widen a k-input truth table (leaf 0 = MSB of the index) to the 8-bit immediate of v_bitop3_b32 with the leaves in
    src0.. and unused trailing sources ignored"""
    out = 0
    for idx in range(8):
        sub = idx >> (3 - k)
        if (tt >> sub) & 1:
            out |= 1 << idx
    return out


def emit_header(luts, path):
    names = {}
    lines = []
    cnt = 0
    for n, leaves, tt in luts:
        cnt += 1
    body = []
    for n, leaves, tt in luts:
        def nm(x):
            if x.startswith("U"):
                return "x%d" % (7 - int(x[1:]))         # U0 = MSB = bit 7
            return "t_" + x
        k = len(leaves)
        args = [nm(l) for l in leaves]
        if k == 1:
            expr = args[0] if tt == 0b10 else "~" + args[0]
        elif k == 2:
            f = {0b0110: "%s ^ %s", 0b1000: "%s & %s", 0b1001: "~(%s ^ %s)", 0b1110: "%s | %s"}.get(tt)
            expr = (f % tuple(args)) if f else "BS_LUT(%s, %s, %s, 0x%02x)" % (args[0], args[1], args[1], tt8(tt, 2))
        else:
            expr = "BS_LUT(%s, %s, %s, 0x%02x)" % (args[0], args[1], args[2], tt8(tt, 3))
        body.append("    const u32 %s = %s;" % (nm(n), expr))
    with open(path, "w") as f:
        f.write("// GENERATED by tools/sbox_lut3.py -- do not edit.  AES S-box (FIPS-197; reference table src/aes_func.vhd:228-301) as %d\n" % len(luts))
        f.write("// three-input boolean instructions (v_bitop3_b32): Boyar-Peralta depth-16 circuit, LUT3-mapped.  x7 = MSB.\n")
        f.write("// BS_LUT(a, b, c, tt): result bit = tt[(a << 2) | (b << 1) | c].\n")
        f.write("#pragma once\n")
        f.write("BS_FN void bs_sbox(u32 &x0, u32 &x1, u32 &x2, u32 &x3, u32 &x4, u32 &x5, u32 &x6, u32 &x7) {\n")
        f.write("\n".join(body) + "\n")
        for i in range(8):
            f.write("    x%d = t_S%d;\n" % (7 - i, i))
        f.write("}\n")


def stats(luts):
    from collections import Counter
    c = Counter(len(l) for _, l, _ in luts)
    return dict(c)


if __name__ == "__main__":
    net = parse_bp()
    assert verify_sbox(net), "the restated Boyar-Peralta circuit does not compute the AES S-box"
    gates = sum(1 for n in net.order if net.nodes[n][0] != "in")
    print("Boyar-Peralta depth-16 circuit: %d gates, verified against the FIPS-197 S-box on all 256 inputs" % gates)
    luts = map_lut3(net, verbose=True)
    assert verify_luts(luts, net), "LUT3 network wrong"
    print("LUT3 mapping: %d instructions (by input count: %s), verified on all 256 inputs" % (len(luts), stats(luts)))
    if len(sys.argv) > 2 and sys.argv[1] == "--emit":
        emit_header(luts, sys.argv[2])
        print("wrote", sys.argv[2])
