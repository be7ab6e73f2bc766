#!/usr/bin/env python3
"""Writes trico_amd/csrc/hip/chain64_bodies.inc: the quad bodies of the double chain decoder (k_fpc64.hip, chain64_run).

Same construction as tools/gen_chain5.py (float chain): the parser wave knows from the headers which of the two predictions every
value takes (F = FCM-coded, D = DFCM-coded; fpsc.c:977-978) and hands the chain the values in quads, each with the address of
straight-line code for exactly its pattern of kinds and the kind of the value behind it.  What that buys here: the ONE table
entry the next value needs is requested as soon as its hash exists - two instructions behind the value for an F successor, six
for a D successor - instead of behind both hashes, both stores and five selects of a branch-free step (30 instructions).

Layout: 69 slots of STRIDE bytes behind a label aligned to 1024:
   slot 32 * parity + kinds     body of a quad (parity = index of the quad in its batch & 1: which of the two record registers
                                holds it), kinds bit j = value j is D, bit 4 = the first value of the next quad is (0 for the
                                last quad of a batch: the code that starts a batch requests the entry itself)
   slot 64                      branch to the end of the batch
Registers (fixed; chain64_run lists them as clobbered):
   s[46:47] FCM table   s[48:49] DFCM table   s[98:99] scratch (ring, counters)
   s[36:37], s[38:39] {stride, value} of odd values / the value before an even one; s[40:41], s[42:43] of even values
   s[44:45] loaded entry   s[50:51] prediction   s52 / s53 FCM offset (current after an odd / even value), s54 / s55 DFCM offset
   s56 forwarded: the entry the value needs is what the value before just stored (a scalar load is not ordered behind a scalar
   store to the same address that is still in flight)   s57, s58 scratch
   s[64:79], s[80:95] record of the current quad, alternating: residuals 0..3 (8 dwords), address of its body (2), bit of its lane,
   offset of the next record from the scratch base, [12] of a batch's first record: its first value is D
EXEC: the upper half stays zero from the first quad to the last; the lower half is the quad's lane.
"""
import os
import sys

STRIDE = 512
A = dict(s="s[36:37]", slo="s36", shi="s37", v="s[38:39]", vlo="s38", vhi="s39")
B = dict(s="s[40:41]", slo="s40", shi="s41", v="s[42:43]", vlo="s42", vhi="s43")
REC = (dict(base=64, regs="s[64:79]", tgt="s[72:73]", lane="s74", nxt="s75"),
       dict(base=80, regs="s[80:95]", tgt="s[88:89]", lane="s90", nxt="s91"))


def value(out, j, kind, nextk, cur, prefetch):
    IN, OUT = (A, B) if j % 2 == 0 else (B, A)
    # offsets: an even value stores under s52 / s54 and leaves the new ones in s53 / s55; an odd one the other way round
    O1O, O1N, O2O, O2N = ("s52", "s53", "s54", "s55") if j % 2 == 0 else ("s53", "s52", "s55", "s54")
    x = "s[%d:%d]" % (cur["base"] + 2 * j, cur["base"] + 2 * j + 1)
    out.append("s_waitcnt lgkmcnt(0)")                       # the entry; and every older store is complete before the next load
    if prefetch:
        out.append(prefetch)
    out.append("s_cmp_lg_u32 s56, 0")
    if kind == "D":
        out.append(f"s_cselect_b64 s[50:51], {IN['s']}, s[44:45]")
        out.append(f"s_add_u32 s50, s50, {IN['vlo']}")
        out.append(f"s_addc_u32 s51, s51, {IN['vhi']}")
    else:
        out.append(f"s_cselect_b64 s[50:51], {IN['v']}, s[44:45]")
    out.append(f"s_xor_b64 {OUT['v']}, {x}, s[50:51]")                                   # fpsc.c:977-981
    sub = [f"s_sub_u32 {OUT['slo']}, {OUT['vlo']}, {IN['vlo']}", f"s_subb_u32 {OUT['shi']}, {OUT['vhi']}, {IN['vhi']}"]
    h1 = [f"s_lshr_b32 s57, {OUT['vhi']}, 9", f"s_and_b32 {O1N}, s57, 0x7ffff8"]        # top 20 bits of the value, as a byte offset
    h2 = [f"s_lshl_b32 s58, {O2O}, 10", f"s_lshr_b32 s57, {OUT['shi']}, 9", "s_xor_b32 s58, s58, s57", f"s_and_b32 {O2N}, s58, 0x7ffff8"]
    stores = [f"s_store_dwordx2 {OUT['v']}, s[46:47], {O1O}", f"s_store_dwordx2 {OUT['s']}, s[48:49], {O2O}"]   # fpsc.c:982-995
    if os.environ.get("CH64_EXPERIMENT") == "no_t2_store":      # timing experiment only: wrong values
        stores = stores[:1]
    if os.environ.get("CH64_EXPERIMENT") == "no_stores":
        stores = []
    if nextk == "F":
        out += h1
        out.append(f"s_load_dwordx2 s[44:45], s[46:47], {O1N}")
        out.append(f"s_cmp_eq_u32 {O1N}, {O1O}")
        out.append("s_cselect_b32 s56, 1, 0")
        out += sub + stores + h2
    else:
        out += sub + h2
        out.append(f"s_load_dwordx2 s[44:45], s[48:49], {O2N}")
        out.append(f"s_cmp_eq_u32 {O2N}, {O2O}")
        out.append("s_cselect_b32 s56, 1, 0")
        out += stores + h1
    out.append(f"v_mov_b32 %[o{j}lo], {OUT['vlo']}")
    out.append(f"v_mov_b32 %[o{j}hi], {OUT['vhi']}")


def body(parity, kinds, after):
    cur, nxt = REC[parity], REC[1 - parity]
    out = [f"s_mov_b32 exec_lo, {cur['lane']}"]
    pf = f"s_load_dwordx16 {nxt['regs']}, s[98:99], {cur['nxt']}"
    for j in range(4):
        nextk = kinds[j + 1] if j < 3 else after
        value(out, j, kinds[j], nextk, cur, pf if j == 0 else None)
    # (the record of the next quad was requested with value 0 and every later value began with a wait)
    out.append(f"s_setpc_b64 {nxt['tgt']}")
    return out


def size_of(line):
    op = line.split()[0]
    if op in ("s_load_dwordx2", "s_load_dwordx16", "s_store_dwordx2"):
        return 8
    if "0x7ffff8" in line:
        return 8
    return 4


def main():
    slots = []
    for parity in range(2):
        for nib in range(32):
            kinds = "".join("D" if (nib >> j) & 1 else "F" for j in range(5))
            slots.append((f"quad {parity} {kinds[:4]} then {kinds[4]}", body(parity, kinds[:4], kinds[4])))
    slots.append(("end of the batch", ["s_branch .Lc64_end_%="]))
    worst = max(sum(size_of(t) for t in text) for _, text in slots)
    assert worst <= STRIDE, worst
    lines = ["// generated by tools/gen_chain64.py - do not edit", "#define CH64_STRIDE %d" % STRIDE, "#define CH64_SLOT_END 64", "#define CH64_BODIES \\"]
    lines.append('  ".p2align 10\\n .Lc64_body_%=:\\n" \\')
    for name, text in slots:
        lines.append('  /* %s */ \\' % name)
        for t in text:
            lines.append('  "%s\\n" \\' % t)
        lines.append('  ".p2align 9\\n" \\')
    lines.append('  ""')
    path = os.environ.get("TRICO_GEN_OUT") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "trico_amd", "csrc", "hip", "chain64_bodies.inc")
    with open(path, "w") as f:
        f.write("\n".join(lines) + "\n")
    print("wrote", os.path.normpath(path), "- largest body", worst, "bytes", file=sys.stderr)


if __name__ == "__main__":
    main()
