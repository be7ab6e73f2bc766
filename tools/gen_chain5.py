#!/usr/bin/env python3
"""Writes trico_amd/csrc/hip/chain5_bodies.inc: the quad bodies of the float chain decoder (k_fpc32_decode.hip, chain5_run).

The chain of a stream (fpsc.c:308-326) runs on the scalar unit of one wave.  Which of the two predictions a value takes is in
the payload's headers, so the parser wave knows it in advance; it hands the chain the values in quads, each with the address of
the code that decodes exactly that pattern of four kinds (F = FCM-coded, D = DFCM-coded).  A body is straight-line code: an
F value needs no DFCM entry, no wait and no forwarding test, and a value whose successor is F does not load one.

Layout: 69 slots of STRIDE bytes behind a label aligned to STRIDE:
   slot 32 * parity + kinds     body of a quad (parity = index of the quad in its batch & 1: which of the two record
                                registers holds it), kinds bit j = value j is D, bit 4 = the first value of the next quad is
                                (0 for the last quad of a batch)
   slot 64, 66                  first quad of a batch the parser flagged as 64 exact FCM / DFCM hits (see run_check; two slots each)
   slot 68                      branch to the end of the batch
Registers (fixed; chain5_run lists them as clobbered):
   s[36:37] scratch base       s40, s41 temporaries        s42 P = (stride & 0xffc00000) << 5 of the previous value
   s43 loaded DFCM entry       s44 / s45 byte address of the current / previous DFCM entry (alternating)
   s47 saved SCC (run check)   s[52:53], s[54:55] {stride, value} of the previous value, alternating
   s56 FCM entry of the current class                      s[84:99] FCM table (s_movrels / s_movreld, M0 = class)
   s[60:67], s[68:75] record of the current quad, alternating: residuals 0..3, address of its body (2), the bit of its lane, offset of
   the next record from the scratch base
EXEC: the upper half stays zero from the first quad to the last; the lower half is the quad's lane.
SCC between values: 1 = the DFCM entry loaded for the next value is valid, 0 = it was stored by the value before (take the
stride from the register: a scalar load behind a scalar store to the same address is not ordered, see k_fpc32_decode.hip).
"""
import os
import sys

STRIDE = 512
NO_FD_WAIT = os.environ.get("CH5_NO_FD_WAIT", "0") == "1"    # timing experiment only: UNSAFE (a load may pass an older store)
STORE_AFTER_LOAD = os.environ.get("CH5_STORE_AFTER_LOAD", "0") == "1"    # experiment: the stride's store behind the next entry's load
A = ("s52", "s53")           # {stride, value}
B = ("s54", "s55")
REC = (dict(x=["s60", "s61", "s62", "s63"], tgt="s[64:65]", lane="s66", nxt="s67", regs="s[60:67]"),
       dict(x=["s68", "s69", "s70", "s71"], tgt="s[72:73]", lane="s74", nxt="s75", regs="s[68:75]"))


def value(out, j, kind, nextk, cur, prefetch):
    """one value; returns nothing, appends lines"""
    IN, OUT = (A, B) if j % 2 == 0 else (B, A)
    AO, AN = ("s44", "s45") if j % 2 == 0 else ("s45", "s44")
    load = nextk in "DU"
    # every store older than this value's own must be complete before a load is issued (only lgkmcnt(0) means anything for scalar
    # memory), and a D value needs its entry: both wait here
    if kind == "D" or (load and not NO_FD_WAIT):
        out.append("s_waitcnt lgkmcnt(0)")
    if prefetch:
        out.append(prefetch)
    if kind == "D":
        out.append(f"s_cselect_b32 s40, s43, {IN[0]}")
        out.append(f"s_add_u32 s40, s40, {IN[1]}")
        out.append(f"s_xor_b32 {OUT[1]}, {cur['x'][j]}, s40")
    else:
        out.append(f"s_xor_b32 {OUT[1]}, {cur['x'][j]}, s56")
    out.append(f"s_sub_u32 {OUT[0]}, {OUT[1]}, {IN[1]}")
    if not (load and STORE_AFTER_LOAD):
        out.append(f"s_store_dword {OUT[0]}, s[36:37], {AO}")
    out.append(f"s_and_b32 s41, {OUT[0]}, 0xffc00000")
    out.append(f"s_xor_b32 s40, s41, s42")
    out.append(f"s_lshr_b32 {AN}, s40, 20")
    if load:
        out.append(f"s_load_dword s43, s[36:37], {AN}")
        if STORE_AFTER_LOAD:
            out.append(f"s_store_dword {OUT[0]}, s[36:37], {AO}")
    out.append(f"s_movreld_b32 s84, {OUT[1]}")
    out.append(f"s_lshr_b32 m0, {OUT[1]}, 28")
    out.append(f"s_lshl_b32 s42, s41, 5")
    if nextk in "FU":
        out.append("s_movrels_b32 s56, s84")
    out.append(f"v_mov_b32 %[o{j}], {OUT[1]}")
    if load:
        out.append(f"s_cmp_lg_u32 {AN}, {AO}")


def run_check(out, kind):
    """After value 1 of a batch the parser flagged (kind 1: 64 values FCM-coded without residual, 2: DFCM-coded with a zero
    residual): s52 = stride, s53 = value 1, s54 = stride of value 0, s44 = current DFCM address, s45 = the one before.  If
    both strides are one S (0 for kind 1; with a stationary hash for kind 2), |S| < 2^25 and value 1 and value 62 are in one
    FCM class, values 2..63 are value 1 + (K - 1) S: one multiply-add on the vector unit, and the tables end as 62 steps would
    leave them (the entry of that class = value 63, the DFCM entry of the hash = S)."""
    out.append("s_cselect_b32 s47, 1, 0")
    out.append("s_cmp_eq_u32 s52, s54")
    out.append("s_cbranch_scc0 .Lc5_norun%d_%%=" % kind)
    if kind == 1:
        out.append("s_cmp_eq_u32 s52, 0")
    else:
        out.append("s_cmp_eq_u32 s45, s44")
    out.append("s_cbranch_scc0 .Lc5_norun%d_%%=" % kind)
    out.append("s_abs_i32 s40, s52")
    out.append("s_cmp_lt_u32 s40, 0x2000000")
    out.append("s_cbranch_scc0 .Lc5_norun%d_%%=" % kind)
    out.append("s_mul_i32 s40, s52, 61")
    out.append("s_add_u32 s40, s40, s53")                  # value 62
    out.append("s_xor_b32 s41, s40, s53")
    out.append("s_lshr_b32 s41, s41, 28")
    out.append("s_cmp_eq_u32 s41, 0")
    out.append("s_cbranch_scc0 .Lc5_norun%d_%%=" % kind)
    # lane q of output register j holds value 4 q + j = value 1 + (4 q + j - 1) S;  %[lm] = 4 lane - 1
    out.append("s_mov_b64 exec, 0xfffe")
    out.append("v_mul_lo_u32 %[vt], %[lm], s52")
    out.append("v_add_u32 %[o0], s53, %[vt]")
    out.append("v_add_u32 %[vt], s52, %[vt]")
    out.append("v_add_u32 %[o1], s53, %[vt]")
    out.append("s_mov_b64 exec, 0xffff")
    out.append("v_mul_lo_u32 %[vt], %[lm], s52")
    out.append("v_add_u32 %[vt], s52, %[vt]")
    out.append("v_add_u32 %[vt], s52, %[vt]")
    out.append("v_add_u32 %[o2], s53, %[vt]")
    out.append("v_add_u32 %[vt], s52, %[vt]")
    out.append("v_add_u32 %[o3], s53, %[vt]")
    out.append("s_lshr_b32 m0, s40, 28")
    out.append("s_add_u32 s53, s40, s52")                  # value 63
    out.append("s_movreld_b32 s84, s53")
    out.append("s_lshr_b32 m0, s53, 28")
    out.append("s_store_dword s52, s[36:37], s44")
    out.append("s_movrels_b32 s56, s84")
    out.append("s_waitcnt lgkmcnt(0)")
    out.append("s_load_dword s43, s[36:37], s44")
    out.append("s_cmp_eq_u32 s52, s52")                    # SCC = 1: the loaded entry is valid
    out.append("s_branch .Lc5_end_%=")
    out.append(".Lc5_norun%d_%%=:" % kind)
    out.append("s_cmp_lg_u32 s47, 0")


def body(parity, kinds, after, run=0):
    """after = kind of the first value of the next quad ("F" also for the last quad of a batch: the end of the batch loads the
    DFCM entry itself)"""
    cur, nxt = REC[parity], REC[1 - parity]
    out = []
    pf = f"s_load_dwordx8 {nxt['regs']}, s[36:37], {cur['nxt']}"
    # the values of quad q go to lane q of the four output registers: EXEC = that lane (v_writelane cannot take both the value and
    # the lane from scalar registers)
    out.append(f"s_mov_b32 exec_lo, {cur['lane']}")
    for j in range(4):
        nextk = kinds[j + 1] if j < 3 else after
        value(out, j, kinds[j], nextk, cur, pf if j == 0 else None)
        if run and j == 1:
            run_check(out, run)
    # the record of the next quad, requested with value 0, must be there
    if not any(t.startswith("s_waitcnt") for t in out[out.index(pf) + 1:]):
        out.append("s_waitcnt lgkmcnt(0)")
    out.append(f"s_setpc_b64 {nxt['tgt']}")
    return out


def main():
    slots = []
    for parity in range(2):
        for nib in range(32):
            kinds = "".join("D" if (nib >> j) & 1 else "F" for j in range(5))
            slots.append((f"quad {parity} {kinds[:4]} then {kinds[4]}", body(parity, kinds[:4], kinds[4])))
    slots.append(("run of FCM hits", body(0, "FFFF", "F", run=1)))
    slots.append(("run of DFCM hits", body(0, "DDDD", "D", run=2)))
    slots.append(("end of the batch", ["s_branch .Lc5_end_%="]))
    lines = ["// generated by tools/gen_chain5.py - do not edit", "#define CH5_STRIDE %d" % STRIDE, "#define CH5_SLOT_RUN1 64",
             "#define CH5_SLOT_RUN2 66", "#define CH5_SLOT_END 68", "#define CH5_BODIES \\"]
    lines.append('  ".p2align 10\\n .Lc5_body_%=:\\n" \\')
    for name, text in slots:
        lines.append('  /* %s */ \\' % name)
        for t in text:
            lines.append('  "%s\\n" \\' % t)
        lines.append('  ".p2align %d\\n" \\' % (10 if "run" in name else 9))
    lines.append('  ""')
    path = os.environ.get("TRICO_GEN_OUT") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "trico_amd", "csrc", "hip", "chain5_bodies.inc")
    with open(path, "w") as f:
        f.write("\n".join(lines) + "\n")
    n = max(len(t) for _, t in slots)
    print("wrote", os.path.normpath(path), "- longest body", n, "instructions", file=sys.stderr)


if __name__ == "__main__":
    main()
