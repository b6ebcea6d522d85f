#!/usr/bin/env python3
"""asm_dump_instrument.py - insert whole-register-file dumps into ONE kernel of a hipcc -S device listing, at the assembly level.

Round 6 root-cause aid (profiles/r06_chain_rootcause.txt): every SOURCE-level probe cured the failing build of
lcp_primal_kernel<56, ...>, so the probes go in behind the compiler: `s_call_b64` to one dump routine at chosen line numbers of the
.s (places where the execution mask is full), register allocation and scheduling of the failing build untouched.

Registers the routine owns: s[100:101] (return address; the kernel's .amdhsa_next_free_sgpr is raised to 102), a246 .. a255
(.amdhsa_next_free_vgpr raised to 512).  It touches neither SCC, VCC nor EXEC and restores every VGPR it borrows.

Memory: the dump goes to  base = ((ws + 0x1fffffff) & ~0x0fffffff) + scene * 2^22 + slot * 2^18, `ws` the kernel's workspace pointer
(kernarg offset --ws-off), slot = the running count of dumps this wavefront made.  Row r of a slot (256 bytes, one dword per lane):
  0: tag (the dump point's number)    1 .. 256: v0 .. v255    257 .. 356: a0 .. a99    357 .. 460: s0 .. s99, vcc_lo, vcc_hi, exec_lo, exec_hi
  512 .. 639: LDS bytes 0 .. 32767 (dword 64 r' + lane)
The caller allocates a workspace of >= 512 MB + B * 4 MB (tools/experiments/chain_regdump.py).

usage: asm_dump_instrument.py in.s out.s --kernel SUBSTR --ws-off 0xc8 --at LINE[,LINE...]     (1-based line numbers of in.s; the call
       is inserted BEFORE that line)
"""
import re
import sys


def routine(label):
    o = []
    A = o.append
    A("%s:" % label)
    A("\ts_waitcnt vmcnt(0) lgkmcnt(0)")
    for i in range(4):
        A("\tv_accvgpr_write_b32 a%d, v%d" % (250 + i, i))
    A("\tv_accvgpr_read_b32 v0, a254")
    A("\tv_accvgpr_read_b32 v1, a255")
    A("\tv_accvgpr_read_b32 v2, a248")
    A("\tv_lshlrev_b32_e32 v3, 18, v2")
    A("\tv_add_u32_e32 v0, v0, v3")
    A("\tv_add_u32_e32 v2, 1, v2")
    A("\tv_accvgpr_write_b32 a248, v2")
    A("\tv_mbcnt_lo_u32_b32 v2, -1, 0")
    A("\tv_mbcnt_hi_u32_b32 v2, -1, v2")
    A("\tv_lshlrev_b32_e32 v2, 2, v2")
    A("\tv_accvgpr_write_b32 a249, v2")           # lane * 4, for the LDS reads
    A("\tv_add_u32_e32 v0, v0, v2")
    row = [0]

    def store(reg):
        off = (row[0] % 16) * 256
        A("\tglobal_store_dword v[0:1], %s, off offset:%d" % (reg, off))
        row[0] += 1
        if row[0] % 16 == 0:
            A("\tv_add_u32_e32 v0, 0x1000, v0")

    def skip_to(r):
        while row[0] < r:
            row[0] += 1
            if row[0] % 16 == 0:
                A("\tv_add_u32_e32 v0, 0x1000, v0")

    A("\tv_accvgpr_read_b32 v3, a247")
    store("v3")                                     # row 0: tag
    for i in range(4):                              # v0 .. v3 from their parking places
        A("\tv_accvgpr_read_b32 v3, a%d" % (250 + i))
        store("v3")
    for i in range(4, 256):
        store("v%d" % i)
    for i in range(100):
        A("\tv_accvgpr_read_b32 v3, a%d" % i)
        store("v3")
    for i in range(100):
        A("\tv_mov_b32_e32 v3, s%d" % i)
        store("v3")
    for nm in ("vcc_lo", "vcc_hi", "exec_lo", "exec_hi"):
        A("\tv_mov_b32_e32 v3, %s" % nm)
        store("v3")
    skip_to(512)
    A("\tv_accvgpr_read_b32 v2, a249")
    for r in range(128):
        A("\tds_read_b32 v3, v2 offset:%d" % (256 * r))
        A("\ts_waitcnt lgkmcnt(0)")
        store("v3")
    A("\ts_waitcnt vmcnt(0)")
    for i in range(4):
        A("\tv_accvgpr_read_b32 v%d, a%d" % (i, 250 + i))
    A("\ts_nop 4")
    A("\ts_setpc_b64 s[100:101]")
    return o


def main():
    a = sys.argv[1:]
    src, dst = a[0], a[1]
    sub = a[a.index("--kernel") + 1]
    ws_off = int(a[a.index("--ws-off") + 1], 0)
    at = [int(x) for x in a[a.index("--at") + 1].split(",")]
    lines = open(src).read().split("\n")
    start = None
    for i, l in enumerate(lines):
        if re.match(r"^_Z\w+:", l) and sub in l:
            start = i
            break
    assert start is not None, "kernel not found"
    end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))
    last_endpgm = max(i for i in range(start, end) if re.match(r"^\s*s_endpgm", lines[i]))
    label = ".Llcp_dump_%d" % start
    out = []
    tag = {ln: k + 1 for k, ln in enumerate(at)}
    for i, l in enumerate(lines):
        ln = i + 1
        if ln in tag and start < i < end:
            out.append("\tv_accvgpr_write_b32 a247, %d" % tag[ln])
            out.append("\ts_call_b64 s[100:101], %s" % label)
        out.append(l)
        if i == start + 1:                           # behind "; %bb.0:": s[0:1] = kernarg pointer, s2 = workgroup id, v0 = lane; v1 is free
            out += ["\ts_load_dwordx2 s[100:101], s[0:1], 0x%x" % ws_off,
                    "\ts_waitcnt lgkmcnt(0)",
                    "\ts_add_u32 s100, s100, 0x1fffffff",
                    "\ts_addc_u32 s101, s101, 0",
                    "\ts_and_b32 s100, s100, 0xf0000000",
                    "\tv_mov_b32_e32 v1, s2",
                    "\tv_lshlrev_b32_e32 v1, 22, v1",
                    "\tv_add_u32_e32 v1, s100, v1",
                    "\tv_accvgpr_write_b32 a254, v1",
                    "\tv_mov_b32_e32 v1, s101",
                    "\tv_accvgpr_write_b32 a255, v1",
                    "\tv_accvgpr_write_b32 a248, 0"]
        if i == last_endpgm:
            out += routine(label)
    text = "\n".join(out)
    # the kernel descriptor: more registers
    ks = text.index(".amdhsa_kernel " + lines[start].split(":")[0])
    ke = text.index(".end_amdhsa_kernel", ks)
    desc = text[ks:ke]
    desc = re.sub(r"\.amdhsa_next_free_vgpr \d+", ".amdhsa_next_free_vgpr 512", desc)
    desc = re.sub(r"\.amdhsa_next_free_sgpr \d+", ".amdhsa_next_free_sgpr 102", desc)
    text = text[:ks] + desc + text[ke:]
    open(dst, "w").write(text)
    print("instrumented %s: %d dump points, routine of %d lines" % (lines[start].split(":")[0][:60], len(at), len(routine(label))))


if __name__ == "__main__":
    main()
