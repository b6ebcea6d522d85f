#!/usr/bin/env python3
"""isa_lint.py - a must-be-initialised dataflow check over gfx950 assembly (hipcc -S --cuda-device-only).

Round 6: lcp_primal_kernel<56, ...> returned garbage that changed from build to build when the
reciprocal step lengths were compiled in.  Garbage that moves with the build is what a read of a register nobody wrote looks
like, so this tool walks every kernel of a .s file, builds its control-flow graph from the labels and branches, and reports

  * every VGPR / AGPR / SGPR read that is not preceded by a write on ALL paths from the kernel's entry,
  * the same per LANE for the VGPRs that hold spilled SGPRs (v_writelane_b32 / v_readlane_b32 with an immediate lane),
  * (--exec) every VGPR / AGPR read whose reaching writes all happened under a NARROWER execution mask nesting than the read
    (s_and_saveexec depth by linear scan - a heuristic: lanes that were off at the write hold whatever was there before).

usage: tools/isa_lint.py file.s [--kernel SUBSTR] [--exec] [--max N]
"""
import re
import sys
from collections import defaultdict

REG = re.compile(r"\b([vsa])(\d+)\b|\b([vsa])\[(\d+):(\d+)\]|\b(vcc|exec|m0|scc)(_lo|_hi)?\b")


def regs_of(tok):
    out = []
    for m in REG.finditer(tok):
        if m.group(1):
            out.append((m.group(1), int(m.group(2))))
        elif m.group(3):
            out += [(m.group(3), i) for i in range(int(m.group(4)), int(m.group(5)) + 1)]
        else:
            out.append((m.group(6), 0))
    return out


NO_DEF = ("s_cmp", "s_bitcmp", "s_cbranch", "s_branch", "s_waitcnt", "s_nop", "s_endpgm", "s_barrier", "s_setprio", "s_sleep", "s_setreg",
          "ds_write", "ds_add_f64", "ds_add_u32", "ds_add_f32", "ds_max", "ds_min", "ds_or_b", "ds_and_b", "global_store", "scratch_store",
          "flat_store", "buffer_store", "global_atomic", "s_sendmsg", "s_icache", "s_dcache", "buffer_wbl2", "buffer_inv", "s_trap", "s_sethalt",
          "s_set_gpr", "s_code_end", "global_wb", "global_inv")
SCC_DEF = ("s_add_", "s_sub_", "s_addc", "s_subb", "s_and_", "s_or_", "s_xor_", "s_andn2", "s_orn2", "s_nand", "s_nor", "s_xnor", "s_lshl", "s_lshr",
           "s_ashr", "s_bfe", "s_not", "s_abs", "s_min", "s_max", "s_cmp", "s_bitcmp", "s_wqm", "s_bcnt", "s_absdiff", "s_mul_hi" "s_quadmask")
SCC_USE = ("s_cselect", "s_cbranch_scc", "s_addc", "s_subb", "s_cmov")


class Ins:
    __slots__ = ("line", "text", "op", "defs", "uses", "lane_def", "lane_use", "partial")


def parse_ins(text, lineno):
    t = text.split(";")[0].strip()
    if not t or t.startswith(".") or t.endswith(":"):
        return None
    parts = t.split(None, 1)
    op = parts[0]
    ops = [o.strip() for o in parts[1].split(",")] if len(parts) > 1 else []
    # modifiers behind the last operand (dpp controls, offsets, op_sel ...) carry no registers except in "s[..]" of e64 carry forms
    ins = Ins()
    ins.line, ins.text, ins.op = lineno, t, op
    ins.defs, ins.uses, ins.lane_def, ins.lane_use, ins.partial = [], [], None, None, False
    clean = []
    for o in ops:
        o = re.sub(r"\b(quad_perm:\[[^\]]*\]|row_\w+(:\d+)?|bank_mask:\S+|bound_ctrl:\d|offset\d?:\S+|op_sel\S*|neg_\w+:\S+|clamp|mul:\d|div:\d|nt|sc0|sc1|glc|slc|off|abs|neg)\b", " ", o)
        clean.append(o)
    ops = clean
    if op.startswith(NO_DEF):
        for o in ops:
            ins.uses += regs_of(o)
    elif op == "v_readlane_b32":
        ins.defs += regs_of(ops[0])
        lane = ops[2].strip()
        v = regs_of(ops[1])
        if lane.isdigit() and v:
            ins.lane_use = (v[0], int(lane))
        else:
            ins.uses += v + regs_of(ops[2])
    elif op == "v_writelane_b32":
        lane = ops[2].strip()
        v = regs_of(ops[0])
        ins.uses += regs_of(ops[1])
        if lane.isdigit():
            ins.lane_def = (v[0], int(lane))
        else:
            ins.uses += regs_of(ops[2]); ins.defs += v
    elif op.startswith(("v_cmpx",)):
        ins.defs.append(("exec", 0))
        for o in ops:
            ins.uses += regs_of(o)
    elif op.startswith("v_cmp"):
        ins.defs += regs_of(ops[0])
        for o in ops[1:]:
            ins.uses += regs_of(o)
    elif op.startswith(("v_div_scale", "v_add_co", "v_sub_co", "v_subrev_co", "v_addc_co", "v_subb_co", "v_subbrev_co", "v_mad_u64_u32", "v_mad_i64_i32")):
        ins.defs += regs_of(ops[0]) + regs_of(ops[1])
        for o in ops[2:]:
            ins.uses += regs_of(o)
    elif op.startswith(("s_and_saveexec", "s_or_saveexec", "s_xor_saveexec", "s_andn2_saveexec", "s_orn2_saveexec", "s_andn1_saveexec")):
        ins.defs += regs_of(ops[0]) + [("exec", 0), ("scc", 0)]
        ins.uses += regs_of(ops[1]) + [("exec", 0)]
    elif op.startswith("v_permlane") and "swap" in op:
        ins.defs += regs_of(ops[0]) + regs_of(ops[1])
        ins.uses += regs_of(ops[0]) + regs_of(ops[1])
    elif op in ("s_swappc_b64",):
        ins.defs += regs_of(ops[0]); ins.uses += regs_of(ops[1])
    else:
        if ops:
            ins.defs += regs_of(ops[0])
        for o in ops[1:]:
            ins.uses += regs_of(o)
        if op.startswith(("v_fmac", "v_mac", "v_pk_fmac", "v_dot2c", "v_mfma", "v_smfmac")) and not op.startswith("v_mfma"):
            ins.uses += regs_of(ops[0])
        if "dpp" in op and "bound_ctrl:1" not in t:
            ins.uses += regs_of(ops[0])                      # the old value survives in lanes without a source
        if op.startswith(("v_cndmask_b32_e32", "v_addc_u32_e32", "v_subb_u32_e32", "v_div_fmas")) or (op.startswith("v_cndmask") and len(ops) == 3):
            ins.uses.append(("vcc", 0))
    if op.startswith(SCC_DEF) and not op.startswith(("s_and_saveexec", "s_or_saveexec")):
        ins.defs.append(("scc", 0))
    if op.startswith(SCC_USE):
        ins.uses.append(("scc", 0))
    if op.startswith("s_cbranch_vcc"):
        ins.uses.append(("vcc", 0))
    if op.startswith("s_cbranch_exec"):
        ins.uses.append(("exec", 0))
    return ins


def kernels_of(lines):
    """(name, first line index, last line index) of every function body of the file"""
    out, cur = [], None
    for i, l in enumerate(lines):
        m = re.match(r"^(_Z\w+|[A-Za-z_]\w*):\s*(;.*)?$", l)
        if m and not l.startswith(".L"):
            cur = (m.group(1), i)
        if cur and re.match(r"^\s*s_endpgm", l):
            pass
        if cur and l.startswith(".Lfunc_end"):
            out.append((cur[0], cur[1], i)); cur = None
    return out


def lint(lines, name, lo, hi, exec_mode=False, maxrep=40):
    # blocks
    blocks, label_of, cur = [], {}, []
    def close():
        nonlocal cur
        if cur:
            blocks.append(cur); cur = []
    depth, depth_at = 0, {}
    pending_labels = []
    block_labels = {}
    for i in range(lo + 1, hi):
        l = lines[i]
        m = re.match(r"^(\.LBB\d+_\d+):", l)
        if m:
            close(); pending_labels.append(m.group(1)); continue
        ins = parse_ins(l, i + 1)
        if ins is None:
            continue
        if not cur:
            for lb in pending_labels:
                label_of[lb] = len(blocks)
            pending_labels = []
        if re.match(r"s_(and|or|xor|andn2)_saveexec_b64", ins.op):
            depth += 1
        elif re.match(r"s_or_b64 exec, exec,", ins.text) and depth > 0:
            depth -= 1
        depth_at[ins.line] = depth
        cur.append(ins)
        if ins.op.startswith(("s_branch", "s_cbranch", "s_endpgm")):
            close()
    close()
    nb = len(blocks)
    succ = [[] for _ in range(nb)]
    for b, blk in enumerate(blocks):
        last = blk[-1]
        if last.op == "s_endpgm":
            continue
        if last.op.startswith(("s_branch", "s_cbranch")):
            tgt = last.text.split()[-1]
            if tgt in label_of:
                succ[b].append(label_of[tgt])
            if last.op.startswith("s_cbranch") and b + 1 < nb:
                succ[b].append(b + 1)
        elif b + 1 < nb:
            succ[b].append(b + 1)
    pred = [[] for _ in range(nb)]
    for b in range(nb):
        for s in succ[b]:
            pred[s].append(b)
    # entry state: kernarg pointer, workgroup id, workitem id, exec, vcc (reserved)
    entry = {("s", k) for k in range(0, 16)} | {("v", 0), ("exec", 0)}
    ALL = None
    gen = []
    for blk in blocks:
        g = set()
        for ins in blk:
            g.update(ins.defs)
            if ins.lane_def:
                g.add(("lane",) + ins.lane_def[0] + (ins.lane_def[1],))
        gen.append(g)
    IN = [ALL] * nb
    OUT = [ALL] * nb
    IN[0] = set(entry)
    work = list(range(nb))
    while work:
        b = work.pop(0)
        if b == 0:
            inn = set(entry)
        else:
            ps = [OUT[p] for p in pred[b] if OUT[p] is not ALL]
            if not ps:
                continue
            inn = set.intersection(*ps) if len(ps) > 1 else set(ps[0])
        out = inn | gen[b]
        if OUT[b] is ALL or out != OUT[b] or IN[b] is ALL or inn != IN[b]:
            IN[b], OUT[b] = inn, out
            for s in succ[b]:
                if s not in work:
                    work.append(s)
    reports = []
    for b, blk in enumerate(blocks):
        if IN[b] is ALL:
            continue
        have = set(IN[b])
        for ins in blk:
            for r in ins.uses:
                if r not in have and r[0] in ("v", "a", "s", "vcc", "scc"):
                    reports.append((ins.line, "%s%s" % (r[0], r[1] if r[0] in "vas" else ""), ins.text))
            if ins.lane_use:
                key = ("lane",) + ins.lane_use[0] + (ins.lane_use[1],)
                if key not in have and ins.lane_use[0] not in have:
                    reports.append((ins.line, "%s%d.lane%d" % (ins.lane_use[0][0], ins.lane_use[0][1], ins.lane_use[1]), ins.text))
            have.update(ins.defs)
            if ins.lane_def:
                have.add(("lane",) + ins.lane_def[0] + (ins.lane_def[1],))
    print("== %s: %d blocks, %d instructions, %d possibly-uninitialised reads" % (name[:110], nb, sum(len(b) for b in blocks), len(reports)))
    seen = set()
    n = 0
    for line, reg, text in reports:
        if (reg, text) in seen:
            continue
        seen.add((reg, text))
        if n < maxrep:
            print("   line %6d  %-12s %s" % (line, reg, text))
        n += 1
    return len(reports)


VECTOR = ("v_", "ds_", "global_", "scratch_", "flat_", "buffer_")
LANE_OPS = ("v_readlane_b32", "v_writelane_b32", "v_readfirstlane_b32")


def exec_shadow(lines, name, lo, hi, quiet=None):
    """Vector instructions between the top of a block and the `s_or_b64 exec, exec, sN` that ends a divergent region there.
    The structurizer re-enables the lanes of a finished `if` with that s_or at the top of the join block; spill / copy code the
    register allocator puts IN FRONT of it (it does when an SGPR copy already sits there: SIInstrInfo::isBasicBlockPrologue stops
    at the copy) runs with the if's execution mask: a VGPR -> AGPR spill there saves only the lanes that took the branch and the
    reload later hands the others whatever the register held before.  That is the defect of lcp_primal_kernel<56, ...> in round 5
    (profiles/r06_chain_rootcause.txt).  Lane-indexed moves (v_readlane / v_writelane: SGPR spills) ignore EXEC and are fine."""
    finds = []
    block_top, label, after_execnz = lo, None, False
    shadow = []                      # vector instructions since the top of the current block, as long as nothing else intervened
    clean = False                    # inside the head of a block that an execz / execnz branch skips to, no exec write / branch since its top
    # blocks that are the target of a branch on EXEC: the join blocks of divergent regions (a block that merely follows the body it was
    # merged with holds the body's own instructions in front of the restore - those are not in anybody's shadow)
    skip_targets = set()             # (label, register pair holding the saved mask): the header's `s_and_saveexec sN` / `s_mov sN, exec` + `s_cbranch_execz L`
    recent = []
    for i in range(lo + 1, hi):
        t = lines[i].split(";")[0].split("//")[0].strip()
        if not t or t.startswith(".") or t.endswith(":"):
            if t.endswith(":"):
                recent = []
            continue
        m = re.match(r"^s_cbranch_exec(z|nz)\s+(\S+)", t)
        if m:
            if m.group(1) == "z":                    # (execnz goes to the BODY of a region - tail-duplicated headers, loop back edges)
                for r in recent[-12:]:
                    skip_targets.add((m.group(2), r))
            continue
        m = re.match(r"^s_\w+_saveexec_b64 (s\[\d+:\d+\])", t) or re.match(r"^s_mov_b64 (s\[\d+:\d+\]), exec$", t)
        if m:
            recent.append(m.group(1))
    for i in range(lo + 1, hi):
        l = lines[i]
        t = l.split(";")[0].split("//")[0].strip()
        m = re.match(r"^(\.LBB\d+_\d+|L\d+):", t) or re.match(r"^[0-9a-f]+ <(L\d+)>:", t)
        if m:
            block_top, shadow, clean, label, after_execnz = i, [], True, m.group(1), False
            continue
        if not t or t.startswith(".") or t.endswith(":"):
            continue
        op = t.split()[0]
        if re.match(r"s_or_b64 exec, exec, s\[\d+:\d+\]", t) or re.match(r"s_or_b64 exec, exec, vcc", t) or re.match(r"s_or_saveexec_b64 s\[\d+:\d+\], s\[\d+:\d+\]", t):
            if clean and shadow and ((label, t.split()[-1]) in skip_targets or after_execnz):
                finds.append((i + 1, t, shadow[:]))
            shadow, clean, after_execnz = [], False, False
            continue
        if op.startswith("s_cbranch_execnz"):           # what follows runs with EXEC == 0 until somebody restores it: a spill there saves nothing
            shadow, clean, after_execnz, label = [], True, True, None
            continue
        if op.startswith(("s_cbranch", "s_branch", "s_endpgm", "s_setpc", "s_call")) or (" exec" in t and op.startswith("s_") and t.split()[1].startswith("exec")) \
           or op.startswith(("s_and_saveexec", "s_or_saveexec", "s_xor_saveexec", "s_andn2_saveexec", "v_cmpx")):
            clean, after_execnz = False, False
            continue
        if clean and op.startswith(VECTOR) and op not in LANE_OPS:
            shadow.append((i + 1, t))
    if quiet is not None:
        quiet.extend(finds)
        return sum(len(f[2]) for f in finds)
    if finds or "--verbose" in sys.argv:
        print("== %s: %d vector instructions in the shadow of an exec restore" % (name[:110], sum(len(f[2]) for f in finds)))
    for ln, t, sh in finds:
        print("   line %6d  %s   <- preceded in its block by:" % (ln, t))
        for l2, t2 in sh[:6]:
            print("        line %6d  %s" % (l2, t2))
    return sum(len(f[2]) for f in finds)


SPILL_OPS = ("v_accvgpr_write_b32", "v_accvgpr_read_b32", "scratch_store_dword", "scratch_load_dword")


def fix_file(src, dst):
    """Move the spill instructions found in the shadow of an exec restore to just behind the restore (there they save / reload every
    lane the join block runs with - a superset of the lanes they would have touched).  Anything in a shadow that is not a plain spill
    instruction, or a spill whose registers something between it and the restore touches, is left alone and reported (exit status 2)."""
    lines = open(src).read().split("\n")
    moved, refused = 0, 0
    for name, lo, hi in kernels_of(lines):
        finds = []
        exec_shadow(lines, name, lo, hi, quiet=finds)
        for rl, rt, sh in finds:                        # rl: 1-based line of the restore
            movable = []
            for ln, t in sh:
                ins = parse_ins(t, ln)
                ok = t.split()[0].startswith(SPILL_OPS)
                if ok:
                    mine = set(ins.defs) | set(ins.uses)
                    for j in range(ln, rl - 1):          # lines between this instruction and the restore (0-based j = 1-based j + 1)
                        if (j + 1) in [m[0] for m in movable] or j + 1 == ln:
                            continue
                        other = parse_ins(lines[j], j + 1)
                        if other is None:
                            continue
                        theirs = set(other.defs) | set(other.uses)
                        if other.lane_def: theirs.add(other.lane_def[0])
                        if other.lane_use: theirs.add(other.lane_use[0])
                        if (mine & theirs) - {("exec", 0), ("vcc", 0), ("scc", 0)}:
                            ok = False
                if ok:
                    movable.append((ln, t))
                else:
                    refused += 1
                    print("isa_lint --fix: NOT moved  %s line %d: %s" % (name[:60], ln, t))
            if movable:
                keep = [lines[m[0] - 1] for m in movable]
                for m in movable:
                    lines[m[0] - 1] = None
                lines[rl - 1] = [lines[rl - 1]] + [k + "    ; moved behind the exec restore (tools/isa_lint.py --fix)" for k in keep]
                moved += len(movable)
                print("isa_lint --fix: %s: %d spill instructions moved behind `%s` (line %d)" % (name[:80], len(movable), rt, rl))
    out = []
    for l in lines:
        if l is None:
            continue
        if isinstance(l, list):
            out += l
        else:
            out.append(l)
    open(dst, "w").write("\n".join(out))
    return moved, refused


def main():
    args = sys.argv[1:]
    if "--fix" in args:
        i = args.index("--fix")
        moved, refused = fix_file(args[i + 1], args[i + 2])
        sys.exit(2 if refused else 0)
    if "--shadow-all" in args:                       # every .s of a directory; exit status 1 when anything was found
        import glob
        import os
        d = args[args.index("--shadow-all") + 1]
        total, nk = 0, 0
        fixed_only = "--fixed" in args
        for f in sorted(glob.glob(os.path.join(d, "*.s"))):
            if fixed_only != f.endswith(".fixed.s"):
                continue
            lines = open(f).read().split("\n")
            for name, lo, hi in kernels_of(lines):
                nk += 1
                n = exec_shadow(lines, os.path.basename(f) + ":" + name, lo, hi)
                total += n
        print("isa_lint --shadow-all: %d kernels, %d vector instructions in the shadow of an exec restore" % (nk, total))
        sys.exit(1 if total else 0)
    if "--shadow" in args:
        lines = open(args[0]).read().split("\n")
        sub = args[args.index("--kernel") + 1] if "--kernel" in args else ""
        total = 0
        ks = kernels_of(lines)
        if not ks:                                   # llvm-objdump -d --symbolize-operands: "0000 <name>:" headers
            heads = [i for i, l in enumerate(lines) if re.match(r"^[0-9a-f]+ <[A-Za-z_]\w*>:", l) and not re.match(r"^[0-9a-f]+ <L\d+>:", l)]
            ks = [(re.match(r"^[0-9a-f]+ <(\w+)>:", lines[h]).group(1), h, (heads[j + 1] if j + 1 < len(heads) else len(lines))) for j, h in enumerate(heads)]
        for name, lo, hi in ks:
            if sub in name:
                total += exec_shadow(lines, name, lo, hi)
        print("total", total)
        sys.exit(1 if total else 0)
    path = args[0]
    sub = args[args.index("--kernel") + 1] if "--kernel" in args else ""
    maxrep = int(args[args.index("--max") + 1]) if "--max" in args else 40
    lines = open(path).read().split("\n")
    total = 0
    for name, lo, hi in kernels_of(lines):
        if sub in name:
            total += lint(lines, name, lo, hi, "--exec" in args, maxrep)
    print("total", total)


if __name__ == "__main__":
    main()
