"""The exec-shadow check of tools/isa_lint.py (round 6, profiles/r06_chain_rootcause.txt).

ROCm 7.2's register allocator can put VGPR -> AGPR spill code at the top of the join block of a divergent `if`, in front of the
`s_or_b64 exec, exec, sN` that re-enables the lanes: the spill saves only the lanes that took the branch.  That was the wrong answer of
`lcp_primal_kernel<56, ...>` in round 5, and `lcp_big_kernel<32, true, false>` carried the same pattern.  The build
(`csrc/compile_unit.sh`) now moves such spill code behind the restore; these tests hold the tool to the recorded pattern and the
shipped assembly to zero findings.  No GPU needed."""
import glob
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import isa_lint  # noqa: E402

# the join block of the round-5 defect (lcp_primal_chain.hip, <56>-column forward kernel, reciprocal step lengths), cut down to its
# control flow: the header saves EXEC in s[0:1] and skips to the join when no lane takes the branch
RECORDED = """
_Z6kernelv:
; %bb.0:
\ts_mov_b64 s[0:1], exec
\tv_readlane_b32 s2, v253, 32
\tv_readlane_b32 s3, v253, 33
\ts_and_b64 s[2:3], s[0:1], s[2:3]
\ts_mov_b64 exec, s[2:3]
\ts_cbranch_execz .LBB0_80
; %bb.79:
\tv_mov_b64_e32 v[40:41], 0
\tv_mov_b64_e32 v[2:3], v[40:41]
.LBB0_80:
\tv_accvgpr_write_b32 a62, v114
\tv_accvgpr_write_b32 a63, v115
\ts_mov_b64 s[18:19], s[54:55]
\ts_or_b64 exec, exec, s[0:1]
\tv_readlane_b32 s0, v253, 34
\tv_accvgpr_read_b32 v0, a62
\ts_endpgm
.Lfunc_end0:
"""

# what must NOT be flagged: the body of a region merged with its own join (tail-duplicated header, `s_cbranch_execnz` to the body),
# and SGPR spills to VGPR lanes (they ignore EXEC)
BENIGN = """
_Z6kernelv:
; %bb.0:
\ts_and_saveexec_b64 s[4:5], s[6:7]
\ts_cbranch_execnz .LBB0_2
\ts_branch .LBB0_3
.LBB0_2:
\tv_cmp_eq_u32_e32 vcc, 3, v43
\tv_cndmask_b32_e32 v7, 0, v5, vcc
\ts_or_b64 exec, exec, s[4:5]
.LBB0_3:
\ts_and_saveexec_b64 s[8:9], s[6:7]
\ts_cbranch_execz .LBB0_5
; %bb.4:
\tv_mov_b32_e32 v1, 0
.LBB0_5:
\tv_writelane_b32 v255, s0, 13
\ts_or_b64 exec, exec, s[8:9]
\ts_endpgm
.Lfunc_end0:
"""


def _finds(text):
    lines = text.split("\n")
    out = []
    for name, lo, hi in isa_lint.kernels_of(lines):
        isa_lint.exec_shadow(lines, name, lo, hi, quiet=out)
    return out


def test_the_recorded_defect_is_found_and_the_fix_moves_exactly_the_two_spills(tmp_path):
    finds = _finds(RECORDED)
    assert len(finds) == 1 and [t for _, t in finds[0][2]] == ["v_accvgpr_write_b32 a62, v114", "v_accvgpr_write_b32 a63, v115"]
    src, dst = tmp_path / "a.s", tmp_path / "b.s"
    src.write_text(RECORDED)
    moved, refused = isa_lint.fix_file(str(src), str(dst))
    assert (moved, refused) == (2, 0)
    fixed = dst.read_text()
    assert _finds(fixed) == []
    body = [l.split(";")[0].strip() for l in fixed.split("\n")]
    i = body.index("s_or_b64 exec, exec, s[0:1]")
    assert body[i + 1: i + 3] == ["v_accvgpr_write_b32 a62, v114", "v_accvgpr_write_b32 a63, v115"]         # behind the restore, in order
    assert body[i - 1] == "s_mov_b64 s[18:19], s[54:55]" and body[i - 2] == ".LBB0_80:"                       # nothing else moved
    assert sorted(x for x in body if x) == sorted(l.split(";")[0].strip() for l in RECORDED.split("\n") if l.split(";")[0].strip())


def test_merged_bodies_and_sgpr_spills_are_not_flagged():
    assert _finds(BENIGN) == []


def test_a_non_spill_instruction_in_a_shadow_is_refused(tmp_path):
    bad = RECORDED.replace("\tv_accvgpr_write_b32 a63, v115\n", "\tv_add_f64 v[10:11], v[10:11], v[12:13]\n")
    src, dst = tmp_path / "a.s", tmp_path / "b.s"
    src.write_text(bad)
    moved, refused = isa_lint.fix_file(str(src), str(dst))
    assert moved == 1 and refused == 1 and len(_finds(dst.read_text())) == 1


def test_every_kernel_the_library_ships_is_free_of_shadow_spills():
    """csrc/asm/*.fixed.s is what compile_unit.sh assembled into liblcp_hip.so (`__graft_entry__.build()` / `make` leave it there);
    every unit of the Makefile must be present, newer than its source, and clean."""
    csrc = os.path.join(ROOT, "lcp_physics_amd", "csrc")
    mk = open(os.path.join(csrc, "Makefile")).read()
    units = [w[:-4] for w in mk.split("SRCS =")[1].split("\n")[0].split() if w.endswith(".hip")]
    missing = [u for u in units if not os.path.exists(os.path.join(csrc, "asm", u + ".fixed.s"))
               or os.path.getmtime(os.path.join(csrc, "asm", u + ".fixed.s")) < os.path.getmtime(os.path.join(csrc, u + ".hip"))]
    if missing:
        subprocess.run(["make", "-C", csrc, "-j8"], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    nk, total = 0, 0
    for u in units:
        lines = open(os.path.join(csrc, "asm", u + ".fixed.s")).read().split("\n")
        for name, lo, hi in isa_lint.kernels_of(lines):
            nk += 1
            out = []
            isa_lint.exec_shadow(lines, name, lo, hi, quiet=out)
            total += len(out)
            assert out == [], (u, name, out[:2])
    assert nk >= 140, nk                                  # (152 kernels at the time of writing)
