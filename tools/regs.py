"""Compact register / spill / occupancy table of one kernel file, from the compiler (no GPU needed):
    python tools/regs.py lcp_quad.hip [extra hipcc flags]"""
import os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "lcp_physics_amd", "csrc")
f = sys.argv[1]
extra = sys.argv[2:] + (["-fno-slp-vectorize"] if (f.startswith("lcp_quad") or f == "lcp_solo.hip") and "-fslp-vectorize" not in sys.argv else [])   # (the Makefile's flags for these units)
obj = os.path.join(CSRC, f.replace(".hip", ".o")) if "--keep" in extra else "/dev/null"
extra = [x for x in extra if x != "--keep"]
cmd = ["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-I" + os.path.join(ROOT, "include"), "-Wall",
       "-Wno-unused-function", "-x", "hip", "-c", os.path.join(CSRC, f), "-o", obj, "-Rpass-analysis=kernel-resource-usage"] + extra
err = subprocess.run(cmd, capture_output=True, text=True, cwd=CSRC).stderr
rows, cur = [], None
for line in err.splitlines():
    if "error" in line:
        print(line)
    m = re.search(r"Function Name: (\S+)", line)
    if m:
        cur = {"name": m.group(1)}; rows.append(cur); continue
    m = re.search(r"remark:\s+([A-Za-z \[\]/]+): (\d+)", line)
    if m and cur is not None:
        cur[m.group(1).strip()] = int(m.group(2))
names = subprocess.run(["c++filt"], input="\n".join(r["name"] for r in rows), capture_output=True, text=True).stdout.split("\n")
print("%5s %5s %5s %6s %6s %7s %4s  %s" % ("vgpr", "agpr", "sgpr", "vspill", "sspill", "scratch", "occ", "kernel"))
for r, n in zip(rows, names):
    n = re.sub(r"\(.*", "", n).replace("void lcp::", "")
    print("%5d %5d %5d %6d %6d %7d %4d  %s" % (r.get("VGPRs", -1), r.get("AGPRs", -1), r.get("TotalSGPRs", -1), r.get("VGPRs Spill", -1),
                                             r.get("SGPRs Spill", -1), r.get("ScratchSize [bytes/lane]", -1), r.get("Occupancy [waves/SIMD]", -1), n))
