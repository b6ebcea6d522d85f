"""One fingerprint of the kernel sources: the profile JSONs bench.py quotes (rocprofv3 counters, HBM traffic, the compiler's register
report) are stamped with it when they are made (tools/make_profile_json.py, tools/kernel_resources.py) and bench.py compares it with
the sources it runs on - a figure measured on other kernels is printed as stale instead of being quoted silently."""
import glob
import hashlib
import os

CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc")
PATTERNS = ("*.hip", "*.inc", "*.h", "*.cpp", "Makefile", "compile_unit.sh")


def source_sha256():
    h = hashlib.sha256()
    files = sorted(f for pat in PATTERNS for f in glob.glob(os.path.join(CSRC, pat)))
    for f in files:
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()
