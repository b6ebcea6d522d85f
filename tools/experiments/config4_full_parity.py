"""One-off (GPU box): the configs[4] parity report of tests/test_hip_headline_parity.py on EVERY scene of the 4096 x 64 batch instead of the
512 the test samples (the fp64 oracle factors 256 x 256 systems: ~a minute on the box's host).   python tools/experiments/config4_full_parity.py [sample]"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tests import test_hip_headline_parity as T

sample = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
case = next(c for c in T.CASES if c[0].startswith("configs4"))
label, kind, B, nbox, seed, rows, _, gates = case
t = time.time()
rep, out, grads, scg = T._run_case(kind, B, nbox, seed, rows, sample)
rep["wall_s"] = time.time() - t
print(json.dumps({"case": label, "sample": sample, "report": rep}))
