"""GPU bring-up diagnostics (not a test): runs small problems through the HIP path with the
per-iteration trace enabled and prints them next to the oracle's trace."""
import ctypes
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from lcp_physics_amd import _lib, scenes  # noqa: E402
from lcp_physics_amd.lcp import lcp_backward, lcp_solve  # noqa: E402
from oracle import pdipm_oracle as O  # noqa: E402
from tests import parity  # noqa: E402


def run(name, lcp32, max_iter=10, compute="f64", dtype=torch.float32):
    dev = "cuda"
    lcp64 = [None if t is None else t.double() for t in lcp32]
    tr = []
    ref = O.lcp_forward(*lcp64, max_iter=max_iter, trace=tr)
    B = lcp32[0].shape[0]
    trace = torch.zeros(B, max_iter, 4, dtype=torch.float64, device=dev)
    lib = _lib.load()
    lib.lcp_debug_set_trace(ctypes.c_void_p(trace.data_ptr()))
    ins = [None if t is None else t.to(device=dev, dtype=dtype).contiguous() for t in lcp32]
    sol = lcp_solve(*ins, max_iter=max_iter, compute=compute)
    torch.cuda.synchronize()
    lib.lcp_debug_set_trace(None)
    ex = parity.err_x(sol.x.double().cpu(), ref.x, lcp64[0], lcp64[1])
    print("== %s  B=%d  err_x max %.3e  iters hip %s  oracle %s  status %s" % (
        name, B, float(ex.max()), sol.iters.cpu().tolist()[:8], ref.iters.tolist()[:8], sol.status.cpu().tolist()[:8]))
    i = int(ex.argmax())
    tc = trace.cpu()
    for it, t in enumerate(tr):
        print("   scene %d it %d  oracle resid %.6e mu %.3e | hip resid %.6e mu %.3e sig %.3e alpha %.4f" % (
            i, it, float(t["resid"][i]), float(t["mu"][i]), float(tc[i, it, 0]), float(tc[i, it, 1]),
            float(tc[i, it, 2]), float(tc[i, it, 3])))
    print("   x hip   ", sol.x[i].double().cpu().numpy()[:8])
    print("   x oracle", ref.x[i].numpy()[:8])
    return sol, ref


if __name__ == "__main__":
    torch.set_printoptions(precision=6, linewidth=200)
    print(torch.cuda.get_device_name(0), torch.cuda.get_device_properties(0))
    run("random m8 noeq f64io", scenes.make_random_lcp(4, 5, 8, 0, seed=1, dtype=torch.float64), dtype=torch.float64)
    run("random m12 eq f64io", scenes.make_random_lcp(4, 6, 12, 2, seed=2, dtype=torch.float64), dtype=torch.float64)
    sc = scenes.make_stack_scenes(B=8, nbox=2, pts_per_interface=4, seed=3, dtype=torch.float32)
    run("stack2x4 f32io/f64", O.assemble_lcp(*sc.assembly_args()))
    sc = scenes.make_stack_scenes(B=8, nbox=4, pts_per_interface=4, seed=3, dtype=torch.float32)
    run("stack4x4 f32io/f64", O.assemble_lcp(*sc.assembly_args()))
    run("stack4x4 f32io/f32", O.assemble_lcp(*sc.assembly_args()), compute="f32")
