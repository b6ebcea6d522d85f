"""A/B aid: run the fused step on a seeded stack batch with the library named by LCP_HIP_LIB and dump the outputs.
    LCP_HIP_LIB=tools/liblcp_hip_prev.so python tools/gpu_ab_dump.py gpurun_out/a.pt ; python tools/gpu_ab_dump.py gpurun_out/b.pt
    python tools/gpu_ab_dump.py --diff gpurun_out/a.pt gpurun_out/b.pt"""
import sys

import torch

import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    if sys.argv[1] == "--diff":
        a, b = torch.load(sys.argv[2]), torch.load(sys.argv[3])
        for k in a:
            d = (a[k].double() - b[k].double()).abs()
            print("%-8s max|a-b| %.3e  (max|a| %.3e)  worst scene %d" % (k, float(d.max()), float(a[k].double().abs().max()),
                                                                       int(d.reshape(d.shape[0], -1).max(dim=1)[0].argmax())))
        return
    from lcp_physics_amd.physics.batched_world import fused_step
    from lcp_physics_amd.scenes import make_stack_scenes
    sc = make_stack_scenes(64, nbox=4, pts_per_interface=4, seed=1).to("cuda", torch.float32)
    out = fused_step(sc)
    torch.cuda.synchronize()
    torch.save({k: out[k].cpu() for k in ("v_new", "z", "s", "iters", "status")}, sys.argv[1])
    print("iters", out["iters"][:8].tolist(), "status", out["status"][:8].tolist(), "v_new[0]", out["v_new"][0].reshape(-1)[:6].tolist())


if __name__ == "__main__":
    main()
