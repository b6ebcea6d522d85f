"""Soak run (GPU box): B scenes of `nbox` boxes dropped on a floor, stepped for thousands of steps through ContactWorld.run (HIP graph replay);
reports what a long roll-out leaves behind - status bits ever raised (singular / NaN / truncated), NaN poses, how far the settled stacks
sit from rest, the deepest penetration, memory growth.   python tools/experiments/soak_world.py [--batch 4096] [--steps 3000] [--nbox 4]"""
import argparse, json, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=4096)
    ap.add_argument("--nbox", type=int, default=4)
    ap.add_argument("--steps", type=int, default=3000)
    ap.add_argument("--maxc", type=int, default=16)
    args = ap.parse_args()
    from lcp_physics_amd import _lib, scenes
    from lcp_physics_amd.physics import batched_world as bw
    from lcp_physics_amd.physics import contacts as ct
    dev = torch.device("cuda")
    w = scenes.make_drop_world(args.batch, nbox=args.nbox)
    geom = ct.GeometryBatch.from_shapes(w["shapes"], args.batch).to(dev)
    g = lambda k: w[k].to(dev)
    world = bw.ContactWorld(geom, g("p"), g("v"), g("Mdiag"), g("f"), g("rest"), g("fric"), Je=g("Je"), maxc=args.maxc)
    world.run(8, graph=True)
    torch.cuda.synchronize()
    mem0 = torch.cuda.memory_allocated()
    t0 = time.perf_counter()
    done, chunks = 8, []
    while done < args.steps:
        n = min(500, args.steps - done)
        world.run(n, graph=True)
        done += n
        torch.cuda.synchronize()
        chunks.append({"steps": done, "max_speed": float(world.v[:, 1:].abs().max()), "max_penetration": float(world.contacts.max_pen.max()),
                       "mean_contacts": float(world.contacts.count.float().mean())})
    wall = time.perf_counter() - t0
    st = world.sticky_status
    bits = {name: int(((st & bit) != 0).sum()) for name, bit in (("singular_Q", _lib.ST_SINGULAR_Q), ("singular_S11", _lib.ST_SINGULAR_S11),
                                                                  ("singular_T", _lib.ST_SINGULAR_T), ("nan", _lib.ST_NAN), ("truncated", _lib.ST_TRUNCATED))}
    out = {"experiment": "soak: ContactWorld.run, HIP graph replay", "batch": args.batch, "bodies": args.nbox + 1, "steps": done,
           "sim_steps_per_s": args.batch * (done - 8) / wall, "scenes_with_status_bit_ever_set": bits,
           "nan_poses": int(torch.isnan(world.p).any(dim=2).any(dim=1).sum()), "floor_moved_max": float((world.p[:, 0] - g("p")[:, 0]).abs().max()),
           "clock_min_max_s": [float(world.t.min()), float(world.t.max())], "memory_growth_bytes": int(torch.cuda.memory_allocated() - mem0),
           "progress": chunks}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
