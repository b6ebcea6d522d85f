"""Round 6, chain root cause: the iterate the <56>-column kernel keeps after max_iter = 1, 2, 3, 4 passes (its workspace block:
x[64] y[24] z[4][64] s[4][64]), per library variant (LCP_HIP_LIB); `compare` prints where two variants first part."""
import sys
import numpy as np
import torch
sys.path.insert(0, ".")


def dump(tag):
    from tests.test_hip_primal import _with_joint_rows
    from lcp_physics_amd import scenes, _lib
    from lcp_physics_amd.physics.batched_world import solve_dynamics
    from lcp_physics_amd.physics.contacts import ContactBuffers
    B, nbox, pts, e = 32, 8, 2, 16
    sc = _with_joint_rows(scenes.make_stack_scenes(B=B, nbox=nbox, pts_per_interface=pts, seed=900 + nbox + e, dtype=torch.float32), e)
    scg = sc.to(device="cuda")
    cb = ContactBuffers(sc.B, sc.nb, sc.nc, "cuda")
    cb.c_n, cb.c_p1, cb.c_p2, cb.c_i1, cb.c_i2 = scg.c_n, scg.c_p1, scg.c_p2, scg.c_i1, scg.c_i2
    count = torch.full((B,), sc.nc, dtype=torch.int32, device="cuda")
    rec = {}
    for k in (1, 2, 3, 4, 10):
        out = solve_dynamics(sc.B, sc.nb, sc.nc, e, count, scg.Mdiag, scg.v, scg.f, scg.rest, scg.fric, cb, scg.Je, sc.dt, max_iter=k)
        torch.cuda.synchronize()
        ws = out["ws"].view(torch.float64)[: B * 792].reshape(B, 792).cpu().numpy()
        rec["ws%d" % k] = ws
        if k == 1:
            rec["dbg"] = out["ws"].view(torch.float64)[B * 792: B * 792 + B * 64 * 80].reshape(B, 64, 80).cpu().numpy()
        rec["it%d" % k] = out["iters"].cpu().numpy()
        rec["v%d" % k] = out["v_new"].cpu().numpy()
    np.savez("gpurun_out/r06_chain_it_%s.npz" % tag, nz=3 * sc.nb, e=e, nc=sc.nc, **rec)


def compare(a, b):
    A, Bq = np.load("gpurun_out/r06_chain_it_%s.npz" % a), np.load("gpurun_out/r06_chain_it_%s.npz" % b)
    nz, e, nc = int(A["nz"]), int(A["e"]), int(A["nc"])
    print("compare", a, "vs", b, "nz", nz, "e", e, "nc", nc)
    if "--dbg" in sys.argv:
        da, db = A["dbg"], Bq["dbg"]
        names = ["t%d" % j for j in range(56)] + ["udinv", "rx", "ry", "rz.n", "qd", "p", "idn", "kap", "b00", "b10", "b11", "ox", "oy", "os.n", "os.f1", "os.f2", "os.g",
                                                  "oz.n", "oz.f1", "oz.f2", "oz.g"]
        for j, nm in enumerate(names):
            d = np.abs(da[:, :, j] - db[:, :, j])
            sc_ = max(np.abs(db[:, :, j]).max(), 1e-300)
            both_nan = np.isnan(da[:, :, j]) & np.isnan(db[:, :, j])
            bad = np.argwhere((d > 1e-9 * sc_) & ~both_nan | (np.isnan(d) & ~both_nan))
            if len(bad):
                lanes = sorted(set(int(l) for _, l in bad))
                s0 = int(bad[0][0]); l0 = int(bad[0][1])
                print("  dbg %-6s differs: %d scenes, lanes %s  e.g. scene %d lane %d: %r vs %r" % (nm, len(set(int(s) for s, _ in bad)), lanes, s0, l0, float(da[s0, l0, j]), float(db[s0, l0, j])))
    for k in (1, 2, 3, 4, 10):
        wa, wb = A["ws%d" % k][:, 64:], Bq["ws%d" % k][:, 64:]
        parts = {"x": (0, nz), "y": (64, 64 + e)}
        for j, nm in enumerate(("z.n", "z.f1", "z.f2", "z.g", "s.n", "s.f1", "s.f2", "s.g")):
            parts[nm] = (88 + 64 * j, 88 + 64 * j + 64)
        msg = []
        for nm, (lo, hi) in parts.items():
            d = np.abs(wa[:, lo:hi] - wb[:, lo:hi])
            sc_ = np.maximum(np.abs(wb[:, lo:hi]).max(), 1e-300)
            bad = np.argwhere(d > 1e-9 * sc_)
            if len(bad):
                lanes = sorted(set(int(l) for _, l in bad))
                msg.append("%s: %d scenes, lanes %s%s worst %.2e" % (nm, len(set(int(s) for s, _ in bad)), lanes[:12], "..." if len(lanes) > 12 else "", float(d.max() / sc_)))
        print("  max_iter %2d  iters %s vs %s :" % (k, A["it%d" % k][:6].tolist(), Bq["it%d" % k][:6].tolist()), "; ".join(msg) if msg else "identical to 1e-9")


if __name__ == "__main__":
    if sys.argv[1] == "dump":
        dump(sys.argv[2])
    else:
        compare(sys.argv[2], sys.argv[3])
