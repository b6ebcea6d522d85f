"""Round 6: match the assembly-level register dumps of the failing build (chain_regdump.py) against the values the passing build's
source-level probe recorded (chain_iterates.py dump dhot: t[0..55], udinv, ..., ox, oy)."""
import sys
import numpy as np

tag = sys.argv[1]
D = np.load("gpurun_out/r06_chain_regdump_%s.npz" % tag)["dump"]
ref = np.load("gpurun_out/r06_chain_it_dhot.npz")["dbg"]
names = ["t%d" % j for j in range(56)] + ["udinv", "rx", "ry", "rz.n", "qd", "p", "idn", "kap", "b00", "b10", "b11", "ox", "oy", "os.n", "os.f1", "os.f2", "os.g",
                                          "oz.n", "oz.f1", "oz.f2", "oz.g"]
LDK, NCOL = 57, 56
scene = int(sys.argv[2]) if len(sys.argv) > 2 else 0


def pairs(rows):          # rows [n, 64] int32 -> dict (i -> double[64]) for every consecutive pair (i, i + 1) = (lo, hi)
    lo = rows[:-1].astype(np.uint32).astype(np.uint64)
    hi = rows[1:].astype(np.uint32).astype(np.uint64)
    return ((hi << np.uint64(32)) | lo).view(np.float64)


def close(a, b, lanes):
    a, b = a[lanes], b[lanes]
    return np.all((a == b) | (np.abs(a - b) <= 1e-12 * np.maximum(np.abs(b), 1e-300)) | (np.isnan(a) & np.isnan(b)))


for k in range(D.shape[1]):
    d = D[scene, k]
    t = int(d[0, 0])
    if t == 0:
        continue
    V, A, S, L = pairs(d[1:257]), pairs(d[257:357]), d[357:461], d[512:640]
    lds = L.reshape(-1).astype(np.uint32)
    ldsd = ((lds[1::2].astype(np.uint64) << np.uint64(32)) | lds[0::2].astype(np.uint64)).view(np.float64)     # LDS as doubles
    print("== slot %d, dump point %d   exec %08x%08x" % (k, t, np.uint32(S[103, 0]), np.uint32(S[102, 0])))
    found, missing = [], []
    for j, nm in enumerate(names):
        r = ref[scene, :, j]
        lanes = np.arange(64) if not nm.startswith("t") else np.arange(43)          # rows of the system (n = 43)
        if nm in ("ox", "rx", "p", "qd"): lanes = np.arange(27)
        if nm == "oy": lanes = np.arange(27, 43)
        if nm.startswith(("os", "oz", "rz")) or nm in ("idn", "kap", "b00", "b10", "b11"): lanes = np.arange(16)
        if np.all(r[lanes] == 0) or np.all(r[lanes] == 1.0):
            continue
        hit = [("v[%d:%d]" % (i, i + 1)) for i in range(255) if close(V[i], r, lanes)] + [("a[%d:%d]" % (i, i + 1)) for i in range(99) if close(A[i], r, lanes)]
        (found if hit else missing).append((nm, hit))
    print("   found  :", " ".join("%s=%s" % (n, h[0]) for n, h in found))
    print("   missing:", " ".join(n for n, h in missing))
    if t in (1, 2):
        # the matrix image before the LU: factor it here (no pivoting, as the kernel) and compare with the passing build's factors
        K = ldsd[: NCOL * LDK].reshape(NCOL, LDK)[:, :NCOL].copy()
        n = 43
        Kf = K[:n, :n].copy()
        for c in range(n):
            Kf[c + 1:, c] /= Kf[c, c]
            Kf[c + 1:, c + 1:] -= np.outer(Kf[c + 1:, c], Kf[c, c + 1:])
        tref = ref[scene, :n, :n]
        err = np.abs(Kf - tref) / np.maximum(np.abs(tref), 1e-9 * np.abs(tref).max())
        print("   LDS image factored on the host vs the passing build's factors: worst rel %.2e at %s; K symmetric to %.1e" % (
            err.max(), np.unravel_index(err.argmax(), err.shape), np.abs(K[:n, :n] - K[:n, :n].T).max()))

# the right-hand side the passing build must have solved for: K sol (K = the image of dump point 1, sol = (ox, oy) of the passing build)
d0 = D[scene, 0]
lds = d0[512:640].reshape(-1).astype(np.uint32)
ldsd = ((lds[1::2].astype(np.uint64) << np.uint64(32)) | lds[0::2].astype(np.uint64)).view(np.float64)
K = ldsd[: NCOL * LDK].reshape(NCOL, LDK)[:, :NCOL]
n = 43
sol = np.where(np.arange(64) < 27, ref[scene, :, names.index("ox")], ref[scene, :, names.index("oy")])
rhs = np.zeros(64); rhs[:n] = K[:n, :n] @ sol[:n]
# forward sweep on the host: y = L^-1 rhs, then the vector after the backward sweep
tref = ref[scene, :n, :n]
Lm = np.tril(tref, -1) + np.eye(n); U = np.triu(tref)
y = np.zeros(64); y[:n] = np.linalg.solve(Lm, rhs[:n])
print("rhs (host) lanes 0..8:", rhs[:9])
for k in range(D.shape[1]):
    d = D[scene, k]
    if int(d[0, 0]) == 0: continue
    V, A = pairs(d[1:257]), pairs(d[257:357])
    for nm, vec in (("rhs", rhs), ("y=L^-1 rhs", y), ("sol", sol)):
        sc_ = np.abs(vec[:n]).max()
        hit = [("v[%d:%d]" % (i, i + 1)) for i in range(255) if np.all(np.abs(V[i][:n] - vec[:n]) <= 1e-9 * sc_)] + [("a[%d:%d]" % (i, i + 1)) for i in range(99) if np.all(np.abs(A[i][:n] - vec[:n]) <= 1e-9 * sc_)]
        # partial matches: the register that agrees on the most lanes
        best = max(range(255), key=lambda i: int(np.sum(np.abs(V[i][:n] - vec[:n]) <= 1e-9 * sc_)))
        nb_ = int(np.sum(np.abs(V[best][:n] - vec[:n]) <= 1e-9 * sc_))
        print("slot %d point %d  %-12s %s   best partial v[%d:%d] agrees on %d / %d lanes%s" % (k, int(d[0, 0]), nm, hit, best, best + 1, nb_, n,
              "" if nb_ in (0, n) else "  differs on lanes %s" % [int(l) for l in np.nonzero(np.abs(V[best][:n] - vec[:n]) > 1e-9 * sc_)[0]][:20]))

d = D[scene, 3]
V = pairs(d[1:257])
rxv = ref[scene, :, names.index("rx")]
print("lane  rhs(kernel v[0:1])   rhs(host)   rx   gu(kernel)=rhs+rx   gu(host)")
for l in range(0, 30):
    print("%3d  % .6e  % .6e  % .6e  % .6e  % .6e" % (l, V[0][l], rhs[l], rxv[l], V[0][l] + rxv[l], rhs[l] + rxv[l]))
