"""Batched narrow-phase contact generation on the GPU.

Host-side mirror of the reference's contact pipeline for B independent scenes:
`World.find_contacts` (`physics/world.py:139-142`) + `DiffContactHandler.__call__`
(`physics/contacts.py:57-205`) + the move / penetration-check / dt-halving loop of `World.step_dt`
(`world.py:88-101`), all inside ONE launch of `lcp_move_find_contacts_f64` (include/lcp_hip.h).
The contact record has the reference's format `((normal, p1, p2, penetration), i1, i2)`
(`contacts.py:203-204`), stored structure-of-arrays and padded to `maxc` contacts per scene with a
per-scene `count`.  No CPU fallback.
"""
from dataclasses import dataclass

import numpy as np
import torch

from .. import _lib

CIRCLE, HULL = 0, 1
NV = 8                   # vertex capacity of a hull (lcp_contacts.hip)
EPSILON = 0.1            # physics/utils.py:16  (contact detection margin)
TOL = 1e-6               # physics/utils.py:17  (allowed penetration)


@dataclass
class GeometryBatch:
    """Collision geometry of the bodies of B scenes (constant over a simulation).

    kind [B,nb] int32 (0 circle, 1 hull), radius [B,nb] f64, verts_local [B,nb,8,2] f64 (body frame, the
    order of the reference's `Hull.verts`), nverts [B,nb] int32, no_contact [B,nb,nb] uint8 or None
    (the pairs `World` excludes through `add_no_contact`, bodies.py:117-118)."""
    kind: torch.Tensor
    radius: torch.Tensor
    verts_local: torch.Tensor
    nverts: torch.Tensor
    no_contact: torch.Tensor = None

    @property
    def B(self):
        return self.kind.shape[0]

    @property
    def nb(self):
        return self.kind.shape[1]

    def to(self, device):
        mv = lambda t: None if t is None else t.to(device).contiguous()
        return GeometryBatch(mv(self.kind), mv(self.radius), mv(self.verts_local), mv(self.nverts), mv(self.no_contact))

    @staticmethod
    def from_shapes(shapes, B=1):
        """`shapes`: per body ('circle', rad) or ('rect', (w, h)) or ('hull', verts[nv,2]); replicated B times."""
        nb = len(shapes)
        kind = torch.zeros(nb, dtype=torch.int32)
        radius = torch.zeros(nb, dtype=torch.float64)
        verts = torch.zeros(nb, NV, 2, dtype=torch.float64)
        nverts = torch.zeros(nb, dtype=torch.int32)
        for i, (k, a) in enumerate(shapes):
            if k == "circle":
                kind[i], radius[i] = CIRCLE, float(a)
            else:
                if k == "rect":                      # bodies.py:261-264: [half, half * (-1, 1), -half, -half * (-1, 1)]
                    hw, hh = float(a[0]) / 2, float(a[1]) / 2
                    vs = [[hw, hh], [-hw, hh], [-hw, -hh], [hw, -hh]]
                else:
                    vs = np.asarray(a, dtype=np.float64).tolist()
                if len(vs) > NV:
                    raise ValueError("hulls are limited to %d vertices" % NV)
                kind[i], nverts[i] = HULL, len(vs)
                verts[i, :len(vs)] = torch.tensor(vs, dtype=torch.float64)
        rep = lambda t: t.unsqueeze(0).repeat(B, *([1] * t.dim())).contiguous()
        return GeometryBatch(rep(kind), rep(radius), rep(verts), rep(nverts), None)


class ContactBuffers:
    """Device buffers the contact kernel fills (allocated once and re-used every step by `ContactWorld.step`; one fresh set per
    differentiable step, which keeps the previous set as that step's contact snapshot).  ONE zero-filled allocation, carved into
    the typed arrays: a single memset launch instead of twelve."""

    _LAYOUTS = {}                                                    # (B, nb, maxc) -> total bytes, [(name, dtype, shape, offset, bytes)]

    @classmethod
    def _layout(cls, B, nb, maxc):
        key = (B, nb, maxc)
        lay = cls._LAYOUTS.get(key)
        if lay is None:
            spec = (("p_out", torch.float64, (B, nb, 3)), ("c_pen", torch.float64, (B, maxc)), ("max_pen", torch.float64, (B,)),
                    ("dt_used", torch.float64, (B,)), ("c_n", torch.float32, (B, maxc, 2)), ("c_p1", torch.float32, (B, maxc, 2)),
                    ("c_p2", torch.float32, (B, maxc, 2)), ("c_i1", torch.int32, (B, maxc)), ("c_i2", torch.int32, (B, maxc)),
                    ("count", torch.int32, (B,)), ("trials", torch.int32, (B,)))
            off, items = 0, []
            for name, dt, sh in spec:
                n = int(np.prod(sh)) * (8 if dt == torch.float64 else 4)
                items.append((name, dt, sh, off, n))
                off += (n + 15) // 16 * 16
            lay = cls._LAYOUTS[key] = (off, items)
        return lay

    def __init__(self, B, nb, maxc, device):
        self.maxc = maxc
        total, items = self._layout(B, nb, maxc)
        self._carve(torch.zeros(total, dtype=torch.uint8, device=device), items)

    def _carve(self, backing, items):
        self._backing, self._dims = backing, None
        for name, dt, sh, off, n in items:
            setattr(self, name, backing[off:off + n].view(dt).view(sh))

    def clone(self):
        """A copy in storage of its own (one device copy): what a world keeps when the original became part of an autograd graph."""
        B, nb = self.p_out.shape[0], self.p_out.shape[1]
        new = ContactBuffers.__new__(ContactBuffers)
        new.maxc = self.maxc
        new._carve(self._backing.clone(), self._layout(B, nb, self.maxc)[1])
        if self.p_out.data_ptr() != self._backing.data_ptr():          # (ContactWorld.step swaps p_out with its pose buffer)
            new.p_out = self.p_out.clone()
        return new


def move_and_find_contacts(geom, p_start, v, dt, maxc=16, eps=EPSILON, tol=TOL, strict=True, dt_floor=None,
                           max_trials=64, t=None, out=None):
    """`p <- p_start + v dt`, contacts at the new pose, dt halving while a contact penetrates by more than `tol`
    (`world.py:88-101`) for every scene.  `v=None` detects at `p_start` (the `find_contacts()` of
    `World.__init__`, `world.py:65-66`).  Returns the `ContactBuffers` (p_out = accepted pose)."""
    lib = _lib.load()
    B, nb = geom.B, geom.nb
    _lib.require_gpu_tensor(geom.kind, "kind", torch.int32)
    _lib.require_gpu_tensor(geom.nverts, "nverts", torch.int32)
    _lib.require_gpu_tensor(geom.radius, "radius", torch.float64)
    _lib.require_gpu_tensor(geom.verts_local, "verts_local", torch.float64)
    _lib.require_gpu_tensor(p_start, "p_start", torch.float64)
    if tuple(p_start.shape) != (B, nb, 3) or tuple(geom.verts_local.shape) != (B, nb, NV, 2):
        raise RuntimeError("p_start must be [B,nb,3] and verts_local [B,nb,%d,2]" % NV)
    if v is not None:
        _lib.require_gpu_tensor(v, "v", torch.float32)
    if geom.no_contact is not None:
        _lib.require_gpu_tensor(geom.no_contact, "no_contact", torch.uint8)
    if t is not None:
        _lib.require_gpu_tensor(t, "t", torch.float64)
    dev = p_start.device
    if out is None:
        out = ContactBuffers(B, nb, maxc, dev)
    P = _lib.ptr
    with torch.cuda.device(dev):
        rc = lib.lcp_move_find_contacts_f64(
            B, nb, out.maxc, P(geom.kind), P(geom.radius), P(geom.verts_local), P(geom.nverts), P(geom.no_contact),
            P(p_start), P(v), float(dt), float(dt / 4 if dt_floor is None else dt_floor), int(bool(strict)),
            int(max_trials), float(eps), float(tol), P(out.p_out), P(out.c_n), P(out.c_p1), P(out.c_p2),
            P(out.c_pen), P(out.c_i1), P(out.c_i2), P(out.count), P(out.max_pen), P(out.dt_used), P(t),
            P(out.trials), _lib.stream_ptr(dev))
    _lib.check(rc, "lcp_move_find_contacts_f64")
    return out


def find_contacts(geom, p, maxc=16, eps=EPSILON, out=None):
    """`World.find_contacts` (`world.py:139-142`) for every scene at pose `p` [B,nb,3] (float64)."""
    return move_and_find_contacts(geom, p, None, 0.0, maxc=maxc, eps=eps, out=out, max_trials=1)


def contact_frame_backward(geom, p, cb, g_n, g_p1, g_p2, eps=EPSILON):
    """d(loss)/d(pose) through the contact frame (`lcp_contact_frame_backward_f64`): the chain rule of the reference's
    differentiable contact handler (`contacts.py:57-352`) for the contacts in `cb` detected at pose `p` [B,nb,3] float64 with
    margin `eps` - every record type (circle / circle, circle / hull, hull / hull)."""
    lib = _lib.load()
    B, nb = geom.B, geom.nb
    dev = p.device
    for name, t in (("g_n", g_n), ("g_p1", g_p1), ("g_p2", g_p2)):
        _lib.require_gpu_tensor(t, name, torch.float32)
    _lib.require_gpu_tensor(p, "p", torch.float64)
    dp = torch.empty(B, nb, 3, dtype=torch.float64, device=dev)
    P = _lib.ptr
    with torch.cuda.device(dev):
        rc = lib.lcp_contact_frame_backward_f64(B, nb, cb.c_n.shape[1], P(geom.kind), P(geom.radius), P(geom.verts_local),
                                                P(geom.nverts), P(geom.no_contact), P(p), float(eps), P(cb.count), P(g_n), P(g_p1),
                                                P(g_p2), P(dp), _lib.stream_ptr(dev))
    _lib.check(rc, "lcp_contact_frame_backward_f64")
    return dp


class _FrameSnapshot:
    __slots__ = ("c_n", "c_p1", "c_p2", "c_i1", "c_i2", "count")


class ContactFrameFunction(torch.autograd.Function):
    """The contact list as a differentiable function of the poses: forward hands out the records the detection kernel
    found at `p` (their values are constants of the launch), backward is `lcp_contact_frame_backward_f64`.

        c_n, c_p1, c_p2 = ContactFrameFunction.apply(p, geom, frame)        # frame: a snapshot of the ContactBuffers"""

    @staticmethod
    def forward(ctx, p, geom, frame, eps=EPSILON):
        # The kernels write contact buffers through raw pointers (autograd's version counters never see it), so the records this node
        # hands out and keeps for its backward must belong to nobody else: a `snapshot_frame()`, or the ContactBuffers a
        # differentiable step RETIRED (`ContactWorld.step_autograd` marks them `retired`: the world detects into a fresh set from
        # then on) are used as they are; a LIVE buffer set (e.g. `world.contacts`) is copied first.
        if isinstance(frame, ContactBuffers) and not getattr(frame, "retired", False):
            frame = snapshot_frame(frame)
        ctx.geom, ctx.frame, ctx.eps = geom, frame, eps
        ctx.save_for_backward(p)
        return frame.c_n.detach(), frame.c_p1.detach(), frame.c_p2.detach()

    @staticmethod
    def backward(ctx, g_n, g_p1, g_p2):
        (p,) = ctx.saved_tensors
        z = lambda g, like: torch.zeros_like(like) if g is None else g.contiguous()
        fr = ctx.frame
        dp = contact_frame_backward(ctx.geom, p, fr, z(g_n, fr.c_n), z(g_p1, fr.c_p1), z(g_p2, fr.c_p2), eps=ctx.eps)
        return dp, None, None, None


def snapshot_frame(cb):
    """Copies of the contact records in `cb` (the buffers are re-used by the next detection launch)."""
    fr = _FrameSnapshot()
    fr.c_n, fr.c_p1, fr.c_p2 = cb.c_n.clone(), cb.c_p1.clone(), cb.c_p2.clone()
    fr.c_i1, fr.c_i2, fr.count = cb.c_i1.clone(), cb.c_i2.clone(), cb.count.clone()
    return fr
