"""Seeded synthetic batched scenes (workload generators for bench.py and the tests).

The reference cannot produce a batch of scenes itself (its physics layer is un-batched,
SURVEY.md §0.2), so the BASELINE.json workloads are synthesised here following
SURVEY.md §8(d): a fixed floor plus a stack (or pyramid) of 60x60 boxes, per-scene
jitter, contacts in the format the reference's contact handler emits
(`physics/contacts.py:203-204`: ((normal, p1, p2, penetration), i1, i2), p1/p2 relative
to the body centres, screen coordinates with +y = down = gravity).

Everything here is host-side torch on the CPU; `SceneBatch.to()` moves a batch to a GPU.
"""
from dataclasses import dataclass, fields

import torch

FLOOR_DIMS = (900.0, 10.0)
BOX = 60.0
GRAVITY = 100.0
DT = 1.0 / 30.0          # physics/utils.py:27-28
GAP = 0.05               # inside the eps = 0.1 detection margin (physics/utils.py:16)


@dataclass
class SceneBatch:
    """Structure-of-arrays description of B independent 2-D rigid-body scenes.

    p, v, Mdiag, f : [B, nb, 3]  (rot, x, y) pose / velocity / diagonal mass matrix / force
    rest, fric     : [B, nb]     per-body restitution and friction coefficient
    c_n, c_p1, c_p2: [B, nc, 2]  contact normal and arms;  c_i1, c_i2: [B, nc] int32
    Je             : [B, e, 3nb] equality (joint) Jacobian;  dt: float
    """
    p: torch.Tensor
    v: torch.Tensor
    Mdiag: torch.Tensor
    f: torch.Tensor
    rest: torch.Tensor
    fric: torch.Tensor
    c_n: torch.Tensor
    c_p1: torch.Tensor
    c_p2: torch.Tensor
    c_i1: torch.Tensor
    c_i2: torch.Tensor
    Je: torch.Tensor
    dt: float = DT

    @property
    def B(self):
        return self.v.shape[0]

    @property
    def nb(self):
        return self.v.shape[1]

    @property
    def nc(self):
        return self.c_n.shape[1]

    def to(self, device=None, dtype=None):
        kw = {}
        for fl in fields(self):
            t = getattr(self, fl.name)
            if isinstance(t, torch.Tensor):
                if t.is_floating_point():
                    t = t.to(device=device, dtype=dtype)
                else:
                    t = t.to(device=device)
                t = t.contiguous()
            kw[fl.name] = t
        return SceneBatch(**kw)

    def slice(self, lo, hi):
        kw = {}
        for fl in fields(self):
            t = getattr(self, fl.name)
            kw[fl.name] = t[lo:hi] if isinstance(t, torch.Tensor) else t
        return SceneBatch(**kw)

    def assembly_args(self):
        """Argument tuple of oracle.pdipm_oracle.assemble_lcp / solve_dynamics."""
        return (self.Mdiag, self.v, self.f, self.dt, self.c_n, self.c_p1, self.c_p2,
                self.c_i1, self.c_i2, self.rest, self.fric, self.Je)

    def phys_dict(self):
        d = {k: getattr(self, k) for k in ("Mdiag", "v", "f", "c_n", "c_p1", "c_p2", "rest", "fric")}
        d["c_i1"], d["c_i2"], d["Je"] = self.c_i1, self.c_i2, self.Je
        return d


def _rect_mdiag(mass, w, h):
    """bodies.py:269-270 (Rect inertia) and :44-47 (M = diag(I, m, m))."""
    inertia = mass * (w * w + h * h) / 12.0
    return torch.stack([inertia, mass, mass], dim=-1)


def _interface_contacts(xl, hl, xu, hu, wl, wu, pts, gap):
    """Contact points of one horizontal interface (lower body l, upper body u).

    Returns p1 [B,pts,2] (relative to the lower centre, on its top face) and p2 [B,pts,2]
    (relative to the upper centre).  The reference puts 2 points at the ends of the overlap
    segment (`contacts.py:170-204`); the 4-point variant adds two interior points."""
    lo = torch.maximum(xl - wl / 2, xu - wu / 2)
    hi = torch.minimum(xl + wl / 2, xu + wu / 2)
    ctr, ov = 0.5 * (lo + hi), hi - lo
    if pts == 2:
        fr = torch.tensor([0.5, -0.5], dtype=xl.dtype)
    elif pts == 4:
        fr = torch.tensor([0.5, -0.5, 1.0 / 6.0, -1.0 / 6.0], dtype=xl.dtype)
    else:
        fr = torch.linspace(0.5, -0.5, pts, dtype=xl.dtype)
    xc = ctr.unsqueeze(1) + ov.unsqueeze(1) * fr.unsqueeze(0)            # [B,pts]
    p1 = torch.stack([xc - xl.unsqueeze(1), torch.full_like(xc, -hl / 2)], dim=-1)
    p2 = torch.stack([xc - xu.unsqueeze(1), torch.full_like(xc, hu / 2 + gap)], dim=-1)
    return p1, p2


def _finish(B, nb, centres, interfaces, pts, g, dtype, rest_hi=0.5, ang_sigma=0.05,
            lin_sigma=1.0, gap=GAP):
    """Common tail: per-scene jitter of mass / friction / restitution / velocity, contacts."""
    rnd = lambda *s: torch.rand(*s, generator=g, dtype=torch.float64)
    mass = torch.ones(B, nb, dtype=torch.float64)
    mass[:, 1:] = 0.5 + 1.5 * rnd(B, nb - 1)
    fric = 0.2 + 0.8 * rnd(B, nb)
    rest = rest_hi * rnd(B, nb)
    dims = torch.tensor([[FLOOR_DIMS[0], FLOOR_DIMS[1]]] + [[BOX, BOX]] * (nb - 1), dtype=torch.float64)
    Mdiag = _rect_mdiag(mass, dims[:, 0].unsqueeze(0), dims[:, 1].unsqueeze(0))
    v = torch.randn(B, nb, 3, generator=g, dtype=torch.float64)
    v[:, :, 0] *= ang_sigma
    v[:, :, 1:] *= lin_sigma
    v[:, 0] = 0.0                                            # the floor is pinned
    f = torch.zeros(B, nb, 3, dtype=torch.float64)
    f[:, 1:, 2] = GRAVITY * mass[:, 1:]                      # forces.py:51-67 (Gravity)
    p = torch.zeros(B, nb, 3, dtype=torch.float64)
    p[:, :, 1:] = centres
    nI = len(interfaces)
    nc = nI * pts
    c_p1 = torch.zeros(B, nc, 2, dtype=torch.float64)
    c_p2 = torch.zeros(B, nc, 2, dtype=torch.float64)
    c_i1 = torch.zeros(B, nc, dtype=torch.int32)
    c_i2 = torch.zeros(B, nc, dtype=torch.int32)
    for k, (lo_b, up_b) in enumerate(interfaces):
        p1, p2 = _interface_contacts(centres[:, lo_b, 0], float(dims[lo_b, 1]), centres[:, up_b, 0],
                                     float(dims[up_b, 1]), float(dims[lo_b, 0]), float(dims[up_b, 0]),
                                     pts, gap)
        c_p1[:, k * pts:(k + 1) * pts] = p1
        c_p2[:, k * pts:(k + 1) * pts] = p2
        c_i1[:, k * pts:(k + 1) * pts] = lo_b
        c_i2[:, k * pts:(k + 1) * pts] = up_b
    c_n = torch.zeros(B, nc, 2, dtype=torch.float64)
    c_n[:, :, 1] = 1.0
    Je = torch.zeros(B, 3, 3 * nb, dtype=torch.float64)      # TotalConstraint on body 0
    Je[:, 0, 0] = Je[:, 1, 1] = Je[:, 2, 2] = 1.0            # constraints.py:190-192
    sb = SceneBatch(p=p, v=v, Mdiag=Mdiag, f=f, rest=rest, fric=fric, c_n=c_n, c_p1=c_p1,
                    c_p2=c_p2, c_i1=c_i1, c_i2=c_i2, Je=Je, dt=DT)
    return sb.to(dtype=dtype)


def make_stack_scenes(B, nbox, pts_per_interface=4, seed=1234, dtype=torch.float32, **kw):
    """Floor + `nbox` stacked 60x60 boxes; nc = nbox * pts_per_interface.

    BASELINE configs: nbox=2,pts=4 -> 8 contacts (nineq 32); nbox=4,pts=4 -> 16 (nineq 64);
    pts=2 gives the scene-faithful 4 / 8 contacts the reference's handler would emit."""
    g = torch.Generator().manual_seed(seed)
    nb = nbox + 1
    centres = torch.zeros(B, nb, 2, dtype=torch.float64)
    centres[:, 0] = torch.tensor([500.0, 500.0], dtype=torch.float64)
    top = 500.0 - FLOOR_DIMS[1] / 2
    for k in range(nbox):
        off = -10.0 + 20.0 * torch.rand(B, generator=g, dtype=torch.float64)
        centres[:, k + 1, 0] = 500.0 + off
        centres[:, k + 1, 1] = top - GAP * (k + 1) - BOX / 2 - BOX * k
    interfaces = [(k, k + 1) for k in range(nbox)]
    return _finish(B, nb, centres, interfaces, pts_per_interface, g, dtype, **kw)


def make_pile_scenes(B, pts_per_interface=4, seed=1239, dtype=torch.float32, **kw):
    """Config 5: 10 boxes in a 4-3-2-1 pyramid on the floor, 16 interfaces (64 contacts)."""
    g = torch.Generator().manual_seed(seed)
    nb = 11
    centres = torch.zeros(B, nb, 2, dtype=torch.float64)
    centres[:, 0] = torch.tensor([500.0, 500.0], dtype=torch.float64)
    top = 500.0 - FLOOR_DIMS[1] / 2
    rows = [4, 3, 2, 1]
    pitch = BOX + 8.0
    idx = 1
    row_ids = []
    for r, n in enumerate(rows):
        ids = []
        x0 = 500.0 - pitch * (n - 1) / 2
        for j in range(n):
            off = -3.0 + 6.0 * torch.rand(B, generator=g, dtype=torch.float64)
            centres[:, idx, 0] = x0 + pitch * j + off
            centres[:, idx, 1] = top - GAP * (r + 1) - BOX / 2 - BOX * r
            ids.append(idx)
            idx += 1
        row_ids.append(ids)
    interfaces = [(0, b) for b in row_ids[0]]
    for r in range(1, len(rows)):
        for j, b in enumerate(row_ids[r]):
            interfaces.append((row_ids[r - 1][j], b))
            interfaces.append((row_ids[r - 1][j + 1], b))
    assert len(interfaces) == 16
    return _finish(B, nb, centres, interfaces, pts_per_interface, g, dtype, **kw)


def make_random_lcp(B, nz, m, e, seed=7, dtype=torch.float32, skew=0.5):
    """Random-structure dense LCPs (stress inputs, not physics): Q SPD, G/A dense, F = PSD +
    skew part (monotone LCP, so it is solvable), h > 0 so that x = 0 is strictly feasible."""
    g = torch.Generator().manual_seed(seed)
    rn = lambda *s: torch.randn(*s, generator=g, dtype=torch.float64)
    L = rn(B, nz, nz)
    Q = L @ L.transpose(1, 2) / nz + torch.eye(nz, dtype=torch.float64)
    G = rn(B, m, nz)
    H = rn(B, m, max(1, m // 4))
    S = rn(B, m, m)
    F = 0.1 * (H @ H.transpose(1, 2)) + skew * 0.2 * (S - S.transpose(1, 2))
    p = rn(B, nz)
    h = 0.1 + torch.rand(B, m, generator=g, dtype=torch.float64)
    if e > 0:
        A = rn(B, e, nz)
        b = torch.zeros(B, e, dtype=torch.float64)
    else:
        A, b = None, None
    cast = lambda t: None if t is None else t.to(dtype)
    return tuple(cast(t) for t in (Q, p, G, h, A, b, F))


def make_drop_world(B, nbox=4, box=40.0, seed=2024, gap=(0.3, 2.0), xjit=6.0):
    """Inputs of a `physics.batched_world.ContactWorld`: a fixed floor (TotalConstraint, `constraints.py:176-192`) and
    `nbox` boxes released slightly above each other (per-scene jitter of gaps, offsets, masses, friction and
    restitution), so that contacts are created by the detection kernel as the stack settles.

    Returns dict(shapes, p [B,nb,3] f64, v, Mdiag, f [B,nb,3] f32, rest, fric [B,nb] f32, Je [B,3,3nb] f32)."""
    g = torch.Generator().manual_seed(seed)
    rnd = lambda *s: torch.rand(*s, generator=g, dtype=torch.float64)
    nb = nbox + 1
    shapes = [("rect", FLOOR_DIMS)] + [("rect", (box, box))] * nbox
    p = torch.zeros(B, nb, 3, dtype=torch.float64)
    p[:, 0, 1], p[:, 0, 2] = 500.0, 500.0
    y = torch.full((B,), 500.0 - FLOOR_DIMS[1] / 2, dtype=torch.float64)
    for i in range(1, nb):
        y = y - (gap[0] + (gap[1] - gap[0]) * rnd(B)) - box / 2
        p[:, i, 1] = 500.0 + xjit * (2 * rnd(B) - 1)
        p[:, i, 2] = y
        y = y - box / 2
    mass = torch.ones(B, nb, dtype=torch.float64)
    mass[:, 1:] = 0.5 + 1.5 * rnd(B, nbox)
    dims = torch.tensor([FLOOR_DIMS] + [(box, box)] * nbox, dtype=torch.float64)
    Mdiag = _rect_mdiag(mass, dims[:, 0].unsqueeze(0), dims[:, 1].unsqueeze(0))
    f = torch.zeros(B, nb, 3, dtype=torch.float64)
    f[:, 1:, 2] = mass[:, 1:] * GRAVITY
    rest = 0.1 + 0.4 * rnd(B, nb)
    fric = 0.3 + 0.5 * rnd(B, nb)
    Je = torch.zeros(B, 3, 3 * nb, dtype=torch.float64)
    Je[:, :, :3] = torch.eye(3, dtype=torch.float64)
    f32 = lambda t: t.to(torch.float32)
    return dict(shapes=shapes, p=p, v=torch.zeros(B, nb, 3), Mdiag=f32(Mdiag), f=f32(f), rest=f32(rest), fric=f32(fric),
                Je=f32(Je))


class ChainWorlds:
    """The scene of the reference's `experiments/inference.py:92-125`, B times: a chain of `links` rectangles 20 x 60 hinged by
    revolute joints (the first one to the world, 2 equality rows each), gravity on all links but the first, and a projectile
    (disc of radius 20, restitution 1) pushed towards the last link for `push_time` seconds.

        chains = ChainWorlds(B, links=10)
        world = chains.world(mass, push)          # a fresh `ContactWorld` at the initial pose

    `mass` ([B] tensor or float: the mass of every link - inertia, mass and gravity follow it, bodies.py:44-47,269-270,
    forces.py:51-67) and `push` ([B,3] or 3 floats, times `push_multiplier`: forces.py:29-48) may require grad; the world is
    differentiable with `step(differentiable=True)`.  Everything that needs the host (shapes, joint anchors, index plans) is
    prepared once in the constructor: `world()` issues device work only, so a whole roll-out - forward and backward - can be
    captured into a HIP graph (tools/experiments/mass_inference.py --graph)."""

    def __init__(self, B, links=10, projectile_x=231.7, device="cuda", post_stab=True, maxc=8, dt=1.0 / 30, push_time=0.1,
                 push_multiplier=1500.0):
        from .physics.contacts import GeometryBatch
        from .physics.joints import JointSet
        self.B, self.links, self.nb, self.device = B, links, links + 1, device
        self.post_stab, self.maxc, self.dt, self.push_time, self.push_multiplier = post_stab, maxc, dt, push_time, push_multiplier
        nb = self.nb
        f32 = lambda t: torch.as_tensor(t, dtype=torch.float32, device=device)
        shapes = [("rect", (20.0, 60.0))] * links + [("circle", 20.0)]
        geom = GeometryBatch.from_shapes(shapes, B)
        nocon = torch.zeros(B, nb, nb, dtype=torch.uint8)
        for i in range(1, links):                                                   # bodies[-1].add_no_contact(bodies[-2])
            nocon[:, i, i - 1] = nocon[:, i - 1, i] = 1
        geom.no_contact = nocon
        self.geom = geom.to(device)
        p0 = torch.zeros(nb, 3, dtype=torch.float64)
        for i in range(links):
            p0[i, 1], p0[i, 2] = 300.0, 50.0 + 50.0 * i
        p0[links, 1], p0[links, 2] = projectile_x, float(p0[links - 1, 2]) + 3.3
        joints = [("joint", 0, None, (300.0, 30.0))] + [("joint", i, i - 1, (300.0, 25.0 + 50.0 * i)) for i in range(1, links)]
        self.joints = JointSet.from_list(joints, p0, B=B).to(device)
        self.p0 = p0.unsqueeze(0).repeat(B, 1, 1).to(device)
        self.per_mass = torch.zeros(nb, 3, dtype=torch.float32, device=device)
        self.per_mass[:links] = f32([(20.0 ** 2 + 60.0 ** 2) / 12.0, 1.0, 1.0])     # Rect: inertia = mass (w^2 + h^2) / 12
        self.fixed = torch.zeros(nb, 3, dtype=torch.float32, device=device)
        self.fixed[links] = f32([0.5 * 20.0 ** 2, 1.0, 1.0])                         # Circle of mass 1: inertia = mass r^2 / 2
        self.grav = torch.zeros(nb, 3, dtype=torch.float32, device=device)
        self.grav[1:links, 2] = 100.0                                                # Gravity(g=100) on every link but the first
        self.sel = torch.zeros(1, nb, 1, dtype=torch.float32, device=device)
        self.sel[0, links, 0] = 1.0
        self.rest = torch.full((B, nb), 0.5, dtype=torch.float32, device=device)
        self.rest[:, links] = 1.0
        self.fric = torch.full((B, nb), 0.9, dtype=torch.float32, device=device)
        self.zeros = torch.zeros(B, nb, 3, dtype=torch.float32, device=device)
        self.unit_push = f32([0.0, 1.0, 0.0]).expand(B, 3)                            # ExternalForce.RIGHT
        if str(device).startswith("cuda"):                                          # the joints' host-side plans (one device read each)
            self.joints.pose_dependent; self.joints.revolute_mask; self.joints._torch_plan(B, nb, torch.float64, self.p0.device)

    def world(self, mass=1.0, push=None):
        """`mass`, `push`: floats / sequences (copied to the device: not inside a graph capture) or device tensors."""
        from .physics.batched_world import ContactWorld
        from .physics.joints import JointSet
        B, nb, dev = self.B, self.nb, self.device
        f32 = lambda t: t.to(device=dev, dtype=torch.float32) if torch.is_tensor(t) else torch.tensor(t, dtype=torch.float32, device=dev)
        mass = f32(mass)
        mass = mass.expand(B) if mass.dim() == 0 else mass
        push = self.unit_push if push is None else f32(push)
        push = push.expand(B, 3) if push.dim() == 1 else push
        j0 = self.joints
        js = JointSet(j0.jtype, j0.jb1, j0.jb2, j0.jr1, j0.jrot1.clone(), j0.e)      # fresh joint state, shared plans
        js.__dict__.update({k: v for k, v in j0.__dict__.items() if k.startswith("_")})
        Mdiag = self.per_mass.unsqueeze(0) * mass.reshape(B, 1, 1) + self.fixed.unsqueeze(0)
        grav, sel, t_push, mult = self.grav, self.sel, self.push_time, self.push_multiplier

        def force_fn(t):
            on = (t < t_push).to(torch.float32).reshape(B, 1, 1)
            return grav.unsqueeze(0) * mass.reshape(B, 1, 1) + sel * (push * mult).unsqueeze(1) * on

        return ContactWorld(self.geom, self.p0.clone(), self.zeros.clone(), Mdiag, self.zeros, self.rest, self.fric, joints=js, dt=self.dt,
                            maxc=self.maxc, force_fn=force_fn, post_stab=self.post_stab, check=False)


def make_chain_world(B, links=10, mass=1.0, push=None, device="cuda", **kw):
    """One world of `ChainWorlds` (see there)."""
    return ChainWorlds(B, links=links, device=device, **kw).world(mass, push)
