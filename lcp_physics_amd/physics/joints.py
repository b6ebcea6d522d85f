"""Joints of a batched world (`physics/constraints.py:13-217` of the reference), evaluated on the device.

The reference's `World.Je()` (`world.py:156-170`) asks every joint for its Jacobian blocks at the current pose
(`Joint.J()`, `FixedJoint.J()`, the X / Y / Rot / Total constraints) and `step_dt` moves the joints with the bodies
(`Joint.move`, `world.py:91-92,102-107`).  `JointSet` is that state for B scenes, `JointSet.jacobian()` one launch of
`lcp_joint_jacobian_f64` (include/lcp_hip.h).  No CPU fallback.
"""
import math
from dataclasses import dataclass

import torch

from .. import _lib

JOINT, FIXED, XCON, YCON, ROTCON, TOTAL = 1, 2, 3, 4, 5, 6
ROWS = {JOINT: 2, FIXED: 3, XCON: 1, YCON: 1, ROTCON: 1, TOTAL: 3}
_NAMES = {"joint": JOINT, "fixed": FIXED, "x": XCON, "y": YCON, "rot": ROTCON, "total": TOTAL}


@dataclass
class JointSet:
    """jtype, jb1, jb2 [B,nj] int32; jr1, jrot1 [B,nj] float64 (polar coordinates of a `Joint`'s anchor relative to body 1;
    `jrot1` is state, advanced with the bodies); `e` = number of equality rows."""
    jtype: torch.Tensor
    jb1: torch.Tensor
    jb2: torch.Tensor
    jr1: torch.Tensor
    jrot1: torch.Tensor
    e: int

    @property
    def pose_dependent(self):
        c = self.__dict__.get("_pose_dep")
        if c is None:                                                      # one device read, then cached (the types never change)
            c = self.__dict__["_pose_dep"] = bool(((self.jtype == JOINT) | (self.jtype == FIXED)).any())
        return c

    @property
    def revolute_mask(self):
        """1.0 where the joint is a revolute `Joint` (the only type whose `move` turns an angle), [B,nj] float64."""
        m = self.__dict__.get("_rev_mask")
        if m is None or m.device != self.jtype.device:
            m = self.__dict__["_rev_mask"] = (self.jtype == JOINT).to(torch.float64)
        return m

    def to(self, device):
        mv = lambda t: t.to(device).contiguous()
        return JointSet(mv(self.jtype), mv(self.jb1), mv(self.jb2), mv(self.jr1), mv(self.jrot1), self.e)

    @staticmethod
    def from_list(joints, p0, B=1):
        """`joints`: [("joint", b1, b2_or_None, (x, y)), ("fixed", b1, b2), ("x", b), ("y", b), ("rot", b), ("total", b)];
        `p0` [nb,3] (or [B,nb,3]) the poses the joints are created at (`Joint.__init__`: pos1 = pos - body1.pos, polar
        coordinates with the positive-angle rule, constraints.py:21-23, utils.py:75-82).  Replicated over B scenes."""
        p0 = torch.as_tensor(p0, dtype=torch.float64)
        if p0.dim() == 2:
            p0 = p0.unsqueeze(0).expand(B, -1, -1)
        B, dev = p0.shape[0], p0.device
        nj = len(joints)
        jtype = torch.zeros(B, nj, dtype=torch.int32, device=dev)
        jb1 = torch.zeros(B, nj, dtype=torch.int32, device=dev)
        jb2 = torch.full((B, nj), -1, dtype=torch.int32, device=dev)
        # the anchors' polar coordinates are differentiable functions of `p0`, like the reference's (`pos - body1.pos` ->
        # cart_to_polar, constraints.py:21-23): with a `p0` that requires grad a differentiable roll-out carries d(loss)/d(r1),
        # d(loss)/d(rot1) back to it (ContactWorld.step_autograd -> _JointJacobianFn)
        r_cols, rot_cols = [], []
        zero = torch.zeros(B, dtype=torch.float64, device=dev)
        e = 0
        for k, j in enumerate(joints):
            t = _NAMES[j[0]]
            jtype[:, k], jb1[:, k] = t, int(j[1])
            if t in (JOINT, FIXED) and j[2] is not None:
                jb2[:, k] = int(j[2])
            if t == JOINT:
                d = torch.tensor(j[3], dtype=torch.float64, device=dev).unsqueeze(0) - p0[:, int(j[1]), 1:]
                th = torch.atan2(d[:, 1], d[:, 0])
                r_cols.append(d.norm(dim=1)); rot_cols.append(torch.where(th < 0, th + 2 * math.pi, th))
            else:
                r_cols.append(zero); rot_cols.append(zero)
            e += ROWS[t]
        jr1 = torch.stack(r_cols, dim=1) if nj else torch.zeros(B, 0, dtype=torch.float64, device=dev)
        jrot1 = torch.stack(rot_cols, dim=1) if nj else torch.zeros(B, 0, dtype=torch.float64, device=dev)
        return JointSet(jtype, jb1, jb2, jr1.contiguous(), jrot1.contiguous(), e)

    @staticmethod
    def from_arrays(jtype, jb1, jb2, jr1, jrot1, B):
        """The encoding itself (what oracle/make_golden_world.py records), replicated over B scenes."""
        rep = lambda a, dt_: torch.as_tensor(a).to(dt_).reshape(1, -1).repeat(B, 1).contiguous()
        e = sum(ROWS[int(t)] for t in torch.as_tensor(jtype).tolist())
        return JointSet(rep(jtype, torch.int32), rep(jb1, torch.int32), rep(jb2, torch.int32), rep(jr1, torch.float64),
                        rep(jrot1, torch.float64), e)

    def jacobian(self, p, v=None, dt_scene=None, dt=0.0, vscale=1.0, out=None):
        """Je [B,e,3nb] float32 at pose `p` [B,nb,3] float64.  With `v` the revolute joints are moved first
        (`rot1 += vscale v[body1][0] dt`, dt per scene from `dt_scene` when given)."""
        lib = _lib.load()
        B, nb = p.shape[0], p.shape[1]
        _lib.require_gpu_tensor(p, "p", torch.float64)
        for name, t, dt_ in (("jtype", self.jtype, torch.int32), ("jb1", self.jb1, torch.int32), ("jb2", self.jb2, torch.int32),
                             ("jr1", self.jr1, torch.float64), ("jrot1", self.jrot1, torch.float64)):
            _lib.require_gpu_tensor(t, name, dt_)
        if v is not None:
            _lib.require_gpu_tensor(v, "v", torch.float32)
        if dt_scene is not None:
            _lib.require_gpu_tensor(dt_scene, "dt_scene", torch.float64)
        if out is None:
            out = torch.empty(B, self.e, 3 * nb, dtype=torch.float32, device=p.device)
        P = _lib.ptr
        with torch.cuda.device(p.device):
            rc = lib.lcp_joint_jacobian_f64(B, nb, self.jtype.shape[1], self.e, P(self.jtype), P(self.jb1), P(self.jb2), P(self.jr1),
                                            P(self.jrot1), P(p), P(v), P(dt_scene), float(dt), float(vscale), P(out),
                                            _lib.stream_ptr(p.device))
        _lib.check(rc, "lcp_joint_jacobian_f64")
        return out

    def jacobian_backward(self, nb, rot, gJe):
        """d(loss)/d(pose) [B,nb,3] and d(loss)/d(revolute angles) [B,nj] (float64) from d(loss)/dJe [B,e,3nb] (float32): the
        backward of `jacobian()` at the angles `rot` [B,nj] it was evaluated at - one launch of `lcp_joint_jacobian_backward_f64`
        (the reference's autograd through `Joint.J()` / `update_pos`, constraints.py:26-50)."""
        lib = _lib.load()
        B, nj = self.jtype.shape[0], self.jtype.shape[1]
        gJe = _lib.require_gpu_tensor(gJe.to(torch.float32).contiguous(), "gJe", torch.float32)
        rot = _lib.require_gpu_tensor(rot.contiguous(), "rot", torch.float64)
        g_p = torch.empty(B, nb, 3, dtype=torch.float64, device=gJe.device)
        g_rot = torch.empty(B, nj, dtype=torch.float64, device=gJe.device)
        P = _lib.ptr
        with torch.cuda.device(gJe.device):
            rc = lib.lcp_joint_jacobian_backward_f64(B, nb, nj, self.e, P(self.jtype), P(self.jb1), P(self.jb2), P(self.jr1), P(rot),
                                                     P(gJe), P(g_p), P(g_rot), _lib.stream_ptr(gJe.device))
        _lib.check(rc, "lcp_joint_jacobian_backward_f64")
        return g_p, g_rot

    def anchor_radius_backward(self, nb, rot, gJe):
        """d(loss)/d(jr1) [B,nj] float64 from d(loss)/dJe: the one input of `Joint.J()` the kernel above does not differentiate (a
        constant unless the joints were created at poses that require grad).  pos1 = r1 (cos rot1, sin rot1) enters four entries per
        revolute joint (constraints.py:26-36 with pos2 = body1.pos + pos1 - body2.pos): a gather on the index plan of `jacobian_torch`."""
        B, nj = self.jtype.shape[0], self.jtype.shape[1]
        plan = self._torch_plan(B, nb, torch.float64, gJe.device)
        out = torch.zeros(B, nj, dtype=torch.float64, device=gJe.device)
        if not plan["ks"]:
            return out
        kk, has2, rev = plan["kk"], plan["has2"], plan["rev"]
        ext = torch.cat([gJe.to(torch.float64), gJe.new_zeros(B, self.e, 1, dtype=torch.float64)], dim=2).reshape(B, -1)
        gv = ext.gather(1, plan["flat"]).reshape(B, len(plan["ks"]), 4)       # d/d(-pos1_y), d/d(pos1_x), d/d(pos2_y), d/d(-pos2_x)
        gx, gy = gv[..., 1] - has2 * gv[..., 3], -gv[..., 0] + has2 * gv[..., 2]
        th = rot[:, kk]
        out[:, kk] = rev * (gx * torch.cos(th) + gy * torch.sin(th))
        return out

    def _torch_plan(self, B, nb, dtype, dev):
        """What `jacobian_torch` needs that does not change with the pose: the constant entries of Je ([B,e,3nb]) and, for the
        revolute / fixed joints, the (batch, row, column) indices of their four pose-dependent entries - built once."""
        key = (B, nb, dtype, str(dev))
        plan = self.__dict__.get("_plan")
        if plan is not None and plan["key"] == key:
            return plan
        types = [int(t) for t in self.jtype[0].tolist()]                  # (one list replicated over the batch)
        ar = torch.arange(B, device=dev)
        const = torch.zeros(B, self.e, 3 * nb, dtype=dtype, device=dev)
        one = torch.ones(B, dtype=dtype, device=dev)
        rows, ks = [], []
        row = 0
        for k, t in enumerate(types):
            b1 = self.jb1[:, k].long()
            if t in (JOINT, FIXED):
                b2 = self.jb2[:, k].long()
                has2 = (b2 >= 0).to(dtype)
                b2c = b2.clamp_min(0)
                const[ar, row, 3 * b1 + 1] += one; const[ar, row + 1, 3 * b1 + 2] += one          # J1 = [[-pos1_y, 1, 0], [pos1_x, 0, 1]]
                const[ar, row, 3 * b2c + 1] -= has2; const[ar, row + 1, 3 * b2c + 2] -= has2      # J2 = [[pos2_y, -1, 0], [-pos2_x, 0, -1]]
                if t == FIXED:
                    const[ar, row + 2, 3 * b1] += one; const[ar, row + 2, 3 * b2c] -= has2
                rows.append(row); ks.append(k)
            elif t == XCON:
                const[ar, row, 3 * b1 + 1] += one
            elif t == YCON:
                const[ar, row, 3 * b1 + 2] += one
            elif t == ROTCON:
                const[ar, row, 3 * b1] += one
            elif t == TOTAL:
                for q in range(3):
                    const[ar, row + q, 3 * b1 + q] += one
            row += ROWS.get(t, 0)
        plan = {"key": key, "const": const, "ks": ks}
        if ks:
            nz = 3 * nb
            kk = torch.tensor(ks, dtype=torch.long, device=dev)
            r0 = torch.tensor(rows, dtype=torch.long, device=dev).unsqueeze(0).expand(B, -1)     # [B,njf]
            b1 = self.jb1[:, kk].long()
            b2 = self.jb2[:, kk].long()
            b2c = b2.clamp_min(0)
            c2 = torch.where(b2 >= 0, 3 * b2c, torch.full_like(b2c, nz))                          # no second body: a spare column
            ri = torch.stack([r0, r0 + 1, r0, r0 + 1], dim=2)
            ci = torch.stack([3 * b1, 3 * b1, c2, c2], dim=2)
            ext = torch.zeros(B, self.e, nz + 1, dtype=dtype, device=dev)
            ext[:, :, :nz] = const
            gi = lambda b: b.unsqueeze(-1).expand(-1, -1, 2).contiguous()
            plan.update(kk=kk, g1=gi(b1), g2=gi(b2c), has2=(b2 >= 0).to(dtype), rev=(self.jtype[:, kk] == JOINT).to(dtype),
                        ext=ext.reshape(B, -1), flat=(ri * (nz + 1) + ci).reshape(B, -1), nz=nz)
        self.__dict__["_plan"] = plan
        return plan

    def jacobian_torch(self, p, jrot1=None):
        """The same Jacobian as a differentiable torch expression of the pose `p` [B,nb,3] (float64) and of the revolute joints'
        angles `jrot1` [B,nj] (default: the state) - `Joint.J()` / `FixedJoint.J()` with `update_pos` (constraints.py:26-50,
        64-85): pos1 = r1 (cos rot1, sin rot1), pos2 = body1.pos + pos1 - body2.pos.  Used for the GRADIENT of a
        differentiable step (the values come from `jacobian()`); float64 [B,e,3nb].  The joint types are those of scene 0
        (one list replicated over the batch: `from_list` / `from_arrays`); the bodies may differ per scene.  A dozen tensor
        operations whatever the number of joints (the index plan is built once)."""
        B, nb = p.shape[0], p.shape[1]
        plan = self._torch_plan(B, nb, p.dtype, p.device)
        if not plan["ks"]:
            return plan["const"]
        jrot1 = self.jrot1 if jrot1 is None else jrot1
        kk, has2, rev, nz = plan["kk"], plan["has2"], plan["rev"], plan["nz"]
        r1, th = self.jr1[:, kk] * rev, jrot1[:, kk]                      # (a FixedJoint's anchor is body 1 itself: pos1 = 0)
        pos1 = torch.stack([r1 * torch.cos(th), r1 * torch.sin(th)], dim=2)                       # [B,njf,2]
        pxy = p[:, :, 1:]
        pos2 = pxy.gather(1, plan["g1"]) + pos1 - pxy.gather(1, plan["g2"])
        vals = torch.stack([-pos1[..., 1], pos1[..., 0], has2 * pos2[..., 1], -has2 * pos2[..., 0]], dim=2)
        # the four pose-dependent entries of every such joint land on distinct zeros of the constant part: one scatter
        Je = plan["ext"].scatter(1, plan["flat"], vals.reshape(B, -1))
        return Je.reshape(B, self.e, nz + 1)[:, :, :nz].contiguous()
