"""CPU oracle for the narrow-phase contact generation.  TEST INFRASTRUCTURE ONLY.

Clean-room numpy restatement (float64, plain Python loops - it is only used on small cases) of the
reference's differentiable contact handler, `/root/reference/lcp_physics/physics/contacts.py`:

  :57-205   DiffContactHandler.__call__   -> `collide_pair`
     :68-79    circle vs circle            -> `_circle_circle`
     :80-141   circle vs hull (GJK, then SAT when the centre is inside)  -> `_circle_hull`
     :142-201  hull vs hull (SAT both ways, incident edge, clipping)      -> `_hull_hull`
  :207-217  get_support                    -> `_support`
  :220-250  test_separations               -> `_test_separations`
  :253-268  get_incident_edge              -> `_incident_edge`
  :270-292  clip_segment_to_line           -> `_clip`
  :295-330  get_closest                    -> `_closest`
  :332-352  get_barycentric_coords         -> `_bary2`, `_bary3`

and of the pieces of `physics/world.py` that drive it: the pair enumeration of `find_contacts`
(:139-142; the reference delegates the broadphase to ODE - here, as in oracle/_stubs/ode.py, all pairs
i < j in body order) and the contact tuple format `((normal, p1, p2, penetration), i1, i2)` (:45, :203-204).

Two sources of history-dependence in the reference are fixed to their fresh-body values, which is what the
fixtures are generated with: the SAT warm start `last_sat_idx = 0` (bodies.py:171) and the GJK start vertex
(`random.choice`, contacts.py:87) = vertex 0.  Neither changes the result except on exact ties.

Parity status: pinned by `tests/golden/contacts_*.npz` (outputs of the unmodified reference on seeded random
pair configurations, `oracle/make_golden_contacts.py`) in `tests/test_contacts_oracle.py`.

A body is a dict: kind 'circle' {pos[2], rad} or 'hull' {pos[2], verts[nv,2]} with `verts` already rotated
into the world frame, relative to the body centre (that is what `Hull.verts` holds, bodies.py:211-214).
"""
import numpy as np


def left_orthogonal(v):
    """physics/utils.py:99-102."""
    return np.array([v[1], -v[0]])


def rotation_matrix(ang):
    """physics/utils.py:105-112."""
    s, c = np.sin(ang), np.cos(ang)
    return np.array([[c, -s], [s, c]])


def rect_verts(dims, rot):
    """Rect vertices (bodies.py:261-264) rotated by `rot` (bodies.py:278-283)."""
    half = np.asarray(dims, dtype=np.float64) / 2
    v0, v1 = half, half * np.array([-1.0, 1.0])
    R = rotation_matrix(rot)
    v0, v1 = R @ v0, R @ v1
    return np.stack([v0, v1, -v0, -v1])


def _support(points, direction):
    """contacts.py:207-217 (`>=`: the LAST maximiser wins)."""
    best, best_idx, best_norm = None, -1, -1.0
    for i, p in enumerate(points):
        cur = float(p @ direction)
        if cur >= best_norm:
            best, best_idx, best_norm = p, i, cur
    return best, best_idx


def _bary2(point, a, b):
    """contacts.py:334-340."""
    diff = b - a
    n = np.linalg.norm(diff)
    nd = diff / n
    return float((b - point) @ nd / n), float((point - a) @ nd / n)


def _bary3(point, a, b, c):
    """contacts.py:341-350."""
    M = np.array([[a[0], b[0], c[0]], [a[1], b[1], c[1]], [1.0, 1.0, 1.0]])
    uvw = np.linalg.inv(M) @ np.array([point[0], point[1], 1.0])
    return float(uvw[0]), float(uvw[1]), float(uvw[2])


def _closest(point, simplex):
    """contacts.py:295-330."""
    if len(simplex) == 1:
        return simplex[0], [0]
    if len(simplex) == 2:
        u, v = _bary2(point, simplex[0], simplex[1])
        if u <= 0:
            return simplex[1], [1]
        if v <= 0:
            return simplex[0], [0]
        return u * simplex[0] + v * simplex[1], [0, 1]
    uAB, vAB = _bary2(point, simplex[0], simplex[1])
    uBC, vBC = _bary2(point, simplex[1], simplex[2])
    uCA, vCA = _bary2(point, simplex[2], simplex[0])
    uABC, vABC, wABC = _bary3(point, simplex[0], simplex[1], simplex[2])
    if vAB <= 0 and uCA <= 0:
        return simplex[0], [0]
    if vBC <= 0 and uAB <= 0:
        return simplex[1], [1]
    if vCA <= 0 and uBC <= 0:
        return simplex[2], [2]
    if uAB > 0 and vAB > 0 and wABC <= 0:
        return uAB * simplex[0] + vAB * simplex[1], [0, 1]
    if uBC > 0 and vBC > 0 and uABC <= 0:
        return uBC * simplex[1] + vBC * simplex[2], [1, 2]
    if uCA > 0 and vCA > 0 and vABC <= 0:
        return uCA * simplex[2] + vCA * simplex[0], [2, 0]
    if uABC > 0 and vABC > 0 and wABC > 0:
        return point, [0, 1, 2]
    raise ValueError("point does not satisfy any condition in get_closest()")


def _circle_circle(b1, b2, eps):
    """contacts.py:68-79."""
    r = b1["rad"] + b2["rad"]
    normal = b1["pos"] - b2["pos"]
    dist = np.linalg.norm(normal)
    pen = r - dist
    if pen < -eps:
        return []
    normal = normal / dist
    p1 = -normal * (b1["rad"] - pen / 2)
    p2 = normal * (b2["rad"] - pen / 2)
    return [(normal, p1, p2, pen)]


def _circle_hull(circ, hull, eps, circle_is_g2):
    """contacts.py:80-141.  `circ` plays b1, `hull` b2 (the reference swaps when the circle is geom2)."""
    verts = hull["verts"]
    test_point = circ["pos"] - hull["pos"]
    simplex = [verts[0]]                                    # reference: random.choice(b2.verts)
    while True:
        closest, ids = _closest(test_point, simplex)
        if len(ids) == 3:
            break
        if len(ids) == 2:
            sd = left_orthogonal(simplex[ids[0]] - simplex[ids[1]])
            if float(sd @ (test_point - simplex[ids[0]])) < 0:
                sd = -sd
        else:
            sd = test_point - closest
        if sd[0] == 0 and sd[1] == 0:
            break
        support, _ = _support(verts, sd)
        if any(support is s or (support[0] == s[0] and support[1] == s[1]) for s in simplex):
            break
        simplex = [simplex[i] for i in ids]
        simplex.append(support)
    if len(ids) < 3:
        best_pt2 = closest
        cw = closest + hull["pos"]
        best_pt1 = cw - circ["pos"]
        best_dist = np.linalg.norm(cw - circ["pos"]) - circ["rad"]
        if best_dist > eps:
            return []
        best_normal = -best_pt1 / np.linalg.norm(best_pt1)
    else:                                                    # centre inside the hull: SAT (contacts.py:114-137)
        best_dist = -1e10
        nv = len(verts)
        best_normal = best_pt1 = best_pt2 = None
        for idx in range(nv):                                # start_edge = last_sat_idx = 0
            edge = verts[(idx + 1) % nv] - verts[idx]
            normal = left_orthogonal(edge) / np.linalg.norm(edge)
            center = circ["pos"] - hull["pos"]
            dist = float(normal @ (center - verts[idx])) - circ["rad"]
            if dist > best_dist:
                if dist > eps:
                    return []
                best_dist = dist
                best_normal = normal
                best_pt2 = center + normal * -(dist + circ["rad"])
                best_pt1 = best_pt2 + hull["pos"] - circ["pos"]
    if circle_is_g2:
        best_normal = -best_normal
        best_pt1, best_pt2 = best_pt2, best_pt1
    return [(best_normal, best_pt1, best_pt2, -best_dist)]


def _test_separations(h1, h2, eps):
    """contacts.py:220-250."""
    v1, v2 = h1["verts"], h2["verts"]
    nv = len(v1)
    best = dict(dist=-1e10, normal=None, vertex=-1, edge_norm=None, edge=0)
    for idx in range(nv):                                    # start_edge = last_sat_idx = 0
        edge = v1[(idx + 1) % nv] - v1[idx]
        edge_norm = np.linalg.norm(edge)
        normal = left_orthogonal(edge) / edge_norm
        sp, sidx = _support(v2, -normal)
        sp = sp + h2["pos"] - h1["pos"]
        dist = float(normal @ (sp - v1[idx]))
        if dist > best["dist"]:
            if dist > eps:
                return dict(dist=dist, normal=None, vertex=None, edge_norm=None, edge=idx)
            best = dict(dist=dist, normal=-normal, vertex=sidx, edge_norm=edge_norm, edge=idx)
    return best


def _incident_edge(ref_normal, inc_verts, inc_vertex):
    """contacts.py:253-268."""
    nv = len(inc_verts)
    min_dot, best_edge = 1e10, -1
    for i in ((inc_vertex - 1) % nv, inc_vertex):
        edge = inc_verts[(i + 1) % nv] - inc_verts[i]
        inc_normal = left_orthogonal(edge) / np.linalg.norm(edge)
        dot = float(ref_normal @ inc_normal)
        if dot < min_dot:
            min_dot, best_edge = dot, i
    return best_edge


def _clip(verts, normal, offset):
    """contacts.py:270-292."""
    out = []
    d0 = float(normal @ verts[0]) + offset
    d1 = float(normal @ verts[1]) + offset
    if d0 >= 0.0:
        out.append(verts[0])
    if d1 >= 0.0:
        out.append(verts[1])
    if d0 * d1 < 0.0 or len(out) < 2:
        interp = d0 / (d0 - d1)
        out.append(verts[0] + interp * (verts[1] - verts[0]))
    return out


def _hull_hull(b1, b2, eps):
    """contacts.py:142-201."""
    c1 = _test_separations(b1, b2, eps)
    if c1["dist"] > eps:
        return []
    c2 = _test_separations(b2, b1, eps)
    if c2["dist"] > eps:
        return []
    if c2["dist"] > c1["dist"]:
        ref, inc, c, flip = b2, b1, c2, False
    else:
        ref, inc, c, flip = b1, b2, c1, True
    normal = -c["normal"]
    half_edge = c["edge_norm"] / 2
    ie = _incident_edge(normal, inc["verts"], c["vertex"])
    iv = [inc["verts"][ie], inc["verts"][(ie + 1) % len(inc["verts"])]]
    iv = [v + inc["pos"] - ref["pos"] for v in iv]
    plane = left_orthogonal(normal)
    cl = _clip(iv, plane, half_edge)
    if len(cl) < 2:
        return []
    cl = _clip(cl, -plane, half_edge)
    pts = []
    for v in cl:
        dist = float(normal @ (v - ref["verts"][c["edge"]]))
        if dist <= eps:
            pt1 = v + normal * -dist
            pt2 = pt1 + ref["pos"] - inc["pos"]
            if flip:      # reference body is b1 (contacts.py:198-201)
                pts.append((-normal, pt1, pt2, -dist))
            else:         # reference body is b2 (contacts.py:170-175)
                pts.append((normal, pt2, pt1, -dist))
    return pts


def collide_pair(b1, b2, eps=0.1):
    """contacts.py:57-205 for one (geom1, geom2) pair; returns [(normal, p1, p2, penetration), ...]."""
    c1, c2 = b1["kind"] == "circle", b2["kind"] == "circle"
    if c1 and c2:
        return _circle_circle(b1, b2, eps)
    if c1:
        return _circle_hull(b1, b2, eps, circle_is_g2=False)
    if c2:
        return _circle_hull(b2, b1, eps, circle_is_g2=True)
    return _hull_hull(b1, b2, eps)


def find_contacts(bodies, eps=0.1, no_contact=()):
    """world.py:139-142 with an all-pairs broadphase (i < j in body order).
    Returns [((normal, p1, p2, penetration), i1, i2), ...] like `world.contacts`."""
    out = []
    nc = {frozenset(p) for p in no_contact}
    for i in range(len(bodies)):
        for j in range(i + 1, len(bodies)):
            if frozenset((i, j)) in nc:
                continue
            for pt in collide_pair(bodies[i], bodies[j], eps):
                out.append((pt, i, j))
    return out
