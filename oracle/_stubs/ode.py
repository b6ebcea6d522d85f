"""Minimal stand-in for py3ode (test infrastructure only; NOT product code).

The reference (`/root/reference/lcp_physics/physics/world.py:40-43,139-142`,
`bodies.py:129-132,197-200,273-276`) only uses ODE as a broadphase container:
geoms remember a pose, the space calls a callback for candidate pairs.  An
all-pairs broadphase is a superset of ODE's hash space, and the reference's
`DiffContactHandler` early-outs on separated pairs, so the generated contact
lists are identical up to ordering.  `ode.collide` (narrow phase) is only
needed by `OdeContactHandler`, which the default configuration never calls.
"""


class _Geom:
    def __init__(self, space=None, *args):
        self._pos = (0.0, 0.0, 0.0)
        self._quat = (1.0, 0.0, 0.0, 0.0)

    def setPosition(self, pos):
        self._pos = tuple(float(x) for x in pos)

    def getPosition(self):
        return self._pos

    def setQuaternion(self, quat):
        self._quat = tuple(float(x) for x in quat)

    def getQuaternion(self):
        return self._quat


class GeomSphere(_Geom):
    pass


class GeomBox(_Geom):
    pass


class HashSpace:
    def __init__(self):
        self._geoms = []

    def add(self, geom):
        self._geoms.append(geom)

    def collide(self, args, callback):
        n = len(self._geoms)
        for i in range(n):
            for j in range(i + 1, n):
                callback(args, self._geoms[i], self._geoms[j])


def collide(geom1, geom2):
    raise NotImplementedError("ode narrow phase is not available in the stub")
